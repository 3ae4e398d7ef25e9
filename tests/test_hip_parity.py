"""GPU parity tests proper: the HIP path, called through the C ABI, against (a) the committed
golden records produced by the reference's own src/simplex.ts and (b) the CPU oracle on seeded
inputs.  Bit-exact: status, result, pivot count, permutations, RHS column, whole matrix."""
import numpy as np
import pytest

import os

from tests import _golden as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    from yalps_amd import _native
    assert _native.lib().yalps_device_count() >= 1, "no HIP device: the GPU tests need a real MI355X"
    return _native


@pytest.fixture(scope="module")
def ctx(nat):
    c = nat.Context(0)
    yield c
    c.close()


GOLDEN = [pytest.param(r, id=G.label(r)) for kind in ("cases", "mixed", "dense") for r in G.records(kind)]


@pytest.mark.parametrize("rec", GOLDEN)
def test_dropin_matches_reference_golden(nat, oracle, rec):
    """yalps_simplex_f64 (host arrays in/out) == reference simplex() on every golden record."""
    m = G.initial_matrix(rec, oracle, dense_gen=nat.dense_lp)
    pos, var = G.identity_perms(rec)
    exp = G.expected(rec)
    status, result, npiv = nat.simplex_host(m, rec["width"], rec["height"], pos, var, **G.options(rec))
    assert status == exp["status"]
    assert G.same_number(result, exp["result"])
    assert npiv == exp["n_pivots"]
    assert np.array_equal(pos, exp["pos"]) and np.array_equal(var, exp["var"])
    col0 = m.reshape(rec["height"], rec["width"])[:, 0]
    assert np.array_equal(col0.view(np.int64), exp["col0"].view(np.int64))
    assert G.sha256(m) == exp["final_sha256"]


def test_dense_generator_matches_oracle(nat, oracle):
    for M, N, seed in ((3, 5, 42), (64, 64, 7), (300, 200, 1)):
        assert np.array_equal(nat.dense_lp(M, N, seed), oracle.dense_lp(M, N, seed))


@pytest.mark.parametrize("shape", [(4, 3), (17, 130), (200, 513), (513, 200), (700, 1100)])
def test_single_pivot_matches_oracle(nat, ctx, oracle, shape):
    """One bare Gauss-Jordan pivot (src/simplex.ts:5-39) on random data with exact zeros and
    entries around the 1e-16 flush / skip threshold."""
    h, w = shape
    rng = np.random.default_rng(h * 1000 + w)
    m = rng.uniform(-1, 1, h * w)
    m[rng.random(h * w) < 0.3] = 0.0
    tiny = rng.random(h * w) < 0.05
    m[tiny] = rng.uniform(-2e-16, 2e-16, tiny.sum())
    m[rng.random(h * w) < 0.01] = -0.0
    row, col = int(rng.integers(1, h)), int(rng.integers(1, w))
    m[row * w + col] = 0.37
    pos = np.arange(w + h, dtype=np.int32)
    var = np.arange(w + h, dtype=np.int32)
    ref, rpos, rvar = m.copy(), pos.copy(), var.copy()
    oracle.pivot(ref, w, h, rpos, rvar, row, col)
    t = nat.DeviceTableau(ctx, w, h)
    try:
        t.upload(m, h, pos, var)
        t.pivot(row, col)
        got, gpos, gvar = t.download()
    finally:
        t.close()
    assert np.array_equal(got.view(np.int64), ref.view(np.int64))
    assert np.array_equal(gpos, rpos) and np.array_equal(gvar, rvar)


@pytest.mark.parametrize("M,N,seed", [(300, 300, 5), (150, 700, 11), (900, 400, 3)])
def test_device_solve_matches_oracle_dense(nat, ctx, oracle, M, N, seed):
    w, h = N + 1, M + 1
    m = nat.dense_lp(M, N, seed)
    pos = np.arange(w + h, dtype=np.int32)
    var = np.arange(w + h, dtype=np.int32)
    ref, rpos, rvar = m.copy(), pos.copy(), var.copy()
    est, eres, epiv, _ = oracle.simplex(ref, w, h, rpos, rvar, max_pivots=np.inf)
    t = nat.DeviceTableau(ctx, w, h)
    try:
        t.upload(m, h, pos, var)
        status, result, npiv, ms = t.solve(max_pivots=np.inf)
        got, gpos, gvar = t.download()
        rhs = t.download_rhs()
    finally:
        t.close()
    assert (status, npiv) == (est, epiv) and G.same_number(result, eres)
    assert np.array_equal(got.view(np.int64), ref.view(np.int64))
    assert np.array_equal(rhs.view(np.int64), ref.reshape(h, w)[:, 0].view(np.int64))
    assert np.array_equal(gpos, rpos) and np.array_equal(gvar, rvar)


def test_copyback_solution_only(nat, oracle):
    rec = next(r for r in G.records("dense") if r["M"] == 128)
    m = G.initial_matrix(rec, oracle, dense_gen=nat.dense_lp)
    init = m.copy()
    pos, var = G.identity_perms(rec)
    exp = G.expected(rec)
    status, result, npiv = nat.simplex_host(m, rec["width"], rec["height"], pos, var, copyback=nat.COPYBACK_SOLUTION,
                                            **G.options(rec))
    assert status == exp["status"] and result == exp["result"]
    mm = m.reshape(rec["height"], rec["width"])
    assert np.array_equal(mm[:, 0].view(np.int64), exp["col0"].view(np.int64))
    assert np.array_equal(mm[:, 1:], init.reshape(mm.shape)[:, 1:])  # untouched on the host
    assert np.array_equal(pos, exp["pos"]) and np.array_equal(var, exp["var"])


def test_bad_arguments_fail_loudly(nat, ctx):
    with pytest.raises(nat.NativeError):
        nat.DeviceTableau(ctx, 0, 4)
    t = nat.DeviceTableau(ctx, 4, 4)
    try:
        with pytest.raises(nat.NativeError):
            t.solve()  # nothing uploaded
        with pytest.raises(nat.NativeError):
            t.upload(np.zeros(4 * 9), 9, np.arange(13, dtype=np.int32), np.arange(13, dtype=np.int32))
    finally:
        t.close()


# ---- end to end: solve() with the HIP simplex behind it (reference test-suite semantics) --------
from tests import _cases as K  # noqa: E402


@pytest.mark.parametrize("name", K.names())
def test_solve_matches_reference_cases(nat, oracle, name):
    """solve(model, options) on the GPU reproduces the reference test-suite's expected status /
    objective / feasibility for every case (tests/solver.ts:23-25, tests/additional/json.ts), and
    equals the oracle-backed solve exactly (same host code, bit-identical simplex)."""
    from tests.test_host_model import oracle_backend
    from yalps_amd import solve as S
    case = K.load(name)
    sol = S.solve(case["model"], case["options"])
    assert K.valid_solution_and_status(sol, case["expected"], case["model"], case["options"]), sol["status"]
    ref = S._solve_with(oracle_backend(oracle), case["model"], case["options"])
    assert sol["status"] == ref["status"] and G.same_number(sol["result"], ref["result"])
    assert sol["variables"] == ref["variables"]


def test_solve_readme_example(nat):
    from yalps_amd import model as M
    from yalps_amd import solve as S
    model = {"direction": "maximize", "objective": "profit",
             "constraints": {"wood": M.less_eq(300), "labor": M.less_eq(110), "storage": M.less_eq(400)},
             "variables": {"table": {"wood": 30, "labor": 5, "profit": 1200, "storage": 30},
                           "dresser": {"wood": 20, "labor": 10, "profit": 1600, "storage": 50}},
             "integers": ["table", "dresser"]}
    assert S.solve(model) == {"status": "optimal", "result": 14400.0, "variables": [("table", 8.0), ("dresser", 3.0)]}


def test_resident_fallback_resumes_with_streaming_kernel(oracle, tmp_path):
    """If the resident kernel ever reports a failed hand-off, the solve continues with the next
    path (the in-place kernel) from the last consistent state.  Forced here after two chunks of 40 pivots
    (in a child process: the switches are read when the context is created)."""
    import subprocess
    import sys
    rec = next(r for r in G.records("dense") if r["M"] == 256)
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "from tests import _golden as G, _oracle\n"
        "from yalps_amd import _native as n\n"
        "rec = next(r for r in G.records('dense') if r['M'] == 256)\n"
        "m = G.initial_matrix(rec, _oracle.load(), dense_gen=n.dense_lp); pos, var = G.identity_perms(rec)\n"
        "ctx = n.Context(0); t = n.DeviceTableau(ctx, rec['width'], rec['height']); t.upload(m, rec['height'], pos, var)\n"
        "st, res, piv, _ = t.solve(max_pivots=float('inf')); info = t.info(); gm, gp, gv = t.download()\n"
        "exp = G.expected(rec)\n"
        "assert info['last_path'] == 'resident+inplace', info\n"
        "assert (st, res, piv) == (exp['status'], exp['result'], exp['n_pivots']), (st, res, piv)\n"
        "assert G.sha256(gm) == exp['final_sha256'] and np.array_equal(gp, exp['pos']) and np.array_equal(gv, exp['var'])\n"
        "print('ok')\n" % ROOT)
    env = dict(__import__("os").environ, YALPS_HIP_RESIDENT_CHUNK="40", YALPS_HIP_RESIDENT_FAULT="2")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


def test_resident_chunking_and_repeatability(nat, ctx, oracle):
    """Several resident launches per solve (state carried across launches) and 10 repeated solves
    of the 1025 x 1025 case: every run must reproduce the reference bit for bit."""
    rec = next(r for r in G.records("dense") if r["M"] == 1024)
    exp = G.expected(rec)
    m0 = G.initial_matrix(rec, oracle, dense_gen=nat.dense_lp)
    pos0, var0 = G.identity_perms(rec)
    t = nat.DeviceTableau(ctx, rec["width"], rec["height"])
    try:
        for _ in range(10):
            t.upload(m0, rec["height"], pos0, var0)
            st, res, piv, _ = t.solve(max_pivots=float("inf"))
            assert (st, res, piv) == (exp["status"], exp["result"], exp["n_pivots"])
            gm, gp, gv = t.download()
            assert G.sha256(gm) == exp["final_sha256"] and np.array_equal(gp, exp["pos"])
        assert t.info()["last_path"] == "resident"
    finally:
        t.close()


# ---- on-device tableau assembly -------------------------------------------
def _cells(m, w):
    idx = np.flatnonzero(m.view(np.int64) != 0)  # every cell that is not +0.0 (a -0.0 cell is kept: bit-exact)
    return (idx // w).astype(np.int32), (idx % w).astype(np.int32), m[idx]


@pytest.mark.parametrize("rec", [pytest.param(r, id=G.label(r)) for r in G.records("cases")])
def test_assemble_then_solve_matches_reference_golden(nat, ctx, oracle, rec):
    """yalps_tableau_assemble builds the reference's initial tableau in HBM from its non-zero cells
    (bit-identical matrix, identity permutations), and yalps_simplex_sparse_f64 returns the
    reference's status / result / pivot count / column 0 / permutations."""
    w, h = rec["width"], rec["height"]
    m = G.initial_matrix(rec, oracle)
    row, col, val = _cells(m, w)
    t = nat.DeviceTableau(ctx, w, h + 3)  # spare capacity: stale rows beyond `height` must not matter
    try:
        junk = np.full(w * (h + 3), 7.5)
        t.upload(junk, h + 3, np.arange(w + h + 3, dtype=np.int32)[::-1].copy(), np.arange(w + h + 3, dtype=np.int32)[::-1].copy())
        t.assemble(h, row, col, val)
        got, pos, var = t.download()
    finally:
        t.close()
    assert t.capacity == h + 3 and got.size == w * h
    assert np.array_equal(got.view(np.int64), m.view(np.int64))
    assert np.array_equal(pos, np.arange(w + h)) and np.array_equal(var, np.arange(w + h))
    exp = G.expected(rec)
    status, result, npiv, col0, pos, var = nat.simplex_sparse(w, h, row, col, val, **G.options(rec))
    assert (status, npiv) == (exp["status"], exp["n_pivots"]) and G.same_number(result, exp["result"])
    assert np.array_equal(pos, exp["pos"]) and np.array_equal(var, exp["var"])
    assert np.array_equal(col0.view(np.int64), exp["col0"].view(np.int64))


def test_assemble_rejects_bad_cells(nat, ctx):
    t = nat.DeviceTableau(ctx, 4, 3)
    i32, f64 = lambda *a: np.array(a, np.int32), lambda *a: np.array(a, np.float64)
    try:
        for row, col in ((i32(1, 1), i32(2, 2)), (i32(2, 1), i32(0, 0)), (i32(0, 3), i32(0, 0)), (i32(0, 0), i32(1, 4)),
                         (i32(-1, 0), i32(0, 1))):
            with pytest.raises(nat.NativeError):
                t.assemble(3, row, col, f64(1, 2))
        t.assemble(3, i32(), i32(), f64())  # no cells: the all-zero tableau
        got, pos, var = t.download()
        assert not got.any() and pos.tolist() == list(range(7))
    finally:
        t.close()


def test_solve_sparse_equals_dense_on_netlib(nat):
    """solve(): the sparse route (cells up, column 0 + permutations back) returns the same Solution
    as the dense host tableau through yalps_simplex_f64."""
    from yalps_amd import mps, solve as S
    for b in mps.read_benchmarks(os.path.join(G.GOLDEN, "netlib")):
        if b["name"] not in ("SC105", "SC205", "AGG2", "SCTAP1", "SCAGR7", "BEACONFD", "KLEIN2"):
            continue
        a = S.solve(b["model"], b["options"], sparse=True)
        d = S.solve(b["model"], b["options"], sparse=False)
        assert a["status"] == d["status"] and G.same_number(a["result"], d["result"]) and a["variables"] == d["variables"]


# ---- small tableaux: one workgroup, tableau in LDS (small_kernel) --------------------------------
@pytest.mark.parametrize("M,N,seed", [(3, 2, 1), (35, 100, 2), (99, 149, 3), (60, 290, 4), (1100, 14, 5), (2, 2000, 6)])
def test_small_path_matches_oracle(nat, ctx, oracle, M, N, seed):
    """Both variants of small_kernel (256 / 1024 lanes), through the resident-tableau API (HBM
    layout) and through the host-array drop-in (reference layout in pinned host memory)."""
    w, h = N + 1, M + 1
    m = nat.dense_lp(M, N, seed)
    m[np.random.default_rng(seed).random(m.size) < 0.2] = 0.0
    m[h // 2 * w:(h // 2 + 1) * w] *= -1.0  # a row "-a x <= -b": the start is infeasible, phase 1 runs too
    pos, var = np.arange(w + h, dtype=np.int32), np.arange(w + h, dtype=np.int32)
    ref, rpos, rvar = m.copy(), pos.copy(), var.copy()
    est, eres, epiv, _ = oracle.simplex(ref, w, h, rpos, rvar, max_pivots=np.inf)
    t = nat.DeviceTableau(ctx, w, h)
    try:
        t.upload(m, h, pos, var)
        status, result, npiv, ms = t.solve(max_pivots=np.inf)
        assert t.info()["last_path"] == "small"
        got, gpos, gvar = t.download()
    finally:
        t.close()
    assert (status, npiv) == (est, epiv) and G.same_number(result, eres)
    assert np.array_equal(got.view(np.int64), ref.view(np.int64))
    assert np.array_equal(gpos, rpos) and np.array_equal(gvar, rvar)
    hm, hpos, hvar = m.copy(), pos.copy(), var.copy()
    status, result, npiv = nat.simplex_host(hm, w, h, hpos, hvar, max_pivots=np.inf)
    assert (status, npiv) == (est, epiv) and G.same_number(result, eres)
    assert np.array_equal(hm.view(np.int64), ref.view(np.int64))
    assert np.array_equal(hpos, rpos) and np.array_equal(hvar, rvar)
    # maxPivots exhausted -> "cycled" with the tableau as it is after that many pivots
    if epiv > 3:
        ref2, rpos2, rvar2 = m.copy(), pos.copy(), var.copy()
        est2, _, epiv2, _ = oracle.simplex(ref2, w, h, rpos2, rvar2, max_pivots=3)
        hm, hpos, hvar = m.copy(), pos.copy(), var.copy()
        status, result, npiv = nat.simplex_host(hm, w, h, hpos, hvar, max_pivots=3)
        assert (status, npiv) == (est2, epiv2) and result != result
        assert np.array_equal(hm.view(np.int64), ref2.view(np.int64)) and np.array_equal(hpos, rpos2)


def test_small_path_boundary(nat, ctx):
    """Just under / just over the LDS budget: the second one must take the multi-workgroup path."""
    for (w, h), path in (((101, 180), "small"), ((101, 184), "resident")):
        m = nat.dense_lp(h - 1, w - 1, 9)
        t = nat.DeviceTableau(ctx, w, h)
        try:
            t.upload(m, h, np.arange(w + h, dtype=np.int32), np.arange(w + h, dtype=np.int32))
            assert t.solve(max_pivots=np.inf)[0] == "optimal" and t.info()["last_path"] == path
        finally:
            t.close()


@pytest.mark.parametrize("tag", ["1", "0"], ids=["tagged-rows", "second-generation-loop"])
def test_golden_cases_without_small_path(nat, oracle, monkeypatch, tag):
    """The multi-workgroup kernels (resident, streaming) on the small golden tableaux too: the same
    records as test_dropin_matches_reference_golden on a context created with YALPS_HIP_SMALL=0; once with the tagged
    candidate rows these shapes take by default, once with YALPS_HIP_TAG=0 through resident2_kernel (sparse rows with
    entries around the 1e-16 flush: its non-zero-mask path, phase 1 and checkCycles included)."""
    monkeypatch.setenv("YALPS_HIP_SMALL", "0")
    monkeypatch.setenv("YALPS_HIP_TAG", tag)
    c = nat.Context(0)
    paths = set()
    try:
        for rec in G.records("cases") + G.records("mixed"):
            m = G.initial_matrix(rec, oracle)
            pos, var = G.identity_perms(rec)
            exp = G.expected(rec)
            t = nat.DeviceTableau(c, rec["width"], rec["height"])
            try:
                t.upload(m, rec["height"], pos, var)
                st, res, piv, _ = t.solve(**G.options(rec))
                paths.add(t.info()["last_path"])
                gm, gp, gv = t.download()
            finally:
                t.close()
            assert (st, piv) == (exp["status"], exp["n_pivots"]) and G.same_number(res, exp["result"]), G.label(rec)
            assert G.sha256(gm) == exp["final_sha256"] and np.array_equal(gp, exp["pos"]) and np.array_equal(gv, exp["var"])
    finally:
        c.close()
    assert paths == {"resident"}, paths  # (the checkCycles records too: hasCycle verdicts travel with one more exchange)


def test_small_path_check_cycles_history_growth(nat, oracle, monkeypatch):
    """checkCycles on the single-workgroup path: hasCycle runs on the device; when a phase outgrows
    the history buffer the kernel leaves the tableau untouched and the host reruns it with a larger
    one (forced here with a first buffer of 8 entries)."""
    monkeypatch.setenv("YALPS_HIP_SMALL_HIST", "8")
    c = nat.Context(0)
    try:
        for rec in G.records("cases") + G.records("mixed"):
            if not rec["options"]["checkCycles"]:
                continue
            m = G.initial_matrix(rec, oracle)
            pos, var = G.identity_perms(rec)
            exp = G.expected(rec)
            t = nat.DeviceTableau(c, rec["width"], rec["height"])
            try:
                t.upload(m, rec["height"], pos, var)
                st, res, piv, _ = t.solve(**G.options(rec))
                assert t.info()["last_path"] == "small"
                gm, gp, gv = t.download()
            finally:
                t.close()
            assert (st, piv) == (exp["status"], exp["n_pivots"]) and G.same_number(res, exp["result"]), G.label(rec)
            assert G.sha256(gm) == exp["final_sha256"] and np.array_equal(gp, exp["pos"]) and np.array_equal(gv, exp["var"])
    finally:
        c.close()


@pytest.mark.parametrize("M,N,kernel", [(600, 2500, "resident_kernel<512,3,4>"), (1400, 2700, "resident_kernel<512,3,6>"),
                                        (700, 4300, "resident_kernel<512,5,4>"), (4500, 200, "resident_kernel<512,1,24>"),
                                        (5000, 700, "resident_kernel<512,1,24>"), (1300, 4300, "resident_kernel<512,5,6>"), (7000, 300, "resident_kernel<512,1,32>"), (9000, 250, "resident_kernel<512,1,40>"),
                                        (2600, 2600, "resident_kernel<512,3,12>"), (900, 5800, "resident_kernel<512,6,4>"),
                                        (2100, 2480, "resident_kernel<512,3,9>")])
@pytest.mark.parametrize("gen", ["1", "2"])
def test_resident_variants_match_oracle(nat, ctx, oracle, monkeypatch, M, N, kernel, gen):
    """The odd-J and tall variants of the resident kernel, 200 pivots each against the oracle
    (maxPivots exhausted -> "cycled" with the tableau as it stands), bit for bit; gen 2: with the second-generation pivot
    loop (resident2_kernel) where that shape has one, the first generation elsewhere."""
    monkeypatch.setenv("YALPS_HIP_RESIDENT_GEN", gen)  # (read when the tableau is created)
    w, h = N + 1, M + 1
    m = nat.dense_lp(M, N, 7)
    m[(h // 3) * w:(h // 3 + 1) * w] *= -1.0  # a row "-a x <= -b": phase 1 first
    pos, var = np.arange(w + h, dtype=np.int32), np.arange(w + h, dtype=np.int32)
    ref, rpos, rvar = m.copy(), pos.copy(), var.copy()
    est, eres, epiv, _ = oracle.simplex(ref, w, h, rpos, rvar, max_pivots=200)
    assert epiv >= 200
    t = nat.DeviceTableau(ctx, w, h)
    try:
        t.upload(m, h, pos, var)
        status, result, npiv, _ = t.solve(max_pivots=200)
        info = t.info()
        got, gpos, gvar = t.download()
    finally:
        t.close()
    names = (kernel,) if gen == "1" else (kernel, kernel.replace("resident_kernel", "resident2_kernel"))
    assert info["last_path"] == "resident" and info["resident"].startswith(names), info
    assert (status, npiv) == (est, epiv) and G.same_number(result, eres)
    assert np.array_equal(got.view(np.int64), ref.view(np.int64))
    assert np.array_equal(gpos, rpos) and np.array_equal(gvar, rvar)


# ---- resident kernel with rows parked in LDS (tableaux a little beyond the register files) --------
@pytest.mark.parametrize("M,N,kernel,lds_rows", [
    (2800, 3300, "resident_kernel<512,4,7,lds>", 4), (11000, 900, "resident_kernel<512,1,38,lds>", 5),
    (3072, 3072, "resident_kernel<512,3,11,lds>", 2), (4300, 2040, "resident_kernel<512,2,16,lds>", 1),
    (1700, 4300, "resident_kernel<512,5,5,lds>", 2), (1200, 5800, "resident_kernel<512,6,3,lds>", 2)])
def test_resident_lds_rows_match_restatement(nat, ctx, M, N, kernel, lds_rows):
    """Tableaux with more rows per workgroup than any register variant holds: the missing rows live in LDS.
    100 pivots (phase 1 first, exact zeros in the data) against the pinned numpy restatement, bit for bit."""
    from tests import _np_simplex as NP
    w, h = N + 1, M + 1
    m = nat.dense_lp(M, N, 13)
    m.reshape(h, w)[h // 3] *= -1.0
    m.reshape(h, w)[5::7, 3::5] = 0.0
    pos, var = np.arange(w + h, dtype=np.int32), np.arange(w + h, dtype=np.int32)
    ref, rpos, rvar = m.copy(), pos.copy(), var.copy()
    est, eres, epiv = NP.simplex(ref, w, h, rpos, rvar, max_pivots=100)
    t = nat.DeviceTableau(ctx, w, h)
    try:
        t.upload(m, h, pos, var)
        status, result, npiv, _ = t.solve(max_pivots=100)
        info = t.info()
        got, gpos, gvar = t.download()
    finally:
        t.close()
    assert info["last_path"] == "resident" and info["resident"].startswith(kernel), info
    assert info["lds_rows"] == str(lds_rows), info
    assert (status, npiv) == (est, epiv) and G.same_number(result, eres)
    assert np.array_equal(gpos, rpos) and np.array_equal(gvar, rvar)
    assert np.array_equal(got.view(np.int64), ref.view(np.int64))


@pytest.mark.parametrize("variant,M,N", [("512,2,16", 256 * 17 + 40, 60), ("512,1,38", 256 * 42 - 3, 33), ("512,3,11", 256 * 19 - 1, 70),
                                         ("512,6,3", 256 * 10, 25), ("512,4,7", 256 * 8 + 1, 48)])
@pytest.mark.parametrize("check", [False, True])
def test_resident_lds_rows_whole_solves(nat, ctx, oracle, monkeypatch, variant, M, N, check):
    """Every LDS-row variant forced onto tall, narrow LPs (cheap for the oracle): whole solves -- phase 1, phase 2, the
    pivot row / the candidate row / the entering column falling on parked rows many times -- with and without hasCycle."""
    monkeypatch.setenv("YALPS_HIP_RVARIANT", variant)
    w, h = N + 1, M + 1
    m = nat.dense_lp(M, N, 5)
    mm = m.reshape(h, w)
    mm[1::3, 0] *= -0.05  # negative right-hand sides: phase 1 runs for a while
    mm[1::3, 1:] *= -1.0
    mm[2::5, 2::3] = 0.0
    pos, var = np.arange(w + h, dtype=np.int32), np.arange(w + h, dtype=np.int32)
    ref, rpos, rvar = m.copy(), pos.copy(), var.copy()
    est, eres, epiv, _ = oracle.simplex(ref, w, h, rpos, rvar, max_pivots=3000, check_cycles=check)
    t = nat.DeviceTableau(ctx, w, h)
    try:
        t.upload(m, h, pos, var)
        status, result, npiv, _ = t.solve(max_pivots=3000, check_cycles=check)
        info = t.info()
        got, gpos, gvar = t.download()
    finally:
        t.close()
    assert info["last_path"] == "resident" and info["resident"].startswith("resident_kernel<%s,lds>" % variant), info
    assert (status, npiv) == (est, epiv) and G.same_number(result, eres)
    assert np.array_equal(gpos, rpos) and np.array_equal(gvar, rvar)
    assert np.array_equal(got.view(np.int64), ref.view(np.int64))


# ---- persistent in-place kernels (stream2_kernel: two pivots per sweep; stream_kernel) for tableaux beyond the on-chip size ----
@pytest.mark.parametrize("delay", ["3", "2", "0"], ids=["delayed", "stream2", "stream"])
@pytest.mark.parametrize("M,N,pivots", [(2800, 3300, 120), (1400, 8000, 81), (12000, 1500, 60), (11000, 900, 61), (12000, 400, 60),
                                        (2800, 3300, 1), (1400, 8000, 2), (2800, 3300, 3), (4400, 5000, 37)])
def test_inplace_path_matches_restatement(nat, ctx, monkeypatch, M, N, pivots, delay):
    """Dense tableaux that do not fit the register-resident kernel: `pivots` pivots (phase 1 first) through the kernels with
    delayed row updates (the rows get several pivots' eliminations per sweep; budgets that are no multiple of the depth
    leave through a shorter flush) -- the default choice (stream3_kernel), stream2_kernel wherever it applies -- and through stream_kernel, against the pinned numpy restatement, bit for bit."""
    from tests import _np_simplex as NP
    monkeypatch.setenv("YALPS_HIP_DELAY", "0" if delay == "0" else "1")
    if delay == "2":
        monkeypatch.setenv("YALPS_HIP_DELAY_KERNEL", delay)
    monkeypatch.setenv("YALPS_HIP_SWEEP", "0")  # (the 8001-column shape: stream_kernel<1024,4>, not sweep_kernel)
    monkeypatch.setenv("YALPS_HIP_LDS_ROWS", "0")  # (two of the shapes would fit with rows parked in LDS)
    w, h = N + 1, M + 1
    m = nat.dense_lp(M, N, 11)
    m.reshape(h, w)[h // 3] *= -1.0
    m.reshape(h, w)[5::7, 3::5] = 0.0  # exact zeros: untouched rows / flushed columns
    pos, var = np.arange(w + h, dtype=np.int32), np.arange(w + h, dtype=np.int32)
    ref, rpos, rvar = m.copy(), pos.copy(), var.copy()
    est, eres, epiv = NP.simplex(ref, w, h, rpos, rvar, max_pivots=pivots)
    t = nat.DeviceTableau(ctx, w, h)
    try:
        t.upload(m, h, pos, var)
        status, result, npiv, _ = t.solve(max_pivots=pivots)
        info = t.info()
        got, gpos, gvar = t.download()
    finally:
        t.close()
    want = "stream_kernel" if delay == "0" else "stream2_kernel" if delay == "2" else "stream3_kernel"
    assert info["last_path"] == "inplace" and info["inplace"].startswith(want), info
    assert (status, npiv) == (est, epiv) and G.same_number(result, eres)
    assert np.array_equal(gpos, rpos) and np.array_equal(gvar, rvar)
    assert np.array_equal(got.view(np.int64), ref.view(np.int64))


@pytest.mark.parametrize("panel", ["1", "0"], ids=["panels", "from-L2"])
@pytest.mark.parametrize("M,N,pivots", [(6000, 2060, 41), (3100, 4102, 37)])
def test_sweep_tail_units_and_last_columns(nat, ctx, monkeypatch, oracle, M, N, pivots, panel):
    """The sweep of the delayed kernels (panel_flush.cuh) takes the units behind the last full 1024-column panel through a routine
    of their own (a lane per row and unit) -- rows of 2^k + 1 columns have one such unit.  Here the widths leave 8 and 4 tail units,
    the objective makes the LAST columns enter first (the pivot-column patch, src/simplex.ts:25,36, lands in the tail; pending pivot
    rows have flushed entries there, :17-24), one row starts infeasible: whole tableau, basis and status against the oracle."""
    monkeypatch.setenv("YALPS_HIP_LDS_ROWS", "0")  # (99 and 102 MB: beyond the register files either way)
    monkeypatch.setenv("YALPS_HIP_STREAM3_PANEL", panel)
    w, h = N + 1, M + 1
    m = nat.dense_lp(M, N, 23)
    A = m.reshape(h, w)
    A[0, w - 14:] *= 60.0          # Dantzig pricing (:71-79) takes the last columns first
    A[h // 4] *= -1.0              # an infeasible start: phase 1 (:106-142)
    A[3::5, w - 9::2] = 0.0        # exact zeros in the tail columns: flushed entries of pending pivot rows, untouched rows
    pos, var = np.arange(w + h, dtype=np.int32), np.arange(w + h, dtype=np.int32)
    ref, rpos, rvar = m.copy(), pos.copy(), var.copy()
    est, eres, epiv, trace = oracle.simplex(ref, w, h, rpos, rvar, max_pivots=float(pivots), trace_cap=256)
    assert (trace[:epiv, 1] >= w - 16).sum() >= 4, trace[:epiv]  # (the case does what it is for: pivot columns in the tail)
    t = nat.DeviceTableau(ctx, w, h)
    try:
        t.upload(m, h, pos, var)
        status, result, npiv, _ = t.solve(max_pivots=float(pivots))
        info = t.info()
        got, gpos, gvar = t.download()
    finally:
        t.close()
    assert info["last_path"] == "inplace" and info["inplace"].startswith("stream3_kernel") and info["sweep"] == ("panels" if panel == "1" else "direct"), info
    assert (status, npiv) == (est, epiv) and G.same_number(result, eres)
    assert np.array_equal(gpos, rpos) and np.array_equal(gvar, rvar)
    assert np.array_equal(got.view(np.int64), ref.view(np.int64))


def test_inplace_path_sparse_netlib_whole_solve(nat, ctx, oracle):
    """SHIP12L (2197 x 5428, 95 MB, 0.4 % of the rows touched per pivot): the whole solve in place,
    every bit of the final tableau against the oracle."""
    from yalps_amd import mps, model as M
    b = next(x for x in mps.read_benchmarks(os.path.join(G.GOLDEN, "netlib")) if x["name"] == "SHIP12L")
    t0 = M.tableau_model(b["model"]).tableau
    ref, rpos, rvar = t0.matrix.copy(), t0.position_of_variable.copy(), t0.variable_at_position.copy()
    est, eres, epiv, _ = oracle.simplex(ref, t0.width, t0.height, rpos, rvar, max_pivots=np.inf)
    t = nat.DeviceTableau(ctx, t0.width, t0.height)
    try:
        t.upload(t0.matrix, t0.height, t0.position_of_variable, t0.variable_at_position)
        status, result, npiv, _ = t.solve(max_pivots=np.inf)
        assert t.info()["last_path"] == "inplace"
        got, gpos, gvar = t.download()
    finally:
        t.close()
    assert (status, npiv) == (est, epiv) and G.same_number(result, eres)
    assert np.array_equal(gpos, rpos) and np.array_equal(gvar, rvar)
    assert np.array_equal(got.view(np.int64), ref.view(np.int64))


@pytest.mark.parametrize("M,N,check", [(300, 200, False), (1000, 500, False), (477, 54, True), (64, 512, False)])
def test_resident_tagged_rows_equal_the_flag_protocol(nat, ctx, oracle, monkeypatch, M, N, check):
    """resident_kernel<256,1,4,tag> (candidate rows as self-validating granules, the default for these shapes) against the
    oracle and against the same variant with the drain + flag hand-off (YALPS_HIP_TAG=0): whole solves with a long phase 1."""
    monkeypatch.setenv("YALPS_HIP_SMALL", "0")
    w, h = N + 1, M + 1
    m = nat.dense_lp(M, N, 9)
    mm = m.reshape(h, w)
    mm[1::3, 0] *= -0.05
    mm[1::3, 1:] *= -1.0
    mm[2::5, 2::3] = 0.0
    pos = np.arange(w + h, dtype=np.int32)
    ref, rp, rv = m.copy(), pos.copy(), pos.copy()
    est, eres, epiv, _ = oracle.simplex(ref, w, h, rp, rv, max_pivots=5000, check_cycles=check)
    for tag, gen, name in (("1", "2", "resident_kernel<256,1,4,tag>"), ("0", "1", "resident_kernel<256,1,4>"),
                           ("0", "2", "resident2_kernel<256,1,4>")):  # (gen 2: the second-generation pivot loop, resident2_kernel.cuh)
        monkeypatch.setenv("YALPS_HIP_TAG", tag)
        monkeypatch.setenv("YALPS_HIP_RESIDENT_GEN", gen)
        c = nat.Context(0)
        t = nat.DeviceTableau(c, w, h)
        try:
            t.upload(m, h, pos, pos.copy())
            st, res, piv, _ = t.solve(max_pivots=5000, check_cycles=check)
            info = t.info()
            gm, gp, gv = t.download()
        finally:
            t.close()
            c.close()
        assert info["last_path"] == "resident" and info["resident"] == name, info
        assert (st, piv) == (est, epiv) and G.same_number(res, eres)
        assert np.array_equal(gm.view(np.int64), ref.view(np.int64)) and np.array_equal(gp, rp) and np.array_equal(gv, rv)


def test_two_contexts_solve_concurrently_from_two_threads(nat, oracle):
    """Two contexts on one device, one thread each (ctypes releases the GIL during a solve): whole-chip persistent
    launches take turns inside the library, nobody falls off the resident path, both get the oracle's answer."""
    import threading
    M = 1024
    w = h = M + 1
    m = nat.dense_lp(M, M, 42)
    pos = np.arange(w + h, dtype=np.int32)
    ref, rp, rv = m.copy(), pos.copy(), pos.copy()
    est, eres, epiv, _ = oracle.simplex(ref, w, h, rp, rv, max_pivots=np.inf)
    out, errors = {}, []

    def work(k):
        try:
            c = nat.Context(0)
            t = nat.DeviceTableau(c, w, h)
            try:
                res = []
                for _ in range(4):
                    t.upload(m, h, pos, pos.copy())
                    st, r, piv, _ms = t.solve(max_pivots=np.inf)
                    res.append((st, r, piv, t.info()["last_path"]))
                out[k] = (res, t.download())
            finally:
                t.close()
                c.close()
        except Exception as e:  # noqa: BLE001 (reported by the main thread)
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [x.start() for x in threads]
    [x.join() for x in threads]
    assert not errors, errors
    for k in range(2):
        res, (gm, gp, gv) = out[k]
        assert all(r[0] == est and G.same_number(r[1], eres) and r[2] == epiv and r[3] == "resident" for r in res), res
        assert np.array_equal(gm.view(np.int64), ref.view(np.int64)) and np.array_equal(gp, rp) and np.array_equal(gv, rv)


def test_two_tableaux_share_a_kernel_with_different_lds_needs(nat, ctx):
    """The >48 KB dynamic-LDS permission belongs to the kernel function, not to a tableau: A (65.6 KB of LDS in
    stream_kernel<1024,4>), then B (49.7 KB, same function), then A again -- the second tableau must not lower what the
    first one is allowed to launch with."""
    from tests import _np_simplex as NP
    runs = []
    tabs = []
    try:
        for M, N in ((256, 8192), (299, 6200)):
            w, h = N + 1, M + 1
            m = nat.dense_lp(M, N, 3)
            pos = np.arange(w + h, dtype=np.int32)
            t = nat.DeviceTableau(ctx, w, h)
            tabs.append((t, m, w, h, pos))
        for k in (0, 1, 0):
            t, m, w, h, pos = tabs[k]
            t.upload(m, h, pos, pos.copy())
            st, res, piv, _ = t.solve(max_pivots=20)
            assert t.info()["last_path"] == "inplace" and t.info()["inplace"] == "stream_kernel<1024,4>", t.info()  # (two rows per workgroup: one sweep per pivot)
            runs.append((k, st, res, piv, t.download()))
    finally:
        for t, *_ in tabs:
            t.close()
    for k, st, res, piv, (gm, gp, gv) in runs:
        _, m, w, h, pos = tabs[k]
        ref, rp, rv = m.copy(), pos.copy(), pos.copy()
        est, eres, epiv = NP.simplex(ref, w, h, rp, rv, max_pivots=20)
        assert (st, piv) == (est, epiv) and G.same_number(res, eres)
        assert np.array_equal(gm.view(np.int64), ref.view(np.int64)) and np.array_equal(gp, rp) and np.array_equal(gv, rv)


@pytest.mark.parametrize("lds_rows,path", [("0", "inplace+streaming"), ("1", "resident+inplace")])
def test_inplace_fallback_restores_the_tableau(oracle, lds_rows, path):
    """A failed hand-off leaves the in-place tableau half updated: the host restores the copy it made before
    the launch and continues with the launch-per-pivot kernels (forced after two chunks of 30 pivots).  With rows
    parked in LDS the same tableau is resident: its forced failure continues in place from the untouched buffer."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "from tests import _np_simplex as NP\n"
        "from yalps_amd import _native as n\n"
        "M, N = 2800, 3300; w, h = N + 1, M + 1\n"
        "m = n.dense_lp(M, N, 3); pos = np.arange(w + h, dtype=np.int32); var = pos.copy()\n"
        "ref, rp, rv = m.copy(), pos.copy(), var.copy(); e = NP.simplex(ref, w, h, rp, rv, max_pivots=100)\n"
        "ctx = n.Context(0); t = n.DeviceTableau(ctx, w, h); t.upload(m, h, pos, var)\n"
        "st, res, piv, _ = t.solve(max_pivots=100); info = t.info(); gm, gp, gv = t.download()\n"
        "assert info['last_path'] == %r, info\n"
        "assert (st, piv) == (e[0], e[2]), (st, piv, e)\n"
        "assert np.array_equal(gm.view(np.int64), ref.view(np.int64)) and np.array_equal(gp, rp) and np.array_equal(gv, rv)\n"
        "print('ok')\n" % (ROOT, path))
    env = dict(os.environ, YALPS_HIP_RESIDENT_CHUNK="30", YALPS_HIP_RESIDENT_FAULT="2", YALPS_HIP_LDS_ROWS=lds_rows)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


TALL = [("inplace", "0"), ("resident", "1")]  # 11001 rows: stream_kernel, or resident_kernel<512,1,38> + 5 rows per workgroup in LDS


@pytest.mark.parametrize("path,lds_rows,delay", [("inplace", "0", "3"), ("inplace", "0", "2"), ("inplace", "0", "0"), ("resident", "1", "3")],
                         ids=["stream3", "stream2", "stream", "resident-lds"])
@pytest.mark.parametrize("kind", ["unbounded", "infeasible", "optimal", "optimal-degenerate"])
def test_inplace_path_terminal_statuses(nat, ctx, monkeypatch, kind, path, lds_rows, delay):
    """The persistent kernels' exits other than the pivot budget, on a tall narrow LP (11001 x 61: beyond the register
    variants) solved to the end through stream3_kernel and stream2_kernel (which may leave with pivots still pending: the
    flush), through stream_kernel and through the resident kernel with LDS rows: unbounded after 124 pivots (result = the column),
    infeasible after 41 phase-1 pivots, optimal, and optimal with every 10th right-hand side zero (ties, ratios <= precision)."""
    from tests import _np_simplex as NP
    monkeypatch.setenv("YALPS_HIP_LDS_ROWS", lds_rows)
    monkeypatch.setenv("YALPS_HIP_DELAY", "0" if delay == "0" else "1")
    monkeypatch.setenv("YALPS_HIP_DELAY_KERNEL", "2" if delay == "2" else "3")
    M, N = 11000, 60
    w, h = N + 1, M + 1
    m = nat.dense_lp(M, N, 21)
    A = m.reshape(h, w)
    if kind == "unbounded":
        A[0, 7] = 0.9          # a good reduced cost ...
        A[1:, 7] = -A[1:, 7]   # ... on a column nothing bounds
    elif kind == "infeasible":
        A[h // 2] = -A[h // 2]  # "a x >= 1e6" against thousands of rows "a' x <= ~20"
        A[h // 2, 0] = -1e6
    elif kind == "optimal-degenerate":
        A[1::10, 0] = 0.0
    pos, var = np.arange(w + h, dtype=np.int32), np.arange(w + h, dtype=np.int32)
    ref, rpos, rvar = m.copy(), pos.copy(), var.copy()
    est, eres, epiv = NP.simplex(ref, w, h, rpos, rvar, max_pivots=np.inf)
    assert est == kind.split("-")[0] and (epiv > 20 or kind == "optimal-degenerate"), (est, epiv)
    t = nat.DeviceTableau(ctx, w, h)
    try:
        t.upload(m, h, pos, var)
        status, result, npiv, _ = t.solve(max_pivots=np.inf)
        assert t.info()["last_path"] == path, t.info()
        got, gpos, gvar = t.download()
    finally:
        t.close()
    assert (status, npiv) == (est, epiv) and G.same_number(result, eres)
    assert np.array_equal(gpos, rpos) and np.array_equal(gvar, rvar)
    assert np.array_equal(got.view(np.int64), ref.view(np.int64))


@pytest.mark.parametrize("path,lds_rows,delay", [("inplace", "0", "1"), ("inplace", "0", "0"), ("resident", "1", "1")],
                         ids=["stream3-check", "stream-check", "resident-lds"])
def test_inplace_path_check_cycles(nat, ctx, oracle, monkeypatch, path, lds_rows, delay):
    """options.checkCycles through stream3_kernel<.., true> (delayed row updates: a cycle verdict leaves with pivots
    pending), stream_kernel<.., true> and the resident kernel with LDS rows: the tall narrow LP solved to optimality with
    the verdict exchange in every pivot (no cycle), and a Chvatal-style cycling LP embedded in a tall tableau (rows of
    zeros below it) that must stop "cycled" at the same pivot as the oracle."""
    monkeypatch.setenv("YALPS_HIP_LDS_ROWS", lds_rows)
    monkeypatch.setenv("YALPS_HIP_DELAY", delay)
    M, N = 11000, 60
    w, h = N + 1, M + 1
    m = nat.dense_lp(M, N, 21)
    rec = next(r for r in G.records("cases") if r["name"] == "Chvatal Cycling")
    small = G.initial_matrix(rec, oracle).reshape(rec["height"], rec["width"])
    big = np.zeros((11001, rec["width"]))
    big[:rec["height"]] = small
    for matrix, width, height, opts in ((m, w, h, dict(precision=1e-8, max_pivots=np.inf, check_cycles=True)),
                                        (big.reshape(-1), rec["width"], 11001, dict(G.options(rec)))):
        pos, var = np.arange(width + height, dtype=np.int32), np.arange(width + height, dtype=np.int32)
        ref, rpos, rvar = matrix.copy(), pos.copy(), var.copy()
        est, eres, epiv, _ = oracle.simplex(ref, width, height, rpos, rvar, **opts)
        t = nat.DeviceTableau(ctx, width, height)
        try:
            t.upload(matrix, height, pos, var)
            status, result, npiv, _ = t.solve(**opts)
            assert t.info()["last_path"] == path, t.info()
            if path == "inplace":
                assert t.info()["inplace"].startswith("stream3_kernel" if delay == "1" else "stream_kernel"), t.info()
            got, gpos, gvar = t.download()
        finally:
            t.close()
        assert (status, npiv) == (est, epiv) and G.same_number(result, eres)
        assert np.array_equal(gpos, rpos) and np.array_equal(gvar, rvar)
        assert np.array_equal(got.view(np.int64), ref.view(np.int64))
    assert est == "cycled"


@pytest.mark.parametrize("M,N,delay", [(1400, 8000, "0"), (300, 16000, "0"), (1400, 8000, "1"), (300, 16000, "1")])
def test_check_cycles_on_wide_tableaux_launch_per_pivot(nat, ctx, monkeypatch, M, N, delay):
    """checkCycles beyond the on-chip size.  With delayed updates switched off no persistent kernel applies by
    default (4098+ columns): DECIDE launches of pivot_kernel<1024,4,..> + APPLY launches of wide_kernel, or
    (8194+ columns) the any-shape pair.  With them on (the default) stream3_kernel<..,true> takes the shapes
    that have 4+ rows per workgroup (1401 rows here; 301 rows stay on the launch-per-pivot path).  40 pivots
    against the numpy restatement (which
    has no hasCycle: no cycle can close within 40 pivots of these LPs, the check only has to stay silent)."""
    from tests import _np_simplex as NP
    monkeypatch.setenv("YALPS_HIP_DELAY", delay)
    w, h = N + 1, M + 1
    m = nat.dense_lp(M, N, 13)
    m.reshape(h, w)[h // 3] *= -1.0
    pos, var = np.arange(w + h, dtype=np.int32), np.arange(w + h, dtype=np.int32)
    ref, rpos, rvar = m.copy(), pos.copy(), var.copy()
    est, eres, epiv = NP.simplex(ref, w, h, rpos, rvar, max_pivots=40)
    t = nat.DeviceTableau(ctx, w, h)
    try:
        t.upload(m, h, pos, var)
        status, result, npiv, _ = t.solve(max_pivots=40, check_cycles=True)
        if delay == "1" and M >= 1024:
            assert t.info()["last_path"] == "inplace" and t.info()["inplace"] == "stream3_kernel<512,8>", t.info()
        else:
            assert t.info()["last_path"] == ("generic" if N > 8192 else "streaming"), t.info()
        got, gpos, gvar = t.download()
    finally:
        t.close()
    assert (status, npiv) == (est, epiv) and G.same_number(result, eres)
    assert np.array_equal(gpos, rpos) and np.array_equal(gvar, rvar)
    assert np.array_equal(got.view(np.int64), ref.view(np.int64))


@pytest.mark.parametrize("M,N,kernel", [(4000, 2000, "pivot_kernel<1024,1,16>"), (4000, 500, "pivot_kernel<256,1,16>"),
                                        (2000, 1000, "pivot_kernel<256,2,8>")])
def test_launch_per_pivot_fallback_variants(nat, oracle, monkeypatch, M, N, kernel):
    """The launch-per-pivot kernels a tableau falls back to when the persistent ones are unavailable (context
    created with YALPS_HIP_RESIDENT=0 / YALPS_HIP_INPLACE=0), including the register-heaviest pivot_kernel
    variants: 60 pivots against the numpy restatement."""
    from tests import _np_simplex as NP
    monkeypatch.setenv("YALPS_HIP_RESIDENT", "0")
    monkeypatch.setenv("YALPS_HIP_INPLACE", "0")
    c = nat.Context(0)
    w, h = N + 1, M + 1
    m = nat.dense_lp(M, N, 17)
    m.reshape(h, w)[h // 3] *= -1.0
    pos, var = np.arange(w + h, dtype=np.int32), np.arange(w + h, dtype=np.int32)
    ref, rpos, rvar = m.copy(), pos.copy(), var.copy()
    est, eres, epiv = NP.simplex(ref, w, h, rpos, rvar, max_pivots=60)
    t = nat.DeviceTableau(c, w, h)
    try:
        t.upload(m, h, pos, var)
        status, result, npiv, _ = t.solve(max_pivots=60)
        info = t.info()
        got, gpos, gvar = t.download()
    finally:
        t.close()
        c.close()
    assert info["last_path"] == "streaming" and info["streaming"] == kernel, info
    assert (status, npiv) == (est, epiv) and G.same_number(result, eres)
    assert np.array_equal(gpos, rpos) and np.array_equal(gvar, rvar)
    assert np.array_equal(got.view(np.int64), ref.view(np.int64))


SPILLING = [  # M, N, checkCycles, env, streaming kernel, DECIDE kernel -- the seven launch-per-pivot instantiations with scratch
    (2000, 2000, False, {}, "pivot_kernel<1024,1,9>", None),          # 12 B (bench.py's streaming_apply_only kernel)
    (4000, 2000, False, {}, "pivot_kernel<1024,1,16>", None),         # 200 B
    (4000, 500, False, {}, "pivot_kernel<256,1,16>", None),           # 20 B
    (1500, 3000, True, {}, "wide_kernel<1024,2>", "pivot_kernel<1024,2,8>"),   # 348 B, DECIDE launches only
    (900, 7000, True, {}, "wide_kernel<1024,4>", "pivot_kernel<1024,4,4>"),    # 452 B, DECIDE launches only
    (300, 12000, True, {"YALPS_HIP_WIDE8": "1"}, "wide_kernel<1024,8>", "pivot_kernel<1024,8,2>"),  # 868 B DECIDE; wide_kernel<1024,8> 32 B
    (300, 12000, False, {"YALPS_HIP_WIDE8": "1"}, "wide_kernel<1024,8>", None),
]


@pytest.mark.parametrize("M,N,check,env,kernel,decide", SPILLING)
def test_every_spilling_launch_per_pivot_instantiation(nat, monkeypatch, M, N, check, env, kernel, decide):
    """the register gate (yalps_amd/build.py NO_SCRATCH) exempts the launch-per-pivot family; seven of its
    instantiations use scratch (12 - 868 bytes per lane).  Each of them by name: 48 pivots with a phase-1 start and exact
    zeros against the oracle, every bit of the tableau (checkCycles where the instantiation only runs as the DECIDE launch)."""
    from tests import _oracle
    orc = _oracle.load(omp=True)
    orc.set_threads(8)
    monkeypatch.setenv("YALPS_HIP_RESIDENT", "0")
    monkeypatch.setenv("YALPS_HIP_INPLACE", "0")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    c = nat.Context(0)
    w, h = N + 1, M + 1
    m = nat.dense_lp(M, N, 23)
    A = m.reshape(h, w)
    A[h // 3] *= -1.0
    A[5::7, 3::5] = 0.0
    pos, var = np.arange(w + h, dtype=np.int32), np.arange(w + h, dtype=np.int32)
    ref, rpos, rvar = m.copy(), pos.copy(), var.copy()
    est, eres, epiv, _ = orc.simplex(ref, w, h, rpos, rvar, max_pivots=48.0, check_cycles=check)
    t = nat.DeviceTableau(c, w, h)
    try:
        t.upload(m, h, pos, var)
        status, result, npiv, _ = t.solve(max_pivots=48, check_cycles=check)
        info = t.info()
        got, gpos, gvar = t.download()
    finally:
        t.close()
        c.close()
    assert info["last_path"] == "streaming" and info["streaming"] == kernel, info
    assert decide is None or info["decide"] == decide, info
    assert (status, npiv) == (est, epiv) and G.same_number(result, eres)
    assert np.array_equal(gpos, rpos) and np.array_equal(gvar, rvar)
    assert np.array_equal(got.view(np.int64), ref.view(np.int64))


def test_golden_cases_through_the_inplace_kernel(nat, oracle, monkeypatch):
    """Every golden record of the reference (all statuses, checkCycles, odd precisions / maxPivots) through
    stream_kernel: a context without the single-workgroup and the register-resident paths."""
    monkeypatch.setenv("YALPS_HIP_SMALL", "0")
    monkeypatch.setenv("YALPS_HIP_RESIDENT", "0")
    c = nat.Context(0)
    paths = set()
    try:
        for rec in G.records("cases") + G.records("mixed") + [r for r in G.records("dense") if r["M"] <= 256]:
            m = G.initial_matrix(rec, oracle, dense_gen=nat.dense_lp)
            pos, var = G.identity_perms(rec)
            exp = G.expected(rec)
            t = nat.DeviceTableau(c, rec["width"], rec["height"])
            try:
                t.upload(m, rec["height"], pos, var)
                st, res, piv, _ = t.solve(**G.options(rec))
                paths.add(t.info()["last_path"])
                gm, gp, gv = t.download()
            finally:
                t.close()
            assert (st, piv) == (exp["status"], exp["n_pivots"]) and G.same_number(res, exp["result"]), G.label(rec)
            assert G.sha256(gm) == exp["final_sha256"] and np.array_equal(gp, exp["pos"]) and np.array_equal(gv, exp["var"])
    finally:
        c.close()
    assert paths == {"inplace"}, paths


@pytest.mark.parametrize("path,env", [("small", {}), ("resident", {"YALPS_HIP_SMALL": "0"}),
                                      ("resident", {"YALPS_HIP_SMALL": "0", "YALPS_HIP_TAG": "0"}),  # resident2_kernel for the narrow ones too
                                      ("inplace", {"YALPS_HIP_SMALL": "0", "YALPS_HIP_RESIDENT": "0"}),
                                      ("streaming", {"YALPS_HIP_SMALL": "0", "YALPS_HIP_RESIDENT": "0", "YALPS_HIP_INPLACE": "0"})])
def test_degenerate_integer_lps_on_every_path(nat, oracle, monkeypatch, path, env):
    """Seeded random tableaux with small integer entries (ties everywhere: equal reduced costs, equal ratios,
    zero right-hand sides, exact zeros) and random options, on each of the four device paths: the lowest-index
    tie-breaks, the ratio <= precision early exit and the 1e-16 rules must hold across workgroup boundaries."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    c = nat.Context(0)
    rng = np.random.default_rng(20240607)
    seen = set()
    try:
        for case in range(70 if path == "small" else 84):
            big = case >= 70  # several rows per workgroup (not for the single-workgroup path: they exceed LDS)
            h, w = (int(rng.integers(300, 1200)), int(rng.integers(40, 500))) if big else (int(rng.integers(2, 70)), int(rng.integers(2, 90)))
            m = rng.integers(-3, 4, size=(h, w)).astype(np.float64)
            m[rng.random((h, w)) < rng.choice([0.0, 0.3, 0.7])] = 0.0
            m[1:, 0] = rng.integers(-1 if case % 3 == 0 else 0, 5, size=h - 1)  # mostly feasible starts, many zeros
            m[0, 0] = 0.0
            if case % 5 == 0:
                m *= 0.5
            m = m.reshape(-1)
            opts = dict(precision=float(rng.choice([1e-8, 1e-6, 1e-12])), max_pivots=float(rng.choice([3000, 7, 60])),  # (finite: these LPs may cycle)
                        check_cycles=bool(rng.integers(0, 2)))
            pos, var = np.arange(w + h, dtype=np.int32), np.arange(w + h, dtype=np.int32)
            ref, rpos, rvar = m.copy(), pos.copy(), var.copy()
            est, eres, epiv, _ = oracle.simplex(ref, w, h, rpos, rvar, **opts)
            seen.add(est)
            t = nat.DeviceTableau(c, w, h)
            try:
                t.upload(m, h, pos, var)
                status, result, npiv, _ = t.solve(**opts)
                assert t.info()["last_path"] == path, (case, t.info())
                got, gpos, gvar = t.download()
            finally:
                t.close()
            assert (status, npiv) == (est, epiv) and G.same_number(result, eres), (case, h, w, opts, status, npiv, est, epiv)
            assert np.array_equal(gpos, rpos) and np.array_equal(gvar, rvar), case
            assert np.array_equal(got.view(np.int64), ref.view(np.int64)), case
    finally:
        c.close()
    assert {"optimal", "unbounded"} <= seen, seen


# ---- sweep_kernel: persistent, in place, for what streams from HBM (8194 .. 16385 columns; 4098 .. 8193 beyond the cache) ----
SWEEP = [  # M, N, pivots, env, expected kernel, checkCycles
    (600, 16000, 70, {"YALPS_HIP_DELAY": "0"}, "sweep_kernel<512,16>", False),
    (300, 9000, 60, {"YALPS_HIP_SWEEP_NT": "1", "YALPS_HIP_DELAY": "0"}, "sweep_kernel<512,16,nt>", False),
    (2100, 12345, 50, {"YALPS_HIP_DELAY": "0"}, "sweep_kernel<512,16>", False),
    # stream3_kernel (delayed updates for 8194 .. 16385 columns: objective replica in LDS, pending rows in a global scratch)
    (600, 16000, 71, {"YALPS_HIP_DELAY_MIN_ROWS": "1"}, "stream3_kernel<512,16>", False),
    (300, 9000, 60, {"YALPS_HIP_DELAY_NT": "1", "YALPS_HIP_DELAY_MIN_ROWS": "1"}, "stream3_kernel<512,16,nt>", False),
    (2100, 12345, 51, {"YALPS_HIP_DELAY_DEPTH": "4"}, "stream3_kernel<512,16>", False),
    (2100, 12345, 7, {"YALPS_HIP_DELAY_DEPTH": "3"}, "stream3_kernel<512,16>", False),
    (1400, 8000, 41, {"YALPS_HIP_DELAY_NT": "1"}, "stream3_kernel<512,8,nt>", False),
    (4300, 4000, 33, {}, "stream3_kernel<512,4>", False),
    (4400, 5500, 29, {"YALPS_HIP_DELAY_DEPTH": "5"}, "stream3_kernel<512,6>", False),
    (1400, 8000, 80, {"YALPS_HIP_SWEEP": "2", "YALPS_HIP_DELAY": "0"}, "sweep_kernel<512,8>", False),
    (2500, 5000, 60, {"YALPS_HIP_SWEEP": "2", "YALPS_HIP_SWEEP_NT": "1", "YALPS_HIP_DELAY": "0"}, "sweep_kernel<512,8,nt>", False),
    (900, 7000, 60, {"YALPS_HIP_SWEEP": "2", "YALPS_HIP_DELAY": "0"}, "sweep_kernel<512,8>", True),
    (900, 7000, 61, {}, "stream3_kernel<512,8>", True),
    (2100, 12345, 23, {}, "stream3_kernel<512,16>", True),
    # stream2_kernel (two pivots per sweep) with the same awkward data; its non-temporal forms; odd budgets
    (1400, 8000, 81, {"YALPS_HIP_DELAY_NT": "1", "YALPS_HIP_DELAY_KERNEL": "2"}, "stream2_kernel<512,8,nt>", False),
    (4300, 4000, 61, {"YALPS_HIP_DELAY_NT": "1", "YALPS_HIP_DELAY_KERNEL": "2"}, "stream2_kernel<1024,2,nt>", False),
    (900, 7000, 60, {"YALPS_HIP_DELAY_KERNEL": "2"}, "stream2_kernel<512,8>", False),
]


@pytest.mark.parametrize("M,N,pivots,env,kernel,check", SWEEP)
def test_sweep_kernel_matches_restatement(nat, ctx, monkeypatch, M, N, pivots, env, kernel, check):
    """sweep_kernel on dense tableaux with a phase-1 start, exact zeros (untouched rows, flushed pivot-row entries: the
    non-zero-mask path) and -- in one case -- checkCycles: `pivots` pivots against the pinned numpy restatement, every
    bit of the tableau, the basis and the pivot count; both cache policies, both unit counts (the switches are read when
    the tableau is created)."""
    from tests import _np_simplex as NP
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    w, h = N + 1, M + 1
    m = nat.dense_lp(M, N, 17)
    A = m.reshape(h, w)
    A[h // 3] *= -1.0          # "-a x <= -b": phase 1 first
    A[5::7, 3::5] = 0.0        # exact zeros
    A[2::9, 0] = 0.0           # degenerate rows (ratio <= precision)
    pos, var = np.arange(w + h, dtype=np.int32), np.arange(w + h, dtype=np.int32)
    ref, rpos, rvar = m.copy(), pos.copy(), var.copy()
    est, eres, epiv = NP.simplex(ref, w, h, rpos, rvar, max_pivots=pivots)
    t = nat.DeviceTableau(ctx, w, h)
    try:
        t.upload(m, h, pos, var)
        status, result, npiv, _ = t.solve(max_pivots=pivots, check_cycles=check)
        info = t.info()
        got, gpos, gvar = t.download()
    finally:
        t.close()
    assert info["last_path"] == "inplace" and info["inplace"] == kernel, info
    assert (status, npiv) == (est, epiv) and G.same_number(result, eres)
    assert np.array_equal(gpos, rpos) and np.array_equal(gvar, rvar)
    assert np.array_equal(got.view(np.int64), ref.view(np.int64))


@pytest.mark.parametrize("delay", ["1", "0"], ids=["stream3", "sweep"])
@pytest.mark.parametrize("kind", ["optimal", "unbounded", "infeasible"])
def test_sweep_kernel_whole_solves(nat, ctx, oracle, monkeypatch, kind, delay):
    """sweep_kernel to the end of a solve (120 x 9001: few, very wide rows): optimal, unbounded (result = the column) and
    infeasible, against the oracle bit for bit."""
    monkeypatch.setenv("YALPS_HIP_DELAY", delay)
    monkeypatch.setenv("YALPS_HIP_DELAY_MIN_ROWS", "1")  # (one row per workgroup here: stream3_kernel by request only)
    M, N = 120, 9000
    w, h = N + 1, M + 1
    m = nat.dense_lp(M, N, 23)
    A = m.reshape(h, w)
    if kind == "unbounded":
        A[0, 7] = 0.99
        A[1:, 7] = -A[1:, 7]
    elif kind == "infeasible":
        A[h // 2] = -A[h // 2]
        A[h // 2, 0] = -1e7
    pos, var = np.arange(w + h, dtype=np.int32), np.arange(w + h, dtype=np.int32)
    ref, rpos, rvar = m.copy(), pos.copy(), var.copy()
    est, eres, epiv, _ = oracle.simplex(ref, w, h, rpos, rvar, max_pivots=np.inf)
    assert est == kind, (est, epiv)
    t = nat.DeviceTableau(ctx, w, h)
    try:
        t.upload(m, h, pos, var)
        status, result, npiv, _ = t.solve(max_pivots=np.inf)
        info = t.info()
        got, gpos, gvar = t.download()
    finally:
        t.close()
    assert info["last_path"] == "inplace" and info["inplace"].startswith("stream3_kernel<512,16" if delay == "1" else "sweep_kernel<512,16"), info
    assert (status, npiv) == (est, epiv) and G.same_number(result, eres)
    assert np.array_equal(gpos, rpos) and np.array_equal(gvar, rvar)
    assert np.array_equal(got.view(np.int64), ref.view(np.int64))


# ---- rows wider than 16385 columns: the any-shape DECIDE + APPLY pair (generic_kernels.cuh) ---------------
@pytest.mark.parametrize("M,N,pivots,check", [(200, 20000, 90, False), (60, 17000, 5000, False), (400, 33000, 60, True)])
def test_generic_path_for_very_wide_tableaux(nat, ctx, M, N, pivots, check):
    """The reference has no width limit; the tuned kernels stop at 16385 columns.  Beyond that every pivot is one
    single-workgroup DECIDE launch + one in-place APPLY launch, bit-exact against the numpy restatement (phase 1,
    exact zeros, a whole solve to optimality, and checkCycles staying silent)."""
    from tests import _np_simplex as NP
    w, h = N + 1, M + 1
    m = nat.dense_lp(M, N, 31)
    A = m.reshape(h, w)
    if pivots < 5000:  # (the whole-solve case stays a plain LP: it reaches its optimum in 230 pivots)
        A[h // 3] *= -1.0
        A[3::5, 7::11] = 0.0
    pos, var = np.arange(w + h, dtype=np.int32), np.arange(w + h, dtype=np.int32)
    ref, rpos, rvar = m.copy(), pos.copy(), var.copy()
    est, eres, epiv = NP.simplex(ref, w, h, rpos, rvar, max_pivots=pivots)
    assert est == ("optimal" if pivots == 5000 else "cycled")
    t = nat.DeviceTableau(ctx, w, h)
    try:
        t.upload(m, h, pos, var)
        status, result, npiv, _ = t.solve(max_pivots=pivots, check_cycles=check)
        assert t.info()["last_path"] == "generic", t.info()
        got, gpos, gvar = t.download()
    finally:
        t.close()
    assert (status, npiv) == (est, epiv) and G.same_number(result, eres)
    assert np.array_equal(gpos, rpos) and np.array_equal(gvar, rvar)
    assert np.array_equal(got.view(np.int64), ref.view(np.int64))
    if pivots == 5000:  # and through the host-array drop-in
        hm, hp, hv = m.copy(), pos.copy(), var.copy()
        status, result, npiv = nat.simplex_host(hm, w, h, hp, hv, max_pivots=5000)
        assert (status, npiv) == (est, epiv) and np.array_equal(hm.view(np.int64), ref.view(np.int64)) and np.array_equal(hp, rpos)


def test_golden_cases_through_the_generic_pair(nat, oracle, monkeypatch):
    """Every small golden record of the reference (all statuses, "cycled" by hasCycle, odd precisions / maxPivots)
    through generic_decide_kernel + generic_apply_kernel (forced with YALPS_HIP_GENERIC=1, read at tableau creation)."""
    monkeypatch.setenv("YALPS_HIP_SMALL", "0")
    monkeypatch.setenv("YALPS_HIP_GENERIC", "1")
    c = nat.Context(0)
    try:
        for rec in G.records("cases") + G.records("mixed") + [r for r in G.records("dense") if r["M"] <= 128]:
            m = G.initial_matrix(rec, oracle, dense_gen=nat.dense_lp)
            pos, var = G.identity_perms(rec)
            exp = G.expected(rec)
            t = nat.DeviceTableau(c, rec["width"], rec["height"])
            try:
                t.upload(m, rec["height"], pos, var)
                st, res, piv, _ = t.solve(**G.options(rec))
                assert t.info()["last_path"] == "generic"
                gm, gp, gv = t.download()
            finally:
                t.close()
            assert (st, piv) == (exp["status"], exp["n_pivots"]) and G.same_number(res, exp["result"]), G.label(rec)
            assert G.sha256(gm) == exp["final_sha256"] and np.array_equal(gp, exp["pos"]) and np.array_equal(gv, exp["var"])
    finally:
        c.close()
