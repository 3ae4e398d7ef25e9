"""BASELINE config 5 -- the 16385 x 16385 dense LP, 2.1 GB -- pinned by the oracle beyond its first pivots (GPU).

333 pivots per phase (tests/_c5.py: no multiple of the delay depth, 41 full depth-8 flushes, both sets of pending rows,
three launch boundaries of the persistent loop at 100 pivots per launch) through every kernel that runs this shape --
stream3_kernel<512,16,nt>, sweep_kernel<512,16,nt>, dshard_kernel<512,16,nt,panel> with one rank over RCCL and with two ranks
over the host transport -- each compared with oracle/liboracle_omp.so on the box's host cores: the whole tableau
(SHA-256 per block of 512 rows), both permutations, column 0, status, result, pivot count.  Two inputs: the LP as
generated (phase-2 pivots, src/simplex.ts:66-103) and an infeasible start with exact zeros (phase 1, :106-142).

The whole solve (83 270 pivots) is out of the oracle's reach (about an hour on 16 cores); its record --
tests/golden/c5_whole_solve.json, written by tools/c5_record.py from a stream3_kernel run -- is a HIP result that the
three kernels, which share no sweep code, have to reproduce bit for bit.
"""
import hashlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import _c5

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
RECORD = os.path.join(ROOT, "tests", "golden", "c5_whole_solve.json")


def _same(a, b):
    return a == b or (a != a and b != b)


def _unsharded(monkeypatch, variant, delay, kernel, budget, chunk=None):
    from yalps_amd import _native as nat
    monkeypatch.setenv("YALPS_HIP_DELAY", delay)
    if chunk:
        monkeypatch.setenv("YALPS_HIP_RESIDENT_CHUNK", str(chunk))
    ctx = nat.Context(0)
    t = nat.DeviceTableau(ctx, _c5.W, _c5.H)
    try:
        ident = np.arange(_c5.W + _c5.H, dtype=np.int32)
        t.upload(_c5.make_input(nat.dense_lp, variant), _c5.H, ident, ident.copy())
        status, result, npiv, _ = t.solve(max_pivots=budget)
        info = t.info()
        assert info["last_path"] == "inplace" and info["inplace"] == kernel, info
        got, gpos, gvar = t.download()
    finally:
        t.close()
        ctx.close()
    return status, result, npiv, got.reshape(_c5.H, _c5.W), gpos, gvar, info


@pytest.mark.parametrize("variant", ["phase2", "phase1"])
@pytest.mark.parametrize("delay,kernel", [("1", "stream3_kernel<512,16,nt>"), ("0", "sweep_kernel<512,16,nt>")], ids=["stream3", "sweep"])
def test_c5_persistent_kernels_against_the_oracle(monkeypatch, variant, delay, kernel):
    """One GPU, in place: 333 pivots per phase in launches of 100 -- stream3_kernel leaves every launch but the third with
    pivots pending (100 = 12 * 8 + 4) and the loop with five; sweep_kernel is the one-sweep-per-pivot form of the same loop."""
    ref = _c5.reference(variant)
    status, result, npiv, got, gpos, gvar, info = _unsharded(monkeypatch, variant, delay, kernel, float(_c5.BUDGET), _c5.CHUNK)
    assert (status, npiv) == (ref["status"], ref["pivots"]) and _same(result, ref["result"]), (status, result, npiv)
    assert npiv >= _c5.BUDGET and int(info["last_resident_launches"]) >= 4, info
    assert np.array_equal(gpos, ref["pos"]) and np.array_equal(gvar, ref["var"])
    assert np.array_equal(got[:, 0].view(np.int64), ref["ref"][:, 0].view(np.int64))
    assert _c5.check_digests(ref["ref"], _c5.digest_rows(got, 0)) == []


def _run_workers(kind, world, variant, budget, tmp_path, env_extra=None):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / ("c5_%s_%d_%s.npz" % (kind, world, variant)))
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
        procs.append(subprocess.Popen([sys.executable, "-m", "tests._shard_worker", kind, str(_c5.M), str(_c5.N), str(_c5.SEED), out,
                                       str(budget), "c5:" + variant], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=1500)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    if os.environ.get("YALPS_TEST_LOG_DIR"):
        with open(os.path.join(os.environ["YALPS_TEST_LOG_DIR"], "c5_%s_%d_%s.log" % (kind, world, variant)), "w") as f:
            f.write("\n".join(logs))
    return np.load(out)


# (two processes sharing one GPU take ~0.24 s per pivot at this size -- the card alternates between their contexts for every
# kernel --, so only one of the two-rank cases runs the full 333 pivots per phase; the others stop after 61 / 45, still between
# two sweeps of the depth-16 form and with both scratch parities used)
@pytest.mark.parametrize("kind,world,variant,budget", [("hip-rccl", 1, "phase2", _c5.BUDGET), ("hip-rccl", 1, "phase1", _c5.BUDGET),
                                                       ("hip-native", 2, "phase2", _c5.BUDGET), ("hip-native", 2, "phase1", 61),
                                                       ("hip", 2, "phase1", 45)])
def test_c5_row_shards_against_the_oracle(tmp_path, kind, world, variant, budget):
    """The row-sharded path with delayed row updates (dshard_kernel<512,16,nt,panel>, depth 16, the library's own loop): one rank
    holding all rows over RCCL (ncclAllGather between the kernels, batches of 64 pivots as hipGraph replays), and two ranks
    sharing the test GPU over the host transport (8192 rows each, the candidate rows travel with the pending pivots applied);
    the last case drives the same kernels from the Python loop over gloo (yalps_amd/sharded.py::sharded_simplex)."""
    ref = _c5.reference(variant, budget)
    res = _run_workers(kind, world, variant, budget, tmp_path)
    assert str(res["kernel"]).startswith("dshard_kernel<512,16,nt,panel>"), res["kernel"]
    assert (str(res["status"]), int(res["pivots"])) == (ref["status"], ref["pivots"]) and _same(float(res["result"]), ref["result"])
    assert np.array_equal(res["pos"], ref["pos"]) and np.array_equal(res["var"], ref["var"])
    assert np.array_equal(res["col0"].view(np.int64), ref["ref"][:, 0].view(np.int64))
    digests = list(zip(res["lo"].tolist(), res["hi"].tolist(), [str(x) for x in res["sha"]]))
    assert len(digests) >= world + (_c5.H - 1) // _c5.BLOCK and _c5.check_digests(ref["ref"], digests) == []


def record_of(status, result, npiv, m00, col0, pos, var):
    return {"status": status, "result": result, "pivots": int(npiv), "m00_hex": float(m00).hex(),
            "col0_sha256": hashlib.sha256(np.ascontiguousarray(col0).tobytes()).hexdigest(),
            "pos_sha256": hashlib.sha256(np.ascontiguousarray(pos).tobytes()).hexdigest(),
            "var_sha256": hashlib.sha256(np.ascontiguousarray(var).tobytes()).hexdigest()}


@pytest.mark.parametrize("delay,kernel", [("1", "stream3_kernel<512,16,nt>"), ("0", "sweep_kernel<512,16,nt>")], ids=["stream3", "sweep"])
def test_c5_whole_solve_record_persistent_kernels(monkeypatch, delay, kernel):
    rec = json.load(open(RECORD))
    status, result, npiv, got, gpos, gvar, _ = _unsharded(monkeypatch, "phase2", delay, kernel, float("inf"))
    assert record_of(status, result, npiv, got[0, 0], got[:, 0], gpos, gvar) == rec["record"]


def test_c5_whole_solve_record_row_shard_one_rank(tmp_path):
    rec = json.load(open(RECORD))["record"]
    res = _run_workers("hip-rccl", 1, "phase2", "inf", tmp_path)
    assert str(res["kernel"]).startswith("dshard_kernel<512,16,nt,panel>"), res["kernel"]
    got = record_of(str(res["status"]), float(res["result"]), int(res["pivots"]), res["col0"][0], res["col0"], res["pos"], res["var"])
    assert got == rec
