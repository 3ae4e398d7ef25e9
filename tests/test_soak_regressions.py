"""What the randomised soaks (tools/soak_delay.py, tools/soak_dshard.py) found, as deterministic tests (GPU).

Round 2's soak of the delayed-update kernels hit ONE wrong workgroup in 1 147 random cases: sparse tableaux (2 % dense:
most pivot-row entries flushed, most rows untouched), 16 units per lane, depth 8 -- the scratch of pending pivot rows
was shared by all XCDs with write-back stores, and a dirty line of an earlier generation in another XCD's L2 came back
over the fresh row.  stream3_kernel has one scratch per XCD since; dshard_kernel keeps ONE (`d.dpend`) because a kernel
boundary lies between the launch that stores a pending row and the launches that read it.  These cases are that family,
seeded, through both kernels, compared with the oracle bit for bit; they also check that the padding behind the device
rows stays finite (the select-free path used to multiply padding lanes by the FLUSHED marker).
"""
import numpy as np
import pytest

from tests import _golden as G

pytestmark = pytest.mark.gpu


def sparse_tableau(seed, h, w, density, flip, degenerate):
    """tools/soak_delay.py's generator: uniform(-1, 1) entries kept with probability `density`, right-hand sides positive
    or (flip) of random sign -- a phase-1 start --, every seventh right-hand side 0 (degenerate) on request."""
    rng = np.random.default_rng(seed)
    m = rng.uniform(-1, 1, (h, w))
    m[rng.random((h, w)) > density] = 0.0
    m[1:, 0] = np.abs(m[1:, 0]) * (rng.choice([-1, 1], h - 1) if flip else 1)
    if degenerate:
        m[1::7, 0] = 0.0
    m[0, 0] = 0.0
    return m.reshape(-1)


CASES = [  # seed, h, w, density, phase-1 start, degenerate rows, pivot budget
    (101, 1100, 16385, 0.02, False, False, 77),
    (102, 2500, 12345, 0.02, True, False, 77),
    (103, 1800, 9000, 0.02, False, True, 40),
    (104, 2048, 16384, 0.02, True, True, 77),
    (105, 1300, 14001, 0.1, True, False, 41),
    (106, 2599, 8200, 0.02, False, False, 17),
]


@pytest.fixture(scope="module")
def omp_oracle():
    from tests import _oracle
    orc = _oracle.load(omp=True)
    orc.set_threads(8)
    return orc


def _expect(omp_oracle, case):
    seed, h, w, density, flip, degenerate, budget = case
    m = sparse_tableau(seed, h, w, density, flip, degenerate)
    pos = np.arange(w + h, dtype=np.int32)
    var = pos.copy()
    ref, rpos, rvar = m.copy(), pos.copy(), var.copy()
    est, eres, epiv, _ = omp_oracle.simplex(ref, w, h, rpos, rvar, max_pivots=float(budget))
    return m, pos, var, ref, rpos, rvar, est, eres, epiv


@pytest.mark.parametrize("nt", ["0", "1"], ids=["plain", "nt"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "seed%d-%dx%d" % c[:3])
def test_sparse_family_through_stream3(monkeypatch, omp_oracle, case, nt):
    from yalps_amd import _native as nat
    monkeypatch.setenv("YALPS_HIP_DELAY_DEPTH", "8")
    monkeypatch.setenv("YALPS_HIP_DELAY_NT", nt)
    seed, h, w, *_ , budget = case
    m, pos, var, ref, rpos, rvar, est, eres, epiv = _expect(omp_oracle, case)
    ctx = nat.Context(0)
    t = nat.DeviceTableau(ctx, w, h)
    try:
        t.upload(m, h, pos, var)
        status, result, npiv, _ = t.solve(max_pivots=float(budget))
        info = t.info()
        got, gpos, gvar = t.download()
        pad = t.padding_check()
    finally:
        t.close()
        ctx.close()
    assert info["last_path"] == "inplace" and info["inplace"] == "stream3_kernel<512,16%s>" % (",nt" if nt == "1" else "") and info["delay_depth"] == "8", info
    assert (status, npiv) == (est, epiv) and G.same_number(result, eres), (status, npiv, est, epiv)
    assert np.array_equal(gpos, rpos) and np.array_equal(gvar, rvar)
    assert np.array_equal(got.view(np.int64), ref.view(np.int64))
    assert pad == (0, 0), pad


DSHARD_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
import torch
torch.cuda.init()  # (torch's HIP runtime first, as in every multi-rank launch: the other order leaves torch without a GPU)
from yalps_amd import sharded
m = np.load(%(inp)r)
h, w, budget = %(h)d, %(w)d, %(budget)d
pos = np.arange(w + h, dtype=np.int32)
ops = sharded.HipShardOps(m, w, sharded.partition(h, 1), 0, h, pos, pos.copy(), device=0)
status, result, npiv = sharded.sharded_simplex(ops, sharded.TorchComm(), max_pivots=float(budget), check_every=8)
kernel = ops.tab.info()["streaming"]
got, gpos, gvar = ops.download()
pad = ops.tab.padding_check()
ops.close()
np.savez(%(out)r, matrix=got, pos=gpos, var=gvar, status=status, result=result, pivots=npiv, kernel=kernel, pad=np.array(pad))
print("ok")
"""


@pytest.mark.parametrize("panel", ["0", "1"], ids=["direct", "panel"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "seed%d-%dx%d" % c[:3])
def test_sparse_family_through_dshard(monkeypatch, omp_oracle, tmp_path, case, panel):
    """The same tableaux as ONE row shard (rank 0 of 1) through dshard_select_kernel / dshard_kernel<512,16>, depth 8, the
    Python loop with the status polled every 8 pivots: every pending row is stored by one launch and read by later ones --
    the launch boundary dshard_kernel.cuh relies on for its single `d.dpend`.  (In a child process: the candidate slots are
    torch tensors, and torch has to initialise its HIP runtime before the library does.)"""
    import os
    import subprocess
    import sys
    seed, h, w, *_ , budget = case
    m, pos, var, ref, rpos, rvar, est, eres, epiv = _expect(omp_oracle, case)
    inp, out = str(tmp_path / "in.npy"), str(tmp_path / "out.npz")
    np.save(inp, m)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, YALPS_HIP_DELAY_DEPTH="8", YALPS_HIP_SHARD_PANEL=panel, HSA_ENABLE_IPC_MODE_LEGACY="0")  # (the sweep through LDS panels and straight from L2)
    run = subprocess.run([sys.executable, "-c", DSHARD_CHILD % dict(root=root, inp=inp, out=out, h=h, w=w, budget=budget)],
                         capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert run.returncode == 0 and "ok" in run.stdout, run.stdout + run.stderr
    res = np.load(out)
    kernel = str(res["kernel"])
    assert kernel.startswith("dshard_kernel<512,16") and kernel.endswith("delay_depth:8") and (",panel" in kernel) == (panel == "1"), kernel
    assert (str(res["status"]), int(res["pivots"])) == (est, epiv) and G.same_number(float(res["result"]), eres), (res["status"], res["pivots"], est, epiv)
    assert np.array_equal(res["pos"], rpos) and np.array_equal(res["var"], rvar)
    assert np.array_equal(res["matrix"].view(np.int64), ref.view(np.int64))
    assert tuple(res["pad"]) == (0, 0), res["pad"]
