"""Parity at BASELINE's full sizes (GPU).  Config 2 (2049 x 2049, whole solve) is pinned by the
reference's own golden record in test_hip_parity.py.  Config 5 (16385 x 16385, 2.1 GB) is too big
for the scalar oracle: the first pivots of that tableau -- phase 1 and phase 2, unsharded on one GPU and
through the row-sharded path (wide_kernel<1024,8>) with two ranks -- are compared bit
for bit with tests/_np_simplex.py, the vectorised restatement that test_oracle_golden.py pins to
the reference's golden records."""
import hashlib
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import _np_simplex as NP

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
M = N = 16384
MAX_PIVOTS = 3


def reference(negate):
    """dense-LP(16384,16384,42) (negate: with one row turned into "-a x <= -b", which makes the start
    infeasible: the pivots are then phase-1 pivots) and what the reference's arithmetic leaves after
    MAX_PIVOTS pivots (budget exhausted -> "cycled")."""
    from yalps_amd import _native as nat
    w, h = N + 1, M + 1
    m = nat.dense_lp(M, N, 42)
    if negate:
        m.reshape(h, w)[h // 3] *= -1.0
    ref = m.copy()
    pos, var = np.arange(w + h, dtype=np.int32), np.arange(w + h, dtype=np.int32)
    status, _, npiv = NP.simplex(ref, w, h, pos, var, max_pivots=MAX_PIVOTS)
    assert status == "cycled" and npiv == MAX_PIVOTS
    return dict(nat=nat, w=w, h=h, m=m, ref=ref, pos=pos, var=var)


@pytest.mark.parametrize("delay,kernel", [("1", "stream3_kernel<512,16,nt>"), ("0", "sweep_kernel<512,16,nt>")], ids=["stream3", "sweep"])
def test_c5_streaming_kernel_matches_restatement(monkeypatch, delay, kernel):
    """One GPU, phase-2 pivots, through the two persistent in-place kernels that unsharded 8194..16385-column tableaux take --
    stream3_kernel (the rows get two pivots' eliminations per sweep; the odd budget leaves through the one-pivot flush) and
    sweep_kernel --, non-temporal row traffic; the whole 2.1 GB tableau is compared."""
    monkeypatch.setenv("YALPS_HIP_DELAY", delay)
    c5 = reference(negate=False)
    nat, w, h = c5["nat"], c5["w"], c5["h"]
    ctx = nat.Context(0)
    t = nat.DeviceTableau(ctx, w, h)
    try:
        ident = np.arange(w + h, dtype=np.int32)
        t.upload(c5["m"], h, ident, ident.copy())
        status, result, npiv, _ = t.solve(max_pivots=MAX_PIVOTS)
        assert t.info()["last_path"] == "inplace" and t.info()["inplace"] == kernel, t.info()
        got, gpos, gvar = t.download()
    finally:
        t.close()
        ctx.close()
    assert (status, npiv) == ("cycled", MAX_PIVOTS) and result != result
    assert np.array_equal(gpos, c5["pos"]) and np.array_equal(gvar, c5["var"])
    assert np.array_equal(got.view(np.int64), c5["ref"].view(np.int64))


def test_c5_row_sharded_two_ranks_matches_restatement(tmp_path):
    """Two processes share the test GPU, each holds half of the rows (+ the objective row); the
    candidates + candidate rows travel through gloo; phase-1 pivots.  SHA-256 of every rank's block."""
    c5 = reference(negate=True)
    w, h = c5["w"], c5["h"]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "c5.npz")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-m", "tests._shard_worker", "hip", str(M), str(N), "42", out,
                                       str(MAX_PIVOTS), "digest"], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    res = np.load(out)
    ref = c5["ref"].reshape(h, w)
    assert str(res["status"]) == "cycled" and int(res["pivots"]) == MAX_PIVOTS
    assert np.array_equal(res["pos"], c5["pos"]) and np.array_equal(res["var"], c5["var"])
    bounds = res["bounds"]
    for rank in range(2):
        assert str(res["row0"][rank]) == hashlib.sha256(ref[0].tobytes()).hexdigest()
        assert str(res["blocks"][rank]) == hashlib.sha256(ref[bounds[rank]:bounds[rank + 1]].tobytes()).hexdigest()


def _run_bnb(extra, env=None):
    import json
    out = subprocess.run([sys.executable] + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_c4_every_node_of_the_batch_matches_oracle():
    """BASELINE config 4 at full size: 1024 branch-and-cut nodes of a 512-variable MILP in one batch;
    bench_bnb.py checks status, pivot count, permutations and the whole tableau of EVERY node against the
    CPU oracle (and a sample against the one-node-at-a-time drop-in call)."""
    rec = _run_bnb(["bench_bnb.py", "--check", "1024", "--seq-sample", "16"])
    assert rec["oracle_checked_nodes"] == 1024 and rec["batch"]["status_counts"].get("optimal", 0) > 0


def test_c4_node_queue_dealt_to_two_ranks():
    """The same queue dealt round-robin to two processes (they share the test GPU; no data-path collective)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    rec = _run_bnb(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", str(port), "bench_bnb.py", "--gpus", "2", "--check", "64", "--seq-sample", "8"],
                   env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert rec["n_gpus"] == 2 and rec["config"]["nodes_per_rank"] == 512 and rec["oracle_checked_nodes"] == 64
