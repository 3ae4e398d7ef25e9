"""Row-sharded solve: the distributed driver must reproduce the single-process
oracle bit for bit -- status, result, pivot count, permutations and the assembled tableau."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import _golden as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_world(kind, world, M, N, seed, tmp_path):
    out = str(tmp_path / f"res_{kind}_{world}_{M}_{N}_{seed}.npz")
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-m", "tests._shard_worker", kind, str(M), str(N), str(seed), out],
                                      cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    return np.load(out)


def check_against_oracle(oracle, res, M, N, seed):
    w, h = N + 1, M + 1
    m = oracle.dense_lp(M, N, seed)
    if seed % 2:
        m.reshape(h, w)[1::3, 0] *= -0.05
    pos = np.arange(w + h, dtype=np.int32)
    var = pos.copy()
    status, result, npiv, _ = oracle.simplex(m, w, h, pos, var, max_pivots=np.inf)
    assert str(res["status"]) == status and int(res["pivots"]) == npiv
    assert G.same_number(float(res["result"]), result)
    assert np.array_equal(res["pos"], pos) and np.array_equal(res["var"], var)
    assert np.array_equal(res["matrix"].view(np.int64), m.view(np.int64))


def test_partition():
    from yalps_amd import sharded
    assert sharded.partition(10, 2) == [1, 6, 10]
    assert sharded.partition(2049, 8)[0] == 1 and sharded.partition(2049, 8)[-1] == 2049
    assert sharded.partition(3, 4) == [1, 2, 3, 3, 3]  # ranks without rows are allowed


@pytest.mark.parametrize("world,M,N,seed", [(2, 40, 30, 4), (2, 37, 50, 5), (3, 25, 25, 7)])
def test_sharded_driver_gloo_cpu(oracle, tmp_path, world, M, N, seed):
    """world_size > 1 on CPU: gloo all-gather + the numpy stand-in for the per-rank steps."""
    res = run_world("numpy", world, M, N, seed, tmp_path)
    check_against_oracle(oracle, res, M, N, seed)


@pytest.mark.gpu
@pytest.mark.parametrize("world,M,N,seed", [(1, 200, 150, 4), (2, 300, 280, 5), (2, 90, 700, 2), (3, 64, 64, 9)])
def test_sharded_hip_steps(oracle, tmp_path, world, M, N, seed):
    """The real HIP per-rank kernels: `world` processes share the one GPU of the test box and
    exchange their candidate slots through gloo (host-staged); with RCCL on a multi-GPU node only
    the transport differs."""
    res = run_world("hip", world, M, N, seed, tmp_path)
    check_against_oracle(oracle, res, M, N, seed)
