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


def run_world(kind, world, M, N, seed, tmp_path, extra=()):
    out = str(tmp_path / f"res_{kind}_{world}_{M}_{N}_{seed}.npz")
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-m", "tests._shard_worker", kind, str(M), str(N), str(seed), out] + [str(a) for a in extra],
                                      cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    return np.load(out)


def check_against_oracle(oracle, res, M, N, seed, max_pivots=np.inf, phase1=False):
    w, h = N + 1, M + 1
    m = oracle.dense_lp(M, N, seed)
    if phase1:  # (tests/_shard_worker.py, `phase1`)
        A = m.reshape(h, w)
        A[h // 3] *= -1.0
        A[5::7, 3::5] = 0.0
    elif seed % 2:
        m.reshape(h, w)[1::3, 0] *= -0.05
    pos = np.arange(w + h, dtype=np.int32)
    var = pos.copy()
    status, result, npiv, _ = oracle.simplex(m, w, h, pos, var, max_pivots=max_pivots)
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


@pytest.mark.parametrize("world,M,N,seed,depth,budget,phase1", [(2, 40, 30, 4, 3, None, False), (2, 37, 50, 5, 8, None, False),
                                                                (3, 25, 25, 7, 2, None, False), (2, 30, 60, 9, 4, 33, True),
                                                                (3, 45, 45, 10, 5, 50, True)])
def test_sharded_delayed_protocol_gloo_cpu(oracle, tmp_path, world, M, N, seed, depth, budget, phase1):
    """The delayed row updates of the row shards (dshard_kernel / dshard_select_kernel) as a numpy stand-in over gloo, world
    sizes 2 and 3: candidates from scalar chains, candidate rows sent with the pending pivots applied, the sweep every `depth`
    pivots and on the way out -- bit for bit the single-process oracle (the GPU tests run the same cases' kernels)."""
    extra = ["inf" if budget is None else budget] + (["phase1"] if phase1 else [])
    res = run_world("numpy-delayed%d" % depth, world, M, N, seed, tmp_path, extra)
    check_against_oracle(oracle, res, M, N, seed, max_pivots=np.inf if budget is None else float(budget), phase1=phase1)


@pytest.mark.gpu
@pytest.mark.parametrize("world,M,N,seed", [(1, 200, 150, 4), (2, 300, 280, 5), (2, 90, 700, 2), (3, 64, 64, 9),
                                            (2, 120, 3000, 6), (2, 100, 9000, 8)])
def test_sharded_hip_steps(oracle, tmp_path, world, M, N, seed):
    """The real HIP per-rank kernels: `world` processes share the one GPU of the test box and
    exchange their candidate slots through gloo (host-staged); with RCCL on a multi-GPU node only
    the transport differs.  The last two are wide enough for wide_kernel and so for its in-place shard form
    (<1024,2,in place>, <512,16,in place>: several waves per row, rows updated where they are)."""
    res = run_world("hip", world, M, N, seed, tmp_path)
    check_against_oracle(oracle, res, M, N, seed)


DELAYED = [
    # forced onto shards with one row per workgroup (the switch is read by yalps_tableau_set_shard): a whole solve; phase 1
    # first with pivot budgets (these LPs cycle without the budget), depths 2 / 3 / 4 / 8, <512,4> / <512,6> / <512,16>,
    # two and three ranks, both drivers
    ("hip", 2, 120, 3000, 6, None, False, {"YALPS_HIP_DELAY_MIN_ROWS": "1", "YALPS_HIP_DELAY_DEPTH": "4"}, "dshard_kernel<512,4>,delay_depth:4"),
    ("hip", 3, 100, 9000, 8, 401, True, {"YALPS_HIP_DELAY_MIN_ROWS": "1", "YALPS_HIP_DELAY_DEPTH": "3"}, "dshard_kernel<512,16>,delay_depth:3"),
    ("hip-native", 2, 150, 6000, 11, 397, True, {"YALPS_HIP_DELAY_MIN_ROWS": "1", "YALPS_HIP_DELAY_DEPTH": "8"}, "dshard_kernel<512,6>,delay_depth:8"),
    ("hip-native", 3, 90, 3500, 4, 211, True, {"YALPS_HIP_DELAY_MIN_ROWS": "1", "YALPS_HIP_DELAY_DEPTH": "2"}, "dshard_kernel<512,4>,delay_depth:2"),
    # what takes them by default (4+ rows per workgroup): 1150 / 1100 rows per rank; budgets that end between two sweeps
    ("hip", 2, 2300, 4200, 6, 150, False, {}, "dshard_kernel<512,6>,delay_depth:8"),
    ("hip-native", 3, 3300, 4200, 5, 131, True, {}, "dshard_kernel<512,6>,delay_depth:8"),
    # ... and the same shard one sweep per pivot, by request
    ("hip", 2, 2300, 4200, 6, 37, False, {"YALPS_HIP_SHARD_DELAY": "0"}, "wide_kernel<1024,4>"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("kind,world,M,N,seed,budget,phase1,env,kernel", DELAYED)
def test_sharded_delayed_row_updates(oracle, tmp_path, monkeypatch, kind, world, M, N, seed, budget, phase1, env, kernel):
    """dshard_kernel / dshard_select_kernel: a pivot costs a shard its scalars, the rows are swept once per `depth` pivots
    and on the way out; the candidate rows travel with the pending pivots applied.  Status, result, pivot count, basis
    and every bit of the assembled tableau against the oracle."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    extra = ["inf" if budget is None else budget] + (["phase1"] if phase1 else [])
    res = run_world(kind, world, M, N, seed, tmp_path, extra)
    assert str(res["kernel"]) == kernel
    check_against_oracle(oracle, res, M, N, seed, max_pivots=np.inf if budget is None else float(budget), phase1=phase1)


@pytest.mark.gpu
@pytest.mark.parametrize("world,M,N,seed", [(2, 300, 280, 5), (3, 64, 64, 9), (2, 150, 6000, 10)])
def test_sharded_native_loop_host_transport(oracle, tmp_path, world, M, N, seed):
    """yalps_shard_run -- select, exchange, apply and the status polls all inside the library -- with `world` processes
    sharing the test GPU; the exchange is the library's host transport (a callback) carried by gloo.  An odd pivot
    budget is not a multiple of the batch: the loop must stop exactly where the oracle does."""
    res = run_world("hip-native", world, M, N, seed, tmp_path)
    check_against_oracle(oracle, res, M, N, seed)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,seed", [(200, 150, 4), (520, 1100, 3), (130, 9000, 12)])
def test_sharded_native_loop_rccl_one_rank(oracle, tmp_path, M, N, seed):
    """The native loop over RCCL itself: ncclCommInitRank (one rank), ncclAllGather on the context's stream between the
    select and the apply kernel, every pivot enqueued by the library's own loop (one hipGraph replay per batch where the
    runtime lets RCCL be captured) -- bit for bit the oracle.  In a child process that imports torch first, like every
    multi-rank launch does.  (More than one rank per GPU is something RCCL refuses; the multi-rank control flow is covered
    above and on CPU.)"""
    out_npz = str(tmp_path / "rccl1.npz")
    code = (
        "import sys, json, numpy as np; sys.path.insert(0, %r)\n"
        "import torch\n"
        "from tests import _oracle\n"
        "from yalps_amd import sharded\n"
        "M, N, seed = %d, %d, %d; w, h = N + 1, M + 1\n"
        "m = _oracle.load().dense_lp(M, N, seed)\n"
        "if seed %% 2: m.reshape(h, w)[1::3, 0] *= -0.05\n"
        "ident = np.arange(w + h, dtype=np.int32); bounds = sharded.partition(h, 1)\n"
        "ops = sharded.HipShardOps(sharded.local_rows(m, w, h, bounds, 0), w, bounds, 0, h, ident, ident.copy(), device=0, private_stream=True)\n"
        "comm = sharded.native_comm(ops.ctx, 0, 1, transport='rccl')\n"
        "status, result, pivots = sharded.sharded_simplex_native(ops, comm, max_pivots=float('inf'), check_every=16)\n"
        "info = comm.info(); lm, pos, var = ops.download(); comm.close(); ops.close()\n"
        "assert info['transport'] == 'rccl' and int(info['collectives']) >= pivots, info\n"
        "np.savez(%r, matrix=lm, pos=pos, var=var, status=status, result=result, pivots=pivots)\n"
        "print('info', json.dumps(info)); print('ok')\n" % (ROOT, M, N, seed, out_npz))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr
    check_against_oracle(oracle, np.load(out_npz), M, N, seed)


@pytest.mark.gpu
def test_sharded_python_driver_over_torch_nccl_backend(oracle, tmp_path):
    """torch.distributed's "nccl" backend (= RCCL) carrying the Python driver's all-gather on device tensors: a group
    of one rank still initialises RCCL, runs all_gather_into_tensor on torch's stream and orders it against the select
    and apply kernels enqueued on the same stream through yalps_ctx_create_on_stream."""
    code = (
        "import os, sys, numpy as np; sys.path.insert(0, %r)\n"
        "import torch, torch.distributed as dist\n"
        "from tests import _oracle\n"
        "from yalps_amd import sharded\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "M, N, seed = 200, 150, 4; w, h = N + 1, M + 1\n"
        "m = _oracle.load().dense_lp(M, N, seed); ident = np.arange(w + h, dtype=np.int32)\n"
        "bounds = sharded.partition(h, 1)\n"
        "ops = sharded.HipShardOps(sharded.local_rows(m, w, h, bounds, 0), w, bounds, 0, h, ident, ident.copy(), device=0)\n"
        "comm = sharded.TorchComm(always_collective=True); assert not comm.stage\n"
        "status, result, pivots = sharded.sharded_simplex(ops, comm, max_pivots=float('inf'), check_every=8)\n"
        "lm, pos, var = ops.download(); ops.close()\n"
        "np.savez(%r, matrix=lm, pos=pos, var=var, status=status, result=result, pivots=pivots)\n"
        "dist.destroy_process_group(); print('ok')\n" % (ROOT, str(tmp_path / "nccl1.npz")))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr
    check_against_oracle(oracle, np.load(str(tmp_path / "nccl1.npz")), 200, 150, 4)
