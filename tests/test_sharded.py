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
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=int(os.environ.get("YALPS_TEST_WORLD_TIMEOUT", "300")))[0])
    except subprocess.TimeoutExpired:  # (a rank that waits for a collective its peers never enter: show what everybody printed)
        for p in procs:
            p.kill()
        logs = [p.communicate()[0] for p in procs]
        raise AssertionError("ranks did not finish:\n" + "\n---\n".join(logs))
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
                                                                (3, 25, 25, 7, 2, None, False), (2, 30, 60, 9, 4, 33, True), (2, 37, 50, 5, 16, None, False), (3, 41, 33, 12, 13, 90, True),
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
    # round 3: up to 16 pending pivots (the sweep stages them in LDS, panel_flush.cuh); budgets between two sweeps
    # (the sweep through LDS panels, forced onto small shards -- by default it is taken from 12 rows per workgroup on --, and the
    # pending rows straight from L2 at depths beyond 8)
    ("hip", 2, 2300, 4200, 6, 150, False, {"YALPS_HIP_DELAY_DEPTH": "16", "YALPS_HIP_SHARD_PANEL": "1"}, "dshard_kernel<512,6,panel>,delay_depth:16"),
    ("hip-native", 3, 3300, 4200, 5, 131, True, {"YALPS_HIP_DELAY_DEPTH": "12", "YALPS_HIP_SHARD_PANEL": "1"}, "dshard_kernel<512,6,panel>,delay_depth:12"),
    ("hip", 2, 100, 9000, 8, 401, True, {"YALPS_HIP_DELAY_MIN_ROWS": "1", "YALPS_HIP_DELAY_DEPTH": "16", "YALPS_HIP_SHARD_PANEL": "1"}, "dshard_kernel<512,16,panel>,delay_depth:16"),
    ("hip-native", 2, 4700, 1000, 3, 90, False, {"YALPS_HIP_DELAY_DEPTH": "13", "YALPS_HIP_SHARD_PANEL": "1"}, "dshard_kernel<512,1,panel>,delay_depth:13"),
    ("hip", 2, 2300, 4200, 6, 150, False, {"YALPS_HIP_DELAY_DEPTH": "14"}, "dshard_kernel<512,6>,delay_depth:14"),
    ("hip-native", 2, 13000, 2100, 7, 45, True, {}, "dshard_kernel<512,4,panel>,delay_depth:16"),  # 26 rows per workgroup: the default
    # the sweep as a launch of its own (dsweep_kernel.cuh: rows mapped to workgroups by panel and row block) is what the native loop
    # takes below 24 rows per workgroup whenever a batch is a whole number of sweeps (the hip-native cases above with depths 8 and 2:
    # batches of 8 pivots); here switched off for such a shard, and forced onto one with 26 rows per workgroup
    ("hip-native", 3, 3300, 4200, 5, 131, True, {"YALPS_HIP_SHARD_XSWEEP": "0"}, "dshard_kernel<512,6>,delay_depth:8"),
    ("hip-native", 2, 13000, 2100, 7, 45, True, {"YALPS_HIP_SHARD_XSWEEP": "1", "YALPS_HIP_DELAY_DEPTH": "8"}, "dshard_kernel<512,4,panel>,delay_depth:8"),
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


# ---- options.checkCycles on row shards (src/simplex.ts:44-63,98,137): permutations and pivot history are replicated, every
# rank runs the detector on the pivot everybody is about to decide -- no communication ------------------------------------
def _cycling_inputs(oracle, tmp_path):
    """name -> (file, w, h): Chvatal's cycling LP (the reference's own test case) padded with all-zero rows (they never leave
    the basis) so that every rank owns rows, alone and embedded in 3001 columns (zero columns never enter); KLEIN2 of the
    reference's netlib selection (478 x 55, 4644 pivots with checkCycles)."""
    from yalps_amd import mps
    from yalps_amd.model import tableau_model
    rec = next(r for r in G.records("cases") if r["name"] == "Chvatal Cycling")
    small = G.initial_matrix(rec, oracle).reshape(rec["height"], rec["width"])
    out = {}
    for name, hh, ww in (("chvatal", 40, rec["width"]), ("chvatal-wide", 40, 3001)):
        big = np.zeros((hh, ww))
        big[:rec["height"], :rec["width"]] = small
        out[name] = (big, ww, hh)
    prob = mps.read_benchmarks(os.path.join(G.GOLDEN, "netlib"), names=("KLEIN2",))[0]
    tab = tableau_model(prob["model"]).tableau
    out["klein2"] = (np.asarray(tab.matrix, np.float64).reshape(tab.height, tab.width), tab.width, tab.height)
    files = {}
    for name, (mat, ww, hh) in out.items():
        path = str(tmp_path / (name + ".npy"))
        np.save(path, mat)
        files[name] = (path, mat, ww, hh)
    return files


def _check_cycles_case(oracle, tmp_path, kind, world, name, env=None, monkeypatch=None):
    path, mat, w, h = _cycling_inputs(oracle, tmp_path)[name]
    for k, v in (env or {}).items():
        monkeypatch.setenv(k, v)
    res = run_world(kind, world, h - 1, w - 1, 0, tmp_path, ["inf", "npy:" + path + ":check"])
    m = mat.reshape(-1).copy()
    pos = np.arange(w + h, dtype=np.int32)
    var = pos.copy()
    status, result, npiv, _ = oracle.simplex(m, w, h, pos, var, max_pivots=np.inf, check_cycles=True)
    assert str(res["status"]) == status and int(res["pivots"]) == npiv, (res["status"], res["pivots"], status, npiv)
    assert G.same_number(float(res["result"]), result)
    assert np.array_equal(res["pos"], pos) and np.array_equal(res["var"], var)
    assert np.array_equal(res["matrix"].view(np.int64), m.view(np.int64))
    return status, str(res["kernel"])


@pytest.mark.parametrize("kind,world", [("numpy", 2), ("numpy", 3), ("numpy-delayed4", 2), ("numpy-delayed3", 3)])
def test_sharded_check_cycles_protocol_gloo_cpu(oracle, tmp_path, kind, world):
    """The replicated hasCycle as a numpy stand-in over gloo: Chvatal's LP must stop "cycled" at the oracle's pivot, with one
    sweep per pivot and with delayed row updates (the pending pivots are carried out, the cycling one is not)."""
    status, _ = _check_cycles_case(oracle, tmp_path, kind, world, "chvatal")
    assert status == "cycled"


@pytest.mark.gpu
@pytest.mark.parametrize("kind,world,name,env,kernel", [
    ("hip", 2, "chvatal", {}, "pivot_kernel"),
    ("hip-native", 3, "chvatal", {}, "pivot_kernel"),
    ("hip", 2, "chvatal-wide", {}, "wide_kernel"),
    ("hip-native", 3, "chvatal-wide", {}, "wide_kernel"),
    ("hip", 3, "chvatal-wide", {"YALPS_HIP_DELAY_MIN_ROWS": "1", "YALPS_HIP_DELAY_DEPTH": "4"}, "dshard_kernel<512,4>"),
    ("hip-native", 2, "chvatal-wide", {"YALPS_HIP_DELAY_MIN_ROWS": "1"}, "dshard_kernel<512,4>"),
    ("hip-native", 2, "klein2", {}, "pivot_kernel"),
    ("hip-native", 3, "klein2", {}, "pivot_kernel"),
])
def test_sharded_check_cycles(oracle, tmp_path, monkeypatch, kind, world, name, env, kernel):
    """yalps_shard_begin / yalps_shard_run with checkCycles: shard_cycle_kernel between the all-gather and the step launch,
    through all three step kernels (pivot_kernel, wide_kernel in place, dshard_kernel with pivots pending when the cycle
    closes), Python loop and native loop, two and three ranks sharing the test GPU: status, pivot count, basis and every
    bit of the tableau against the oracle (KLEIN2: 4644 pivots, the history grows twice on the way)."""
    status, got_kernel = _check_cycles_case(oracle, tmp_path, kind, world, name, env, monkeypatch)
    assert got_kernel.startswith(kernel), got_kernel
    if name != "klein2":
        assert status == "cycled"


@pytest.mark.gpu
def test_sharded_graph_cache_follows_the_tableau(oracle, tmp_path):
    """one communicator serving several tableaux.  The batch captured for the first tableau holds its device
    pointers by value; a second tableau -- same shape, very likely the same host address after the first is destroyed -- and a
    re-partition of a tableau must each get a fresh capture (yalps_tableau::generation), and a checkCycles run after a plain
    one on the same tableau as well (the captured batch lacks the detector's launch)."""
    out_npz = str(tmp_path / "gen.npz")
    code = (
        "import sys, json, numpy as np; sys.path.insert(0, %r)\n"
        "import torch\n"
        "from tests import _oracle\n"
        "from yalps_amd import sharded, _native\n"
        "M, N = 520, 1100; w, h = N + 1, M + 1\n"
        "orc = _oracle.load(); ident = np.arange(w + h, dtype=np.int32); bounds = sharded.partition(h, 1)\n"
        "ctx = _native.Context(0)\n"
        "comm = None; res = {}\n"
        "for k, seed in enumerate((3, 8, 11)):\n"
        "    m = orc.dense_lp(M, N, seed)\n"
        "    ops = sharded.HipShardOps.__new__(sharded.HipShardOps)\n"
        "    ops.torch = torch; ops.ctx = ctx\n"
        "    ops.tab = _native.DeviceTableau(ctx, w, h); ops.tab.upload(m, h, ident, ident.copy())\n"
        "    ops.tab.set_shard(0, 1, bounds, h, ident, ident.copy()); ops.perm_len = w + h\n"
        "    if comm is None: comm = sharded.native_comm(ctx, 0, 1, transport='rccl')\n"
        "    check = k == 2\n"
        "    status, result, pivots = sharded.sharded_simplex_native(ops, comm, max_pivots=float('inf'), check_every=16, check_cycles=check)\n"
        "    if k == 1:  # the same tableau again, re-partitioned and refilled: new device arrays behind the same handle\n"
        "        ops.tab.close(); ops.tab = _native.DeviceTableau(ctx, w, h); ops.tab.upload(m, h, ident, ident.copy())\n"
        "        ops.tab.set_shard(0, 1, bounds, h, ident, ident.copy())\n"
        "        status, result, pivots = sharded.sharded_simplex_native(ops, comm, max_pivots=float('inf'), check_every=16)\n"
        "    lm, pos, var = ops.tab.download(perm_len=w + h)\n"
        "    res['m%%d' %% k], res['pos%%d' %% k], res['var%%d' %% k] = lm, pos, var\n"
        "    res['st%%d' %% k], res['piv%%d' %% k] = status, pivots\n"
        "    ops.tab.close()\n"
        "info = comm.info(); comm.close(); ctx.close()\n"
        "assert int(info['graph_replays']) > 0, info\n"
        "np.savez(%r, **res); print('ok')\n" % (ROOT, out_npz))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr
    res = np.load(out_npz)
    M, N = 520, 1100
    w, h = N + 1, M + 1
    for k, seed in enumerate((3, 8, 11)):
        m = oracle.dense_lp(M, N, seed)
        pos = np.arange(w + h, dtype=np.int32)
        var = pos.copy()
        status, result, npiv, _ = oracle.simplex(m, w, h, pos, var, max_pivots=np.inf, check_cycles=k == 2)
        assert (str(res["st%d" % k]), int(res["piv%d" % k])) == (status, npiv)
        assert np.array_equal(res["pos%d" % k], pos) and np.array_equal(res["var%d" % k], var)
        assert np.array_equal(res["m%d" % k].view(np.int64), m.view(np.int64))
