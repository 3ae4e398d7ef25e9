"""bench.py as the driver runs it (small sizes): one JSON line with the contract's keys, `roofline` and `cpu_baseline`
at N=1; the sharded workload's line.  A crash in either path is a lost round-end measurement."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline")


def _bench(args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "bench.py"] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_default_workload_line():
    rec = _bench(["--size", "300", "--steps", "2", "--warmup", "1", "--cpu-pivots", "200", "--sweep-launches", "4"])
    for k in KEYS + ("cpu_baseline",):
        assert k in rec, k
    assert rec["n_gpus"] == 1 and rec["steps"] == 2 and rec["vs_baseline"] is None and rec["dtype"] == "f64"
    assert rec["value"] > 0 and rec["roofline"]["bound"] in ("hbm", "mfma", "onchip") and rec["cpu_baseline"]["value"] > 0
    assert "workload" in rec["config"]
    rf = rec["roofline"]
    if rf["bound"] == "onchip":  # the register-resident kernel: one bound stated consistently, the HBM figure beside it
        assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and rf["unit"] == "pivots/s" and rf["frac_hbm"] > 0
        floor = rf["onchip_floor"]
        assert 0 < floor["measured_exchange"]["flags_only_us"] <= floor["measured_exchange"]["publish_flags_fetch_us"] < floor["us"] < 50


@pytest.mark.gpu
@pytest.mark.parametrize("extra,kernel", [([], "dshard_kernel"), (["--shard-rows", "300"], "wide_kernel")])
def test_sharded_workload_line(extra, kernel):
    """--workload sharded at one rank (RCCL transport, world 1): 5 000 columns; 1 200 rows take the delayed shard kernel,
    300 rows (two per workgroup) one sweep per pivot."""
    rec = _bench(["--workload", "sharded", "--size", "5000", "--steps", "1", "--warmup", "1", "--pivots-per-step", "24"]
                 + (extra if extra else ["--shard-rows", "1200"]))
    for k in KEYS:
        assert k in rec, k
    assert rec["scaling"] == "strong" and rec["value"] > 0
    assert rec["roofline"]["kernel"].startswith(kernel), rec["roofline"]
    assert rec["parity"]["ok"] is True and rec["parity"]["pivots"] == 48 and rec["cpu_baseline"]["value"] > 0, rec["parity"]
    if kernel == "dshard_kernel":
        assert rec["roofline"]["delay_depth"] == 8 and "algorithmic_equiv" in rec["roofline"]


def _bench_ranks(n, args):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
                          "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", str(n)] + args, cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_ranks_line_carries_the_row_sharded_measurement():
    """The driver's multi-GPU command, rehearsed with two ranks sharing the test GPU: the replicas headline, and beside it
    `sharded_c5` -- ONE tableau row-sharded over the ranks, measured by child processes in a group of their own (here a
    4097 x 4097 stand-in for config 5 over the host transport; on a node: 16385 x 16385 over RCCL)."""
    rec = _bench_ranks(2, ["--size", "300", "--steps", "2", "--warmup", "1", "--sharded-c5-size", "4096", "--pivots-per-step", "40"])
    for k in KEYS[:-1]:
        assert k in rec, k
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    c5 = rec["sharded_c5"]
    assert "error" not in c5, c5
    assert c5["n_gpus"] == 2 and c5["scaling"] == "strong" and c5["value"] > 0 and c5["us_per_pivot"] > 0
    assert c5["roofline"]["kernel"].startswith("dshard_kernel") and c5["exchange"]["transport"] == "host"
    assert "4097x4097" in c5["workload"] and c5["parity"]["ok"] is True, c5.get("parity")


@pytest.mark.gpu
def test_a_failing_row_sharded_measurement_leaves_the_headline_alone():
    """Children that do not finish in time are killed and reported; the line, its headline and the return code stay."""
    rec = _bench_ranks(2, ["--size", "300", "--steps", "2", "--warmup", "1", "--sharded-c5-size", "4096", "--sharded-c5-timeout", "0.05"])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and "timed out" in rec["sharded_c5"]["error"]
