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
    assert rec["value"] > 0 and rec["roofline"]["bound"] in ("hbm", "mfma") and rec["cpu_baseline"]["value"] > 0
    assert "workload" in rec["config"]


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
    if kernel == "dshard_kernel":
        assert rec["roofline"]["delay_depth"] == 8 and "algorithmic_equiv" in rec["roofline"]
