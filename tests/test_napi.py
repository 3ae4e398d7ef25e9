"""The N-API shim (yalps_amd/napi): builds against node's headers, loads under node, exposes
simplex(tableau, options) -> [status, number] like the reference export (src/simplex.ts:144)."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "yalps_amd", "napi", "run_simplex.js")
README = {"matrix": [0, 1200, 1600, 300, 30, 20, 110, 5, 10, 400, 30, 50], "width": 3, "height": 4, "viewOffset": 4}

pytestmark = pytest.mark.skipif(shutil.which("node") is None or not os.path.exists("/usr/include/node/node_api.h"),
                                reason="node / node_api.h not available")


def run_node(job):
    from yalps_amd import build
    build.build_hip()
    assert build.build_napi() is not None
    out = subprocess.run(["node", DRIVER], input=json.dumps(job), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_addon_loads_and_fails_loudly_without_gpu():
    from yalps_amd import _native
    if _native.lib().yalps_device_count() > 0:
        pytest.skip("a GPU is present")
    res = run_node(README)
    assert "no HIP device" in res["error"]  # thrown as a JS Error; no CPU fallback


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 2, 0), (40, 30, 7), (120, 200, 3)])
def test_addon_matches_oracle(oracle, shape):
    M, N, off = shape
    w, h = N + 1, M + 1
    m = oracle.dense_lp(M, N, 11) if M > 3 else np.array(README["matrix"], np.float64)
    if M <= 3:
        w, h = 3, 4
    res = run_node({"matrix": m.tolist(), "width": w, "height": h, "viewOffset": off,
                    "options": {"maxPivots": "Infinity"}})
    assert "error" not in res, res
    ref = m.copy()
    pos = np.arange(w + h, dtype=np.int32)
    var = pos.copy()
    status, result, _, _ = oracle.simplex(ref, w, h, pos, var, max_pivots=np.inf)
    assert res["status"] == status and float(res["result"]) == result
    assert np.array_equal(np.array(res["matrix"]).view(np.int64), ref.view(np.int64))  # JSON round-trips doubles exactly
    assert res["positionOfVariable"] == pos.tolist() and res["variableAtPosition"] == var.tolist()
    assert res["guardsIntact"]  # typed-array views honoured: nothing written outside them
