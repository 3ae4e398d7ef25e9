"""CPU-side check of the boundary: the C-ABI library loads and exports every symbol that
include/yalps_hip.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def nat():
    from yalps_amd import build, _native
    build.build_hip()
    return _native


def test_header_symbols_are_exported(nat):
    text = open(os.path.join(ROOT, "include", "yalps_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(yalps_[a-z0-9_]+)\s*\(", text))
    assert declared, "no declarations parsed"
    L = nat.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(nat.SYMBOLS)


def test_no_cpu_fallback(nat):
    import numpy as np
    if nat.lib().yalps_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(nat.NativeError, match="no HIP device"):
        nat.Context(0)
    m = np.zeros(12)
    p = np.arange(7, dtype=np.int32)
    with pytest.raises(nat.NativeError):
        nat.simplex_host(m, 3, 4, p, p.copy())


def test_product_does_not_touch_the_oracle():
    """Nothing under yalps_amd/ may import, link or execute oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "yalps_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".cc", ".h", ".js", ".mjs")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in src and "simplex_oracle" not in src and "oracle/" not in src.replace(
                    "Nothing here imports the oracle", ""), os.path.join(dirpath, f)


def test_kernel_register_budgets(nat):
    """The build itself refuses a library whose register-resident / persistent / single-workgroup kernels use
    accumulator registers or scratch (yalps_amd/build.py check_register_budgets: a variant that did computed wrong
    rows on the GPU); here the same check runs on the library the tests load."""
    from yalps_amd import build
    ks = build.check_register_budgets()
    assert sum("resident_kernel" in k for k in ks) >= 15, sorted(ks)
