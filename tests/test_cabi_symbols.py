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


def _kernel_metadata():
    """{demangled-ish kernel name: {vgpr_count, agpr_count, private_segment_fixed_size, ...}} of the gfx950
    code object inside the built library (clang-offload-bundler + llvm-readelf of this image's ROCm)."""
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    lib = os.path.join(ROOT, "yalps_amd", "libyalps_hip.so")
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.run([f"{llvm}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        # one bundle per translation unit (yalps_hip.hip + persistent_*.hip), back to back in the section
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        blob = open(fat, "rb").read()
        starts = [i for i in range(len(blob)) if blob.startswith(magic, i)]
        assert starts, "no offload bundle in .hip_fatbin"
        notes = ""
        for k, lo in enumerate(starts):
            part = os.path.join(tmp, "part%d.bin" % k)
            with open(part, "wb") as f:
                f.write(blob[lo:starts[k + 1] if k + 1 < len(starts) else len(blob)])
            subprocess.run([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
            notes += subprocess.run([f"{llvm}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    kernels, cur = {}, {}
    for line in notes.splitlines():
        m = re.match(r"\s*(?:- )?\.(\w+):\s+(\S+)\s*$", line)
        if not m:
            continue
        key, val = m.groups()
        if key == "agpr_count" and line.lstrip().startswith("- "):  # first key of a kernel's record
            cur = {}
        cur[key] = val
        if key == "name" and val.startswith("_Z"):
            kernels[val] = cur
    return kernels


def test_kernel_register_budgets(nat):
    """Built code object: every kernel is there for gfx950, the resident variants (whose rows live in
    registers for the whole solve) need no AGPRs -- a variant that did left rows unwritten on the GPU --
    and neither they nor the other persistent / single-workgroup kernels spill to scratch (SGPR spills that end
    up in scratch computed garbage in a stream_kernel variant)."""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("ROCm LLVM tools not installed")
    ks = _kernel_metadata()
    resident = {k: v for k, v in ks.items() if "resident_kernel" in k}
    assert len(resident) >= 15, sorted(ks)
    for name, md in resident.items():
        assert int(md["vgpr_count"]) <= 256 and int(md["agpr_count"]) == 0, (name, md["vgpr_count"], md["agpr_count"])
    for name, md in ks.items():
        if "small_kernel" in name or "batch_kernel" in name or "assemble" in name or "resident_kernel" in name \
                or "stream_kernel" in name:
            assert int(md["private_segment_fixed_size"]) == 0, (name, md["private_segment_fixed_size"])
