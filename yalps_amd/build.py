"""Builds the native pieces in-tree (gfx950 only).  Used by __graft_entry__.build().

  yalps_amd/libyalps_hip.so   HIP kernels + C ABI (include/yalps_hip.h)      hipcc
  yalps_amd/napi/yalps_napi.node  thin N-API shim over the C ABI (optional)  g++

The .so files are git-ignored but travel to the GPU box with the tree.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libyalps_hip.so")
LIB_STAMPS = os.path.join(HERE, "libyalps_hip_stamps.so")
LINK_LIBS = []
CSRC = os.path.join(HERE, "csrc")
HIP_SRC = os.path.join(CSRC, "yalps_hip.hip")  # host side + C ABI + the launch-per-pivot / single-workgroup / batch kernels
# the persistent kernels' instantiations, one translation unit per group: compiled side by side (the device compile of
# ~40 register-heavy kernels in one unit took 2.5 minutes)
HIP_UNITS = [HIP_SRC] + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.startswith("persistent_") and f.endswith(".hip")]
HIP_DEPS = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".cuh", ".inc", ".h"))]
OBJ_DIR = os.path.join(HERE, "build")
HEADER = os.path.join(ROOT, "include", "yalps_hip.h")
NAPI_SRC = os.path.join(HERE, "napi", "yalps_napi.cc")
NAPI_OUT = os.path.join(HERE, "napi", "yalps_napi.node")

# -ffp-contract=off: the reference (V8) rounds the product and the difference of
# M[r,c] - coef*M[row,c] separately (src/simplex.ts:33); an fma would change pivot paths.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]


def _stale(out, *srcs):
    return not os.path.exists(out) or any(os.path.getmtime(out) < os.path.getmtime(s) for s in srcs if os.path.exists(s))


def kernel_metadata(lib=LIB):
    """{mangled kernel name: {vgpr_count, agpr_count, private_segment_fixed_size, ...}} of the gfx950 code objects inside a
    built library (llvm-objcopy + clang-offload-bundler + llvm-readelf of this image's ROCm)."""
    import re
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.run([f"{llvm}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        # one bundle per translation unit (yalps_hip.hip + persistent_*.hip), back to back in the section
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        blob = open(fat, "rb").read()
        starts = [i for i in range(len(blob)) if blob.startswith(magic, i)]
        if not starts:
            raise RuntimeError("no offload bundle in .hip_fatbin of %s" % lib)
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


# Kernels whose rows / tableaux live in registers or LDS behind hand-written sc1 loads and stores: built without
# scratch and without accumulator registers, or not at all.  (Two instantiations that broke this rule computed wrong
# rows on the GPU -- DESIGN.md 4.7 -- so the rule is part of the build, not of an optional test.)
NO_SCRATCH = ("dshard_kernel", "dshard_select_kernel", "dshard_sweep_kernel", "small_kernel", "batch_kernel", "assemble", "resident_kernel", "resident2_kernel", "stream_kernel", "stream2_kernel", "stream3_kernel", "sweep_kernel")


def check_register_budgets(lib=LIB, min_resident=15):
    ks = kernel_metadata(lib)
    resident = {k: v for k, v in ks.items() if "resident_kernel" in k or "resident2_kernel" in k}
    bad = []
    if len(resident) < min_resident:
        bad.append("only %d resident_kernel instantiations in the code object" % len(resident))
    for name, md in sorted(ks.items()):
        if ("resident_kernel" in name or "resident2_kernel" in name) and (int(md["vgpr_count"]) > 256 or int(md["agpr_count"]) != 0):
            bad.append("%s: vgpr_count %s agpr_count %s" % (name, md["vgpr_count"], md["agpr_count"]))
        if any(tag in name for tag in NO_SCRATCH) and int(md["private_segment_fixed_size"]) != 0:
            bad.append("%s: private_segment_fixed_size %s (scratch)" % (name, md["private_segment_fixed_size"]))
    if bad and os.environ.get("YALPS_BUILD_ALLOW_SCRATCH") == "1":  # (experiments only: a same-box A/B of a form that does not fit yet)
        print("register budget violated (YALPS_BUILD_ALLOW_SCRATCH=1: building anyway):\n  " + "\n  ".join(bad))
        return ks
    if bad:
        raise RuntimeError("register budget violated (a spilling variant computes wrong rows):\n  " + "\n  ".join(bad))
    return ks


def build_hip(force=False, verbose=False, stamps=False):
    """stamps=True: the diagnostic build with in-kernel stage stamps (-DYALPS_STAMPS -> libyalps_hip_stamps.so; selected
    with YALPS_HIP_LIB by tools/resident_stages.py, never loaded by default)."""
    lib_out = LIB_STAMPS if stamps else LIB
    obj_dir = OBJ_DIR + ("_stamps" if stamps else "")
    if not force and not _stale(lib_out, HEADER, *HIP_UNITS, *HIP_DEPS):
        return lib_out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(obj_dir, exist_ok=True)
    objs = [os.path.join(obj_dir, os.path.basename(u)[:-4] + ".o") for u in HIP_UNITS]
    flags = HIPCC_FLAGS + (["-DYALPS_STAMPS"] if stamps else [])

    def compile_unit(pair):
        unit, obj = pair
        if not force and not _stale(obj, unit, HEADER, *HIP_DEPS):
            return
        cmd = [hipcc, *flags, "-c", "-o", obj, unit]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(objs), os.cpu_count() or 1)) as pool:
        list(pool.map(compile_unit, zip(HIP_UNITS, objs)))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_out + ".tmp", *objs, *LINK_LIBS]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    if not stamps:
        check_register_budgets(lib_out + ".tmp")  # (the stamped kernels keep their sums in extra scalar registers)
    os.replace(lib_out + ".tmp", lib_out)  # (never a half-written library under the final name)
    return lib_out


def build_napi(force=False, verbose=False):
    """The Node addon; skipped (returns None) where node's headers are absent."""
    inc = "/usr/include/node"
    if not os.path.exists(os.path.join(inc, "node_api.h")) or not os.path.exists(NAPI_SRC):
        return None
    if not force and not _stale(NAPI_OUT, NAPI_SRC, HEADER):
        return NAPI_OUT
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I", inc, "-I", os.path.join(ROOT, "include"),
           "-o", NAPI_OUT, NAPI_SRC, "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return NAPI_OUT


if __name__ == "__main__":
    import sys
    if "stamps" in sys.argv[1:]:
        print(build_hip(verbose=True, stamps=True))
        raise SystemExit(0)
    print(build_hip(force=True, verbose=True))
    print(build_napi(force=True, verbose=True))
