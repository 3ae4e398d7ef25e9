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


def build_hip(force=False, verbose=False):
    if not force and not _stale(LIB, HEADER, *HIP_UNITS, *HIP_DEPS):
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs = [os.path.join(OBJ_DIR, os.path.basename(u)[:-4] + ".o") for u in HIP_UNITS]

    def compile_unit(pair):
        unit, obj = pair
        if not force and not _stale(obj, unit, HEADER, *HIP_DEPS):
            return
        cmd = [hipcc, *HIPCC_FLAGS, "-c", "-o", obj, unit]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(objs), os.cpu_count() or 1)) as pool:
        list(pool.map(compile_unit, zip(HIP_UNITS, objs)))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp", *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)  # (never a half-written library under the final name)
    return LIB


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
    print(build_hip(force=True, verbose=True))
    print(build_napi(force=True, verbose=True))
