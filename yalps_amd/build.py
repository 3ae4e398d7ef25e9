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
HIP_SRC = os.path.join(HERE, "csrc", "yalps_hip.hip")
HIP_DEPS = [os.path.join(HERE, "csrc", f) for f in sorted(os.listdir(os.path.join(HERE, "csrc"))) if f.endswith((".cuh", ".inc"))]
HEADER = os.path.join(ROOT, "include", "yalps_hip.h")
NAPI_SRC = os.path.join(HERE, "napi", "yalps_napi.cc")
NAPI_OUT = os.path.join(HERE, "napi", "yalps_napi.node")

# -ffp-contract=off: the reference (V8) rounds the product and the difference of
# M[r,c] - coef*M[row,c] separately (src/simplex.ts:33); an fma would change pivot paths.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def _stale(out, *srcs):
    return not os.path.exists(out) or any(os.path.getmtime(out) < os.path.getmtime(s) for s in srcs if os.path.exists(s))


def build_hip(force=False, verbose=False):
    if not force and not _stale(LIB, HIP_SRC, HEADER, *HIP_DEPS):
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, *HIPCC_FLAGS, "-o", LIB, HIP_SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
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
