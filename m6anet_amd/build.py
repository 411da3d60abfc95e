"""Builds libm6a_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(os.path.dirname(PKG), "include")
LIB = os.path.join(PKG, "libm6a_hip.so")
IO_LIB = os.path.join(PKG, "libm6a_io.so")
SOURCES = ["m6a_kernels.hip", "m6a_pool_reg.hip", "m6a_pool_rtab.hip", "m6a_api.hip", "m6a_host_ring.hip", "m6a_job.hip", "m6a_comm.hip",
           "m6a_validate.hip"]
DEPS = SOURCES + ["m6a_kernels.h", "m6a_ctx.h", "m6a_host_cpus.h", os.path.join(INCLUDE, "m6a.h"), os.path.join(PKG, "assets", "mt19937_jump.bin")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -ffp-contract=off: the pooling arithmetic must round where NumPy rounds (a fused 1 - a*b would
    # not); the encoder's FMAs are explicit fmaf()
    jump = os.path.join(PKG, "assets", "mt19937_jump.bin")      # embedded into the library (tools/make_mt_jump.py writes it)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wextra", "-fPIC", "-shared",
           '-DM6A_MT_JUMP_PATH="%s"' % jump, "-I" + INCLUDE, "-I" + CSRC] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


def build_io(force=False, verbose=False):
    """libm6a_io.so: host-only C++ (loader + CSV writers, include/m6a_io.h)."""
    src = os.path.join(CSRC, "m6a_io.cpp")
    hdr = os.path.join(INCLUDE, "m6a_io.h")
    deps = [src, hdr, os.path.join(CSRC, "m6a_host_cpus.h")]
    if not force and os.path.exists(IO_LIB) and os.path.getmtime(IO_LIB) >= max(os.path.getmtime(d) for d in deps):
        return IO_LIB
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", "-Wextra",
           "-I" + INCLUDE, "-I" + CSRC, src, "-o", IO_LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return IO_LIB


if __name__ == "__main__":
    print(build_io(force="--force" in sys.argv, verbose=True))
    print(build(force="--force" in sys.argv, verbose=True))
