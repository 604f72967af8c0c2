"""Builds typesense_amd/libtsgpu.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libtsgpu.so")
SOURCES = ["tsgpu.hip", "tsgpu_index.hip", "tsgpu_vec.hip", "tsgpu_facet.hip"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libtsgpu.so cannot be built (there is no CPU fallback)")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if os.path.isfile(os.path.join(CSRC, f))] + [os.path.join(HERE, "..", "include", "tsgpu.h")]
    return any(os.path.getmtime(d) > t for d in deps)


LOADGEN_SRC = os.path.join(CSRC, "host", "tsgpu_loadgen.cpp")
LOADGEN_OUT = os.path.join(HERE, "libtsgpu_loadgen.so")


def build_loadgen(force=False):
    """measurement tooling (bench.py `concurrency`, tests): T native threads issuing 1-query calls through the C-ABI"""
    if force or not os.path.exists(LOADGEN_OUT) or os.path.getmtime(LOADGEN_SRC) > os.path.getmtime(LOADGEN_OUT) \
            or os.path.getmtime(os.path.join(HERE, "..", "include", "tsgpu.h")) > os.path.getmtime(LOADGEN_OUT):
        subprocess.check_call([shutil.which("g++") or "g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", LOADGEN_OUT, LOADGEN_SRC, "-lpthread"])
    return LOADGEN_OUT


def build(force=False, verbose=False):
    build_loadgen(force)
    if not force and not needs_build():
        return OUT
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-Wno-pass-failed",
           "-o", OUT] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
