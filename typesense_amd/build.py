"""Builds typesense_amd/libtsgpu.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libtsgpu.so")
SOURCES = ["tsgpu.hip", "tsgpu_index.hip", "tsgpu_vec.hip", "tsgpu_facet.hip", "tsgpu_group.hip"]


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


def build(force=False, verbose=False, extra_flags=(), out=None, only_kw=False, only=None):
    """one object per translation unit (compiled in parallel, kept under typesense_amd/_obj), then one link.
    extra_flags/out: variant builds for tools/ experiments (-D knobs); only_kw / only=<source>: the flags touch that translation unit only (default tsgpu.hip), the other objects are reused"""
    build_loadgen(force)
    out = out or OUT
    if not force and not extra_flags and not needs_build():
        return out
    tag = "default" if not extra_flags else "v_" + "_".join(f.lstrip("-D").replace("=", "-") for f in extra_flags)
    objdir = os.path.join(HERE, "_obj", tag)
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "tsgpu.h")]
    hdr_t = max(os.path.getmtime(h) for h in headers)
    base = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed"]
    procs, objs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        flags = list(extra_flags) if ((only or "tsgpu.hip") == s or not (only_kw or only)) else []
        obj = os.path.join(objdir if flags or not extra_flags else os.path.join(HERE, "_obj", "default"), s + ".o")
        os.makedirs(os.path.dirname(obj), exist_ok=True)
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(hdr_t, os.path.getmtime(src)):
            cmd = base + flags + ["-c", "-o", obj, src]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
