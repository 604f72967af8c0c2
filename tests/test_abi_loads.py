"""`not gpu`: libtsgpu.so builds for gfx950 without a GPU, loads, and exports every symbol include/tsgpu.h declares.
No compute is attempted here; creating a context without a GPU must fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from typesense_amd import build, _lib
    so = build.build()
    return _lib.lib(so), so


def test_every_declared_symbol_is_exported(lib):
    L, so = lib
    hdr = open(os.path.join(ROOT, "include", "tsgpu.h")).read()
    declared = set(re.findall(r"\b(tsgpu_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"tsgpu_status"}
    assert len(declared) >= 25
    raw = C.CDLL(so)
    missing = [s for s in sorted(declared) if not hasattr(raw, s)]
    assert not missing, missing
    assert L.tsgpu_abi_version() == 6


def test_binding_covers_the_header(lib):
    from typesense_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "tsgpu.h")).read()
    declared = set(re.findall(r"\b(tsgpu_[a-z0-9_]+)\s*\(", hdr))
    assert declared <= set(_lib.EXPORTS) | {"tsgpu_status"}, declared - set(_lib.EXPORTS)


def test_no_gpu_means_loud_failure_not_a_cpu_path(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L, _ = lib
    h = C.c_void_p()
    rc = L.tsgpu_create(0, C.byref(h))
    assert rc != 0 and not h.value
    assert b"no HIP device" in L.tsgpu_last_error() or rc == 500


def test_struct_layouts_match_the_header():
    """ctypes mirrors vs a C program compiled against include/tsgpu.h"""
    import subprocess, tempfile
    from typesense_amd import _lib as B
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "tsgpu.h"
    int main(){ printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(tsgpu_kw_query), offsetof(tsgpu_kw_query, sort),
      offsetof(tsgpu_kw_query, excluded_ids), offsetof(tsgpu_kw_query, deadline_us), sizeof(tsgpu_hits), sizeof(tsgpu_vec_query),
      sizeof(tsgpu_hybrid_params), sizeof(tsgpu_timings), offsetof(tsgpu_timings, kw_algorithmic_bytes),
      sizeof(tsgpu_group_by), offsetof(tsgpu_group_by, wildcard), sizeof(tsgpu_grouped_hits), offsetof(tsgpu_grouped_hits, loglog_registers)); return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")])
        out = list(map(int, subprocess.check_output([os.path.join(d, "t")]).split()))
    K = B.KwQueryC
    assert out == [C.sizeof(K), K.sort.offset, K.excluded_ids.offset, K.deadline_us.offset, C.sizeof(B.HitsC), C.sizeof(B.VecQueryC),
                   C.sizeof(B.HybridParamsC), C.sizeof(B.TimingsC), B.TimingsC.kw_algorithmic_bytes.offset,
                   C.sizeof(B.GroupByC), B.GroupByC.wildcard.offset, C.sizeof(B.GroupedHitsC), B.GroupedHitsC.loglog_registers.offset]
