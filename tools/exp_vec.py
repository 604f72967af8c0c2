"""GPU experiment (not part of the product): the k-NN scan at B = 256 on 10M x 768 — ablation builds (VEC_ABL bit 0 no epilogue, bit 1 no
MFMA, bit 2 no DMA), sample-pass sizes, per-phase timings. Usage: python tools/exp_vec.py [n_rows]"""
import glob
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import typesense_amd as T  # noqa: E402
from typesense_amd import _lib as B, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim, k = 768, 100
libs = [None] + sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(T.__file__)), "abl", "libtsgpu_abl*.so")))
Q = synth.random_vectors(1024, dim, seed=4, device="cuda")


def run(g, nq, steps=5):
    d = torch.zeros((nq, k), dtype=torch.float32, device="cuda")
    l = torch.zeros((nq, k), dtype=torch.int64, device="cuda")
    c = torch.zeros(nq, dtype=torch.int32, device="cuda")
    g.vec_knn_batch_raw(1, Q.data_ptr(), B.MEM_DEVICE, nq, k, d.data_ptr(), l.data_ptr(), c.data_ptr(), B.MEM_DEVICE)
    t0 = time.perf_counter()
    scan, knn, post = [], [], []
    for _ in range(steps):
        g.vec_knn_batch_raw(1, Q.data_ptr(), B.MEM_DEVICE, nq, k, d.data_ptr(), l.data_ptr(), c.data_ptr(), B.MEM_DEVICE)
        tm = g.timings()
        scan.append(tm.vec_scan_ms); knn.append(tm.vec_knn_ms); post.append(tm.vec_merge_ms)
    return (time.perf_counter() - t0) / steps * 1e3, float(np.mean(scan)), float(np.mean(knn)), float(np.mean(post))


for lib in libs:
    g = T.GpuIndex(0, lib)
    g.vec_create(1, dim, B.METRIC_IP, n)
    S = 1 << 20
    for a in range(0, n, S):
        b = min(n, a + S)
        x = synth.random_vectors(b - a, dim, seed=3 + a, device="cuda")
        lab = torch.arange(a, b, dtype=torch.int64, device="cuda")
        g.vec_upsert_device(1, lab.data_ptr(), x.data_ptr(), b - a)
        del x
    torch.cuda.synchronize()
    name = os.path.basename(lib) if lib else "product"
    if lib is None:
        for st in (1024, 2048, 4096, 8192):
            g.set_option("vec_sample_tiles", st)
            g.set_option("vec_count_rescored", 1)
            step, scan, knn, post = run(g, 256)
            print("%s sample_tiles %5d: step %.3f ms scan %.3f pre %.3f post %.3f | rescored/query %.0f fallbacks %d overflow %d" %
                  (name, st, step, scan, knn - scan, post, g.counter("vec_rescored_rows") / 256, g.counter("vec_prefilter_fallbacks"), g.counter("vec_overflow_rounds")), flush=True)
        g.set_option("vec_sample_tiles", 0)
        g.set_option("vec_count_rescored", 0)
    for nq in (64, 256):
        try:
            step, scan, knn, post = run(g, nq, 3)
            print("%s B=%d: step %.3f ms scan %.3f pre %.3f post %.3f" % (name, nq, step, scan, knn - scan, post), flush=True)
        except Exception as e:  # ablation builds break exactness: overflow loops may fail
            print("%s B=%d: %s" % (name, nq, str(e)[:100]), flush=True)
    g.close()
