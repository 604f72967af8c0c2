"""GPU experiment (not part of the product): the 256-query scan with the row operand read from global memory (option vec_rows_direct)
against the LDS-DMA form, same context, same 10M x 768 collection. Needs tools/experiments/vec_rows_direct_64x128.patch applied (the
option does not exist in the shipped library; results: profiles/r02/exp_vec_rows_direct.txt).
usage: python tools/exp_vec_direct.py [n_rows]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import typesense_amd as T  # noqa: E402
from typesense_amd import _lib as B, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim, k = 768, 100
Q = synth.random_vectors(1024, dim, seed=4, device="cuda")
g = T.GpuIndex(0)
g.vec_create(1, dim, B.METRIC_IP, n)
S = 1 << 20
for a in range(0, n, S):
    b = min(n, a + S)
    x = synth.random_vectors(b - a, dim, seed=3 + a, device="cuda")
    lab = torch.arange(a, b, dtype=torch.int64, device="cuda")
    g.vec_upsert_device(1, lab.data_ptr(), x.data_ptr(), b - a)
    del x
torch.cuda.synchronize()
ref = {}
for direct in (0, 1, 0, 1):
    g.set_option("vec_rows_direct", direct)
    for nq in (256, 1024):
        d = torch.zeros((nq, k), dtype=torch.float32, device="cuda"); l = torch.zeros((nq, k), dtype=torch.int64, device="cuda"); c = torch.zeros(nq, dtype=torch.int32, device="cuda")
        for _ in range(2):
            g.vec_knn_batch_raw(1, Q.data_ptr(), B.MEM_DEVICE, nq, k, d.data_ptr(), l.data_ptr(), c.data_ptr(), B.MEM_DEVICE)
        scan, knn = [], []
        t0 = time.perf_counter()
        for _ in range(5):
            g.vec_knn_batch_raw(1, Q.data_ptr(), B.MEM_DEVICE, nq, k, d.data_ptr(), l.data_ptr(), c.data_ptr(), B.MEM_DEVICE)
            tm = g.timings(); scan.append(tm.vec_scan_ms); knn.append(tm.vec_knn_ms)
        step = (time.perf_counter() - t0) / 5 * 1e3
        key = (l.cpu().numpy().tobytes(), d.cpu().numpy().tobytes())
        same = ref.setdefault(nq, key) == key
        print("direct=%d B=%d: step %.3f ms  scan %.3f ms (%.2f TB/s of bf16 rows)  pre+post %.3f  results identical to the first run: %s  fallbacks %d" %
              (direct, nq, step, np.mean(scan), n * dim * 2 * ((nq + 255) // 256) / (np.mean(scan) * 1e-3) / 1e12, np.mean(knn) - np.mean(scan), same, g.counter("vec_prefilter_fallbacks")), flush=True)
g.close()
