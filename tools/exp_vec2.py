"""GPU experiment (not part of the product): k-NN scan timings at B = 256 (and others) on 10M x 768 for a list of library builds.
usage: TSGPU_LIBS="-,typesense_amd/variants/libtsgpu_x.so" python tools/exp_vec2.py [n_rows] [batches]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import typesense_amd as T  # noqa: E402
from typesense_amd import _lib as B, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
batches = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "256").split(",")]
dim, k = 768, 100
Q = synth.random_vectors(1024, dim, seed=4, device="cuda")
for lib in os.environ.get("TSGPU_LIBS", "-").split(","):
    g = T.GpuIndex(0, None if lib == "-" else lib)
    g.vec_create(1, dim, B.METRIC_IP, n)
    S = 1 << 20
    for a in range(0, n, S):
        b = min(n, a + S)
        x = synth.random_vectors(b - a, dim, seed=3 + a, device="cuda")
        lab = torch.arange(a, b, dtype=torch.int64, device="cuda")
        g.vec_upsert_device(1, lab.data_ptr(), x.data_ptr(), b - a)
        del x
    torch.cuda.synchronize()
    for nq in batches:
        d = torch.zeros((nq, k), dtype=torch.float32, device="cuda"); l = torch.zeros((nq, k), dtype=torch.int64, device="cuda"); c = torch.zeros(nq, dtype=torch.int32, device="cuda")
        for _ in range(2 if "abl" not in lib else 0):
            g.vec_knn_batch_raw(1, Q.data_ptr(), B.MEM_DEVICE, nq, k, d.data_ptr(), l.data_ptr(), c.data_ptr(), B.MEM_DEVICE)
        scan, knn, post = [], [], []
        t0 = time.perf_counter()
        for _ in range(5):
            try:
                g.vec_knn_batch_raw(1, Q.data_ptr(), B.MEM_DEVICE, nq, k, d.data_ptr(), l.data_ptr(), c.data_ptr(), B.MEM_DEVICE)
            except Exception as e:                        # ablation builds break exactness: the overflow rounds may give up
                print("  (", str(e)[:80], ")")
            tm = g.timings(); scan.append(tm.vec_scan_ms); knn.append(tm.vec_knn_ms); post.append(tm.vec_merge_ms)
        step = (time.perf_counter() - t0) / 5 * 1e3
        print("%s B=%d: step %.3f ms scan %.3f pre %.3f post %.3f  scan-rate %.2f TB/s  checksum %d" % (os.path.basename(lib), nq, step, np.mean(scan), np.mean(knn) - np.mean(scan), np.mean(post),
              n * dim * 2 / (np.mean(scan) * 1e-3) / 1e12, int(l.sum().item())), flush=True)
    if "abl" in lib:
        g.close(); continue
    # exactness: the prefilter path against the fp32 scan of every row (8 queries)
    nq = 8
    d1 = torch.zeros((nq, k), dtype=torch.float32, device="cuda"); l1 = torch.zeros((nq, k), dtype=torch.int64, device="cuda"); c1 = torch.zeros(nq, dtype=torch.int32, device="cuda")
    d2 = torch.zeros_like(d1); l2 = torch.zeros_like(l1)
    Q256 = Q[:256].contiguous()
    dd = torch.zeros((256, k), dtype=torch.float32, device="cuda"); ll = torch.zeros((256, k), dtype=torch.int64, device="cuda"); cc = torch.zeros(256, dtype=torch.int32, device="cuda")
    g.vec_knn_batch_raw(1, Q256.data_ptr(), B.MEM_DEVICE, 256, k, dd.data_ptr(), ll.data_ptr(), cc.data_ptr(), B.MEM_DEVICE)
    g.set_option("vec_prefilter", 0)
    g.vec_knn_batch_raw(1, Q256.data_ptr(), B.MEM_DEVICE, nq, k, d2.data_ptr(), l2.data_ptr(), c1.data_ptr(), B.MEM_DEVICE)
    print("%s exact check: labels equal %s, distance bits equal %s" % (os.path.basename(lib), bool((ll[:nq] == l2).all().item()), bool((dd[:nq].view(torch.int32) == d2.view(torch.int32)).all().item())), flush=True)
    g.close()
