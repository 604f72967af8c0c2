"""Round-6 experiment: tsgpu_vec_hnsw_build (the bulk build on the device) at the HNSW bench leg's collection (synth.latent_vectors, 768 dims, M 16,
ef_construction 200, seed 100): build time by phase, recall@100 and q/s at ef 100 / 200 / 400 (batch 4096) against the exact scan; at the first size also
hnswlib's row-by-row insertion inside the library (tsgpu_vec_hnsw_enable) for comparison.   python tools/experiments/exp_r06_hnsw_bulk.py 300000 2000000 10000000"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import typesense_amd as T                                   # noqa: E402
from typesense_amd import _lib as B, synth                  # noqa: E402


def measure(g, field, Q, k, le_h, n):
    out = []
    for ef in (100, 200, 400):
        nq = Q.shape[0]
        d = torch.zeros((nq, k), dtype=torch.float32, device="cuda"); l = torch.zeros((nq, k), dtype=torch.int64, device="cuda"); c = torch.zeros(nq, dtype=torch.int32, device="cuda")
        for _ in range(2):
            g.vec_hnsw_search_batch_raw(field, Q.data_ptr(), B.MEM_DEVICE, nq, k, ef, d.data_ptr(), l.data_ptr(), c.data_ptr(), B.MEM_DEVICE)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(3):
            g.vec_hnsw_search_batch_raw(field, Q.data_ptr(), B.MEM_DEVICE, nq, k, ef, d.data_ptr(), l.data_ptr(), c.data_ptr(), B.MEM_DEVICE)
        torch.cuda.synchronize()
        el = (time.time() - t0) / 3
        lh = l[:256].cpu().numpy()
        rec = float(np.mean([len(set(lh[i].tolist()) & set(le_h[i].tolist())) / k for i in range(256)]))
        out.append({"ef": ef, "batch": nq, "queries_per_s": nq / el, "ms_per_batch": 1e3 * el, "recall_at_%d" % k: rec, "distances_per_query": g.counter("hnsw_last_distances") / nq})
    return out


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [300000]
    dim, k, M, efc, field = 768, 100, 16, 200, 7
    threads = int(os.environ.get("HNSW_THREADS", "16"))
    max_batch = int(os.environ.get("HNSW_MAX_BATCH", "0"))
    res = []
    for si, n in enumerate(sizes):
        X = synth.latent_vectors(n, dim, seed=3, device="cuda")
        Q = synth.latent_vectors(4096, dim, seed=4, device="cuda")
        lab = torch.arange(n, dtype=torch.int64, device="cuda")
        hows = ("bulk", "inserted") if si == 0 and n <= 400000 and not os.environ.get("HNSW_SKIP_INSERTED") else ("bulk",)
        if os.environ.get("HNSW_ONLY_INSERTED"):              # the yardstick at any size: hnswlib's row-by-row insertion inside the library (10M x 768: ~20 min on 16 host threads)
            hows = ("inserted",)
        for how in hows:
            g = T.GpuIndex(0)
            g.vec_create(field, dim, B.METRIC_IP, n)
            t0 = time.time()
            if how == "inserted":
                g.vec_hnsw_enable(field, M=M, ef_construction=efc, seed=100, threads=threads)
                for a in range(0, n, 1 << 16):
                    b = min(n, a + (1 << 16))
                    g.vec_upsert_device(field, lab[a:b].data_ptr(), X[a:b].data_ptr(), b - a)
                info = {}
            else:
                g.vec_upsert_device(field, lab.data_ptr(), X.data_ptr(), n)
                torch.cuda.synchronize()
                t0 = time.time()
                info = g.vec_hnsw_build(field, M=M, ef_construction=efc, seed=100, threads=threads, max_batch=max_batch)
            torch.cuda.synchronize()
            t_build = time.time() - t0
            de = torch.zeros((256, k), dtype=torch.float32, device="cuda"); le = torch.zeros((256, k), dtype=torch.int64, device="cuda"); ce = torch.zeros(256, dtype=torch.int32, device="cuda")
            g.vec_knn_batch_raw(field, Q.data_ptr(), B.MEM_DEVICE, 256, k, de.data_ptr(), le.data_ptr(), ce.data_ptr(), B.MEM_DEVICE)
            torch.cuda.synchronize()
            r = {"rows": n, "graph": how, "build_s": t_build, "rows_per_s": n / t_build, "info": info, "runs": measure(g, field, Q, k, le.cpu().numpy(), n)}
            if n <= 2_000_000:
                gr = g.vec_hnsw_export(field)
                cn = gr["link0"][:, 0]
                r["mean_level0_degree"] = float(cn.mean()); r["min_level0_degree"] = int(cn.min()); r["maxlevel"] = int(gr["maxlevel"])
            print(json.dumps(r), flush=True)
            res.append(r)
            g.close()
        del X, Q, lab
        torch.cuda.empty_cache()
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "exp_hnsw_bulk.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
