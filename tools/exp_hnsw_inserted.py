"""HNSW on an hnswlib-SHAPED graph: the graph is built by the oracle's restatement of hnswlib's addPoint, rows inserted
in label order, random_seed 100, M 16, ef_construction 200 (include/index.h:365-367) — not the GPU-built knn-heuristic graph of bench.py's
hnsw leg — then mirrored (tsgpu_vec_hnsw_load) and searched at ef = 10 (the reference's default, include/vector_query_ops.h:21), 100, 400:
q/s, recall@k against the exact scan, and the GPU traversal compared bit for bit with the oracle's traversal of the same graph.
The same rows also get the knn-heuristic graph, for the side-by-side. PARITY UNPINNED: hnswlib itself is not under /root/reference.
usage: python tools/exp_hnsw_inserted.py [--rows 100000] [--dim 768] [--batch 4096]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--k", type=int, default=100)
    args = ap.parse_args()
    import torch
    torch.zeros(1, device="cuda")
    import typesense_amd as T
    from typesense_amd import _lib as B, synth, hnsw_synth
    from oracle import oracle_py as O
    n, dim, k = args.rows, args.dim, args.k
    X = synth.latent_vectors(n, dim, seed=3, device="cuda")
    Xh = X.cpu().numpy()
    g = T.GpuIndex(0)
    out = {"rows": n, "dim": dim, "k": k, "graphs": {}}
    nq = max(args.batch, 256)
    Q = synth.latent_vectors(nq, dim, seed=4, device="cuda")
    Qh = Q.cpu().numpy()
    lab = torch.arange(n, dtype=torch.int64, device="cuda")
    orc = O.OracleIndex(1, 1)
    orc.vec_init(dim, O.METRIC_IP)
    orc.vec_add(np.arange(n, dtype=np.uint32), Xh)
    t0 = time.time()
    orc.hnsw_build(M=16, ef_construction=200, seed=100)
    t_build = time.time() - t0
    inserted = orc.hnsw_export()
    for name, field in (("hnswlib-shaped (oracle addPoint, insertion order, seed 100, M 16, ef_construction 200)", 1), ("knn-heuristic (exact 64-NN + getNeighborsByHeuristic2, GPU-built)", 2)):
        g.vec_create(field, dim, B.METRIC_IP, n)
        g.vec_upsert_device(field, lab.data_ptr(), X.data_ptr(), n)
        torch.cuda.synchronize()
        if field == 1:
            graph, build_s = inserted, t_build
        else:
            t1 = time.time()
            graph = hnsw_synth.build_graph(torch, g, field, X, M=16, K0=64, seed=100, batch=1024)
            torch.cuda.synchronize()
            build_s = time.time() - t1
        g.vec_hnsw_load(field, graph)
        de, le, ce = g.vec_knn_batch(field, Qh[:256], k)
        rec = {"build_s": build_s, "maxlevel": int(graph["maxlevel"]), "mean_level0_degree": float(np.asarray(graph["link0"])[:, 0].mean()), "runs": []}
        for ef in (10, 100, 400):
            for b in sorted({256, nq}):
                d = torch.zeros((b, k), dtype=torch.float32, device="cuda"); l = torch.zeros((b, k), dtype=torch.int64, device="cuda"); c = torch.zeros(b, dtype=torch.int32, device="cuda")
                for _ in range(2):
                    g.vec_hnsw_search_batch_raw(field, Q.data_ptr(), B.MEM_DEVICE, b, k, ef, d.data_ptr(), l.data_ptr(), c.data_ptr(), B.MEM_DEVICE)
                t1 = time.perf_counter()
                for _ in range(5):
                    g.vec_hnsw_search_batch_raw(field, Q.data_ptr(), B.MEM_DEVICE, b, k, ef, d.data_ptr(), l.data_ptr(), c.data_ptr(), B.MEM_DEVICE)
                el = time.perf_counter() - t1
                lh = l[:256].cpu().numpy()
                recall = float(np.mean([len(set(lh[i].tolist()) & set(le[i].tolist())) / k for i in range(256)]))
                rec["runs"].append({"ef": ef, "batch": b, "queries_per_s": b * 5 / el, "ms_per_batch": 1e3 * el / 5, "recall_at_%d" % k: recall,
                                    "expansions_per_query": g.counter("hnsw_last_expansions") / b, "distances_per_query": g.counter("hnsw_last_distances") / b})
        if field == 1:
            # the GPU replays the oracle's traversal of the SAME graph: labels, order, distance bits
            bad = 0
            for ef in (10, 100, 400):
                dd, ll, cc = g.vec_hnsw_search_batch(field, Qh[:64], k, ef)
                for i in range(64):
                    od, ol, _ = orc.hnsw_search(Qh[i], k, ef)
                    if cc[i] != od.size or not np.array_equal(ll[i, :od.size], ol) or not np.array_equal(dd[i, :od.size].view(np.uint32), od.view(np.uint32)):
                        bad += 1
            rec["parity"] = {"queries_checked": 3 * 64, "mismatches": bad, "pinned": False, "what": "labels, order, distance bits vs the oracle's traversal of the same graph"}
            t1 = time.time()
            orc.hnsw_search_batch(Qh[:512], k, 100, threads=os.cpu_count() or 1)
            rec["cpu_oracle_traversal_q_per_s (ef 100, all host threads)"] = 512 / (time.time() - t1)
        out["graphs"][name] = rec
        print(json.dumps({name: rec}), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
