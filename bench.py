#!/usr/bin/env python
"""bench.py — throughput of the query-time scoring hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" = one pass of the hot path over one batch of synthetic queries:
  * workload keyword (default, BASELINE config 2): 10M-doc Zipf collection (V=100K, 32 tokens/doc, seed 2),
    a batch of 3-term conjunctive queries (ranks log-uniform [8,2000]), Topster 250 (per_page 100), sort
    [_text_match desc, points desc]; N>1 = the collection split into N contiguous seq_id ranges (doc-range shards),
    every rank scores the whole batch on its shard, RCCL all-gather of per-GPU top-K, exact merge.
  * --workload vector (config 3): 10M x 768 fp32, batched exact inner-product top-100.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, HIP-event timed
inside the library on its launch stream) and, at N=1, `cpu_baseline` (the oracle = a port of the reference's CPU
path, timed on this box's host cores on a bounded sample of the same queries and also used as a parity check).
The oracle is never the thing measured as `value`.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3     # fp32-input MFMA dense peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="keyword", choices=["keyword", "vector"])
    ap.add_argument("--n-docs", type=int, default=10_000_000)
    ap.add_argument("--batch", type=int, default=0, help="queries per step (default 10000 keyword / 256 vector)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="queries of the CPU-baseline sample (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=100)
    return ap.parse_args()


def dist_setup(n_gpus):
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(local)
    assert world == n_gpus, "launch with torch.distributed.run --nproc-per-node %d" % n_gpus
    return rank, world, local


def barrier(world):
    import torch
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world):
    import torch
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ------------------------------------------------------------------------------------------------ keyword
def device_hits(torch, T, n_q, ks):
    from typesense_amd import _lib as B
    d = dict(keys=torch.zeros((n_q, ks), dtype=torch.int64, device="cuda"),
             scores=torch.zeros((n_q, ks, 3), dtype=torch.int64, device="cuda"),
             text_match=torch.zeros((n_q, ks), dtype=torch.int64, device="cuda"),
             vector_distance=torch.zeros((n_q, ks), dtype=torch.float32, device="cuda"),
             match_score_index=torch.zeros((n_q, ks), dtype=torch.int8, device="cuda"),
             n_hits=torch.zeros(n_q, dtype=torch.int32, device="cuda"),
             num_matched=torch.zeros(n_q, dtype=torch.int64, device="cuda"),
             status=torch.zeros(n_q, dtype=torch.int32, device="cuda"),
             search_cutoff=torch.zeros(n_q, dtype=torch.int32, device="cuda"))
    h = B.HitsC()
    h.mem = B.MEM_DEVICE
    h.k_stride = ks
    for k, v in d.items():
        setattr(h, k, v.data_ptr())
    return d, h


def merge_shards_device(torch, gathered, k):
    """exact G-way merge of per-shard Topster lists on the GPU: order = (s0, s1, s2, key) descending
    (include/topster.h:146-149). gathered: dict of [G, B, K, ...] tensors."""
    G, Bq, K = gathered["keys"].shape
    keys = gathered["keys"].permute(1, 0, 2).reshape(Bq, G * K)
    sc = gathered["scores"].permute(1, 0, 2, 3).reshape(Bq, G * K, 3)
    nh = gathered["n_hits"].permute(1, 0)                                    # [B, G]
    valid = (torch.arange(K, device=keys.device)[None, None, :] < nh[:, :, None]).reshape(Bq, G * K)
    order = torch.arange(G * K, device=keys.device)[None, :].expand(Bq, -1)
    # successive stable sorts, least significant key first; invalid slots last
    for col in (keys, sc[..., 2], sc[..., 1], sc[..., 0]):
        v = torch.gather(col, 1, order)
        idx = torch.sort(v, dim=1, descending=True, stable=True).indices
        order = torch.gather(order, 1, idx)
    v = torch.gather(valid.to(torch.int8), 1, order)
    idx = torch.sort(v, dim=1, descending=True, stable=True).indices
    order = torch.gather(order, 1, idx)[:, :k]
    out_keys = torch.gather(keys, 1, order)
    out_sc = torch.gather(sc, 1, order[:, :, None].expand(-1, -1, 3))
    n_out = torch.clamp(nh.sum(1), max=k)
    return out_keys, out_sc, n_out


def run_keyword(args, rank, world):
    import torch
    import typesense_amd as T
    from typesense_amd import _lib as B, synth

    n_docs = args.n_docs
    vocab, tpd = (100_000, 32) if n_docs >= 1_000_000 else (20_000, 16)
    n_q = args.batch or 10_000
    K = 250
    lo = n_docs * rank // world
    hi = n_docs * (rank + 1) // world
    t0 = time.time()
    # every rank draws the SAME collection slice-by-slice: shard r holds docs [lo, hi) (seed derived per shard so
    # that shards are i.i.d. like the unsharded collection; N=1 reproduces seed 2 exactly)
    csr = synth.zipf_corpus_csr(hi - lo, vocab, tpd, seed=2 + 1000 * rank if world > 1 else 2, doc_base=lo)
    pts_all = synth.points_column(n_docs)
    g = T.GpuIndex(torch.cuda.current_device())
    g.field_create(0, False)
    g.terms_load_csr(0, csr["term_ids"], csr["ids_ptr"], csr["ids"], csr["offset_index"], csr["off_ptr"], csr["offsets"])
    g.column_set(0, pts_all)
    g.set_num_docs(n_docs)
    g.commit()
    t_build = time.time() - t0

    qtok = synth.keyword_queries(n_q, 3, 8, 2000, seed=4)
    arr = (B.KwQueryC * n_q)()
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    for i in range(n_q):
        T.KwQuery(qtok[i], sort=sort, topster_size=K).fill(arr[i])

    dev, hs = device_hits(torch, T, n_q, K)
    if world > 1:
        import torch.distributed as dist
        gath = {k: torch.zeros((world,) + tuple(dev[k].shape), dtype=dev[k].dtype, device="cuda") for k in ("keys", "scores", "n_hits", "num_matched")}

    kern_ms, merge_ms, alg_bytes = [], [], []

    def step():
        g.keyword_search_batch_raw(arr, n_q, hs)           # synchronises its stream before returning
        if world > 1:
            for k in ("keys", "scores", "n_hits", "num_matched"):
                dist.all_gather_into_tensor(gath[k], dev[k])
            return merge_shards_device(torch, gath, K)
        return dev["keys"], dev["scores"], dev["n_hits"]

    for _ in range(args.warmup):
        step()
    barrier(world)
    lat = []
    t_start = time.perf_counter()
    for _ in range(args.steps):
        s0 = time.perf_counter()
        out = step()
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - s0)
        tm = g.timings()
        kern_ms.append(tm.kw_search_ms)
        merge_ms.append(tm.kw_merge_ms)
        alg_bytes.append(tm.kw_algorithmic_bytes)
    barrier(world)
    elapsed = max_over_ranks(time.perf_counter() - t_start, world)

    res = dict(elapsed=elapsed, lat=lat, kern_ms=float(np.mean(kern_ms)), merge_ms=float(np.mean(merge_ms)),
               alg_bytes=float(np.mean(alg_bytes)), n_q=n_q, t_build=t_build, n_postings=int(csr["n_postings"]))

    # host copies for the parity check / CPU baseline (untimed)
    keys = out[0].cpu().numpy().astype(np.uint64)
    scores = out[1].cpu().numpy()
    n_hits = out[2].cpu().numpy()
    num_matched = dev["num_matched"].cpu().numpy()
    res["nonempty"] = int((n_hits > 0).sum())

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle_py as O
        ncpu = os.cpu_count() or 1
        sample = args.cpu_sample or max(2 * ncpu, 32)
        sample = min(sample, n_q)
        orc = O.OracleIndex(1, 1)
        orc.set_num_docs(n_docs)
        orc.set_sort_dense(0, pts_all)
        for t in np.unique(qtok[:sample]):
            ids, oi, off = synth.csr_term(csr, t)
            if ids.size:
                orc.load_posting(0, int(t), ids, oi, off)
        base = orc.make_query(qtok[0], sort=((O.SORT_TEXT_MATCH, 0, 1), (O.SORT_INT64_COLUMN, 0, 1)), fetch_size=100)
        orc.bench_keyword(base, qtok[:min(sample, ncpu)], ncpu)               # warm the page cache / allocator
        wall, per = orc.bench_keyword(base, qtok[:sample], ncpu)
        res["cpu"] = dict(value=sample / wall, unit="queries/s", cores=ncpu, kind="port",
                          sample="%d of the %d queries of the step, one query per thread on %d host threads (oracle = port of "
                                 "or_iterator_t::intersect + Match + Topster); p50 %.1f ms/query" % (sample, n_q, ncpu, float(np.median(per)) / 1e3))
        # parity at full size on the sample: identical top-K (keys + all 3 scores) and match counts
        bad = 0
        for i in range(min(sample, 64)):
            oq = orc.make_query(qtok[i], sort=((O.SORT_TEXT_MATCH, 0, 1), (O.SORT_INT64_COLUMN, 0, 1)), fetch_size=100)
            ref = orc.search_keyword(oq)
            n = int(n_hits[i])
            if n != ref.keys.size or not np.array_equal(keys[i, :n], ref.keys) or not np.array_equal(scores[i, :n], ref.scores) \
                    or int(num_matched[i]) != int(ref.num_keyword_matches):
                bad += 1
        res["parity_checked"] = min(sample, 64)
        res["parity_bad"] = bad
    g.close()
    return res


# ------------------------------------------------------------------------------------------------ vector
def run_vector(args, rank, world):
    import torch
    import typesense_amd as T
    from typesense_amd import _lib as B, synth

    n, dim, k = args.n_docs, args.dim, args.k
    n_q = args.batch or 256
    lo = n * rank // world
    hi = n * (rank + 1) // world
    t0 = time.time()
    g = T.GpuIndex(torch.cuda.current_device())
    g.vec_create(1, dim, B.METRIC_IP, hi - lo)
    slab = 1 << 20
    for a in range(lo, hi, slab):                     # base vectors are generated on the device they live on
        b = min(hi, a + slab)
        x = synth.random_vectors(b - a, dim, seed=3 + a, device="cuda")
        labels = torch.arange(a, b, dtype=torch.int64, device="cuda")
        g.vec_upsert_device(1, labels.data_ptr(), x.data_ptr(), b - a)
        del x
    torch.cuda.synchronize()
    t_build = time.time() - t0
    Q = synth.random_vectors(n_q, dim, seed=4, device="cuda")
    dist_o = torch.zeros((n_q, k), dtype=torch.float32, device="cuda")
    lab_o = torch.zeros((n_q, k), dtype=torch.int64, device="cuda")
    cnt_o = torch.zeros(n_q, dtype=torch.int32, device="cuda")
    if world > 1:
        import torch.distributed as dist
        gd = torch.zeros((world, n_q, k), dtype=torch.float32, device="cuda")
        gl = torch.zeros((world, n_q, k), dtype=torch.int64, device="cuda")

    def step():
        g.vec_knn_batch_raw(1, Q.data_ptr(), B.MEM_DEVICE, n_q, k, dist_o.data_ptr(), lab_o.data_ptr(), cnt_o.data_ptr(), B.MEM_DEVICE)
        if world > 1:
            dist.all_gather_into_tensor(gd, dist_o)
            dist.all_gather_into_tensor(gl, lab_o)
            d = gd.permute(1, 0, 2).reshape(n_q, world * k)
            l = gl.permute(1, 0, 2).reshape(n_q, world * k)
            o = torch.sort(l, dim=1, stable=True).indices                      # ties: smaller label first
            d, l = torch.gather(d, 1, o), torch.gather(l, 1, o)
            o = torch.sort(d, dim=1, stable=True).indices[:, :k]
            return torch.gather(d, 1, o), torch.gather(l, 1, o)
        return dist_o, lab_o

    for _ in range(args.warmup):
        step()
    barrier(world)
    lat, kern_ms, flops = [], [], []
    t_start = time.perf_counter()
    for _ in range(args.steps):
        s0 = time.perf_counter()
        out = step()
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - s0)
        tm = g.timings()
        kern_ms.append(tm.vec_knn_ms)
        flops.append(tm.vec_flops)
    barrier(world)
    elapsed = max_over_ranks(time.perf_counter() - t_start, world)
    res = dict(elapsed=elapsed, lat=lat, kern_ms=float(np.mean(kern_ms)), flops=float(np.mean(flops)), n_q=n_q, t_build=t_build)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle_py as O
        ncpu = os.cpu_count() or 1
        ns = min(n, 400_000)                           # bounded sample of the base: rows [0, ns)
        orc = O.OracleIndex(1, 1)
        orc.vec_init(dim, O.METRIC_IP)
        xs = synth.random_vectors(min(slab, n), dim, seed=3, device="cuda")[:ns].cpu().numpy()
        orc.vec_add(np.arange(ns, dtype=np.uint32), xs)
        qs = Q[:max(ncpu, 8)].cpu().numpy()
        orc.bench_vector(qs[:ncpu], k, ncpu)
        wall, per = orc.bench_vector(qs, k, ncpu)
        qps_sample = qs.shape[0] / wall
        res["cpu"] = dict(value=qps_sample * ns / n, unit="queries/s", cores=ncpu, kind="port",
                          sample="exact flat scan (1 - q.x, hnswlib 16-lane order) of %d queries over the first %d of %d base vectors on %d "
                                 "host threads, %.1f q/s on the sample, scaled by %d/%d (cost is linear in N)" % (qs.shape[0], ns, n, ncpu, qps_sample, ns, n))
        # parity on the sample rows: distances of the GPU's hits that fall in [0, ns) must match the oracle's
        d_gpu, l_gpu = out[0].cpu().numpy(), out[1].cpu().numpy()
        bad = 0
        for i in range(min(8, qs.shape[0])):
            for j in range(k):
                if l_gpu[i, j] < ns:
                    ref = float(np.float32(1.0) - np.dot(qs[i].astype(np.float64), xs[l_gpu[i, j]].astype(np.float64)))
                    if abs(ref - d_gpu[i, j]) > 1e-5 * max(1.0, abs(ref)):
                        bad += 1
        res["parity_bad"] = bad
    g.close()
    return res


def main():
    args = parse()
    import torch
    if not torch.cuda.is_available():
        print(json.dumps({"error": "no GPU visible: bench.py measures the HIP path only (there is no CPU fallback)"}))
        sys.exit(2)
    rank, world, _ = dist_setup(args.gpus)
    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    barrier(world)

    if args.workload == "keyword":
        r = run_keyword(args, rank, world)
        qps = r["n_q"] * args.steps / r["elapsed"]
        achieved = r["alg_bytes"] / (r["kern_ms"] * 1e-3) / 1e9 if r["kern_ms"] > 0 else 0.0
        line = {
            "metric": "queries/sec, 10M-doc keyword 3-term AND top-100 (Topster 250)", "value": qps, "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * r["elapsed"] / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32/i64", "data": "synthetic",
            "config": {"workload": "BASELINE config 2: %d-doc Zipf(1.0) text, V=%d, %d tokens/doc, %d postings/shard; %d queries/step, 3 distinct "
                                   "terms ranks log-uniform [8,2000], sort [_text_match desc, points desc], num_typos=0, prefix=false"
                                   % (args.n_docs, 100_000 if args.n_docs >= 1_000_000 else 20_000, 32 if args.n_docs >= 1_000_000 else 16,
                                      r["n_postings"], r["n_q"]),
                       "parallelism": "doc-range shards x%d, RCCL all-gather of per-GPU top-250 + exact merge" % world if world > 1 else "1 GPU",
                       "results_to": "device (tsgpu_hits mem=DEVICE); host delivery is measured in DESIGN.md"},
            "p50_ms_per_batch": 1e3 * float(np.median(r["lat"])),
            "queries_with_hits": r.get("nonempty"),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "kernel": "kw_search_kernel<3,512>", "kernel_ms": r["kern_ms"], "merge_kernel_ms": r["merge_ms"],
                         "algorithmic_bytes_per_launch": r["alg_bytes"]},
            "index_build_s": r["t_build"],
        }
    else:
        r = run_vector(args, rank, world)
        qps = r["n_q"] * args.steps / r["elapsed"]
        tf = r["flops"] / (r["kern_ms"] * 1e-3) / 1e12 if r["kern_ms"] > 0 else 0.0
        line = {
            "metric": "queries/sec, 10M x 768 fp32 exact inner-product top-100", "value": qps, "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * r["elapsed"] / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE config 3: %d x %d fp32 N(0,1) base, %d queries/step, k=%d, dist = 1 - q.x" % (args.n_docs, args.dim, r["n_q"], args.k),
                       "parallelism": "row-range shards x%d, RCCL all-gather of per-GPU top-k + merge" % world if world > 1 else "1 GPU"},
            "p50_ms_per_batch": 1e3 * float(np.median(r["lat"])),
            "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TF,
                         "traffic": None, "kernel": "vec_knn_kernel", "kernel_ms": r["kern_ms"], "flops_per_launch": r["flops"]},
            "index_build_s": r["t_build"],
        }
    if "cpu" in r:
        line["cpu_baseline"] = r["cpu"]
        line["speedup_vs_cpu_baseline"] = line["value"] / r["cpu"]["value"] if r["cpu"]["value"] else None
    if "parity_bad" in r:
        line["parity"] = {"checked": r.get("parity_checked"), "mismatches": r["parity_bad"]}
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
