"""Shard-sized rehearsal on ONE GPU: what a rank of a G-way doc-range sharding of BASELINE config 2 costs per
10 000-query step — its local pass on 1/G of the 10M-doc collection, the pack, and the merge of G gathered blocks — so that the
2/4/8-GPU curve can be predicted before the driver measures it. The all-gather itself cannot be rehearsed on one GPU: its time is
modelled as bytes / bandwidth and stated separately.
usage: python tools/exp_shard_rehearsal.py [--n-docs 10000000] [--batch 10000]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-docs", type=int, default=10_000_000)
    ap.add_argument("--batch", type=int, default=10_000)
    ap.add_argument("--steps", type=int, default=8)
    args = ap.parse_args()
    import torch
    torch.zeros(1, device="cuda")
    import typesense_amd as T
    from typesense_amd import _lib as B, synth
    from bench import device_hits
    n_docs, n_q = args.n_docs, args.batch
    vocab, tpd = (100_000, 32) if n_docs >= 1_000_000 else (20_000, 16)
    pts = synth.points_column(n_docs)
    qtok = synth.keyword_queries(n_q, 3, 8, 2000, seed=4)
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    arr = (B.KwQueryC * n_q)()
    for i in range(n_q):
        T.KwQuery(qtok[i], sort=sort, topster_size=250).fill(arr[i])
    out = {"n_docs": n_docs, "batch": n_q, "shards": {}}
    for G in (1, 2, 4, 8):
        lo, hi = 0, n_docs // G
        csr = synth.zipf_corpus_csr(n_docs, vocab, tpd, seed=2, doc_range=(lo, hi) if G > 1 else None)
        members = []
        for _ in range(1 if G == 1 else 2):          # two members sharing the device are enough to exercise pack + exchange copy + merge of G blocks below
            g = T.GpuIndex(0)
            g.field_create(0, False)
            g.terms_load_csr(0, csr["term_ids"], csr["ids_ptr"], csr["ids"], csr["offset_index"], csr["off_ptr"], csr["offsets"])
            g.column_set(0, pts)
            g.set_num_docs(n_docs)
            g.commit()
            members.append(g)
        g = members[0]
        dev, hs = device_hits(torch, n_q, 250)
        for _ in range(3):
            g.keyword_search_batch_raw(arr, n_q, hs)
        t0 = time.perf_counter()
        kern, merge, find = [], [], []
        for _ in range(args.steps):
            g.keyword_search_batch_raw(arr, n_q, hs)
            tm = g.timings()
            kern.append(tm.kw_search_ms); merge.append(tm.kw_merge_ms); find.append(tm.kw_find_ms)
        local_ms = 1e3 * (time.perf_counter() - t0) / args.steps
        rec = {"local_step_ms": local_ms, "find_plus_score_kernel_ms": float(np.mean(kern)), "find_kernel_ms": float(np.mean(find)), "partial_merge_kernel_ms": float(np.mean(merge)),
               "host_ms (plan + launch + sync)": local_ms - float(np.mean(kern)) - float(np.mean(merge)), "postings": int(csr["n_postings"])}
        if G > 1:
            # pack + the merge of G gathered blocks: a COPY group of G members is emulated by gathering member 0's block G times — the merge
            # kernel's cost depends on G * k entries per query, not on whose they are
            grp = T.GpuGroup(members + [members[1]] * (G - 2), B.XCHG_COPY) if G > 2 else T.GpuGroup(members, B.XCHG_COPY)
            gdev, ghs = device_hits(torch, n_q, 100)
            for _ in range(2):
                grp.keyword_search_batch_raw(arr, n_q, 100, ghs)
            ex = []
            for _ in range(4):
                grp.keyword_search_batch_raw(arr, n_q, 100, ghs)
                ex.append(grp.timings().exchange_merge_ms)
            rec["slices: copies + G slice merges + replication, ALL on this one device (ms)"] = float(np.mean(ex))
            rec["slices: bytes received per GPU"] = int(grp.timings().exchange_bytes_per_member)
            grp.set_option("kw_exchange_slices", 0)
            for _ in range(2):
                grp.keyword_search_batch_raw(arr, n_q, 100, ghs)
            ex = []
            for _ in range(4):
                grp.keyword_search_batch_raw(arr, n_q, 100, ghs)
                ex.append(grp.timings().exchange_merge_ms)
            rec["all-gather form: G block copies + full merge on one device (ms)"] = float(np.mean(ex))
            rec["all-gather form: bytes received per GPU"] = int(grp.timings().exchange_bytes_per_member)
            grp.close()
        out["shards"][str(G)] = rec
        for m in members:
            m.close()
        print(json.dumps({str(G): rec}), flush=True)
    base = out["shards"]["1"]["local_step_ms"]
    pred = {}
    for G in (2, 4, 8):
        r = out["shards"][str(G)]
        # wire time over xGMI modelled at 150 GB/s (conservative) and 300 GB/s (RCCL bus bandwidth on a full mesh) received per GPU
        for form, key_ms, key_b, div in (("slices (all-to-all + slice merge + all-gather of merged lists)", "slices: copies + G slice merges + replication, ALL on this one device (ms)",
                                          "slices: bytes received per GPU", G),
                                         ("one all-gather + full merge", "all-gather form: G block copies + full merge on one device (ms)", "all-gather form: bytes received per GPU", 1)):
            wire = [1e3 * r[key_b] / bw for bw in (150e9, 300e9)]
            dev_ms = r[key_ms] / div            # the slice form's device work is spread over the G GPUs
            step = [r["local_step_ms"] + dev_ms + w for w in wire]
            pred.setdefault(str(G), {})[form] = {"wire_ms@150GB/s,300GB/s": wire, "pack+merge_ms_per_gpu": dev_ms, "predicted_step_ms": step,
                                                 "predicted_speedup_vs_1gpu": [base / x for x in step], "predicted_qps": [n_q / (x * 1e-3) for x in step]}
    out["predicted"] = pred
    print(json.dumps(out))


if __name__ == "__main__":
    main()
