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
        # G DISTINCT doc-range shards, all resident on this one device (round 5: the bound-pruned exchange depends on how the winners spread over
        # the shards — G copies of one shard would prune nothing). Member 0's local step is timed alone; the group call then runs every member's
        # pack / bound / count / pruned-pack kernels, the slice copies and the G slice merges on the ONE device: that time / G = one GPU's share.
        members, postings = [], 0
        for i in range(G):
            lo, hi = i * (n_docs // G), (i + 1) * (n_docs // G) if i + 1 < G else n_docs
            csr = synth.zipf_corpus_csr(n_docs, vocab, tpd, seed=2, doc_range=(lo, hi) if G > 1 else None)
            g = T.GpuIndex(0)
            g.field_create(0, False)
            g.terms_load_csr(0, csr["term_ids"], csr["ids_ptr"], csr["ids"], csr["offset_index"], csr["off_ptr"], csr["offsets"])
            g.column_set(0, pts)
            g.set_num_docs(n_docs)
            g.commit()
            members.append(g)
            if i == 0:
                postings = int(csr["n_postings"])
            del csr
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
               "host_ms (plan + launch + sync)": local_ms - float(np.mean(kern)) - float(np.mean(merge)), "postings": postings}
        if G > 1:
            grp = T.GpuGroup(members, B.XCHG_COPY)
            gdev, ghs = device_hits(torch, n_q, 100)
            for form, slices in (("slices", 1), ("all-gather form", 0)):
                grp.set_option("kw_exchange_slices", slices)
                for pruned in (1, 0):
                    grp.set_option("kw_exchange_pruned", pruned)
                    for _ in range(2):
                        grp.keyword_search_batch_raw(arr, n_q, 100, ghs)
                    ex, loc, kn = [], [], []
                    for _ in range(7):
                        grp.keyword_search_batch_raw(arr, n_q, 100, ghs)
                        tmg = grp.timings()
                        ex.append(tmg.exchange_merge_ms); loc.append(tmg.local_ms); kn.append(tmg.exchange_kernels_ms)
                    tag = "%s, %s" % (form, "bound-pruned" if pruned else "full top-k")
                    rec[tag + ": exchange kernels + copies + merges of ALL %d members on this one device (ms)" % G] = float(np.median(ex))
                    rec[tag + ": ONE member's exchange kernels, HIP events (pack or bounds, count, pruned pack, its merge) (ms)"] = float(np.median(kn))     # (median: a call that re-allocates a staging buffer is an outlier)
                    rec[tag + ": hit bytes received per GPU (bounds + slices / blocks)"] = int(tmg.hit_exchange_bytes_per_member)
                    rec[tag + ": all bytes received per GPU (+ replication of the merged lists)"] = int(tmg.exchange_bytes_per_member)
            grp.close()
        out["shards"][str(G)] = rec
        for m in members:
            m.close()
        print(json.dumps({str(G): rec}), flush=True)
    base = out["shards"]["1"]["local_step_ms"]
    pred = {}
    for G in (2, 4, 8):
        r = out["shards"][str(G)]
        # MODEL, every term listed: step(G) = local step of a 1/G shard (measured alone: kernels + per-batch host work)
        #                                   + ONE member's exchange kernels (HIP events on its stream: pack or bounds, count, pruned pack, the merge of its slice / of everything)
        #                                   + 30 us per collective call (bounds all-gather, totals all-gather, slice exchange: 3 pruned, 1 full)
        #                                   + wire time = bytes received per GPU / bandwidth, at 150 GB/s (one xGMI link, conservative) and 300 GB/s (several links busy)
        # own-slice delivery (rank form, option kw_own_slice_only: what bench.py --gpus N times): the replication of the merged lists is not sent -> hit bytes only.
        for pruned in ("bound-pruned", "full top-k"):
            for form, div in (("slices", G), ("all-gather form", 1)):
                tag = "%s, %s" % (form, pruned)
                # one member's own kernels (HIP events on its stream; in the slice form it merges only its slice). The wall time of all members'
                # work on the one device (also recorded) is dominated by the COPY transport's G x G host-issued copies, which a rank of an RCCL group
                # replaces by one collective call each: COLLECTIVE_LAUNCHES x 30 us are added for those instead
                n_coll = (3 if pruned == "bound-pruned" else 1) + (0 if form != "slices" else 0)
                dev_ms = r[tag + ": ONE member's exchange kernels, HIP events (pack or bounds, count, pruned pack, its merge) (ms)"] + 0.03 * n_coll
                for deliver, key_b in (("replicated result", ": all bytes received per GPU (+ replication of the merged lists)"), ("own slice only", ": hit bytes received per GPU (bounds + slices / blocks)")):
                    if form != "slices" and deliver == "own slice only":
                        continue
                    wire = [1e3 * r[tag + key_b] / bw for bw in (150e9, 300e9)]
                    step = [r["local_step_ms"] + dev_ms + w for w in wire]
                    pred.setdefault(str(G), {})["%s, %s" % (tag, deliver)] = {
                        "terms_ms": {"local_step": r["local_step_ms"], "exchange_device_work_per_gpu": dev_ms, "wire@150GB/s,300GB/s": wire},
                        "bytes_received_per_gpu": r[tag + key_b], "predicted_step_ms": step,
                        "predicted_speedup_vs_1gpu": [base / x for x in step], "predicted_qps": [n_q / (x * 1e-3) for x in step]}
    out["predicted"] = pred
    print(json.dumps(out))


if __name__ == "__main__":
    main()
