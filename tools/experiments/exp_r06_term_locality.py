"""Round 6: what could a term-major work order buy the pair-find kernel AT MOST? Same resident 10M-doc index, 10 000 three-term queries whose two
rarer terms are drawn as in the bench; the THIRD (most frequent) term — the list whose id directory stage 2 probes once per stage-1 survivor — is
  local:   the same list for every query of the batch (its 2.8 MB directory stays in every XCD's 4 MB L2: the best a work order could arrange),
  few:     one of 5 lists, spread: one of ~160 lists with a directory (what the bench batch looks like: 470 MB of directories).
The number of stage-2 probes is the number of stage-1 survivors — independent of the third list — so find-kernel time differences are the probes' cost
by where their directory entries come from. Also the SECOND list: `b_local` fixes the second term instead (the tile DMA's source)."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import typesense_amd as T
from typesense_amd import _lib as B, synth
import bench

n_docs, n_q = 10_000_000, 10_000
csr = synth.zipf_corpus_csr(n_docs, 100_000, 32, seed=2)
pts = synth.points_column(n_docs)
g = T.GpuIndex(0, os.environ.get('TSGPU_LIB') or None)
g.field_create(0, False)
g.terms_load_csr(0, csr["term_ids"], csr["ids_ptr"], csr["ids"], csr["offset_index"], csr["off_ptr"], csr["offsets"])
g.column_set(0, pts); g.set_num_docs(n_docs); g.commit()
sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
rng = np.random.default_rng(11)


def logu(lo, hi, n):
    return np.clip(np.exp(rng.uniform(np.log(lo), np.log(hi + 1), size=n)).astype(np.int64), lo, hi)


def batch(third, second=None):
    a = logu(200, 2000, n_q)
    b = logu(40, 199, n_q) if second is None else second
    q = np.stack([a, b, third], 1).astype(np.uint32)
    return q


cases = {
    "third_local (rank 20 for all)": batch(np.full(n_q, 20)),
    "third_few (ranks 18..22)": batch(rng.integers(18, 23, n_q)),
    "third_spread (ranks 8..39)": batch(rng.integers(8, 40, n_q)),
    "second_local (rank 100), third spread": batch(rng.integers(8, 40, n_q), second=np.full(n_q, 100)),
    "second_and_third_local (100, 20)": batch(np.full(n_q, 20), second=np.full(n_q, 100)),
    "bench-like (3 ranks log-uniform 8..2000)": synth.keyword_queries(n_q, 3, 8, 2000, seed=4),
}
for name, qtok in cases.items():
    arr = (B.KwQueryC * n_q)()
    for i in range(n_q):
        T.KwQuery(qtok[i], sort=sort, topster_size=250).fill(arr[i])
    dev, hs = bench.device_hits(torch, n_q, 250)
    for _ in range(2):
        g.keyword_search_batch_raw(arr, n_q, hs)
    fs, ks = [], []
    for _ in range(5):
        g.keyword_search_batch_raw(arr, n_q, hs)
        tm = g.timings(); fs.append(tm.kw_find_ms); ks.append(tm.kw_search_ms)
    g.set_option("kw_count_touched", 1); g.set_option("kw_device_plan_min_queries", 1 << 30)
    g.keyword_search_batch_raw(arr, n_q, hs)
    t = g.kw_touched()
    g.set_option("kw_count_touched", 0); g.set_option("kw_device_plan_min_queries", 512)
    print(json.dumps({"case": name, "find_ms": round(float(np.mean(fs)), 3), "find+score_ms": round(float(np.mean(ks)), 3), "matched": int(dev["num_matched"].sum().item()),
                      "driver_ids_GB": round(t["find_driver_ids"] / 1e9, 2), "tile_dma_GB": round(t["find_tile_dma"] / 1e9, 2), "probe_GB": round(t["find_probes"] / 1e9, 2),
                      "hit_records": t["find_hit_records"], "ns_per_driver_id": round(1e6 * float(np.mean(fs)) / max(1, t["find_driver_ids"] / 2), 3)}), flush=True)
g.close()
