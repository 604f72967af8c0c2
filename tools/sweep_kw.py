"""Keyword knob sweep on one resident 10M-doc index: kw_chunk_blocks x batch size. Prints one line per setting."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import typesense_amd as T
from typesense_amd import _lib as B, synth
import bench

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
csr = synth.zipf_corpus_csr(n_docs, 100_000, 32, seed=2)
pts = synth.points_column(n_docs)
g = T.GpuIndex(0, os.environ.get('TSGPU_LIB') or None)
g.field_create(0, False)
g.terms_load_csr(0, csr["term_ids"], csr["ids_ptr"], csr["ids"], csr["offset_index"], csr["off_ptr"], csr["offsets"])
g.column_set(0, pts); g.set_num_docs(n_docs); g.commit()
sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
for n_q in [int(x) for x in os.environ.get("KW_BATCHES", "10000,1000,100").split(",")]:
    qtok = synth.keyword_queries(n_q, int(os.environ.get("KW_TOKENS", "3")), 8, 2000, seed=4)
    arr = (B.KwQueryC * n_q)()
    for i in range(n_q):
        T.KwQuery(qtok[i], sort=sort, topster_size=250).fill(arr[i])
    dev, hs = bench.device_hits(torch, n_q, 250)
    if os.environ.get("KW_HOST"):                       # host outputs (the sliced delivery): the arrays the B1 shim requests
        hh = T.Hits(n_q, 250)
        hs = hh.c_struct(seam_arrays_only=True)
    for opts in json.loads(os.environ.get("KW_SWEEP", '[{"kw_chunk_blocks":64}]')):
        for k, v in opts.items():
            g.set_option(k, v)
        for _ in range(2):
            g.keyword_search_batch_raw(arr, n_q, hs)
        prof = None
        if hasattr(g.L, 'tsgpu_debug_prof') and os.environ.get('KW_PROF'):
            import ctypes as C
            prof = (C.c_uint64 * 16)()
            g.L.tsgpu_debug_prof(g.h, 1, None)
        t0 = time.perf_counter(); ks = []; ms = []; fs = []
        for _ in range(5):
            g.keyword_search_batch_raw(arr, n_q, hs)
            tm = g.timings(); ks.append(tm.kw_search_ms); ms.append(tm.kw_merge_ms); fs.append(tm.kw_find_ms)
        wall = (time.perf_counter() - t0) / 5
        print(json.dumps(dict(n_q=n_q, opts=opts, wall_ms=wall * 1e3, qps=n_q / wall, search_ms=float(np.mean(ks)), find_ms=float(np.mean(fs)), merge_ms=float(np.mean(ms)),
                              alg_GBs=tm.kw_algorithmic_bytes / (np.mean(ks) * 1e-3) / 1e9)), flush=True)
        if prof is not None:
            g.L.tsgpu_debug_prof(g.h, 1, prof)
            v = list(prof)
            tot = sum(v[:12]) or 1
            print('PROF wg=%d' % v[12], ' '.join('p%d=%.1f%%' % (i, 100.0 * v[i] / tot) for i in range(12)), 'cycles/wg=%.0f' % (tot / max(v[12], 1)), flush=True)
g.close()
