"""world_size-2 worker of tests/test_dist_gloo.py: doc-range shards on the SIMT-emulator build of the product sources,
gloo all-gather of the per-shard top-K, exact merge, compared on rank 0 with the unsharded oracle."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import typesense_amd as T                         # noqa: E402
from typesense_amd import _lib as B, dist as D    # noqa: E402
from oracle import oracle_py as O                 # noqa: E402
from tests import helpers as H                    # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = os.environ["TSGPU_EMU_LIB"]
    n_docs, dim, K, k_vec = 1500, 24, 40, 12
    docs = H.zipf_docs(n_docs, 80, 10, seed=8)                      # every rank derives the same collection
    pts = H.points_of(n_docs)
    rng = np.random.default_rng(3)
    X = rng.standard_normal((n_docs, dim)).astype(np.float32)
    Q = rng.standard_normal((4, dim)).astype(np.float32)
    lo, hi = D.shard_range(n_docs, rank, world)
    orc = O.OracleIndex(1, 1)                                        # full-collection oracle (checker, rank 0 compares)
    for d in range(n_docs):
        orc.index_plain(d, 0, docs[d])
    orc.set_sort_dense(0, pts)
    orc.vec_init(dim, O.METRIC_IP)
    orc.vec_add(np.arange(n_docs, dtype=np.uint32), X)
    # ---- this rank's shard: postings restricted to [lo, hi), global seq_ids kept ----
    g = T.GpuIndex(0, lib)
    g.field_create(0, False)
    for term in orc.terms(0):
        ids, oi, off = orc.dump_posting(0, int(term))
        sel = np.nonzero((ids >= lo) & (ids < hi))[0]
        if sel.size == 0:
            continue
        ends = np.append(oi[1:], off.size)
        new_off, new_oi = [], []
        for j in sel:
            new_oi.append(len(new_off))
            new_off.extend(off[oi[j]:ends[j]])
        g.term_upsert(0, int(term), ids[sel], new_oi, new_off)
    g.column_set(0, pts)
    g.set_num_docs(n_docs)
    g.commit()
    g.vec_create(1, dim, B.METRIC_IP)
    g.vec_upsert(1, np.arange(lo, hi, dtype=np.uint64), X[lo:hi])
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    toks = ([1, 2], [3, 1, 2], [5], [4, 9])
    qs = [T.KwQuery(t, sort=sort, topster_size=K) for t in toks]
    # ---- keyword: local top-K -> all-gather -> merge ----
    h = g.keyword_search_batch(qs, k_stride=K)
    local = dict(keys=torch.from_numpy(h.keys.astype(np.int64)), scores=torch.from_numpy(h.scores.copy()),
                 n_hits=torch.from_numpy(h.n_hits.astype(np.int32)), num_matched=torch.from_numpy(h.num_matched.astype(np.int64)))
    keys, sc, n, nm = D.sharded_keyword(local, K)
    # ---- vector: local top-k -> all-gather -> merge ----
    dl, ll, cl = g.vec_knn_batch(1, Q, k_vec)
    dm, lm, cm = D.sharded_knn(torch.from_numpy(dl), torch.from_numpy(ll.astype(np.int64)), torch.from_numpy(cl.astype(np.int32)), k_vec)
    # ---- hybrid: fuse AFTER the merge (ranks are global), on the merged lists ----
    merged = T.Hits(len(qs), K)
    merged.keys[:] = keys.numpy().astype(np.uint64)
    merged.scores[:] = sc.numpy()
    merged.n_hits[:] = n.numpy().astype(np.uint32)
    merged.num_matched[:] = nm.numpy().astype(np.uint64)
    merged.match_score_index[:] = 0
    merged.text_match[:] = sc.numpy()[:, :, 0]
    fused = g.hybrid_fuse_batch(qs, merged, dm.numpy(), lm.numpy().astype(np.uint64), cm.numpy().astype(np.uint32), B.METRIC_IP,
                                k=k_vec, fetch_size=10, alpha=0.3, k_stride=K)
    ok = True
    if rank == 0:
        for i, q in enumerate(qs):
            ref = H.oracle_keyword(orc, q)
            m = int(n[i])
            ok &= m == ref.keys.size and np.array_equal(keys[i, :m].numpy().astype(np.uint64), ref.keys)
            ok &= np.array_equal(sc[i, :m].numpy(), ref.scores) and int(nm[i]) == int(ref.num_keyword_matches)
        for i in range(Q.shape[0]):
            d, l = orc.flat_knn(Q[i], k_vec)
            ok &= np.array_equal(lm[i].numpy().astype(np.uint32), l) and np.allclose(dm[i].numpy(), d, rtol=1e-5, atol=1e-5)
        for i, q in enumerate(qs):
            oq = orc.make_query(q.tokens, sort=((O.SORT_TEXT_MATCH, 0, 1), (O.SORT_INT64_COLUMN, 0, 1)), fetch_size=10, topster_size=K)
            ref = orc.search_hybrid(oq, Q[i], k=k_vec, alpha=0.3)
            m = int(fused.n_hits[i])
            ok &= m == ref.keys.size and np.array_equal(fused.keys[i, :m], ref.keys) and np.array_equal(fused.scores[i, :m], ref.scores)
        print("DIST_OK" if ok else "DIST_MISMATCH", flush=True)
    g.close()
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
