"""Bench / test tooling (NOT the product path): an HNSW-SHAPED graph over the rows of a resident vector field, made on the GPU.

The reference builds its graphs with hnswlib's sequential addPoint on the CPU (src/index.cpp:1003-1054); at 1M x 768 that takes
longer than a bench run may, and no graph file can travel with the repo. What the search kernel needs is the structure hnswlib
leaves behind — geometric level assignment, level-0 lists of <= 2M neighbours, upper lists of <= M, all chosen by the
"keep a candidate only if it is closer to the node than to every neighbour already kept" heuristic
(getNeighborsByHeuristic2) with reverse links — so this module derives exactly that from EXACT k-nearest-neighbour lists
(tsgpu_vec_knn_batch: the brute-force path of this library) instead of from beam searches during insertion:
  * level(i) = floor(-ln(U) / ln(M)), hnswlib's getRandomLevel;
  * per level: candidates = the node's K exact nearest neighbours among the nodes of that level; the heuristic keeps <= M of
    them; reverse links fill the lists up to 2M (level 0) / M (upper levels), closest first;
  * entry point = the first node of the top level.
It is NOT hnswlib's graph (insertion order effects are absent) and the numbers measured on it say so ("graph": "knn-heuristic").
The oracle adopts the same arrays (oracle_py.hnsw_import), so GPU traversal and CPU traversal walk the same graph."""
import math

import numpy as np


def _heuristic(torch, cand, dist, X, M):
    """cand [B, K] int64 row ids ascending by dist [B, K] (to the node); -> keep mask [B, K] (<= M per row)"""
    Bn, K = cand.shape
    V = X[cand.reshape(-1)].reshape(Bn, K, -1)
    if V.dtype != torch.bfloat16:
        V = V.to(torch.bfloat16)
    D = 1.0 - torch.bmm(V, V.transpose(1, 2)).float()                 # candidate-to-candidate distances
    keep = torch.zeros((Bn, K), dtype=torch.bool, device=cand.device)
    cnt = torch.zeros(Bn, dtype=torch.int32, device=cand.device)
    valid = cand >= 0
    for i in range(K):
        closer_to_kept = ((D[:, i, :] < dist[:, i:i + 1]) & keep).any(dim=1)     # some kept neighbour is closer to it than the node is
        ok = valid[:, i] & ~closer_to_kept & (cnt < M)
        keep[:, i] = ok
        cnt += ok.to(torch.int32)
    return keep


def _fill_lists(torch, n, src, dst, dist, cap, width):
    """edges (src -> dst, dist) with priority = order given (earlier first); per src keep the first `cap` distinct dst -> [n, width] table (count, ids..)"""
    key = src * (n + 1) + dst
    order = torch.arange(src.numel(), device=src.device)
    # distinct (src, dst): keep the earliest occurrence
    sk, si = torch.sort(key, stable=True)
    first = torch.ones_like(sk, dtype=torch.bool)
    first[1:] = sk[1:] != sk[:-1]
    idx = si[first]
    idx = idx[torch.argsort(order[idx], stable=True)]
    s2, d2 = src[idx], dst[idx]
    so, sp = torch.sort(s2, stable=True)                                # by src, priority order kept
    d2 = d2[sp]
    start = torch.ones_like(so, dtype=torch.bool)
    start[1:] = so[1:] != so[:-1]
    pos_start = torch.nonzero(start).squeeze(1)
    rank = torch.arange(so.numel(), device=so.device) - torch.repeat_interleave(pos_start, torch.diff(torch.cat([pos_start, torch.tensor([so.numel()], device=so.device)])))
    ok = rank < cap
    table = torch.zeros((n, width), dtype=torch.int64, device=src.device)
    table[so[ok], 1 + rank[ok]] = d2[ok]
    table[:, 0] = torch.bincount(so[ok], minlength=n)
    return table


def build_graph(torch, g, field_id, X, M=16, K0=64, seed=100, batch=1024, log=None, knn=None):
    """X: torch cuda [n, dim] fp32 rows of the field in insertion order (labels = row numbers; unit rows for a cosine field).
    knn(a, b, k) -> (labels [b-a, k] int64, dist [b-a, k]) overrides the library call (tests on the SIMT emulator, where a scan is slow).
    -> dict(M, maxlevel, enterpoint, link0[n, 1+2M] u32, upper_ptr[n+1] u64, upper_links[n_upper, 1+M] u32, levels[n])"""
    from . import _lib as B
    n, dim = X.shape
    dev = X.device
    rng = np.random.default_rng(seed)
    levels = np.floor(-np.log(1.0 - rng.random(n)) / math.log(M)).astype(np.int64)
    maxlevel = int(levels.max()) if n else -1
    lev = torch.from_numpy(levels).to(dev)
    Xh = X.to(torch.bfloat16)                                           # the heuristic's candidate-to-candidate distances (structure only)
    # ---- level 0: exact K0 nearest neighbours of every row through the library's own brute-force path ----
    k = K0 + 1
    src_f, dst_f, dist_f = [], [], []
    d = torch.zeros((batch, k), dtype=torch.float32, device=dev); l = torch.zeros((batch, k), dtype=torch.int64, device=dev); c = torch.zeros(batch, dtype=torch.int32, device=dev)
    for a in range(0, n, batch):
        b = min(n, a + batch)
        if knn is None:
            g.vec_knn_batch_raw(field_id, X[a:b].data_ptr(), B.MEM_DEVICE, b - a, k, d.data_ptr(), l.data_ptr(), c.data_ptr(), B.MEM_DEVICE)
            if dev.type == "cuda":
                torch.cuda.synchronize()
            cand, dist = l[:b - a].clone(), d[:b - a].clone()
        else:
            cand, dist = knn(a, b, k)
        rows = torch.arange(a, b, device=dev)
        notself = cand != rows[:, None]
        # drop the node itself (wherever ties put it), keep order
        ordk = torch.argsort((~notself).to(torch.int8), dim=1, stable=True)[:, :K0]
        cand, dist = torch.gather(cand, 1, ordk), torch.gather(dist, 1, ordk)
        keep = _heuristic(torch, cand, dist, Xh, M)
        rr = rows[:, None].expand(-1, K0)
        src_f.append(rr[keep]); dst_f.append(cand[keep]); dist_f.append(dist[keep])
        if log and (a // batch) % 200 == 0:
            log("hnsw_synth: level 0 rows %d / %d" % (b, n))
    src_f, dst_f, dist_f = torch.cat(src_f), torch.cat(dst_f), torch.cat(dist_f)
    # forward (heuristic) links first, then reverse links closest first
    ro = torch.argsort(dist_f, stable=True)
    link0 = _fill_lists(torch, n, torch.cat([src_f, dst_f[ro]]), torch.cat([dst_f, src_f[ro]]), None, 2 * M, 1 + 2 * M)
    # ---- upper levels: the same among the nodes of each level (dense torch: the sets are small) ----
    upper_lists = {}
    for level in range(1, maxlevel + 1):
        ids = torch.nonzero(lev >= level).squeeze(1)
        m = ids.numel()
        if m == 1:
            upper_lists[level] = (ids, torch.zeros((1, 1 + M), dtype=torch.int64, device=dev))
            continue
        Xs = X[ids]
        K = min(2 * M, m - 1)
        cand_l, dist_l = [], []
        for a in range(0, m, 4096):
            S = 1.0 - Xs[a:a + 4096] @ Xs.T
            S[torch.arange(S.shape[0], device=dev), torch.arange(a, a + S.shape[0], device=dev)] = float("inf")
            dv, iv = torch.topk(S, K, dim=1, largest=False, sorted=True)
            cand_l.append(iv); dist_l.append(dv)
        cand, dist = torch.cat(cand_l), torch.cat(dist_l)
        keep = _heuristic(torch, cand, dist, Xs, M)
        rr = torch.arange(m, device=dev)[:, None].expand(-1, K)
        sf, df, ds = rr[keep], cand[keep], dist[keep]
        ro = torch.argsort(ds, stable=True)
        tab = _fill_lists(torch, m, torch.cat([sf, df[ro]]), torch.cat([df, sf[ro]]), None, M, 1 + M)
        tab[:, 1:] = torch.where(torch.arange(M, device=dev)[None, :] < tab[:, :1], ids[tab[:, 1:]], torch.zeros_like(tab[:, 1:]))
        upper_lists[level] = (ids, tab)
    # flat form: node i's upper lists (level 1.. ascending) at upper_ptr[i]
    upper_ptr = np.zeros(n + 1, np.uint64)
    upper_ptr[1:] = np.cumsum(levels)
    n_upper = int(upper_ptr[n])
    upper = torch.zeros((max(n_upper, 1), 1 + M), dtype=torch.int64, device=dev)
    up_t = torch.from_numpy(upper_ptr[:-1].astype(np.int64)).to(dev)
    for level, (ids, tab) in upper_lists.items():
        upper[up_t[ids] + (level - 1)] = tab
    enter = int(torch.nonzero(lev == maxlevel).squeeze(1)[0].item()) if n else 0
    return dict(M=M, maxlevel=maxlevel, enterpoint=enter, levels=levels.astype(np.uint32), link0=link0.to(torch.int32).cpu().numpy().astype(np.uint32),
                upper_ptr=upper_ptr, upper_links=upper[:n_upper].to(torch.int32).cpu().numpy().astype(np.uint32), graph="knn-heuristic")
