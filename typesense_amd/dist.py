"""Multi-GPU plumbing for the doc-range-sharded hot path: one process per GPU
(torch.distributed; backend nccl = RCCL over xGMI on the GPU box, gloo in the CPU tests), every rank scores the whole
query batch on its shard, ONE all-gather of the per-shard top-K, then an exact merge in the Topster order
(include/topster.h:146-149). Nothing here computes scores: ranks come out of libtsgpu.so, this module only moves and
orders them. The same code runs on CUDA tensors (bench.py) and on CPU tensors (tests/test_dist_gloo.py)."""
import numpy as np


def world():
    import torch.distributed as dist
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def all_gather_cat(t):
    """t: [B, ...] on every rank -> [G, B, ...] (one collective per tensor)"""
    import torch
    import torch.distributed as dist
    _, G = world()
    out = torch.empty((G,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    if G == 1:
        out[0] = t
    else:
        dist.all_gather_into_tensor(out.view(G * t.shape[0], *t.shape[1:]) if t.dim() else out, t.contiguous())
    return out


def merge_keyword_topk(keys, scores, n_hits, k):
    """exact G-way merge of per-shard Topster lists. keys [G,B,K] int64, scores [G,B,K,3] int64, n_hits [G,B] ->
    (keys [B,k], scores [B,k,3], n [B]); order = (s0, s1, s2, key) descending, invalid slots last."""
    import torch
    G, Bq, K = keys.shape
    kk = keys.permute(1, 0, 2).reshape(Bq, G * K)
    sc = scores.permute(1, 0, 2, 3).reshape(Bq, G * K, 3)
    nh = n_hits.permute(1, 0).to(torch.int64)                                           # [B, G]
    valid = (torch.arange(K, device=keys.device)[None, None, :] < nh[:, :, None]).reshape(Bq, G * K)
    order = torch.arange(G * K, device=keys.device)[None, :].expand(Bq, -1)
    for col in (kk, sc[..., 2], sc[..., 1], sc[..., 0]):                                 # stable sorts, least significant key first
        v = torch.gather(col, 1, order)
        order = torch.gather(order, 1, torch.sort(v, dim=1, descending=True, stable=True).indices)
    v = torch.gather(valid.to(torch.int8), 1, order)
    order = torch.gather(order, 1, torch.sort(v, dim=1, descending=True, stable=True).indices)[:, :k]
    out_keys = torch.gather(kk, 1, order)
    out_sc = torch.gather(sc, 1, order[:, :, None].expand(-1, -1, 3))
    return out_keys, out_sc, torch.clamp(nh.sum(1), max=k)


def merge_knn_topk(dist_t, labels, cnt, k):
    """dist [G,B,K] f32, labels [G,B,K] int64, cnt [G,B] -> closest first, ties: smaller label first"""
    import torch
    G, Bq, K = dist_t.shape
    d = dist_t.permute(1, 0, 2).reshape(Bq, G * K).clone()
    l = labels.permute(1, 0, 2).reshape(Bq, G * K)
    valid = (torch.arange(K, device=d.device)[None, None, :] < cnt.permute(1, 0).to(torch.int64)[:, :, None]).reshape(Bq, G * K)
    d[~valid] = float("inf")
    o = torch.sort(l, dim=1, stable=True).indices
    d, l, valid = torch.gather(d, 1, o), torch.gather(l, 1, o), torch.gather(valid, 1, o)
    o = torch.sort(d, dim=1, stable=True).indices[:, :k]
    return torch.gather(d, 1, o), torch.gather(l, 1, o), torch.clamp(cnt.to(torch.int64).sum(0), max=k)


def merge_keyword_topk_device(index, g, k):
    """the same merge by libtsgpu's kw_shard_merge_kernel (one workgroup per query, LDS bitonic sort): the gathered CUDA
    tensors are handed over as raw pointers. index: the rank's GpuIndex."""
    import ctypes as C
    import torch
    from . import _lib as B
    G, Bq, K = g["keys"].shape
    dev = g["keys"].device
    out = dict(keys=torch.empty((Bq, k), dtype=torch.int64, device=dev), scores=torch.empty((Bq, k, 3), dtype=torch.int64, device=dev),
               n_hits=torch.empty(Bq, dtype=torch.int32, device=dev), num_matched=torch.empty(Bq, dtype=torch.int64, device=dev))
    hin, hout = B.HitsC(), B.HitsC()
    hin.mem = hout.mem = B.MEM_DEVICE
    hin.k_stride, hout.k_stride = K, k
    for name in ("keys", "scores", "n_hits", "num_matched"):
        setattr(hin, name, g[name].data_ptr())
        setattr(hout, name, out[name].data_ptr())
    torch.cuda.current_stream().synchronize()          # the library runs on its own stream
    B.check(index.L, index.L.tsgpu_merge_shard_hits_device(index.h, C.byref(hin), G, Bq, k, C.byref(hout)))
    return out["keys"], out["scores"], out["n_hits"].to(torch.int64), out["num_matched"]


def merge_gathered_keyword(index, g_pack, g_counts, k):
    """bench.py's shard step: ONE all-gather delivered g_pack [G,B,K,4] = {key, scores[3]} and g_counts [G,B,2] = {n_hits, num_matched};
    exact merge on the device (kw_shard_merge_kernel): global Topster order, num_matched = sum over the shards."""
    import torch
    g = dict(keys=g_pack[..., 0].contiguous(), scores=g_pack[..., 1:].contiguous(),
             n_hits=g_counts[..., 0].to(torch.int32).contiguous(), num_matched=g_counts[..., 1].contiguous())
    if index is not None and g["keys"].is_cuda:
        return merge_keyword_topk_device(index, g, k)
    keys, sc, n = merge_keyword_topk(g["keys"], g["scores"], g["n_hits"], k)
    return keys, sc, n, g["num_matched"].sum(0)


def sharded_keyword(local, k, index=None):
    """local: dict(keys [B,K] int64, scores [B,K,3] int64, n_hits [B] int32, num_matched [B] int64) of THIS shard ->
    merged (keys, scores, n, num_matched) identical on every rank. With `index` (a GpuIndex) and CUDA tensors the merge runs
    in libtsgpu (kw_shard_merge_kernel); otherwise the torch sort-based merge (CPU tests)."""
    g = {name: all_gather_cat(local[name]) for name in ("keys", "scores", "n_hits", "num_matched")}
    if index is not None and g["keys"].is_cuda:
        return merge_keyword_topk_device(index, g, k)
    keys, sc, n = merge_keyword_topk(g["keys"], g["scores"], g["n_hits"], k)
    return keys, sc, n, g["num_matched"].sum(0)


def sharded_knn(dist_t, labels, cnt, k):
    return merge_knn_topk(all_gather_cat(dist_t), all_gather_cat(labels), all_gather_cat(cnt), k)


def shard_range(n, rank, world_size):
    return n * rank // world_size, n * (rank + 1) // world_size


def torch_collectives(group=None):
    """(all_gather, all_to_all) for GpuGroup.join_host / tsgpu_group_create_rank_host: the product's rank-form exchange over
    torch.distributed on HOST memory (gloo). all_to_all uses the backend's all_to_all_single where it exists and otherwise
    an all-gather of the whole send buffers + a local slice pick (same result, more bytes on the wire)."""
    import torch
    import torch.distributed as dist
    rank, G = dist.get_rank(group), dist.get_world_size(group)
    state = {"a2a": True}

    def all_gather(send, recv, nbytes):
        if nbytes == 0:
            return
        dist.all_gather_into_tensor(torch.from_numpy(recv), torch.from_numpy(send).contiguous(), group=group)

    def all_to_all(send, recv, nbytes):
        if nbytes == 0:
            return
        s, r = torch.from_numpy(send), torch.from_numpy(recv)
        if state["a2a"]:
            try:
                dist.all_to_all_single(r, s.contiguous(), group=group)
                return
            except (RuntimeError, NotImplementedError):
                state["a2a"] = False           # every rank runs the same backend: they all land here together
        whole = torch.empty(G * s.numel(), dtype=torch.uint8)
        dist.all_gather_into_tensor(whole, s.contiguous(), group=group)
        w = whole.view(G, G, nbytes)
        r.view(G, nbytes).copy_(w[:, rank, :])

    return all_gather, all_to_all
