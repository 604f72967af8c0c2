"""Plumbing for tsgpu_group's rank form: the doc range of a shard, and the two host-memory collectives the group's HOST
transport takes as callbacks (tsgpu_group_create_rank_host, include/tsgpu.h) — here over torch.distributed (gloo in the CPU tests and the
one-GPU rehearsal). Nothing here computes, orders or merges results: that is the library's (tsgpu_group.hip)."""


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
