"""Multi-GPU layer: one process per GPU, knot points (or whole trajectories) sharded contiguously across ranks,
no data-path collective for the evaluation itself; ONE exchange at the end for callers that need the stacked
Jacobians in one place (torch.distributed over RCCL/xGMI on GPUs, gloo in the CPU tests).

The reference has no distributed code; this is the `for k` loop of its callers split by rank.
"""
import os

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch, dist = None, None


def shard_range(N, rank, world):
    """Contiguous block of ceil(N/world) knot points for `rank` (last ranks may be short or empty)."""
    per = -(-N // world)
    lo = min(rank * per, N)
    return lo, min(lo + per, N)


def shard_sizes(N, world):
    return [shard_range(N, r, world)[1] - shard_range(N, r, world)[0] for r in range(world)]


def env_rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def gather_stacked(local, N, mode="root", root=0, group=None):
    """Stack per-rank result slabs (rows = knot points) along dim 0.
    mode "none": results stay sharded (returns `local`);
    mode "all" : every rank receives the (N, ...) stack (all_gather on equal shards, padded when N % world != 0);
    mode "root": only `root` receives it -- implemented as direct point-to-point sends, which on the fully
                 connected 8-GPU xGMI topology puts each shard on its own link instead of a ring."""
    if mode == "none" or dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = shard_sizes(N, world)
    tail = tuple(local.shape[1:])
    if mode == "all":
        per = max(sizes)
        if all(s == per for s in sizes):
            out = torch.empty((N,) + tail, dtype=local.dtype, device=local.device)
            dist.all_gather_into_tensor(out, local.contiguous(), group=group)
            return out
        pad = torch.zeros((per,) + tail, dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
        buf = torch.empty((world * per,) + tail, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(buf, pad, group=group)
        return torch.cat([buf[r * per: r * per + sizes[r]] for r in range(world)], dim=0)
    if mode == "root":
        if rank == root:
            out = torch.empty((N,) + tail, dtype=local.dtype, device=local.device)
            ops, lo = [], 0
            for r in range(world):
                hi = lo + sizes[r]
                if r == root:
                    out[lo:hi].copy_(local)
                elif sizes[r] > 0:
                    ops.append(dist.P2POp(dist.irecv, out[lo:hi], r, group=group))
                lo = hi
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            return out
        if local.shape[0] > 0:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, local.contiguous(), root, group=group)]):
                w.wait()
        return None
    raise ValueError(f"unknown gather mode {mode!r}")
