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


# ------------------------------------------------------------------------------------------------------------------------------------
# The native exchange (C ABI: rdhip_comm_*, rdhip_gather): RCCL driven by librdhip itself, no torch.distributed in the data path.  This is
# the path a Julia host takes; torch tensors are used here only as device buffers.
# ------------------------------------------------------------------------------------------------------------------------------------
class Comm:
    """One rank per GPU.  Comm.local(devices): ONE process drives all ranks (rdhip_comm_create, ncclCommInitAll).
    Comm.from_torch_distributed(): one process per GPU; rank 0 draws the RCCL unique id and torch.distributed only carries those 128 bytes
    (rdhip_comm_unique_id / rdhip_comm_create_rank)."""

    def __init__(self, handle, ctxs, ranks, nranks):
        self.handle, self.ctxs, self.ranks, self.nranks = handle, ctxs, ranks, nranks

    @classmethod
    def local(cls, devices):
        import ctypes as C
        from . import _lib as L
        ctxs = [L.Context(int(d)) for d in devices]
        arr = (C.c_void_p * len(ctxs))(*[c.handle for c in ctxs])
        h = C.c_void_p()
        L.check(L.lib().rdhip_comm_create(arr, len(ctxs), C.byref(h)))
        return cls(h, ctxs, list(range(len(ctxs))), len(ctxs))

    @classmethod
    def from_torch_distributed(cls, device, stream=None, group=None):
        import ctypes as C
        from . import _lib as L
        world = dist.get_world_size(group) if dist is not None and dist.is_initialized() else 1
        rank = dist.get_rank(group) if world > 1 else 0
        ident = C.create_string_buffer(L.COMM_ID_BYTES)
        if world > 1:
            box = [None]
            if rank == 0:
                L.check(L.lib().rdhip_comm_unique_id(ident))
                box[0] = ident.raw
            dist.broadcast_object_list(box, src=0, group=group)
            ident = C.create_string_buffer(box[0], L.COMM_ID_BYTES)
        elif os.environ.get("RDHIP_COMM_FORCE_RCCL") == "1":       # test switch: a 1-rank communicator that really goes through librccl
            L.check(L.lib().rdhip_comm_unique_id(ident))
        ctx = L.Context(int(device), stream)
        h = C.c_void_p()
        L.check(L.lib().rdhip_comm_create_rank(ctx.handle, world, rank, ident, C.byref(h)), ctx.handle)
        return cls(h, [ctx], [rank], world)

    def gather(self, mode, shards, nbytes, outs, root=0):
        """rdhip_gather: `shards` / `outs` = one device tensor (or None) per LOCAL rank, `nbytes` = shard size in bytes of EVERY rank.
        Asynchronous on each rank's context stream; call synchronize() before reading `outs`."""
        import ctypes as C
        from . import _lib as L
        code = {"none": L.GATHER_NONE, "root": L.GATHER_ROOT, "all": L.GATHER_ALL}[mode]
        nl = len(self.ctxs)
        sh = (C.c_void_p * nl)(*[(s.data_ptr() if s is not None else None) for s in shards])
        ou = (C.c_void_p * nl)(*[(o.data_ptr() if o is not None else None) for o in outs])
        nb = (C.c_ulonglong * self.nranks)(*[int(b) for b in nbytes])
        L.check(L.lib().rdhip_gather(self.handle, code, int(root), sh, nb, ou), self.ctxs[0].handle)

    def synchronize(self):
        for c in self.ctxs:
            c.synchronize()

    def close(self):
        if getattr(self, "handle", None):
            from . import _lib as L
            L.lib().rdhip_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
