"""One-ciphertext-per-GPU sharding of a batch of independent he_mul's.

The reference has no communication layer at all: ciphertext
multiplications are independent, so the only multi-GPU structure is a partition
of the batch.  Nothing is exchanged inside an NTT; the collectives here are the
optional scatter of input slabs from rank 0, the gather of output slabs back,
and the MAX-reduce of the timing.

Two planes.  The CONTROL plane (barriers, the MAX of the timing, small python
objects) is the default process group and may be plain gloo: it carries a few
bytes.  The DATA plane (scatter / gather of slabs) is the group handed to
`use_data_group` -- backend "nccl" (= RCCL over xGMI) on GPUs, created lazily, so
that a broken fabric can only fail the transfer leg, never the set-up of the
ranks or the compute-only measurement.  Without `use_data_group` the slabs
travel over the default group (gloo in the CPU tests: staged through the host).
"""
import torch
import torch.distributed as dist

_data = None          # process group of the slab transfers; None = the default group


def use_data_group(group):
    """Slabs travel over `group` from now on (None = the default group again)."""
    global _data
    _data = group


def data_backend():
    return dist.get_backend(_data)


def shard_range(batch, world, rank):
    """Contiguous block partition of `batch` ciphertexts: [lo, hi) of this rank.
    The first batch % world ranks take one extra ciphertext."""
    base, extra = divmod(batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _wire_device(device):
    """Where a slab sits while it is on the wire: RCCL moves device memory; gloo (CPU tests, and the
    rehearsal of several ranks on one GPU) moves host memory, so device slabs are staged through the host."""
    return device if data_backend() == "nccl" else torch.device("cpu")


def scatter_slab(full, per_ct, batch, src=0, device=None):
    """Rank `src` holds `full` = int64[batch*per_ct] (others pass None); every rank
    gets its own shard int64[(hi-lo)*per_ct] on `device` (default: where `full` lives).
    Ragged shards are allowed."""
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_range(batch, world, rank)
    if device is None:
        device = full.device if full is not None else _default_device()
    device = torch.device(device)
    if world == 1:
        return full.to(device, copy=True)
    wire = _wire_device(device)
    if wire != device:
        staged = scatter_slab(full.to(wire) if full is not None else None, per_ct, batch, src, wire)
        return staged.to(device)
    mine = torch.empty((hi - lo) * per_ct, dtype=torch.int64, device=device)
    ops = []
    if rank == src:
        for r in range(world):
            rlo, rhi = shard_range(batch, world, r)
            piece = full[rlo * per_ct: rhi * per_ct]
            if r == src:
                mine.copy_(piece)
            elif rhi > rlo:
                ops.append(dist.P2POp(dist.isend, piece.contiguous(), r, group=_data))
    elif hi > lo:
        ops.append(dist.P2POp(dist.irecv, mine, src, group=_data))
    if ops:
        for w in dist.batch_isend_irecv(ops):  # grouped send/recv: one RCCL group on GPUs
            w.wait()
    return mine


def gather_slab(mine, per_ct, batch, dst=0):
    """Inverse of scatter_slab: rank `dst` returns int64[batch*per_ct], others None."""
    world, rank = dist.get_world_size(), dist.get_rank()
    if world == 1:
        return mine.clone()
    wire = _wire_device(mine.device)
    if wire != mine.device:
        full = gather_slab(mine.to(wire), per_ct, batch, dst)       # .to() waits for the kernels that produced `mine`
        return full.to(mine.device) if full is not None else None
    ops = []
    full = None
    if rank == dst:
        full = torch.empty(batch * per_ct, dtype=torch.int64, device=mine.device)
        for r in range(world):
            rlo, rhi = shard_range(batch, world, r)
            if r == dst:
                full[rlo * per_ct: rhi * per_ct].copy_(mine)
            elif rhi > rlo:
                ops.append(dist.P2POp(dist.irecv, full[rlo * per_ct: rhi * per_ct], r, group=_data))
    elif mine.numel():
        ops.append(dist.P2POp(dist.isend, mine.contiguous(), dst, group=_data))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return full


def max_over_ranks(seconds, device=None):
    """Wall time of the slowest rank (the number bench.py reports); control plane."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rank_records(mine):
    """Every rank's record (a small dict: rank, device, its OWN rate over the timed steps, where it was placed) on every rank, in rank order;
    control plane.  One rank: [mine]."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [mine]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, mine)
    return out


def summarize_ranks(records):
    """What bench.py prints beside the MAX-time aggregate: the slowest and the fastest rank's own he_mul/s, their sum (what the job would do if no
    rank waited for another), and the records themselves -- a slow device or a badly placed rank is visible, not just a lower `value`."""
    rates = [r["he_mul_per_s"] for r in records]
    return {"he_mul_per_s_min": min(rates), "he_mul_per_s_max": max(rates), "sum_of_own_rates": round(sum(rates), 1), "ranks": records}


def _default_device():
    """Where a shard lands when the caller names no device: the GPU when the slabs travel over RCCL."""
    return torch.device("cuda", torch.cuda.current_device()) if data_backend() == "nccl" else torch.device("cpu")
