"""Target sharding over GPUs (one process per GPU, torch.distributed).

The reference shards particles by Peano-Hilbert key ranges (TopLeaves -> Task, domain.c:154-256) and every rank
walks its own particles (treewalk.c:801-902).  Here the tree order of the device tree is a Morton order, so a
contiguous range of tree slots is a spatially compact domain; rank r owns slots [lo_r, hi_r) and walks them.
After the walk the per-rank result blocks are exchanged with ONE all-gather (RCCL on GPUs, gloo in CPU tests)
and scattered back to caller order.  These helpers hold only index logic so they run on CPU tensors too.
"""
import torch
import torch.distributed as dist


def slot_range(n, rank, world):
    """Tree-slot range [lo, hi) owned by `rank` (balanced to +-1)."""
    return (n * rank) // world, (n * (rank + 1)) // world


def chunk_size(n, world):
    return (n + world - 1) // world


def exchange_results(values, order, rank, world, sbuf=None, gbuf=None, group=None):
    """All-gather per-rank results.

    values : [N, k] tensor in caller order; rows order[lo:hi] hold this rank's fresh results.
    order  : [N] int tensor, tree slot -> caller index (identical on every rank).
    On return every row of `values` holds the result computed by its owner rank."""
    n = values.shape[0]
    if world == 1:
        return values
    chunk = chunk_size(n, world)
    lo, hi = slot_range(n, rank, world)
    if sbuf is None:
        sbuf = torch.zeros((chunk,) + tuple(values.shape[1:]), dtype=values.dtype, device=values.device)
    if gbuf is None:
        gbuf = torch.zeros((world * chunk,) + tuple(values.shape[1:]), dtype=values.dtype, device=values.device)
    sbuf[:hi - lo] = values[order[lo:hi].long()]
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(gbuf, sbuf, group=group)
    else:
        # backends without the flat form (gloo on device tensors): gather a list and copy
        parts = [torch.empty_like(sbuf) for _ in range(world)]
        dist.all_gather(parts, sbuf, group=group)
        for r in range(world):
            gbuf[r * chunk:(r + 1) * chunk] = parts[r]
    for r in range(world):
        rlo, rhi = slot_range(n, r, world)
        values[order[rlo:rhi].long()] = gbuf[r * chunk:r * chunk + (rhi - rlo)]
    return values
