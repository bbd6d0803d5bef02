"""Row exchanges between the processes of a run (one per GPU) over torch.distributed: the personalised exchange of particle rows
(exchange.c's MPI_Alltoallv) used by domain_peano.PeanoDomain, and the all-gather of per-target results the test tools assemble
whole-set outputs with.  RCCL ("nccl") moves device tensors; gloo - the CPU-launched tests that put several ranks on one GPU - has no
all_to_all on device tensors, there the exchanges gather and slice.  (The x-slab domains, the slab PM driver and the replicated-tree
sharding of round 1 that lived next to these helpers - domain.py, pm_slab.py, shard.py - were retired in round 3: the force step
runs on the library's own choreography, csrc/dist.hip; the last commit holding them is b25612b.)"""
import os

import torch
import torch.distributed as dist

# MPG_FORCE_COLLECTIVES=1: issue the collectives even in a one-rank group (lets a single-GPU box exercise the RCCL code path)
FORCE_COLLECTIVES = bool(os.environ.get("MPG_FORCE_COLLECTIVES"))


def _scratch(*shape, **kw):
    """an uninitialised work buffer; MPG_POISON=1 (the engine's debugging aid, csrc/mpg_common.h) fills it with NaN so that a read of
    an element nobody wrote shows up in the results"""
    if os.environ.get("MPG_POISON"):
        return torch.full(shape, float("nan"), **kw)
    return torch.empty(*shape, **kw)


def _has_all_to_all(group=None):
    """RCCL ("nccl") has all_to_all_single / all_gather_into_tensor on device tensors; gloo (the CPU-launched tests that put several
    ranks on one GPU) does not: there the exchanges gather everything and slice.  The path is chosen ONCE from the backend - not by
    catching errors, which would let one rank's genuine failure (out of memory, bad sizes) drop it into a different collective
    than its peers are in."""
    if os.environ.get("MPG_GLOO_TRY_A2A"):       # experiment: use the backend's own all_to_all_single whatever it is
        return True
    return dist.get_backend(group) == "nccl"


def count_matrix(counts, world, device, group=None):
    """allc[s][d] = rows rank s sends to rank d (host tensor)"""
    cnt = torch.tensor(counts, dtype=torch.int64, device=device)
    allc = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(allc, cnt, group=group)
    return torch.stack(allc).cpu()


def exchange_rows(send, counts, world, group=None, allc=None):
    """Personalised exchange: `send` holds the rows for rank 0, 1, ... back to back (counts[d] rows each); returns the rows
    received from all ranks, in rank order.  RCCL all_to_all_single with uneven splits; backends without it (gloo) gather.
    `allc`: the count matrix of count_matrix when the caller already has it (several fields, same lists)."""
    dev = send.device
    if allc is None:
        allc = count_matrix(counts, world, dev, group)
    rank = dist.get_rank(group)
    recv_counts = [int(allc[s][rank]) for s in range(world)]
    out = torch.empty((sum(recv_counts),) + tuple(send.shape[1:]), dtype=send.dtype, device=dev)
    if _has_all_to_all(group):
        dist.all_to_all_single(out, send, recv_counts, list(counts), group=group)
        return out
    nmax = int(allc.sum(1).max())
    pad = torch.zeros((nmax,) + tuple(send.shape[1:]), dtype=send.dtype, device=dev)
    pad[:send.shape[0]] = send
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    o = 0
    for s in range(world):
        off = int(allc[s][:rank].sum())
        c = recv_counts[s]
        out[o:o + c] = parts[s][off:off + c]
        o += c
    return out



def slab_of_cells(pos_x, cellsize, nmesh, world):
    """Owner rank of each particle: the slab holding its base cell floor(x / cellsize) (wrapped), as the engine computes it."""
    ix = torch.floor(pos_x / cellsize).to(torch.int64)
    ix = torch.where(ix >= nmesh, ix - nmesh, ix)
    ix = torch.where(ix < 0, ix + nmesh, ix)
    return ix // (nmesh // world)


class TargetExchange:
    """All-gather of per-target results (one per step): every rank contributes the rows of its own targets."""

    def __init__(self, world, device, group=None):
        self.world, self.device, self.group = world, device, group
        self.cap = 0

    def _reserve(self, nmax, width):
        if nmax > self.cap or getattr(self, "width", None) != width:
            self.cap = int(nmax * 1.05) + 1024
            self.width = width
            f64 = dict(dtype=torch.float64, device=self.device)
            self.sv = torch.zeros(self.cap, width, **f64)
            self.gv = torch.zeros(self.world * self.cap, width, **f64)
            self.si = torch.zeros(self.cap, dtype=torch.int32, device=self.device)
            self.gi = torch.zeros(self.world * self.cap, dtype=torch.int32, device=self.device)

    def exchange(self, values, targets):
        """values: [N, k] caller order, rows `targets` fresh on this rank.  On return every row holds its owner's result."""
        if self.world == 1 and not FORCE_COLLECTIVES:
            return values
        nt = torch.tensor([targets.shape[0]], dtype=torch.int64, device=self.device)
        counts = [torch.zeros_like(nt) for _ in range(self.world)]
        dist.all_gather(counts, nt, group=self.group)
        counts = [int(c.item()) for c in counts]
        self._reserve(max(counts), values.shape[1])
        n = targets.shape[0]
        self.sv[:n] = values[targets.long()]
        self.si[:n] = targets
        for g, s in ((self.gv, self.sv), (self.gi, self.si)):
            if _has_all_to_all(self.group):
                dist.all_gather_into_tensor(g, s, group=self.group)
            else:
                parts = [torch.empty_like(s) for _ in range(self.world)]
                dist.all_gather(parts, s, group=self.group)
                for r in range(self.world):
                    g[r * self.cap:(r + 1) * self.cap] = parts[r]
        for r in range(self.world):
            c = counts[r]
            values[self.gi[r * self.cap:r * self.cap + c].long()] = self.gv[r * self.cap:r * self.cap + c]
        return values
