"""Long-range PM and target sharding over the GPUs of one node (one process per GPU, torch.distributed / RCCL).

The reference decomposes the mesh into 2-D pencils over all MPI ranks, ships each rank's "region" meshes to the pencils
and back (petapm.c:584-885) and lets PFFT transpose between the 1-D transform stages.  Here:

* every rank binds the same particle set; rank r owns the x-planes [r P, (r+1) P), P = Nmesh / world, of the mesh;
* a particle's CIC cloud is deposited by the owner(s) of the planes it touches, so there is no region exchange;
* one 3-D transform = local 2-D transforms, ONE all-to-all transpose, local 1-D transforms; the four inverse transforms
  only the potential is transformed back (the forces are its 4-point differences, the real-space form of the reference's
  Fourier-space force transfer), so there are TWO all-to-alls per PM step; five potential planes go to the neighbours;
* the targets of rank r - for the PM readout and for the short-range walk alike - are the particles whose base mesh
  cell lies in its slab, listed in tree (Morton) order; their accelerations are all-gathered once per step.

xGMI is point-to-point: each all-to-all is 7 contiguous blocks of Nmesh^3 / world^2 complex values per rank (134 MB at
Nmesh 1024 on 8 GPUs), large enough to run every link at its streaming rate.

This module holds the collectives and the index logic only (torch tensors as buffers); all arithmetic is in the engine.
"""
import os

import torch
import torch.distributed as dist

# MPG_FORCE_COLLECTIVES=1: issue the collectives even in a one-rank group (lets a single-GPU box exercise the RCCL code path)
FORCE_COLLECTIVES = bool(os.environ.get("MPG_FORCE_COLLECTIVES"))


A2A_MAX_ELEMENTS = 1 << 26       # elements in one all_to_all_single call, all peers together (see _all_to_all)


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


def _all_to_all(recv, send, world, group=None):
    """recv[s-th block] <- rank s's send[my block].  RCCL all_to_all_single; backends without it (gloo, used by the
    CPU-launched tests that put two ranks on one GPU) gather everything and slice."""
    if world == 1 and not FORCE_COLLECTIVES:
        recv.copy_(send)
        return
    blk = send.numel() // world
    piece = max(A2A_MAX_ELEMENTS // world, 1)
    if blk > piece:
        # RCCL 2.26 (torch 2.10 / ROCm 7) returns garbage in the second half of a message of more than 1 GiB (measured in a one-rank
        # group, where message = whole buffer: tools/a2a_selftest.py; up to 1 GiB it is exact).  Whether the limit is per peer or
        # per call could not be measured on one GPU, so a call never carries more than A2A_MAX_ELEMENTS (512 MiB of doubles) in
        # total: larger transposes go in pieces (one extra copy of each piece, ~0.1 ms per 100 MB)
        sv, rv = send.view(world, blk), recv.view(world, blk)
        for c in range(0, blk, piece):
            part = sv[:, c:c + piece].contiguous()
            got = torch.empty_like(part)
            _all_to_all(got.view(-1), part.view(-1), world, group)
            rv[:, c:c + piece] = got
        return
    if _has_all_to_all(group):
        dist.all_to_all_single(recv, send, group=group)      # errors propagate: every rank must stay in the same collective
        return
    rank = dist.get_rank(group)
    parts = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(parts, send, group=group)
    for s in range(world):
        recv.view(-1)[s * blk:(s + 1) * blk] = parts[s].view(-1)[rank * blk:(rank + 1) * blk]


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


class SlabPM:
    """gravpm_force (gravpm.c:61-119) over `world` GPUs.  `eng` must have gravpm_init_periodic and dev_bind_particles done."""

    def __init__(self, eng, box, nmesh, rank, world, device, group=None):
        if nmesh % world:
            raise ValueError("Nmesh must be a multiple of the number of GPUs")
        self.eng, self.rank, self.world, self.group = eng, rank, world, group
        self.box, self.nmesh, self.cellsize = box, nmesh, box / nmesh
        per_peer, plane = eng.dev_pm_slab_init(rank, world)
        f64 = dict(dtype=torch.float64, device=device)
        self.sendA = _scratch(2 * per_peer * world, **f64)
        self.recvA = _scratch(2 * per_peer * world, **f64)
        self.sendB = _scratch(2 * per_peer * world, **f64)        # the inverse transpose carries the potential only
        self.recvB = _scratch(2 * per_peer * world, **f64)
        self.ghost_send = _scratch(5, plane, **f64)               # first 3 planes (-> previous rank), last 2 (-> next rank)
        self.ghost_recv = _scratch(5, plane, **f64)
        self.device = device

    def targets(self, pos, order):
        """Caller indices (int32, tree order) of the particles whose base cell lies in this rank's slab.
        pos: [N,3] device tensor; order: [N] int32 tree slot -> caller index (engine.dev_tree_order)."""
        o = order.long()
        owner = slab_of_cells(pos[:, 0][o], self.cellsize, self.nmesh, self.world)
        return order[owner == self.rank].contiguous()

    def force(self, targets, gravpm, potential=None):
        """GravPM (assigned) and Potential (incremented) for `targets`; rows of other particles are left untouched."""
        e = self.eng
        e.dev_pm_slab_forward_a(self.sendA)
        _all_to_all(self.recvA, self.sendA, self.world, self.group)
        e.dev_pm_slab_forward_b(self.recvA, self.sendB)
        _all_to_all(self.recvB, self.sendB, self.world, self.group)
        e.dev_pm_slab_inverse_c(self.recvB, self.ghost_send)
        self._ghost_planes()
        e.dev_pm_slab_readout(self.ghost_recv, targets, gravpm, potential)

    def power_spectrum(self, BoxSize_in_MPC):
        """(kk, Power, Nmodes) of the last force() call: the raw sums of this rank's Fourier rows, summed over the ranks
        (the MPI_Allreduce of powerspectrum_sum, powerspectrum.c:68-72), then powerspectrum_sum."""
        acc = torch.zeros(2 * self.nmesh + 1, dtype=torch.float64, device=self.device)
        modes = torch.zeros(self.nmesh, dtype=torch.int64, device=self.device)
        self.eng.dev_gravpm_powerspectrum_raw(acc, modes)
        if self.world > 1 or FORCE_COLLECTIVES:
            dist.all_reduce(acc, group=self.group)
            dist.all_reduce(modes, group=self.group)
        return self.eng.powerspectrum_sum(acc.cpu().numpy(), modes.cpu().numpy(), BoxSize_in_MPC)

    def _ghost_planes(self):
        """ghost_recv <- [first 3 planes of rank+1 | last 2 planes of rank-1] (periodic): the force stencil reaches two planes
        either way and the CIC readout one plane up.  One personalised exchange (both neighbours may be the same rank, or this
        rank itself)."""
        w, r = self.world, self.rank
        if w == 1 and not FORCE_COLLECTIVES:
            self.ghost_recv.copy_(self.ghost_send)
            return
        prev, nxt = (r - 1) % w, (r + 1) % w
        # rows for each destination in rank order; within one destination: planes it uses as its upper ghosts first (our first 3),
        # then those it uses as its lower ghosts (our last 2)
        rows, counts = [], [0] * w
        for d in range(w):
            if d == prev:
                rows.append(self.ghost_send[0:3])
                counts[d] += 3
            if d == nxt:
                rows.append(self.ghost_send[3:5])
                counts[d] += 2
        got = exchange_rows(torch.cat(rows).contiguous(), counts, w, self.group)
        # received in source-rank order; from a source that is both our next and our previous rank: its first 3, then its last 2
        o = 0
        for s_ in range(w):
            if s_ == nxt:
                self.ghost_recv[0:3] = got[o:o + 3]
                o += 3
            if s_ == prev:
                self.ghost_recv[3:5] = got[o:o + 2]
                o += 2


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
