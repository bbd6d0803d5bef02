"""Particles distributed over the GPUs of a node: x-slab domains with ghost import (one process per GPU, torch.distributed).

The reference gives every rank a set of Peano-Hilbert key ranges, builds a replicated "top-tree" whose leaves carry the
moments of the remote sub-trees, and ships walk targets to the ranks that own the nodes they open
(domain.c, forcetree.c:1106-1290, treewalk.c:325-793).  Here:

* rank r OWNS the particles whose base PM-mesh cell lies in its x-slab (the same slabs as pm_slab.py); they are its targets
  for the PM readout and for the short-range walk;
* before every force step it IMPORTS, from the ranks that own them, the particles in the columns of level-La tree cells that
  overlap its slab widened by Rcut on either side (whole columns, so every tree cell at level >= La that one of its targets
  can reach is complete and identical to the global tree's cell; cells that are missing locally lie entirely more than
  Rcut away and would be discarded by the walk on geometry alone);
* the nodes above level La also contain remote particles: their moments come from sums over all ranks
  (engine.dev_tree_top_partial -> all-reduce -> engine.dev_tree_top_set), the counterpart of the reference's top-tree;
* nothing is all-gathered: per step one personalised exchange of ghost particles (32 B each, a surface layer) and one
  all-reduce of (8^La - 1)/7 * 4 doubles.

This module holds the index logic and the collectives; all arithmetic on particles is in the engine."""
import math

import torch
import torch.distributed as dist

from . import pm_slab


def tree_column(x, box, La):
    """x-index of the level-La tree cell of each position: the reference's own floating-point descent
    (forcetree.c:get_subnode via the engine's k_keys: Pos > centre; centre +- len/4; len /= 2), so that ownership of a column
    agrees with the tree bit for bit."""
    cx = torch.full_like(x, box / 2.)
    ln = box * 1.001
    col = torch.zeros_like(x, dtype=torch.int64)
    for _ in range(La):
        q = 0.25 * ln
        b = x > cx
        cx = torch.where(b, cx + q, cx - q)
        col = col * 2 + b.to(torch.int64)
        ln *= 0.5
    return col


def needed_columns(box, nmesh, world, La, margin):
    """[world, 2^La] bool: the columns of level-La cells rank s needs = those overlapping its slab widened by `margin`
    (periodic).  Column k covers [root_lo + k w, root_lo + (k+1) w), root = the tree's root cell (1.001 Box wide)."""
    ncol = 1 << La
    w = 1.001 * box / ncol
    root_lo = box / 2. - 0.5 * 1.001 * box
    need = torch.zeros(world, ncol, dtype=torch.bool)
    slab = box / world
    for s in range(world):
        a, b = s * slab - margin, (s + 1) * slab + margin
        pieces = []
        if b - a >= box:
            pieces = [(0.0, box)]
        elif a < 0:
            pieces = [(a + box, box), (0.0, b)]
        elif b > box:
            pieces = [(a, box), (0.0, b - box)]
        else:
            pieces = [(a, b)]
        for lo, hi in pieces:
            k0 = max(0, int(math.floor((lo - root_lo) / w)) - 0)
            k1 = min(ncol - 1, int(math.floor((hi - root_lo) / w)))
            need[s, k0:k1 + 1] = True
    return need


def _pack_rows(tensors, idx):
    """Rows `idx` of several per-particle tensors ([n] or [n, k], any dtype) side by side as BYTES in one uint8 buffer
    [len(idx), row bytes]: nothing is converted, so 64-bit integer columns (IDs with the generation in bits 56+, keys) survive
    bit for bit.  The reference ships whole 160-byte particle_data records (exchange.c)."""
    cols = []
    for t in tensors:
        r = t[idx].contiguous()
        w = t.element_size()
        for d in t.shape[1:]:
            w *= int(d)
        cols.append(r.view(torch.uint8).reshape(idx.shape[0], w))      # (w spelled out: an empty send list has no -1 to infer)
    return torch.cat(cols, dim=1).contiguous() if len(cols) > 1 else cols[0].contiguous()


def _unpack_rows(buf, like):
    """Inverse of _pack_rows for the received byte buffer: one tensor per entry of `like` (shape[1:] and dtype taken from it)."""
    out, c = [], 0
    for t in like:
        w = t.element_size()
        for d in t.shape[1:]:
            w *= int(d)
        if buf.shape[0] == 0:
            out.append(torch.empty((0,) + tuple(t.shape[1:]), dtype=t.dtype, device=buf.device))
        else:
            out.append(buf[:, c:c + w].contiguous().view(t.dtype).reshape((buf.shape[0],) + tuple(t.shape[1:])))
        c += w
    return out


class SlabDomain:
    """Ownership, ghost import and the global top of the tree for one rank.  `rcut` is the short-range cut-off radius in
    length units (Rcut * Asmth * cell size, gravshort-tree.c:102)."""

    def __init__(self, eng, box, nmesh, rank, world, device, rcut, La=None, group=None, margin=None):
        self.eng, self.box, self.nmesh, self.rank, self.world, self.dev, self.group = eng, box, nmesh, rank, world, device, group
        self.cellsize = box / nmesh
        if La is None:   # column width in [rcut, 2 rcut): little over-import, few cells above the decomposition level
            La = int(math.floor(math.log2(1.001 * box / rcut)))
        self.La = max(1, min(7, La))
        # > Rcut: the walk's discard test is strict.  SPH needs margin >= the largest smoothing length (check_hsml_margin).
        self.margin = max(rcut, 2 * self.cellsize, margin or 0.0) + 0.01 * self.cellsize
        self.need = needed_columns(box, nmesh, world, self.La, self.margin).to(device)
        self.ntop_fine = 8 ** (self.La - 1)
        self.partial = torch.zeros(self.ntop_fine * 4, dtype=torch.float64, device=device)

    def select_own(self, pos):
        """Indices of the particles this rank owns (base PM cell in its slab)."""
        owner = pm_slab.slab_of_cells(pos[:, 0], self.cellsize, self.nmesh, self.world)
        return torch.nonzero(owner == self.rank).squeeze(1)

    def import_ghosts(self, own_pos, own_mass, fields=()):
        """Returns (pos, mass, *fields) of [own | ghosts]: the particle set this rank builds its trees and its PM slab from.
        `fields`: further per-particle tensors ([n_own] or [n_own, k]) to carry along.  The send lists are kept so that
        ghost_update() can refresh per-particle data of the same ghosts later in the step."""
        if self.world == 1 and not pm_slab.FORCE_COLLECTIVES:
            self.send_idx, self.send_counts = None, None
            return (own_pos, own_mass) + tuple(fields)
        col = tree_column(own_pos[:, 0], self.box, self.La)
        idxs, counts = [], []
        for d in range(self.world):
            if d == self.rank:
                counts.append(0)
                continue
            idx = torch.nonzero(self.need[d][col]).squeeze(1)
            counts.append(int(idx.shape[0]))
            idxs.append(idx)
        self.send_idx = torch.cat(idxs) if idxs else torch.zeros(0, dtype=torch.int64, device=self.dev)
        self.send_counts = counts
        self.count_matrix = pm_slab.count_matrix(counts, self.world, self.dev, self.group)
        own = (own_pos, own_mass) + tuple(fields)                 # one message per peer carries all fields
        got = pm_slab.exchange_rows(_pack_rows(own, self.send_idx), self.send_counts, self.world, self.group, self.count_matrix)
        return tuple(torch.cat([t, g]).contiguous() for t, g in zip(own, _unpack_rows(got, own)))

    def ghost_update(self, own_t):
        """Rows of `own_t` ([n_own] or [n_own, k], any dtype) for this rank's ghosts, fetched from their owners, in ghost order."""
        if self.world == 1 and not pm_slab.FORCE_COLLECTIVES:
            return own_t[:0]
        got = pm_slab.exchange_rows(_pack_rows((own_t,), self.send_idx), self.send_counts, self.world, self.group, self.count_matrix)
        return _unpack_rows(got, (own_t,))[0]

    def ghost_update_many(self, own_tensors):
        """ghost_update for several tensors with one message per peer; returns the list of ghost rows."""
        if self.world == 1 and not pm_slab.FORCE_COLLECTIVES:
            return [t[:0] for t in own_tensors]
        got = pm_slab.exchange_rows(_pack_rows(tuple(own_tensors), self.send_idx), self.send_counts, self.world, self.group, self.count_matrix)
        return _unpack_rows(got, tuple(own_tensors))

    def migrate(self, own_pos, fields=()):
        """After a drift: particles whose base PM cell has left this rank's slab go to their new owner, with all the per-particle
        tensors in `fields` ([n_own] or [n_own, k]; the reference: domain_exchange, exchange.c).  Returns (pos, *fields) of the
        new own set: the particles that stayed, in their old order, followed by the arrivals in rank order."""
        if self.world == 1 and not pm_slab.FORCE_COLLECTIVES:
            return (own_pos,) + tuple(fields)
        owner = pm_slab.slab_of_cells(own_pos[:, 0], self.cellsize, self.nmesh, self.world)
        stay = torch.nonzero(owner == self.rank).squeeze(1)
        idxs, counts = [], []
        for d in range(self.world):
            idx = torch.nonzero(owner == d).squeeze(1) if d != self.rank else stay[:0]
            idxs.append(idx)
            counts.append(int(idx.shape[0]))
        send_idx = torch.cat(idxs)
        allc = pm_slab.count_matrix(counts, self.world, self.dev, self.group)
        own = (own_pos,) + tuple(fields)
        got = _unpack_rows(pm_slab.exchange_rows(_pack_rows(own, send_idx), counts, self.world, self.group, allc), own)
        return tuple(torch.cat([t[stay], g]).contiguous() for t, g in zip(own, got))

    def check_hsml_margin(self, own_hsml):
        """The SPH loops on the distributed set need every neighbour within max(Hsml_i, Hsml_j) of an own gas particle to be
        local: the largest smoothing length of any rank must not exceed the import margin."""
        h = own_hsml.max().reshape(1) if own_hsml.numel() else torch.zeros(1, dtype=torch.float64, device=self.dev)
        if self.world > 1 or pm_slab.FORCE_COLLECTIVES:
            dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.group)
        if float(h.item()) > self.margin:
            raise RuntimeError("largest smoothing length %g exceeds the ghost margin %g" % (float(h.item()), self.margin))

    def set_global_top(self, n_own):
        """Moments of the tree nodes above level La from the sums over all ranks (call after dev_force_tree_build)."""
        self.eng.dev_tree_top_partial(self.La, n_own, self.partial)
        if self.world > 1 or pm_slab.FORCE_COLLECTIVES:
            dist.all_reduce(self.partial, group=self.group)
        levels = [self.partial.view(self.ntop_fine, 4)]
        for _ in range(self.La - 1):                       # a parent's 8 children are consecutive (octant-path numbering)
            levels.append(levels[-1].view(-1, 8, 4).sum(1))
        sums = torch.cat(levels[::-1]).contiguous()        # level 0 first
        self.eng.dev_tree_top_set(self.La, sums)

    def own_targets(self, n_own, n_local):
        """Own particles in tree order (int32 caller indices into the [own | ghosts] arrays)."""
        order = self.eng.dev_tree_order(n_local, self.dev)
        return order[order < n_own].contiguous()
