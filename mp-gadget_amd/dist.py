"""The force step on several ranks through the library's own choreography (csrc/dist.hip, `mpg_dist_*` of
include/mpgadget_hip.h): Python here only supplies the communicator.

The reference's entry points are collective over MPI_COMM_WORLD; the C-ABI takes the three collectives it needs as callbacks
(`mpg_comm`).  `TorchComm` implements them over torch.distributed - RCCL over xGMI when the process group's backend is "nccl"
(device pointers go straight into all_to_all_single), gloo on host memory in the CPU-launched tests that put several ranks on
one GPU (the library then stages through pinned host buffers).  A C caller supplies MPI_Allreduce / MPI_Alltoall /
MPI_Alltoallv instead (shim/)."""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import engine as E
from .domain_peano import TopNode

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int)
A2A_I64_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64))
A2AV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64),
                      C.POINTER(C.c_int64), C.c_int)


BIND_STREAM_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)


class MpgComm(C.Structure):     # mpg_comm, include/mpgadget_hip.h
    _fields_ = [("ctx", C.c_void_p), ("ThisTask", C.c_int), ("NTask", C.c_int), ("device_buffers", C.c_int),
                ("allreduce", ALLREDUCE_FN), ("alltoall_i64", A2A_I64_FN), ("alltoallv", A2AV_FN), ("bind_stream", BIND_STREAM_FN)]


class _DevMem:
    """a raw device pointer as a __cuda_array_interface__ object: torch.as_tensor wraps it without a copy"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def _bytes_tensor(ptr, nbytes, on_device, device):
    if nbytes == 0:
        return torch.empty(0, dtype=torch.uint8, device=device if on_device else "cpu")
    if on_device:
        return torch.as_tensor(_DevMem(ptr, nbytes), device=device)
    return torch.frombuffer((C.c_char * int(nbytes)).from_address(int(ptr)), dtype=torch.uint8)


# RCCL 2.26 (torch 2.10 / ROCm 7) returned garbage in the second half of an all_to_all_single message above 1 GiB (measured with
# a one-rank group, tools/a2a_selftest.py): no call carries more than this many bytes in total; larger exchanges go in pieces
A2A_MAX_BYTES = 1 << 29


class TorchComm:
    """mpg_comm over a torch.distributed process group (one process per GPU)."""

    def __init__(self, device, group=None):
        self.group, self.device = group, device
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.on_device = dist.get_backend(group) == "nccl"
        self.errors = []
        self._fns = (ALLREDUCE_FN(self._allreduce), A2A_I64_FN(self._alltoall_i64), A2AV_FN(self._alltoallv))   # keep alive
        self.struct = MpgComm(None, self.rank, self.world, 1 if self.on_device else 0, *self._fns)

    def _guard(self, f, *a):
        try:
            f(*a)
            return 0
        except Exception as e:      # a callback must not raise through the C frames
            self.errors.append(repr(e))
            return 1

    def _allreduce(self, ctx, buf, count, dtype, op, on_device):
        def go():
            t = _bytes_tensor(buf, 8 * count, bool(on_device), self.device).view(torch.int64 if dtype else torch.float64)
            red = dist.ReduceOp.MAX if op else dist.ReduceOp.SUM
            if self.on_device and not on_device:      # RCCL reduces device memory only: a small host array goes through the GPU
                g = t.to(self.device)
                dist.all_reduce(g, op=red, group=self.group)
                t.copy_(g.cpu())
            else:
                dist.all_reduce(t, op=red, group=self.group)
            if on_device:
                torch.cuda.current_stream().synchronize()
        return self._guard(go)

    def _alltoall_i64(self, ctx, send, recv):
        def go():
            s = torch.tensor([send[i] for i in range(self.world)], dtype=torch.int64)
            r = torch.empty_like(s)
            if self.on_device:
                s, r = s.to(self.device), r.to(self.device)
            dist.all_to_all_single(r, s, group=self.group)
            r = r.cpu()
            for i in range(self.world):
                recv[i] = int(r[i])
        return self._guard(go)

    def _alltoallv(self, ctx, send, sbytes, sdispls, recv, rbytes, rdispls, on_device):
        def go():
            w = self.world
            sb, sd = [sbytes[i] for i in range(w)], [sdispls[i] for i in range(w)]
            rb, rd = [rbytes[i] for i in range(w)], [rdispls[i] for i in range(w)]
            S = _bytes_tensor(send, max((d + b for d, b in zip(sd, sb)), default=0), bool(on_device), self.device)
            R = _bytes_tensor(recv, max((d + b for d, b in zip(rd, rb)), default=0), bool(on_device), self.device)
            Rhost = None
            if self.on_device and not on_device:      # RCCL moves device memory only: small host blocks (trees, samples) go through the GPU
                Rhost, S, R = R, S.to(self.device), torch.empty(R.shape[0], dtype=torch.uint8, device=self.device)
            packed = lambda b, d: all(d[i] == sum(b[:i]) for i in range(w))
            piece = max(A2A_MAX_BYTES // w, 1)
            # the number of pieces must be the same on every rank (each piece is one collective): the largest block of ANY rank decides.
            # (Round 2 first took the local maximum; ranks whose largest blocks fell on different sides of a piece boundary then issued
            # different numbers of collectives - gloo aborts the process on the size mismatch, RCCL would hang.)
            big = torch.tensor([max(max(sb), max(rb))], dtype=torch.int64, device=self.device if self.on_device else "cpu")
            dist.all_reduce(big, op=dist.ReduceOp.MAX, group=self.group)
            nchunk = max((int(big.item()) + piece - 1) // piece, 1)
            if nchunk == 1 and packed(sb, sd) and packed(rb, rd):
                dist.all_to_all_single(R[:sum(rb)], S[:sum(sb)], rb, sb, group=self.group)
            else:
                for c in range(nchunk):        # blocks in rank order, each block cut into pieces
                    cs = [min(max(b - c * piece, 0), piece) for b in sb]
                    cr = [min(max(b - c * piece, 0), piece) for b in rb]
                    src = torch.cat([S[d + c * piece:d + c * piece + n] for d, n in zip(sd, cs)]) if sum(cs) else S[:0]
                    dst = torch.empty(sum(cr), dtype=torch.uint8, device=S.device)
                    dist.all_to_all_single(dst, src, cr, cs, group=self.group)
                    o = 0
                    for d, n in zip(rd, cr):
                        R[d + c * piece:d + c * piece + n] = dst[o:o + n]
                        o += n
            if Rhost is not None:
                Rhost.copy_(R.cpu())
            if on_device:
                torch.cuda.current_stream().synchronize()   # the library continues on ITS stream
        return self._guard(go)


class RcclComm:
    """mpg_comm on the library's NATIVE RCCL communicator (csrc/rccl_comm.hip): ncclSend / ncclRecv / ncclAllReduce on the engine's
    stream, no Python frame in any collective.  Python's only part is the bootstrap: rank 0's ncclUniqueId reaches the other ranks
    through the torch.distributed group the launcher set up (any backend)."""

    def __init__(self, lib, device, group=None, selftest_bytes=0):
        self.lib, self.device, self.errors = lib, device, []
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        lib.mpg_rccl_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p, C.c_int]
        lib.mpg_rccl_comm.argtypes = [C.c_void_p, C.POINTER(MpgComm)]
        lib.mpg_rccl_selftest.argtypes = [C.c_void_p, C.c_int64]
        lib.mpg_rccl_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int)]
        lib.mpg_rccl_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.mpg_rccl_destroy.argtypes = [C.c_void_p]
        lib.mpg_rccl_destroy.restype = None
        lib.mpg_rccl_last_error.argtypes = [C.c_void_p]
        lib.mpg_rccl_last_error.restype = C.c_char_p
        idb = (C.c_char * 128)()
        ok = 1
        if self.rank == 0:
            ok = 0 if lib.mpg_rccl_get_unique_id(idb) else 1
        on_gpu = dist.get_backend(group) == "nccl"
        t = torch.tensor([ok] + list(idb.raw), dtype=torch.int32, device=device if on_gpu else "cpu")
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        t = t.cpu()
        if int(t[0]) != 1:
            raise E.EngineError("RCCL: rank 0 could not create a unique id: " + lib.mpg_last_error().decode())
        idb.raw = bytes(int(v) for v in t[1:])
        self.h = C.c_void_p()
        dev_index = device.index if device.index is not None else torch.cuda.current_device()
        if lib.mpg_rccl_create(C.byref(self.h), self.rank, self.world, idb, int(dev_index)):
            raise E.EngineError("RCCL: " + lib.mpg_last_error().decode())
        if selftest_bytes >= 0 and lib.mpg_rccl_selftest(self.h, C.c_int64(selftest_bytes)):
            msg = lib.mpg_last_error().decode()
            self.close()
            raise E.EngineError("RCCL self-test: " + msg)
        self.struct = MpgComm()
        if lib.mpg_rccl_comm(self.h, C.byref(self.struct)):
            raise E.EngineError("RCCL: " + lib.mpg_last_error().decode())
        self.on_device = True

    def stats(self):
        calls, sent, ver = (C.c_int64 * 3)(), C.c_int64(0), C.c_int(0)
        self.lib.mpg_rccl_stats(self.h, calls, C.byref(sent), C.byref(ver))
        return dict(allreduce=calls[0], alltoall_i64=calls[1], alltoallv=calls[2], bytes_sent=sent.value, rccl_version=ver.value)

    def info(self):
        """what RCCL itself reports for the communicator: ncclCommCount, ncclCommUserRank, the HIP device"""
        n, r, d = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        if self.lib.mpg_rccl_comm_info(self.h, C.byref(n), C.byref(r), C.byref(d)):
            raise E.EngineError("RCCL: " + self.lib.mpg_last_error().decode())
        return dict(nranks=n.value, rank=r.value, device=d.value)

    def last_error(self):
        return (self.lib.mpg_rccl_last_error(self.h) or b"").decode() if self.h else ""

    def close(self):
        if self.h:
            self.lib.mpg_rccl_destroy(self.h)
            self.h = C.c_void_p()


class LocalComm:
    """one rank, no process group: NULL callbacks (the library then copies locally)"""

    def __init__(self):
        self.errors = []
        self.rank, self.world = 0, 1
        self.struct = MpgComm(None, 0, 1, 0, ALLREDUCE_FN(), A2A_I64_FN(), A2AV_FN())


def _bind(lib):
    if getattr(lib, "_dist_bound", False):
        return
    lib.mpg_dist_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(MpgComm)]
    lib.mpg_dist_destroy.argtypes = [C.c_void_p]
    lib.mpg_dist_destroy.restype = None
    lib.mpg_dist_set_domain.argtypes = [C.c_void_p, C.c_double, C.POINTER(TopNode), C.c_int, C.POINTER(C.c_int), C.c_int, C.c_double, C.c_int]
    lib.mpg_dist_gravity_step.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 7 + [C.c_double]
    lib.mpg_dist_dev_force_tree_build.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.mpg_dist_dev_density.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.mpg_dist_dev_hydro_force.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.mpg_dist_get_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.mpg_dist_get_times.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib._dist_bound = True


class DistForce:
    """mpg_dist: gravpm_force + force_tree_full + grav_short_tree for particles distributed over the ranks by a Peano-Hilbert
    domain decomposition (domain_peano.PeanoDomain)."""

    def __init__(self, eng, comm):
        self.eng, self.lib, self.comm = eng, eng.lib, comm
        _bind(self.lib)
        self.h = C.c_void_p()
        self._ck(self.lib.mpg_dist_create(C.byref(self.h), eng.h, C.byref(comm.struct)))

    def _ck(self, rc):
        if rc:
            msg = self.lib.mpg_last_error().decode()
            if self.comm.errors:
                msg += " | callback: " + "; ".join(self.comm.errors[-3:])
            if hasattr(self.comm, "last_error") and self.comm.last_error():
                msg += " | RCCL: " + self.comm.last_error()
            raise E.EngineError(msg)

    def set_domain(self, dom, margin, La=0):
        """dom: a decomposed PeanoDomain (TopNodes, leaf_task); margin: at least Rcut in length units"""
        tn = np.ascontiguousarray(dom.TopNodes)
        lt = np.ascontiguousarray(dom.leaf_task, np.int32)
        self._keep = (tn, lt)
        self._ck(self.lib.mpg_dist_set_domain(self.h, C.c_double(dom.box), tn.ctypes.data_as(C.POINTER(TopNode)), int(dom.NTopNodes),
                                              lt.ctypes.data_as(C.POINTER(C.c_int)), int(dom.NTopLeaves), C.c_double(margin), int(La)))

    def gravity_step(self, pos, mass, accel, gravpm, potential=None, oldacc=None, prev_accel=None, rho0=0.0):
        """one force step for this rank's own particles (device tensors, n_own rows)"""
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        self._ck(self.lib.mpg_dist_gravity_step(self.h, C.c_int64(pos.shape[0]), p(pos), p(mass), p(oldacc), p(prev_accel), p(accel), p(gravpm),
                                                p(potential), C.c_double(rho0)))

    def gravpm_force(self, pos, mass, gravpm, potential=None):
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        self._ck(self.lib.mpg_dist_dev_gravpm_force(self.h, C.c_int64(pos.shape[0]), p(pos), p(mass), p(gravpm), p(potential)))

    def grav_short_tree(self, accel, oldacc=None, prev_accel=None, gravpm=None, potential=None, rho0=0.0, active=None):
        """the walk on the tree of the last force_tree_build; active: int32 device tensor of own-particle indices (a sub-step's
        ActiveParticle) or None for all"""
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        self._ck(self.lib.mpg_dist_dev_grav_short_tree_active(self.h, p(active), C.c_int64(0 if active is None else active.shape[0]), p(oldacc),
                                                              p(prev_accel), p(gravpm), p(accel), p(potential), C.c_double(rho0)))

    def grav_short_tree_active_tree(self, pos, mass, accel, oldacc=None, potential=None, rho0=0.0):
        """hierarchical gravity: the active particles (compacted device tensors of this rank's members) on the tree of the active set"""
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        self._ck(self.lib.mpg_dist_dev_grav_short_tree_active_tree(self.h, C.c_int64(pos.shape[0]), p(pos), p(mass), p(oldacc), p(accel),
                                                                   p(potential), C.c_double(rho0)))

    # ---- domain_decompose_full + domain_exchange through the library (the C++ form of domain_peano.PeanoDomain)
    def domain_decompose(self, pos, box, garbage=None, overdecomposition=4, global_sorting=True, cost=None):
        """Returns (NTopNodes, NTopLeaves); the decomposition stays in the library (domain_get copies it out)."""
        ntn, ntl = C.c_int(0), C.c_int(0)
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        self.lib.mpg_dist_domain_decompose.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_void_p,
                                                       C.POINTER(C.c_int), C.POINTER(C.c_int)]
        self._ck(self.lib.mpg_dist_domain_decompose(self.h, C.c_int64(pos.shape[0]), p(pos), p(garbage), C.c_double(box), int(overdecomposition),
                                                    int(bool(global_sorting)), p(cost), C.byref(ntn), C.byref(ntl)))
        self._dom_sizes = (ntn.value, ntl.value)
        return self._dom_sizes

    def domain_get(self):
        from .domain_peano import TOPNODE_DTYPE
        ntn, ntl = self._dom_sizes
        w = self.comm.world
        tn = np.zeros(ntn, TOPNODE_DTYPE)
        lt, st, en, cnt = np.zeros(ntl, np.int32), np.zeros(w, np.int32), np.zeros(w, np.int32), np.zeros(ntl, np.int64)
        P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
        self.lib.mpg_dist_domain_get.argtypes = [C.c_void_p, C.POINTER(TopNode), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                                 C.POINTER(C.c_int64)]
        self._ck(self.lib.mpg_dist_domain_get(self.h, P(tn, TopNode), P(lt, C.c_int), P(st, C.c_int), P(en, C.c_int), P(cnt, C.c_int64)))
        return dict(TopNodes=tn, leaf_task=lt, StartLeaf=st, EndLeaf=en, TopLeafCount=cnt)

    def domain_exchange(self, *columns):
        """domain_exchange: every live particle to the task of its TopLeaf; returns this rank's columns afterwards (copies)."""
        n, k = int(columns[0].shape[0]), len(columns)
        cols = [c.contiguous() for c in columns]
        ptrs = (C.c_void_p * k)(*[c.data_ptr() for c in cols])
        widths = (C.c_int * k)(*[c.element_size() * int(np.prod(c.shape[1:], dtype=np.int64)) for c in cols])
        outp = (C.c_void_p * k)()
        nn = C.c_int64(0)
        self.lib.mpg_dist_domain_exchange.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]
        self._ck(self.lib.mpg_dist_domain_exchange(self.h, C.c_int64(n), k, ptrs, widths, C.byref(nn), outp))
        out = []
        dev = self.comm_device()
        for c, w, ptr in zip(cols, widths, outp):
            shape = (nn.value,) + tuple(c.shape[1:])
            if nn.value == 0:
                out.append(torch.empty(shape, dtype=c.dtype, device=dev))
            else:
                out.append(torch.as_tensor(_DevMem(ptr, w * nn.value), device=dev).view(c.dtype).reshape(shape).clone())
        return out

    def domain_maintain(self, pos, box, garbage=None):
        """domain_maintain after a drift; returns the number of particles about to leave this rank"""
        nl = C.c_int64(0)
        self.lib.mpg_dist_domain_maintain.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_double, C.POINTER(C.c_int64)]
        self._ck(self.lib.mpg_dist_domain_maintain(self.h, C.c_int64(pos.shape[0]), C.c_void_p(pos.data_ptr()),
                                                   None if garbage is None else C.c_void_p(garbage.data_ptr()), C.c_double(box), C.byref(nl)))
        return nl.value

    def use_decomposition(self, box, margin, La=0):
        self.lib.mpg_dist_use_decomposition.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int]
        self._ck(self.lib.mpg_dist_use_decomposition(self.h, C.c_double(box), C.c_double(margin), int(La)))

    def fof_fof(self, pos, mass, ids, linking_length, min_length=32, type=None, vel=None, primary=2, secondary=1 + 16 + 32):
        """fof_fof with groups spanning ranks.  Returns (grnr int64 device tensor over the own particles with GLOBAL group numbers,
        total number of groups, table of the groups this rank keeps as a dict of numpy arrays)."""
        n = int(pos.shape[0])
        par = E.FofParams(int(primary), int(secondary), float(linking_length), int(min_length))
        grnr = torch.zeros(n, dtype=torch.int64, device=pos.device)
        tot, here = C.c_int64(0), C.c_int64(0)
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        self.lib.mpg_dist_dev_fof_fof.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 7 + [C.POINTER(C.c_int64)] * 2
        self._ck(self.lib.mpg_dist_dev_fof_fof(self.h, C.c_int64(n), p(pos), p(mass), p(type), p(ids), p(vel), C.cast(C.pointer(par), C.c_void_p), p(grnr),
                                               C.byref(tot), C.byref(here)))
        m = here.value
        g = dict(MinID=np.zeros(m, np.uint64), Length=np.zeros(m, np.int32), GrNr=np.zeros(m, np.int32), LenType=np.zeros((m, 6), np.int32),
                 Mass=np.zeros(m), MassType=np.zeros((m, 6)), CM=np.zeros((m, 3)), Vel=np.zeros((m, 3)), Jmom=np.zeros((m, 3)),
                 Imom=np.zeros((m, 3, 3)), FirstPos=np.zeros((m, 3), np.float32))
        out = E.FofGroupsC(*[g[k].ctypes.data for k in ("MinID", "Length", "GrNr", "LenType", "Mass", "MassType", "CM", "Vel", "Jmom", "Imom", "FirstPos")])
        self.lib.mpg_dist_fof_groups.argtypes = [C.c_void_p, C.c_void_p]
        self._ck(self.lib.mpg_dist_fof_groups(self.h, C.byref(out)))
        return grnr, tot.value, g

    def force_tree_build(self, pos, mass):
        """mpg_dist_dev_force_tree_build alone: ghost import, local tree, global top (what the SPH loops need when no gravity step
        ran on this particle set)"""
        self._ck(self.lib.mpg_dist_dev_force_tree_build(self.h, C.c_int64(pos.shape[0]), C.c_void_p(pos.data_ptr()), C.c_void_p(mass.data_ptr())))

    def density(self, type, arrays, times, update_hsml=1, DoEgyDensity=0, active=None):
        """density() for the rank's own gas (active: int32 device tensor of own indices, a sub-step's list, or None for all); type: uint8
        device tensor, arrays: dict of device tensors over the own particles"""
        a = self.eng._sph_arrays(arrays)
        ap = None if active is None else C.c_void_p(active.data_ptr())
        self._ck(self.lib.mpg_dist_dev_density_active(self.h, C.c_int64(type.shape[0]), C.c_void_p(type.data_ptr()), C.byref(a), C.byref(times),
                                                      ap, C.c_int64(0 if active is None else active.shape[0]), int(update_hsml),
                                                      int(DoEgyDensity)))

    def hydro_force(self, n_own, arrays, times, active=None):
        a = self.eng._sph_arrays(arrays)
        ap = None if active is None else C.c_void_p(active.data_ptr())
        self._ck(self.lib.mpg_dist_dev_hydro_force_active(self.h, C.c_int64(n_own), C.byref(a), C.byref(times), ap,
                                                          C.c_int64(0 if active is None else active.shape[0])))

    # drop-in forms on the rank's particle_data records (numpy, engine.PARTICLE_DTYPE) and host SPH arrays: what shim/sph-hip.c calls
    def host_force_tree_full(self, P):
        v = self.eng._view(P)
        self._ck(self.lib.mpg_dist_force_tree_full(self.h, C.byref(v)))

    @staticmethod
    def _host_active(active):
        if active is None:
            return None, None, 0
        act = np.ascontiguousarray(active, np.int32)
        return act, act.ctypes.data_as(C.c_void_p), len(act)

    def host_grav_short_tree_active_tree(self, P, AccelStore, ActiveParticle=None, rho0=0.0):
        """hierarchical gravity through the drop-in form: AccelStore [len(P), 3] float64 (numpy) gets the active particles' accelerations"""
        v = self.eng._view(P)
        keep, ap, na = self._host_active(ActiveParticle)
        self._ck(self.lib.mpg_dist_grav_short_tree_active_tree(self.h, C.byref(v), ap, C.c_int64(na), AccelStore.ctypes.data_as(C.c_void_p),
                                                               C.c_double(rho0)))

    def host_density(self, P, arrays, times, update_hsml=1, DoEgyDensity=0, ActiveParticle=None):
        v = self.eng._view(P)
        a = self.eng._sph_host_arrays(arrays)
        keep, ap, na = self._host_active(ActiveParticle)
        self._ck(self.lib.mpg_dist_density(self.h, C.byref(v), C.byref(a), C.byref(times), ap, C.c_int64(na), int(update_hsml), int(DoEgyDensity)))

    def host_hydro_force(self, P, arrays, times, ActiveParticle=None):
        v = self.eng._view(P)
        a = self.eng._sph_host_arrays(arrays)
        keep, ap, na = self._host_active(ActiveParticle)
        self._ck(self.lib.mpg_dist_hydro_force(self.h, C.byref(v), C.byref(a), C.byref(times), ap, C.c_int64(na)))

    def walk_cost(self, n_own):
        """per-particle work of the last walk for the rank's own particles (float32 device tensor, a copy): feed it to
        PeanoDomain.decompose(cost=...)"""
        self.lib.mpg_dist_walk_cost.restype = C.c_void_p
        self.lib.mpg_dist_walk_cost.argtypes = [C.c_void_p]
        ptr = self.lib.mpg_dist_walk_cost(self.h)
        if not ptr or n_own == 0:
            return torch.zeros(n_own, dtype=torch.float32, device=self.comm_device())
        return torch.as_tensor(_DevMem(ptr, 4 * n_own), device=self.comm_device()).view(torch.float32).clone()

    def comm_device(self):
        return getattr(self.comm, "device", None) or torch.device("cuda", torch.cuda.current_device())

    def stats_raw(self):
        s = (C.c_int64 * 8)()
        self._ck(self.lib.mpg_dist_get_stats(self.h, s))
        return list(s)

    def stats(self):
        s = (C.c_int64 * 8)()
        self._ck(self.lib.mpg_dist_get_stats(self.h, s))
        return dict(ghosts=s[0], pm_shipped=s[1], local=s[2], La=s[3], exchange_bytes=s[4], transpose_bytes=s[5])

    def times(self):
        t = (C.c_double * 8)()
        self._ck(self.lib.mpg_dist_get_times(self.h, t))
        return dict(pm=t[0], ghosts=t[1], tree=t[2], walk=t[3], tree_beside_pm=t[4])

    def close(self):
        if self.h:
            self.lib.mpg_dist_destroy(self.h)
            self.h = C.c_void_p()
