"""CPU restatement of the FOF group catalogue -- TEST INFRASTRUCTURE ONLY: only tests/ may import it.

Labels come from oracle/fof_oracle.c (ofof_label: primary linking + secondary attachment, fof.c:366-579, 1175-1327); this module is
the catalogue, following libgadget/fof.c:
  fof_fof                       :157-253   particles in MinID order, groups = runs of equal MinID, P[].GrNr
  fof_compile_base              :758-812   FirstPos (float), Length, groups shorter than FOFHaloMinLength dropped
  fof_assign_grnr               :1106-1155 numbered from 1 by (Length descending, MinID ascending) (radix :1495-1501)
  add_particle_to_group         :631-703   Length, LenType, Mass, MassType, CM, Vel, Jmom, Imom (relative to FirstPos, NEAREST)
  fof_finish_group_properties   :705-755
The first particle of a group (FirstPos) is the one the reference's unstable qsort happens to put first; here it is the member with
the lowest index.  Group sums depend on it only through rounding."""
import ctypes as C

import numpy as np

from . import oracle as O


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def fof_labels(orc, pos, ids, box, LL, type=None, flags=None, hsml=None, primary=2, secondary=1 + 16 + 32):
    pos = np.ascontiguousarray(pos, np.float64)
    ids = np.ascontiguousarray(ids, np.uint64)
    type = None if type is None else np.ascontiguousarray(type, np.uint8)
    flags = None if flags is None else np.ascontiguousarray(flags, np.uint8)
    hsml = None if hsml is None else np.ascontiguousarray(hsml, np.float64)
    label = np.zeros(len(pos), np.uint64)
    f = orc.lib.ofof_label
    f.restype = C.c_int
    f.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p]
    f(len(pos), _vp(pos), _vp(type), _vp(flags), _vp(ids), _vp(hsml), box, LL, primary, secondary, _vp(label))
    return label


def nearest(x, box):
    return np.where(x > 0.5 * box, x - box, np.where(x < -0.5 * box, x + box, x))


def fof_fof(orc, pos, mass, ids, box, LL, min_length, vel=None, type=None, flags=None, hsml=None, primary=2, secondary=1 + 16 + 32):
    """Returns (grnr[n] int64, groups dict in MinID order)."""
    n = len(pos)
    label = fof_labels(orc, pos, ids, box, LL, type, flags, hsml, primary, secondary)
    order = np.argsort(label, kind="stable")
    sl = label[order]
    starts = np.nonzero(np.r_[True, sl[1:] != sl[:-1]])[0] if n else np.zeros(0, np.int64)
    lengths = np.diff(np.r_[starts, n])
    keep = lengths >= min_length
    starts, lengths = starts[keep], lengths[keep]
    ng = len(starts)
    minid = sl[starts] if ng else np.zeros(0, np.uint64)
    rank = np.lexsort((minid, -lengths.astype(np.int64)))
    grnr_g = np.zeros(ng, np.int32)
    grnr_g[rank] = np.arange(1, ng + 1)
    grnr = np.full(n, -1, np.int64)
    typ = np.ones(n, np.int64) if type is None else np.asarray(type, np.int64)
    v = np.zeros((n, 3)) if vel is None else np.asarray(vel, np.float64)
    G = dict(MinID=minid, Length=lengths.astype(np.int32), GrNr=grnr_g, LenType=np.zeros((ng, 6), np.int32), Mass=np.zeros(ng),
             MassType=np.zeros((ng, 6)), CM=np.zeros((ng, 3)), Vel=np.zeros((ng, 3)), Jmom=np.zeros((ng, 3)), Imom=np.zeros((ng, 3, 3)),
             FirstPos=np.zeros((ng, 3), np.float32))
    for g in range(ng):
        mem = order[starts[g]:starts[g] + lengths[g]]
        grnr[mem] = grnr_g[g]
        first = pos[mem[0]].astype(np.float32)
        G["FirstPos"][g] = first
        f64 = first.astype(np.float64)
        m = mass[mem].astype(np.float64)
        rel = nearest(pos[mem] - f64, box)
        xyz = rel + f64
        G["Mass"][g] = m.sum()
        np.add.at(G["LenType"][g], typ[mem], 1)
        np.add.at(G["MassType"][g], typ[mem], m)
        G["CM"][g] = (m[:, None] * xyz).sum(0)
        G["Vel"][g] = (m[:, None] * v[mem]).sum(0)
        G["Jmom"][g] = (m[:, None] * np.cross(rel, v[mem])).sum(0)
        G["Imom"][g] = np.einsum("i,ij,ik->jk", m, rel, rel)
        # fof_finish_group_properties
        M = G["Mass"][g]
        G["Vel"][g] /= M
        cm = G["CM"][g] / M
        relc = nearest(cm - f64, box)
        cm = np.mod(cm, box)
        G["CM"][g] = cm
        G["Jmom"][g] -= np.cross(relc, G["Vel"][g]) * M
        G["Imom"][g] -= M * np.outer(relc, relc)
    return grnr, G
