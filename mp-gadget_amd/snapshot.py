"""Snapshot / IC wire format: MP-Gadget's "bigfile" snapshots as petaio.c writes them (petaio.c:986-1120):
one directory per particle type `0/ .. 5/` with blocks `Position f8 x3, Velocity f4 x3, Mass f4, ID u8, Potential f4,
SmoothingLength f4, Density f4, ...` and a `Header` block whose attributes carry TotNumPart[6], MassTable[6], BoxSize, Time, ...
(petaio.c:418-442).  The file IO is the library's (csrc/snapshot_io.hip through the C-ABI); this module is the thin mirror used by
tools and tests."""
import ctypes as C
import os

import numpy as np

from .engine import load_library, EngineError

_NP = {"f8": np.float64, "f4": np.float32, "i8": np.int64, "u8": np.uint64, "i4": np.int32, "u4": np.uint32, "u1": np.uint8, "i1": np.int8,
       "S1": np.uint8, "i2": np.int16, "u2": np.uint16}
# (block, dtype in the file, members) of the blocks the force path needs (petaio.c:991-998)
CORE_BLOCKS = (("Position", "f8", 3), ("Velocity", "f4", 3), ("Mass", "f4", 1), ("ID", "u8", 1))


class BlockInfo(C.Structure):
    _fields_ = [("dtype", C.c_char * 8), ("nmemb", C.c_int), ("nfile", C.c_int), ("size", C.c_int64)]


def _ck(L, rc):
    if rc:
        raise EngineError(L.mpg_last_error().decode())


def _dt(dtype):
    d = dtype.lstrip("<=|")
    if d not in _NP:
        raise EngineError("snapshot: unsupported dtype %r" % dtype)
    return d


def block_info(path, block):
    L = load_library()
    b = BlockInfo()
    _ck(L, L.mpg_bigfile_block_info(path.encode(), block.encode(), C.byref(b)))
    return dict(dtype=b.dtype.decode(), nmemb=b.nmemb, nfile=b.nfile, size=b.size)


def read_block(path, block, start=0, count=None, dtype=None):
    """numpy array [count, nmemb] (or [count]) of the block, cast to `dtype` (default: the file's)."""
    L = load_library()
    info = block_info(path, block)
    want = _dt(dtype or info["dtype"])
    n = info["size"] - start if count is None else count
    out = np.zeros((n, info["nmemb"]), _NP[want])
    _ck(L, L.mpg_bigfile_read_block(path.encode(), block.encode(), C.c_int64(start), C.c_int64(n), want.encode(), out.ctypes.data_as(C.c_void_p)))
    return out[:, 0] if info["nmemb"] == 1 else out


def write_block(path, block, data, dtype, nfile=1):
    """Writes `data` ([n] or [n, nmemb]) as a block of file dtype `dtype` split over nfile files."""
    L = load_library()
    a = np.ascontiguousarray(data)
    src = next(k for k, v in _NP.items() if v == a.dtype.type and k != "S1")
    nmemb = 1 if a.ndim == 1 else a.shape[1]
    _ck(L, L.mpg_bigfile_write_block(path.encode(), block.encode(), _dt(dtype).encode(), nmemb, int(nfile), C.c_int64(a.shape[0]), src.encode(),
                                     a.ctypes.data_as(C.c_void_p)))


def get_attr(path, block, name, dtype, nmemb=1):
    L = load_library()
    out = np.zeros(nmemb, _NP[_dt(dtype)])
    rc = L.mpg_bigfile_get_attr(path.encode(), block.encode(), name.encode(), _dt(dtype).encode(), out.ctypes.data_as(C.c_void_p), int(nmemb))
    if rc == 2:
        raise KeyError(name)
    _ck(L, rc)
    return out


def set_attr(path, block, name, value, dtype):
    L = load_library()
    d = _dt(dtype)
    a = np.frombuffer(value.encode(), np.uint8) if isinstance(value, str) else np.atleast_1d(np.asarray(value, _NP[d]))
    a = np.ascontiguousarray(a)
    _ck(L, L.mpg_bigfile_set_attr(path.encode(), block.encode(), name.encode(), d.encode(), a.ctypes.data_as(C.c_void_p), int(a.shape[0])))


def write_snapshot(path, parts, box, time, mass_table=None, nfile=1, extra_header=None):
    """parts: {ptype: dict(Position=[n,3], Velocity=[n,3], Mass=[n], ID=[n], ...)}.  Writes the Header attributes petaio.c reads back
    (petaio_read_header_internal, petaio.c:470-540) and the blocks with the reference's on-disk dtypes."""
    os.makedirs(path, exist_ok=True)
    tot = np.zeros(6, np.uint64)
    for t, d in parts.items():
        tot[int(t)] = len(d["Position"])
    set_attr(path, "Header", "TotNumPart", tot, "u8")
    set_attr(path, "Header", "TotNumPartInit", tot, "u8")
    set_attr(path, "Header", "MassTable", np.zeros(6) if mass_table is None else mass_table, "f8")
    set_attr(path, "Header", "BoxSize", box, "f8")
    set_attr(path, "Header", "Time", time, "f8")
    for k, (v, dt) in (extra_header or {}).items():
        set_attr(path, "Header", k, v, dt)
    disk = {b: dt for b, dt, _ in CORE_BLOCKS}
    disk.update(Potential="f4", SmoothingLength="f4", Density="f4", InternalEnergy="f4", GravAccel="f4", GravPM="f4")
    for t, d in parts.items():
        for name, arr in d.items():
            write_block(path, "%d/%s" % (int(t), name), arr, disk.get(name, "f8"), nfile)


def read_snapshot(path, types=(0, 1, 2, 3, 4, 5), blocks=("Position", "Velocity", "Mass", "ID")):
    """Returns (header dict, {ptype: {block: array}}): positions f8, velocities f8, masses f4 (MassTable entries fill a missing Mass
    block as petaio.c does, petaio.c:334-385), IDs u8."""
    hdr = dict(TotNumPart=get_attr(path, "Header", "TotNumPart", "u8", 6), MassTable=get_attr(path, "Header", "MassTable", "f8", 6),
               BoxSize=float(get_attr(path, "Header", "BoxSize", "f8")[0]), Time=float(get_attr(path, "Header", "Time", "f8")[0]))
    want = dict(Position="f8", Velocity="f8", Mass="f4", ID="u8")
    parts = {}
    for t in types:
        n = int(hdr["TotNumPart"][t])
        if n == 0:
            continue
        d = {}
        for b in blocks:
            name = "%d/%s" % (t, b)
            if os.path.exists(os.path.join(path, name, "header")):
                d[b] = read_block(path, name, dtype=want.get(b))
            elif b == "Mass":
                d[b] = np.full(n, hdr["MassTable"][t], np.float32)
        parts[t] = d
    return hdr, parts
