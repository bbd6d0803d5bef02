"""Snapshot / IC wire format (csrc/snapshot_io.hip through the C-ABI; host IO, no GPU): round trips, the on-disk layout, and
interoperability BOTH WAYS with the reference's own library (depends/bigfile/src/bigfile.c built in place into oracle/_ref)."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def snap(pkg):
    import importlib
    return importlib.import_module("mp-gadget_amd.snapshot")


def particles(n, seed=0):
    rng = np.random.RandomState(seed)
    return dict(Position=rng.random_sample((n, 3)) * 25000.0, Velocity=rng.standard_normal((n, 3)).astype(np.float64) * 100,
                Mass=np.full(n, 0.0123, np.float32), ID=rng.permutation(n).astype(np.uint64) + (1 << 40))


def test_round_trip_and_layout(snap, tmp_path):
    path = str(tmp_path / "PART_000")
    dm, gas = particles(1001, 1), particles(77, 2)
    gas["SmoothingLength"] = np.linspace(1, 2, 77)
    snap.write_snapshot(path, {1: dm, 0: gas}, box=25000.0, time=0.1, nfile=3, extra_header={"HubbleParam": (0.697, "f8"), "CodeVersion": ("abc", "S1")})
    # layout: header text, three data files with an even split, attr-v2 lines
    hdr = open(os.path.join(path, "1", "Position", "header")).read().split("\n")
    assert hdr[:3] == ["DTYPE: <f8", "NMEMB: 3", "NFILE: 3"]
    sizes = [int(l.split(":")[1]) for l in hdr[3:6]]
    assert sizes == [1001 * (i + 1) // 3 - 1001 * i // 3 for i in range(3)]
    raw = np.fromfile(os.path.join(path, "1", "Position", "000001"), np.float64).reshape(-1, 3)
    assert np.array_equal(raw, dm["Position"][sizes[0]:sizes[0] + sizes[1]])
    cks = [int(l.split(":")[2]) for l in hdr[3:6]]
    assert cks[1] == int(raw.view(np.uint8).astype(np.uint64).sum() & 0xffffffff)          # sysv byte sum (bigfile.c:1420-1428)
    assert os.path.getsize(os.path.join(path, "1", "Velocity", "000000")) == sizes[0] * 3 * 4   # stored as f4
    attr = open(os.path.join(path, "Header", "attr-v2")).read()
    assert "TotNumPart <u8 6 " in attr and "BoxSize <f8 1 " in attr and "#HUMANE [ 25000 ]" in attr and "CodeVersion <S1 3 616263 #HUMANE [ abc ]" in attr
    # read back
    h, parts = snap.read_snapshot(path)
    assert h["BoxSize"] == 25000.0 and h["Time"] == 0.1 and h["TotNumPart"].tolist() == [77, 1001, 0, 0, 0, 0]
    assert np.array_equal(parts[1]["Position"], dm["Position"]) and np.array_equal(parts[1]["ID"], dm["ID"])
    assert np.array_equal(parts[1]["Velocity"], dm["Velocity"].astype(np.float32).astype(np.float64))
    assert np.array_equal(parts[0]["Mass"], gas["Mass"])
    assert np.allclose(snap.read_block(path, "0/SmoothingLength", dtype="f8"), gas["SmoothingLength"], rtol=1e-7)
    # partial reads across the file boundaries, with a cast
    got = snap.read_block(path, "1/Position", start=300, count=500, dtype="f4")
    assert got.dtype == np.float32 and np.array_equal(got, dm["Position"][300:800].astype(np.float32))
    assert snap.get_attr(path, "Header", "HubbleParam", "f8")[0] == 0.697
    with pytest.raises(KeyError):
        snap.get_attr(path, "Header", "NoSuchThing", "f8")
    with pytest.raises(Exception):
        snap.read_block(path, "1/Position", start=900, count=200)
    with pytest.raises(Exception):
        snap.block_info(path, "1/Nothing")


def test_mass_table_fills_missing_mass_block(snap, tmp_path):
    path = str(tmp_path / "IC")
    dm = particles(10)
    del dm["Mass"]
    snap.write_snapshot(path, {1: dm}, box=100.0, time=0.01, mass_table=[0, 0.5, 0, 0, 0, 0])
    h, parts = snap.read_snapshot(path)
    assert np.all(parts[1]["Mass"] == np.float32(0.5)) and len(parts[1]["Mass"]) == 10


# ---- the reference's own library (built in place, oracle/Makefile) -------------------------------------------------------------

def ref_lib():
    path = os.path.join(ROOT, "oracle", "_ref", "libref_bigfile.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libref_bigfile.so not built (reference tree absent)")
    L = C.CDLL(path)
    L.big_file_get_error_message.restype = C.c_char_p
    return L


class Opaque(C.Structure):
    _fields_ = [("raw", C.c_char * 4096)]      # BigFile / BigBlock / BigBlockPtr / BigArray: opaque, over-sized


def ref_write_block(L, fpath, block, arr, dtype, nfile):
    bf, bb, ptr, ba = Opaque(), Opaque(), Opaque(), Opaque()
    assert L.big_file_create(C.byref(bf), fpath.encode()) == 0, L.big_file_get_error_message()
    n = arr.shape[0]
    nmemb = 1 if arr.ndim == 1 else arr.shape[1]
    fsize = (C.c_size_t * nfile)(*[n * (i + 1) // nfile - n * i // nfile for i in range(nfile)])
    assert L.big_file_create_block(C.byref(bf), C.byref(bb), block.encode(), dtype.encode(), nmemb, nfile, fsize) == 0, L.big_file_get_error_message()
    dims = (C.c_size_t * 2)(n, nmemb)
    src = {np.float64: b"f8", np.float32: b"f4", np.uint64: b"u8"}[arr.dtype.type]
    assert L.big_array_init(C.byref(ba), arr.ctypes.data_as(C.c_void_p), src, 2, dims, None) == 0
    assert L.big_block_seek(C.byref(bb), C.byref(ptr), C.c_ssize_t(0)) == 0
    assert L.big_block_write(C.byref(bb), C.byref(ptr), C.byref(ba)) == 0, L.big_file_get_error_message()
    assert L.big_block_close(C.byref(bb)) == 0
    L.big_file_close(C.byref(bf))


def ref_set_attr(L, fpath, block, name, arr, dtype):
    bf, bb = Opaque(), Opaque()
    assert L.big_file_create(C.byref(bf), fpath.encode()) == 0
    if os.path.exists(os.path.join(fpath, block, "header")):
        assert L.big_file_open_block(C.byref(bf), C.byref(bb), block.encode()) == 0
    else:
        assert L.big_file_create_block(C.byref(bf), C.byref(bb), block.encode(), None, 0, 0, None) == 0
    a = np.ascontiguousarray(arr)
    assert L.big_block_set_attr(C.byref(bb), name.encode(), a.ctypes.data_as(C.c_void_p), dtype.encode(), int(a.shape[0])) == 0
    assert L.big_block_close(C.byref(bb)) == 0
    L.big_file_close(C.byref(bf))


def ref_read_block(L, fpath, block, n, nmemb, npdtype, dtype):
    bf, bb, ptr, ba = Opaque(), Opaque(), Opaque(), Opaque()
    assert L.big_file_open(C.byref(bf), fpath.encode()) == 0, L.big_file_get_error_message()
    assert L.big_file_open_block(C.byref(bf), C.byref(bb), block.encode()) == 0, L.big_file_get_error_message()
    out = np.zeros((n, nmemb), npdtype)
    dims = (C.c_size_t * 2)(n, nmemb)
    assert L.big_array_init(C.byref(ba), out.ctypes.data_as(C.c_void_p), dtype.encode(), 2, dims, None) == 0
    assert L.big_block_seek(C.byref(bb), C.byref(ptr), C.c_ssize_t(0)) == 0
    assert L.big_block_read(C.byref(bb), C.byref(ptr), C.byref(ba)) == 0, L.big_file_get_error_message()
    L.big_block_close(C.byref(bb))
    L.big_file_close(C.byref(bf))
    return out


def ref_get_attr(L, fpath, block, name, npdtype, dtype, nmemb):
    bf, bb = Opaque(), Opaque()
    assert L.big_file_open(C.byref(bf), fpath.encode()) == 0
    assert L.big_file_open_block(C.byref(bf), C.byref(bb), block.encode()) == 0
    out = np.zeros(nmemb, npdtype)
    assert L.big_block_get_attr(C.byref(bb), name.encode(), out.ctypes.data_as(C.c_void_p), dtype.encode(), nmemb) == 0, L.big_file_get_error_message()
    L.big_block_close(C.byref(bb))
    L.big_file_close(C.byref(bf))
    return out


def test_files_of_the_reference_library_are_read(snap, tmp_path):
    L = ref_lib()
    path = str(tmp_path / "REF")
    p = particles(5003, 3)
    ref_write_block(L, path, "1/Position", p["Position"], "f8", 4)
    ref_write_block(L, path, "1/Velocity", p["Velocity"], "f4", 2)
    ref_write_block(L, path, "1/ID", p["ID"], "u8", 1)
    ref_set_attr(L, path, "Header", "TotNumPart", np.array([0, 5003, 0, 0, 0, 0], np.uint64), "u8")
    ref_set_attr(L, path, "Header", "MassTable", np.array([0, 0.25, 0, 0, 0, 0]), "f8")
    ref_set_attr(L, path, "Header", "BoxSize", np.array([25000.0]), "f8")
    ref_set_attr(L, path, "Header", "Time", np.array([0.05]), "f8")
    h, parts = snap.read_snapshot(path)
    assert h["TotNumPart"][1] == 5003 and h["BoxSize"] == 25000.0 and h["Time"] == 0.05
    assert np.array_equal(parts[1]["Position"], p["Position"]) and np.array_equal(parts[1]["ID"], p["ID"])
    assert np.array_equal(parts[1]["Velocity"], p["Velocity"].astype(np.float32).astype(np.float64))
    assert np.all(parts[1]["Mass"] == np.float32(0.25))
    assert snap.block_info(path, "1/Position") == dict(dtype="<f8", nmemb=3, nfile=4, size=5003)
    # ... and the two writers produce the same bytes
    mine = str(tmp_path / "MINE")
    snap.write_block(mine, "1/Position", p["Position"], "f8", nfile=4)
    snap.write_block(mine, "1/Velocity", p["Velocity"], "f4", nfile=2)
    for blk, nf in (("1/Position", 4), ("1/Velocity", 2)):
        assert open(os.path.join(mine, blk, "header")).read() == open(os.path.join(path, blk, "header")).read()
        for f in range(nf):
            assert open(os.path.join(mine, blk, "%06X" % f), "rb").read() == open(os.path.join(path, blk, "%06X" % f), "rb").read()
    snap.set_attr(mine, "Header", "TotNumPart", np.array([0, 5003, 0, 0, 0, 0], np.uint64), "u8")
    snap.set_attr(mine, "Header", "MassTable", np.array([0, 0.25, 0, 0, 0, 0]), "f8")
    snap.set_attr(mine, "Header", "BoxSize", 25000.0, "f8")
    snap.set_attr(mine, "Header", "Time", 0.05, "f8")
    assert open(os.path.join(mine, "Header", "attr-v2")).read() == open(os.path.join(path, "Header", "attr-v2")).read()


def test_files_written_here_are_read_by_the_reference_library(snap, tmp_path):
    L = ref_lib()
    path = str(tmp_path / "PART_001")
    p = particles(4097, 4)
    snap.write_snapshot(path, {1: p}, box=8000.0, time=0.25, nfile=5)
    assert np.array_equal(ref_read_block(L, path, "1/Position", 4097, 3, np.float64, "f8"), p["Position"])
    assert np.array_equal(ref_read_block(L, path, "1/Velocity", 4097, 3, np.float32, "f4"), p["Velocity"].astype(np.float32))
    assert np.array_equal(ref_read_block(L, path, "1/ID", 4097, 1, np.uint64, "u8")[:, 0], p["ID"])
    assert np.array_equal(ref_read_block(L, path, "1/Mass", 4097, 1, np.float64, "f8")[:, 0], p["Mass"].astype(np.float64))   # cast on read
    assert ref_get_attr(L, path, "Header", "TotNumPart", np.uint64, "u8", 6).tolist() == [0, 4097, 0, 0, 0, 0]
    assert ref_get_attr(L, path, "Header", "BoxSize", np.float64, "f8", 1)[0] == 8000.0
    assert ref_get_attr(L, path, "Header", "Time", np.float64, "f8", 1)[0] == 0.25


def test_zero_length_attribute_does_not_break_the_block(snap, tmp_path):
    """bigfile writes "name dtype 0  #HUMANE [  ]" for an attribute with no members (bigfile.c:1615): the "#HUMANE" token is a
    comment, not the hex field - the other attributes of the block must stay readable and writable."""
    d = tmp_path / "S" / "Header"
    d.mkdir(parents=True)
    (d / "header").write_text("DTYPE: <i8\nNMEMB: 1\nNFILE: 0\n")
    (d / "attr-v2").write_text("Empty <S1 0  #HUMANE [  ]\nTime <f8 1 000000000000D03F #HUMANE [ 0.25 ]\n")
    path = str(tmp_path / "S")
    assert snap.get_attr(path, "Header", "Time", "f8")[0] == 0.25
    snap.set_attr(path, "Header", "BoxSize", 8.0, "f8")
    assert snap.get_attr(path, "Header", "BoxSize", "f8")[0] == 8.0 and snap.get_attr(path, "Header", "Time", "f8")[0] == 0.25
