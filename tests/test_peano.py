"""The Peano-Hilbert state machine generated for the device (mp-gadget_amd/csrc/peano_tables.h, made by tools/gen_peano_tables.py)
against golden vectors taken from the reference function (tests/golden/peano_keys.npz, make_peano_golden.py) and, when
oracle/_ref is present, against the reference function itself on fresh random input.  CPU only: the tables are parsed from the header."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_tables():
    txt = open(os.path.join(ROOT, "mp-gadget_amd", "csrc", "peano_tables.h")).read()
    out = {}
    for name in ("MPG_PEANO_SUBPIX", "MPG_PEANO_NEXT"):
        body = txt.split("#define " + name)[1].split("}\n")[0]
        rows = re.findall(r"\{([0-9, ]+)\}", body)
        out[name] = np.array([[int(v) for v in r.split(",")] for r in rows], np.int64)
    n = int(re.search(r"MPG_PEANO_NSTATES (\d+)", txt).group(1))
    assert out["MPG_PEANO_SUBPIX"].shape == (n, 8) and out["MPG_PEANO_NEXT"].shape == (n, 8)
    return out["MPG_PEANO_SUBPIX"], out["MPG_PEANO_NEXT"]


def keys_from_tables(xyz, bits=21):
    sub, nxt = load_tables()
    xyz = np.asarray(xyz, np.int64)
    state = np.zeros(len(xyz), np.int64)
    key = np.zeros(len(xyz), np.uint64)
    for bit in range(bits - 1, -1, -1):
        pix = 4 * ((xyz[:, 0] >> bit) & 1) + 2 * ((xyz[:, 1] >> bit) & 1) + ((xyz[:, 2] >> bit) & 1)
        key = (key << np.uint64(3)) | sub[state, pix].astype(np.uint64)
        state = nxt[state, pix]
    return key


def test_tables_are_a_bijection_per_state():
    sub, nxt = load_tables()
    assert all(sorted(r) == list(range(8)) for r in sub.tolist())       # every orientation numbers its 8 octants 0..7
    assert nxt.min() >= 0 and nxt.max() < len(sub)


def test_golden_keys():
    g = np.load(os.path.join(ROOT, "tests", "golden", "peano_keys.npz"))
    assert np.array_equal(keys_from_tables(g["xyz"]), g["keys"])
    box = float(g["box"])
    fac = 1.0 / (box * 1.001) * float(1 << 21)
    ip = ((g["pos"] + box / 2000) * fac).astype(np.int32)
    assert np.array_equal(keys_from_tables(ip), g["pkeys"])
    # the curve is continuous: consecutive keys of a small cube are face neighbours
    side = 8
    c = np.array([[x, y, z] for x in range(side) for y in range(side) for z in range(side)])
    k = keys_from_tables(c, bits=3)
    assert sorted(k.tolist()) == list(range(side ** 3))
    path = c[np.argsort(k)]
    assert np.all(np.abs(np.diff(path, axis=0)).sum(1) == 1)


def test_against_reference_function_when_built():
    path = os.path.join(ROOT, "oracle", "_ref", "libref_leaf.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built (reference tree absent)")
    lib = C.CDLL(path, mode=os.RTLD_LAZY)
    lib.peano_hilbert_key.restype = C.c_uint64
    lib.peano_hilbert_key.argtypes = [C.c_int] * 4
    rng = np.random.RandomState(5)
    for bits in (1, 2, 5, 13, 21):
        xyz = rng.randint(0, 1 << bits, size=(3000, 3))
        ref = np.array([lib.peano_hilbert_key(int(a), int(b), int(c), bits) for a, b, c in xyz], np.uint64)
        assert np.array_equal(keys_from_tables(xyz, bits), ref), bits
