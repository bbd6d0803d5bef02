"""Generates tests/golden/peano_keys.npz from the reference function itself: peano_hilbert_key of libgadget/utils/peano.c,
compiled in place into oracle/_ref/libref_leaf.so (oracle/Makefile).  Run in the build container: python tests/golden/make_peano_golden.py
Contents: integer triplets (x, y, z) in [0, 2^21) with their 21-bit keys, and positions in a box of 25000 with their PEANO() keys
(peano.h:15-21: key of int((Pos + Box/2000) / (1.001 Box) 2^21) per axis)."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(HERE, "..", "..", "oracle", "_ref", "libref_leaf.so"), mode=os.RTLD_LAZY)
lib.peano_hilbert_key.restype = C.c_uint64
lib.peano_hilbert_key.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]

rng = np.random.RandomState(21)
n = 4096
xyz = rng.randint(0, 1 << 21, size=(n, 3)).astype(np.int32)
xyz[:8] = [[0, 0, 0], [(1 << 21) - 1] * 3, [1, 0, 0], [0, 1, 0], [0, 0, 1], [1 << 20, 0, 0], [0, 1 << 20, 1 << 20], [12345, 54321, 2]]
keys = np.array([lib.peano_hilbert_key(int(a), int(b), int(c), 21) for a, b, c in xyz], np.uint64)
box = 25000.0
pos = rng.random_sample((n, 3)) * box
pos[:4] = [[0, 0, 0], [box, box, box], [box / 2, box / 2, box / 2], [1e-9, box - 1e-9, 0.5]]
fac = 1.0 / (box * 1.001) * float(1 << 21)
ip = ((pos + box / 2000) * fac).astype(np.int32)          # C conversion double -> int: truncation
pkeys = np.array([lib.peano_hilbert_key(int(a), int(b), int(c), 21) for a, b, c in ip], np.uint64)
np.savez_compressed(os.path.join(HERE, "peano_keys.npz"), xyz=xyz, keys=keys, box=box, pos=pos, pkeys=pkeys)
print("wrote peano_keys.npz", keys[:3], pkeys[:3])
