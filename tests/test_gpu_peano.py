"""GPU: Peano-Hilbert keys (bit-identical to the reference's, tests/golden/peano_keys.npz) and the (type, key) order of
slots_gc_sorted (slotsmanager.c:404-452)."""
import os

import numpy as np
import pytest

from test_peano import keys_from_tables

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_peano_keys_bit_identical(engine):
    import torch
    g = np.load(os.path.join(ROOT, "tests", "golden", "peano_keys.npz"))
    box = float(g["box"])
    pos = torch.from_numpy(g["pos"]).cuda()
    keys = torch.zeros(len(g["pos"]), dtype=torch.int64, device="cuda")
    engine.dev_peano_keys(pos, box, keys)
    engine.synchronize()
    assert np.array_equal(keys.cpu().numpy().view(np.uint64), g["pkeys"])
    # a large random set against the table walk on the host (itself pinned to the reference by tests/test_peano.py)
    rng = np.random.RandomState(8)
    n, box = 1000003, 64000.0
    p = rng.random_sample((n, 3)) * box
    p[:3] = [[box, box, box], [0, 0, 0], [box, 0, box / 2]]
    keys = torch.zeros(n, dtype=torch.int64, device="cuda")
    engine.dev_peano_keys(torch.from_numpy(p).cuda(), box, keys)
    engine.synchronize()
    fac = 1.0 / (box * 1.001) * float(1 << 21)
    ref = keys_from_tables(((p + box / 2000) * fac).astype(np.int32))
    assert np.array_equal(keys.cpu().numpy().view(np.uint64), ref)


@pytest.mark.parametrize("n", [1, 1000, 300007])
def test_order_by_type_and_key(engine, n):
    import torch
    rng = np.random.RandomState(n)
    box = 100.0
    p = rng.random_sample((n, 3)) * box
    if n > 10:
        p[5] = p[3]                                                     # equal keys: input order is kept
    typ = rng.choice([0, 1, 4, 5], size=n, p=[0.3, 0.5, 0.15, 0.05]).astype(np.uint8)
    flags = (rng.random_sample(n) < 0.03).astype(np.uint8)              # IsGarbage
    d_p, d_t, d_f = torch.from_numpy(p).cuda(), torch.from_numpy(typ).cuda(), torch.from_numpy(flags).cuda()
    keys = torch.zeros(n, dtype=torch.int64, device="cuda")
    perm = torch.zeros(n, dtype=torch.int32, device="cuda")
    engine.dev_peano_keys(d_p, box, keys)
    live = engine.dev_order_by_type_and_key(keys, perm, type=d_t, flags=d_f)
    engine.synchronize()
    k = keys.cpu().numpy().view(np.uint64)
    tk = np.where(flags & 1, 255, typ).astype(np.int64)
    ref = np.lexsort((np.arange(n), k, tk))                              # (TypeKey, Key), stable
    assert live == int((flags == 0).sum())
    assert np.array_equal(perm.cpu().numpy(), ref.astype(np.int32))
    assert np.all(flags[perm.cpu().numpy()[:live]] == 0)
