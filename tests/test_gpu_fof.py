"""GPU parity of the friends-of-friends finder (csrc/fof.hip, mpg_dev_fof_fof) with the CPU restatement: group membership, numbering and
integer properties exactly, summed properties to rounding; the reference's own known answer (test_fof.c: one group)."""
import numpy as np
import pytest

from oracle import fof_oracle as F
from test_oracle_fof import kat_particles

pytestmark = pytest.mark.gpu


def dev(torch, a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run_engine(engine, pos, mass, ids, box, LL, minlen, vel=None, typ=None, hsml=None, flags=None):
    import torch
    n = len(pos)
    d = dict(pos=dev(torch, pos), mass=dev(torch, mass.astype(np.float32)), ids=dev(torch, ids.view(np.int64)), vel=dev(torch, vel),
             typ=dev(torch, typ), hsml=dev(torch, hsml), flags=dev(torch, flags))
    engine.dev_bind_particles(d["pos"], d["mass"], box, type=d["typ"])
    grnr = torch.zeros(n, dtype=torch.int64, device="cuda")
    ng = engine.dev_fof_fof(d["ids"], LL, minlen, vel=d["vel"], hsml=d["hsml"], flags=d["flags"], grnr=grnr)
    G = {k: v.cpu().numpy() for k, v in engine.dev_fof_groups(ng).items()}
    engine.synchronize()
    G["MinID"] = G["MinID"].view(np.uint64)
    return grnr.cpu().numpy(), G


def compare(grnr, G, grnr_o, Go, box):
    assert np.array_equal(grnr, grnr_o)
    for k in ("MinID", "Length", "GrNr", "LenType", "FirstPos"):
        assert np.array_equal(G[k], Go[k]), k
    for k in ("Mass", "MassType", "Vel"):
        assert np.allclose(G[k], Go[k], rtol=1e-12, atol=1e-12 * (np.abs(Go[k]).max() + 1e-300)), k
    d = np.abs(G["CM"] - Go["CM"])
    assert np.minimum(d, box - d).max() <= 1e-11 * box
    for k in ("Jmom", "Imom"):
        assert np.allclose(G[k], Go[k], rtol=1e-9, atol=1e-9 * (np.abs(Go[k]).max() + 1e-300)), k


def test_reference_known_answer_on_gpu(engine):
    pos, ids, box = kat_particles()
    N = len(pos)
    LL = 0.2 * box / np.cbrt(N)
    grnr, G = run_engine(engine, pos, np.ones(N), ids, box, LL, 5)
    assert len(G["MinID"]) == 1 and G["Length"][0] == N and G["MinID"][0] == 0 and np.all(grnr == 1)   # test_fof.c:93


def test_hand_made_cases(engine, orc):
    box, LL = 100.0, 1.0
    pos = np.array([[99.7, 50, 50], [0.4, 50, 50], [1.2, 50.3, 50], [2.0, 50.3, 50.5], [30, 30, 30], [30.5, 30.5, 30.5], [31.0, 31.0, 31.0],
                    [60.0, 60, 60], [61.5, 60, 60]])
    ids = np.array([40, 41, 42, 43, 7, 8, 9, 100, 101], np.uint64)
    mass = np.arange(1, 10, dtype=np.float32)
    vel = np.arange(27, dtype=np.float64).reshape(9, 3)
    compare(*run_engine(engine, pos, mass, ids, box, LL, 3, vel=vel), *F.fof_fof(orc, pos, mass, ids, box, LL, 3, vel=vel), box)
    dm = np.array([[10, 10, 10], [10.8, 10, 10], [50, 50, 50]])
    gas = np.array([[10.3, 10.2, 10], [52.5, 50, 50], [58.0, 50, 50], [50, 50, 56.3]])
    pos = np.vstack([dm, gas])
    typ = np.array([1, 1, 1, 0, 0, 0, 0], np.uint8)
    ids = np.array([5, 6, 9, 20, 21, 22, 23], np.uint64)
    mass = np.ones(7, np.float32)
    for hs, fl in ((None, None), (np.array([0, 0, 0, 0.1, 0.1, 17.0, 0.1]), None), (None, np.array([0, 1, 0, 0, 0, 0, 0], np.uint8))):
        g, G = run_engine(engine, pos, mass, ids, box, LL, 1, typ=typ, hsml=hs, flags=fl)
        go, Go = F.fof_fof(orc, pos, mass, ids, box, LL, 1, type=typ, hsml=hs, flags=fl)
        compare(g, G, go, Go, box)


def clumpy_set(seed, nclump=80, nback=6000, box=100.0):
    """Gaussian clumps of 5 .. 400 members (some across the periodic boundary) on a uniform background."""
    rng = np.random.RandomState(seed)
    N0 = nback + 40 * nclump
    LL = 0.2 * box / np.cbrt(N0)
    parts = [rng.random_sample((nback, 3)) * box]
    for _ in range(nclump):
        m = int(np.exp(rng.uniform(np.log(5), np.log(400))))
        c = rng.random_sample(3) * box
        if rng.random_sample() < 0.2:
            c[rng.randint(3)] = rng.choice([0.02, box - 0.02])
        parts.append(np.mod(c + rng.standard_normal((m, 3)) * 0.25 * LL * np.cbrt(m), box))
    pos = np.vstack(parts)
    return pos, box, LL


@pytest.mark.parametrize("seed,gas", [(1, False), (2, True)])
def test_clumpy_set_matches_oracle(engine, orc, seed, gas):
    pos, box, LL = clumpy_set(seed)
    N = len(pos)
    rng = np.random.RandomState(seed + 100)
    ids = rng.permutation(N).astype(np.uint64) + 1000
    vel = rng.standard_normal((N, 3)) * 30.0
    mass = np.ones(N, np.float32)
    typ = hsml = flags = None
    if gas:
        typ = np.where(rng.random_sample(N) < 0.4, 0, 1).astype(np.uint8)
        typ[rng.choice(N, 20, replace=False)] = 4
        typ[rng.choice(N, 10, replace=False)] = 2                                     # neither primary nor secondary: stays alone
        hsml = rng.random_sample(N) * 12 * LL
        mass = np.where(typ == 0, 0.19, 0.81).astype(np.float32)
        flags = (rng.random_sample(N) < 0.01).astype(np.uint8)
    minlen = 8
    g, G = run_engine(engine, pos, mass, ids, box, LL, minlen, vel=vel, typ=typ, hsml=hsml, flags=flags)
    go, Go = F.fof_fof(orc, pos, mass, ids, box, LL, minlen, vel=vel, type=typ, hsml=hsml, flags=flags)
    assert len(Go["MinID"]) >= 40 and Go["Length"].max() >= 100                        # the set really has groups
    if gas:
        assert (Go["LenType"][:, 0] > 0).sum() >= 20 and Go["LenType"][:, 4].sum() >= 1
    compare(g, G, go, Go, box)


def test_fof_edge_cases(engine, orc):
    import torch
    box, LL = 10.0, 0.5
    # no particle of a primary type: everybody stays alone
    pos = np.array([[1.0, 1, 1], [1.1, 1, 1], [1.2, 1, 1]])
    typ = np.zeros(3, np.uint8)
    ids = np.array([3, 4, 5], np.uint64)
    g, G = run_engine(engine, pos, np.ones(3), ids, box, LL, 2, typ=typ)
    assert len(G["MinID"]) == 0 and np.all(g == -1)
    g, G = run_engine(engine, pos, np.ones(3), ids, box, LL, 1, typ=typ)
    assert G["MinID"].tolist() == [3, 4, 5] and G["Length"].tolist() == [1, 1, 1] and sorted(g.tolist()) == [1, 2, 3]
    # one particle; identical positions (a full leaf of coincident points links into one group)
    g, G = run_engine(engine, pos[:1], np.ones(1), ids[:1], box, LL, 1)
    assert G["MinID"].tolist() == [3] and g.tolist() == [1]
    same = np.repeat(np.array([[2.0, 3.0, 4.0]]), 8, axis=0)
    g, G = run_engine(engine, same, np.ones(8), np.arange(8, dtype=np.uint64) + 10, box, LL, 1)
    assert G["Length"].tolist() == [8] and G["MinID"].tolist() == [10] and np.abs(G["CM"][0] - [2.0, 3.0, 4.0]).max() < 1e-6
    # no particles at all
    z = torch.zeros(0, 3, dtype=torch.float64, device="cuda")
    engine.dev_bind_particles(z, torch.zeros(0, dtype=torch.float32, device="cuda"), box)
    assert engine.dev_fof_fof(torch.zeros(0, dtype=torch.int64, device="cuda"), LL, 1) == 0


def test_fof_groups_spanning_ranks(tmp_path):
    """fof_fof with the particles on their Peano-Hilbert owners (mpg_dist_dev_fof_fof, csrc/dist.hip): clumps up to 3000 members wide
    enough to straddle domain boundaries, gas attached to the nearest dark matter.  1, 2 and 4 ranks (gloo, sharing this GPU) against
    the single-GPU finder: P[].GrNr of every particle and the number of groups EQUAL; MinID / Length / LenType / GrNr of every group
    equal; Mass, Vel, CM, Jmom, Imom to rounding (the parts of a group are added up in another order)."""
    import os
    import sys
    from conftest import run_ranks
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tools", "mgpu_fof_check.py")

    def run(name, nproc, mode, port):
        out = str(tmp_path / name)
        env = dict(os.environ, MPG_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MPG_MGPU_MODE=mode)
        cmd = [sys.executable, script, out] if nproc == 1 else \
              [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(port), script, out]
        run_ranks(cmd, env, out)
        return np.load(out + ".npz")

    one = run("one", 1, "single", 0)
    assert int(one["total"]) >= 60 and one["G_Length"].max() >= 1000
    box = 100.0
    for name, nproc, port in (("p1", 1, 0), ("p2", 2, 29611), ("p4", 4, 29612)):
        d = run(name, nproc, "peano", port)
        assert int(d["total"]) == int(one["total"]), name
        assert np.array_equal(d["grnr"], one["grnr"]), name
        for k in ("MinID", "Length", "GrNr", "LenType"):
            assert np.array_equal(d["G_" + k].astype(np.int64), one["G_" + k].astype(np.int64)), (name, k)
        for k in ("Mass", "MassType", "Vel"):
            assert np.allclose(d["G_" + k], one["G_" + k], rtol=1e-12, atol=1e-12 * np.abs(one["G_" + k]).max()), (name, k)
        dc = np.abs(d["G_CM"] - one["G_CM"])
        assert np.minimum(dc, box - dc).max() <= 1e-11 * box, name
        for k in ("Jmom", "Imom"):
            assert np.allclose(d["G_" + k], one["G_" + k], rtol=1e-8, atol=1e-8 * np.abs(one["G_" + k]).max()), (name, k)
        if nproc > 1:
            assert int(d["rounds"]) >= 2, name           # some group did cross a boundary: a second round was needed
