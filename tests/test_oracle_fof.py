"""The FOF restatement (oracle/fof_oracle.c + oracle/fof_oracle.py) against the reference's own known answer
(libgadget/tests/test_fof.c:40-56, 80-97: 512 x 512 dark-matter particles on a wrapped diagonal, linking length 0.2 mean
separations, FOFHaloMinLength 5 -> exactly one group) and against hand-made cases."""
import numpy as np

from oracle import fof_oracle as F


def kat_particles(N=512 * 512, box=20000.0):
    ids = np.arange(N, dtype=np.uint64)
    pos = np.zeros((N, 3))
    for j in range(3):
        p = box * (j + 1) * ids.astype(np.float64) / N          # test_fof.c:52
        while np.any(p > box):
            p = np.where(p > box, p - box, p)
        pos[:, j] = p
    return pos, ids, box


def test_reference_known_answer(orc):
    pos, ids, box = kat_particles()
    N = len(pos)
    LL = 0.2 * box / np.cbrt(N)                                  # fof_init(BoxSize / cbrt(NumPart)), set_fof_testpar(1, 0.2, 5)
    grnr, G = F.fof_fof(orc, pos, np.ones(N, np.float32), ids, box, LL, 5)
    assert len(G["MinID"]) == 1                                  # assert_true(fof.TotNgroups == 1)
    assert G["Length"][0] == N and G["MinID"][0] == 0 and G["GrNr"][0] == 1 and np.all(grnr == 1)


def test_two_clumps_wrap_and_minimum_length(orc):
    box, LL = 100.0, 1.0
    a = np.array([[99.7, 50, 50], [0.4, 50, 50], [1.2, 50.3, 50], [2.0, 50.3, 50.5]])      # a chain across the periodic boundary
    b = np.array([[30, 30, 30], [30.5, 30.5, 30.5], [31.0, 31.0, 31.0]])                   # a second chain (steps of 0.87)
    c = np.array([[60.0, 60, 60], [61.5, 60, 60]])                                         # two particles 1.5 apart: not linked
    pos = np.vstack([a, b, c])
    ids = np.array([40, 41, 42, 43, 7, 8, 9, 100, 101], np.uint64)
    lab = F.fof_labels(orc, pos, ids, box, LL)
    assert lab.tolist() == [40, 40, 40, 40, 7, 7, 7, 100, 101]
    mass = np.arange(1, 10, dtype=np.float32)
    vel = np.arange(27, dtype=np.float64).reshape(9, 3)
    grnr, G = F.fof_fof(orc, pos, mass, ids, box, LL, 3, vel=vel)
    assert G["MinID"].tolist() == [7, 40] and G["Length"].tolist() == [3, 4] and G["GrNr"].tolist() == [2, 1]
    assert grnr.tolist() == [1, 1, 1, 1, 2, 2, 2, -1, -1]
    # centre of mass of the wrapped chain: computed in the frame of its first particle
    m = mass[:4].astype(np.float64)
    x = np.array([99.7, 100.4, 101.2, 102.0])
    assert abs(G["CM"][1][0] - np.mod((m * x).sum() / m.sum(), box)) < 1e-5          # (FirstPos is a float)
    assert abs(G["Mass"][1] - m.sum()) < 1e-12 and G["LenType"][1].tolist() == [0, 4, 0, 0, 0, 0]
    assert np.allclose(G["Vel"][0], (mass[4:7, None] * vel[4:7]).sum(0) / mass[4:7].sum())


def test_secondary_attachment(orc):
    box, LL = 100.0, 1.0
    dm = np.array([[10, 10, 10], [10.8, 10, 10], [50, 50, 50]])
    gas = np.array([[10.3, 10.2, 10], [52.5, 50, 50], [58.0, 50, 50], [50, 50, 56.3]])
    pos = np.vstack([dm, gas])
    typ = np.array([1, 1, 1, 0, 0, 0, 0], np.uint8)
    ids = np.array([5, 6, 9, 20, 21, 22, 23], np.uint64)
    lab = F.fof_labels(orc, pos, ids, box, LL, type=typ)
    # radii tried: 0.4, 0.8, 1.6, 3.2, 6.4 (float): 2.5 and 6.3 away attach, 8.0 away stays alone
    assert lab.tolist() == [5, 5, 9, 5, 9, 22, 9]
    # a large smoothing length widens the first radius (half of Hsml): the particle 8.0 away is found at once
    hs = np.array([0, 0, 0, 0.1, 0.1, 17.0, 0.1])
    assert F.fof_labels(orc, pos, ids, box, LL, type=typ, hsml=hs).tolist() == [5, 5, 9, 5, 9, 9, 9]
    # garbage takes no part
    fl = np.array([0, 1, 0, 0, 0, 0, 0], np.uint8)
    assert F.fof_labels(orc, pos, ids, box, LL, type=typ, flags=fl).tolist() == [5, 6, 9, 5, 9, 22, 9]
