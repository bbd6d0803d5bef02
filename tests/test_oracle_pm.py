"""pm_oracle.c (the C + OpenMP form of the reference's particle <-> mesh loops and Fourier sweeps, petapm.c:955-1020,
gravpm.c:383-517) against oracle.py's numpy restatement of the same functions, which tests/test_oracle_kat.py pins with the
reference's test_gravity.c bounds.  CPU only."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def _set(n, box, seed, clump=False):
    rng = np.random.RandomState(seed)
    pos = rng.random_sample((n, 3)) * box
    if clump:
        pos[: n // 2] = (box * 0.999 + rng.normal(0, box * 0.01, (n // 2, 3))) % box     # a clump across the periodic corner
    pos[0] = (0.0, 0.0, 0.0)
    pos[1] = np.nextafter(box, 0) * np.ones(3)                                             # the last cell: its cloud wraps to 0
    mass = (rng.random_sample(n) + 0.5).astype(np.float32)
    return pos, mass


def _check(nmesh, clump, fast):
    box = 1000.0
    pos, mass = _set(3000, box, nmesh, clump)
    orc = O.Oracle(fast=fast)
    a0, p0 = O.gravpm_force(pos, mass, box, nmesh, 1.25, 43.0071)
    tm = {}
    a1, p1 = O.gravpm_force_c(orc, pos, mass, box, nmesh, 1.25, 43.0071, workers=2, timings=tm)
    sa, sp = np.abs(a0).mean(), np.abs(p0).mean()
    tol = 1e-9 if fast else 1e-12          # -ffast-math reorders the sums and takes reciprocals for the quotients
    assert np.abs(a1 - a0).max() / sa < tol
    assert np.abs(p1 - p0).max() / sp < tol
    assert set(tm) == {"deposit", "fft", "transfer", "readout"} and all(v >= 0 for v in tm.values())


@pytest.mark.parametrize("nmesh,clump", [(16, False), (32, True), (24, True)])
def test_c_pm_matches_the_numpy_restatement(nmesh, clump):
    _check(nmesh, clump, False)


def test_c_pm_of_the_reference_flags_build():
    """liboracle_fast.so (-O3 -ffast-math -fopenmp: what bench.py's cpu_baseline times).  In a process of its own: loading a -ffast-math
    library switches the loading thread to flush-to-zero arithmetic, which must not leak into the other tests of this process."""
    code = "import sys; sys.path.insert(0, %r); import tests.test_oracle_pm as t; [t._check(n, c, True) for n, c in ((16, False), (32, True), (24, True))]; print('fast ok')" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "fast ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_c_deposit_and_readout_piecewise():
    box, nmesh = 100.0, 20
    pos, mass = _set(5000, box, 3, True)
    orc = O.Oracle()
    L = O._pm_bind(orc)
    rho = np.zeros(nmesh ** 3)
    L.pmo_cic_deposit(len(pos), pos, mass.astype(np.float64), None, box, nmesh, rho)
    ref = O.pm_cic_deposit(pos, mass, box, nmesh).reshape(-1)
    assert np.abs(rho - ref).max() <= 1e-12 * ref.max()
    assert abs(rho.sum() - mass.astype(np.float64).sum()) < 1e-9 * mass.sum()              # the cloud weights sum to one
    # live flags: a dead record deposits nothing (INACTIVE / RegionInd < 0, petapm.c:964)
    live = np.ones(len(pos), np.uint8)
    live[::3] = 0
    rho2 = np.zeros(nmesh ** 3)
    L.pmo_cic_deposit(len(pos), pos, mass.astype(np.float64), live.ctypes.data_as(C.c_void_p), box, nmesh, rho2)
    ref2 = O.pm_cic_deposit(pos[live > 0], mass[live > 0], box, nmesh).reshape(-1)
    assert np.abs(rho2 - ref2).max() <= 1e-12 * ref2.max()
    # read-out accumulates (P[i].Potential += ..., gravpm.c:506) with a stride
    mesh = np.random.RandomState(5).random_sample(nmesh ** 3)
    out = np.ones((len(pos), 3))
    L.pmo_readout(len(pos), pos, box, nmesh, mesh, 2.0, C.c_void_p(out.ctypes.data + 8), 3)
    want = 1.0 + 2.0 * O.pm_readout(mesh.reshape((nmesh,) * 3), pos, box, nmesh)
    assert np.abs(out[:, 1] - want).max() < 1e-12 and (out[:, 0] == 1).all() and (out[:, 2] == 1).all()
