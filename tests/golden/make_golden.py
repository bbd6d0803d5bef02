#!/usr/bin/env python3
"""Generate the committed golden vectors for the gravity path.

The reference's own TreePM path cannot be built in this image (needs PFFT + GSL headers), so the vectors are
produced by the CPU oracle AFTER it has been pinned to the reference's known answers (tests/test_oracle_kat.py:
SURVEY App. C.5 probe outputs of the unmodified reference objects, and libgadget/tests/test_gravity.c bounds).
The first fixture carries those probe numbers themselves.  Run from the repo root:  python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mp-gadget_amd")
from oracle import oracle as O  # noqa: E402

G = 43.0071
HERE = os.path.dirname(os.path.abspath(__file__))


def run(pos, mass, box, n, nmesh):
    orc = O.Oracle()
    orc.fill_ntab(0, 1.5)
    gpm, pmpot = O.gravpm_force(pos, mass, box, nmesh, 1.5, G)
    tr = orc.tree(pos, mass, box)
    par = O.make_grav_params(box, nmesh, npart_cbrt=n, G=G)
    par.TreeUseBH = 1
    a1, _, c1, _ = tr.grav_short_tree(par, oldacc=np.sqrt((gpm ** 2).sum(1)) / G)
    par.TreeUseBH = 0
    a2, p2, c2, _ = tr.grav_short_tree(par, oldacc=np.sqrt(((a1 + gpm) ** 2).sum(1)) / G, want_pot=True)
    return dict(GravPM=gpm, PMPotential=pmpot, Accel1=a1, Accel2=a2, TreePotential=p2, counters1=c1, counters2=c2,
                numnodes=np.int64(tr.numnodes))


if __name__ == "__main__":
    # reference known answers recorded by the survey probe
    np.savez(os.path.join(HERE, "reference_probe_kat.npz"),
             n=np.array([32, 64, 64]), nmesh=np.array([64, 128, 192]),
             mean_abs_accel=np.array([1.67498e-05, 1.67329e-05, 1.43708e-05]),
             ninteractions_per_particle=np.array([1333.8, 1333.9, 512.0]))
    pos, mass, box = pkg.ics.s_grid(16)
    np.savez_compressed(os.path.join(HERE, "grav_sgrid16.npz"), box=box, n=16, nmesh=32, **run(pos, mass, box, 16, 32))
    pos, mass, box = pkg.ics.s_clust(12, box=8.0, seed=5)
    np.savez_compressed(os.path.join(HERE, "grav_sclust12.npz"), box=box, n=12, nmesh=24, seed=5, **run(pos, mass, box, 12, 24))
    print("golden vectors written to", HERE)
