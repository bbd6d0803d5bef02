"""Accelerations / potentials / per-target walk cost / interaction counters of three force steps of the two-kernel walk, once per
leaf-expansion level (mpg_set_walk_leaf_expand 0, 1, 2, 4), written to an .npz (tests/test_gpu_gravity.py::
test_leaf_expansion_levels_agree).  usage: pair_check.py out.npz ic n"""
import ctypes as C
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mp-gadget_amd")
import torch
out, ic, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
pos, mass, box = getattr(pkg.ics, ic)(n)
N = len(pos)
dev = torch.device("cuda", 0)
res = {}
for kx in (0, 1, 2, 4):
    eng = pkg.Engine(0)   # a fresh engine per level: the list capacity adapts from the same start
    eng.set_walk_variant(6)
    eng.set_walk_leaf_expand(kx)
    eng.set_instrumentation(False, True)
    eng.gravshort_fill_ntab(0, 1.5)
    eng.gravpm_init_periodic(box, 1.5, 2 * n, 43.0071)
    eng.set_gravshort_treepar(ErrTolForceAcc=0.002, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=2, Rcut=6.0, FractionalGravitySoftening=1. / 30.)
    eng.gravshort_set_softenings(box / n)
    p = torch.from_numpy(pos).to(dev)
    m = torch.from_numpy(mass).to(dev)
    z3 = lambda: torch.zeros(N, 3, dtype=torch.float64, device=dev)
    gravpm, acc, prev, pot = z3(), z3(), z3(), torch.zeros(N, dtype=torch.float64, device=dev)
    cost = torch.zeros(N, dtype=torch.float32, device=dev)
    eng.dev_bind_particles(p, m, box)
    eng.lib.mpg_dev_set_walk_cost.argtypes = [C.c_void_p, C.c_void_p]
    eng._ck(eng.lib.mpg_dev_set_walk_cost(eng.h, C.c_void_p(cost.data_ptr())))
    for step in range(3):          # Barnes-Hut first walk, list-capacity adaptation, relative criterion
        pot.zero_()
        eng.dev_gravpm_force(gravpm, pot)
        eng.dev_force_tree_build()
        prev, acc = acc, prev
        eng.dev_grav_short_tree(acc, prev_accel=prev, gravpm=gravpm, potential=pot)
        torch.cuda.synchronize()
        c = eng.walk_counters()
        res["acc%d_k%d" % (step, kx)] = acc.cpu().numpy().copy()
        res["pot%d_k%d" % (step, kx)] = pot.cpu().numpy().copy()
        res["cost%d_k%d" % (step, kx)] = cost.cpu().numpy().copy()
        res["cnt%d_k%d" % (step, kx)] = np.array([c["pp"], c["nodes_visited"], c["nodes_used"]], dtype=np.int64)
        print(ic, n, "kx", kx, "step", step, "walk", eng.walk_choice(), c["pp"], c["nodes_visited"], c["nodes_used"], flush=True)
    eng.close()
np.savez(out, **res)
