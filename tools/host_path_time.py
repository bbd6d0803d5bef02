"""PCIe-inclusive timing of the drop-in (host pointer) gravity path: struct particle_data in host memory in, results written back."""
import importlib, sys, time
sys.path.insert(0, '.')
import numpy as np
pkg = importlib.import_module("mp-gadget_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
G = 43.0071
pos, mass, box = pkg.ics.s_grid(n)
N = len(pos)
P = pkg.make_particles(pos, mass)
eng = pkg.Engine(0)
eng.gravshort_fill_ntab(0, 1.5)
eng.gravpm_init_periodic(box, 1.5, 2 * n, G)
eng.set_gravshort_treepar(TreeUseBH=2)
eng.gravshort_set_softenings(box / n)
use_epoch = len(sys.argv) > 2 and sys.argv[2] == "epoch"
for it in range(5):
    if use_epoch:
        eng.set_particle_epoch(it + 1)       # P is unchanged between the three calls of a step
    t0 = time.perf_counter()
    eng.gravpm_force(P)
    t1 = time.perf_counter()
    eng.force_tree_full(P, box)
    t2 = time.perf_counter()
    eng.grav_short_tree(P)
    t3 = time.perf_counter()
    print("step %d: gravpm_force %.1f ms  force_tree_full %.1f ms  grav_short_tree %.1f ms  total %.1f ms -> %.3g particles/s"
          % (it, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t3 - t0), N / (t3 - t0)), flush=True)
eng.close()
