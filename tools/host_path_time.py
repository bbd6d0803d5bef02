"""Phases of the drop-in host path at 256^3 (GPU box): MPG_HOST_TIMING=1 python tools/host_path_time.py [n]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mp-gadget_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pos, mass, box = pkg.ics.s_zel(n)
eng = pkg.Engine(0)
eng.gravshort_fill_ntab(0, 1.5)
eng.gravpm_init_periodic(box, 1.5, 2 * n, 43.0071)
eng.set_gravshort_treepar(TreeUseBH=0)
eng.gravshort_set_softenings(box / n)
P = pkg.make_particles(pos, mass)
for overlap in (2, True, False):          # 2: overlap + mpg_host_prefetch 40 ms ahead (round 6)
    eng.set_host_overlap(bool(overlap))
    for it in range(5):
        eng.set_particle_epoch(10 * overlap + it + 1)
        if overlap == 2:
            eng.host_prefetch(P, box)
            time.sleep(0.04)
        t0 = time.perf_counter()
        eng.gravpm_force(P)
        t1 = time.perf_counter()
        eng.force_tree_full(P, box)
        t2 = time.perf_counter()
        eng.grav_short_tree(P)
        t3 = time.perf_counter()
        print("overlap %d step %d: gravpm %.1f tree %.1f walk %.1f total %.1f ms" % (overlap, it, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t3 - t0)), flush=True)
eng.close()
