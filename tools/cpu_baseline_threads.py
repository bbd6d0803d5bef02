"""cpu_baseline experiments on the GPU box's host: python tools/cpu_baseline_threads.py n log2(sample) [groups like 1x8, 8x16]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib, numpy as np
import bench
if __name__ == "__main__":
    pkg = importlib.import_module("mp-gadget_amd")
    n = int(sys.argv[1])
    pos, mass, box = pkg.ics.s_zel(n)
    if len(sys.argv) > 3:
        p, t = (int(x) for x in sys.argv[3].split("x"))
        g = bench._numa_core_groups()
        flat = [c for grp in g for c in grp]
        bench._numa_core_groups = lambda nproc_hint=0: [flat[i * t:(i + 1) * t] for i in range(p)]
    r = bench.cpu_baseline(pkg, pos, mass, box, n, 2 * n, np.full((len(pos), 3), 1e-6), 1 << int(sys.argv[2]))
    print(n, sys.argv[3:] , {k: r[k] for k in ('value', 'pairs_per_s_per_thread', 'walk_s_median_of_3', 'tree_build_all_particles_s', 'cores', 'processes')}, flush=True)
