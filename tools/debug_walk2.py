import importlib, sys, os
sys.path.insert(0, '.')
import numpy as np
pkg = importlib.import_module("mp-gadget_amd")
from oracle import oracle as O
G = 43.0071
n, nmesh = 32, 64
ic = sys.argv[1] if len(sys.argv) > 1 else "s_grid"
pos, mass, box = (pkg.ics.s_clust(n, box=8.0, seed=1) if ic == "s_clust" else pkg.ics.s_grid(n))
orc = O.Oracle(); orc.fill_ntab(0, 1.5)
tr = orc.tree(pos, mass, box)
par = O.make_grav_params(box, nmesh, npart_cbrt=n, G=G); par.TreeUseBH = 0
old = np.full(len(pos), 1e-6)
ao, po, co, _ = tr.grav_short_tree(par, oldacc=old, want_pot=True)
eng = pkg.Engine(0)
eng.gravshort_fill_ntab(0, 1.5); eng.gravpm_init_periodic(box, 1.5, nmesh, G)
eng.set_gravshort_treepar(TreeUseBH=0); eng.gravshort_set_softenings(box / n)
P = pkg.make_particles(pos, mass)
P["FullTreeGravAccel"][:, 0] = 1e-6 * G
eng.force_tree_full(P, box)
for variant in (1, 4, 5):
    eng.set_walk_variant(variant)
    store = np.zeros((len(pos), 3))
    P["FullTreeGravAccel"] = 0; P["FullTreeGravAccel"][:, 0] = 1e-6 * G
    eng.set_instrumentation(False, True)
    eng.grav_short_tree(P, AccelStore=store)
    c = eng.walk_counters()
    rel = np.sqrt(((store - ao) ** 2).sum(1)) / np.sqrt((ao ** 2).sum(1))
    w = np.argsort(rel)[-5:]
    print(ic, "variant", variant, "fastwrap env", os.environ.get("MPG_NO_FASTWRAP"), "counters", c["pp"], c["nodes_visited"], c["nodes_used"], "oracle", co,
          "median %.2e max %.2e" % (np.median(rel), rel.max()))
    print("   worst:", w, rel[w], "pos/box", pos[w] / box)
    print("   pot maxrel", np.abs((P["Potential"] - po) / po).max())
