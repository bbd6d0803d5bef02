import importlib, sys, time, os
sys.path.insert(0, '.')
import numpy as np
order_first = os.environ.get("LOAD_ORDER", "lib")
if order_first == "torch":
    import torch
pkg = importlib.import_module("mp-gadget_amd")
eng = pkg.Engine(0)
print(eng.version())
import torch
print("torch", torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0))
from oracle import oracle as O
orc = O.Oracle(fast=False); orc.fill_ntab(0, 1.5)
eng.gravshort_fill_ntab(0, 1.5)
eng.set_instrumentation(True, True)
G = 43.0071
for n, nmesh in ((32, 64), (64, 128)):
    pos, mass, box = pkg.ics.s_grid(n)
    eng.gravpm_init_periodic(box, 1.5, nmesh, G)
    eng.set_gravshort_treepar(TreeUseBH=2)
    eng.gravshort_set_softenings(box / n)
    P = pkg.make_particles(pos, mass)
    eng.gravpm_force(P)
    gpm_o, pot_o = O.gravpm_force(pos, mass, box, nmesh, 1.5, G)
    d = np.abs(P["GravPM"] - gpm_o).max() / np.abs(gpm_o).mean()
    print(n, "PM max diff / mean|GravPM| =", d, "mean|gravpm|", np.abs(gpm_o).mean())
    P["GravPM"] = 0   # short-range-only KAT of SURVEY C.5
    eng.force_tree_full(P, box)
    st = eng.tree_stats()
    print("tree: npart", st.NumParticles, "nodes", st.numnodes, "leaves", st.numleaves, "maxlevel", st.maxlevel, "root mass", st.root_mass)
    eng.grav_short_tree(P)
    c1 = eng.walk_counters()
    eng.grav_short_tree(P)
    c2 = eng.walk_counters()
    print("counters walk2", c2, "Nint/N", c2["pp"]/len(pos), "times", eng.phase_times())
    print("mean|a| =", np.abs(P["FullTreeGravAccel"]).mean())
    # oracle
    tr = orc.tree(pos, mass, box)
    par = O.make_grav_params(box, nmesh, npart_cbrt=n); par.TreeUseBH = 1
    a1, _, k1, _ = tr.grav_short_tree(par, oldacc=np.zeros(len(pos)))
    par.TreeUseBH = 0
    a2, p2, k2, _ = tr.grav_short_tree(par, oldacc=np.sqrt((a1**2).sum(1))/G, want_pot=True)
    rel = np.sqrt(((P["FullTreeGravAccel"]-a2)**2).sum(1))/np.sqrt((a2**2).sum(1))
    print("oracle counters", k2, "rel diff median %.3e 99.9%% %.3e max %.3e" % (np.median(rel), np.quantile(rel, 0.999), rel.max()))
    print("pot rel diff max", np.abs((P["Potential"]-p2)/p2).max())
# device path timing at larger N
for n, nmesh in ((128, 256), (256, 512)):
    pos, mass, box = pkg.ics.s_grid(n)
    dpos = torch.from_numpy(pos).cuda(); dmass = torch.from_numpy(mass).cuda()
    N = len(pos)
    eng.gravpm_init_periodic(box, 1.5, nmesh, G)
    eng.set_gravshort_treepar(TreeUseBH=2)
    eng.gravshort_set_softenings(box / n)
    eng.dev_bind_particles(dpos, dmass, box)
    gpm = torch.zeros(N, 3, dtype=torch.float64, device="cuda"); acc = torch.zeros_like(gpm); pot = torch.zeros(N, dtype=torch.float64, device="cuda")
    for it in range(3):
        for thr in ((8,) if it < 2 else (1, 8, 16, 24, 32, 48)):
            eng.set_walk_threshold(thr)
            t0 = time.time()
            eng.dev_gravpm_force(gpm, pot)
            eng.dev_force_tree_build()
            prev = acc.clone()
            eng.dev_grav_short_tree(acc, prev_accel=prev, gravpm=gpm, potential=pot)
            eng.synchronize(); t1 = time.time()
            print(n, "iter", it, "thr", thr, "wall %.1f ms" % ((t1-t0)*1e3), {k: round(v, 2) for k, v in eng.phase_times().items()}, eng.walk_counters())
    print("mean|acc|", acc.abs().mean().item(), "mean|gpm|", gpm.abs().mean().item())
