"""Per-phase timing of the SPH path on the GPU box (2 x n^3 DM+gas, s_zel)."""
import importlib, sys, time
sys.path.insert(0, '.')
import numpy as np
pkg = importlib.import_module("mp-gadget_amd")
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
G = 43.0071
posd, _, box = pkg.ics.s_zel(n)
sp = box / n
posg = np.mod(posd - 0.425 * sp, box); posd = np.mod(posd + 0.075 * sp, box)
pos = np.concatenate([posg, posd]); N = len(pos)
mass = np.concatenate([np.full(n**3, 0.15, np.float32), np.full(n**3, 0.85, np.float32)])
typ = np.concatenate([np.zeros(n**3, np.uint8), np.ones(n**3, np.uint8)])
dev = "cuda"
f8 = torch.float64
d_pos, d_mass, d_type = torch.from_numpy(pos).to(dev), torch.from_numpy(mass).to(dev), torch.from_numpy(typ).to(dev)
eng = pkg.Engine(0)
eng.use_torch_stream()
eng.set_gravshort_treepar(); eng.gravshort_set_softenings(box / n)
eng.set_densitypar(1.0, 2.0, 2.0, 99999., 2, 0.006); eng.set_hydropar(0, 100.0, 0.75)
eng.dev_bind_particles(d_pos, d_mass, box, type=d_type)
z1 = lambda: torch.zeros(N, dtype=f8, device=dev); z3 = lambda: torch.zeros(N, 3, dtype=f8, device=dev)
a = dict(hsml=z1(), dthsml=z1(), vel=z3(), entropy=torch.ones(N, dtype=f8, device=dev), density=z1(), egywtdensity=z1(), dhsmlegyfac=z1(),
         divvel=z1(), curlvel=z1(), hydroacc_out=z3(), dtentropy_out=z1(), maxsignalvel=z1())
t = pkg.SphTimes(); t.atime, t.hubble = 0.1, 0.1
for i in range(47): t.dloga_bin[i] = 0.01
def timed(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    print("%-28s %8.2f ms" % (name, dt), flush=True)
timed("tree GAS+BH with moments", lambda: eng.dev_force_tree_rebuild_mask(33, with_moments=True))
timed("set_init_hsml", lambda: eng.dev_set_init_hsml(a, box / n))
for it in range(3):
    timed("tree GAS", lambda: eng.dev_force_tree_rebuild_mask(1))
    timed("density", lambda: eng.dev_density(a, t))
    print("    ", eng.sph_stats())
    timed("hmax", lambda: eng.dev_force_tree_calc_hmax())
    timed("hydro", lambda: eng.dev_hydro_force(a, t))
    print("    ", eng.sph_stats())
