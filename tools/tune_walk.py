"""Walk-kernel tuning on the GPU box: timing + burst statistics per variant / threshold."""
import importlib, sys, time, os
sys.path.insert(0, '.')
import numpy as np
pkg = importlib.import_module("mp-gadget_amd")
import torch
G = 43.0071
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ic = sys.argv[2] if len(sys.argv) > 2 else "s_grid"
nmesh = 2 * n
pos, mass, box = getattr(pkg.ics, ic)(n)
N = len(pos)
eng = pkg.Engine(0)
eng.gravshort_fill_ntab(0, 1.5)
eng.gravpm_init_periodic(box, 1.5, nmesh, G)
eng.set_gravshort_treepar(TreeUseBH=0)
eng.gravshort_set_softenings(box / n)
dpos = torch.from_numpy(pos).cuda(); dmass = torch.from_numpy(mass).cuda()
eng.dev_bind_particles(dpos, dmass, box)
gpm = torch.zeros(N, 3, dtype=torch.float64, device="cuda"); acc = torch.zeros_like(gpm); pot = torch.zeros(N, dtype=torch.float64, device="cuda")
eng.dev_gravpm_force(gpm, pot)
eng.dev_force_tree_build()
if 'MPG_SPLIT_MODE' in os.environ:
    ov, cpw = os.environ['MPG_SPLIT_MODE'].split(',')
    eng.set_walk_split_mode(int(ov), int(cpw))
eng.set_walk_variant(1); eng.set_walk_threshold(16)
eng.dev_grav_short_tree(acc, prev_accel=torch.zeros_like(acc), gravpm=gpm, potential=pot)   # BH-free first pass gives OldAcc
prev = acc.clone()
ref = None
VARIANTS = [int(v) for v in os.environ.get('MPG_VARIANTS', '1,4,6').split(',')]
for variant, thr in [(v, 16 if v == 1 else 512) for v in VARIANTS]:
    eng.set_walk_variant(variant); eng.set_walk_threshold(16); eng.set_walk_list_capacity(thr)
    eng.set_instrumentation(False, True)
    eng.dev_grav_short_tree(acc, prev_accel=prev, gravpm=gpm, potential=pot)
    eng.synchronize()
    c = eng.walk_counters()
    eng.set_instrumentation(False, False)
    for _ in range(3):   # let adaptive list capacities settle
        eng.dev_grav_short_tree(acc, prev_accel=prev, gravpm=gpm, potential=pot)
    eng.synchronize()
    eng.walk_events_collect()
    for _ in range(3):
        eng.dev_grav_short_tree(acc, prev_accel=prev, gravpm=gpm, potential=pot)
    ms, k = eng.walk_events_collect()
    a = acc.cpu().numpy()
    if ref is None:
        ref = a.copy()
    d = np.sqrt(((a - ref) ** 2).sum(1)) / np.sqrt((ref ** 2).sum(1))
    s = ""
    if variant in (4, 5) and c["node_steps"] and c["int_steps"]:
        s = "A: %.1f steps/target, %.2f nodes/step; B: %.1f steps/target, lane util %.2f" % (c["node_steps"] / N, c["node_lanes"] / c["node_steps"], c["int_steps"] / 8.0 / N, c["int_lanes"] / c["int_steps"])
        s += "  cycles/step A %.0f B %.0f" % (c["cycles_a"] / max(c["node_steps"], 1) * 8, c["cycles_b"] / max(c["int_steps"] / 8, 1) * 8)
        s += "  tree ms %s" % {k: round(v, 2) for k, v in eng.phase_times().items() if k.startswith("tree")}
    if variant == 6:
        print("   kernel-6 state (variant, list capacity, targets handed to the fallback):", eng.walk_choice(), "of", N, flush=True)
    print("n=%d %s variant %d thr %2d: %.2f ms  pp/N %.1f nodes/N %.1f used/N %.2f  maxrel vs v1 %.1e  %s" % (n, ic, variant, thr, ms / k, c["pp"] / N, c["nodes_visited"] / N, c["nodes_used"] / N, d.max(), s), flush=True)
