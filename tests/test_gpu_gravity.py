"""GPU parity tests of the TreePM gravity path: HIP engine (through the C-ABI) vs the CPU oracle on the same inputs.

Tolerances: the engine takes the reference's opening decisions per target, so interaction sets are
identical and only the summation order differs: median |da|/|a| <= 1e-12, 99.9 % <= 1e-9, and no particle beyond
2 * ErrTolForceAcc * <|a|> (bound of one flipped opening decision).  PM: |dGravPM| / <|GravPM|> <= 1e-11.
Tree: identical node set, moments to 1e-13.  Counters (pair interactions, nodes visited) must be EQUAL.
"""
import os

import numpy as np
import pytest

from conftest import keep_artifacts_on_failure, phase_clock, run_ranks

from oracle import oracle as O

pytestmark = pytest.mark.gpu
G = 43.0071
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def setup_engine(eng, box, n, nmesh, TreeUseBH=2, Rcut=6.0, window=0, mean_sep=None):
    eng.gravshort_fill_ntab(window, 1.5)
    eng.gravpm_init_periodic(box, 1.5, nmesh, G)
    eng.set_gravshort_treepar(TreeUseBH=TreeUseBH, Rcut=Rcut)
    eng.gravshort_set_softenings(box / n if mean_sep is None else mean_sep)


def oracle_two_walks(orc, pos, mass, box, n, nmesh, gpm, want_pot=False):
    tr = orc.tree(pos, mass, box)
    par = O.make_grav_params(box, nmesh, npart_cbrt=n, G=G)
    par.TreeUseBH = 1
    a1, _, c1, _ = tr.grav_short_tree(par, oldacc=np.sqrt((gpm ** 2).sum(1)) / G)
    par.TreeUseBH = 0
    a2, p2, c2, _ = tr.grav_short_tree(par, oldacc=np.sqrt(((a1 + gpm) ** 2).sum(1)) / G, want_pot=want_pot)
    return a1, a2, p2, c1, c2, tr


def assert_accel_parity(a_hip, a_ref, errtol=0.002):
    mag = np.sqrt((a_ref ** 2).sum(1))
    d = np.sqrt(((a_hip - a_ref) ** 2).sum(1))
    rel = d / np.maximum(mag, 1e-300)
    assert np.median(rel) <= 1e-12, np.median(rel)
    assert np.quantile(rel, 0.999) <= 1e-9, np.quantile(rel, 0.999)
    assert d.max() <= 2 * errtol * np.abs(a_ref).mean(), d.max()


# ------------------------------------------------------------------------------- tree
@pytest.mark.parametrize("kind", ["grid", "clust", "tiny9", "tiny8", "one"])
def test_tree_topology_and_moments(pkg, engine, orc, kind):
    """Same node SET as forcetree.c produces (cell is internal iff > 8 particles), same geometry, moments to 1e-13,
    every particle in exactly one leaf (test_forcetree.c:119-171)."""
    if kind == "grid":
        pos, mass, box = pkg.ics.s_grid(20)
    elif kind == "clust":
        pos, mass, box = pkg.ics.s_clust(16, box=8.0, seed=11)
        mass = (1 + np.arange(len(pos)) % 3).astype(np.float32)
    else:
        k = {"tiny9": 9, "tiny8": 8, "one": 1}[kind]
        rng = np.random.RandomState(4)
        pos, mass, box = rng.random_sample((k, 3)) * 10.0, np.ones(k, np.float32), 10.0
    P = pkg.make_particles(pos, mass)
    engine.force_tree_full(P, box)
    t = engine.tree_export()
    ot = orc.tree(pos, mass, box)
    d = ot.export()
    live = d["live"].astype(bool)
    assert live.sum() == len(t["level"])
    # node identity = (level, centre): compare as sorted records
    key_o = np.round(np.c_[d["level"][live], d["center"][live] / box * 2 ** 40]).astype(np.int64)
    key_h = np.round(np.c_[t["level"], t["center"] / box * 2 ** 40]).astype(np.int64)
    io = np.lexsort(key_o.T[::-1])
    ih = np.lexsort(key_h.T[::-1])
    assert np.array_equal(key_o[io], key_h[ih])
    assert np.array_equal(d["center"][live][io], t["center"][ih])        # bit-identical geometry
    assert np.array_equal(d["len"][live][io], t["len"][ih])
    assert np.allclose(d["mass"][live][io], t["mass"][ih], rtol=1e-14, atol=0)
    assert np.abs(d["cofm"][live][io] - t["cofm"][ih]).max() <= 1e-13 * box
    # leaves: occupancy and ownership
    nocc = np.where(d["childtype"][live] == 0, d["noccupied"][live], 0)[io]
    assert np.array_equal(nocc, t["pcount"][ih])
    seen = np.zeros(len(pos), int)
    for j in np.nonzero(t["pcount"] > 0)[0]:
        idx = t["order"][t["pstart"][j]:t["pstart"][j] + t["pcount"][j]]
        seen[idx] += 1
        assert np.all(np.abs(pos[idx] - t["center"][j]) <= t["len"][j] / 2)
    assert np.all(seen == 1)
    st = engine.tree_stats()
    assert st.NumParticles == len(pos) and abs(st.root_mass - mass.astype(np.float64).sum()) < 1e-9 * mass.sum()


def test_tree_type_mask(pkg, engine, orc):
    """force_tree_rebuild_mask (forcetree.c:151-166, :802-807): only types whose bit is set enter the tree; garbage never."""
    pos, mass, box = pkg.ics.s_grid(12)
    P = pkg.make_particles(pos, mass)
    P["Type"] = np.arange(len(pos)) % 2          # gas / dark matter
    P["Flags"][5] = 1                            # IsGarbage
    engine.force_tree_rebuild_mask(P, box, pkg.engine.GASMASK)
    st = engine.tree_stats()
    want = ((P["Type"] == 0) & (P["Flags"] == 0)).sum()
    assert st.NumParticles == want and abs(st.root_mass - want) < 1e-9


def test_coincident_particles_error(pkg, engine):
    """More than 8 particles in one spot exhaust the node pool in the reference (forcetree.c:393-412): here an error."""
    pos = np.tile(np.array([[1.0, 2.0, 3.0]]), (12, 1))
    P = pkg.make_particles(pos, np.ones(12, np.float32))
    with pytest.raises(pkg.EngineError, match="coincident"):
        engine.force_tree_full(P, 10.0)


# ------------------------------------------------------------------------------- PM
@pytest.mark.parametrize("n,nmesh", [(16, 32), (20, 48)])
def test_pm_parity(pkg, engine, n, nmesh):
    pos, mass, box = pkg.ics.s_grid(n)
    pos[0] = [0.0, box, box / 2]                 # edge: Pos == Box lands in cell Nmesh and wraps (petapm.c:903-918)
    setup_engine(engine, box, n, nmesh)
    P = pkg.make_particles(pos, mass)
    P["Potential"] = 0.25                        # readout_potential accumulates (gravpm.c:499-501)
    engine.gravpm_force(P)
    gpm, pot = O.gravpm_force(pos, mass, box, nmesh, 1.5, G)
    assert np.abs(P["GravPM"] - gpm).max() <= 1e-11 * np.abs(gpm).mean()
    assert np.abs(P["Potential"] - (pot + 0.25)).max() <= 1e-11 * np.abs(pot).mean()


def test_host_path_skips_garbage_in_place(pkg, engine, orc):
    """Garbage and swallowed black holes stay in P[] until the next domain_decompose_full; the reference skips them in place: not
    deposited, no mesh force (gravpm.c:176-179: GravPM stays at the zero of gravpm.c:88-92), not in the tree (forcetree.c:806), no walk
    targets (treewalk.c:234).  3 % heavy dead records between the live ones: the live particles get the oracle's forces of the live set
    alone, the dead records come back with GravPM = 0 and their FullTreeGravAccel untouched."""
    n, nmesh = 16, 32
    pos, mass, box = pkg.ics.s_zel(n)
    N = len(pos)
    rng = np.random.RandomState(3)
    dead_src = np.sort(rng.choice(N, N // 33, replace=False))
    npos = np.insert(pos, dead_src, pos[dead_src] + 1e-3 * box / n, axis=0)
    nmass = np.insert(mass, dead_src, np.float32(50.0))
    is_dead = np.zeros(len(npos), bool)
    is_dead[dead_src + np.arange(len(dead_src))] = True
    assert np.array_equal(npos[~is_dead], pos)
    setup_engine(engine, box, n, nmesh, TreeUseBH=0)
    P = pkg.make_particles(npos, nmass)
    di = np.flatnonzero(is_dead)
    P["Flags"][di[0::2]] = 1                      # IsGarbage
    P["Flags"][di[1::2]] = 2                      # Swallowed ...
    P["Type"][di[1::2]] = 5                       # ... black holes
    P["GravPM"][di] = 7.0
    P["FullTreeGravAccel"][di] = 9.0
    engine.set_particle_epoch(0)
    engine.gravpm_force(P)
    engine.force_tree_full(P, box)
    assert engine.tree_stats().NumParticles == N
    engine.grav_short_tree(P)
    gpm, _ = O.gravpm_force(pos, mass, box, nmesh, 1.5, G)
    tr = orc.tree(pos, mass, box)
    par = O.make_grav_params(box, nmesh, npart_cbrt=n, G=G)
    par.TreeUseBH = 0
    a_ref, _, _, _ = tr.grav_short_tree(par, oldacc=np.sqrt((gpm ** 2).sum(1)) / G)
    live = ~is_dead
    assert np.abs(P["GravPM"][live] - gpm).max() <= 1e-11 * np.abs(gpm).mean()
    assert_accel_parity(P["FullTreeGravAccel"][live], a_ref)
    assert np.all(P["GravPM"][di] == 0.0) and np.all(P["FullTreeGravAccel"][di] == 9.0)


def test_pm_linearity_and_momentum(pkg, engine):
    """Size-independent properties: the PM force is linear in the masses and conserves momentum (sum m a = 0)."""
    n, nmesh = 24, 48
    pos, mass, box = pkg.ics.s_clust(n, box=100.0, seed=2)
    setup_engine(engine, box, n, nmesh)
    P = pkg.make_particles(pos, mass)
    engine.gravpm_force(P)
    g1 = P["GravPM"].copy()
    P2 = pkg.make_particles(pos, 3 * mass)
    engine.gravpm_force(P2)
    assert np.abs(P2["GravPM"] - 3 * g1).max() <= 1e-12 * np.abs(g1).max()
    assert np.abs((g1 * mass[:, None]).sum(0)).max() <= 1e-9 * np.abs(g1).sum()


# ------------------------------------------------------------------------------- walk
@pytest.mark.parametrize("ic,n,nmesh", [("s_grid", 32, 64), ("s_grid", 24, 72), ("s_clust", 20, 40), ("s_zel", 24, 48)])
def test_walk_parity(pkg, engine, orc, ic, n, nmesh):
    """PM + full tree + two walks (Barnes-Hut first, relative criterion second: test_gravity.c:211-213) vs the oracle."""
    if ic == "s_clust":
        pos, mass, box = pkg.ics.s_clust(n, box=8.0, seed=1)
    else:
        pos, mass, box = getattr(pkg.ics, ic)(n)
    setup_engine(engine, box, n, nmesh)
    engine.set_instrumentation(False, True)
    P = pkg.make_particles(pos, mass)
    engine.gravpm_force(P)
    gpm_o, _ = O.gravpm_force(pos, mass, box, nmesh, 1.5, G)
    P["GravPM"] = gpm_o          # feed both walks identical OldAcc inputs
    engine.force_tree_full(P, box)
    engine.grav_short_tree(P)
    c1h = engine.walk_counters()
    a1h = P["FullTreeGravAccel"].copy()
    assert engine.get_gravshort_treepar().TreeUseBH == 0      # gravshort-tree.c:148-151
    a1, a2, p2, c1, c2, _ = oracle_two_walks(orc, pos, mass, box, n, nmesh, gpm_o, want_pot=True)
    assert (c1h["pp"], c1h["nodes_visited"], c1h["nodes_used"]) == tuple(c1)
    assert_accel_parity(a1h, a1)
    P["FullTreeGravAccel"] = a1  # identical OldAcc for the second walk
    engine.grav_short_tree(P)
    c2h = engine.walk_counters()
    assert (c2h["pp"], c2h["nodes_visited"], c2h["nodes_used"]) == tuple(c2)
    assert_accel_parity(P["FullTreeGravAccel"], a2)
    assert np.abs(P["Potential"] - p2).max() <= 1e-10 * np.abs(p2).mean()
    engine.set_instrumentation(False, False)


@pytest.mark.parametrize("variant,cap", [(1, 512), (4, 512), (4, 48), (6, 512), (6, 40)])
@pytest.mark.parametrize("ic,n,nmesh", [("s_grid", 24, 48), ("s_clust", 20, 40), ("s_zel", 24, 48)])
def test_walk_kernel_variants(pkg, engine, orc, ic, n, nmesh, variant, cap):
    """Every walk kernel (lane-per-target 1, group-cooperative 4, two-kernel list/evaluate 6) takes the
    reference's decisions per target: equal counters and accelerations against the oracle.  The small list capacities force
    kernel 4 to drain its lists mid-walk and kernel 6 to send overflowing targets to its fallback."""
    if ic == "s_clust":
        pos, mass, box = pkg.ics.s_clust(n, box=8.0, seed=3)
    else:
        pos, mass, box = getattr(pkg.ics, ic)(n)
    setup_engine(engine, box, n, nmesh, TreeUseBH=0)
    gpm_o, _ = O.gravpm_force(pos, mass, box, nmesh, 1.5, G)
    tr = orc.tree(pos, mass, box)
    par = O.make_grav_params(box, nmesh, npart_cbrt=n, G=G)
    par.TreeUseBH = 0
    old = np.sqrt((gpm_o ** 2).sum(1)) / G
    a_ref, p_ref, c_ref, _ = tr.grav_short_tree(par, oldacc=old, want_pot=True)
    P = pkg.make_particles(pos, mass)
    P["GravPM"] = gpm_o
    P["FullTreeGravAccel"] = 0.0
    try:
        engine.set_walk_variant(variant)
        engine.set_walk_list_capacity(cap)
        engine.set_instrumentation(False, True)
        engine.force_tree_full(P, box)
        engine.grav_short_tree(P)
        c = engine.walk_counters()
    finally:
        engine.set_walk_variant(0)
        engine.set_walk_list_capacity(512)
        engine.set_instrumentation(False, False)
    assert (c["pp"], c["nodes_visited"], c["nodes_used"]) == tuple(c_ref)
    assert_accel_parity(P["FullTreeGravAccel"], a_ref)
    assert np.abs(P["Potential"] - p_ref).max() <= 1e-10 * np.abs(p_ref).mean()


@pytest.mark.parametrize("mode", ["relative", "bh", "aold0", "tiny_aold"])
def test_walk_lists_f32_preclassification(pkg, engine, orc, mode, monkeypatch):
    """Round 6: k_walk_lists8 with the node tests of a target pass pre-classified in fp32 (MPG_LISTS_F32=1; off by
    default - it measured slower than the fp64 tests).  The decisions must stay exactly the reference's: interaction counters EQUAL
    to the oracle's and accelerations to rounding, with the relative criterion, with the Barnes-Hut switch (aold = +inf in the kernel), with a
    zero old acceleration (every node with mass opens) and with old accelerations below what fp32 carries (those targets take the fp64 tests:
    the fall-back counter shows it).  The counting build evaluates BOTH forms for every pass and raises a device error on any difference on a
    lane the fp32 form called decided, so a green run also says that no fp32 decision disagreed with fp64 anywhere."""
    n, nmesh = 32, 64
    pos, mass, box = pkg.ics.s_zel(n)
    use_bh = 1 if mode == "bh" else 0
    setup_engine(engine, box, n, nmesh, TreeUseBH=use_bh)
    gpm_o, _ = O.gravpm_force(pos, mass, box, nmesh, 1.5, G)
    tr = orc.tree(pos, mass, box)
    par = O.make_grav_params(box, nmesh, npart_cbrt=n, G=G)
    par.TreeUseBH = use_bh
    scale = {"relative": 1.0, "bh": 1.0, "aold0": 0.0, "tiny_aold": 1e-40}[mode]
    prev = scale * gpm_o
    old = np.sqrt(((prev + scale * gpm_o) ** 2).sum(1)) / G
    a_ref, p_ref, c_ref, _ = tr.grav_short_tree(par, oldacc=old, want_pot=True)
    P = pkg.make_particles(pos, mass)
    P["GravPM"] = scale * gpm_o
    P["FullTreeGravAccel"] = prev
    monkeypatch.setenv("MPG_LISTS_F32", "1")
    try:
        engine.set_walk_variant(6)
        engine.set_instrumentation(False, True)
        engine.force_tree_full(P, box)
        engine.grav_short_tree(P)
        c = engine.walk_counters()
        fallback, f32_waves = engine.walk_f32_stats()
    finally:
        engine.set_walk_variant(0)
        engine.set_instrumentation(False, False)
    assert (c["pp"], c["nodes_visited"], c["nodes_used"]) == tuple(c_ref)
    assert_accel_parity(P["FullTreeGravAccel"], a_ref)
    assert np.abs(P["Potential"] - p_ref).max() <= 1e-10 * np.abs(p_ref).mean()
    assert f32_waves > 0.2 * len(pos) / 8, (f32_waves, len(pos) // 8)        # the interior waves (no target near a face) did run the fp32 tests
    if mode == "tiny_aold":
        assert fallback > f32_waves          # every pass of those waves went to fp64 (aold outside 1e-30 .. 1e30)
    else:
        assert fallback < 0.01 * c["nodes_visited"] / 8, (fallback, c["nodes_visited"])


def test_resident_walk_in_place_with_list_retry(pkg, engine, orc):
    """in resident mode the walk writes FullTreeGravAccel over its own opening input.  The two-kernel walk runs
    its list pass again with longer lists when more than a fifth of the targets overflow - after the evaluation has already stored new
    accelerations for the targets that fitted - so the opening input is now taken once, before the first kernel (k_oldacc).  A capacity
    of 16 entries forces such retries; results must equal the host path's (separate buffers) and the oracle's."""
    n, nmesh = 24, 48
    pos, mass, box = pkg.ics.s_zel(n)
    setup_engine(engine, box, n, nmesh, TreeUseBH=0)
    gpm_o, _ = O.gravpm_force(pos, mass, box, nmesh, 1.5, G)
    tr = orc.tree(pos, mass, box)
    par = O.make_grav_params(box, nmesh, npart_cbrt=n, G=G)
    par.TreeUseBH = 0
    rng = np.random.RandomState(5)
    prev = 1e-3 * np.abs(gpm_o).mean() * rng.standard_normal(gpm_o.shape)   # a previous acceleration unlike the new one
    a_ref, _, _, _ = tr.grav_short_tree(par, oldacc=np.sqrt(((prev + gpm_o) ** 2).sum(1)) / G)
    res = {}
    try:
        engine.set_walk_variant(6)
        for mode in ("host", "resident"):
            P = pkg.make_particles(pos, mass)
            P["GravPM"] = gpm_o
            P["FullTreeGravAccel"] = prev
            engine.set_walk_list_capacity(16)
            if mode == "resident":
                engine.resident_begin(P, box)
            engine.force_tree_full(P, box)
            engine.grav_short_tree(P)
            assert engine.walk_choice()[1] > 16     # the retry ran
            if mode == "resident":
                engine.resident_end(P)
            res[mode] = P["FullTreeGravAccel"].copy()
    finally:
        engine.set_walk_variant(0)
        engine.set_walk_list_capacity(512)
    assert_accel_parity(res["host"], a_ref)
    assert_accel_parity(res["resident"], a_ref)
    assert np.abs(res["resident"] - res["host"]).max() <= 1e-12 * np.abs(a_ref).mean()


def test_leaf_blocks_behind_the_tree_event(pkg, engine):
    """Round 5: with a PM force queued, force_tree_build runs on the engine's second stream and queues the leaves' source blocks (the
    evaluation kernel's input) BEHIND the event the main stream waits for, so that they are made beside the list kernel.  Every order of
    calls must give the walk of a tree built on the main stream: walk at once; a second build on the main stream while the first one's
    leaf-block kernel may still run (nothing waited for it); a gas-tree density in between is covered by the hydro tests."""
    import torch
    n, nmesh = 32, 64
    pos, mass, box = pkg.ics.s_zel(n)
    setup_engine(engine, box, n, nmesh, TreeUseBH=1)   # (the geometric criterion: the lists do not depend on an earlier walk)
    dev = torch.device("cuda", 0)
    d_pos, d_mass = torch.from_numpy(pos).to(dev), torch.from_numpy(mass).to(dev)
    N = len(pos)
    z3 = lambda: torch.zeros(N, 3, dtype=torch.float64, device=dev)
    gpm, pot = z3(), torch.zeros(N, dtype=torch.float64, device=dev)
    engine.dev_bind_particles(d_pos, d_mass, box)
    res = {}
    try:
        engine.set_walk_variant(6)
        for mode in ("main_stream", "beside_pm", "beside_pm_then_rebuilt", "beside_pm_twice"):
            acc, p2 = z3(), torch.zeros(N, dtype=torch.float64, device=dev)
            if mode != "main_stream":
                engine.dev_gravpm_force(gpm, pot)          # queues the PM force: the next build goes to the second stream
            engine.dev_force_tree_build()
            if mode == "beside_pm_then_rebuilt":
                engine.dev_force_tree_build()              # (no PM queued any more: on the main stream, over the tree the leaf-block kernel reads)
            if mode == "beside_pm_twice":
                engine.dev_gravpm_force(gpm, pot)
                engine.dev_force_tree_build()
            engine.dev_grav_short_tree(acc, potential=p2)
            torch.cuda.synchronize()
            res[mode] = (acc.cpu().numpy(), p2.cpu().numpy())
    finally:
        engine.set_walk_variant(0)
    a0, p0 = res["main_stream"]
    assert np.isfinite(a0).all() and np.abs(a0).max() > 0
    for mode, (a, p) in res.items():
        assert np.array_equal(a, a0) and np.array_equal(p, p0), mode


@pytest.mark.parametrize("ic", ["s_zel", "s_grid"])
def test_walk_64bit_offset_kernels(pkg, engine, orc, ic):
    """The two-kernel walk has variants with 64-bit offsets into the source / node arrays, taken when those exceed 4 GiB (512^3 particles in
    one tree: BASELINE configs[3] on one GPU); mpg_set_walk_offsets64 selects them at a size the oracle can check.  Same decisions, same
    results."""
    n, nmesh = 24, 48
    pos, mass, box = getattr(pkg.ics, ic)(n)
    setup_engine(engine, box, n, nmesh, TreeUseBH=0)
    gpm_o, _ = O.gravpm_force(pos, mass, box, nmesh, 1.5, G)
    tr = orc.tree(pos, mass, box)
    par = O.make_grav_params(box, nmesh, npart_cbrt=n, G=G)
    par.TreeUseBH = 0
    a_ref, p_ref, c_ref, _ = tr.grav_short_tree(par, oldacc=np.sqrt((gpm_o ** 2).sum(1)) / G, want_pot=True)
    P = pkg.make_particles(pos, mass)
    P["GravPM"] = gpm_o
    P["FullTreeGravAccel"] = 0.0
    try:
        engine.set_walk_variant(6)
        engine.set_walk_offsets64(True)
        engine.set_instrumentation(False, True)
        engine.force_tree_full(P, box)
        engine.grav_short_tree(P)
        c = engine.walk_counters()
    finally:
        engine.set_walk_variant(0)
        engine.set_walk_offsets64(False)
        engine.set_instrumentation(False, False)
    assert (c["pp"], c["nodes_visited"], c["nodes_used"]) == tuple(c_ref)
    assert_accel_parity(P["FullTreeGravAccel"], a_ref)
    assert np.abs(P["Potential"] - p_ref).max() <= 1e-10 * np.abs(p_ref).mean()


@pytest.mark.parametrize("name", ["grav_sgrid16", "grav_sclust12"])
def test_committed_oracle_vectors(pkg, engine, name):
    """The committed vectors of tests/golden/grav_*.npz are outputs of the ORACLE (make_golden.py), frozen at the state in which it
    reproduced the reference's known answers - a regression fixture for both sides, not an independent reference-derived pin (those
    are test_reference_probe_known_answer, test_reference_force_accuracy_vs_direct_sum_on_gpu and tests/test_oracle_kat.py)."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    n, nmesh, box = int(g["n"]), int(g["nmesh"]), float(g["box"])
    if name == "grav_sgrid16":
        pos, mass, _ = pkg.ics.s_grid(n)
    else:
        pos, mass, _ = pkg.ics.s_clust(n, box=box, seed=int(g["seed"]))
    setup_engine(engine, box, n, nmesh)
    engine.set_instrumentation(False, True)
    P = pkg.make_particles(pos, mass)
    engine.gravpm_force(P)
    assert np.abs(P["GravPM"] - g["GravPM"]).max() <= 1e-11 * np.abs(g["GravPM"]).mean()
    P["GravPM"] = g["GravPM"]
    engine.force_tree_full(P, box)
    assert engine.tree_stats().numnodes <= int(g["numnodes"])       # the oracle count includes pruned empty cells
    engine.grav_short_tree(P)
    assert_accel_parity(P["FullTreeGravAccel"], g["Accel1"])
    P["FullTreeGravAccel"] = g["Accel1"]
    engine.grav_short_tree(P)
    c = engine.walk_counters()
    assert (c["pp"], c["nodes_visited"], c["nodes_used"]) == tuple(g["counters2"])
    assert_accel_parity(P["FullTreeGravAccel"], g["Accel2"])
    engine.set_instrumentation(False, False)


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_reference_probe_known_answer(pkg, engine, idx):
    """the unmodified reference gives mean|FullTreeGravAccel| = 1.67498e-05, Ninteractions/N = 1333.8 on the 32^3
    S-grid set with Nmesh 64, 1.67329e-05 / 1333.9 on 64^3 with Nmesh 128, and 1.43708e-05 / 512.0 on BASELINE configs[0]'s shape
    (examples/dm-small: 64^3 particles, Nmesh = 3 x 64 = 192); short range only, GravPM = 0, second walk."""
    k = np.load(os.path.join(GOLD, "reference_probe_kat.npz"))
    n, nmesh = int(k["n"][idx]), int(k["nmesh"][idx])
    assert (n, nmesh) == ((32, 64), (64, 128), (64, 192))[idx]
    pos, mass, box = pkg.ics.s_grid(n)
    setup_engine(engine, box, n, nmesh)
    engine.set_instrumentation(False, True)
    P = pkg.make_particles(pos, mass)
    engine.force_tree_full(P, box)
    engine.grav_short_tree(P)
    engine.grav_short_tree(P)
    c = engine.walk_counters()
    assert abs(np.abs(P["FullTreeGravAccel"]).mean() / k["mean_abs_accel"][idx] - 1) < 5e-6
    assert abs(c["pp"] / len(pos) - k["ninteractions_per_particle"][idx]) < 0.06
    engine.set_instrumentation(False, False)


def test_active_subset_and_accelstore(pkg, engine, orc):
    """ActiveParticle list + external AccelStore (gravshort-tree.c:106-111, hierarchical gravity timestep.c:454-456)."""
    n, nmesh = 20, 40
    pos, mass, box = pkg.ics.s_zel(n)
    setup_engine(engine, box, n, nmesh, TreeUseBH=0)
    P = pkg.make_particles(pos, mass)
    rng = np.random.RandomState(0)
    P["FullTreeGravAccel"] = rng.standard_normal((len(pos), 3)) * 1e-5
    old = np.sqrt((P["FullTreeGravAccel"] ** 2).sum(1)) / G
    before = P["FullTreeGravAccel"].copy()
    engine.force_tree_full(P, box)
    act = np.sort(rng.choice(len(pos), 777, replace=False)).astype(np.int32)
    store = np.full((len(pos), 3), np.nan)
    engine.grav_short_tree(P, ActiveParticle=act, AccelStore=store)
    tr = orc.tree(pos, mass, box)
    par = O.make_grav_params(box, nmesh, npart_cbrt=n, G=G)
    par.TreeUseBH = 0
    a, _, _, _ = tr.grav_short_tree(par, oldacc=old, active=act)
    assert_accel_parity(store[act], a[act])
    inactive = np.setdiff1d(np.arange(len(pos)), act)
    assert np.all(np.isnan(store[inactive]))                       # untouched
    assert np.array_equal(P["FullTreeGravAccel"][inactive], before[inactive])
    assert_accel_parity(P["FullTreeGravAccel"][act], a[act])       # full tree: postprocess stores it (gravshort.h:54-59)


def test_erfc_window(pkg, engine, orc):
    n, nmesh = 16, 32
    pos, mass, box = pkg.ics.s_grid(n)
    setup_engine(engine, box, n, nmesh, TreeUseBH=1, window=1)
    P = pkg.make_particles(pos, mass)
    engine.force_tree_full(P, box)
    engine.grav_short_tree(P)
    orc.fill_ntab(1, 1.5)
    try:
        tr = orc.tree(pos, mass, box)
        par = O.make_grav_params(box, nmesh, npart_cbrt=n, G=G)
        par.TreeUseBH = 1
        a, _, _, _ = tr.grav_short_tree(par, oldacc=np.zeros(len(pos)))
    finally:
        orc.fill_ntab(0, 1.5)
    assert_accel_parity(P["FullTreeGravAccel"], a)
    engine.gravshort_fill_ntab(0, 1.5)


def test_error_behaviour(pkg, engine):
    """The reference endrun()s in these situations; the C-ABI returns an error that the mirror raises."""
    with pytest.raises(pkg.EngineError, match="calibrated for Asmth = 1.5"):
        engine.gravshort_fill_ntab(0, 1.25)                       # gravity.c:25-29
    engine.gravshort_fill_ntab(0, 1.5)
    pos, mass, box = pkg.ics.s_grid(8)
    P = pkg.make_particles(pos, mass)
    engine.force_tree_free()
    with pytest.raises(pkg.EngineError, match="before tree moments"):
        engine.grav_short_tree(P)                                 # gravshort-tree.c:113-114


@pytest.mark.parametrize("kind", ["close", "random", "random2"])
def test_reference_force_accuracy_vs_direct_sum_on_gpu(pkg, engine, orc, kind):
    """The reference's own acceptance test of this path, do_force_test + check_against_force_direct (test_gravity.c:146-219), run on
    the engine through the drop-in calls: 16^3 particles in a box of 8, Nmesh 48, Asmth 1.5, Rcut 7, Barnes-Hut opening on both
    walks; PM + tree against the direct sum over 27 images: max relative error < 3 ErrTolForceAcc, mean < 0.8 ErrTolForceAcc.
    "random" / "random2" are the two particle sets test_force_random draws from gsl_rng_mt19937 (seed 0), regenerated bit for bit
    by oracle/mt19937.py."""
    from test_oracle_kat import _test_gravity_sets
    pos, mass, box, n = _test_gravity_sets(kind)
    nmesh, err = 48, 0.002
    engine.gravshort_fill_ntab(0, 1.5)
    engine.gravpm_init_periodic(box, 1.5, nmesh, G)
    engine.set_gravshort_treepar(ErrTolForceAcc=err, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.0, TreeUseBH=1, Rcut=7.0)
    engine.gravshort_set_softenings(box / n)
    P = pkg.make_particles(pos, mass)
    engine.gravpm_force(P)
    engine.force_tree_full(P, box)
    engine.grav_short_tree(P)
    engine.grav_short_tree(P)
    par = O.make_grav_params(box, nmesh, npart_cbrt=n, G=G)
    direct = orc.force_direct(pos, mass, box, par.h, G)
    relerr = np.abs(direct - (P["GravPM"] + P["FullTreeGravAccel"])) / np.abs(direct).mean()
    assert relerr.max() < 3 * err, relerr.max()
    assert relerr.mean() < 0.8 * err, relerr.mean()


def test_load_order_torch_first():
    """The library must also work when torch (with its bundled HIP runtime) was imported first."""
    import subprocess
    import sys
    code = ("import torch, importlib, sys; sys.path.insert(0, %r); torch.zeros(1).cuda();"
            "p = importlib.import_module('mp-gadget_amd'); e = p.Engine(0); print(e.version()); e.close()")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code % root], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "gfx950" in out.stdout, out.stderr[-2000:]


def _run_mgpu(tmp_path, name, nproc, mode, port, ic="s_zel", n=40, env_extra=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / name)
    env = dict(os.environ, MPG_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MPG_MGPU_MODE=mode, MPG_MGPU_IC=ic, **(env_extra or {}))
    script = os.path.join(root, "tools", "mgpu_check.py")
    if nproc == 1:
        cmd = [sys.executable, script, out, str(n)]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(port), script, out, str(n)]
    run_ranks(cmd, env, out)
    return np.load(out)


@keep_artifacts_on_failure
@pytest.mark.parametrize("ic", ["s_zel", "s_clust"])
def test_peano_domain_ranks_match_one(tmp_path, ic):
    """The force step on the reference's own decomposition, through the library's choreography (csrc/dist.hip, mpg_dist_*): particles on
    the owners of their Peano-Hilbert TopLeaves (domain_decompose_full + exchange), PM by shipping particles to the x-slabs and the
    results back, ghosts in whole level-La tree cells around the rank's TopLeaves, the nodes above from an all-reduce.  1 rank (no
    communicator), 2, 3 and 4 ranks (gloo, sharing this GPU; the collectives are the mpg_comm callbacks) against one GPU: the same
    decisions except where a node's moments, summed in another order, sit within an ulp of an opening threshold."""
    n = 36                                                            # Nmesh = 72: a multiple of 2, 3 and 4
    one = _run_mgpu(tmp_path, "one.npy", 1, "single", 0, ic=ic, n=n)
    for name, nproc, port in (("p1.npy", 1, 0), ("p2.npy", 2, 29601), ("p3.npy", 3, 29602), ("p4.npy", 4, 29603)):
        d = _run_mgpu(tmp_path, name, nproc, "peano" if nproc > 1 else "peano1", port, ic=ic, n=n)
        assert_accel_parity(d[:, 0:3], one[:, 0:3])
        assert np.abs(d[:, 3:6] - one[:, 3:6]).max() <= 1e-11 * np.abs(one[:, 3:6]).mean(), name
        # P[].Potential after a PM step is the tree's (grav_short_reduce assigns it in primary mode, gravshort.h:94-95); the one-rank
        # run of the same code is the reference for it
        if nproc == 1:
            pot1 = d[:, 6]
        assert np.abs(d[:, 6] - pot1).max() <= 1e-9 * np.abs(pot1).mean(), name


@keep_artifacts_on_failure
def test_peano_domain_walk_in_place_on_ranks(tmp_path):
    """mpg_dist_gravity_step with prev_accel aliased to accel (the resident caller's FullTreeGravAccel) on 1 and 3 ranks: the opening
    input is taken for the walk's targets only - the engine is bound to own + ghost rows, the caller's arrays hold the own rows - and the
    result equals the walk into a separate array (asserted inside the tool on every rank) and the one-GPU second step."""
    n = 36
    env = {"MPG_INPLACE": "1"}
    for name, nproc, port in (("p1.npy", 1, 0), ("p3.npy", 3, 29612)):
        d = _run_mgpu(tmp_path, name, nproc, "peano" if nproc > 1 else "peano1", port, ic="s_clust", n=n, env_extra=env)
        if nproc == 1:
            one = d
        assert np.abs(d[:, 0:3]).min() > 0
        assert_accel_parity(d[:, 0:3], one[:, 0:3])


@keep_artifacts_on_failure
def test_peano_domain_substep_active_subset(tmp_path):
    """A sub-step on several ranks (run.c:392-470 without the hierarchical trees): the tree holds every particle, every fifth one is
    active and walked (mpg_dist_dev_grav_short_tree_active).  The active particles get the accelerations of the one-GPU sub-step,
    nothing else is written."""
    n = 36
    env = {"MPG_ACTIVE_EVERY": "5"}
    one = _run_mgpu(tmp_path, "one.npy", 1, "single", 0, ic="s_clust", n=n, env_extra=env)
    act = np.arange(len(one)) % 5 == 0
    assert np.abs(one[act, 0:3]).min() > 0 and np.all(one[~act, 0:3] == 0)
    for name, nproc, port in (("p1.npy", 1, 0), ("p3.npy", 3, 29608)):
        d = _run_mgpu(tmp_path, name, nproc, "peano" if nproc > 1 else "peano1", port, ic="s_clust", n=n, env_extra=env)
        assert np.all(d[~act, 0:3] == 0), name
        assert_accel_parity(d[act, 0:3], one[act, 0:3])


@keep_artifacts_on_failure
def test_peano_domain_hierarchical_active_tree(tmp_path):
    """The hierarchical gravity loop's force on several ranks (force_tree_active_moments + grav_short_tree on the ACTIVE particles only):
    every seventh particle is active, spread over 3 ranks; mpg_dist_dev_grav_short_tree_active_tree gathers the (small) active set on
    every rank, builds the tree one GPU builds and walks the rank's members: the one-GPU accelerations, to rounding."""
    n = 36
    env = {"MPG_ACTIVE_EVERY": "7", "MPG_ACTIVE_TREE": "1"}
    one = _run_mgpu(tmp_path, "one.npy", 1, "single", 0, ic="s_clust", n=n, env_extra=env)
    act = np.arange(len(one)) % 7 == 0
    assert np.abs(one[act, 0:3]).min() > 0 and np.all(one[~act, 0:3] == 0)
    for name, nproc, port, form in (("p1.npy", 1, 0, "1"), ("p3.npy", 3, 29611, "1"), ("h2.npy", 2, 29612, "host")):
        d = _run_mgpu(tmp_path, name, nproc, "peano" if nproc > 1 else "peano1", port, ic="s_clust", n=n,
                      env_extra=dict(env, MPG_ACTIVE_TREE=form))       # host: mpg_dist_grav_short_tree_active_tree on particle_data records
        assert np.all(d[~act, 0:3] == 0), name
        assert np.abs(d[act, 0:3] - one[act, 0:3]).max() <= 1e-13 * np.abs(one[act, 0:3]).max(), name


@pytest.mark.parametrize("ic", ["s_grid", "s_zel", "s_clust"])
def test_full_size_256_properties(pkg, orc, ic):
    _full_size_properties(pkg, orc, ic, 256)


def test_c4_full_size_512_against_oracle(pkg, orc):
    """BASELINE configs[3]'s WHOLE particle set (512^3 = 134 M particles, Nmesh 1024) on the one GPU, CHECKED against the oracle as the
    256^3 sets are: GravPM of all 134 M particles against oracle/pm_oracle.c at Nmesh 1024 (<= 1e-10 of the mean) and 2048 sampled
    targets against the oracle walking the oracle-built tree of all particles (assert_accel_parity) - the walk here runs its
    64-bit-offset kernels and slices its list area (this run asserted counter ranges only)."""
    _full_size_properties(pkg, orc, "s_zel", 512)


def _full_size_properties(pkg, orc, ic, n):
    """256^3, Nmesh 512 (BASELINE configs[1]) on the device-resident path, on the three input sets of SURVEY 8(d) (the jittered grid,
    the Zel'dovich-displaced grid of the headline, the strongly clustered set): size-independent properties + a sampled oracle
    comparison.
      * S-grid opens every node, so the short-range force is a pure pair sum; pairs are antisymmetric except where
        only one of the two targets keeps the other's leaf (cube cut at Rcut + len/2, gravshort-tree.c:198-215), where
        the window has already suppressed the force by >1e4: |sum_i a_i| <= 1e-6 sum_i |a_i|;
      * PM momentum conservation: sum_i m_i GravPM_i = 0;
      * GravPM of ALL 16.8 M particles against the CPU long-range step at the same size (oracle/pm_oracle.c's loops and sweeps +
        pocketfft on the 512^3 mesh): 1e-10 of the mean |GravPM|;
      * 2048 random targets agree with the oracle walking the oracle-built tree of all 16.8 M particles."""
    import torch
    clk = phase_clock("full_size_%d[%s]" % (n, ic))
    nmesh = 2 * n
    pos, mass, box = getattr(pkg.ics, ic)(n)
    clk.mark("ics")
    N = len(pos)
    eng = pkg.Engine(0)
    setup_engine(eng, box, n, nmesh, TreeUseBH=0)
    eng.set_instrumentation(False, True)
    dpos, dmass = torch.from_numpy(pos).cuda(), torch.from_numpy(mass).cuda()
    eng.dev_bind_particles(dpos, dmass, box)
    gpm = torch.zeros(N, 3, dtype=torch.float64, device="cuda")
    acc = torch.zeros_like(gpm)
    old = torch.full((N,), 1e-7, dtype=torch.float64, device="cuda")
    eng.dev_gravpm_force(gpm, None)
    eng.dev_force_tree_build()
    if ic != "s_grid":       # a realistic OldAcc for the relative criterion (with 1e-7 every node of the dense clump would be opened)
        old = torch.clamp(gpm.norm(dim=1) / G, min=1e-7).contiguous()
    eng.synchronize()
    clk.mark("pm+tree")
    eng.dev_grav_short_tree(acc, oldacc=old)
    eng.synchronize()
    clk.mark("walk (kernel, capacity, fallback targets) = %s" % (eng.walk_choice(),))
    c = eng.walk_counters()
    a = acc.cpu().numpy()
    g = gpm.cpu().numpy()
    if ic == "s_grid":                                            # (every node is opened there: a pure pair sum)
        assert c["nodes_used"] == 0
        assert np.abs(a.sum(0)).max() <= 1e-6 * np.abs(a).sum()
    else:                                                         # monopoles break the pair antisymmetry at the level of ErrTolForceAcc
        assert c["nodes_used"] > 0
        assert np.abs(a.sum(0)).max() <= 0.002 * np.abs(a).sum()
    assert np.all(np.isfinite(a)) and np.all(np.isfinite(g))
    assert np.abs(g.sum(0)).max() <= 1e-9 * np.abs(g).sum()
    st = eng.tree_stats()
    assert st.NumParticles == N and abs(st.root_mass - N) < 1e-6
    clk.mark("checks")
    gc, _ = O.gravpm_force_c(orc, pos, mass, box, nmesh, 1.5, G, want_potential=False)
    clk.mark("oracle PM, all particles")
    assert np.abs(g - gc).max() <= 1e-10 * np.abs(gc).mean()
    del gc
    tr = orc.tree(pos, mass, box, father=False)
    clk.mark("oracle tree")
    assert tr.numnodes >= st.numnodes
    par = O.make_grav_params(box, nmesh, npart_cbrt=n, G=G)
    par.TreeUseBH = 0
    act = np.sort(np.random.RandomState(1).choice(N, 2048, replace=False)).astype(np.int32)
    ao, _, co, _ = tr.grav_short_tree(par, oldacc=old.cpu().numpy(), active=act)
    clk.mark("oracle walk")
    clk.write()
    assert_accel_parity(a[act], ao[act])
    eng.close()


@pytest.mark.parametrize("ic,n", [("s_grid", 20), ("s_clust", 16)])
def test_grav_short_pair(pkg, engine, orc, ic, n):
    """grav_short_pair (gravshort-pair.c): the exact pair-wise short-range force within the Rcut sphere, against the oracle's
    O(N^2) restatement; and, as runtests.c:131-176 does with it, the tree force against the pair-wise force."""
    nmesh = 2 * n
    pos, mass, box = (pkg.ics.s_clust(n, box=8.0, seed=2) if ic == "s_clust" else pkg.ics.s_grid(n))
    setup_engine(engine, box, n, nmesh, TreeUseBH=0)
    P = pkg.make_particles(pos, mass)
    P["GravPM"] = 0.0
    engine.force_tree_full(P, box)
    engine.grav_short_pair(P, 6.0)
    a_pair = P["FullTreeGravAccel"].copy()
    par = O.make_grav_params(box, nmesh, npart_cbrt=n, G=G)
    a_ref = orc.grav_short_pair(pos, mass, box, par, 6.0 * 1.5 * box / nmesh)
    assert_accel_parity(a_pair, a_ref)
    # active subset through the device entry point
    import torch
    act = np.sort(np.random.RandomState(0).choice(len(pos), 300, replace=False)).astype(np.int32)
    d_acc = torch.zeros(len(pos), 3, dtype=torch.float64, device="cuda")
    engine.dev_grav_short_pair(d_acc, 6.0, active=torch.from_numpy(act).cuda())
    engine.synchronize()
    got = d_acc.cpu().numpy()
    assert_accel_parity(got[act], a_ref[act])
    assert np.all(np.delete(got, act, axis=0) == 0)
    # tree force vs pair-wise force, as runtests.c:131-176 compares them: they differ by the tree's node approximations
    # (ErrTolForceAcc 0.002) and by the window's tail between the Rcut sphere and the cube the walk accepts
    P["FullTreeGravAccel"] = a_pair
    engine.grav_short_tree(P)
    rel = np.sqrt(((P["FullTreeGravAccel"] - a_pair) ** 2).sum(1)) / np.sqrt((a_pair ** 2).sum(1)).mean()
    assert rel.mean() < 2.5 * 0.002 and rel.max() < 50 * 0.002, (rel.mean(), rel.max())


def test_force_tree_active_moments(pkg, engine, orc):
    """force_tree_active_moments (forcetree.c:129-148): a tree of the active particles only (the hierarchical-gravity level loop
    re-enters tree build + walk per time bin with such trees).  Tree and walk equal the oracle's on the sub-set."""
    n, nmesh = 20, 40
    pos, mass, box = pkg.ics.s_zel(n)
    N = len(pos)
    setup_engine(engine, box, n, nmesh, TreeUseBH=0)
    act = np.sort(np.random.RandomState(7).choice(N, N // 3, replace=False)).astype(np.int32)
    P = pkg.make_particles(pos, mass)
    rng = np.random.RandomState(8)
    P["GravPM"] = rng.standard_normal((N, 3)) * 1e-3
    P["FullTreeGravAccel"] = rng.standard_normal((N, 3)) * 1e-3
    before = P["FullTreeGravAccel"].copy()
    engine.force_tree_active_moments(P, box, act)
    st = engine.tree_stats()
    tr = orc.tree(pos[act], mass[act], box)
    assert st.NumParticles == len(act) and st.numnodes == int(tr.export()["live"].sum())   # (the oracle also counts empty child slots)
    assert abs(st.root_mass - mass[act].astype(np.float64).sum()) <= 1e-9 * len(act)
    store = np.zeros((N, 3))
    engine.set_instrumentation(False, True)
    engine.grav_short_tree(P, ActiveParticle=act, AccelStore=store)
    c = engine.walk_counters()
    engine.set_instrumentation(False, False)
    par = O.make_grav_params(box, nmesh, npart_cbrt=n, G=G)
    par.TreeUseBH = 0
    old = np.sqrt(((before + P["GravPM"]) ** 2).sum(1)) / G
    a_ref, _, c_ref, _ = tr.grav_short_tree(par, oldacc=old[act])
    assert (c["pp"], c["nodes_visited"], c["nodes_used"]) == tuple(c_ref)
    assert_accel_parity(store[act], a_ref)
    assert np.array_equal(P["FullTreeGravAccel"], before)            # not a full particle tree: P is left alone (gravshort.h:54-66)
    assert np.all(np.delete(store, act, axis=0) == 0)
    # ActiveParticle == NULL: the full tree, flagged as such
    engine.force_tree_active_moments(P, box, None)
    engine.grav_short_tree(P)
    assert not np.array_equal(P["FullTreeGravAccel"], before)


def test_host_path_particle_epoch(pkg, engine):
    """mpg_set_particle_epoch: with the same non-zero epoch the upload of Pos / Mass / Type is reused (results identical); a new epoch
    (or epoch 0) picks up changed positions."""
    n, nmesh = 12, 24
    pos, mass, box = pkg.ics.s_zel(n)
    setup_engine(engine, box, n, nmesh, TreeUseBH=0)
    P = pkg.make_particles(pos, mass)

    start = [None]

    def step():
        if start[0] is not None:
            P["FullTreeGravAccel"] = start[0]            # the same OldAcc for every step that is compared
        engine.gravpm_force(P)
        engine.force_tree_full(P, box)
        engine.grav_short_tree(P)
        return P["GravPM"].copy(), P["FullTreeGravAccel"].copy()

    engine.set_particle_epoch(0)
    start[0] = step()[1]
    g0, a0 = step()
    engine.set_particle_epoch(7)
    g1, a1 = step()
    same = lambda x, y: np.abs(x - y).max() <= 1e-10 * np.abs(y).max()                      # (the CIC deposit sums with atomics)
    assert same(g1, g0) and np.abs(a1 - a0).max() <= 1e-9 * np.abs(a0).max()               # (a1 uses a0 as OldAcc: same decisions)
    P["Pos"][:, 0] = np.mod(P["Pos"][:, 0] + 0.37 * box / n, box)                           # the table changes ...
    engine.set_particle_epoch(7)
    P["FullTreeGravAccel"] = start[0]
    engine.force_tree_full(P, box)                                                          # ... but the epoch says it did not: stale upload
    engine.grav_short_tree(P)
    stale = P["FullTreeGravAccel"].copy()
    assert np.abs(stale - a1).max() <= 1e-9 * np.abs(a1).max()
    engine.set_particle_epoch(8)
    g2, a2 = step()
    assert not same(g2, g1) and np.abs(a2 - stale).max() > 1e-6 * np.abs(a2).max()
    engine.set_particle_epoch(0)
    g3, a3 = step()
    assert same(g3, g2)


@pytest.mark.parametrize("n", [20, 64])
def test_host_path_overlap_gives_the_synchronous_results(pkg, engine, n):
    """mpg_set_host_overlap: one packing pass per epoch (Pos / Mass / Type / Potential / FullTreeGravAccel), OldAcc on the device, and the
    write-back of gravpm_force's GravPM / Potential on a copy stream + host thread while the tree build and the walk run.  The three calls
    leave in P[] what the synchronous path leaves (GravPM, FullTreeGravAccel, the TREE's Potential - the PM step's write-back must land
    before it), over three steps with changing positions, with garbage in the table, with mpg_host_results_sync between the calls, and
    with the walk cut into slices whose results go down while the next slice is walked."""
    nmesh = 2 * n                                               # (64^3: slices long enough for the write-back thread to run beside a walk)
    pos, mass, box = pkg.ics.s_clust(n, seed=3) if n < 32 else pkg.ics.s_zel(n)
    setup_engine(engine, box, n, nmesh, TreeUseBH=0)
    N = len(pos)
    rng = np.random.RandomState(2)
    dead = rng.random_sample(N) < 0.02
    res = {}
    for mode in ("sync", "overlap", "overlap+sync_call", "overlap_sliced", "overlap+prefetch"):
        P = pkg.make_particles(pos, mass)
        P["Flags"][dead] = 1
        P["Potential"] = 0.125                                  # (gravpm_force accumulates onto it; the walk then assigns the tree's)
        # (3: the walk in three slices of the tree order, each written back while the next is walked - the path a 256^3 table takes)
        engine.set_host_overlap(0 if mode == "sync" else (3 if mode == "overlap_sliced" else 1))       # ("overlap+prefetch": 1)
        out = []
        for step in range(3):
            engine.set_particle_epoch(100 * (1 + len(res)) + step + 1)
            if mode == "overlap+prefetch":
                # (round 6) the packing pass + uploads started as soon as P[] is final for the step - the end of drift_all_particles in run.c -
                # on a host thread; gravpm_force joins it.  Once a step the upload is made useless by a new epoch (an exchange after the
                # drift): the calls must then pack again and not use what the prefetch staged.
                engine.host_prefetch(P, box)
                if step == 1:
                    engine.set_particle_epoch(100 * (1 + len(res)) + 50)     # (what mpg_shim_particles_changed leads to: joins the prefetch FIRST)
                    P["Pos"][:, 2] = np.mod(P["Pos"][:, 2] + 0.13 * box / n, box)
                    P["Pos"][P["Pos"] <= 0] += box
            elif step == 1:
                P["Pos"][:, 2] = np.mod(P["Pos"][:, 2] + 0.13 * box / n, box)       # (the same move in every mode)
                P["Pos"][P["Pos"] <= 0] += box
            engine.gravpm_force(P)
            if mode == "overlap+sync_call":
                engine.host_results_sync()
                assert np.abs(P["GravPM"][~dead]).max() > 0      # (in P[] now, as a host module reading it between the calls needs it)
            engine.force_tree_full(P, box)
            engine.grav_short_tree(P)
            out.append((P["GravPM"].copy(), P["FullTreeGravAccel"].copy(), P["Potential"].copy()))
            P["Pos"][:, 1] = np.mod(P["Pos"][:, 1] + 0.21 * box / n, box)
            P["Pos"][P["Pos"] <= 0] += box
        res[mode] = out
        engine.set_particle_epoch(0)
    engine.set_host_overlap(False)
    for mode in ("overlap", "overlap+sync_call", "overlap_sliced", "overlap+prefetch"):
        for (g0, a0, p0), (g1, a1, p1) in zip(res["sync"], res[mode]):
            assert np.abs(g1 - g0).max() <= 1e-10 * np.abs(g0).max() and np.abs(a1 - a0).max() <= 1e-9 * np.abs(a0).max(), mode
            assert np.abs(p1 - p0).max() <= 1e-9 * np.abs(p0).max(), mode
            assert np.all(g1[dead] == 0) and np.all(a1[dead] == 0) and np.all(p1[dead] == 0.125)         # garbage: GravPM zeroed, nothing else touched
    assert np.abs(res["sync"][0][2][~dead] - 0.125).min() > 0 and np.abs(res["sync"][1][0] - res["sync"][0][0]).max() > 0


def test_host_overlap_second_walk_of_an_epoch_opens_with_the_new_acceleration(pkg, engine):
    """with mpg_set_host_overlap the first walk of an epoch takes OldAcc on the device from the FullTreeGravAccel that
    went up with the epoch's one packing pass.  A SECOND grav_short_tree in the same epoch (the hierarchical level loop, a repeated call) must
    open its nodes with what the first walk wrote into P[] - grav_short_copy reads P[].FullTreeGravAccel (gravshort.h:82-86) - not with the
    staged pre-step copy: results of two walks per epoch with overlap equal those of the synchronous calls, and differ from a walk that keeps
    the first opening input (so the test can see the difference)."""
    n, nmesh = 24, 48
    pos, mass, box = pkg.ics.s_clust(n, seed=5)
    setup_engine(engine, box, n, nmesh, TreeUseBH=0)
    N = len(pos)
    rng = np.random.RandomState(4)
    res = {}
    for mode in ("sync", "overlap"):
        P = pkg.make_particles(pos, mass)
        P["FullTreeGravAccel"] = 1e-6 * rng.standard_normal((N, 3)) if mode == "sync" else res["start"]
        if mode == "sync":
            res["start"] = P["FullTreeGravAccel"].copy()          # a tiny old acceleration: the first walk opens far more nodes than the second
        engine.set_host_overlap(0 if mode == "sync" else 1)
        engine.set_particle_epoch(7000 + len(res))
        engine.gravpm_force(P)
        engine.force_tree_full(P, box)
        engine.grav_short_tree(P)
        first = P["FullTreeGravAccel"].copy()
        engine.grav_short_tree(P)                                 # same epoch, same tree: OldAcc is now |first + GravPM| / G
        res[mode] = (first, P["FullTreeGravAccel"].copy())
        engine.set_particle_epoch(0)
    engine.set_host_overlap(False)
    (f0, s0), (f1, s1) = res["sync"], res["overlap"]
    scale = np.abs(f0).max()
    assert np.abs(f1 - f0).max() <= 1e-9 * scale and np.abs(s1 - s0).max() <= 1e-9 * scale
    assert np.abs(s0 - f0).max() > 1e-6 * scale                  # (the two opening inputs do give different forces: the check above is not vacuous)


def test_pm_power_spectrum(pkg, engine, tmp_path):
    """The matter power spectrum measured during gravpm_force (gravpm.c:331-382, powerspectrum.c:55-122) against the numpy
    restatement: mode counts equal, k and P(k) to rounding; the saved file has the reference's columns."""
    import torch
    n, nmesh = 24, 48
    pos, mass, box = pkg.ics.s_zel(n)
    setup_engine(engine, box, n, nmesh)
    mpc = box / 1000.0
    d_pos, d_mass = torch.from_numpy(pos).cuda(), torch.from_numpy(mass).cuda()
    gpm = torch.zeros(len(pos), 3, dtype=torch.float64, device="cuda")
    engine.dev_bind_particles(d_pos, d_mass, box)
    engine.dev_gravpm_force(gpm, None)
    k, P, N = engine.gravpm_get_powerspectrum(nmesh, mpc)
    ko, Po, No = O.pm_power_spectrum(pos, mass, box, nmesh, mpc)
    assert np.array_equal(N, No) and N.sum() == nmesh ** 3 - 1                              # every mode but k = 0, weights included
    assert np.allclose(k, ko, rtol=1e-12) and np.allclose(P, Po, rtol=1e-9)
    assert len(k) > 20 and np.all(np.diff(k) > 0)
    engine.powerspectrum_save(str(tmp_path), "powerspectrum", 0.1, 0.5, k, P, N)
    lines = open(tmp_path / "powerspectrum-0.1000.txt").read().split("\n")
    assert lines[0] == "# in Mpc/h Units " and lines[2] == "# k P N P(z=0)"
    row = lines[3].split()
    assert abs(float(row[0]) / k[0] - 1) < 1e-5 and int(row[2]) == N[0] and abs(float(row[3]) / (P[0] / 0.25) - 1) < 1e-5
    # the measurement can be switched off; the forces do not depend on it
    g1 = gpm.cpu().numpy().copy()
    engine.gravpm_measure_power(False)
    engine.dev_gravpm_force(gpm, None)
    engine.gravpm_measure_power(True)
    engine.synchronize()
    assert np.abs(gpm.cpu().numpy() - g1).max() <= 1e-10 * np.abs(g1).max()


def test_rccl_one_rank_group_matches_single(tmp_path):
    """The library's multi-rank choreography (PM shipping, ghost import, all-reduced top of the tree: csrc/dist.hip) with its collectives
    through RCCL itself (backend "nccl", a one-rank group on this GPU) against the plain single-GPU step.  (The gloo tests above cover
    several ranks; this one covers the backend the multi-GPU runs use.  Round 1 found RCCL returning garbage beyond 1 GiB per
    all_to_all_single call - tools/a2a_selftest.py - hence TorchComm's piecewise path, A2A_MAX_BYTES.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tools", "mgpu_check.py")
    res = {}
    for k, (mode, force) in enumerate((("single", ""), ("peano1", "1"))):
        out = str(tmp_path / (mode + ".npy"))
        env = dict(os.environ, MPG_DIST_BACKEND="nccl", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29590 + k), MPG_MGPU_MODE=mode, MPG_MGPU_IC="s_zel")
        if force:
            env["MPG_FORCE_COLLECTIVES"] = force
        r = subprocess.run([sys.executable, script, out, "48"], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        res[mode] = np.load(out)
    one, got = res["single"], res["peano1"]
    assert_accel_parity(got[:, 0:3], one[:, 0:3])
    assert np.abs(got[:, 3:6] - one[:, 3:6]).max() <= 1e-11 * np.abs(one[:, 3:6]).mean()
