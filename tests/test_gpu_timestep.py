"""GPU parity of the time-integration kernels (csrc/timestep.hip) with the CPU oracle: BIT-IDENTICAL results (the kernels are
compiled without FMA contraction, like the reference's loops), error codes included."""
import numpy as np
import pytest

from conftest import keep_artifacts_on_failure, run_ranks

from oracle import oracle as O
from test_oracle_timestep import make_set

pytestmark = pytest.mark.gpu


def dev(torch, a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("n", [1, 1000, 200003])
def test_drift_and_kicks_bit_exact(pkg, engine, orc, n):
    import torch
    pos, vel, typ, flags, hsml, dthsml, box = make_set(n, seed=n)
    rng = np.random.RandomState(9)
    gpm, gacc, hacc = (rng.standard_normal((n, 3)) for _ in range(3))
    ent, dte = 1.0 + rng.random_sample(n), rng.standard_normal(n)
    tbg = rng.randint(0, 4, n).astype(np.uint8)
    tbh = rng.randint(0, 4, n).astype(np.uint8)
    Ko, Kd = O.KickFactors(), pkg.KickFactors()
    for K in (Ko, Kd):
        for b in range(4):
            K.gravkick[b], K.hydrokick[b], K.dt_entr[b] = 0.1 * (b + 1) * (b != 2), 0.05 * (b + 1), 0.01 * (b + 1)
            K.bin_active[b] = b != 2
        K.atime, K.MaxGasVel = 0.5, 8.0
    d = {k: dev(torch, v) for k, v in dict(pos=pos, vel=vel, typ=typ, flags=flags, hsml=hsml, dthsml=dthsml, gpm=gpm, gacc=gacc, hacc=hacc,
                                            ent=ent, dte=dte, tbg=tbg, tbh=tbh).items()}
    act = np.sort(rng.choice(n, max(1, n // 3), replace=False)).astype(np.int32)
    # oracle sequence: PM half kick, half kick (subset), drift
    O.apply_pm_half_kick(orc, vel, gpm, 0.25, flags=flags)
    assert O.apply_half_kick(orc, vel, gacc, Ko, active=act, type=typ, flags=flags, tb_grav=tbg, tb_hydro=tbh, hydroaccel=hacc, entropy=ent,
                             dtentropy=dte) == 0
    assert O.drift_all_particles(orc, pos, vel, 0.37, box, (0.11, -0.07, 0.02), type=typ, flags=flags, hsml=hsml, dthsml=dthsml) == 0
    engine.dev_apply_pm_half_kick(d["vel"], d["gpm"], 0.25, flags=d["flags"])
    engine.dev_apply_half_kick(d["vel"], d["gacc"], Kd, active=dev(torch, act), type=d["typ"], flags=d["flags"], tb_grav=d["tbg"], tb_hydro=d["tbh"],
                               hydroaccel=d["hacc"], entropy=d["ent"], dtentropy=d["dte"])
    engine.dev_drift_all_particles(d["pos"], d["vel"], 0.37, box, (0.11, -0.07, 0.02), type=d["typ"], flags=d["flags"], hsml=d["hsml"],
                                   dthsml=d["dthsml"])
    engine.synchronize()
    assert np.array_equal(d["vel"].cpu().numpy(), vel)
    assert np.array_equal(d["ent"].cpu().numpy(), ent)
    assert np.array_equal(d["pos"].cpu().numpy(), pos)
    assert np.array_equal(d["hsml"].cpu().numpy(), hsml)


def test_drift_error_codes(pkg, engine):
    import torch
    pos, vel, typ, flags, hsml, dthsml, box = make_set(64)
    flags[:] = 0
    typ[0] = 0
    hsml[0], dthsml[0] = 0.1, -1.0
    with pytest.raises(pkg.EngineError, match="Hsml <= 0"):
        engine.dev_drift_all_particles(dev(torch, pos), dev(torch, vel), 1.0, box, type=dev(torch, typ), flags=dev(torch, flags),
                                       hsml=dev(torch, hsml), dthsml=dev(torch, dthsml))
    vel[3, 1] = np.inf
    with pytest.raises(pkg.EngineError, match="non-finite"):
        engine.dev_drift_all_particles(dev(torch, pos), dev(torch, vel), 1.0, box)


def test_three_force_kick_drift_steps_track_the_oracle(pkg, engine, orc):
    """End to end across steps, everything device-resident: (PM + tree build + short-range walk) -> PM kick + short-range kick ->
    drift, three times, against the same sequence on the CPU oracle.  The forces agree to ~1e-13 and the kicks / drifts are
    bit-exact, so the trajectories stay together far below the force accuracy."""
    import torch
    n, nmesh, G = 16, 32, 43.0071
    pos, mass, box = pkg.ics.s_zel(n)
    N = len(pos)
    dt = 2e-4 * box / np.sqrt(G)                                  # moves particles by a few per cent of the spacing per step
    rng = np.random.RandomState(4)
    vel = rng.standard_normal((N, 3)) * 0.02 * box / n / dt
    eng = engine
    eng.gravshort_fill_ntab(0, 1.5)
    eng.gravpm_init_periodic(box, 1.5, nmesh, G)
    eng.set_gravshort_treepar(TreeUseBH=0)
    eng.gravshort_set_softenings(box / n)
    f8 = dict(dtype=torch.float64, device="cuda")
    d_pos, d_mass, d_vel = dev(torch, pos), dev(torch, mass), dev(torch, vel)
    d_gpm, d_acc, d_prev, d_pot = torch.zeros(N, 3, **f8), torch.zeros(N, 3, **f8), torch.zeros(N, 3, **f8), torch.zeros(N, **f8)
    Kd, Ko = pkg.KickFactors(), O.KickFactors()
    for K in (Kd, Ko):
        K.gravkick[0], K.bin_active[0], K.atime, K.MaxGasVel = dt, 1, 1.0, 1e30
    par = O.make_grav_params(box, nmesh, npart_cbrt=n, G=G)
    par.TreeUseBH = 0
    o_pos, o_vel, o_prev = pos.copy(), vel.copy(), np.zeros((N, 3))
    for step in range(3):
        # device
        eng.dev_bind_particles(d_pos, d_mass, box)
        eng.dev_gravpm_force(d_gpm, None)
        eng.dev_force_tree_build()
        d_prev, d_acc = d_acc, d_prev
        eng.dev_grav_short_tree(d_acc, prev_accel=d_prev, gravpm=d_gpm)
        eng.dev_apply_pm_half_kick(d_vel, d_gpm, dt)
        eng.dev_apply_half_kick(d_vel, d_acc, Kd)
        eng.dev_drift_all_particles(d_pos, d_vel, dt, box)
        eng.synchronize()
        # oracle
        gpm, _ = O.gravpm_force(o_pos, mass, box, nmesh, 1.5, G)
        tr = orc.tree(o_pos, mass, box)
        acc, _, _, _ = tr.grav_short_tree(par, oldacc=np.sqrt(((o_prev + gpm) ** 2).sum(1)) / G)
        O.apply_pm_half_kick(orc, o_vel, gpm, dt)
        assert O.apply_half_kick(orc, o_vel, acc, Ko) == 0
        assert O.drift_all_particles(orc, o_pos, o_vel, dt, box) == 0
        o_prev = acc
        dp = np.abs(np.mod(d_pos.cpu().numpy() - o_pos + box / 2, box) - box / 2).max()
        dv = np.abs(d_vel.cpu().numpy() - o_vel).max() / np.abs(o_vel).max()
        assert dp <= 1e-11 * box / n and dv <= 1e-10, (step, dp, dv)
    moved = np.abs(np.mod(o_pos - pos + box / 2, box) - box / 2).max()
    assert moved > 0.02 * box / n                                   # the particles did move: the tree of step 3 is not the tree of step 1


@keep_artifacts_on_failure
def test_distributed_evolution_matches_one_gpu(tmp_path):
    """A complete distributed loop - ghost import, tree with the global top, slab PM, walk, kicks, drift, particle migration - for
    three steps on 2 and 4 ranks (sharing this GPU over gloo) against the same three steps on one GPU."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(name, nproc, mode, port):
        out = str(tmp_path / name)
        env = dict(os.environ, MPG_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MPG_MGPU_MODE=mode)
        script = os.path.join(root, "tools", "mgpu_evolve_check.py")
        cmd = [sys.executable, script, out, "24"] if nproc == 1 else \
              [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(port), script, out, "24"]
        run_ranks(cmd, env, out)
        return np.load(out)

    one = run("one.npy", 1, "single", 0)
    # the library's own choreography on the Peano-Hilbert domains: decomposition, exchange, force, kicks, drift, domain_maintain +
    # exchange (mpg_dist_*, csrc/dist.hip)
    for name, nproc, port, mode in (("p1.npy", 1, 0, "peano"), ("p2.npy", 2, 29597, "peano"), ("p4.npy", 4, 29598, "peano")):
        d = run(name, nproc, mode, port)
        dp = np.abs(d[:, 0:3] - one[:, 0:3])
        dp = np.minimum(dp, np.abs(dp - np.abs(one[:, 0:3]).max()))       # (a particle sitting on the periodic seam)
        assert np.median(dp) <= 1e-12 * np.abs(one[:, 0:3]).max(), name
        assert np.abs(d[:, 3:6] - one[:, 3:6]).max() <= 1e-9 * np.abs(one[:, 3:6]).max(), name
        assert np.abs(d[:, 6:9] - one[:, 6:9]).max() <= 2 * 0.002 * np.abs(one[:, 6:9]).mean(), name


def test_timestep_gravity_dloga(pkg, engine, orc):
    import torch
    rng = np.random.RandomState(2)
    n = 50001
    acc, gpm = rng.standard_normal((n, 3)) * 1e-3, rng.standard_normal((n, 3)) * 1e-4
    acc[7], gpm[7] = 0.0, 0.0                                     # zero acceleration: the 1e-60 guard (timestep.c:1054-1055)
    engine.set_gravshort_treepar()
    engine.gravshort_set_softenings(0.5)                          # GravitySoftening = 0.5 / 30; FORCE_SOFTENING = 2.8 times that
    out = torch.zeros(n, dtype=torch.float64, device="cuda")
    engine.dev_timestep_gravity_dloga(dev(torch, acc), dev(torch, gpm), 0.25, 0.7, 0.025, out)
    engine.synchronize()
    ref = O.timestep_gravity_dloga(orc, acc, gpm, 0.25, 0.7, 0.025, 2.8 * 0.5 / 30.)
    assert np.abs(out.cpu().numpy() / ref - 1).max() <= 4e-16


def test_hydro_timestep_criterion_and_bins(pkg, engine):
    """get_timestep_hydro_dloga (timestep.c:1076-1118) and find_hydro_timesteps (:617-733) on the device against the restatement
    (oracle/hiergrav_oracle.py; the reference holds no test of them: parity unpinned, see its header): dloga and the criterion per particle
    BIT-IDENTICAL (Courant, the Gadget-4 smoothing-length criterion, the black holes' neighbour limiter, dt = 1 for the rest), then the
    new hydro bins of an active list at two points of the timeline (bins never above the gravity bin, longer steps only onto active bins,
    garbage skipped), the counts per criterion, the smallest bin and the DriftKickTimes update - the first call as the first time step
    (set_bh_first_timestep)."""
    import torch
    from oracle import hiergrav_oracle as H
    from test_gpu_hiergrav import to_struct, from_struct
    rng = np.random.RandomState(11)
    n = 50021
    typ = rng.choice([0, 0, 0, 1, 1, 4, 5], n).astype(np.uint8)
    flags = ((rng.random_sample(n) < 0.03) * rng.randint(1, 4, n)).astype(np.uint8)
    hsml = 10 ** rng.uniform(-1.5, 1.0, n)
    dthsml = rng.standard_normal(n) * 10 ** rng.uniform(-3, 1, n)
    dthsml[rng.random_sample(n) < 0.1] = 0.0
    maxsig = 10 ** rng.uniform(0.5, 3.5, n)
    bhmin = rng.randint(0, 12, n).astype(np.uint8)
    atime, hubble, courant = 0.37, 0.21, 0.15
    sync = np.log(np.array([0.1, 0.25, 0.5, 1.0]))
    tl = H.Timeline(sync)
    d = {k: dev(torch, v) for k, v in dict(type=typ, flags=flags, hsml=hsml, dthsml=dthsml, maxsignalvel=maxsig, bh_mintimebin=bhmin).items()}
    # ---- the criterion alone
    Ti = (1 << H.TIMEBINS) + (3 << 30)
    logDTime = tl.dloga_interval_ti(Ti)
    table = [H.dti_from_timebin(b) * logDTime for b in range(H.TIMEBINS + 1)]
    dl = torch.zeros(n, dtype=torch.float64, device="cuda")
    tt = torch.zeros(n, dtype=torch.uint8, device="cuda")
    engine.dev_timestep_hydro_dloga(d["type"], d["hsml"], d["dthsml"], d["maxsignalvel"], atime, hubble, courant, dl, tt, bh_mintimebin=d["bh_mintimebin"],
                                    dloga_for_bin=table)
    engine.synchronize()
    ref = [H.get_timestep_hydro_dloga(int(typ[i]), float(hsml[i]), float(dthsml[i]), float(maxsig[i]), atime, hubble, courant, int(bhmin[i]), table)
           for i in range(n)]
    assert np.array_equal(dl.cpu().numpy(), np.array([r[0] for r in ref]))                  # bit-identical
    assert np.array_equal(tt.cpu().numpy(), np.array([r[1] for r in ref], np.uint8))
    assert set(np.unique(tt.cpu().numpy())) == {0, 1, 3, 4}
    # (without the limiter's arrays a black hole takes dt = 1)
    engine.dev_timestep_hydro_dloga(d["type"], d["hsml"], d["dthsml"], d["maxsignalvel"], atime, hubble, courant, dl, tt)
    engine.synchronize()
    assert np.all(dl.cpu().numpy()[typ == 5] == hubble) and np.all(tt.cpu().numpy()[typ == 5] == 0)
    # ---- the bins
    engine.dev_bind_particles(torch.zeros(n, 3, dtype=torch.float64, device="cuda") + 0.5, torch.ones(n, dtype=torch.float32, device="cuda"), 1.0)
    for case, (Ti_cur, first) in enumerate((((1 << H.TIMEBINS), True), ((1 << H.TIMEBINS) + (5 << 33), False))):
        tbg = rng.randint(30, 42, n).astype(np.uint8)
        tbh = np.minimum(rng.randint(28, 42, n), tbg).astype(np.uint8)
        act = np.sort(rng.choice(n, n // 2, replace=False)).astype(np.int32) if case else None
        times = dict(mintimebin=30, maxtimebin=41, mingravtimebin=31, Ti_Current=Ti_cur, PM_length=1 << 41, PM_start=Ti_cur - (1 << 41) * case, PM_kick=0,
                     Ti_kick=[0] * (H.TIMEBINS + 1))
        S = dict(type=typ, flags=flags, hsml=hsml, dthsml=dthsml, maxsignalvel=maxsig, tb_grav=tbg, tb_hydro=tbh.copy(), bh_mintimebin=bhmin)
        t_o = dict(times, Ti_kick=list(times["Ti_kick"]))
        ro = H.find_hydro_timesteps(S, act, t_o, tl, 1e-7, courant, atime, hubble, isFirstTimeStep=first)
        d_tbh = dev(torch, tbh)
        ts = to_struct(pkg, times)
        rg = engine.dev_find_hydro_timesteps(dict(d, tb_grav=dev(torch, tbg), tb_hydro=d_tbh), dev(torch, act), ts, sync, 1e-7, courant, atime, hubble,
                                             isFirstTimeStep=first)
        engine.synchronize()
        t_g = dict(times)
        from_struct(ts, t_g)
        assert np.array_equal(d_tbh.cpu().numpy(), S["tb_hydro"]), case
        assert rg == ro, (rg, ro)
        assert t_g["mintimebin"] == t_o["mintimebin"] and sum(rg["ntitype"]) > 0 and rg["ntitype"][0] > 0 and rg["ntitype"][1] > 0
        changed = d_tbh.cpu().numpy() != tbh
        dead = ((flags & 3) != 0) & (typ != 5)                      # garbage keeps its bin (set_bh_first_timestep sets every black hole's)
        assert changed.any() and not changed[dead].any()


def test_find_timesteps_bins(pkg, engine, orc):
    """find_timesteps (timestep.c:739-849, the step assignment of run.c:756 - a run without SplitGravityTimestepsOn) on the device against
    the restatement (oracle/hiergrav_oracle.py::find_timesteps; the reference holds no test of it: parity unpinned): the new bin of every
    active particle from the gravity criterion and, for gas / black holes, the hydro criteria where they are shorter - BOTH time bins equal
    to the restatement's, the counts per criterion, the smallest / largest bin, and the DriftKickTimes update; once as a PM step (new PM
    length handed in, shrunk onto the longest tree step) with every particle active, once between PM steps with an active list."""
    import torch
    from oracle import hiergrav_oracle as H
    from test_gpu_hiergrav import to_struct, from_struct
    rng = np.random.RandomState(17)
    n = 30011
    typ = rng.choice([0, 0, 0, 1, 1, 4, 5], n).astype(np.uint8)
    flags = ((rng.random_sample(n) < 0.03) * rng.randint(1, 4, n)).astype(np.uint8)
    hsml = 10 ** rng.uniform(-1.5, 1.0, n)
    dthsml = rng.standard_normal(n) * 10 ** rng.uniform(-3, 1, n)
    maxsig = 10 ** rng.uniform(0.5, 3.5, n)
    bhmin = rng.randint(0, 12, n).astype(np.uint8)
    gacc = rng.standard_normal((n, 3)) * 10 ** rng.uniform(-2, 4, (n, 1))     # (the zero-acceleration guard: test_timestep_gravity_dloga)
    gpm = rng.standard_normal((n, 3)) * 1e-1
    atime, hubble, courant, errtol = 0.37, 0.21, 0.15, 0.025
    sync = np.log(np.array([0.1, 0.25, 0.5, 1.0]))
    tl = H.Timeline(sync)
    engine.set_gravshort_treepar()
    engine.gravshort_set_softenings(0.5)
    soft = 2.8 * 0.5 / 30.
    engine.dev_bind_particles(torch.zeros(n, 3, dtype=torch.float64, device="cuda") + 0.5, torch.ones(n, dtype=torch.float32, device="cuda"), 1.0)
    d = {k: dev(torch, v) for k, v in dict(type=typ, flags=flags, hsml=hsml, dthsml=dthsml, maxsignalvel=maxsig, bh_mintimebin=bhmin).items()}
    seen = set()
    for case in range(2):
        Ti_cur = (1 << H.TIMEBINS) + ((5 << 33) if case else 0)
        tb = rng.randint(28, 42, n).astype(np.uint8)
        act = np.sort(rng.choice(n, n // 2, replace=False)).astype(np.int32) if case else None
        # case 0: a PM step (Ti_Current == PM_start + PM_length); case 1: inside a PM step
        times = dict(mintimebin=30, maxtimebin=41, mingravtimebin=31, Ti_Current=Ti_cur, PM_length=1 << 41, PM_start=Ti_cur - ((1 << 41) if not case else (5 << 33)),
                     PM_kick=Ti_cur if not case else Ti_cur - (5 << 33), Ti_kick=[0] * (H.TIMEBINS + 1))
        dti_pm = 1 << 43
        S = dict(type=typ, flags=flags, gacc=gacc, gravpm=gpm, hsml=hsml, dthsml=dthsml, maxsignalvel=maxsig, tb_grav=tb.copy(), tb_hydro=tb.copy(),
                 bh_mintimebin=bhmin)
        t_o = dict(times, Ti_kick=list(times["Ti_kick"]))
        ro = H.find_timesteps(orc, S, act, t_o, tl, errtol, 1e-7, courant, atime, hubble, soft, dti_max_pm=dti_pm)
        d_tbg, d_tbh = dev(torch, tb), dev(torch, tb)
        ts = to_struct(pkg, times)
        rg = engine.dev_find_timesteps(dict(d, tb_grav=d_tbg, tb_hydro=d_tbh), dev(torch, gacc), dev(torch, gpm), dev(torch, act), ts, sync, errtol, 1e-7,
                                       courant, atime, hubble, dti_max_pm=dti_pm)
        engine.synchronize()
        t_g = dict(times)
        from_struct(ts, t_g)
        assert np.array_equal(d_tbh.cpu().numpy(), S["tb_hydro"]) and np.array_equal(d_tbg.cpu().numpy(), S["tb_grav"]), case
        assert rg == ro, (rg, ro)
        assert rg["isPM"] == (0 if case else 1)
        for k in ("mintimebin", "maxtimebin", "PM_length", "PM_start"):
            assert t_g[k] == t_o[k], (case, k, t_g[k], t_o[k])
        if not case:                                                  # the PM step was shrunk onto the longest tree step
            assert t_g["PM_length"] == 1 << rg["maxTimeBin"] < dti_pm and t_g["PM_start"] == times["PM_kick"]
        changed = d_tbh.cpu().numpy() != tb
        assert changed.any() and not changed[(flags & 3) != 0].any()
        seen |= {k for k in range(5) if rg["ntitype"][k] > 0}
    assert {0, 1, 3, 4} <= seen          # gravity, Courant, the black holes' neighbour limiter and the smoothing-length criterion all decided bins


# ---- a resident gas run (round 5): gravity + density + hydro + time bins + kicks + drift, three steps, one upload and one final fetch
def _gas_run_setup(pkg, n=12):
    G = 43.0071
    pos, mass, typ8, box = pkg.ics.hydro_pair(n)
    N = len(pos)
    ph = 2 * np.pi * pos / box
    dt = 2e-4 * box / np.sqrt(G)
    vel = 0.03 * (box / n) / dt * np.stack([np.sin(ph[:, 1]) + np.cos(ph[:, 2]), np.sin(ph[:, 2]) + np.cos(ph[:, 0]), np.sin(ph[:, 0]) * np.cos(ph[:, 1])], 1)
    ent = 2e11 * (1.0 + 0.2 * np.sin(ph[:, 0]) * np.sin(ph[:, 1]))     # (sound speed ~ the flow's velocities: weak shocks)
    return dict(G=G, n=n, nmesh=2 * n, pos=pos, mass=mass, typ=typ8, box=box, N=N, dt=dt, vel=vel, ent=ent, atime=0.5, hubble=0.05, courant=0.15,
                bg=40, sync=np.log(np.array([0.25, 1.0])), pe=1)


def _gas_run_factors(s, KF, sph_times):
    """kick factors per time bin (a bin b steps 2^(b - bg) of dt) and the SphTimes of the predictions"""
    K = KF()
    f = [0.0] + [s["dt"] * 2.0 ** (b - s["bg"]) for b in range(1, 47)]
    for b in range(47):
        K.gravkick[b], K.hydrokick[b], K.dt_entr[b], K.bin_active[b] = 0.5 * f[b], 0.5 * f[b], 0.5 * f[b] * 1e-3, 1
    K.atime, K.MaxGasVel = s["atime"], 1e30
    t = sph_times(atime=s["atime"], hubble=s["hubble"], FgravkickB=0.25 * s["dt"], gravkicks=[0.25 * x for x in f], hydrokicks=[0.25 * x for x in f],
                  drifts=[0.5 * x for x in f], dloga_kick=[0.25e-3 * x for x in f], dloga_bin=[1e-3 * x for x in f])
    return K, t


def _gas_run_times(s, step):
    if s.get("pm_every_step"):          # (every step ends a PM step: Ti_Current == PM_start + PM_length)
        return dict(mintimebin=30, maxtimebin=41, mingravtimebin=s["bg"], Ti_Current=step << 41, PM_length=1 << 41, PM_start=(step - 1) << 41,
                    PM_kick=step << 41, Ti_kick=[0] * 47)
    return dict(mintimebin=30, maxtimebin=41, mingravtimebin=s["bg"], Ti_Current=step << 41, PM_length=1 << 41, PM_start=0, PM_kick=0, Ti_kick=[0] * 47)


def _oracle_gas_run(pkg, orc, s, nsteps, run_c_order=False):
    """run_c_order: the sequence of run.c without SplitGravityTimestepsOn (run.c:553-565, 754-794) - forces, second half kick of the step
    that ends, find_timesteps (both bins), first half kick of the step that begins, drift - instead of the round-5 sequence."""
    from oracle import hiergrav_oracle as H
    N, box, n, G = s["N"], s["box"], s["n"], s["G"]
    pos, vel, mass, typ = s["pos"].copy(), s["vel"].copy(), s["mass"], s["typ"].astype(np.int32)
    K, to = _gas_run_factors(s, O.KickFactors, O.sph_times)
    dp = O.DensityParams(1.0, 2.0, 2.0, 99999., 2, 0.006)
    O.sph_set_softening(orc, 2.8 * (box / n) / 30.)
    A = O.SphArrays(pos, mass, type=typ, hsml=np.full(N, 2.5 * box / n), vel=vel, entropy=s["ent"].copy())
    A.tb_grav[:] = s["bg"]
    A.tb_hydro[:] = 38
    par = O.make_grav_params(box, s["nmesh"], npart_cbrt=n, G=G)
    par.TreeUseBH = 0
    acc, gpm = np.zeros((N, 3)), np.zeros((N, 3))
    typ8 = s["typ"].astype(np.uint8)
    tl = H.Timeline(s["sync"])
    hist = []
    for step in range(nsteps):
        gpm, _ = O.gravpm_force(A.pos, mass, box, s["nmesh"], 1.5, G)
        tr = orc.tree(A.pos, mass, box)
        acc, _, _, _ = tr.grav_short_tree(par, oldacc=np.sqrt(((acc + gpm) ** 2).sum(1)) / G)
        A.gacc[:], A.gpm[:] = acc, gpm
        A.hydroacc_in[:], A.dtentropy_in[:] = A.hydroacc_out, A.dtentropy_out
        trg = orc.tree(A.pos, mass, box, type=typ, hsml=A.hsml, hydro_active=np.ones(N, np.uint8), mask=1, moments=False)
        O.sph_density(orc, trg, dp, A, to, DoEgyDensity=s["pe"])
        trg.calc_moments()
        O.sph_hydro_force(orc, trg, dp, O.HydroParams(s["pe"], 100.0, 0.75), A, to)
        S = dict(type=typ8, hsml=A.hsml, dthsml=A.dthsml, maxsignalvel=A.maxsignalvel, tb_grav=A.tb_grav, tb_hydro=A.tb_hydro)
        times = _gas_run_times(s, step)
        if run_c_order:
            kick = lambda: O.apply_half_kick(orc, A.vel, acc, K, type=typ8, tb_grav=A.tb_grav, tb_hydro=A.tb_hydro, hydroaccel=A.hydroacc_out,
                                             entropy=A.entropy, dtentropy=A.dtentropy_out)
            assert kick() == 0                                                       # run.c:558: the second half of the step that ends
            O.apply_pm_half_kick(orc, A.vel, gpm, 0.25 * s["dt"])                    # run.c:565
            S.update(gacc=acc, gravpm=gpm)
            r = H.find_timesteps(orc, S, None, times, tl, s["errtol_int"], 1e-9, s["courant"], s["atime"], s["hubble"], 2.8 * (box / n) / 30.,
                                 dti_max_pm=1 << 41)                                 # run.c:756
            hist.append((r, times["mintimebin"], np.bincount(A.tb_hydro, minlength=47)))
            assert kick() == 0                                                       # run.c:759: the first half of the step that begins
            O.apply_pm_half_kick(orc, A.vel, gpm, 0.25 * s["dt"])                    # run.c:794
            assert O.drift_all_particles(orc, A.pos, A.vel, s["dt"], box, type=typ8, hsml=A.hsml, dthsml=A.dthsml) == 0
            continue
        r = H.find_hydro_timesteps(S, None, times, tl, 1e-9, s["courant"], s["atime"], s["hubble"], isFirstTimeStep=(step == 0))
        hist.append((r, times["mintimebin"], np.bincount(A.tb_hydro[typ == 0], minlength=47)))
        O.apply_pm_half_kick(orc, A.vel, gpm, 0.5 * s["dt"])
        assert O.apply_half_kick(orc, A.vel, acc, K, type=typ8, tb_grav=A.tb_grav, tb_hydro=A.tb_hydro, hydroaccel=A.hydroacc_out, entropy=A.entropy,
                                 dtentropy=A.dtentropy_out) == 0
        assert O.drift_all_particles(orc, A.pos, A.vel, s["dt"], box, type=typ8, hsml=A.hsml, dthsml=A.dthsml) == 0
    return A, acc, gpm, hist


def test_three_resident_gas_steps_track_the_oracle(pkg, orc):
    """A gas run that stays resident (mpg_resident_begin + mpg_resident_sph_begin): per step gravpm_force, force_tree_full, grav_short_tree,
    density, hydro_force - the reference's own call sequence on the host view - then find_hydro_timesteps, apply_PM_half_kick,
    apply_half_kick and drift_all_particles on the device copies; three steps with ONE upload (the two begins) and ONE fetch (the two ends),
    against the same sequence on the CPU oracle: positions, velocities, entropies, smoothing lengths, the hydro time bins (spread over four
    bins by the Courant criterion) and the SPH fields.  The kick factors per bin and the SphTimes are the same synthetic ones on both sides."""
    from oracle import hiergrav_oracle as H
    from test_gpu_hiergrav import to_struct, from_struct
    s = _gas_run_setup(pkg)
    N, box, n = s["N"], s["box"], s["n"]
    nsteps = 3
    A, o_acc, o_gpm, hist = _oracle_gas_run(pkg, orc, s, nsteps)
    # ---- the engine, resident
    eng = pkg.Engine(0)
    eng.gravshort_fill_ntab(0, 1.5)
    eng.gravpm_init_periodic(box, 1.5, s["nmesh"], s["G"])
    eng.set_gravshort_treepar(TreeUseBH=0)
    eng.gravshort_set_softenings(box / n)
    eng.set_densitypar(1.0, 2.0, 2.0, 99999., pkg.engine.DENSITY_KERNEL_QUINTIC_SPLINE, 0.006)
    eng.set_hydropar(s["pe"], 100.0, 0.75)
    P = pkg.make_particles(s["pos"], s["mass"], type=s["typ"])
    P["Vel"] = s["vel"]
    z = lambda *sh: np.zeros(sh)
    a = dict(hsml=np.full(N, 2.5 * box / n), dthsml=z(N), vel=s["vel"].copy(), gacc=z(N, 3), gpm=z(N, 3), hydroacc_in=z(N, 3),
             tb_hydro=np.full(N, 38, np.uint8), tb_grav=np.full(N, s["bg"], np.uint8), entropy=s["ent"].copy(), dtentropy_in=z(N), density=z(N),
             egywtdensity=z(N), dhsmlegyfac=z(N), divvel=z(N), curlvel=z(N), hydroacc_out=z(N, 3), dtentropy_out=z(N), maxsignalvel=z(N))
    K, t = _gas_run_factors(s, pkg.KickFactors, lambda **kw: make_times_like(pkg, **kw))
    eng.resident_begin(P, box)                       # the one upload ...
    eng.resident_sph_begin(P, a)
    res = []
    for step in range(nsteps):
        eng.gravpm_force(P)
        eng.force_tree_full(P, box)
        eng.grav_short_tree(P)
        eng.density(P, box, a, t, DoEgyDensity=s["pe"])
        eng.hydro_force(P, a, t)
        ts = to_struct(pkg, _gas_run_times(s, step))
        res.append((eng.resident_find_hydro_timesteps(P, ts, s["sync"], 1e-9, s["courant"], s["atime"], s["hubble"], isFirstTimeStep=(step == 0)),
                    int(ts.mintimebin)))
        eng.resident_apply_pm_half_kick(P, 0.5 * s["dt"])
        eng.resident_apply_half_kick(P, K)
        eng.resident_drift_all_particles(P, s["dt"])
    assert np.array_equal(P["Pos"], s["pos"]) and np.array_equal(a["entropy"], s["ent"])      # (the host copies went stale: nothing came back yet)
    eng.resident_sph_end(a)                          # ... and the one fetch
    eng.resident_end(P)
    eng.close()
    # ---- against the oracle
    for (rg, mb_g), (ro, mb_o, _) in zip(res, hist):
        assert rg == ro and mb_g == mb_o, (rg, ro)
    sp = box / n
    gas = s["typ"] == 0
    dpos = np.abs(np.mod(P["Pos"] - A.pos + box / 2, box) - box / 2).max()
    assert dpos <= 1e-9 * sp, dpos
    assert np.abs(P["Vel"] - A.vel).max() <= 1e-9 * np.abs(A.vel).max()
    assert np.abs(P["FullTreeGravAccel"] - o_acc).max() <= 1e-10 * np.abs(o_acc).max() and np.abs(P["GravPM"] - o_gpm).max() <= 1e-10 * np.abs(o_gpm).max()
    assert np.array_equal(a["tb_hydro"], A.tb_hydro) and len(np.unique(a["tb_hydro"][gas])) >= 3
    for k in ("hsml", "dthsml", "entropy", "density", "egywtdensity", "divvel", "curlvel", "dtentropy_out", "maxsignalvel"):
        g, o = a[k][gas], getattr(A, k)[gas]
        assert np.abs(g - o).max() <= 1e-8 * np.abs(o).max(), (k, np.abs(g - o).max() / np.abs(o).max())
    assert np.abs(a["hydroacc_out"][gas] - A.hydroacc_out[gas]).max() <= 1e-8 * np.abs(A.hydroacc_out[gas]).max()
    moved = np.abs(np.mod(A.pos - s["pos"] + box / 2, box) - box / 2).max()
    assert moved > 0.05 * sp and np.abs(A.entropy[gas] / s["ent"][gas] - 1).max() > 1e-6        # the run did move particles and entropies


def test_resident_gas_steps_in_run_c_order_without_split_gravity(pkg, orc):
    """the resident stretch must carry a set of integrator functions that one branch of run.c actually calls.  This
    drives the branch WITHOUT SplitGravityTimestepsOn in run.c's own order (run.c:522-565, 754-794) on the C-ABI the shim forwards to - per
    step gravpm_force, force_tree_full, grav_short_tree, density, hydro_force, apply_half_kick (second half), apply_PM_half_kick,
    find_timesteps (gravity + hydro criteria, BOTH time bins, every step a PM step), apply_half_kick (first half), apply_PM_half_kick,
    drift_all_particles - three steps, ONE upload and ONE fetch, against the same sequence on the CPU oracle.  The time bins are fetched after
    every find_timesteps (mpg_resident_fetch_timebins: what build_active_particles reads on the host) and must equal the oracle's there too."""
    from oracle import hiergrav_oracle as H
    from test_gpu_hiergrav import to_struct, from_struct
    s = _gas_run_setup(pkg)
    s["errtol_int"] = 0.05
    s["pm_every_step"] = True
    N, box, n = s["N"], s["box"], s["n"]
    nsteps = 3
    A, o_acc, o_gpm, hist = _oracle_gas_run(pkg, orc, s, nsteps, run_c_order=True)
    eng = pkg.Engine(0)
    eng.gravshort_fill_ntab(0, 1.5)
    eng.gravpm_init_periodic(box, 1.5, s["nmesh"], s["G"])
    eng.set_gravshort_treepar(TreeUseBH=0)
    eng.gravshort_set_softenings(box / n)
    eng.set_densitypar(1.0, 2.0, 2.0, 99999., pkg.engine.DENSITY_KERNEL_QUINTIC_SPLINE, 0.006)
    eng.set_hydropar(s["pe"], 100.0, 0.75)
    P = pkg.make_particles(s["pos"], s["mass"], type=s["typ"])
    P["Vel"] = s["vel"]
    z = lambda *sh: np.zeros(sh)
    a = dict(hsml=np.full(N, 2.5 * box / n), dthsml=z(N), vel=s["vel"].copy(), gacc=z(N, 3), gpm=z(N, 3), hydroacc_in=z(N, 3),
             tb_hydro=np.full(N, 38, np.uint8), tb_grav=np.full(N, s["bg"], np.uint8), entropy=s["ent"].copy(), dtentropy_in=z(N), density=z(N),
             egywtdensity=z(N), dhsmlegyfac=z(N), divvel=z(N), curlvel=z(N), hydroacc_out=z(N, 3), dtentropy_out=z(N), maxsignalvel=z(N))
    K, t = _gas_run_factors(s, pkg.KickFactors, lambda **kw: make_times_like(pkg, **kw))
    eng.resident_begin(P, box)
    eng.resident_sph_begin(P, a)
    res, bins = [], []
    for step in range(nsteps):
        eng.gravpm_force(P)
        eng.force_tree_full(P, box)
        eng.grav_short_tree(P)
        eng.density(P, box, a, t, DoEgyDensity=s["pe"])
        eng.hydro_force(P, a, t)
        eng.resident_apply_half_kick(P, K)
        eng.resident_apply_pm_half_kick(P, 0.25 * s["dt"])
        ts = to_struct(pkg, _gas_run_times(s, step))
        res.append((eng.resident_find_timesteps(P, ts, s["sync"], s["errtol_int"], 1e-9, s["courant"], s["atime"], s["hubble"], dti_max_pm=1 << 41),
                    int(ts.mintimebin)))
        bins.append(eng.resident_fetch_timebins(N))
        eng.resident_apply_half_kick(P, K)
        eng.resident_apply_pm_half_kick(P, 0.25 * s["dt"])
        eng.resident_drift_all_particles(P, s["dt"])
    eng.resident_sph_end(a)
    eng.resident_end(P)
    eng.close()
    for (rg, mb_g), (ro, mb_o, hb), (tbh, tbg) in zip(res, hist, bins):
        assert rg == ro and mb_g == mb_o, (rg, ro)
        assert np.array_equal(np.bincount(tbh, minlength=47), hb) and np.array_equal(tbh, tbg)   # find_timesteps sets both bins alike
    sp = box / n
    gas = s["typ"] == 0
    dpos = np.abs(np.mod(P["Pos"] - A.pos + box / 2, box) - box / 2).max()
    assert dpos <= 1e-9 * sp, dpos
    assert np.abs(P["Vel"] - A.vel).max() <= 1e-9 * np.abs(A.vel).max()
    assert np.array_equal(a["tb_hydro"], A.tb_hydro) and np.array_equal(a["tb_grav"], A.tb_grav)
    assert len(np.unique(a["tb_hydro"])) >= 3 and len(np.unique(a["tb_grav"][~gas])) >= 1
    assert res[-1][0]["ntitype"][0] > 0 and res[-1][0]["ntitype"][1] > 0                 # gravity and Courant both set bins
    for k in ("hsml", "entropy", "density", "dtentropy_out", "maxsignalvel"):
        g, o = a[k][gas], getattr(A, k)[gas]
        assert np.abs(g - o).max() <= 1e-8 * np.abs(o).max(), (k, np.abs(g - o).max() / np.abs(o).max())


def make_times_like(pkg, atime=1.0, hubble=0.1, **kw):
    t = pkg.SphTimes()
    t.atime, t.hubble = atime, hubble
    for k, v in kw.items():
        if isinstance(v, (int, float)):
            setattr(t, k, v)
        else:
            arr = getattr(t, k)
            for i, x in enumerate(v):
                arr[i] = x
    return t
