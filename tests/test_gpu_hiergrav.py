"""GPU parity of the hierarchical gravity level loop (mpg_dev_hierarchical_gravity_and_timesteps / _accelerations,
libgadget/timestep.c:239-599) with its CPU restatement (oracle/hiergrav_oracle.py): the same driver - the order of run.c:366-794
reduced to gravity - steps both through several sub-steps of a clustered particle set whose accelerations span many time bins."""
import numpy as np
import pytest

from oracle import oracle as O
from oracle import hiergrav_oracle as H
from test_gpu_gravity import setup_engine, G

pytestmark = pytest.mark.gpu

TIMEBINS = H.TIMEBINS


def dev(torch, a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


def update_kick_times(t):
    """timestep.c:214-235 on a dict of the DriftKickTimes fields"""
    if t["mintimebin"] == 0 and t["maxtimebin"] == 0:
        return
    for b in range(t["mintimebin"], TIMEBINS + 1):
        if H.is_timebin_active(b, t["Ti_Current"]):
            t["Ti_kick"][b] += H.dti_from_timebin(b) // 2
    for b in range(1, t["mintimebin"]):
        t["Ti_kick"][b] += H.dti_from_timebin(t["mintimebin"]) // 2


def to_struct(pkg, t):
    s = pkg.engine.DriftKickTimes()
    for k in ("mintimebin", "maxtimebin", "mingravtimebin", "Ti_Current", "PM_length", "PM_start", "PM_kick"):
        setattr(s, k, int(t[k]))
    for b in range(TIMEBINS + 1):
        s.Ti_kick[b] = int(t["Ti_kick"][b])
    return s


def from_struct(s, t):
    for k in ("mintimebin", "maxtimebin", "mingravtimebin", "Ti_Current", "PM_length", "PM_start", "PM_kick"):
        t[k] = int(getattr(s, k))
    t["Ti_kick"] = [int(s.Ti_kick[b]) for b in range(TIMEBINS + 1)]


def test_build_active_sublist(pkg, engine):
    import torch
    rng = np.random.RandomState(3)
    n = 100003
    tb = rng.randint(0, 9, n).astype(np.uint8)
    flags = (rng.random_sample(n) < 0.05).astype(np.uint8) * rng.randint(1, 4, n).astype(np.uint8)
    act = np.sort(rng.choice(n, n // 2, replace=False)).astype(np.int32)
    S = dict(tb_grav=tb, flags=flags)
    for a, maxbin, tic in ((act, 5, 48), (None, 8, 0), (act, 3, 7), (act[:0], 3, 16)):
        ref = H.build_active_sublist(S, a, maxbin, tic)
        out = torch.zeros(n, dtype=torch.int32, device="cuda")
        m = engine.dev_build_active_sublist(dev(torch, a) if a is not None and len(a) else (None if a is None else torch.zeros(0, dtype=torch.int32, device="cuda")),
                                            dev(torch, tb), dev(torch, flags), maxbin, tic, out)
        assert m == len(ref)
        assert np.array_equal(out[:m].cpu().numpy(), ref)


def test_hierarchical_gravity_steps_track_the_oracle(pkg, engine, orc):
    import torch
    n, nmesh = 14, 28
    pos, mass, box = pkg.ics.s_clust(n)
    N = len(pos)
    setup_engine(engine, box, n, nmesh, TreeUseBH=0)
    soft = 2.8 * (box / n) / 30.
    par = O.make_grav_params(box, nmesh, npart_cbrt=n, G=G)
    par.TreeUseBH = 0
    rho0 = 0.0
    # common initial state: GravPM and the accelerations of a full tree, from the engine
    f8 = dict(dtype=torch.float64, device="cuda")
    d_pos, d_mass = dev(torch, pos), dev(torch, mass)
    d_gpm, d_full, d_prev = torch.zeros(N, 3, **f8), torch.zeros(N, 3, **f8), torch.zeros(N, 3, **f8)
    engine.dev_bind_particles(d_pos, d_mass, box)
    engine.dev_gravpm_force(d_gpm, None)
    engine.dev_force_tree_build()
    engine.dev_grav_short_tree(d_full, prev_accel=d_prev, gravpm=d_gpm)
    engine.synchronize()
    rng = np.random.RandomState(11)
    S = dict(pos=pos.copy(), mass=mass.copy(), box=box, vel=rng.standard_normal((N, 3)) * 1e-2, gravpm=d_gpm.cpu().numpy().copy(),
             fulltree=d_full.cpu().numpy().copy(), tb_grav=np.zeros(N, np.uint8), flags=None, stored=None)
    d_vel, d_tb = dev(torch, S["vel"]), dev(torch, S["tb_grav"])
    # timeline and parameters: scale H so that the steps spread over bins around 2^36 .. 2^40 of an interval of 2^46
    loga = [np.log(0.1), np.log(0.5), np.log(1.0)]
    tl = H.Timeline(loga)
    atime, ErrTol, MinSize = 0.1, 0.025, 0.0
    dl1 = O.timestep_gravity_dloga(orc, S["fulltree"], S["gravpm"], atime, 1.0, ErrTol, soft)
    hubble = 6e-3 / np.median(dl1)
    dti_max_pm = 1 << 40
    ckick = 1e-3 / float(1 << 40)
    gravkick = lambda t0, t1: (t1 - t0) * ckick
    t_o = dict(mintimebin=0, maxtimebin=0, mingravtimebin=0, Ti_kick=[0] * (TIMEBINS + 1), Ti_Current=0, PM_length=0, PM_start=0, PM_kick=0)
    t_d = dict(t_o, Ti_kick=[0] * (TIMEBINS + 1))
    act = None
    levels_seen, spread = set(), 0
    for step in range(4):
        nag = N if act is None else len(act)
        d_act = None if act is None else dev(torch, act)
        stored_o = np.zeros((N, 3))
        d_stored = torch.zeros(N, 3, **f8)
        S["stored"] = stored_o
        A = engine._hier_arrays(d_vel, d_gpm, d_full, d_tb, stored_accel=d_stored)
        # second half of the previous step: accelerations of all active bins + kicks
        H.hierarchical_gravity_accelerations(orc, S, act, nag, t_o, par, G, gravkick)
        ts = to_struct(pkg, t_d)
        engine.dev_bind_particles(d_pos, d_mass, box)
        engine.dev_hierarchical_gravity_accelerations(A, d_act, nag, ts, rho0, gravkick)
        engine.synchronize()
        from_struct(ts, t_d)
        idx = np.arange(N) if act is None else act
        so, sd = stored_o[idx], d_stored.cpu().numpy()[idx]
        rel = np.sqrt(((so - sd) ** 2).sum(1)) / np.sqrt((so ** 2).sum(1))
        assert np.median(rel) <= 1e-12 and rel.max() <= 1e-8, (step, np.median(rel), rel.max())
        if act is None:
            assert np.abs(d_full.cpu().numpy() - S["fulltree"]).max() <= 1e-9 * np.abs(S["fulltree"]).max()
        update_kick_times(t_o)
        update_kick_times(t_d)
        # first half of the next step: new bins, accelerations per level, kicks
        if t_o["Ti_Current"] == t_o["PM_start"] + t_o["PM_length"]:
            t_o["PM_kick"] = t_d["PM_kick"] = t_o["Ti_Current"]          # (the two PM half kicks have been done, timestep.c:990)
        bad_o = H.hierarchical_gravity_and_timesteps(orc, S, act, nag, t_o, tl, ErrTol, MinSize, atime, hubble, dti_max_pm, par, G, soft, gravkick)
        ts = to_struct(pkg, t_d)
        bad_d = engine.dev_hierarchical_gravity_and_timesteps(A, d_act, nag, ts, loga, ErrTol, MinSize, atime, hubble, dti_max_pm, rho0, gravkick)
        engine.synchronize()
        from_struct(ts, t_d)
        assert bad_d == bad_o == 0
        assert t_d == t_o, (step, t_d, t_o)
        tb_d = d_tb.cpu().numpy()
        assert np.array_equal(tb_d, S["tb_grav"]), (step, np.nonzero(tb_d != S["tb_grav"])[0][:10])
        dv = np.abs(d_vel.cpu().numpy() - S["vel"]).max() / np.abs(S["vel"]).max()
        assert dv <= 1e-11, (step, dv)
        levels_seen |= set(np.unique(S["tb_grav"]).tolist())
        spread = max(spread, t_o["maxtimebin"] - t_o["mintimebin"])
        update_kick_times(t_o)
        update_kick_times(t_d)
        # advance to the next kick of the smallest occupied bin (find_next_kick, timestep.c:1324-1328), drift, new active set
        ti_next = t_o["Ti_Current"] + H.dti_from_timebin(t_o["mintimebin"])
        ddrift = (ti_next - t_o["Ti_Current"]) * ckick
        assert O.drift_all_particles(orc, S["pos"], S["vel"], ddrift, box) == 0
        engine.dev_drift_all_particles(d_pos, d_vel, ddrift, box)
        t_o["Ti_Current"] = t_d["Ti_Current"] = ti_next
        act = np.array([i for i in range(N) if H.is_timebin_active(int(S["tb_grav"][i]), ti_next)], np.int32)
        assert 0 < len(act) <= N
    assert len(levels_seen) >= 3, levels_seen                        # several levels were really exercised
    assert spread >= 2, spread                                        # ... within one call
