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
