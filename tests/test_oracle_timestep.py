"""CPU checks of the time-integration oracle (oracle/timestep_oracle.c) against the behaviour drift.c / timestep.c specify."""
import numpy as np

from oracle import oracle as O


def make_set(n=4000, box=10.0, seed=11):
    rng = np.random.RandomState(seed)
    pos = rng.random_sample((n, 3)) * box
    pos[pos <= 0] = box
    vel = rng.standard_normal((n, 3)) * 3.0
    typ = (rng.random_sample(n) < 0.5).astype(np.uint8)          # 0 gas, 1 dark matter
    flags = np.zeros(n, np.uint8)
    flags[rng.choice(n, n // 50, replace=False)] = 1               # garbage
    flags[rng.choice(n, n // 50, replace=False)] |= 2              # swallowed
    hsml = 0.1 + rng.random_sample(n)
    dthsml = rng.standard_normal(n) * 0.05
    return pos, vel, typ, flags, hsml, dthsml, box


def test_drift_wraps_into_half_open_box(orc):
    pos, vel, typ, flags, hsml, dthsml, box = make_set()
    p0, h0 = pos.copy(), hsml.copy()
    rc = O.drift_all_particles(orc, pos, vel, 0.7, box, (0.3, -0.2, 0.05), type=typ, flags=flags, hsml=hsml, dthsml=dthsml)
    assert rc == 0
    assert pos.min() > 0 and pos.max() <= box                    # (0, BoxSize], drift.c:78-81
    live = (flags & 3) == 0
    exp = p0 + vel * 0.7 + np.array([0.3, -0.2, 0.05])
    d = np.abs(np.mod(pos[live] - exp[live] + box / 2, box) - box / 2)
    assert d.max() < 1e-12
    dead = ~live                                                  # only the random shift is applied (drift.c:21-30)
    d = np.abs(np.mod(pos[dead] - p0[dead] - np.array([0.3, -0.2, 0.05]) + box / 2, box) - box / 2)
    assert d.max() < 1e-12 and np.array_equal(hsml[dead], h0[dead])
    gas = live & (typ == 0)
    assert np.array_equal(hsml[gas], np.minimum(h0[gas] + dthsml[gas] * 0.7, box / 2))
    assert np.array_equal(hsml[live & (typ != 0)], h0[live & (typ != 0)])


def test_drift_reports_bad_hsml_and_positions(orc):
    pos, vel, typ, flags, hsml, dthsml, box = make_set(100)
    flags[:] = 0
    typ[0] = 0
    hsml[0], dthsml[0] = 0.1, -1.0
    assert O.drift_all_particles(orc, pos.copy(), vel, 1.0, box, type=typ, flags=flags, hsml=hsml.copy(), dthsml=dthsml) == 5
    vel[3, 1] = np.inf
    assert O.drift_all_particles(orc, pos.copy(), vel, 1.0, box) == 5


def test_kicks(orc):
    pos, vel, typ, flags, hsml, dthsml, box = make_set()
    n = len(vel)
    rng = np.random.RandomState(5)
    gpm, gacc, hacc = (rng.standard_normal((n, 3)) for _ in range(3))
    v0 = vel.copy()
    O.apply_pm_half_kick(orc, vel, gpm, 0.25, flags=flags)
    live = (flags & 3) == 0
    assert np.array_equal(vel[live], v0[live] + gpm[live] * 0.25) and np.array_equal(vel[~live], v0[~live])
    K = O.KickFactors()
    tbg = rng.randint(0, 4, n).astype(np.uint8)
    for b in range(4):
        K.gravkick[b], K.hydrokick[b], K.dt_entr[b] = 0.1 * (b + 1), 0.05 * (b + 1), 0.01 * (b + 1)
        K.bin_active[b] = b != 2                                   # bin 2 is not active: no gravity kick
    K.gravkick[2] = 0.0
    K.atime, K.MaxGasVel = 0.5, 8.0
    ent, dte = 1.0 + rng.random_sample(n), rng.standard_normal(n)
    v1, e0 = vel.copy(), ent.copy()
    assert O.apply_half_kick(orc, vel, gacc, K, type=typ, flags=flags, tb_grav=tbg, tb_hydro=tbg, hydroaccel=hacc, entropy=ent, dtentropy=dte) == 0
    gk = np.array([K.gravkick[b] for b in tbg])
    act = np.array([K.bin_active[b] for b in tbg]).astype(bool)
    dm = live & (typ != 0)
    assert np.array_equal(vel[dm & act], v1[dm & act] + gacc[dm & act] * gk[dm & act][:, None])
    assert np.array_equal(vel[dm & ~act], v1[dm & ~act]) and np.array_equal(vel[~live], v1[~live])
    gas = live & (typ == 0)
    assert np.array_equal(ent[gas], e0[gas] + dte[gas] * np.array([K.dt_entr[b] for b in tbg])[gas]) and np.array_equal(ent[~gas], e0[~gas])
    speed = np.sqrt((vel[gas] ** 2).sum(1)) / K.atime
    assert speed.max() <= K.MaxGasVel * (1 + 1e-12) and (speed > 0.99 * K.MaxGasVel).any()   # the limiter acted (timestep.c:1026-1030)
