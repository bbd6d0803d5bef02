"""GPU parity of the time-integration kernels (csrc/timestep.hip) with the CPU oracle: BIT-IDENTICAL results (the kernels are
compiled without FMA contraction, like the reference's loops), error codes included."""
import numpy as np
import pytest

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
