"""The integer-timeline helpers of oracle/hiergrav_oracle.py against the known answers of the reference's own test
(libgadget/tests/test_timebinmgr.c:22-43, 72-92: sync points at a = 0.1, 0.2, 0.8, 1.0) and the bin arithmetic of timestep.c."""
import numpy as np

from oracle import hiergrav_oracle as H

TIMEBASE = H.TIMEBASE
outs = [0.1, 0.2, 0.8, 1.0]
logouts = [np.log(a) for a in outs]


def test_conversions_known_answers():
    tl = H.Timeline(logouts)
    assert abs(tl.loga_from_ti(0) - logouts[0]) < 1e-6
    assert abs(tl.loga_from_ti(TIMEBASE) - logouts[1]) < 1e-6
    assert abs(tl.loga_from_ti(TIMEBASE - 1) - (logouts[0] + (logouts[1] - logouts[0]) * (TIMEBASE - 1) / TIMEBASE)) < 1e-6
    assert abs(tl.loga_from_ti(TIMEBASE + 1) - (logouts[1] + (logouts[2] - logouts[1]) / TIMEBASE)) < 1e-6
    assert abs(tl.loga_from_ti(2 * TIMEBASE) - logouts[2]) < 1e-6
    assert tl.ti_from_loga(logouts[0])[0] == 0
    assert tl.ti_from_loga(logouts[1])[0] == TIMEBASE
    assert tl.ti_from_loga(logouts[2])[0] == 2 * TIMEBASE
    midpt = (logouts[2] + logouts[1]) / 2
    assert tl.ti_from_loga(midpt)[0] == TIMEBASE + TIMEBASE // 2
    assert abs(tl.loga_from_ti(TIMEBASE + TIMEBASE // 2) - midpt) < 1e-6
    assert tl.ti_from_loga(0.0)[0] == 3 * TIMEBASE                       # past the end
    assert abs(tl.loga_from_ti(int(tl.ti_from_loga(np.log(0.1))[0])) - np.log(0.1)) < 1e-6


def test_dloga_and_power_of_two_known_answers():
    tl = H.Timeline(logouts)
    ti = int(tl.ti_from_loga(np.log(0.55))[0])
    dl = tl.dloga_interval_ti(ti)
    assert abs(H.dti_from_timebin(0) * dl) < 1e-6                       # get_dloga_for_bin(0)
    assert abs(H.dti_from_timebin(H.TIMEBINS) * dl - (logouts[2] - logouts[1])) < 1e-6
    assert abs(H.dti_from_timebin(H.TIMEBINS - 2) * dl - (logouts[2] - logouts[1]) / 4) < 1e-6
    r = H.round_down_power_of_two(np.array([TIMEBASE, TIMEBASE + 1, TIMEBASE - 1, 0, 1, 5]))
    assert r.tolist() == [TIMEBASE, TIMEBASE, TIMEBASE // 2, 0, 1, 4]


def test_bins_and_activity():
    assert H.get_timestep_bin(np.array([0, 1, 2, 3, 4, 1 << 40, (1 << 40) + 5])).tolist() == [0, 0, 1, 1, 2, 40, 40]
    assert H.is_timebin_active(0, 12345) and H.is_timebin_active(7, 0) and H.is_timebin_active(3, 16) and not H.is_timebin_active(3, 12)
    tl = H.Timeline(logouts)
    # convert_timestep_to_ti: capped at dti_max, floored at MinSizeTimestep, 0 when dti_max == 0
    cur = 1 << 40
    d = H.convert_timestep_to_ti(np.array([1e-30, 1e-3, 10.0]), 1 << 42, cur, tl, 1e-5)
    iv = (logouts[1] - logouts[0]) / TIMEBASE
    assert abs(int(d[0]) - 1e-5 / iv) <= 2 and abs(int(d[1]) - 1e-3 / iv) <= 2 and int(d[2]) == 1 << 42
    assert H.convert_timestep_to_ti(np.array([1e-3]), 0, cur, tl, 0.0).tolist() == [0]
    S = dict(tb_grav=np.array([0, 1, 2, 3, 4], np.uint8), flags=np.array([0, 0, 1, 0, 0], np.uint8))
    assert H.build_active_sublist(S, None, 3, 8).tolist() == [0, 1, 3]
    assert H.build_active_sublist(S, np.array([4, 3, 1]), 4, 8).tolist() == [3, 1]   # order of the input list; bin 4 inactive at 8


def test_hydro_timestep_restatement_closed_forms():
    """get_timestep_hydro_dloga / get_timebin_from_dti / find_hydro_timesteps as restated in oracle/hiergrav_oracle.py (timestep.c:1076-1118,
    166-182, 617-733; the reference holds no test of them) against values worked out by hand."""
    import math
    from oracle import hiergrav_oracle as H
    a, hub, C = 0.5, 0.3, 0.15
    # gas, Courant: dt = 2 C a h / (a^(3 (1 - 5/3) / 2) vsig) = 2 C a^2 h / vsig
    dl, tt = H.get_timestep_hydro_dloga(0, 0.2, 0.0, 40.0, a, hub, C)
    assert tt == H.TI_COURANT and abs(dl / (2 * C * a * a * 0.2 / 40.0 * hub) - 1) < 1e-14
    # gas, a fast change of the smoothing length wins: dt = C a^2 |h / dh|
    dl, tt = H.get_timestep_hydro_dloga(0, 0.2, -50.0, 40.0, a, hub, C)
    assert tt == H.TI_HSML and abs(dl / (C * a * a * 0.2 / 50.0 * hub) - 1) < 1e-12
    # black hole: the bin above the shortest neighbour's; none without a neighbour bin; other types dt = 1
    tab = [float(b) for b in range(H.TIMEBINS + 1)]
    assert H.get_timestep_hydro_dloga(5, 0, 0, 0, a, hub, C, 7, tab) == (tab[8] / hub * hub, H.TI_NEIGH)
    assert H.get_timestep_hydro_dloga(5, 0, 0, 0, a, hub, C, 0, tab) == (hub, H.TI_ACCEL)
    assert H.get_timestep_hydro_dloga(1, 0, 0, 0, a, hub, C) == (hub, H.TI_ACCEL)
    # get_timebin_from_dti: a power of two rounded down; a longer step only onto an active bin
    assert H.get_timebin_from_dti((1 << 20) + 5, 25, 1 << 30) == 20
    assert H.get_timebin_from_dti(1 << 20, 12, 1 << 15) == 15 and H.get_timebin_from_dti(1 << 20, 12, 3 << 10) == 12
    assert H.get_timebin_from_dti(1 << 20, 12, 1 << 11) == 12      # (bin 11 is active, but never below the old bin)
    # find_hydro_timesteps: the hydro bin never exceeds the gravity bin, and the shortest bin is the new mintimebin
    import numpy as np
    tl = H.Timeline(np.log(np.array([0.25, 1.0])))
    S = dict(type=np.array([0, 0, 1], np.uint8), hsml=np.array([0.5, 0.5, 0.5]), dthsml=np.zeros(3), maxsignalvel=np.array([1e3, 1e-6, 1.0]),
             tb_grav=np.array([40, 33, 40], np.uint8), tb_hydro=np.array([30, 30, 30], np.uint8))
    t = dict(mintimebin=30, maxtimebin=41, mingravtimebin=33, Ti_Current=0, PM_length=1 << 41, PM_start=0, PM_kick=0, Ti_kick=[0] * 47)
    r = H.find_hydro_timesteps(S, None, t, tl, 1e-9, C, a, hub)
    dl0 = 2 * C * a * a * 0.5 / 1e3 * hub
    b0 = int(math.floor(math.log2(dl0 / (math.log(4.0) / (1 << H.TIMEBINS)))))
    assert S["tb_hydro"][0] == b0 and S["tb_hydro"][1] == 33 and S["tb_hydro"][2] == 30
    assert r["ntitype"] == [1, 1, 0, 0, 0] and r["mTimeBin"] == b0 and t["mintimebin"] == min(b0, 33)


def test_find_timesteps_restatement_closed_forms(orc):
    """find_timesteps as restated in oracle/hiergrav_oracle.py (timestep.c:739-849; the reference holds no test of it) against values worked
    out by hand: the gravity step dt = sqrt(2 eta a eps / |a_phys|), eps = FORCE_SOFTENING / 2.8, a_phys = (FullTreeGravAccel + GravPM) / a^2;
    gas takes the shorter of the two steps and BOTH bins follow; a PM step takes the handed-in length and is shrunk onto the longest tree
    step; an inactive old bin keeps its bins but still enters the extrema."""
    import math
    a, hub, C, eta, soft = 0.5, 0.3, 0.15, 0.025, 0.07
    tl = H.Timeline(np.log(np.array([0.25, 1.0])))
    iv = math.log(4.0) / (1 << H.TIMEBINS)
    g = np.array([[3.0, 0, 0], [0, 4.0e4, 0], [0, 0, 1e-3], [1.0, 1.0, 1.0]])
    pm = np.array([[1.0, 0, 0], [0, 0, 0], [0, 0, 0], [0, 0, 0]])
    S = dict(type=np.array([1, 1, 0, 0], np.uint8), gacc=g, gravpm=pm, hsml=np.full(4, 0.5), dthsml=np.zeros(4), maxsignalvel=np.array([1.0, 1.0, 1e3, 1e-9]),
             tb_grav=np.array([30, 30, 30, 30], np.uint8), tb_hydro=np.array([30, 30, 30, 30], np.uint8))
    t = dict(mintimebin=30, maxtimebin=41, mingravtimebin=30, Ti_Current=0, PM_length=1 << 40, PM_start=-(1 << 40), PM_kick=0, Ti_kick=[0] * 47)
    r = H.find_timesteps(orc, S, None, t, tl, eta, 1e-12, C, a, hub, soft, dti_max_pm=1 << 44)

    def bin_of(dloga):
        return int(math.floor(math.log2(dloga / iv)))
    dl_g = [math.sqrt(2 * eta * a * (soft / 2.8) / (x / (a * a))) * hub for x in (4.0, 4.0e4, 1e-3, math.sqrt(3.0))]
    dl_h2 = 2 * C * a * a * 0.5 / 1e3 * hub
    want = [bin_of(dl_g[0]), bin_of(dl_g[1]), min(bin_of(dl_g[2]), bin_of(dl_h2)), bin_of(dl_g[3])]
    want = [min(b, 44) for b in want]                                   # capped at the PM step
    assert S["tb_hydro"].tolist() == want and S["tb_grav"].tolist() == want
    assert bin_of(dl_h2) < bin_of(dl_g[2])                              # (the gas particle's bin did come from the Courant criterion)
    assert r["ntitype"] == [3, 1, 0, 0, 0] and r["isPM"] == 1
    assert r["mTimeBin"] == min(want) and r["maxTimeBin"] == max(want)
    assert t["mintimebin"] == min(want) and t["maxtimebin"] == max(want)
    assert t["PM_length"] == min(1 << 44, 1 << max(want)) and t["PM_start"] == 0
    # between PM steps, at a time where the old bin 30 is not active: the bins stay, the extrema still see the new bins
    S2 = dict(S, tb_grav=np.full(4, 30, np.uint8), tb_hydro=np.full(4, 30, np.uint8))
    t2 = dict(mintimebin=30, maxtimebin=41, mingravtimebin=30, Ti_Current=1 << 29, PM_length=1 << 40, PM_start=0, PM_kick=0, Ti_kick=[0] * 47)
    r2 = H.find_timesteps(orc, S2, None, t2, tl, eta, 1e-12, C, a, hub, soft)
    assert S2["tb_hydro"].tolist() == [30] * 4 and S2["tb_grav"].tolist() == [30] * 4 and r2["isPM"] == 0 and t2["PM_length"] == 1 << 40
    assert r2["mTimeBin"] <= 30 and t2["mintimebin"] == r2["mTimeBin"]
