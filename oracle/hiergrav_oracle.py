"""CPU restatement of the hierarchical gravity level loop -- TEST INFRASTRUCTURE ONLY (see oracle/README.md): only tests/ may import it.

Follows libgadget/timestep.c:
  apply_hierarchical_grav_kick            :238-278
  grav_short_tree_build_tree              :281-291
  hierarchical_gravity_and_timesteps      :293-490
  hierarchical_gravity_accelerations      :495-599
  get_timestep_gravity_dloga              :1045-1074   (via oracle.timestep_gravity_dloga, C)
  convert_timestep_to_ti                  :1155-1175
  get_timestep_bin                        :1301-1315
  build_active_sublist                    :1435-1478
  is_timebin_active                       :143-150
and libgadget/timebinmgr.c: Dloga_interval_ti :372-385, loga_from_ti :388-398, ti_from_loga :400-417, dti_from_dloga :434-440,
round_down_power_of_two :449-462; timebinmgr.h: dti_from_timebin :47-50.

Trees and walks are the oracle's (oracle.Oracle.tree on the active sub-set, OracleTree.grav_short_tree).  The orchestration has no
known-answer test in the reference (there is no test_timestep.c): parity of this layer is pinned by this restatement only.

State is a dict of numpy arrays in particle order: pos[N,3], mass[N] (f32), vel[N,3], gravpm[N,3], fulltree[N,3] (FullTreeGravAccel),
tb_grav[N] (u8), flags[N] (u8, optional), stored[N,3] or None (StoredGravAccel.GravAccel).
`times` is a dict with the DriftKickTimes fields (timestep.h:10-27)."""
import numpy as np

from . import oracle as O

TIMEBINS = 46
TIMEBASE = 1 << TIMEBINS


def dti_from_timebin(b):
    return (1 << int(b)) if b > 0 else 0


def is_timebin_active(i, current):
    if i <= 0 or current <= 0:
        return True
    return current % dti_from_timebin(i) == 0


class Timeline:
    """SyncPoints[].loga"""

    def __init__(self, loga):
        self.loga = [float(x) for x in loga]
        self.n = len(self.loga)

    def dloga_interval_ti(self, ti):
        lastsnap = ti >> TIMEBINS
        if lastsnap >= self.n - 1:
            return 0.0
        return (self.loga[lastsnap + 1] - self.loga[lastsnap]) / float(TIMEBASE)

    def loga_from_ti(self, ti):
        lastsnap = ti >> TIMEBINS
        dti = ti & (TIMEBASE - 1)
        return self.loga[lastsnap] + dti * self.dloga_interval_ti(ti)

    def ti_from_loga(self, loga):
        """vectorised over `loga`"""
        loga = np.atleast_1d(np.asarray(loga, np.float64))
        out = np.zeros(loga.shape, np.int64)
        for k, x in enumerate(loga):
            i = 1
            while i < self.n - 1:
                if self.loga[i] > x:
                    break
                i += 1
            logDTime = np.float64(self.loga[i] - self.loga[i - 1]) / np.float64(TIMEBASE)
            ti = np.float64((i - 1) << TIMEBINS)
            out[k] = np.int64(np.trunc(ti + np.float64(x - self.loga[i - 1]) / logDTime))
        return out


def convert_timestep_to_ti(dloga, dti_max, Ti_Current, tl, MinSizeTimestep):
    if dti_max == 0:
        return np.zeros(len(dloga), np.int64)
    dloga = np.where(dloga < MinSizeTimestep, MinSizeTimestep, dloga)
    loga_cur = tl.loga_from_ti(Ti_Current)
    ti = tl.ti_from_loga(loga_cur)[0]
    dti = tl.ti_from_loga(dloga + loga_cur) - ti
    return np.where((dti > dti_max) | (dti < 0), dti_max, dti).astype(np.int64)


def round_down_power_of_two(dti):
    out = np.zeros(len(dti), np.int64)
    for k, d in enumerate(dti):
        ti_min = TIMEBASE
        d = int(d)
        sign = 1
        if d < 0:
            d, sign = -d, -1
        while ti_min > d:
            ti_min >>= 1
        out[k] = ti_min * sign
    return out


def get_timestep_bin(dti):
    out = np.zeros(len(dti), np.int64)
    for k, d in enumerate(dti):
        d = int(d)
        if d <= 1:
            continue
        b = -1
        while d:
            b += 1
            d >>= 1
        out[k] = b
    return out


def build_active_sublist(S, act, maxtimebin, Ti_Current):
    idx = np.arange(len(S["tb_grav"]), dtype=np.int64) if act is None else np.asarray(act, np.int64)
    keep = []
    for pi in idx:
        b = int(S["tb_grav"][pi])
        if S.get("flags") is not None and (S["flags"][pi] & 3):
            continue
        if b > maxtimebin:
            continue
        if not is_timebin_active(b, Ti_Current):
            continue
        keep.append(pi)
    return np.array(keep, np.int32)


def _listed(S, act):
    idx = np.arange(len(S["tb_grav"]), dtype=np.int64) if act is None else np.asarray(act, np.int64)
    if S.get("flags") is not None:
        idx = idx[(S["flags"][idx] & 3) == 0]
    return idx


def apply_hierarchical_grav_kick(S, act, times, accel, ti, largest_active, gravkick):
    dti = dti_from_timebin(ti)
    gk = gravkick(times["Ti_kick"][ti], times["Ti_kick"][ti] + dti // 2)
    if ti < largest_active:
        lowerdti = dti_from_timebin(ti + 1)
        gk -= gravkick(times["Ti_kick"][ti + 1], times["Ti_kick"][ti + 1] + lowerdti // 2)
    idx = _listed(S, act)
    a = S["fulltree"] if accel is None else accel
    S["vel"][idx] += a[idx] * gk


def grav_short_tree_build_tree(orc, S, act, accel_store, par, G):
    """Tree of the listed particles (all if act is None), walk for them; results into accel_store (if given) and, for a tree of all
    particles, into FullTreeGravAccel (grav_short_postprocess, gravshort.h:47-67)."""
    N = len(S["tb_grav"])
    idx = np.arange(N) if act is None else np.asarray(act, np.int64)
    if len(idx) == 0:
        return
    tr = orc.tree(S["pos"][idx], S["mass"][idx], S["box"])
    old = np.sqrt(((S["fulltree"] + S["gravpm"]) ** 2).sum(1)) / G          # grav_get_abs_accel, gravshort.h:70-80
    a, _, _, _ = tr.grav_short_tree(par, oldacc=old[idx])
    tr.free()
    if accel_store is not None:
        accel_store[idx] = a
    if act is None:
        S["fulltree"][idx] = a
    if par.TreeUseBH > 1:                                                      # gravshort-tree.c:148-151
        par.TreeUseBH = 0


def _largest_active(times):
    largest, ti = TIMEBINS, TIMEBINS
    while ti >= 0:
        if is_timebin_active(ti, times["Ti_Current"]) and dti_from_timebin(ti) <= times["PM_length"]:
            largest = ti
            break
        ti -= 1
    return largest, ti


def _dloga(orc, S, accel, idx, atime, hubble, ErrTol, soft):
    return O.timestep_gravity_dloga(orc, np.ascontiguousarray(accel[idx]), np.ascontiguousarray(S["gravpm"][idx]), atime, hubble, ErrTol, soft)


def hierarchical_gravity_and_timesteps(orc, S, act, num_active_gravity, times, tl, ErrTolIntAccuracy, MinSizeTimestep, atime, hubble,
                                       dti_max_pm, par, G, soft, gravkick):
    N = len(S["tb_grav"])
    nact = N if act is None else len(act)
    nag = nact if act is None else num_active_gravity
    assert times["Ti_Current"] <= times["PM_start"] + times["PM_length"]
    isPM = times["Ti_Current"] == times["PM_start"] + times["PM_length"]
    dti_max = times["PM_length"]
    if isPM:
        dti_max = dti_max_pm
        times["PM_length"] = dti_max
        times["PM_start"] = times["PM_kick"]
    largest_active, _ = _largest_active(times)
    if nag == nact or isPM:
        subact = act
    else:
        subact = build_active_sublist(S, act, largest_active, times["Ti_Current"])
    idx = _listed(S, subact)
    counts = np.zeros(TIMEBINS + 1, np.int64)
    bad = 0
    topacc = S["stored"] if S.get("stored") is not None else S["fulltree"]
    if len(idx):
        dl = _dloga(orc, S, topacc, idx, atime, hubble, ErrTolIntAccuracy, soft)
        dti = convert_timestep_to_ti(dl, dti_max, times["Ti_Current"], tl, MinSizeTimestep)
        dti = round_down_power_of_two(dti)
        bad += int(((dti <= 1) | (dti > TIMEBASE)).sum())
        b = np.minimum(get_timestep_bin(dti), largest_active)
        np.add.at(counts, b, 1)
        S["tb_grav"][idx] = b.astype(np.uint8)
    for ti in range(largest_active, 0, -1):
        if counts[ti] > 0:
            largest_active = ti
            break
    push_down_bin = largest_active
    if isPM:
        for ti in range(largest_active, 0, -1):
            if counts[ti] // 3 > counts[ti - 1]:
                break
            push_down_bin = ti - 1
            counts[ti - 1] += counts[ti]
    assert push_down_bin != 0
    if push_down_bin != largest_active:
        ii = np.arange(N) if subact is None else np.asarray(subact, np.int64)
        S["tb_grav"][ii] = np.minimum(S["tb_grav"][ii], push_down_bin)
        largest_active = push_down_bin
    times["maxtimebin"] = largest_active
    apply_hierarchical_grav_kick(S, subact, times, S.get("stored"), largest_active, largest_active, gravkick)
    lastact = subact
    for ti in range(largest_active - 1, 0, -1):
        sub = build_active_sublist(S, lastact, ti, times["Ti_Current"])
        if len(sub) == 0:
            times["mingravtimebin"] = ti + 1
            break
        grav = np.zeros((N, 3))
        grav_short_tree_build_tree(orc, S, sub, grav, par, G)
        idx = _listed(S, sub)
        dl = _dloga(orc, S, grav, idx, atime, hubble, ErrTolIntAccuracy, soft)
        dti = convert_timestep_to_ti(dl, dti_max, times["Ti_Current"], tl, MinSizeTimestep)
        down = dti < dti_from_timebin(ti)
        S["tb_grav"][idx[down]] = ti - 1
        if ti == 1:
            bad += int(down.sum())
        apply_hierarchical_grav_kick(S, sub, times, grav, ti, largest_active, gravkick)
        lastact = sub
    times["mintimebin"] = times["mingravtimebin"]
    return bad


def hierarchical_gravity_accelerations(orc, S, act, num_active_gravity, times, par, G, gravkick):
    N = len(S["tb_grav"])
    nact = N if act is None else len(act)
    nag = nact if act is None else num_active_gravity
    largest_active, ti = _largest_active(times)
    if nag == nact:
        lastact, last_grav = act, nag
    else:
        lastact = build_active_sublist(S, act, ti, times["Ti_Current"])
        last_grav = len(lastact)
    grav_short_tree_build_tree(orc, S, lastact, S.get("stored"), par, G)
    apply_hierarchical_grav_kick(S, lastact, times, S.get("stored"), ti, largest_active, gravkick)
    grav = None
    for ti in range(largest_active - 1, times["mingravtimebin"] - 1, -1):
        sub = build_active_sublist(S, lastact, ti, times["Ti_Current"])
        if len(sub) != last_grav:
            grav = np.zeros((N, 3))
            grav_short_tree_build_tree(orc, S, sub, grav, par, G)
        tmp = grav if grav is not None else S.get("stored")
        apply_hierarchical_grav_kick(S, sub, times, tmp, ti, largest_active, gravkick)
        lastact, last_grav = sub, len(sub)


# ---- the hydro time step (round 5): get_timestep_hydro_dloga, timestep.c:1076-1118; get_timebin_from_dti :166-182;
# find_hydro_timesteps :617-733 (without the dynamic-friction bins of the black holes, :676-695).  No reference test covers them: parity
# unpinned, pinned by this restatement (plain IEEE arithmetic in the reference's order of operations: the device results are compared bit
# for bit).
GAMMA = 5.0 / 3.0       # physconst.h:35
TI_ACCEL, TI_COURANT, TI_ACCRETE, TI_NEIGH, TI_HSML = 0, 1, 2, 3, 4     # enum TimeStepType, timestep.c:89-96


def get_timestep_hydro_dloga(ptype, hsml, dthsml, maxsignalvel, atime, hubble, CourantFac, bh_mintimebin=None, dloga_for_bin=None):
    """one particle; returns (dloga, titype)"""
    import math
    dt, titype = 1.0, TI_ACCEL
    if ptype == 0:
        fac3 = math.pow(atime, 3 * (1 - GAMMA) / 2.0)
        dt_courant = 2 * CourantFac * atime * hsml / (fac3 * maxsignalvel)
        dt, titype = dt_courant, TI_COURANT
        dt_hsml = CourantFac * atime * atime * abs(hsml / (dthsml + 1e-20))
        if dt_hsml < dt:
            dt, titype = dt_hsml, TI_HSML
    elif ptype == 5 and bh_mintimebin is not None and dloga_for_bin is not None:
        if bh_mintimebin > 0 and bh_mintimebin + 1 < TIMEBINS:
            dt, titype = dloga_for_bin[bh_mintimebin + 1] / hubble, TI_NEIGH
    return dt * hubble, titype


def get_timebin_from_dti(dti, binold, Ti_Current):
    dti = int(round_down_power_of_two(np.array([dti]))[0])
    b = int(get_timestep_bin(np.array([dti]))[0])
    if b > binold:
        while (not is_timebin_active(b, Ti_Current)) and b > binold and b > 1:
            b -= 1
    return b


def find_hydro_timesteps(S, act, times, tl, MinSizeTimestep, CourantFac, atime, hubble, isFirstTimeStep=False):
    """S: dict with type, flags (optional), hsml, dthsml, maxsignalvel, tb_grav, tb_hydro (updated in place), bh_mintimebin (optional).
    Returns dict(mTimeBin (before the tail), ntitype[5], badstepsizecount, badtimebins) and updates times['mintimebin'] as the tail does
    (one rank: no all-reduce)."""
    dti_max = times["PM_length"]
    Ti = times["Ti_Current"]
    ntitype = [0] * 5
    bad, badbins = 0, 0
    mTimeBin = TIMEBINS
    logDTime = tl.dloga_interval_ti(Ti)
    dloga_for_bin = [dti_from_timebin(b) * logDTime for b in range(TIMEBINS + 1)]
    flags = S.get("flags")
    bhmin = S.get("bh_mintimebin")
    for i in _listed(S, act):
        if flags is not None and (flags[i] & 3):
            continue
        ty = int(S["type"][i]) & 7
        if ty != 0 and ty != 5:
            continue
        dloga, titype = get_timestep_hydro_dloga(ty, float(S["hsml"][i]), float(S["dthsml"][i]), float(S["maxsignalvel"][i]), atime, hubble,
                                                 CourantFac, None if bhmin is None else int(bhmin[i]), dloga_for_bin)
        dti = int(convert_timestep_to_ti(np.array([dloga]), dti_max, Ti, tl, MinSizeTimestep)[0])
        if dti <= 1 or dti > TIMEBASE:
            badbins += 1
        b = get_timebin_from_dti(dti, int(S["tb_hydro"][i]), Ti)
        if b > int(S["tb_grav"][i]):
            b, titype = int(S["tb_grav"][i]), TI_ACCEL
        if b < 1:
            bad += 1
        ntitype[titype] += 1
        if is_timebin_active(int(S["tb_hydro"][i]), Ti) and is_timebin_active(b, Ti):
            S["tb_hydro"][i] = b
        mTimeBin = min(mTimeBin, b)
    res = dict(mTimeBin=mTimeBin, ntitype=ntitype, badstepsizecount=bad, badtimebins=badbins)
    if not is_timebin_active(mTimeBin, Ti):
        mTimeBin = times["mintimebin"]
        if is_timebin_active(mTimeBin + 1, Ti):
            mTimeBin += 1
    if isFirstTimeStep:
        S["tb_hydro"][(S["type"] & 7) == 5] = mTimeBin
    times["mintimebin"] = mTimeBin
    if times["mintimebin"] > times["mingravtimebin"] and times["mingravtimebin"] > 0:
        times["mintimebin"] = times["mingravtimebin"]
    return res


def find_timesteps(orc, S, act, times, tl, ErrTolIntAccuracy, MinSizeTimestep, CourantFac, atime, hubble, soft, dti_max_pm=0):
    """find_timesteps, timestep.c:739-849 (the step assignment of a run without SplitGravityTimestepsOn, run.c:756), without
    ForceEqualTimesteps (:759-761, :778-779) and set_bh_first_timestep (:844-845).  S: dict with type, flags (optional), gacc
    (FullTreeGravAccel), gravpm, hsml, dthsml, maxsignalvel, tb_grav and tb_hydro (both updated in place), bh_mintimebin (optional).
    Updates times['PM_length' / 'PM_start'] on a PM step and times['mintimebin' / 'maxtimebin'] (one rank: no all-reduce).  No reference test
    covers this function: parity unpinned, as for find_hydro_timesteps above."""
    Ti = times["Ti_Current"]
    assert Ti <= times["PM_start"] + times["PM_length"], "Passed end of PM step!"         # is_PM_timestep :153-159
    isPM = Ti == times["PM_start"] + times["PM_length"]
    dti_max = times["PM_length"]
    if isPM:                                                                                # :748-755
        dti_max = dti_max_pm
        times["PM_length"] = dti_max
        times["PM_start"] = times["PM_kick"]
    ntitype = [0] * 5
    bad, badbins = 0, 0
    mTimeBin, maxTimeBin = TIMEBINS, 0
    logDTime = tl.dloga_interval_ti(Ti)
    dloga_for_bin = [dti_from_timebin(b) * logDTime for b in range(TIMEBINS + 1)]
    flags = S.get("flags")
    bhmin = S.get("bh_mintimebin")
    idx = np.asarray(list(_listed(S, act)), dtype=np.int64)
    dl_grav = _dloga(orc, S, S["gacc"], idx, atime, hubble, ErrTolIntAccuracy, soft) if len(idx) else np.zeros(0)
    for k, i in enumerate(idx):
        if flags is not None and (flags[i] & 3):
            continue
        ty = int(S["type"][i]) & 7
        titype = TI_ACCEL
        dti = int(convert_timestep_to_ti(np.array([dl_grav[k]]), dti_max, Ti, tl, MinSizeTimestep)[0])
        if ty == 0 or ty == 5:                                                              # :786-794
            dl_h, th = get_timestep_hydro_dloga(ty, float(S["hsml"][i]), float(S["dthsml"][i]), float(S["maxsignalvel"][i]), atime, hubble,
                                                CourantFac, None if bhmin is None else int(bhmin[i]), dloga_for_bin)
            dti_h = int(convert_timestep_to_ti(np.array([dl_h]), dti_max, Ti, tl, MinSizeTimestep)[0])
            if dti_h < dti:
                dti, titype = dti_h, th
        if dti <= 1 or dti > TIMEBASE:
            badbins += 1
        ntitype[titype] += 1
        b = get_timebin_from_dti(dti, int(S["tb_hydro"][i]), Ti)
        if b < 1:
            bad += 1
        if is_timebin_active(int(S["tb_hydro"][i]), Ti) and is_timebin_active(b, Ti):       # :815-818
            S["tb_hydro"][i] = b
            S["tb_grav"][i] = b
        mTimeBin = min(mTimeBin, b)
        maxTimeBin = max(maxTimeBin, b)
    if isPM and times["PM_length"] > dti_from_timebin(maxTimeBin):                          # :835-836
        times["PM_length"] = dti_from_timebin(maxTimeBin)
    times["mintimebin"] = mTimeBin
    times["maxtimebin"] = maxTimeBin
    return dict(mTimeBin=mTimeBin, maxTimeBin=maxTimeBin, isPM=int(isPM), ntitype=ntitype, badstepsizecount=bad, badtimebins=badbins)
