// timestep.hip -- the streaming time-integration loops either side of the force path, on device-resident
// arrays: with positions, velocities and accelerations kept in HBM between force steps, the per-step host <-> device traffic
// of the drop-in path is no longer needed.
//
//   k_drift          drift_all_particles / real_drift_particle   libgadget/drift.c:18-102
//   k_pm_half_kick   apply_PM_half_kick                          libgadget/timestep.c:964-985
//   k_half_kick      apply_half_kick + do_grav_short_range_kick + do_hydro_kick (gas part)   timestep.c:873-929, 988-1036
//   k_assign_gravity_bins, k_level_gravity_bins, k_push_down_bins, k_kick_list, k_sublist_flags
//                    the per-particle loops of the hierarchical gravity level loop             timestep.c:239-599, 1435-1478
// The per-bin factors (get_exact_drift/gravkick/hydrokick_factor, dloga_from_dti) are computed by the caller with the
// reference's own functions and passed in (mpg_kick_factors), as for the SPH loops.  Not carried: black-hole repositioning
// and the dynamic-friction / drag kicks of type-5 particles (sub-grid physics, out of scope).
// Pure HBM streaming: per particle 73 B read + 32 B written (drift), 48 + 24 B (PM kick), up to 98 + 32 B (half kick).
// Floating-point contraction is off in this file: the reference's loops are compiled without FMA, and these results are
// required to be bit-identical to it (tests/test_gpu_timestep.py).
#include "mpg_common.h"
#include "../../include/mpgadget_hip.h"
#include "timestep.h"
#include <cstring>
#include <rocprim/rocprim.hpp>

#pragma clang fp contract(off)

namespace mpg {

__global__ void __launch_bounds__(256) k_drift(int64_t n, double *__restrict__ pos, const double *__restrict__ vel, const uint8_t *__restrict__ type,
                                               const uint8_t *__restrict__ flags, double *__restrict__ hsml, const double *__restrict__ dthsml,
                                               double ddrift, double box, double s0, double s1, double s2, unsigned *__restrict__ err)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    const double shift[3] = {s0, s1, s2};
    double p[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
    const bool gone = flags && (flags[i] & 3); // IsGarbage | Swallowed: only the random shift is kept up to date (drift.c:21-30)
    if(!gone) {
        if(type && type[i] == 0 && hsml) { // drift.c:56-70
            double h = hsml[i] + dthsml[i] * ddrift;
            if(h <= 0)
                atomicExch(err, 5u);
            const double maxh = box / 2.;
            if(h > maxh)
                h = maxh;
            hsml[i] = h;
        }
#pragma unroll
        for(int j = 0; j < 3; j++) {
            p[j] += vel[3 * i + j] * ddrift + shift[j];
            if(!isfinite(p[j]))
                atomicExch(err, 5u);
        }
    }
    else {
#pragma unroll
        for(int j = 0; j < 3; j++)
            p[j] += shift[j];
    }
#pragma unroll
    for(int j = 0; j < 3; j++) {
        if(isfinite(p[j])) { // (the reference aborts before wrapping a non-finite position)
            while(p[j] > box)
                p[j] -= box;
            while(p[j] <= 0)
                p[j] += box;
        }
        pos[3 * i + j] = p[j];
    }
}

__global__ void __launch_bounds__(256) k_pm_half_kick(int64_t n, double *__restrict__ vel, const double *__restrict__ gravpm,
                                                      const uint8_t *__restrict__ flags, double F)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n || (flags && (flags[i] & 3)))
        return;
#pragma unroll
    for(int j = 0; j < 3; j++)
        vel[3 * i + j] += gravpm[3 * i + j] * F;
}

__global__ void __launch_bounds__(256) k_half_kick(int64_t n, const int *__restrict__ active, int64_t nactive, double *__restrict__ vel,
                                                   const double *__restrict__ gacc, const uint8_t *__restrict__ type, const uint8_t *__restrict__ flags,
                                                   const uint8_t *__restrict__ tbg, const uint8_t *__restrict__ tbh, const double *__restrict__ hacc,
                                                   double *__restrict__ entropy, const double *__restrict__ dtentropy, const mpg_kick_factors K,
                                                   unsigned *__restrict__ err)
{
    const int64_t pa = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(pa >= (active ? nactive : n))
        return;
    const int64_t i = active ? active[pa] : pa;
    if(flags && (flags[i] & 3))
        return;
    const int bg = tbg ? tbg[i] : 0;
    if(bg > MPG_TIMEBINS) {
        atomicExch(err, 4u);
        return;
    }
    double v[3] = {vel[3 * i], vel[3 * i + 1], vel[3 * i + 2]};
    if(K.bin_active[bg]) {
#pragma unroll
        for(int j = 0; j < 3; j++)
            v[j] += gacc[3 * i + j] * K.gravkick[bg];
    }
    const int ty = type ? type[i] : 1;
    if(ty == 0) {
        const int bh = tbh ? tbh[i] : 0;
#pragma unroll
        for(int j = 0; j < 3; j++)
            v[j] += hacc[3 * i + j] * K.hydrokick[bh];
        double vv = 0;
#pragma unroll
        for(int j = 0; j < 3; j++)
            vv += v[j] * v[j];
        vv = sqrt(vv);
        if(vv > 0 && vv / K.atime > K.MaxGasVel) {
#pragma unroll
            for(int j = 0; j < 3; j++)
                v[j] *= K.MaxGasVel * K.atime / vv;
        }
        entropy[i] += dtentropy[i] * K.dt_entr[bh];
    }
#pragma unroll
    for(int j = 0; j < 3; j++)
        vel[3 * i + j] = v[j];
}

// get_timestep_gravity_dloga, timestep.c:1039-1074
__global__ void __launch_bounds__(256) k_timestep_gravity(int64_t n, const double *__restrict__ gacc, const double *__restrict__ gpm, double atime,
                                                          double hubble, double errtol, double soft, double *__restrict__ dloga)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    const double a2inv = 1 / (atime * atime);
    double ax = a2inv * gacc[3 * i + 0];
    double ay = a2inv * gacc[3 * i + 1];
    double az = a2inv * gacc[3 * i + 2];
    ay += a2inv * gpm[3 * i + 1];
    ax += a2inv * gpm[3 * i + 0];
    az += a2inv * gpm[3 * i + 2];
    double ac2 = ax * ax + ay * ay + az * az;
    if(ac2 == 0)
        ac2 = 1.0e-60;
    const double ac = sqrt(ac2);
    const double dt = sqrt(2 * errtol * atime * (soft / 2.8) / ac);
    dloga[i] = dt * hubble;
}


// ---- hierarchical gravity (timestep.c:239-599): per-particle loops -------------------------------------------------------------

// ti_from_loga, timebinmgr.c:400-417 (sp = SyncPoints[].loga, nsync >= 2)
__device__ __forceinline__ int64_t ti_from_loga_dev(const double loga, const double *__restrict__ sp, const int nsync)
{
    int i;
    for(i = 1; i < nsync - 1; i++)
        if(sp[i] > loga)
            break;
    const double logDTime = (sp[i] - sp[i - 1]) / (double)(1ull << MPG_TIMEBINS);
    int64_t ti = (int64_t)(i - 1) << MPG_TIMEBINS;
    ti = (int64_t)((double)ti + (loga - sp[i - 1]) / logDTime); // "ti += double": converted, added, truncated
    return ti;
}

// convert_timestep_to_ti (timestep.c:1155-1175) with dti_from_dloga (timebinmgr.c:434-440): loga_cur = loga_from_ti(Ti_Current)
// and ti0 = ti_from_loga(loga_cur) are the same for every particle and come from the host
__device__ __forceinline__ int64_t convert_timestep_to_ti_dev(double dloga, const int64_t dti_max, const HierTimeline &T)
{
    if(dti_max == 0)
        return 0;
    if(dloga < T.MinSizeTimestep)
        dloga = T.MinSizeTimestep;
    int64_t dti = ti_from_loga_dev(dloga + T.loga_cur, T.sp, T.nsync) - T.ti0;
    if(dti > dti_max || dti < 0)
        dti = dti_max;
    return dti;
}

// get_timestep_gravity_dloga, timestep.c:1045-1074
__device__ __forceinline__ double gravity_dloga_dev(const int64_t i, const double *__restrict__ gacc, const double *__restrict__ gpm,
                                                    const double atime, const double hubble, const double errtol, const double soft)
{
    const double a2inv = 1 / (atime * atime);
    double ax = a2inv * gacc[3 * i + 0];
    double ay = a2inv * gacc[3 * i + 1];
    double az = a2inv * gacc[3 * i + 2];
    ay += a2inv * gpm[3 * i + 1];
    ax += a2inv * gpm[3 * i + 0];
    az += a2inv * gpm[3 * i + 2];
    double ac2 = ax * ax + ay * ay + az * az;
    if(ac2 == 0)
        ac2 = 1.0e-60;
    const double ac = sqrt(ac2);
    const double dt = sqrt(2 * errtol * atime * (soft / 2.8) / ac);
    return dt * hubble;
}

// get_timestep_hydro_dloga, timestep.c:1076-1118: the Courant criterion from the signal velocity and the Gadget-4 criterion on the change
// of the smoothing length for gas; the neighbour limiter for black holes (minTimeBin of their gas neighbours, one bin up); dt = 1 for
// every other type.  fac3 = pow(atime, 3 (1 - GAMMA) / 2) comes from the host (the caller's libm, as the reference computes it).
// titype: enum TimeStepType, timestep.c:89-96 (0 ACCEL, 1 COURANT, 3 NEIGH, 4 HSML).
__device__ __forceinline__ double hydro_dloga_dev(const int64_t i, const uint8_t *__restrict__ type, const double *__restrict__ hsml,
                                                  const double *__restrict__ dthsml, const double *__restrict__ maxsig,
                                                  const uint8_t *__restrict__ bh_mintimebin, const double *__restrict__ dloga_for_bin, const double atime,
                                                  const double hubble, const double courant, const double fac3, int &titype)
{
    double dt = 1;
    titype = 0;
    const int ty = type ? (type[i] & 7) : 1;
    if(ty == 0) {
        const double dt_courant = 2 * courant * atime * hsml[i] / (fac3 * maxsig[i]);
        dt = dt_courant;
        titype = 1;
        const double dt_hsml = courant * atime * atime * fabs(hsml[i] / ((dthsml ? dthsml[i] : 0.0) + 1e-20));
        if(dt_hsml < dt) {
            dt = dt_hsml;
            titype = 4;
        }
    }
    else if(ty == 5 && bh_mintimebin && dloga_for_bin) {
        const int mb = bh_mintimebin[i];
        if(mb > 0 && mb + 1 < MPG_TIMEBINS) {
            dt = dloga_for_bin[mb + 1] / hubble; // get_dloga_for_bin(minTimeBin + 1, Ti_Current) / hubble
            titype = 3;
        }
    }
    return dt * hubble;
}

__global__ void __launch_bounds__(256) k_timestep_hydro(int64_t n, const uint8_t *__restrict__ type, const double *__restrict__ hsml,
                                                        const double *__restrict__ dthsml, const double *__restrict__ maxsig,
                                                        const uint8_t *__restrict__ bh_mintimebin, const double *__restrict__ dloga_for_bin, double atime,
                                                        double hubble, double courant, double fac3, double *__restrict__ dloga, uint8_t *__restrict__ titype)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    int tt;
    dloga[i] = hydro_dloga_dev(i, type, hsml, dthsml, maxsig, bh_mintimebin, dloga_for_bin, atime, hubble, courant, fac3, tt);
    if(titype)
        titype[i] = (uint8_t)tt;
}

// is_timebin_active, timestep.c:143-150
__device__ __forceinline__ bool timebin_active_dev(const int bin, const int64_t Ti_Current)
{
    const int64_t dti = bin > 0 ? ((int64_t)1 << bin) : 0;
    return bin <= 0 || Ti_Current <= 0 || (Ti_Current % dti) == 0; // ("bin 0 is always active and at time 0 all bins are active")
}

// The particle loop of find_hydro_timesteps (timestep.c:629-698) without the dynamic-friction bins of the black holes: the new hydro bin
// of every active gas / black-hole particle.  out[0..4]: particles by criterion (TI_ACCEL, TI_COURANT, TI_ACCRETE, TI_NEIGH, TI_HSML),
// out[5]: badstepsizecount (bin_hydro < 1), out[6]: print_bad_timebin cases (dti <= 1 or > TIMEBASE), out[7]: the smallest bin, by
// atomicMin - initialised to TIMEBINS by the caller.
__global__ void __launch_bounds__(256) k_find_hydro_timesteps(const int *__restrict__ list, int64_t nlist, const uint8_t *__restrict__ type,
                                                              const uint8_t *__restrict__ flags, const double *__restrict__ hsml,
                                                              const double *__restrict__ dthsml, const double *__restrict__ maxsig,
                                                              const uint8_t *__restrict__ bh_mintimebin, const double *__restrict__ dloga_for_bin,
                                                              const uint8_t *__restrict__ tb_grav, uint8_t *__restrict__ tb_hydro, double atime,
                                                              double hubble, double courant, double fac3, HierTimeline T, int64_t dti_max,
                                                              int64_t Ti_Current, unsigned long long *__restrict__ out)
{
    __shared__ unsigned s_cnt[8];
    __shared__ unsigned s_min;
    if(threadIdx.x < 8)
        s_cnt[threadIdx.x] = 0;
    if(threadIdx.x == 0)
        s_min = MPG_TIMEBINS + 1;
    __syncthreads();
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k < nlist) {
        const int64_t i = list ? list[k] : k;
        const int ty = type ? (type[i] & 7) : 1;
        if(!(flags && (flags[i] & 3)) && (ty == 0 || ty == 5)) {
            int titype;
            const double dloga = hydro_dloga_dev(i, type, hsml, dthsml, maxsig, bh_mintimebin, dloga_for_bin, atime, hubble, courant, fac3, titype);
            int64_t dti = convert_timestep_to_ti_dev(dloga, dti_max, T);
            if(dti <= 1 || dti > ((int64_t)1 << MPG_TIMEBINS))
                atomicAdd(&s_cnt[6], 1u);
            // get_timebin_from_dti, timestep.c:166-182: round_down_power_of_two, get_timestep_bin, and a longer step only onto an active bin
            int64_t ti_min = (int64_t)1 << MPG_TIMEBINS;
            while(ti_min > dti)
                ti_min >>= 1;
            dti = ti_min;
            int bin = 0;
            if(dti > 1)
                bin = 63 - __clzll((unsigned long long)dti);
            const int binold = tb_hydro[i];
            if(bin > binold)
                while(!timebin_active_dev(bin, Ti_Current) && bin > binold && bin > 1)
                    bin--;
            // the hydro step never exceeds the gravity step (timestep.c:651-655)
            const int bg = tb_grav ? tb_grav[i] : MPG_TIMEBINS;
            if(bin > bg) {
                bin = bg;
                titype = 0;
            }
            if(bin < 1)
                atomicAdd(&s_cnt[5], 1u);
            atomicAdd(&s_cnt[titype], 1u);
            if(timebin_active_dev(binold, Ti_Current) && timebin_active_dev(bin, Ti_Current))
                tb_hydro[i] = (uint8_t)bin;
            atomicMin(&s_min, (unsigned)bin);
        }
    }
    __syncthreads();
    if(threadIdx.x < 7 && s_cnt[threadIdx.x])
        atomicAdd(&out[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
    if(threadIdx.x == 0 && s_min <= MPG_TIMEBINS)
        atomicMin(&out[7], (unsigned long long)s_min);
}

// get_timebin_from_dti, timestep.c:166-182: round_down_power_of_two, get_timestep_bin, and a longer step only onto an active bin
__device__ __forceinline__ int timebin_from_dti_dev(int64_t dti, const int binold, const int64_t Ti_Current)
{
    int64_t ti_min = (int64_t)1 << MPG_TIMEBINS;
    while(ti_min > dti)
        ti_min >>= 1;
    dti = ti_min;
    int bin = 0;
    if(dti > 1)
        bin = 63 - __clzll((unsigned long long)dti);
    if(bin > binold)
        while(!timebin_active_dev(bin, Ti_Current) && bin > binold && bin > 1)
            bin--;
    return bin;
}

// The particle loop of find_timesteps (timestep.c:765-823; the step assignment of a run WITHOUT SplitGravityTimestepsOn, run.c:756): per
// active particle the gravity step from FullTreeGravAccel + GravPM, for gas / black holes the hydro step where it is shorter, both bins set
// to the new bin when old and new bin are active.  Not carried: ForceEqualTimesteps (the caller refuses it).
// out[0..4]: particles by criterion, out[5]: badstepsizecount (bin < 1), out[6]: print_bad_timebin cases, out[7]: smallest bin (atomicMin,
// initialised to TIMEBINS), out[8]: largest bin (atomicMax, initialised to 0).
__global__ void __launch_bounds__(256) k_find_timesteps(const int *__restrict__ list, int64_t nlist, const uint8_t *__restrict__ type,
                                                        const uint8_t *__restrict__ flags, const double *__restrict__ gacc, const double *__restrict__ gpm,
                                                        const double *__restrict__ hsml, const double *__restrict__ dthsml,
                                                        const double *__restrict__ maxsig, const uint8_t *__restrict__ bh_mintimebin,
                                                        const double *__restrict__ dloga_for_bin, uint8_t *__restrict__ tb_grav,
                                                        uint8_t *__restrict__ tb_hydro, double atime, double hubble, double errtol, double soft,
                                                        double courant, double fac3, HierTimeline T, int64_t dti_max, int64_t Ti_Current,
                                                        unsigned long long *__restrict__ out)
{
    __shared__ unsigned s_cnt[8];
    __shared__ unsigned s_min, s_max;
    if(threadIdx.x < 8)
        s_cnt[threadIdx.x] = 0;
    if(threadIdx.x == 0) {
        s_min = MPG_TIMEBINS + 1;
        s_max = 0;
    }
    __syncthreads();
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k < nlist) {
        const int64_t i = list ? list[k] : k;
        if(!(flags && (flags[i] & 3))) {
            const int ty = type ? (type[i] & 7) : 1;
            int titype = 0;
            const double dloga_gravity = gravity_dloga_dev(i, gacc, gpm, atime, hubble, errtol, soft);
            int64_t dti = convert_timestep_to_ti_dev(dloga_gravity, dti_max, T);
            if(ty == 0 || ty == 5) { // the hydro step: "always shorter"
                int th;
                const double dloga_hydro = hydro_dloga_dev(i, type, hsml, dthsml, maxsig, bh_mintimebin, dloga_for_bin, atime, hubble, courant, fac3, th);
                const int64_t dti_hydro = convert_timestep_to_ti_dev(dloga_hydro, dti_max, T);
                if(dti_hydro < dti) {
                    dti = dti_hydro;
                    titype = th;
                }
            }
            if(dti <= 1 || dti > ((int64_t)1 << MPG_TIMEBINS))
                atomicAdd(&s_cnt[6], 1u);
            atomicAdd(&s_cnt[titype], 1u);
            const int binold = tb_hydro[i];
            const int bin = timebin_from_dti_dev(dti, binold, Ti_Current);
            if(bin < 1)
                atomicAdd(&s_cnt[5], 1u);
            if(timebin_active_dev(binold, Ti_Current) && timebin_active_dev(bin, Ti_Current)) {
                tb_hydro[i] = (uint8_t)bin;
                tb_grav[i] = (uint8_t)bin;
            }
            atomicMin(&s_min, (unsigned)bin);
            atomicMax(&s_max, (unsigned)bin);
        }
    }
    __syncthreads();
    if(threadIdx.x < 7 && s_cnt[threadIdx.x])
        atomicAdd(&out[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
    if(threadIdx.x == 0 && s_min <= MPG_TIMEBINS) {
        atomicMin(&out[7], (unsigned long long)s_min);
        atomicMax(&out[8], (unsigned long long)s_max);
    }
}

// The first loop of hierarchical_gravity_and_timesteps (timestep.c:345-370): new gravity bin of every particle of the list from
// the stored acceleration; counts[bin] += 1; bad += 1 for dti <= 1 or > TIMEBASE (print_bad_timebin).
__global__ void __launch_bounds__(256) k_assign_gravity_bins(const int *__restrict__ list, int64_t nlist, const double *__restrict__ gacc,
                                                             const double *__restrict__ gpm, const uint8_t *__restrict__ flags, double atime,
                                                             double hubble, double errtol, double soft, HierTimeline T, int64_t dti_max,
                                                             int largest_active, uint8_t *__restrict__ tb,
                                                             unsigned long long *__restrict__ counts, unsigned long long *__restrict__ bad)
{
    __shared__ unsigned s_cnt[MPG_TIMEBINS + 2];
    for(int b = threadIdx.x; b < MPG_TIMEBINS + 2; b += blockDim.x)
        s_cnt[b] = 0;
    __syncthreads();
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k < nlist) {
        const int64_t pa = list ? list[k] : k;
        if(!(flags && (flags[pa] & 3))) {
            const double dloga = gravity_dloga_dev(pa, gacc, gpm, atime, hubble, errtol, soft);
            int64_t dti = convert_timestep_to_ti_dev(dloga, dti_max, T);
            // round_down_power_of_two, timebinmgr.c:449-462 (dti >= 0 here)
            int64_t ti_min = (int64_t)1 << MPG_TIMEBINS;
            while(ti_min > dti)
                ti_min >>= 1;
            dti = ti_min;
            if(dti <= 1 || dti > ((int64_t)1 << MPG_TIMEBINS))
                atomicAdd(&s_cnt[MPG_TIMEBINS + 1], 1u);
            // get_timestep_bin, timestep.c:1301-1315
            int bin = 0;
            if(dti > 1)
                bin = 63 - __clzll((unsigned long long)dti);
            if(bin > largest_active)
                bin = largest_active;
            atomicAdd(&s_cnt[bin], 1u);
            tb[pa] = (uint8_t)bin;
        }
    }
    __syncthreads();
    for(int b = threadIdx.x; b < MPG_TIMEBINS + 2; b += blockDim.x)
        if(s_cnt[b]) {
            if(b <= MPG_TIMEBINS)
                atomicAdd(&counts[b], (unsigned long long)s_cnt[b]);
            else
                atomicAdd(bad, (unsigned long long)s_cnt[b]);
        }
}

// The loop of the lower levels (timestep.c:461-477): a particle whose step from the acceleration AT THIS LEVEL is shorter than
// the level's goes one bin down; bad += 1 if that happens at ti == 1.
__global__ void __launch_bounds__(256) k_level_gravity_bins(const int *__restrict__ list, int64_t nlist, const double *__restrict__ gacc,
                                                            const double *__restrict__ gpm, const uint8_t *__restrict__ flags, double atime,
                                                            double hubble, double errtol, double soft, HierTimeline T, int64_t dti_max, int ti,
                                                            uint8_t *__restrict__ tb, unsigned long long *__restrict__ bad)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= nlist)
        return;
    const int64_t pa = list ? list[k] : k;
    if(flags && (flags[pa] & 3))
        return;
    const double dloga = gravity_dloga_dev(pa, gacc, gpm, atime, hubble, errtol, soft);
    const int64_t dti = convert_timestep_to_ti_dev(dloga, dti_max, T);
    const int64_t dti_bin = ti > 0 ? ((int64_t)1 << ti) : 0; // dti_from_timebin
    if(dti < dti_bin) {
        tb[pa] = (uint8_t)(ti - 1);
        if(ti == 1)
            atomicAdd(bad, 1ull);
    }
}

// "Pushing down top bin" (timestep.c:404-411)
__global__ void __launch_bounds__(256) k_push_down_bins(const int *__restrict__ list, int64_t nlist, int push_down_bin, uint8_t *__restrict__ tb)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= nlist)
        return;
    const int64_t pa = list ? list[k] : k;
    if(tb[pa] > push_down_bin)
        tb[pa] = (uint8_t)push_down_bin;
}

// apply_hierarchical_grav_kick (timestep.c:238-278) + do_grav_short_range_kick (:995-1001)
__global__ void __launch_bounds__(256) k_kick_list(const int *__restrict__ list, int64_t nlist, double *__restrict__ vel,
                                                   const double *__restrict__ acc, const uint8_t *__restrict__ flags, double gravkick)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= nlist)
        return;
    const int64_t pa = list ? list[k] : k;
    if(flags && (flags[pa] & 3))
        return;
    for(int j = 0; j < 3; j++)
        vel[3 * pa + j] += acc[3 * pa + j] * gravkick;
}

// build_active_sublist (timestep.c:1435-1478): keep[k] = 1 for the entries that stay; value[k] = the particle index
__global__ void __launch_bounds__(256) k_sublist_flags(const int *__restrict__ list, int64_t nlist, const uint8_t *__restrict__ tb,
                                                       const uint8_t *__restrict__ flags, int maxtimebin, int64_t Ti_Current,
                                                       int *__restrict__ value, uint8_t *__restrict__ keep)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= nlist)
        return;
    const int64_t pa = list ? list[k] : k;
    const int bin = tb[pa];
    bool ok = !(flags && (flags[pa] & 3)) && bin <= maxtimebin;
    if(ok && bin > 0 && Ti_Current > 0) // is_timebin_active, timestep.c:143-150
        ok = (Ti_Current % ((int64_t)1 << bin)) == 0;
    value[k] = (int)pa;
    keep[k] = ok ? 1 : 0;
}

static inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }

void launch_timestep_gravity(int64_t n, const double *gacc, const double *gpm, double atime, double hubble, double errtol, double soft,
                             double *dloga, hipStream_t st)
{
    if(n > 0)
        hipLaunchKernelGGL(k_timestep_gravity, dim3(nblk(n)), dim3(256), 0, st, n, gacc, gpm, atime, hubble, errtol, soft, dloga);
    MPG_HIP(hipGetLastError());
}


void launch_timestep_hydro(int64_t n, const uint8_t *type, const double *hsml, const double *dthsml, const double *maxsig, const uint8_t *bh_mintimebin,
                           const double *dloga_for_bin, double atime, double hubble, double courant, double fac3, double *dloga, uint8_t *titype,
                           hipStream_t st)
{
    if(n > 0)
        hipLaunchKernelGGL(k_timestep_hydro, dim3(nblk(n)), dim3(256), 0, st, n, type, hsml, dthsml, maxsig, bh_mintimebin, dloga_for_bin, atime, hubble,
                           courant, fac3, dloga, titype);
    MPG_HIP(hipGetLastError());
}

void launch_find_hydro_timesteps(const int *list, int64_t nlist, const uint8_t *type, const uint8_t *flags, const double *hsml, const double *dthsml,
                                 const double *maxsig, const uint8_t *bh_mintimebin, const double *dloga_for_bin, const uint8_t *tb_grav,
                                 uint8_t *tb_hydro, double atime, double hubble, double courant, double fac3, const HierTimeline &T, int64_t dti_max,
                                 int64_t Ti_Current, unsigned long long *out, hipStream_t st)
{
    if(nlist > 0)
        hipLaunchKernelGGL(k_find_hydro_timesteps, dim3(nblk(nlist)), dim3(256), 0, st, list, nlist, type, flags, hsml, dthsml, maxsig, bh_mintimebin,
                           dloga_for_bin, tb_grav, tb_hydro, atime, hubble, courant, fac3, T, dti_max, Ti_Current, out);
    MPG_HIP(hipGetLastError());
}

void launch_find_timesteps(const int *list, int64_t nlist, const uint8_t *type, const uint8_t *flags, const double *gacc, const double *gpm,
                           const double *hsml, const double *dthsml, const double *maxsig, const uint8_t *bh_mintimebin, const double *dloga_for_bin,
                           uint8_t *tb_grav, uint8_t *tb_hydro, double atime, double hubble, double errtol, double soft, double courant, double fac3,
                           const HierTimeline &T, int64_t dti_max, int64_t Ti_Current, unsigned long long *out, hipStream_t st)
{
    if(nlist > 0)
        hipLaunchKernelGGL(k_find_timesteps, dim3(nblk(nlist)), dim3(256), 0, st, list, nlist, type, flags, gacc, gpm, hsml, dthsml, maxsig, bh_mintimebin,
                           dloga_for_bin, tb_grav, tb_hydro, atime, hubble, errtol, soft, courant, fac3, T, dti_max, Ti_Current, out);
    MPG_HIP(hipGetLastError());
}

void launch_assign_gravity_bins(const int *list, int64_t nlist, const double *gacc, const double *gpm, const uint8_t *flags, double atime,
                                double hubble, double errtol, double soft, const HierTimeline &T, int64_t dti_max, int largest_active, uint8_t *tb,
                                unsigned long long *counts, unsigned long long *bad, hipStream_t st)
{
    if(nlist > 0)
        hipLaunchKernelGGL(k_assign_gravity_bins, dim3(nblk(nlist)), dim3(256), 0, st, list, nlist, gacc, gpm, flags, atime, hubble, errtol, soft, T,
                           dti_max, largest_active, tb, counts, bad);
    MPG_HIP(hipGetLastError());
}

void launch_level_gravity_bins(const int *list, int64_t nlist, const double *gacc, const double *gpm, const uint8_t *flags, double atime, double hubble,
                               double errtol, double soft, const HierTimeline &T, int64_t dti_max, int ti, uint8_t *tb, unsigned long long *bad,
                               hipStream_t st)
{
    if(nlist > 0)
        hipLaunchKernelGGL(k_level_gravity_bins, dim3(nblk(nlist)), dim3(256), 0, st, list, nlist, gacc, gpm, flags, atime, hubble, errtol, soft, T,
                           dti_max, ti, tb, bad);
    MPG_HIP(hipGetLastError());
}

void launch_push_down_bins(const int *list, int64_t nlist, int push_down_bin, uint8_t *tb, hipStream_t st)
{
    if(nlist > 0)
        hipLaunchKernelGGL(k_push_down_bins, dim3(nblk(nlist)), dim3(256), 0, st, list, nlist, push_down_bin, tb);
    MPG_HIP(hipGetLastError());
}

void launch_kick_list(const int *list, int64_t nlist, double *vel, const double *acc, const uint8_t *flags, double gravkick, hipStream_t st)
{
    if(nlist > 0)
        hipLaunchKernelGGL(k_kick_list, dim3(nblk(nlist)), dim3(256), 0, st, list, nlist, vel, acc, flags, gravkick);
    MPG_HIP(hipGetLastError());
}

void launch_sublist_flags(const int *list, int64_t nlist, const uint8_t *tb, const uint8_t *flags, int maxtimebin, int64_t Ti_Current, int *value,
                          uint8_t *keep, hipStream_t st)
{
    if(nlist > 0)
        hipLaunchKernelGGL(k_sublist_flags, dim3(nblk(nlist)), dim3(256), 0, st, list, nlist, tb, flags, maxtimebin, Ti_Current, value, keep);
    MPG_HIP(hipGetLastError());
}

// order-preserving compaction of value[k] with keep[k] != 0 (the merge step of build_active_sublist); *d_count receives the count
void compact_flagged(const int *value, const uint8_t *keep, int64_t n, int *out, unsigned long long *d_count, DevBuf<char> &tmp, hipStream_t st)
{
    if(n <= 0) {
        MPG_HIP(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), st));
        return;
    }
    size_t bytes = 0;
    MPG_HIP(rocprim::select(nullptr, bytes, value, keep, out, d_count, (size_t)n, st));
    tmp.reserve(bytes + 16);
    MPG_HIP(rocprim::select((void *)tmp.p, bytes, value, keep, out, d_count, (size_t)n, st));
}

void launch_drift(int64_t n, double *pos, const double *vel, const uint8_t *type, const uint8_t *flags, double *hsml, const double *dthsml,
                  double ddrift, double box, const double shift[3], unsigned *err, hipStream_t st)
{
    if(n > 0)
        hipLaunchKernelGGL(k_drift, dim3(nblk(n)), dim3(256), 0, st, n, pos, vel, type, flags, hsml, dthsml, ddrift, box, shift[0], shift[1],
                           shift[2], err);
    MPG_HIP(hipGetLastError());
}

void launch_pm_half_kick(int64_t n, double *vel, const double *gravpm, const uint8_t *flags, double F, hipStream_t st)
{
    if(n > 0)
        hipLaunchKernelGGL(k_pm_half_kick, dim3(nblk(n)), dim3(256), 0, st, n, vel, gravpm, flags, F);
    MPG_HIP(hipGetLastError());
}

void launch_half_kick(int64_t n, const int *active, int64_t nactive, double *vel, const double *gacc, const uint8_t *type, const uint8_t *flags,
                      const uint8_t *tbg, const uint8_t *tbh, const double *hacc, double *entropy, const double *dtentropy,
                      const mpg_kick_factors &K, unsigned *err, hipStream_t st)
{
    const int64_t na = active ? nactive : n;
    if(na > 0)
        hipLaunchKernelGGL(k_half_kick, dim3(nblk(na)), dim3(256), 0, st, n, active, nactive, vel, gacc, type, flags, tbg, tbh, hacc, entropy,
                           dtentropy, K, err);
    MPG_HIP(hipGetLastError());
}

} // namespace mpg
