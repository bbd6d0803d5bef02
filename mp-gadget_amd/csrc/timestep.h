// timestep.h -- launch interface of the time-integration kernels (see timestep.hip)
#pragma once
#include "mpg_common.h"
#include "../../include/mpgadget_hip.h"

namespace mpg {
// what the per-particle timestep conversion needs of the integer timeline (timebinmgr.c): the sync points (device copy), the
// current log a = loga_from_ti(Ti_Current) and ti0 = ti_from_loga(that), TimestepParams.MinSizeTimestep
struct HierTimeline {
    const double *sp;
    int nsync;
    double loga_cur;
    int64_t ti0;
    double MinSizeTimestep;
};
void launch_timestep_hydro(int64_t n, const uint8_t *type, const double *hsml, const double *dthsml, const double *maxsig, const uint8_t *bh_mintimebin,
                           const double *dloga_for_bin, double atime, double hubble, double courant, double fac3, double *dloga, uint8_t *titype,
                           hipStream_t st);
void launch_find_hydro_timesteps(const int *list, int64_t nlist, const uint8_t *type, const uint8_t *flags, const double *hsml, const double *dthsml,
                                 const double *maxsig, const uint8_t *bh_mintimebin, const double *dloga_for_bin, const uint8_t *tb_grav,
                                 uint8_t *tb_hydro, double atime, double hubble, double courant, double fac3, const HierTimeline &T, int64_t dti_max,
                                 int64_t Ti_Current, unsigned long long *out, hipStream_t st);
void launch_find_timesteps(const int *list, int64_t nlist, const uint8_t *type, const uint8_t *flags, const double *gacc, const double *gpm,
                           const double *hsml, const double *dthsml, const double *maxsig, const uint8_t *bh_mintimebin, const double *dloga_for_bin,
                           uint8_t *tb_grav, uint8_t *tb_hydro, double atime, double hubble, double errtol, double soft, double courant, double fac3,
                           const HierTimeline &T, int64_t dti_max, int64_t Ti_Current, unsigned long long *out, hipStream_t st);
void launch_assign_gravity_bins(const int *list, int64_t nlist, const double *gacc, const double *gpm, const uint8_t *flags, double atime,
                                double hubble, double errtol, double soft, const HierTimeline &T, int64_t dti_max, int largest_active, uint8_t *tb,
                                unsigned long long *counts, unsigned long long *bad, hipStream_t st);
void launch_level_gravity_bins(const int *list, int64_t nlist, const double *gacc, const double *gpm, const uint8_t *flags, double atime, double hubble,
                               double errtol, double soft, const HierTimeline &T, int64_t dti_max, int ti, uint8_t *tb, unsigned long long *bad,
                               hipStream_t st);
void launch_push_down_bins(const int *list, int64_t nlist, int push_down_bin, uint8_t *tb, hipStream_t st);
void launch_kick_list(const int *list, int64_t nlist, double *vel, const double *acc, const uint8_t *flags, double gravkick, hipStream_t st);
void launch_sublist_flags(const int *list, int64_t nlist, const uint8_t *tb, const uint8_t *flags, int maxtimebin, int64_t Ti_Current, int *value,
                          uint8_t *keep, hipStream_t st);
void compact_flagged(const int *value, const uint8_t *keep, int64_t n, int *out, unsigned long long *d_count, DevBuf<char> &tmp, hipStream_t st);
void launch_drift(int64_t n, double *pos, const double *vel, const uint8_t *type, const uint8_t *flags, double *hsml, const double *dthsml,
                  double ddrift, double box, const double shift[3], unsigned *err, hipStream_t st);
void launch_pm_half_kick(int64_t n, double *vel, const double *gravpm, const uint8_t *flags, double F, hipStream_t st);
void launch_half_kick(int64_t n, const int *active, int64_t nactive, double *vel, const double *gacc, const uint8_t *type, const uint8_t *flags,
                      const uint8_t *tbg, const uint8_t *tbh, const double *hacc, double *entropy, const double *dtentropy,
                      const mpg_kick_factors &K, unsigned *err, hipStream_t st);
void launch_timestep_gravity(int64_t n, const double *gacc, const double *gpm, double atime, double hubble, double errtol, double soft,
                             double *dloga, hipStream_t st);
} // namespace mpg
