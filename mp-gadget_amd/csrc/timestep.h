// timestep.h -- launch interface of the time-integration kernels (see timestep.hip)
#pragma once
#include "mpg_common.h"
#include "../../include/mpgadget_hip.h"

namespace mpg {
void launch_drift(int64_t n, double *pos, const double *vel, const uint8_t *type, const uint8_t *flags, double *hsml, const double *dthsml,
                  double ddrift, double box, const double shift[3], unsigned *err, hipStream_t st);
void launch_pm_half_kick(int64_t n, double *vel, const double *gravpm, const uint8_t *flags, double F, hipStream_t st);
void launch_half_kick(int64_t n, const int *active, int64_t nactive, double *vel, const double *gacc, const uint8_t *type, const uint8_t *flags,
                      const uint8_t *tbg, const uint8_t *tbh, const double *hacc, double *entropy, const double *dtentropy,
                      const mpg_kick_factors &K, unsigned *err, hipStream_t st);
void launch_timestep_gravity(int64_t n, const double *gacc, const double *gpm, double atime, double hubble, double errtol, double soft,
                             double *dloga, hipStream_t st);
} // namespace mpg
