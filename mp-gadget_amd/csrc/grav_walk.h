// grav_walk.h -- launch interface of the short-range gravity walk (see grav_walk.hip)
#pragma once
#include "mpg_common.h"

namespace mpg {

struct WalkIO {
    int64_t ntargets = 0;
    const int *targets = nullptr;      // caller indices of the targets, or null: every particle of the tree, in tree order
    const double *pos = nullptr;       // caller order [n][3]
    const float *mass = nullptr;       // caller order [n]
    const double *oldacc = nullptr;    // caller order [n]: |a_old| / G, or null
    const double *prev_accel = nullptr;// caller order [n][3] (used when oldacc is null)
    const double *gravpm = nullptr;    // caller order [n][3] (may be null)
    double *accel = nullptr;           // caller order [n][3]
    double *potential = nullptr;       // caller order [n] or null
    const float *tab_force = nullptr;  // [NTAB] shortrange_table           (gravity.c:20)
    const float *tab_pot = nullptr;    // [NTAB] shortrange_table_potential
    unsigned long long *counters = nullptr; // [3] pp interactions, nodes visited, nodes used (COUNT builds only)
};

void launch_grav_walk(const TreeView &tv, const GravParams &gp, const WalkIO &io, bool want_pot, bool count, int thresh, hipStream_t st);

} // namespace mpg
