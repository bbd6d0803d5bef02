// grav_pair.h -- the pair arithmetic shared by the walk kernels of grav_walk.hip, grav_walk_coop.hip and grav_walk_shared.hip
// (grav_walk_split.hip carries its own variant, tuned for register pressure: see the comments there).
#pragma once
#include "mpg_common.h"

namespace mpg {

struct WTab {
    double a, b; // T[t], T[t+1]: one 16-byte LDS read per table lookup
};

__device__ __forceinline__ double tab_lo(const WTab &p) { return p.a; }
__device__ __forceinline__ double tab_hi(const WTab &p) { return p.b; }
__device__ __forceinline__ double tab_lo(const float2 &p) { return (double)p.x; }
__device__ __forceinline__ double tab_hi(const float2 &p) { return (double)p.y; }

__device__ __forceinline__ double rsqrt_nr(double x)
{
    // v_rsq_f64 + one cubic Newton step -> full double precision; x > 0
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * y), y, 1.0);
    return fma(y * e, fma(e, 0.375, 0.5), y);
}

// apply_accn_to_output, gravshort-tree.c:158-193, for one source (particle or node used unopened) at separation (dx, dy, dz).
// wf / wp: the force / potential window tables as (T[t], T[t+1]) pairs (doubles or floats) in LDS.
template <bool POT, typename PTab>
__device__ __forceinline__ void pair_force(const Src4 s, const double dx, const double dy, const double dz, const GravParams &gp,
                                           const WTab *__restrict__ wf, const PTab *__restrict__ wp, double &ax, double &ay, double &az,
                                           double &pot)
{
    const double r2 = dx * dx + dy * dy + dz * dz;
    const double rinv = rsqrt_nr(fmax(r2, 1e-300));
    const double r = r2 * rinv;                   // exactly 0 for the self interaction
    const double ti = r * gp.inv_cell_dx;         // r / cellsize / dx, gravity.c:57-58
    const bool inrange = ti < (double)(NTAB - 1); // tabindex >= NTAB-1 contributes nothing (gravity.c:60-61)
    double fac = s.m * rinv * rinv * rinv;
    double facpot = -s.m * rinv;
    if(r2 < gp.h * gp.h) { // Gadget-2 softening spline with the reference's truncated constants
        const double u = r / gp.h;
        double wpk;
        if(u < 0.5) {
            fac = s.m * gp.h3inv * (10.666666666667 + u * u * (32.0 * u - 38.4));
            wpk = -2.8 + u * u * (5.333333333333 + u * u * (6.4 * u - 9.6));
        }
        else {
            fac = s.m * gp.h3inv * (21.333333333333 - 48.0 * u + 38.4 * u * u - 10.666666666667 * u * u * u - 0.066666666667 / (u * u * u));
            wpk = -3.2 + 0.066666666667 / u + u * u * (10.666666666667 + u * (-16.0 + u * (9.6 - 2.133333333333 * u)));
        }
        facpot = s.m / gp.h * wpk;
    }
    const double tcl = inrange ? ti : 0.0;
    const int t = (int)tcl;
    // (t + 1 - i) and (i - t) of gravity.c:63 are both exact, so 1 - (i - t) is the same number as (t + 1 - i)
    const double w1 = tcl - (double)t, w0 = 1.0 - w1;
    const WTab f = wf[t];
    const double wgt = inrange ? (w0 * f.a + w1 * f.b) : 0.0;
    fac *= wgt;
    ax = fma(dx, fac, ax);
    ay = fma(dy, fac, ay);
    az = fma(dz, fac, az);
    if(POT) {
        const PTab p = wp[t];
        const double wpot = inrange ? (w0 * tab_lo(p) + w1 * tab_hi(p)) : 0.0;
        pot = fma(facpot, wpot, pot);
    }
}

} // namespace mpg
