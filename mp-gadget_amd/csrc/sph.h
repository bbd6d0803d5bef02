// sph.h -- SPH density + hydro force on the device tree (see sph.hip)
#pragma once
#include <cstdlib>
#include "mpg_common.h"
#include "tree_build.h"

namespace mpg {

struct alignas(32) Aux4 { // tree-ordered density source record: predicted velocity, predicted entropy^(1/gamma)
    double x, y, z, w;
};

struct HydroSrc { // tree-ordered hydro source record (96 bytes)
    double vx, vy, vz, hsml;
    double density, eomdensity, pressure, soundspeed;
    double f2, dhsml, entvarpred, dloga;
};

// device view of the caller's particle table (caller order, n entries each; SPH slot fields are indexed by particle)
struct SphView {
    const double *pos;
    const float *mass;
    const uint8_t *type;
    double *hsml, *dthsml;
    const double *vel, *gacc, *gpm, *hydroacc_in;
    const uint8_t *tb_hydro, *tb_grav;
    const double *entropy, *dtentropy_in;
    double *density, *egywtdensity, *dhsmlegyfac, *divvel, *curlvel;
    double *gradrho;
    double *hydroacc_out, *dtentropy_out, *maxsignalvel;
};

struct DensityCtl {
    int ktype, update_hsml, DoEgyDensity, BlackHoleOn;
    double DesNumNgb, MinGasHsml;
    double *Left, *Right, *NumNgb;
    const double *entvarpred;
};

struct HydroCtl {
    int ktype;
    double fac_mu, fac_vsic_fix, hubble_a2;
};

double sph_desnumngb(const mpg_density_params &P);

struct SphEngine {
    DevBuf<double> left, right, numngb, entvarpred, hsml_tree;
    DevBuf<int> queue_a, queue_b, slot_of;
    DevBuf<uint8_t> active_flags;
    DevBuf<Aux4> aux;
    DevBuf<HydroSrc> hsrc;
    DevBuf<double> hsml_t; // the sources' smoothing lengths in tree order (hydro distance tests)
    DevBuf<unsigned> ctr;
    DevBuf<unsigned long long> stats;
    bool hmax_pending = false;
    SphView hsml_view{}; // the caller's arrays at the last density(): calc_hmax gathers Hsml from them
    int64_t last_iterations = 0, last_targets = 0, last_interactions = 0, last_candidates = 0;

    const uint8_t *mark_active(const int *d_active, int64_t nactive, int64_t n, hipStream_t st);
    // density(), density.c:234-355
    void density(TreeBuilder &tree, const SphView &A, const mpg_sph_times &T, const mpg_density_params &P, double force_softening,
                 const int *d_active, int64_t nactive, int64_t n, int update_hsml, int DoEgyDensity, int BlackHoleOn, bool bh_in_tree,
                 hipStream_t st);
    // set_init_hsml(), density.c:691-749
    void set_init_hsml(TreeBuilder &tree, const SphView &A, const mpg_density_params &P, double MeanGasSeparation, hipStream_t st);
    // the hmax half of force_tree_calc_moments after density (run.c:477)
    void calc_hmax(TreeBuilder &tree, hipStream_t st);
    // hydro_force(), hydra.c:153-245
    void hydro_force(TreeBuilder &tree, const SphView &A, const mpg_sph_times &T, const mpg_density_params &P, const mpg_hydro_params &HP,
                     const int *d_active, int64_t nactive, int64_t n, hipStream_t st);
};

} // namespace mpg
