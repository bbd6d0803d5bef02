/* oracle/pm_oracle.c -- TEST INFRASTRUCTURE ONLY (the CPU checker and bench.py's cpu_baseline leg; the product never links it).
 *
 * The particle <-> mesh loops and the Fourier-space sweeps of the reference's long-range step, restated in C with OpenMP as the
 * reference threads them, for ONE rank that holds the whole Nmesh^3 mesh (one region = the whole periodic mesh, so the region
 * offsets of petapm.c:969 are zero and the wrap of petapm.c:903-918 is applied at deposit / read-out).  The transforms themselves are
 * PFFT's in the reference (third party, 1.0.8-alpha3-fftw3-2don2d, not vendored, not in this image): oracle.py runs pocketfft in
 * their place.  Pinned against oracle.py's numpy restatement of the same functions (tests/test_oracle_pm.py), which the
 * reference's test_gravity.c bounds pin end to end (tests/test_oracle_kat.py).
 *
 * Mesh layout: row-major [x][y][z], real mesh Nmesh^3, Fourier mesh Nmesh x Nmesh x (Nmesh/2+1) complex (numpy rfftn).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

/* the cell, residual and eight (linear index, weight) of a cloud follow pm_iterate_one (petapm.c:955-1004) */
static inline int wrap(int i, int n)
{
    while (i < 0) i += n;                          /* petapm.c:905-906 */
    while (i >= n) i -= n;
    return i;
}

/* put_particle_to_mesh through pm_iterate (gravpm.c:499-505, petapm.c:1011-1017): `#pragma omp atomic update` per mesh point */
void pmo_cic_deposit(int64_t n, const double *pos, const double *mass, const unsigned char *live, double box, int nmesh, double *rho)
{
    const double cell = box / nmesh;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        if (live && !live[i]) continue;            /* INACTIVE(i) / RegionInd < 0 (petapm.c:964) */
        int ic[3];
        double res[3];
        for (int k = 0; k < 3; k++) {
            double tmp = pos[3 * i + k] / cell;
            double fl = floor(tmp);
            ic[k] = (int) fl;
            res[k] = tmp - fl;
        }
        const double m = mass[i];
        for (int c = 0; c < 8; c++) {
            double w = 1.0;
            size_t lin = 0;
            for (int k = 0; k < 3; k++) {
                int off = (c >> k) & 1;
                lin = lin * (size_t) nmesh + (size_t) wrap(ic[k] + off, nmesh);
                w *= off ? res[k] : (1 - res[k]);
            }
#pragma omp atomic update
            rho[lin] += w * m;
        }
    }
}

/* readout_potential / readout_force_{x,y,z} (gravpm.c:506-517): out[i * stride] += sum over the cloud of weight * mesh */
void pmo_readout(int64_t n, const double *pos, double box, int nmesh, const double *mesh, double scale, double *out, int stride)
{
    const double cell = box / nmesh;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        int ic[3];
        double res[3];
        for (int k = 0; k < 3; k++) {
            double tmp = pos[3 * i + k] / cell;
            double fl = floor(tmp);
            ic[k] = (int) fl;
            res[k] = tmp - fl;
        }
        double acc = 0;
        for (int c = 0; c < 8; c++) {
            double w = 1.0;
            size_t lin = 0;
            for (int k = 0; k < 3; k++) {
                int off = (c >> k) & 1;
                lin = lin * (size_t) nmesh + (size_t) wrap(ic[k] + off, nmesh);
                w *= off ? res[k] : (1 - res[k]);
            }
            acc += w * (mesh[lin] * scale);
        }
        out[(size_t) i * stride] += acc;
    }
}

/* gravpm.c:295-302 */
static inline double sinc_unnormed(double x)
{
    if (x < 1e-5 && x > -1e-5) {
        double x2 = x * x;
        return 1.0 - x2 / 6. + x2 * x2 / 120.;
    }
    return sin(x) / x;
}

/* petapm_mesh_to_k-style signed wavenumber of a mesh index: 0 .. N/2 stay, above N/2 become negative (petapm.c:1067-1075) */
static inline int mesh_to_k(int i, int nmesh)
{
    return i <= nmesh / 2 ? i : i - nmesh;
}

/* potential_transfer (gravpm.c:383-454) over the whole Fourier mesh, in place; no neutrino response; the power spectrum the
 * reference accumulates in the same sweep is not taken here (oracle.py::pm_power_spectrum has it) */
void pmo_potential_transfer(int nmesh, double box, double Asmth, double G, double *cplx)
{
    const int nz = nmesh / 2 + 1;
    const double asmth2 = pow((2 * M_PI) * Asmth / nmesh, 2);
    const double pot_factor = -G / (M_PI * box);
#pragma omp parallel for collapse(2) schedule(static)
    for (int ix = 0; ix < nmesh; ix++)
        for (int iy = 0; iy < nmesh; iy++) {
            const int kx = mesh_to_k(ix, nmesh), ky = mesh_to_k(iy, nmesh);
            double tx = sinc_unnormed(kx * M_PI / nmesh), ty = sinc_unnormed(ky * M_PI / nmesh);
            const double fxy = (1. / (tx * tx)) * (1. / (ty * ty));
            double *row = cplx + 2 * ((size_t) ix * nmesh + iy) * nz;
            for (int kz = 0; kz < nz; kz++) {
                const int64_t k2 = (int64_t) kx * kx + (int64_t) ky * ky + (int64_t) kz * kz;
                if (k2 == 0) {
                    row[0] = row[1] = 0.0;         /* the mean (gravpm.c:445-448) */
                    continue;
                }
                double tz = sinc_unnormed(kz * M_PI / nmesh);
                const double f = fxy * (1. / (tz * tz));
                const double smth = exp(-(double) k2 * asmth2) / (double) k2;
                const double fac = pot_factor * smth * f * f;
                row[2 * kz] *= fac;
                row[2 * kz + 1] *= fac;
            }
        }
}

/* force_{x,y,z}_transfer (gravpm.c:458-489): dst = i * fac(k_axis) * src over the whole Fourier mesh */
void pmo_force_transfer(int nmesh, double box, int axis, const double *src, double *dst)
{
    const int nz = nmesh / 2 + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int ix = 0; ix < nmesh; ix++)
        for (int iy = 0; iy < nmesh; iy++) {
            const int kx = mesh_to_k(ix, nmesh), ky = mesh_to_k(iy, nmesh);
            const double *s = src + 2 * ((size_t) ix * nmesh + iy) * nz;
            double *d = dst + 2 * ((size_t) ix * nmesh + iy) * nz;
            for (int kz = 0; kz < nz; kz++) {
                const int k = axis == 0 ? kx : axis == 1 ? ky : kz;
                const double w = k * (2 * M_PI / nmesh);
                const double fac = -1 * (1 / 6.0 * (8 * sin(w) - sin(2 * w))) * (nmesh / box);
                const double re = s[2 * kz], im = s[2 * kz + 1];
                d[2 * kz] = -im * fac;
                d[2 * kz + 1] = re * fac;
            }
        }
}
