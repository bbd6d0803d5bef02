/* oracle/timestep_oracle.c -- CPU restatement of the streaming time-integration loops either side of the force path
 * (SURVEY 8(f) row 1).  TEST INFRASTRUCTURE ONLY: used by tests/ to check the HIP kernels of csrc/timestep.hip.
 *
 * Follows the reference line by line, on plain arrays instead of struct particle_data:
 *   ots_drift_all_particles   libgadget/drift.c:18-102   (real_drift_particle / drift_all_particles)
 *   ots_apply_pm_half_kick    libgadget/timestep.c:964-985
 *   ots_apply_half_kick       libgadget/timestep.c:873-929 with do_grav_short_range_kick (:988-995) and
 *                             do_hydro_kick (:997-1036)
 * Not carried (sub-grid black-hole physics, out of scope): BH repositioning (drift.c:33-55), the dynamic-friction and drag
 * kicks of type-5 particles (timestep.c:1003-1010).
 * flags[i]: bit 0 IsGarbage, bit 1 Swallowed (the bit-field byte of struct particle_data, partmanager.h:29-40).
 * Return value: 0, or the reference's endrun() code (5: bad Hsml / non-finite position). */
#include <math.h>
#include <stdint.h>

#define TIMEBINS 46 /* timebinmgr.h:8 */

typedef struct {
    double gravkick[TIMEBINS + 1], hydrokick[TIMEBINS + 1]; /* get_exact_*kick_factor per bin; 0 for inactive bins (timestep.c:878-890) */
    double dt_entr[TIMEBINS + 1];                           /* dloga_from_dti(dti_from_timebin(bin) / 2)  (timestep.c:917) */
    uint8_t bin_active[TIMEBINS + 1];                       /* is_timebin_active(bin, Ti_Current) */
    double atime, MaxGasVel;
} ots_kick_factors;

int ots_drift_all_particles(int64_t n, double *pos, const double *vel, const uint8_t *type, const uint8_t *flags, double *hsml,
                            const double *dthsml, double ddrift, double BoxSize, const double random_shift[3])
{
    int rc = 0;
    for(int64_t i = 0; i < n; i++) {
        double *p = pos + 3 * i;
        if(flags && (flags[i] & 3)) { /* drift.c:21-30 */
            for(int j = 0; j < 3; j++) {
                p[j] += random_shift[j];
                while(p[j] > BoxSize) p[j] -= BoxSize;
                while(p[j] <= 0) p[j] += BoxSize;
            }
            continue;
        }
        if(type && type[i] == 0 && hsml) { /* drift.c:56-70 */
            hsml[i] += dthsml[i] * ddrift;
            if(hsml[i] <= 0)
                rc = 5;
            const double Maxhsml = BoxSize / 2.;
            if(hsml[i] > Maxhsml)
                hsml[i] = Maxhsml;
        }
        for(int j = 0; j < 3; j++) { /* drift.c:71-77 */
            p[j] += vel[3 * i + j] * ddrift + random_shift[j];
            if(!isfinite(p[j]))
                rc = 5;
        }
        for(int j = 0; j < 3; j++) { /* drift.c:78-81 (the reference has aborted above for a non-finite position) */
            if(!isfinite(p[j]))
                continue;
            while(p[j] > BoxSize) p[j] -= BoxSize;
            while(p[j] <= 0) p[j] += BoxSize;
        }
    }
    return rc;
}

void ots_apply_pm_half_kick(int64_t n, double *vel, const double *gravpm, const uint8_t *flags, double Fgravkick)
{
    for(int64_t i = 0; i < n; i++) {
        if(flags && (flags[i] & 3))
            continue;
        for(int j = 0; j < 3; j++)
            vel[3 * i + j] += gravpm[3 * i + j] * Fgravkick;
    }
}

int ots_apply_half_kick(int64_t n, const int *active, int64_t nactive, double *vel, const double *gravaccel, const uint8_t *type,
                        const uint8_t *flags, const uint8_t *tb_grav, const uint8_t *tb_hydro, const double *hydroaccel, double *entropy,
                        const double *dtentropy, const ots_kick_factors *K)
{
    const int64_t na = active ? nactive : n;
    int rc = 0;
    for(int64_t pa = 0; pa < na; pa++) {
        const int64_t i = active ? active[pa] : pa;
        if(flags && (flags[i] & 3))
            continue;
        const int bg = tb_grav ? tb_grav[i] : 0;
        if(bg > TIMEBINS) {
            rc = 4;
            continue;
        }
        if(K->bin_active[bg]) /* do_grav_short_range_kick */
            for(int j = 0; j < 3; j++)
                vel[3 * i + j] += gravaccel[3 * i + j] * K->gravkick[bg];
        const int ty = type ? type[i] : 1;
        if(ty == 0) { /* do_hydro_kick, gas part (timestep.c:1014-1034) */
            const int bh = tb_hydro ? tb_hydro[i] : 0;
            for(int j = 0; j < 3; j++)
                vel[3 * i + j] += hydroaccel[3 * i + j] * K->hydrokick[bh];
            double vv = 0;
            for(int j = 0; j < 3; j++)
                vv += vel[3 * i + j] * vel[3 * i + j];
            vv = sqrt(vv);
            if(vv > 0 && vv / K->atime > K->MaxGasVel)
                for(int j = 0; j < 3; j++)
                    vel[3 * i + j] *= K->MaxGasVel * K->atime / vv;
            entropy[i] += dtentropy[i] * K->dt_entr[bh];
        }
    }
    return rc;
}

/* get_timestep_gravity_dloga, timestep.c:1039-1074 (grav_acceleration2 + the gravity time-step criterion): dloga per particle */
void ots_timestep_gravity_dloga(int64_t n, const double *gravaccel, const double *gravpm, double atime, double hubble,
                                double ErrTolIntAccuracy, double force_softening, double *dloga)
{
    for(int64_t i = 0; i < n; i++) {
        const double a2inv = 1 / (atime * atime);
        double ax = a2inv * gravaccel[3 * i + 0];
        double ay = a2inv * gravaccel[3 * i + 1];
        double az = a2inv * gravaccel[3 * i + 2];
        ay += a2inv * gravpm[3 * i + 1];
        ax += a2inv * gravpm[3 * i + 0];
        az += a2inv * gravpm[3 * i + 2];
        double ac2 = ax * ax + ay * ay + az * az; /* this is now the physical acceleration */
        if(ac2 == 0)
            ac2 = 1.0e-60;
        const double ac = sqrt(ac2);
        /* mind the factor 2.8 difference between gravity and softening used here. */
        const double dt = sqrt(2 * ErrTolIntAccuracy * atime * (force_softening / 2.8) / ac);
        dloga[i] = dt * hubble; /* d a / a = dt * H */
    }
}
