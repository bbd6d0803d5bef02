/* test_cabi.c -- a C caller of libmpgadget_hip.so, as the reference's run.c would be one: plain gcc, no Python,
 * no torch.  Built and run by tests/test_gpu_cabi.py on the GPU box.
 *
 *   test_cabi single <table.f64> <pos.f64> <expect.f64> n nmesh box
 *       the drop-in (host pointer) calls on an array of 160-byte `struct particle_data` records: mpg_gravpm_force,
 *       mpg_force_tree_full, mpg_grav_short_tree (twice: Barnes-Hut opening, then the relative criterion, test_gravity.c:211-213);
 *       P[].GravPM and P[].FullTreeGravAccel against the committed vectors of tests/golden/ (expect = GravPM[n][3], Accel2[n][3]).
 *   test_cabi run <table.f64> <pos.f64> <expect.f64> n nmesh box <expect_active.f64>
 *       one rank in run.c's order with shim/forcetree-hip.c in the link ("Tree constructors"): the tree constructors
 *       only record mask / active list, no host tree exists anywhere; PM step (run.c:522-548), a refused walk after force_tree_free,
 *       and one level of the hierarchical loop (timestep.c:287-289: tree of every third particle, results in AccelStore only).
 *   test_cabi ranks|ranks_host <table.f64> <pos.f64> <expect.f64> n nmesh box NTask
 *       NTask processes (fork; the collectives of mpg_comm are implemented on a shared-memory segment with a process-shared
 *       barrier - what MPI_Allreduce / MPI_Alltoall / MPI_Alltoallv would do), every one with its own engine on GPU 0.
 *       ranks:      mpg_dist_domain_decompose + mpg_dist_domain_exchange (domain_decompose_full and the particle exchange through
 *                   the library), mpg_dist_use_decomposition, mpg_dist_gravity_step twice on device arrays;
 *       ranks_host: particles handed to the owners of the 8 top-level Peano-Hilbert cells (a legal, unbalanced domain given from
 *                   outside: mpg_dist_set_domain), then the drop-in calls mpg_dist_gravpm_force / _force_tree_full /
 *                   _grav_short_tree on each rank's table of 160-byte records, then a sub-step (every third particle active:
 *                   mpg_dist_grav_short_tree_active) that must reproduce the full walk's accelerations (to rounding: 1e-13);
 *       the assembled GravPM / accelerations against the same vectors; the matter power spectrum of the PM step
 *       (mpg_gravpm_get_powerspectrum / mpg_dist_gravpm_get_powerspectrum) is printed as a "pk:" line that must not depend on NTask.
 *   test_cabi sph <in.f64> <expect.f64> N box NTask BlackHoleOn kernel
 *       density() -> hydro_force() as shim/sph-hip.c issues them (run.c:466-489) on 160-byte records plus mpg_sph_arrays in host
 *       memory: one rank mpg_density / mpg_hydro_force, several ranks mpg_dist_force_tree_full / mpg_dist_density /
 *       mpg_dist_hydro_force (gas and, with BlackHoleOn, black holes as density targets).  in = [N][10] Pos, Type, Vel, Entropy,
 *       Hsml, Mass; expect = [N][6] Hsml, Density, HydroAccel, DtEntropy from the CPU oracle.  Also checks that a gravity walk
 *       after the SPH loops is refused until the gravity tree is rebuilt (the density loop replaces the tree in the library).
 * Built with -DMPG_TEST_MPI and shim/mpg_mpi_comm.c + shim/mpg_rccl_mpi.c in the link (tests/test_gpu_cabi.py::test_c_caller_real_mpi),
 * `ranks` / `ranks_host` run under mpiexec instead: one MPI process per rank, the collectives are the shim's MPI_Allreduce /
 * MPI_Alltoall / MPI_Alltoallv on host buffers (mpg_mpi_comm, what shim/gravity-hip.c hands to the library), or with MPG_TEST_COMM=rccl
 * and one process the RCCL communicator bootstrapped by the shim's MPI_Bcast (mpg_rccl_mpi_comm); rank 0 assembles and checks.
 * Exit code 0 and a last line "PASS ..." on success. */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include "mpgadget_hip.h"
#ifdef MPG_TEST_MPI
#include "mpg_mpi_comm.h" /* shim/ */
#endif

/* the four HIP runtime calls a device-resident caller needs (declared here so that plain gcc compiles this file) */
extern int hipMalloc(void **p, size_t n);
extern int hipFree(void *p);
extern int hipMemcpy(void *dst, const void *src, size_t n, int kind); /* 1 = host to device, 2 = device to host */
extern int hipDeviceSynchronize(void);

/* struct particle_data, libgadget/partmanager.h:9-71 (160 bytes; the fields the force path touches, the rest as padding) */
struct particle_data {
    double Pos[3];            /* 0 */
    int TopLeaf;              /* 24 */
    float Mass;               /* 28 */
    int PI;                   /* 32 */
    unsigned char flags[3];   /* 36: bit fields (IsGarbage, ...), TimeBinHydro, TimeBinGravity */
    unsigned char Type;       /* 39 */
    double Vel[3];            /* 40 */
    double FullTreeGravAccel[3]; /* 64 */
    double GravPM[3];         /* 88 */
    int64_t Ti_drift;         /* 112 */
    double Hsml, DtHsml;      /* 120, 128 */
    uint64_t ID;              /* 136 */
    int64_t GrNr;             /* 144 */
    double Potential;         /* 152 */
};
_Static_assert(sizeof(struct particle_data) == 160, "particle_data is 160 bytes");
_Static_assert(__builtin_offsetof(struct particle_data, Mass) == 28 && __builtin_offsetof(struct particle_data, Type) == 39 &&
               __builtin_offsetof(struct particle_data, FullTreeGravAccel) == 64 && __builtin_offsetof(struct particle_data, GravPM) == 88 &&
               __builtin_offsetof(struct particle_data, Potential) == 152, "particle_data layout");

#define CK(call)                                                                    \
    do {                                                                            \
        if((call) != 0) {                                                           \
            fprintf(stderr, "FAIL %s:%d %s: %s\n", __FILE__, __LINE__, #call, mpg_last_error()); \
            exit(1);                                                                \
        }                                                                           \
    } while(0)

static double *read_f64(const char *path, size_t count)
{
    FILE *f = fopen(path, "rb");
    if(!f) {
        fprintf(stderr, "FAIL cannot open %s\n", path);
        exit(1);
    }
    double *d = malloc(count * sizeof(double));
    if(fread(d, sizeof(double), count, f) != count) {
        fprintf(stderr, "FAIL short read of %s\n", path);
        exit(1);
    }
    fclose(f);
    return d;
}

static mpg_engine *make_engine(const double *table, double box, int n, int nmesh)
{
    mpg_engine *e = NULL;
    CK(mpg_engine_create(&e, 0));
    CK(mpg_gravshort_fill_ntab(e, 0, 1.5, table, 512));
    CK(mpg_gravpm_init_periodic(e, box, 1.5, nmesh, 43.0071));
    mpg_gravshort_tree_params tp = {0.002, 0.175, 0.9, 2, 6.0, 1.0 / 30.};
    CK(mpg_set_gravshort_treepar(e, &tp));
    CK(mpg_gravshort_set_softenings(e, box / n));
    return e;
}

/* max |a - b| / mean |b| over count values */
static double relerr(const double *a, const double *b, size_t count)
{
    double mx = 0, mean = 0;
    for(size_t i = 0; i < count; i++) {
        const double d = fabs(a[i] - b[i]);
        mx = d > mx ? d : mx;
        mean += fabs(b[i]);
    }
    return mx / (mean / count);
}

static int run_single(const double *table, const double *pos, const double *expect, int n, int nmesh, double box)
{
    const int64_t N = (int64_t)n * n * n;
    struct particle_data *P = calloc(N, sizeof(struct particle_data));
    for(int64_t i = 0; i < N; i++) {
        memcpy(P[i].Pos, pos + 3 * i, 3 * sizeof(double));
        P[i].Mass = 1.0f;
        P[i].Type = 1;
        P[i].ID = (uint64_t)i;
    }
    mpg_engine *e = make_engine(table, box, n, nmesh);
    mpg_particle_view v;
    mpg_particle_view_reference_layout(&v, P, N);
    CK(mpg_gravpm_force(e, &v));
    {
        double *kk = malloc(2 * nmesh * sizeof(double)), *pw = kk + nmesh, sp = 0;
        int64_t *nm = malloc(nmesh * sizeof(int64_t)), sn = 0;
        int nz = 0;
        CK(mpg_gravpm_get_powerspectrum(e, box / 1000., kk, pw, nm, &nz));
        for(int i = 0; i < nz; i++) {
            sp += pw[i] * (double)nm[i];
            sn += nm[i];
        }
        printf("pk: bins %d modes %lld sumPN %.12e k0 %.12e\n", nz, (long long)sn, sp, nz ? kk[0] : 0.0);
        free(nm);
        free(kk);
    }
    CK(mpg_force_tree_full(e, &v, box));
    CK(mpg_grav_short_tree(e, &v, NULL, 0, NULL, 0.0)); /* Barnes-Hut opening */
    CK(mpg_grav_short_tree(e, &v, NULL, 0, NULL, 0.0)); /* relative criterion, OldAcc = |FullTreeGravAccel + GravPM| / G */
    double *gpm = malloc(3 * N * sizeof(double)), *acc = malloc(3 * N * sizeof(double));
    for(int64_t i = 0; i < N; i++)
        for(int k = 0; k < 3; k++) {
            gpm[3 * i + k] = P[i].GravPM[k];
            acc[3 * i + k] = P[i].FullTreeGravAccel[k];
        }
    const double e_pm = relerr(gpm, expect, 3 * N), e_tr = relerr(acc, expect + 3 * N, 3 * N);
    mpg_engine_destroy(e);
    printf("single: N %lld  GravPM err %.3e  FullTreeGravAccel err %.3e\n", (long long)N, e_pm, e_tr);
    if(!(e_pm < 1e-10 && e_tr < 1e-10)) {
        printf("FAIL\n");
        return 1;
    }
    printf("PASS single\n");
    return 0;
}

/* ---- one rank, the order of run.c with shim/forcetree-hip.c in the link: the tree constructors only record what was asked for, no
 * host tree exists at any point, the consumer builds the device tree from the record ----------------------------------------- */
enum { TREE_NONE = 0, TREE_FULL = 1, TREE_ACTIVE = 2 };
struct tree_record { /* what forcetree-hip.c keeps per ForceTree (shim/mpg_shim.h: struct mpg_deferred_tree) */
    int kind, mask, HybridNuTracer;
    const int *ActiveParticle;
    int64_t NumActiveParticle;
};
/* grav_short_tree of shim/gravity-hip.c, NTask == 1 */
static int shim_grav_short_tree(mpg_engine *e, const mpg_particle_view *v, double box, const struct tree_record *t, const int *act, int64_t nact,
                                double (*AccelStore)[3])
{
    if(t->kind == TREE_NONE)
        return mpg_grav_short_tree(e, v, act, nact, AccelStore, 0.0); /* (must fail: "Gravtree called before tree moments computed") */
    if(t->kind == TREE_ACTIVE && t->ActiveParticle)
        CK(mpg_force_tree_active_moments(e, v, box, t->ActiveParticle, t->NumActiveParticle, t->HybridNuTracer));
    else
        CK(mpg_force_tree_rebuild_mask(e, v, box, t->mask));
    return mpg_grav_short_tree(e, v, act, nact, AccelStore, 0.0);
}

static int run_order(const double *table, const double *pos, const double *expect, const char *expect_active_path, int n, int nmesh, double box)
{
    const int64_t N = (int64_t)n * n * n;
    struct particle_data *P = calloc(N, sizeof(struct particle_data));
    for(int64_t i = 0; i < N; i++) {
        memcpy(P[i].Pos, pos + 3 * i, 3 * sizeof(double));
        P[i].Mass = 1.0f;
        P[i].Type = 1;
        P[i].ID = (uint64_t)i;
    }
    mpg_engine *e = make_engine(table, box, n, nmesh);
    mpg_particle_view v;
    mpg_particle_view_reference_layout(&v, P, N);
    int64_t epoch = 1;
    CK(mpg_set_particle_epoch(e, epoch)); /* mpg_shim_sync: one upload serves the calls of this step */
    /* ---- a PM step, run.c:522-548 ---- */
    CK(mpg_gravpm_force(e, &v));
    struct tree_record Tree = {TREE_FULL, 63, 0, NULL, 0};   /* force_tree_full(&Tree, ...): recorded, nothing built */
    CK(shim_grav_short_tree(e, &v, box, &Tree, NULL, 0, NULL));
    Tree = (struct tree_record){TREE_NONE, 0, 0, NULL, 0};    /* force_tree_free(&Tree) */
    CK(mpg_force_tree_free(e));
    if(shim_grav_short_tree(e, &v, box, &Tree, NULL, 0, NULL) == 0) {
        printf("FAIL a walk after force_tree_free was not refused\n");
        return 1;
    }
    /* second walk of the start-up sequence (relative criterion; test_gravity.c:211-213) */
    Tree = (struct tree_record){TREE_FULL, 63, 0, NULL, 0};
    CK(shim_grav_short_tree(e, &v, box, &Tree, NULL, 0, NULL));
    CK(mpg_force_tree_free(e));
    double *gpm = malloc(3 * N * sizeof(double)), *acc = malloc(3 * N * sizeof(double));
    for(int64_t i = 0; i < N; i++)
        for(int k = 0; k < 3; k++) {
            gpm[3 * i + k] = P[i].GravPM[k];
            acc[3 * i + k] = P[i].FullTreeGravAccel[k];
        }
    const double e_pm = relerr(gpm, expect, 3 * N), e_tr = relerr(acc, expect + 3 * N, 3 * N);
    printf("run: N %lld  GravPM err %.3e  FullTreeGravAccel err %.3e\n", (long long)N, e_pm, e_tr);
    /* ---- a level of the hierarchical gravity loop, timestep.c:287-289: force_tree_active_moments(&Tree, subact) records the list;
     * grav_short_tree walks the tree of those particles for those particles into AccelStore and leaves P[] alone (gravshort.h:57-67) ---- */
    int64_t nact = 0;
    int *act = malloc(N * sizeof(int));
    for(int64_t i = 0; i < N; i += 3)
        act[nact++] = (int)i;
    double *exa = read_f64(expect_active_path, 3 * (size_t)nact);
    double(*store)[3] = calloc(N, sizeof(double[3]));
    Tree = (struct tree_record){TREE_ACTIVE, 63, 0, act, nact};
    CK(shim_grav_short_tree(e, &v, box, &Tree, act, nact, store));
    CK(mpg_force_tree_free(e));
    double *got = malloc(3 * nact * sizeof(double));
    int touched = 0;
    for(int64_t k = 0; k < nact; k++)
        for(int j = 0; j < 3; j++)
            got[3 * k + j] = store[act[k]][j];
    for(int64_t i = 0; i < N; i++)
        for(int k = 0; k < 3; k++)
            touched |= (P[i].FullTreeGravAccel[k] != acc[3 * i + k]);
    const double e_act = relerr(got, exa, 3 * nact);
    printf("run: active-only tree of %lld particles  AccelStore err %.3e  P[] touched %d\n", (long long)nact, e_act, touched);
    mpg_engine_destroy(e);
    if(!(e_pm < 1e-10 && e_tr < 1e-10 && e_act < 1e-10 && !touched)) {
        printf("FAIL\n");
        return 1;
    }
    printf("PASS run\n");
    return 0;
}

/* ---- NTask processes on one box: the collectives of mpg_comm on shared memory ---------------------------------------------- */
#define MAXT 8
struct shm {
    pthread_barrier_t bar;
    int64_t cnt[MAXT][MAXT];
    int64_t off[MAXT][MAXT];
    size_t cap;         /* bytes of data[] per rank */
    int64_t N;
    int failed;
    int substep_bad;    /* a rank's sub-step check failed (ranks_host) */
    /* followed by: data[MAXT][cap], result[N][6] */
};
struct ctx {
    struct shm *S;
    int me, nt;
};
static char *shm_data(struct shm *S, int r) { return (char *)(S + 1) + (size_t)r * S->cap; }
static double *shm_result(struct shm *S) { return (double *)((char *)(S + 1) + (size_t)MAXT * S->cap); }

static int cb_allreduce(void *c_, void *buf, int64_t count, int dtype, int op, int on_device)
{
    struct ctx *c = c_;
    if(on_device || (size_t)count * 8 > c->S->cap)
        return 1;
    memcpy(shm_data(c->S, c->me), buf, (size_t)count * 8);
    pthread_barrier_wait(&c->S->bar);
    for(int64_t i = 0; i < count; i++) {
        if(dtype) {
            int64_t a = ((int64_t *)shm_data(c->S, 0))[i];
            for(int r = 1; r < c->nt; r++) {
                const int64_t b = ((int64_t *)shm_data(c->S, r))[i];
                a = op ? (b > a ? b : a) : a + b;
            }
            ((int64_t *)buf)[i] = a;
        }
        else {
            double a = ((double *)shm_data(c->S, 0))[i];
            for(int r = 1; r < c->nt; r++) {
                const double b = ((double *)shm_data(c->S, r))[i];
                a = op ? (b > a ? b : a) : a + b;
            }
            ((double *)buf)[i] = a;
        }
    }
    pthread_barrier_wait(&c->S->bar);
    return 0;
}

static int cb_alltoall_i64(void *c_, const int64_t *send, int64_t *recv)
{
    struct ctx *c = c_;
    for(int d = 0; d < c->nt; d++)
        c->S->cnt[c->me][d] = send[d];
    pthread_barrier_wait(&c->S->bar);
    for(int s = 0; s < c->nt; s++)
        recv[s] = c->S->cnt[s][c->me];
    pthread_barrier_wait(&c->S->bar);
    return 0;
}

static int cb_alltoallv(void *c_, const void *send, const int64_t *sb, const int64_t *sd, void *recv, const int64_t *rb, const int64_t *rd,
                        int on_device)
{
    struct ctx *c = c_;
    if(on_device)
        return 1;
    size_t o = 0;
    for(int d = 0; d < c->nt; d++) {
        if(o + (size_t)sb[d] > c->S->cap)
            return 1;
        memcpy(shm_data(c->S, c->me) + o, (const char *)send + sd[d], (size_t)sb[d]);
        c->S->off[c->me][d] = (int64_t)o;
        c->S->cnt[c->me][d] = sb[d];
        o += (size_t)sb[d];
    }
    pthread_barrier_wait(&c->S->bar);
    int bad = 0;
    for(int s = 0; s < c->nt; s++) {
        if(c->S->cnt[s][c->me] != rb[s])
            bad = 1;
        else
            memcpy((char *)recv + rd[s], shm_data(c->S, s) + c->S->off[s][c->me], (size_t)rb[s]);
    }
    pthread_barrier_wait(&c->S->bar);
    return bad;
}

static void rank_main(struct shm *S, int me, int nt, const double *table, const double *pos, int n, int nmesh, double box, int host)
{
    const int64_t N = S->N;
    alarm(300); /* a rank that dies leaves its peers at a barrier: do not hang the test */
    struct ctx cx = {S, me, nt};
    mpg_comm comm = {&cx, me, nt, 0, cb_allreduce, cb_alltoall_i64, cb_alltoallv};
    /* MPG_TEST_COMM=rccl: the library's native RCCL communicator instead (mpg_rccl_*, csrc/rccl_comm.hip), bootstrapped as a C caller
     * does it - rank 0's unique id handed to every rank (here: one rank; RCCL refuses several ranks on one GPU), mpg_rccl_create, the
     * self-test, the callbacks.  Collectives then run on the engine's stream with device pointers, no host staging. */
    mpg_rccl *RC = NULL;
#ifdef MPG_TEST_MPI
    /* real MPI processes: the shim's communicator (shim/mpg_mpi_comm.c), or - one process - its RCCL bootstrap (shim/mpg_rccl_mpi.c) */
    static MPI_Comm world;
    world = MPI_COMM_WORLD;
    comm = mpg_mpi_comm(&world);
    if(comm.ThisTask != me || comm.NTask != nt) {
        fprintf(stderr, "FAIL mpg_mpi_comm reports task %d of %d, expected %d of %d\n", comm.ThisTask, comm.NTask, me, nt);
        exit(1);
    }
    if(getenv("MPG_TEST_COMM") && !strcmp(getenv("MPG_TEST_COMM"), "rccl")) {
        if(nt != 1) {
            fprintf(stderr, "FAIL RCCL refuses several ranks on one GPU: one MPI process with MPG_TEST_COMM=rccl\n");
            exit(1);
        }
        if(mpg_rccl_mpi_comm(world, 0, &RC, &comm)) {
            fprintf(stderr, "FAIL mpg_rccl_mpi_comm\n");
            exit(1);
        }
    }
#else
    if(getenv("MPG_TEST_COMM") && !strcmp(getenv("MPG_TEST_COMM"), "rccl")) {
        if(nt != 1) {
            fprintf(stderr, "FAIL RCCL refuses several ranks on one GPU: NTask must be 1 with MPG_TEST_COMM=rccl\n");
            exit(1);
        }
        char id[MPG_RCCL_ID_BYTES];
        CK(mpg_rccl_get_unique_id(id));
        CK(mpg_rccl_create(&RC, me, nt, id, 0));
        CK(mpg_rccl_selftest(RC, 0));
        CK(mpg_rccl_comm(RC, &comm));
    }
#endif
    mpg_engine *e = make_engine(table, box, n, nmesh);
    /* the domain: the root of the Peano-Hilbert key space cut into its 8 cells, cell k owned by task k % NTask
     * (struct topnode_data, domain.h:12-18: StartKey, Shift, Daughter, Leaf) */
    mpg_topnode tn[9];
    int leaf_task[8];
    memset(tn, 0, sizeof(tn));
    tn[0].StartKey = 0;
    tn[0].Shift = 63;
    tn[0].Daughter = 1;
    tn[0].Parent = -1;
    tn[0].Leaf = -1;
    for(int k = 0; k < 8; k++) {
        tn[1 + k].StartKey = (uint64_t)k << 60;
        tn[1 + k].Shift = 60;
        tn[1 + k].Daughter = -1;
        tn[1 + k].Parent = 0;
        tn[1 + k].Leaf = k;
        leaf_task[k] = k % nt;
    }
    mpg_dist *D = NULL;
    CK(mpg_dist_create(&D, e, &comm));
    int64_t n_own = 0;
    int64_t *ids = malloc(N * sizeof(int64_t));
    double *opos = malloc(3 * (N + 1) * sizeof(double));
    float *omass = malloc((N + 1) * sizeof(float));
    if(host) {
        /* own particles: those whose key falls into my cells (keys from the engine: PEANO(), peano.h:15-21) */
        double *d_all = NULL;
        uint64_t *d_keys = NULL, *keys = malloc(N * sizeof(uint64_t));
        if(hipMalloc((void **)&d_all, 3 * N * sizeof(double)) || hipMalloc((void **)&d_keys, N * sizeof(uint64_t)) ||
           hipMemcpy(d_all, pos, 3 * N * sizeof(double), 1)) {
            fprintf(stderr, "FAIL hipMalloc\n");
            exit(1);
        }
        CK(mpg_dev_peano_keys(e, N, d_all, box, d_keys));
        CK(mpg_engine_synchronize(e));
        hipMemcpy(keys, d_keys, N * sizeof(uint64_t), 2);
        for(int64_t i = 0; i < N; i++)
            if(leaf_task[keys[i] >> 60] == me)
                ids[n_own++] = i;
        for(int64_t k = 0; k < n_own; k++)
            memcpy(opos + 3 * k, pos + 3 * ids[k], 3 * sizeof(double));
        free(keys);
        hipFree(d_all);
        hipFree(d_keys);
    }
    else {
        /* domain_decompose_full + domain_exchange through the library (mpg_dist_domain_*): every rank starts from a contiguous
         * share of the particle set, as after reading a snapshot */
        const int64_t lo = N * me / nt, hi = N * (me + 1) / nt, ns = hi - lo;
        double *d_sp = NULL;
        int64_t *d_sid = NULL, *sid = malloc((ns + 1) * sizeof(int64_t));
        for(int64_t i = 0; i < ns; i++)
            sid[i] = lo + i;
        if(hipMalloc((void **)&d_sp, 3 * (ns + 1) * sizeof(double)) || hipMalloc((void **)&d_sid, (ns + 1) * sizeof(int64_t)) ||
           hipMemcpy(d_sp, pos + 3 * lo, 3 * ns * sizeof(double), 1) || hipMemcpy(d_sid, sid, ns * sizeof(int64_t), 1)) {
            fprintf(stderr, "FAIL hipMalloc\n");
            exit(1);
        }
        int ntn = 0, ntl = 0;
        CK(mpg_dist_domain_decompose(D, ns, d_sp, NULL, box, 4, 1, NULL, &ntn, &ntl));
        const void *cols[2] = {d_sp, d_sid};
        const int widths[2] = {24, 8};
        void *newc[2];
        CK(mpg_dist_domain_exchange(D, ns, 2, cols, widths, &n_own, newc));
        hipMemcpy(opos, newc[0], 3 * n_own * sizeof(double), 2);
        hipMemcpy(ids, newc[1], n_own * sizeof(int64_t), 2);
        if(me == 0)
            printf("decomposition: %d TopNodes, %d TopLeaves\n", ntn, ntl);
        free(sid);
        hipFree(d_sp);
        hipFree(d_sid);
    }
    for(int64_t k = 0; k < n_own; k++)
        omass[k] = 1.0f;
    double *d_pos, *d_acc, *d_prev, *d_gpm, *d_pot;
    float *d_mass;
    const size_t b3 = 3 * (n_own + 1) * sizeof(double);
    if(hipMalloc((void **)&d_pos, b3) || hipMalloc((void **)&d_acc, b3) || hipMalloc((void **)&d_prev, b3) || hipMalloc((void **)&d_gpm, b3) ||
       hipMalloc((void **)&d_pot, b3) || hipMalloc((void **)&d_mass, (n_own + 1) * sizeof(float))) {
        fprintf(stderr, "FAIL hipMalloc\n");
        exit(1);
    }
    hipMemcpy(d_pos, opos, 3 * n_own * sizeof(double), 1);
    hipMemcpy(d_mass, omass, n_own * sizeof(float), 1);
    const double rcut = 6.0 * 1.5 * box / nmesh; /* Rcut * Asmth * cell size, gravshort-tree.c:102 */
    if(host)
        CK(mpg_dist_set_domain(D, box, tn, 9, leaf_task, 8, rcut, 0));
    else
        CK(mpg_dist_use_decomposition(D, box, rcut, 0));
    double *acc = malloc(b3), *gpm = malloc(b3);
    if(host) {
        /* the drop-in calls on this rank's table of 160-byte records, in run.c's order; twice (Barnes-Hut, then relative).
         * MPG_TEST_GARBAGE=<percent>: the table also holds that many garbage / swallowed records between the live ones (heavy, next to
         * live particles: any leak into the mesh, a tree or a target list shows in the forces) - what P[] looks like in the sub-steps
         * after star formation or a black-hole merger, until the next domain_decompose_full collects them.  The reference skips them
         * in place (treewalk.c:234, forcetree.c:806, gravpm.c:176-179); the expected forces of the live particles do not change. */
        const int gpct = getenv("MPG_TEST_GARBAGE") ? atoi(getenv("MPG_TEST_GARBAGE")) : 0;
        const int64_t gevery = gpct > 0 ? 100 / gpct : 0;
        const int64_t n_tab = n_own + (gevery ? n_own / gevery : 0);
        struct particle_data *P = calloc(n_tab + 1, sizeof(struct particle_data));
        int64_t *tab = malloc((n_own + 1) * sizeof(int64_t)); /* live particle k sits at P[tab[k]] */
        unsigned char *dead = calloc(n_tab + 1, 1);
        {
            int64_t t = 0;
            for(int64_t k = 0; k < n_own; k++) {
                if(gevery && k % gevery == gevery - 1) {
                    memcpy(P[t].Pos, opos + 3 * k, 3 * sizeof(double));
                    P[t].Pos[0] += 1e-3 * box / n;
                    P[t].Mass = 50.0f;
                    P[t].ID = (uint64_t)(N + t);
                    if((k / gevery) & 1) { /* a swallowed black hole (partmanager.h:33-37: Swallowed is bit 1 of the flag byte) */
                        P[t].Type = 5;
                        P[t].flags[0] = 2;
                    }
                    else { /* IsGarbage: bit 0 */
                        P[t].Type = 1;
                        P[t].flags[0] = 1;
                    }
                    P[t].GravPM[0] = 7.0; /* (must come back zeroed: gravpm.c:88-92) */
                    P[t].FullTreeGravAccel[1] = 9.0; /* (must come back untouched) */
                    dead[t++] = 1;
                }
                tab[k] = t;
                memcpy(P[t].Pos, opos + 3 * k, 3 * sizeof(double));
                P[t].Mass = 1.0f;
                P[t].Type = 1;
                P[t].ID = (uint64_t)ids[k];
                t++;
            }
            if(t != n_tab) {
                fprintf(stderr, "FAIL table construction\n");
                exit(1);
            }
        }
        mpg_particle_view v;
        mpg_particle_view_reference_layout(&v, P, n_tab);
        double *prev = malloc((3 * n_tab + 3) * sizeof(double));
        for(int it = 0; it < 2; it++) {
            for(int64_t t = 0; t < n_tab; t++)
                memcpy(prev + 3 * t, P[t].FullTreeGravAccel, 3 * sizeof(double));
            CK(mpg_dist_gravpm_force(D, &v));
            if(it == 0) { /* gravpm.c:110-118 on several ranks: the slab sums all-reduced, the same spectrum on every rank */
                double *kk = malloc(2 * nmesh * sizeof(double)), *pw = kk + nmesh, sp = 0;
                int64_t *nm = malloc(nmesh * sizeof(int64_t)), sn = 0;
                int nz = 0;
                CK(mpg_dist_gravpm_get_powerspectrum(D, box / 1000., kk, pw, nm, &nz));
                for(int i = 0; i < nz; i++) {
                    sp += pw[i] * (double)nm[i];
                    sn += nm[i];
                }
                if(me == nt - 1) /* (any rank: the call is collective and returns the summed spectrum everywhere) */
                    printf("pk: bins %d modes %lld sumPN %.12e k0 %.12e\n", nz, (long long)sn, sp, nz ? kk[0] : 0.0);
                free(nm);
                free(kk);
            }
            CK(mpg_dist_force_tree_full(D, &v));
            CK(mpg_dist_grav_short_tree(D, &v, NULL, 0.0));
        }
        for(int64_t k = 0; k < n_own; k++)
            for(int j = 0; j < 3; j++) {
                gpm[3 * k + j] = P[tab[k]].GravPM[j];
                acc[3 * k + j] = P[tab[k]].FullTreeGravAccel[j];
            }
        for(int64_t t = 0; t < n_tab; t++)
            if(dead[t] && !(P[t].GravPM[0] == 0.0 && P[t].GravPM[1] == 0.0 && P[t].FullTreeGravAccel[1] == 9.0 && P[t].FullTreeGravAccel[0] == 0.0)) {
                fprintf(stderr, "rank %d: FAIL a garbage record was given a force (GravPM %g, accel %g %g)\n", me, P[t].GravPM[0],
                        P[t].FullTreeGravAccel[0], P[t].FullTreeGravAccel[1]);
                S->substep_bad = 1;
                break;
            }
        /* a sub-step on the same tree: every third record active (mpg_dist_grav_short_tree_active; garbage on the list is skipped in
         * place), P[] put back to what the second walk saw.  The active particles must get that walk's accelerations (same tree, same
         * OldAcc, same lists), in P[] and in AccelStore; the others keep what they had. */
        {
            int64_t nact = 0, nbad = 0;
            int *act = malloc((n_tab + 1) * sizeof(int));
            double(*store)[3] = calloc(n_tab + 1, sizeof(*store));
            double *full = malloc((3 * n_tab + 3) * sizeof(double)); /* the second walk's result per record */
            for(int64_t t = 0; t < n_tab; t++) {
                memcpy(full + 3 * t, P[t].FullTreeGravAccel, 3 * sizeof(double));
                memcpy(P[t].FullTreeGravAccel, prev + 3 * t, 3 * sizeof(double));
                if(t % 3 == 0)
                    act[nact++] = (int)t;
            }
            CK(mpg_dist_grav_short_tree_active(D, &v, act, nact, store, 0.0));
            /* (the inactive particles keep what they had, bit for bit; the active ones get the full walk's accelerations to rounding:
             * with k_walk_lists8 the ORDER of a target's list entries - not the entries - depends on the 7 targets that share its
             * wave, and a sub-step groups other targets together) */
            double amax = 0;
            for(int64_t k = 0; k < 3 * n_own; k++)
                amax = fmax(amax, fabs(acc[k]));
            for(int64_t t = 0; t < n_tab; t++) {
                const int walked = (t % 3 == 0) && !dead[t];
                const double *want = walked ? full + 3 * t : prev + 3 * t;
                for(int j = 0; j < 3; j++) {
                    const double tol = walked ? 1e-13 * amax : 0.0;
                    if(!(fabs(P[t].FullTreeGravAccel[j] - want[j]) <= tol))
                        nbad++;
                    if(walked ? !(fabs(store[t][j] - want[j]) <= tol) : store[t][j] != 0)
                        nbad++;
                }
            }
            if(nbad) {
                fprintf(stderr, "rank %d: FAIL sub-step: %lld entries differ from the full walk\n", me, (long long)nbad);
                S->substep_bad = 1;
            }
            free(act);
            free(store);
            free(full);
        }
        if(gevery && me == 0)
            printf("garbage: %lld of %lld records of rank 0 are garbage / swallowed\n", (long long)(n_tab - n_own), (long long)n_tab);
        free(tab);
        free(dead);
        free(prev);
        free(P);
    }
    else {
        CK(mpg_dist_gravity_step(D, n_own, d_pos, d_mass, NULL, NULL, d_prev, d_gpm, d_pot, 0.0));  /* Barnes-Hut opening */
        CK(mpg_dist_gravity_step(D, n_own, d_pos, d_mass, NULL, d_prev, d_acc, d_gpm, d_pot, 0.0)); /* relative criterion */
        hipMemcpy(acc, d_acc, 3 * n_own * sizeof(double), 2);
        hipMemcpy(gpm, d_gpm, 3 * n_own * sizeof(double), 2);
    }
    int64_t st[8];
    CK(mpg_dist_get_stats(D, st));
    double *R = shm_result(S);
    for(int64_t k = 0; k < n_own; k++)
        for(int j = 0; j < 3; j++) {
            R[6 * ids[k] + j] = gpm[3 * k + j];
            R[6 * ids[k] + 3 + j] = acc[3 * k + j];
        }
    printf("rank %d of %d: own %lld ghosts %lld local %lld La %lld\n", me, nt, (long long)n_own, (long long)st[0], (long long)st[2], (long long)st[3]);
    if(RC) {
        int64_t calls[3], sent = 0;
        int ver = 0;
        CK(mpg_rccl_stats(RC, calls, &sent, &ver));
        printf("rccl: version %d allreduce %lld alltoall_i64 %lld alltoallv %lld\n", ver, (long long)calls[0], (long long)calls[1], (long long)calls[2]);
        if(calls[0] < 1 || calls[2] < 4) {
            fprintf(stderr, "FAIL the collectives did not go through the RCCL communicator\n");
            exit(1);
        }
    }
    fflush(stdout);
    mpg_dist_destroy(D);
    mpg_engine_destroy(e);
    mpg_rccl_destroy(RC);
    pthread_barrier_wait(&S->bar);
}

/* ---- density() -> hydro_force() through the host drop-in forms, one or several ranks ---------------------------------------- */
struct sph_in {
    double Pos[3], Type, Vel[3], Entropy, Hsml, Mass;
};

static void sph_rank(struct shm *S, int me, int nt, const struct sph_in *in, double box, int bh, int kernel)
{
    const int64_t N = S->N;
    alarm(300);
    struct ctx cx = {S, me, nt};
    mpg_comm comm = {&cx, me, nt, 0, cb_allreduce, cb_alltoall_i64, cb_alltoallv};
#ifdef MPG_TEST_MPI
    static MPI_Comm world;
    world = MPI_COMM_WORLD;
    comm = mpg_mpi_comm(&world); /* shim/mpg_mpi_comm.c */
#endif
    mpg_engine *e = NULL;
    CK(mpg_engine_create(&e, 0));
    const double meansep = box / cbrt((double)N);
    mpg_gravshort_tree_params tp = {0.002, 0.175, 0.9, 2, 6.0, 1.0 / 30.};
    CK(mpg_set_gravshort_treepar(e, &tp));
    CK(mpg_gravshort_set_softenings(e, meansep));
    mpg_density_params dp = {1.0, 2.0, 2.0, 99999., kernel, 0.006};
    CK(mpg_set_densitypar(e, &dp));
    mpg_hydro_params hp = {0, 100.0, 0.75};
    CK(mpg_set_hydropar(e, &hp));
    /* my particles: by the top-level Peano-Hilbert cell of the position, cell k owned by task k % NTask */
    int64_t n_own = 0, *ids = malloc(N * sizeof(int64_t));
    int leaf_task[8];
    for(int k = 0; k < 8; k++)
        leaf_task[k] = k % nt;
    {
        double *pos = malloc(3 * N * sizeof(double)), *d_all = NULL;
        uint64_t *d_keys = NULL, *keys = malloc(N * sizeof(uint64_t));
        for(int64_t i = 0; i < N; i++)
            memcpy(pos + 3 * i, in[i].Pos, 3 * sizeof(double));
        if(hipMalloc((void **)&d_all, 3 * N * sizeof(double)) || hipMalloc((void **)&d_keys, N * sizeof(uint64_t)) ||
           hipMemcpy(d_all, pos, 3 * N * sizeof(double), 1)) {
            fprintf(stderr, "FAIL hipMalloc\n");
            exit(1);
        }
        CK(mpg_dev_peano_keys(e, N, d_all, box, d_keys));
        CK(mpg_engine_synchronize(e));
        hipMemcpy(keys, d_keys, N * sizeof(uint64_t), 2);
        for(int64_t i = 0; i < N; i++)
            if(leaf_task[keys[i] >> 60] == me)
                ids[n_own++] = i;
        free(keys);
        free(pos);
        hipFree(d_all);
        hipFree(d_keys);
    }
    struct particle_data *P = calloc(n_own + 1, sizeof(struct particle_data));
    const size_t n1 = (size_t)n_own + 1;
    double *hs = calloc(n1, 8), *dth = calloc(n1, 8), *vel = calloc(3 * n1, 8), *ent = calloc(n1, 8), *den = calloc(n1, 8), *egy = calloc(n1, 8),
           *dhs = calloc(n1, 8), *dv = calloc(n1, 8), *cv = calloc(n1, 8), *hacc = calloc(3 * n1, 8), *dte = calloc(n1, 8), *msv = calloc(n1, 8);
    double hmax = 0;
    for(int64_t k = 0; k < n_own; k++) {
        const struct sph_in *q = &in[ids[k]];
        memcpy(P[k].Pos, q->Pos, 3 * sizeof(double));
        P[k].Mass = (float)q->Mass;
        P[k].Type = (unsigned char)q->Type;
        P[k].ID = (uint64_t)ids[k];
        P[k].Hsml = q->Hsml;
        memcpy(P[k].Vel, q->Vel, 3 * sizeof(double));
        hs[k] = q->Hsml;
        memcpy(vel + 3 * k, q->Vel, 3 * sizeof(double));
        ent[k] = q->Entropy;
        if(q->Hsml > hmax)
            hmax = q->Hsml;
    }
    mpg_particle_view v;
    mpg_particle_view_reference_layout(&v, P, n_own);
    mpg_sph_arrays A;
    memset(&A, 0, sizeof(A));
    A.hsml = hs, A.dthsml = dth, A.vel = vel, A.entropy = ent, A.density = den, A.egywtdensity = egy, A.dhsmlegyfac = dhs, A.divvel = dv, A.curlvel = cv;
    A.hydroacc_out = hacc, A.dtentropy_out = dte, A.maxsignalvel = msv;
    mpg_sph_times T;
    memset(&T, 0, sizeof(T));
    T.atime = 1.0;
    T.hubble = 0.1;
    mpg_dist *D = NULL;
    if(nt == 1) {
        CK(mpg_density(e, &v, box, &A, &T, NULL, 0, 1, 0, bh));
        CK(mpg_hydro_force(e, &v, &A, &T, NULL, 0));
    }
    else {
        mpg_topnode tn[9];
        memset(tn, 0, sizeof(tn));
        tn[0].Shift = 63, tn[0].Daughter = 1, tn[0].Parent = -1, tn[0].Leaf = -1;
        for(int k = 0; k < 8; k++) {
            tn[1 + k].StartKey = (uint64_t)k << 60;
            tn[1 + k].Shift = 60, tn[1 + k].Daughter = -1, tn[1 + k].Parent = 0, tn[1 + k].Leaf = k;
        }
        CK(mpg_dist_create(&D, e, &comm));
        /* the ghost margin: every neighbour within a smoothing length must be local - deliberately too small at first, so that the
         * library's check fires and the retry of shim/sph-hip.c (margin from mpg_dist_last_max_hsml) is exercised */
        cb_allreduce(&cx, &hmax, 1, 0, 1, 0);
        double margin = 0.8 * hmax;
        CK(mpg_dist_set_sph_options(D, bh));
        int attempt = 0;
        for(;; attempt++) {
            CK(mpg_dist_set_domain(D, box, tn, 9, leaf_task, 8, margin, 0));
            CK(mpg_dist_force_tree_full(D, &v));
            if(mpg_dist_density(D, &v, &A, &T, NULL, 0, 1, 0) == 0)
                break;
            const double need = mpg_dist_last_max_hsml(D);
            if(attempt >= 3 || !(need > margin)) {
                fprintf(stderr, "FAIL mpg_dist_density: %s\n", mpg_last_error());
                exit(1);
            }
            margin = 1.26 * need;
        }
        if(attempt == 0) {
            fprintf(stderr, "FAIL the margin check of mpg_dist_density did not fire (0.8 of the largest initial Hsml)\n");
            exit(1);
        }
        CK(mpg_dist_hydro_force(D, &v, &A, &T, NULL, 0));
        /* the density loop put a gas tree in place of the gravity tree: a walk now must be refused, not run on the gas tree */
        if(mpg_dist_grav_short_tree(D, &v, NULL, 0.0) == 0 || !strstr(mpg_last_error(), "replaced")) {
            fprintf(stderr, "FAIL a gravity walk on the gas tree was not refused (%s)\n", mpg_last_error());
            exit(1);
        }
    }
    double *R = shm_result(S);
    for(int64_t k = 0; k < n_own; k++) {
        double *r = R + 6 * ids[k];
        r[0] = hs[k], r[1] = den[k], r[2] = hacc[3 * k], r[3] = hacc[3 * k + 1], r[4] = hacc[3 * k + 2], r[5] = dte[k];
    }
    printf("sph rank %d of %d: own %lld\n", me, nt, (long long)n_own);
    fflush(stdout);
    if(D)
        mpg_dist_destroy(D);
    mpg_engine_destroy(e);
    pthread_barrier_wait(&S->bar);
}

static int run_sph(const char *in_path, const char *expect_path, int64_t N, double box, int nt, int bh, int kernel)
{
    struct sph_in *in = (struct sph_in *)read_f64(in_path, 10 * (size_t)N);
    double *expect = read_f64(expect_path, 6 * (size_t)N);
    if(nt < 1 || nt > MAXT)
        return 1;
    const size_t cap = 512 * (size_t)N + (1 << 20);
    const size_t total = sizeof(struct shm) + MAXT * cap + 6 * N * sizeof(double);
    struct shm *S = mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if(S == MAP_FAILED) {
        perror("mmap");
        return 1;
    }
    memset(S, 0, sizeof(*S));
    S->cap = cap;
    S->N = N;
    pthread_barrierattr_t ba;
    pthread_barrierattr_init(&ba);
    pthread_barrierattr_setpshared(&ba, PTHREAD_PROCESS_SHARED);
    int bad = 0;
#ifdef MPG_TEST_MPI
    int me = 0, size = 1; /* (as run_ranks: this process is one rank, rank 0 collects) */
    MPI_Comm_rank(MPI_COMM_WORLD, &me);
    MPI_Comm_size(MPI_COMM_WORLD, &size);
    if(size != nt || 6 * N > 2000000000) {
        fprintf(stderr, "FAIL NTask %d but mpiexec started %d processes (or the result does not fit one MPI_Reduce)\n", nt, size);
        return 1;
    }
    pthread_barrier_init(&S->bar, &ba, 1u);
    sph_rank(S, me, nt, in, box, bh, kernel);
    MPI_Reduce(me == 0 ? MPI_IN_PLACE : (void *)shm_result(S), shm_result(S), (int)(6 * N), MPI_DOUBLE, MPI_SUM, 0, MPI_COMM_WORLD);
    if(me != 0)
        return 0;
#else
    pthread_barrier_init(&S->bar, &ba, (unsigned)nt);
    pid_t pids[MAXT];
    for(int r = 0; r < nt; r++) {
        pids[r] = fork();
        if(pids[r] == 0) {
            sph_rank(S, r, nt, in, box, bh, kernel);
            _exit(0);
        }
    }
    for(int r = 0; r < nt; r++) {
        int status = 0;
        waitpid(pids[r], &status, 0);
        if(!WIFEXITED(status) || WEXITSTATUS(status) != 0)
            bad = 1;
    }
#endif
    if(bad) {
        printf("FAIL a rank exited with an error\n");
        return 1;
    }
    /* gas: all six columns; black holes (BlackHoleOn): Hsml and Density; everything else untouched (zero outputs) */
    const double *R = shm_result(S);
    double eh = 0, ed = 0, ea = 0, ee = 0, na = 0, ne = 0, nd = 0;
    int64_t ngas = 0, nbh = 0, nloose = 0;
    for(int64_t i = 0; i < N; i++) {
        const int ty = (int)in[i].Type;
        const double *r = R + 6 * i, *x = expect + 6 * i;
        if(ty == 5)
            nbh++;
        if(ty == 0 || ty == 5) { /* (black holes are density targets whatever BlackHoleOn says: density_haswork) */
            const double dh = fabs(r[0] / x[0] - 1);
            if(dh > 1e-12) { /* a target within an ulp of the NumNgb window's edge may take one iteration more or fewer (tests/test_gpu_sph.py) */
                nloose++;
                if(dh > 2.0 / 33.)
                    eh = dh > eh ? dh : eh;
                continue;
            }
            ed = fmax(ed, fabs(r[1] - x[1]));
            nd = fmax(nd, fabs(x[1]));
        }
        if(ty == 0) {
            ngas++;
            for(int j = 0; j < 3; j++) {
                ea = fmax(ea, fabs(r[2 + j] - x[2 + j]));
                na = fmax(na, fabs(x[2 + j]));
            }
            ee = fmax(ee, fabs(r[5] - x[5]));
            ne = fmax(ne, fabs(x[5]));
        }
        else if(ty != 5 && (r[1] != 0 || r[2] != 0 || r[5] != 0))
            eh = 1;
    }
    printf("sph %d rank(s): gas %lld bh %lld loose-Hsml %lld  Hsml-out-of-bound %.2e  Density %.2e  HydroAccel %.2e  DtEntropy %.2e\n", nt,
           (long long)ngas, (long long)nbh, (long long)nloose, eh, ed / nd, ea / na, ee / (ne > 0 ? ne : 1));
    if(!(eh == 0 && nloose * 1000 <= ngas + nbh && ed <= 1e-9 * nd && ea <= 1e-9 * na && ee <= 1e-9 * (ne > 0 ? ne : 1))) {
        printf("FAIL\n");
        return 1;
    }
    printf("PASS sph %d\n", nt);
    return 0;
}

static int run_ranks(const double *table, const double *pos, const double *expect, int n, int nmesh, double box, int nt, int host)
{
    const int64_t N = (int64_t)n * n * n;
    if(nt < 1 || nt > MAXT || nmesh % nt) {
        fprintf(stderr, "FAIL NTask must be in 1..%d and divide Nmesh\n", MAXT);
        return 1;
    }
    const size_t cap = (size_t)nmesh * nmesh * (nmesh / 2 + 1) * 16 * 2 + 64 * N + (1 << 20); /* a transpose, the particle rows, slack */
    const size_t total = sizeof(struct shm) + MAXT * cap + 6 * N * sizeof(double);
    struct shm *S = mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if(S == MAP_FAILED) {
        perror("mmap");
        return 1;
    }
    memset(S, 0, sizeof(*S));
    S->cap = cap;
    S->N = N;
    pthread_barrierattr_t ba;
    pthread_barrierattr_init(&ba);
    pthread_barrierattr_setpshared(&ba, PTHREAD_PROCESS_SHARED);
    int bad = 0;
#ifdef MPG_TEST_MPI
    /* this process IS one rank; the segment is private to it (the barrier counts one), the results are summed onto rank 0 (every
     * particle is written by its owner only, the others hold zeros) */
    int me = 0, size = 1;
    MPI_Comm_rank(MPI_COMM_WORLD, &me);
    MPI_Comm_size(MPI_COMM_WORLD, &size);
    if(size != nt) {
        fprintf(stderr, "FAIL NTask %d but mpiexec started %d processes\n", nt, size);
        return 1;
    }
    pthread_barrier_init(&S->bar, &ba, 1u);
    rank_main(S, me, nt, table, pos, n, nmesh, box, host);
    MPI_Allreduce(MPI_IN_PLACE, &S->substep_bad, 1, MPI_INT, MPI_MAX, MPI_COMM_WORLD);
    if(6 * N > 2000000000) {
        fprintf(stderr, "FAIL result too large for one MPI_Reduce\n");
        return 1;
    }
    MPI_Reduce(me == 0 ? MPI_IN_PLACE : (void *)shm_result(S), shm_result(S), (int)(6 * N), MPI_DOUBLE, MPI_SUM, 0, MPI_COMM_WORLD);
    if(me != 0)
        return 0;
#else
    pthread_barrier_init(&S->bar, &ba, (unsigned)nt);
    pid_t pids[MAXT];
    for(int r = 0; r < nt; r++) { /* (fork before anything touches the HIP runtime) */
        pids[r] = fork();
        if(pids[r] == 0) {
            rank_main(S, r, nt, table, pos, n, nmesh, box, host);
            _exit(0);
        }
    }
    for(int r = 0; r < nt; r++) {
        int status = 0;
        waitpid(pids[r], &status, 0);
        if(!WIFEXITED(status) || WEXITSTATUS(status) != 0)
            bad = 1;
    }
#endif
    if(bad) {
        printf("FAIL a rank exited with an error\n");
        return 1;
    }
    if(S->substep_bad) {
        printf("FAIL the sub-step (mpg_dist_grav_short_tree_active) differs from the full walk\n");
        return 1;
    }
    const double *R = shm_result(S);
    double *gpm = malloc(3 * N * sizeof(double)), *acc = malloc(3 * N * sizeof(double));
    for(int64_t i = 0; i < N; i++)
        for(int j = 0; j < 3; j++) {
            gpm[3 * i + j] = R[6 * i + j];
            acc[3 * i + j] = R[6 * i + 3 + j];
        }
    const double e_pm = relerr(gpm, expect, 3 * N), e_tr = relerr(acc, expect + 3 * N, 3 * N);
#ifdef MPG_TEST_MPI
    printf("MPI processes, collectives by %s\n", getenv("MPG_TEST_COMM") && !strcmp(getenv("MPG_TEST_COMM"), "rccl") ? "RCCL (bootstrapped by shim/mpg_rccl_mpi.c)" : "shim/mpg_mpi_comm.c");
#endif
    printf("ranks %d (%s): N %lld  GravPM err %.3e  acceleration err %.3e\n", nt, host ? "host tables" : "device arrays", (long long)N, e_pm, e_tr);
    if(!(e_pm < 1e-10 && e_tr < 1e-10)) {
        printf("FAIL\n");
        return 1;
    }
    printf("PASS ranks %d\n", nt);
    return 0;
}

int main(int argc, char **argv)
{
    if(argc >= 9 && !strcmp(argv[1], "sph")) {
#ifdef MPG_TEST_MPI
        MPI_Init(&argc, &argv);
        const int rc = run_sph(argv[2], argv[3], atoll(argv[4]), atof(argv[5]), atoi(argv[6]), atoi(argv[7]), atoi(argv[8]));
        if(rc)
            MPI_Abort(MPI_COMM_WORLD, rc);
        MPI_Finalize();
        return rc;
#else
        return run_sph(argv[2], argv[3], atoll(argv[4]), atof(argv[5]), atoi(argv[6]), atoi(argv[7]), atoi(argv[8]));
#endif
    }
    if(argc < 8) {
        fprintf(stderr, "usage: %s single|ranks|ranks_host table.f64 pos.f64 expect.f64 n nmesh box [NTask]\n       %s sph in.f64 expect.f64 N box NTask BlackHoleOn kernel\n", argv[0], argv[0]);
        return 2;
    }
    const int n = atoi(argv[5]), nmesh = atoi(argv[6]);
    const double box = atof(argv[7]);
    const size_t N = (size_t)n * n * n;
    double *table = read_f64(argv[2], 512 * 5), *pos = read_f64(argv[3], 3 * N), *expect = read_f64(argv[4], 6 * N);
    if(!strcmp(argv[1], "single"))
        return run_single(table, pos, expect, n, nmesh, box);
    if(!strcmp(argv[1], "run"))
        return run_order(table, pos, expect, argv[8], n, nmesh, box);
#ifdef MPG_TEST_MPI
    MPI_Init(&argc, &argv);
    const int rc = run_ranks(table, pos, expect, n, nmesh, box, argc > 8 ? atoi(argv[8]) : 2, !strcmp(argv[1], "ranks_host"));
    if(rc)
        MPI_Abort(MPI_COMM_WORLD, rc);
    MPI_Finalize();
    return rc;
#else
    return run_ranks(table, pos, expect, n, nmesh, box, argc > 8 ? atoi(argv[8]) : 2, !strcmp(argv[1], "ranks_host"));
#endif
}
