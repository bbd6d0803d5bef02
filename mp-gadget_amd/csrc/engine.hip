// engine.hip -- C-ABI (include/mpgadget_hip.h) and host-side orchestration of the gfx950 TreePM engine.
//
// Host side mirrors the reference's call sequence: gravpm_force (gravpm.c:61-119), force_tree_full
// (forcetree.c:110-128), grav_short_tree (gravshort-tree.c:96-154) with fill / reduce / postprocess of
// gravshort.h:47-96.  Every entry point catches mpg::Error and returns non-zero (the reference has no
// return codes here; the in-tree shim maps failures to endrun()).  There is no CPU fallback anywhere:
// without a HIP device mpg_engine_create fails.
#include "engine_internal.h"
#include <atomic>
#include <chrono>
#include <cstdio>

static thread_local std::string g_err;
std::string &mpg_err_slot() { return g_err; }

extern "C" {

const char *mpg_last_error(void) { return g_err.c_str(); }
const char *mpg_version(void) { return "mpgadget_hip 0.1 (gfx950; TreePM gravity: PM + tree + short-range walk)"; }

int mpg_engine_create(mpg_engine **out, int device)
{
    API_BEGIN
    MPG_CHECK(out != nullptr, "mpg_engine_create: null output pointer");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if(e != hipSuccess || ndev <= 0)
        fail(__FILE__, __LINE__, "mpg_engine_create: no HIP device available (this engine has no CPU path)");
    MPG_CHECK(device >= 0 && device < ndev, "mpg_engine_create: bad device index");
    MPG_HIP(hipSetDevice(device));
    mpg_engine *eng = new mpg_engine();
    eng->device = device;
    MPG_HIP(hipStreamCreateWithFlags(&eng->stream, hipStreamNonBlocking));
    eng->own_stream = true;
    eng->counters.reserve(16);
    *out = eng;
    API_END
}

void mpg_engine_destroy(mpg_engine *eng)
{
    if(!eng)
        return;
    (void)hipSetDevice(eng->device);
    (void)hipStreamSynchronize(eng->stream);
    if(eng->w3.split_stream)
        (void)hipStreamSynchronize(eng->w3.split_stream);
    for(auto &v : {&eng->walk_events, &eng->free_events})
        for(auto &e : *v) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
    for(auto &v : {&eng->walk_mid, &eng->free_mid})
        for(auto &e : *v)
            if(e)
                (void)hipEventDestroy(e);
    eng->host_join();
    for(auto &e : eng->chunk_ev)
        if(e)
            (void)hipEventDestroy(e);
    for(auto &e : eng->gchunk_ev)
        if(e)
            (void)hipEventDestroy(e);
    if(eng->ev_pm_done)
        (void)hipEventDestroy(eng->ev_pm_done);
    if(eng->ev_acc_up)
        (void)hipEventDestroy(eng->ev_acc_up);
    for(auto &e : eng->slice_ev)
        if(e)
            (void)hipEventDestroy(e);
    if(eng->copy_stream) {
        (void)hipStreamSynchronize(eng->copy_stream);
        (void)hipStreamDestroy(eng->copy_stream);
    }
    eng->pm.destroy();
    if(eng->aux_stream) {
        (void)hipStreamSynchronize(eng->aux_stream);
        (void)hipStreamDestroy(eng->aux_stream);
        (void)hipEventDestroy(eng->ev_inputs);
        (void)hipEventDestroy(eng->ev_tree_done);
        if(eng->ev_pad_done)
            (void)hipEventDestroy(eng->ev_pad_done);
    }
    if(eng->own_stream && eng->stream)
        (void)hipStreamDestroy(eng->stream);
    delete eng;
}

int mpg_engine_set_stream(mpg_engine *eng, void *hip_stream)
{
    API_BEGIN
    MPG_CHECK(eng, "null engine");
    if(eng->own_stream && eng->stream) {
        MPG_HIP(hipStreamSynchronize(eng->stream));
        MPG_HIP(hipStreamDestroy(eng->stream));
    }
    if(hip_stream) {
        eng->stream = (hipStream_t)hip_stream;
        eng->own_stream = false;
    }
    else {
        MPG_HIP(hipStreamCreateWithFlags(&eng->stream, hipStreamNonBlocking));
        eng->own_stream = true;
    }
    API_END
}

void *mpg_engine_get_stream(mpg_engine *eng) { return eng ? (void *)eng->stream : nullptr; }

int mpg_engine_synchronize(mpg_engine *eng)
{
    API_BEGIN
    MPG_CHECK(eng, "null engine");
    MPG_HIP(hipSetDevice(eng->device));
    MPG_HIP(hipStreamSynchronize(eng->stream));
    MPG_CHECK(walk_coop_error(eng->w3, eng->stream) == 0, "short-range walk aborted by its loop guard (corrupt tree?)");
    API_END
}

int mpg_set_gravshort_treepar(mpg_engine *eng, const mpg_gravshort_tree_params *par)
{
    API_BEGIN
    MPG_CHECK(eng && par, "null argument");
    eng->treepar = *par;
    API_END
}

int mpg_get_gravshort_treepar(mpg_engine *eng, mpg_gravshort_tree_params *par)
{
    API_BEGIN
    MPG_CHECK(eng && par, "null argument");
    *par = eng->treepar;
    API_END
}

int mpg_gravshort_set_softenings(mpg_engine *eng, double MeanSeparation)
{
    API_BEGIN
    MPG_CHECK(eng, "null engine");
    eng->GravitySoftening = eng->treepar.FractionalGravitySoftening * MeanSeparation; // gravshort-tree.c:47
    API_END
}

double mpg_force_softening(mpg_engine *eng) { return eng ? 2.8 * eng->GravitySoftening : 0.0; } // gravshort-tree.c:37-41

int mpg_gravshort_fill_ntab(mpg_engine *eng, int window_type, double Asmth, const double *table, int nrows)
{
    API_BEGIN
    MPG_CHECK(eng && table, "null argument");
    MPG_CHECK(nrows == NTAB, "gravshort_fill_ntab: the short-range table must have 512 rows");
    MPG_CHECK(window_type == 0 || window_type == 1, "gravshort_fill_ntab: unknown ShortRangeForceWindowType");
    if(window_type == 0 && Asmth != 1.5) // gravity.c:25-29
        fail(__FILE__, __LINE__, "The short range force window is calibrated for Asmth = 1.5, but running with " + std::to_string(Asmth));
    std::vector<float> f(NTAB), p(NTAB);
    for(int i = 0; i < NTAB; i++) {
        const double u = table[5 * i + 0] * 0.5 / Asmth;
        if(window_type == 0) {
            f[i] = (float)table[5 * i + 2];
            p[i] = (float)table[5 * i + 1];
        }
        else {
            f[i] = (float)(erfc(u) + 2.0 * u / sqrt(M_PI) * exp(-u * u));
            p[i] = (float)erfc(u);
        }
    }
    eng->tab_dx = table[5 * 1 + 0];
    eng->tab_force.reserve(NTAB);
    eng->tab_pot.reserve(NTAB);
    MPG_HIP(hipSetDevice(eng->device));
    MPG_HIP(hipMemcpy(eng->tab_force.p, f.data(), NTAB * sizeof(float), hipMemcpyHostToDevice));
    MPG_HIP(hipMemcpy(eng->tab_pot.p, p.data(), NTAB * sizeof(float), hipMemcpyHostToDevice));
    eng->have_tab = true;
    API_END
}

int mpg_gravpm_init_periodic(mpg_engine *eng, double BoxSize, double Asmth, int Nmesh, double G)
{
    API_BEGIN
    if(eng)
        eng->pm.kspace_force = getenv("MPG_PM_KSPACE_FORCE") != nullptr;
    MPG_CHECK(eng, "null engine");
    MPG_HIP(hipSetDevice(eng->device));
    eng->pm.init(BoxSize, Asmth, Nmesh, G, eng->stream);
    API_END
}

int mpg_petapm_destroy(mpg_engine *eng)
{
    API_BEGIN
    MPG_CHECK(eng, "null engine");
    MPG_HIP(hipSetDevice(eng->device));
    eng->pm.destroy();
    API_END
}

int mpg_init_forcetree_params(mpg_engine *eng, double TreeAllocFactor)
{
    API_BEGIN
    MPG_CHECK(eng, "null engine");
    eng->TreeAllocFactor = TreeAllocFactor;
    API_END
}

void mpg_particle_view_reference_layout(mpg_particle_view *v, void *P, int64_t NumPart)
{
    // struct particle_data, partmanager.h:9-71 (160 bytes; offsets as measured in SURVEY 8(a))
    v->base = P;
    v->n = NumPart;
    v->stride = 160;
    v->off_pos = 0;
    v->off_mass = 28;
    v->off_pi = 32;
    v->off_flags = 36;
    v->off_type = 39;
    v->off_vel = 40;
    v->off_accel = 64;
    v->off_gravpm = 88;
    v->off_hsml = 120;
    v->off_potential = 152;
}

/* ------------------------------ device-resident path ------------------------------ */

int mpg_dev_bind_particles(mpg_engine *eng, int64_t n, const double *d_pos, const float *d_mass, const uint8_t *d_type, double BoxSize)
{
    API_BEGIN
    MPG_CHECK(eng, "null engine");
    MPG_CHECK(n >= 0 && (n == 0 || (d_pos && d_mass)), "mpg_dev_bind_particles: null particle arrays");
    MPG_CHECK(BoxSize > 0, "mpg_dev_bind_particles: BoxSize must be positive");
    eng->host_join(); // (a prefetch or a write-back of the host-pointer calls still uses the binding and the stream)
    eng->n = n;
    eng->d_pos = d_pos;
    eng->d_mass = d_mass;
    eng->d_type = d_type;
    eng->box = BoxSize;
    eng->pm_queued = false; // a tree build that follows sees new arrays: it stays behind everything on the main stream
    API_END
}

int mpg_dev_gravpm_force(mpg_engine *eng, double *d_gravpm, double *d_potential)
{
    API_BEGIN
    MPG_CHECK(eng && d_gravpm, "null argument");
    MPG_HIP(hipSetDevice(eng->device));
    MPG_CHECK(eng->pm.nmesh > 0, "gravpm_force called before gravpm_init_periodic");
    MPG_CHECK(eng->pm.box == eng->box, "gravpm_force: BoxSize of the mesh differs from the bound particles");
    static const bool no_overlap = getenv("MPG_NO_TREE_OVERLAP") != nullptr;
    if(!no_overlap && !eng->timer.enabled) {
        if(!eng->aux_stream) {
            MPG_HIP(hipStreamCreateWithFlags(&eng->aux_stream, hipStreamNonBlocking));
            MPG_HIP(hipEventCreateWithFlags(&eng->ev_inputs, hipEventDisableTiming));
            MPG_HIP(hipEventCreateWithFlags(&eng->ev_tree_done, hipEventDisableTiming));
            MPG_HIP(hipEventCreateWithFlags(&eng->ev_pad_done, hipEventDisableTiming));
        }
        MPG_HIP(hipEventRecord(eng->ev_inputs, eng->stream)); // everything queued so far: the particle arrays are final, the last walk is done
        eng->pm_queued = true;
    }
    eng->pm.force(eng->n, eng->d_pos, eng->d_mass, nullptr, d_gravpm, d_potential, eng->stream, &eng->timer);
    API_END
}

int mpg_dev_tree_top_partial(mpg_engine *eng, int La, int64_t n_own, double *d_out)
{
    API_BEGIN
    MPG_CHECK(eng && d_out, "null argument");
    MPG_CHECK(eng->tree_allocated, "no tree");
    MPG_HIP(hipSetDevice(eng->device));
    eng->tree.top_partial(La, n_own, d_out, eng->stream);
    API_END
}

int mpg_dev_tree_top_set(mpg_engine *eng, int La, const double *d_sums)
{
    API_BEGIN
    MPG_CHECK(eng && d_sums, "null argument");
    MPG_CHECK(eng->tree_allocated, "no tree");
    MPG_HIP(hipSetDevice(eng->device));
    eng->tree.top_set(La, d_sums, eng->stream);
    API_END
}

int mpg_dev_pm_slab_init(mpg_engine *eng, int rank, int world, int64_t *cplx_per_peer, int64_t *plane_doubles)
{
    API_BEGIN
    MPG_CHECK(eng, "null engine");
    MPG_HIP(hipSetDevice(eng->device));
    eng->pm.slab_init(rank, world);
    if(cplx_per_peer)
        *cplx_per_peer = (int64_t)eng->pm.slab_cplx_per_peer();
    if(plane_doubles)
        *plane_doubles = (int64_t)eng->pm.nmesh * eng->pm.nmesh;
    API_END
}

int mpg_dev_pm_slab_forward_a(mpg_engine *eng, double *sendA)
{
    API_BEGIN
    MPG_CHECK(eng && sendA, "null argument");
    MPG_HIP(hipSetDevice(eng->device));
    MPG_CHECK(eng->pm.box == eng->box, "gravpm_force: BoxSize of the mesh differs from the bound particles");
    eng->pm.slab_forward_a(eng->n, eng->d_pos, eng->d_mass, sendA, eng->stream);
    API_END
}

int mpg_dev_pm_slab_forward_b(mpg_engine *eng, double *recvA, double *sendB)
{
    API_BEGIN
    MPG_CHECK(eng && recvA && sendB, "null argument");
    MPG_HIP(hipSetDevice(eng->device));
    eng->pm.slab_forward_b(recvA, sendB, eng->stream);
    API_END
}

int mpg_dev_pm_slab_inverse_c(mpg_engine *eng, const double *recvB, double *ghost_send)
{
    API_BEGIN
    MPG_CHECK(eng && recvB && ghost_send, "null argument");
    MPG_HIP(hipSetDevice(eng->device));
    eng->pm.slab_inverse_c(recvB, ghost_send, eng->stream);
    API_END
}

int mpg_dev_pm_slab_readout(mpg_engine *eng, const double *ghost_recv, const int *d_targets, int64_t ntargets, double *d_gravpm,
                            double *d_potential)
{
    API_BEGIN
    MPG_CHECK(eng && ghost_recv && d_gravpm && (d_targets || ntargets == 0), "null argument");
    MPG_HIP(hipSetDevice(eng->device));
    eng->pm.slab_readout(ghost_recv, d_targets, ntargets, eng->d_pos, d_gravpm, d_potential, eng->stream);
    API_END
}

// tree of the bound particles + moments on the given stream (csrc/dist.hip: beside the PM step, from a second host thread)
// What the default walk kernels read beside the depth-first tree - the level-ordered copy and the leaves' particles in blocks of 8 - made
// with the tree, on the tree's stream (beside the PM force when one is queued), not at the start of the walk (rounds 1-5: 0.5 ms of every
// walk at 256^3).  Not when the phases are being timed (the tree's phases are its own) or another kernel was selected.
static bool tree_walk_copies(mpg_engine *eng, hipStream_t st, bool leaf_blocks = true)
{
    if(eng->timer.enabled || !(eng->walk_variant == 0 || eng->walk_variant == 6) || eng->tree.npart < 4096)
        return false;
    eng->tree.ensure_level_order(st);
    if(leaf_blocks)
        eng->tree.ensure_leaf_pad(st);
    return true;
}

// a tree build on `st` overwrites what a leaf-block kernel still running on the tree stream reads
static void wait_for_leaf_blocks(mpg_engine *eng, hipStream_t st)
{
    if(eng->pad_pending && st != eng->aux_stream)
        MPG_HIP(hipStreamWaitEvent(st, eng->ev_pad_done, 0));
    eng->pad_pending = false;
}

void engine_tree_build_on(mpg_engine *eng, int mask, hipStream_t st)
{
    eng->pm_queued = false;
    wait_for_leaf_blocks(eng, st);
    eng->tree.build(eng->n, eng->d_pos, eng->d_mass, eng->d_type, mask, eng->box, st, nullptr);
    eng->tree.calc_moments(nullptr, st, nullptr);
    tree_walk_copies(eng, st);
    eng->tree_allocated = true;
    eng->tree_mask = mask;
    eng->full_particle_tree = (eng->tree.npart == eng->n);
}

int mpg_dev_force_tree_build(mpg_engine *eng, int mask)
{
    API_BEGIN
    MPG_CHECK(eng, "null engine");
    MPG_HIP(hipSetDevice(eng->device));
    if(eng->pm_queued && !eng->timer.enabled) {
        // next to the PM force queued on the main stream; whatever is queued on the main stream after this call waits for the tree
        eng->pm_queued = false;
        eng->pad_pending = false; // (a leaf-block kernel of the last tree runs on this same stream: ordered)
        MPG_HIP(hipStreamWaitEvent(eng->aux_stream, eng->ev_inputs, 0));
        eng->tree.build(eng->n, eng->d_pos, eng->d_mass, eng->d_type, mask, eng->box, eng->aux_stream, nullptr);
        eng->tree.calc_moments(nullptr, eng->aux_stream, nullptr);
        const bool copies = tree_walk_copies(eng, eng->aux_stream, false);
        MPG_HIP(hipEventRecord(eng->ev_tree_done, eng->aux_stream));
        MPG_HIP(hipStreamWaitEvent(eng->stream, eng->ev_tree_done, 0));
        if(copies) { // the leaf blocks are read by the walk's SECOND kernel only: made behind the event, beside the list kernel
            eng->tree.ensure_leaf_pad(eng->aux_stream);
            MPG_HIP(hipEventRecord(eng->ev_pad_done, eng->aux_stream));
            eng->pad_pending = true;
        }
    }
    else {
        eng->pm_queued = false;
        wait_for_leaf_blocks(eng, eng->stream);
        eng->tree.build(eng->n, eng->d_pos, eng->d_mass, eng->d_type, mask, eng->box, eng->stream, &eng->timer);
        eng->tree.calc_moments(nullptr, eng->stream, &eng->timer);
        tree_walk_copies(eng, eng->stream);
    }
    if(eng->timer.enabled)
        eng->timer.t.tree_total = eng->timer.t.tree_keys + eng->timer.t.tree_sort + eng->timer.t.tree_nodes + eng->timer.t.tree_moments;
    eng->tree_allocated = true;
    eng->tree_mask = mask;
    // full_particle_tree_flag (forcetree.c:127,164-165): all particle types are in the tree
    eng->full_particle_tree = (eng->tree.npart == eng->n);
    API_END
}

static GravParams make_gp(mpg_engine *eng, double rho0)
{
    GravParams gp{};
    MPG_CHECK(eng->pm.nmesh > 0, "grav_short_tree needs gravpm_init_periodic first (cell size, Asmth, G)");
    MPG_CHECK(eng->have_tab, "grav_short_tree called before gravshort_fill_ntab");
    const double cellsize = eng->tree.box / eng->pm.nmesh;   // gravshort-tree.c:101
    gp.box = eng->tree.box;
    gp.invbox = 1.0 / gp.box;
    gp.rcut = eng->treepar.Rcut * eng->pm.Asmth * cellsize;  // :102
    gp.rcut2 = gp.rcut * gp.rcut;
    gp.h = 2.8 * eng->GravitySoftening;
    MPG_CHECK(gp.h > 0, "grav_short_tree called before gravshort_set_softenings");
    gp.hinv = 1.0 / gp.h;
    gp.h3inv = 1.0 / gp.h / gp.h / gp.h;
    gp.h2 = gp.h * gp.h;
    gp.inv_cell_dx = 1.0 / (cellsize * eng->tab_dx);
    gp.errtol = eng->treepar.ErrTolForceAcc;
    gp.use_bh = eng->treepar.TreeUseBH != 0;
    // gravshort-tree.c:266-270
    gp.bhangle2 = eng->treepar.BHOpeningAngle * eng->treepar.BHOpeningAngle;
    if(eng->treepar.TreeUseBH == 0)
        gp.bhangle2 = eng->treepar.MaxBHOpeningAngle * eng->treepar.MaxBHOpeningAngle;
    gp.G = eng->pm.G;
    gp.cbrtrho0 = pow(rho0, 1.0 / 3);
    gp.full_tree = eng->full_particle_tree ? 1 : 0;
    return gp;
}

int mpg_dev_set_walk_cost(mpg_engine *eng, float *d_cost)
{
    API_BEGIN
    MPG_CHECK(eng, "null engine");
    eng->d_walk_cost = d_cost;
    API_END
}

// OldAcc = |FullTreeGravAccel + GravPM| / G of every particle (grav_get_abs_accel, gravshort.h:70-80), the arithmetic of the walk kernels
// (rows: the targets of the walk, or all n when the list is NULL - only the targets' rows of prev / gravpm are read: under mpg_dist the
// engine is bound to own + ghost rows while the caller's arrays hold the own rows only)
__global__ void __launch_bounds__(256) k_oldacc(int64_t n, const int *__restrict__ list, const double *__restrict__ prev,
                                                const double *__restrict__ gravpm, double G, double *__restrict__ old)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if(k >= n)
        return;
    const int64_t i = list ? (int64_t)list[k] : k;
    double s2 = 0;
    for(int j = 0; j < 3; j++) {
        const double a = prev[3 * i + j] + (gravpm ? gravpm[3 * i + j] : 0.0);
        s2 += a * a;
    }
    old[i] = sqrt(s2) / G;
}

int mpg_dev_grav_short_tree(mpg_engine *eng, const double *d_oldacc, const double *d_prev_accel, const double *d_gravpm,
                            const int *d_active, int64_t nactive, double *d_accel, double *d_potential, double rho0)
{
    API_BEGIN
    MPG_CHECK(eng && d_accel, "null argument");
    MPG_HIP(hipSetDevice(eng->device));
    MPG_CHECK(eng->tree_allocated && eng->tree.has_moments, "Gravtree called before tree moments computed!"); // gravshort-tree.c:113-114
    const GravParams gp = make_gp(eng, rho0);
    if(!d_oldacc && d_prev_accel && d_prev_accel == d_accel) {
        // Results written in place over the previous acceleration (resident mode, mpg_dist callers): the two-kernel walk may run its
        // list pass a second time with longer lists after the evaluation has stored new values for the targets that fitted, and the
        // fallback kernels run after it, so the opening input is taken once, before any kernel of this walk writes.
        // Taken for the walk's TARGETS only: with an active list the caller's arrays need not hold eng->n rows (mpg_dist binds the engine
        // to own + ghost particles and passes arrays of the own rows; its targets are own rows).
        const int64_t nrows = d_active ? nactive : eng->n;
        eng->w_old.reserve((size_t)eng->n + 1);
        if(nrows > 0)
            hipLaunchKernelGGL(k_oldacc, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, eng->stream, nrows, d_active, d_prev_accel, d_gravpm,
                               gp.G, eng->w_old.p);
        d_oldacc = eng->w_old.p;
        d_prev_accel = nullptr;
    }
    WalkIO io;
    io.targets = d_active;
    if(d_active)
        io.ntargets = nactive;
    else {
        // ActiveParticle == NULL: all particles (timestep.c:77-84).  When the tree holds them all, walk in tree order.
        // (a tree of all types holds every LIVE particle: the garbage / swallowed records it leaves out are skipped by the reference's
        // queue as well, treewalk.c:92-96,234)
        MPG_CHECK(eng->tree.npart == eng->n || eng->full_particle_tree, "grav_short_tree with ActiveParticle == NULL needs a tree of all particles "
                                                                         "(pass the active list for masked trees)");
        io.ntargets = eng->tree.npart;
    }
    io.pos = eng->d_pos;
    io.mass = eng->d_mass;
    io.oldacc = d_oldacc;
    io.prev_accel = d_prev_accel;
    io.gravpm = d_gravpm;
    io.accel = d_accel;
    io.potential = eng->full_particle_tree ? d_potential : nullptr;
    io.tab_force = eng->tab_force.p;
    io.tab_pot = eng->tab_pot.p;
    io.counters = eng->counters.p;
    io.cost = eng->d_walk_cost;
    {
        static const int lp = getenv("MPG_LIST_PRIO") ? atoi(getenv("MPG_LIST_PRIO")) : 0, ep = getenv("MPG_EVAL_PRIO") ? atoi(getenv("MPG_EVAL_PRIO")) : 0;
        io.list_prio = lp;
        io.eval_prio = ep;
    }
    if(eng->count)
        MPG_HIP(hipMemsetAsync(eng->counters.p, 0, 16 * sizeof(unsigned long long), eng->stream));
    eng->timer.start(eng->stream);
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    if(!eng->free_events.empty()) {
        ev = eng->free_events.back();
        eng->free_events.pop_back();
    }
    else {
        MPG_HIP(hipEventCreate(&ev.first));
        MPG_HIP(hipEventCreate(&ev.second));
    }
    MPG_HIP(hipEventRecord(ev.first, eng->stream));
    hipEvent_t mid = nullptr;
    if(!eng->free_mid.empty()) {
        mid = eng->free_mid.back();
        eng->free_mid.pop_back();
    }
    else
        MPG_HIP(hipEventCreate(&mid));
    eng->w3.ev_mid = mid;
    eng->w3.mid_recorded = false;
    //  a walk that throws - a loop guard, an overflow - must not leave w3.ev_mid pointing at an event nobody owns, nor
    // lose the three events: the guard hands them back unless the normal path below has filed them
    struct EventGuard {
        mpg_engine *e;
        std::pair<hipEvent_t, hipEvent_t> *ev;
        hipEvent_t *mid;
        bool filed = false;
        ~EventGuard()
        {
            if(filed)
                return;
            e->w3.ev_mid = nullptr;
            e->w3.mid_recorded = false;
            e->free_events.push_back(*ev);
            e->free_mid.push_back(*mid);
        }
    } ev_guard{eng, &ev, &mid};
    // hoisting the minimum-image wrap out of the pair loop is valid when every source range shares the image of its
    // node: Rcut + 1.5 * (largest leaf side) < Box/2, and Rcut well below Box/4 (see grav_walk_coop.hip)
    const double maxleaf = 1.001 * gp.box / (double)(1 << eng->tree.minleaflevel);
    const bool fastwrap = !getenv("MPG_NO_FASTWRAP") && (gp.rcut + 1.5 * maxleaf < 0.49 * gp.box) && (gp.rcut < 0.2 * gp.box);
    auto run_variant_io = [&](int v, const WalkIO &w) {
        if(v != 1)
            eng->tree.ensure_level_order(eng->stream); // variants 4 and 6 walk the level-ordered copy of the tree
        if(v == 6) {
            eng->tree.ensure_leaf_pad(eng->stream);    // ... and 6 the leaves' particles in blocks of 8 (both made with the tree as a rule)
            eng->w3.ev_before_eval = eng->pad_pending ? eng->ev_pad_done : nullptr;
        }
        if(v == 1)
            launch_grav_walk(eng->tree.view(), gp, w, w.potential != nullptr, eng->count, fastwrap, eng->walk_thresh, eng->stream);
        else if(v == 6)
            launch_grav_walk_split(eng->tree.view(), gp, w, w.potential != nullptr, eng->count, fastwrap, eng->walk_thresh, eng->w3, eng->stream);
        else
            launch_grav_walk_coop(eng->tree.view(), gp, w, w.potential != nullptr, eng->count, fastwrap, eng->w3, eng->stream);
    };
    auto run_variant = [&](int v) { run_variant_io(v, io); };
    int variant = eng->walk_variant;
    if(variant == 0) {
        // Default: the two-kernel walk (6) with the cooperative kernel (4) for the targets whose lists overflow; the lane-per-target
        // kernel (1) for small target sets (fewer than 4096), where launch count matters more than lane use.  (This used to be decided by timing
        // kernels 1, 4 and 6 on the first walk and every 64th.  On the measured sets - grid, Zel'dovich, clustered, 96^3 .. 256^3 -
        // kernel 6 now always wins, and the trial itself is unaffordable on clustered sets: 6.7 s for kernel 1 and 2 s for kernel 4
        // against 0.18 s for kernel 6 at 256^3; timing a sample of the targets instead misjudges kernel 6, whose fixed costs -
        // slices, two streams, the control-word read-back - weigh on a small sample.)
        // (round 3, with k_walk_lists8: 32 768 active targets of a 256^3 tree take 1.09 ms with kernel 6 against 4.6 ms with kernel 1;
        // the threshold was 65 536 when the list kernels needed that many targets to fill the chip)
        static const int64_t split_min = getenv("MPG_SPLIT_MIN_TARGETS") ? atoll(getenv("MPG_SPLIT_MIN_TARGETS")) : 4096;
        variant = io.ntargets >= split_min ? 6 : 1;
        eng->walk_choice = variant;
        eng->walks_since_tune++;
    }
    if(variant > 0)
        run_variant(variant);
    MPG_HIP(hipEventRecord(ev.second, eng->stream));
    ev_guard.filed = true;
    eng->walk_events.push_back(ev);
    eng->w3.ev_mid = nullptr;
    if(eng->w3.mid_recorded)
        eng->walk_mid.push_back(mid);
    else {
        eng->walk_mid.push_back(nullptr);
        eng->free_mid.push_back(mid);
    }
    if(eng->walk_events.size() > 4096) { // nobody is collecting: recycle the oldest
        eng->free_events.push_back(eng->walk_events.front());
        eng->walk_events.erase(eng->walk_events.begin());
        if(eng->walk_mid.front())
            eng->free_mid.push_back(eng->walk_mid.front());
        eng->walk_mid.erase(eng->walk_mid.begin());
    }
    eng->timer.lap(eng->stream, &eng->timer.t.walk);
    eng->timer.t.walk_launches = 1;
    eng->last_targets = io.ntargets;
    // TreeUseBH > 1: Barnes-Hut on the first walk only (gravshort-tree.c:148-151)
    if(eng->treepar.TreeUseBH > 1)
        eng->treepar.TreeUseBH = 0;
    API_END
}

static unsigned read_flag(mpg_engine *eng, unsigned *d);

int mpg_dev_grav_short_pair(mpg_engine *eng, const int *d_active, int64_t nactive, double Rcut, double *d_accel, double *d_potential, double rho0)
{
    API_BEGIN
    MPG_CHECK(eng && d_accel, "null argument");
    MPG_HIP(hipSetDevice(eng->device));
    MPG_CHECK(eng->tree_allocated, "grav_short_pair: no tree");
    const GravParams gp = make_gp(eng, rho0);
    WalkIO io;
    io.targets = d_active;
    if(d_active)
        io.ntargets = nactive;
    else {
        MPG_CHECK(eng->tree.npart == eng->n, "grav_short_pair with ActiveParticle == NULL needs a tree of all particles");
        io.ntargets = eng->tree.npart;
    }
    io.pos = eng->d_pos;
    io.mass = eng->d_mass;
    io.accel = d_accel;
    io.potential = eng->full_particle_tree ? d_potential : nullptr;
    io.tab_force = eng->tab_force.p;
    io.tab_pot = eng->tab_pot.p;
    eng->tree.ensure_level_order(eng->stream);
    eng->ts_flag.reserve(4);
    MPG_HIP(hipMemsetAsync(eng->ts_flag.p, 0, sizeof(unsigned), eng->stream));
    const double rcut_abs = Rcut * eng->pm.Asmth * (eng->tree.box / eng->pm.nmesh); // gravshort-pair.c:27-28
    launch_grav_short_pair(eng->tree.view(), gp, io, rcut_abs, io.potential != nullptr, eng->ts_flag.p, eng->stream);
    MPG_CHECK(read_flag(eng, eng->ts_flag.p) == 0, "grav_short_pair: neighbour-search stack overflow");
    API_END
}

/* ------------------------------ time integration (device-resident arrays) ------------------------------ */
static unsigned read_flag(mpg_engine *eng, unsigned *d)
{
    unsigned e = 0;
    MPG_HIP(hipMemcpyAsync(&e, d, sizeof(e), hipMemcpyDeviceToHost, eng->stream));
    MPG_HIP(hipStreamSynchronize(eng->stream));
    return e;
}

int mpg_dev_drift_all_particles(mpg_engine *eng, int64_t n, double *d_pos, const double *d_vel, const unsigned char *d_type,
                                const unsigned char *d_flags, double *d_hsml, const double *d_dthsml, double ddrift, double BoxSize,
                                const double random_shift[3])
{
    API_BEGIN
    MPG_CHECK(eng && d_pos && d_vel && random_shift && n >= 0, "null argument");
    MPG_CHECK(!d_hsml || (d_dthsml && d_type), "drift: Hsml needs DtHsml and Type");
    MPG_HIP(hipSetDevice(eng->device));
    eng->pm_queued = false; // positions move: a tree build that follows must stay behind this on the main stream
    eng->ts_flag.reserve(4);
    MPG_HIP(hipMemsetAsync(eng->ts_flag.p, 0, sizeof(unsigned), eng->stream));
    launch_drift(n, d_pos, d_vel, d_type, d_flags, d_hsml, d_dthsml, ddrift, BoxSize, random_shift, eng->ts_flag.p, eng->stream);
    MPG_CHECK(read_flag(eng, eng->ts_flag.p) == 0, "drift: a particle has Hsml <= 0 or a non-finite position (drift.c:62-75)");
    API_END
}

int mpg_dev_apply_pm_half_kick(mpg_engine *eng, int64_t n, double *d_vel, const double *d_gravpm, const unsigned char *d_flags, double Fgravkick)
{
    API_BEGIN
    MPG_CHECK(eng && d_vel && d_gravpm && n >= 0, "null argument");
    MPG_HIP(hipSetDevice(eng->device));
    launch_pm_half_kick(n, d_vel, d_gravpm, d_flags, Fgravkick, eng->stream);
    API_END
}

int mpg_dev_apply_half_kick(mpg_engine *eng, int64_t n, const int *d_active, int64_t nactive, double *d_vel, const double *d_gravaccel,
                            const unsigned char *d_type, const unsigned char *d_flags, const unsigned char *d_tb_grav,
                            const unsigned char *d_tb_hydro, const double *d_hydroaccel, double *d_entropy, const double *d_dtentropy,
                            const mpg_kick_factors *K)
{
    API_BEGIN
    MPG_CHECK(eng && d_vel && d_gravaccel && K && n >= 0, "null argument");
    MPG_CHECK(!d_type || (d_hydroaccel && d_entropy && d_dtentropy), "half kick: gas needs HydroAccel, Entropy and DtEntropy");
    MPG_HIP(hipSetDevice(eng->device));
    eng->ts_flag.reserve(4);
    MPG_HIP(hipMemsetAsync(eng->ts_flag.p, 0, sizeof(unsigned), eng->stream));
    launch_half_kick(n, d_active, nactive, d_vel, d_gravaccel, d_type, d_flags, d_tb_grav, d_tb_hydro, d_hydroaccel, d_entropy, d_dtentropy, *K,
                     eng->ts_flag.p, eng->stream);
    MPG_CHECK(read_flag(eng, eng->ts_flag.p) == 0, "half kick: a particle has an unexpected time bin (timestep.c:900-901)");
    API_END
}

int mpg_dev_timestep_gravity_dloga(mpg_engine *eng, int64_t n, const double *d_gravaccel, const double *d_gravpm, double atime, double hubble,
                                   double ErrTolIntAccuracy, double *d_dloga)
{
    API_BEGIN
    MPG_CHECK(eng && d_gravaccel && d_gravpm && d_dloga && n >= 0, "null argument");
    MPG_CHECK(eng->GravitySoftening > 0, "timestep: gravshort_set_softenings has not been called");
    MPG_HIP(hipSetDevice(eng->device));
    launch_timestep_gravity(n, d_gravaccel, d_gravpm, atime, hubble, ErrTolIntAccuracy, 2.8 * eng->GravitySoftening, d_dloga, eng->stream);
    API_END
}

int mpg_dev_timestep_hydro_dloga(mpg_engine *eng, int64_t n, const unsigned char *d_type, const double *d_hsml, const double *d_dthsml,
                                 const double *d_maxsignalvel, const unsigned char *d_bh_mintimebin, const double *dloga_for_bin, double atime,
                                 double hubble, double CourantFac, double *d_dloga, unsigned char *d_titype)
{
    API_BEGIN
    MPG_CHECK(eng && d_dloga && n >= 0, "null argument");
    MPG_CHECK(!d_type || (d_hsml && d_maxsignalvel), "timestep_hydro_dloga: gas needs Hsml and MaxSignalVel");
    MPG_HIP(hipSetDevice(eng->device));
    const double *d_bins = nullptr;
    if(d_bh_mintimebin && dloga_for_bin) {
        eng->hier_sp.reserve((size_t)MPG_TIMEBINS + 2);
        MPG_HIP(hipMemcpyAsync(eng->hier_sp.p, dloga_for_bin, (MPG_TIMEBINS + 1) * sizeof(double), hipMemcpyHostToDevice, eng->stream));
        d_bins = eng->hier_sp.p;
    }
    const double fac3 = pow(atime, 3 * (1 - 5.0 / 3.0) / 2.0); // timestep.c:1083, GAMMA = 5/3 (physconst.h:35)
    launch_timestep_hydro(n, d_type, d_hsml, d_dthsml, d_maxsignalvel, d_bins ? d_bh_mintimebin : nullptr, d_bins, atime, hubble, CourantFac, fac3,
                          d_dloga, d_titype, eng->stream);
    if(d_bins)
        MPG_HIP(hipStreamSynchronize(eng->stream)); // (the table sits in a scratch buffer the hierarchical loop shares)
    API_END
}

int mpg_dev_peano_keys(mpg_engine *eng, int64_t n, const double *d_pos, double BoxSize, uint64_t *d_keys)
{
    API_BEGIN
    MPG_CHECK(eng && n >= 0 && (n == 0 || (d_pos && d_keys)) && BoxSize > 0, "peano_keys: bad argument");
    MPG_HIP(hipSetDevice(eng->device));
    launch_peano_keys(n, d_pos, BoxSize, d_keys, eng->stream);
    API_END
}

int mpg_dev_order_by_type_and_key(mpg_engine *eng, int64_t n, const unsigned char *d_type, const unsigned char *d_flags,
                                  const uint64_t *d_keys, int *d_perm, int64_t *n_live)
{
    API_BEGIN
    MPG_CHECK(eng && n >= 0 && n < ((int64_t)1 << 31) && (n == 0 || (d_keys && d_perm)) && n_live, "order_by_type_and_key: bad argument");
    MPG_HIP(hipSetDevice(eng->device));
    *n_live = order_by_type_and_key(n, d_type, d_flags, d_keys, d_perm, eng->peano, eng->stream);
    API_END
}

// ---- Peano-Hilbert domain decomposition (domain.hip) ------------------------------------------------------------------------
static_assert(sizeof(mpg_topnode) == sizeof(TopNode) && offsetof(mpg_topnode, Count) == offsetof(TopNode, Count), "mpg_topnode layout");

int mpg_dev_domain_sample(mpg_engine *eng, int64_t n, const double *d_pos, const unsigned char *d_garbage, double BoxSize, int PreSort,
                          int SubSampleDistance, uint64_t *keys_out, int64_t cap, int64_t *nsample)
{
    API_BEGIN
    MPG_CHECK(eng && n >= 0 && (n == 0 || d_pos) && BoxSize > 0 && keys_out && nsample, "domain_sample: bad argument");
    MPG_HIP(hipSetDevice(eng->device));
    *nsample = domain_sample(n, d_pos, d_garbage, BoxSize, PreSort, SubSampleDistance, keys_out, cap, eng->domain, eng->stream);
    API_END
}

int mpg_domain_local_refine(const uint64_t *keys, const int64_t *costs, int64_t nsample, mpg_topnode *tree, int *size, int MaxTopNodes, int *failed)
{
    API_BEGIN
    MPG_CHECK((keys || nsample == 0) && nsample >= 0 && tree && size && failed, "domain_local_refine: bad argument");
    *failed = toptree_local_refine(keys, costs, nsample, (TopNode *)tree, size, MaxTopNodes) ? 0 : 1;
    API_END
}

int mpg_domain_toptree_truncate(mpg_topnode *tree, int *size, int64_t countlimit, int64_t costlimit)
{
    API_BEGIN
    MPG_CHECK(tree && size && *size >= 1, "domain_toptree_truncate: bad argument");
    toptree_truncate((TopNode *)tree, size, countlimit, costlimit);
    API_END
}

int mpg_domain_toptree_merge(mpg_topnode *A, int *sizeA, const mpg_topnode *B, int sizeB, int MaxTopNodes, int *failed)
{
    API_BEGIN
    MPG_CHECK(A && sizeA && *sizeA >= 1 && (B || sizeB == 0) && sizeB >= 0 && failed, "domain_toptree_merge: bad argument");
    *failed = toptree_merge((TopNode *)A, sizeA, (const TopNode *)B, sizeB, MaxTopNodes) ? 0 : 1;
    API_END
}

int mpg_domain_global_refine(mpg_topnode *tree, int *size, int MaxTopNodes, int64_t countlimit, int64_t costlimit, int *failed)
{
    API_BEGIN
    MPG_CHECK(tree && size && *size >= 1 && failed, "domain_global_refine: bad argument");
    *failed = toptree_global_refine((TopNode *)tree, size, MaxTopNodes, countlimit, costlimit) ? 0 : 1;
    API_END
}

int mpg_domain_create_topleaves(mpg_topnode *tree, int size, int *leaf_topnode, int *nleaves)
{
    API_BEGIN
    MPG_CHECK(tree && size >= 1 && leaf_topnode && nleaves, "domain_create_topleaves: bad argument");
    *nleaves = toptree_create_leaves((TopNode *)tree, size, leaf_topnode);
    API_END
}

int mpg_domain_assign_topleaves_balanced(mpg_topnode *tree, int size, int *leaf_topnode, int nleaves, const int64_t *cost, int NTask,
                                         int NsegmentPerTask, int *leaf_task, int *StartLeaf, int *EndLeaf)
{
    API_BEGIN
    MPG_CHECK(tree && size >= 1 && leaf_topnode && nleaves >= 1 && cost && NTask >= 1 && NsegmentPerTask >= 1 && leaf_task && StartLeaf && EndLeaf,
              "domain_assign_topleaves_balanced: bad argument");
    toptree_assign_balanced((TopNode *)tree, size, leaf_topnode, nleaves, cost, NTask, NsegmentPerTask, leaf_task, StartLeaf, EndLeaf);
    API_END
}

int mpg_dev_domain_topleaves(mpg_engine *eng, int64_t n, const double *d_pos, const unsigned char *d_garbage, double BoxSize,
                             const mpg_topnode *tree, int size, int nleaves, const int *leaf_task, int NTask, int32_t *d_topleaf,
                             int32_t *d_task, int64_t *leaf_counts, int64_t *task_counts)
{
    API_BEGIN
    MPG_CHECK(eng && n >= 0 && (n == 0 || d_pos) && BoxSize > 0 && tree && NTask >= 1, "domain_topleaves: bad argument");
    MPG_HIP(hipSetDevice(eng->device));
    domain_topleaves(n, d_pos, d_garbage, BoxSize, (const TopNode *)tree, size, nleaves, leaf_task, NTask, d_topleaf, d_task, leaf_counts,
                     task_counts, eng->domain, eng->stream);
    API_END
}

__global__ void __launch_bounds__(256) k_include_live(int64_t n, const uint8_t *__restrict__ flags, uint8_t *__restrict__ incl)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        incl[i] = (flags[i] & 3) ? 0 : 1; // garbage and swallowed particles are not in the tree (forcetree.c:357-365)
}

int mpg_dev_fof_fof(mpg_engine *eng, const mpg_fof_params *par, const uint64_t *d_id, const double *d_vel, const double *d_hsml,
                    const unsigned char *d_flags, int64_t *d_grnr, int64_t *ngroups)
{
    API_BEGIN
    MPG_CHECK(eng && par && ngroups && (eng->n == 0 || d_id), "fof_fof: null argument");
    MPG_CHECK(eng->d_pos || eng->n == 0, "fof_fof: no particles bound (mpg_dev_bind_particles)");
    MPG_CHECK(par->FOFHaloComovingLinkingLength > 0 && par->FOFHaloMinLength >= 1, "fof_fof: bad parameters");
    MPG_CHECK((par->FOFPrimaryLinkTypes & par->FOFSecondaryLinkTypes) == 0, "fof_fof: primary and secondary link types must be disjoint");
    MPG_HIP(hipSetDevice(eng->device));
    // the tree of the primary linking particles, no moments (fof.c:178-180)
    const uint8_t *incl = nullptr;
    if(d_flags && eng->n > 0) {
        eng->tree_incl.reserve((size_t)eng->n + 1);
        hipLaunchKernelGGL(k_include_live, dim3((unsigned)((eng->n + 255) / 256)), dim3(256), 0, eng->stream, eng->n, d_flags, eng->tree_incl.p);
        incl = eng->tree_incl.p;
    }
    wait_for_leaf_blocks(eng, eng->stream);
    eng->tree.build(eng->n, eng->d_pos, eng->d_mass, eng->d_type, par->FOFPrimaryLinkTypes, eng->box, eng->stream, &eng->timer, incl);
    eng->tree_allocated = true;
    eng->tree_mask = par->FOFPrimaryLinkTypes;
    eng->full_particle_tree = false;
    eng->sph.hmax_pending = false;
    FofInput in;
    in.n = eng->n;
    in.pos = eng->d_pos;
    in.vel = d_vel;
    in.mass = eng->d_mass;
    in.type = eng->d_type;
    in.flags = d_flags;
    in.id = (const unsigned long long *)d_id;
    in.hsml = d_hsml;
    in.box = eng->box;
    in.LL = par->FOFHaloComovingLinkingLength;
    in.minlen = par->FOFHaloMinLength;
    in.secondary_mask = par->FOFSecondaryLinkTypes;
    *ngroups = eng->fof.run(eng->tree, in, eng->stream);
    if(d_grnr && eng->n > 0)
        MPG_HIP(hipMemcpyAsync(d_grnr, eng->fof.p_grnr.p, (size_t)eng->n * sizeof(int64_t), hipMemcpyDeviceToDevice, eng->stream));
    API_END
}

int mpg_dev_fof_groups(mpg_engine *eng, const mpg_fof_groups *out)
{
    API_BEGIN
    MPG_CHECK(eng && out, "null argument");
    MPG_HIP(hipSetDevice(eng->device));
    FofTable t;
    t.MinID = (unsigned long long *)out->MinID;
    t.Length = out->Length;
    t.GrNr = out->GrNr;
    t.LenType = out->LenType;
    t.Mass = out->Mass;
    t.MassType = out->MassType;
    t.CM = out->CM;
    t.Vel = out->Vel;
    t.Jmom = out->Jmom;
    t.Imom = out->Imom;
    t.FirstPos = out->FirstPos;
    eng->fof.export_groups(t, eng->stream);
    API_END
}

/* ------------------------------ matter power spectrum of the PM step ------------------------------ */
int mpg_gravpm_measure_power(mpg_engine *eng, int on)
{
    API_BEGIN
    MPG_CHECK(eng, "null engine");
    eng->pm.measure_power = on != 0;
    API_END
}

int mpg_dev_gravpm_powerspectrum_raw(mpg_engine *eng, double *d_acc, int64_t *d_modes)
{
    API_BEGIN
    MPG_CHECK(eng && d_acc && d_modes, "null argument");
    MPG_CHECK(eng->pm.nmesh > 0 && eng->pm.ps_valid, "power spectrum: no PM step has been run with the measurement on");
    MPG_HIP(hipSetDevice(eng->device));
    const size_t nb = (size_t)eng->pm.nmesh;
    MPG_HIP(hipMemcpyAsync(d_acc, eng->pm.ps_acc.p, (2 * nb + 1) * sizeof(double), hipMemcpyDeviceToDevice, eng->stream));
    MPG_HIP(hipMemcpyAsync(d_modes, eng->pm.ps_modes.p, nb * sizeof(int64_t), hipMemcpyDeviceToDevice, eng->stream));
    API_END
}

int mpg_powerspectrum_sum(int nbins, const double *acc, const int64_t *modes, double BoxSize_in_MPC, double *kk, double *Power,
                          int64_t *Nmodes, int *nonzero)
{
    API_BEGIN
    MPG_CHECK(nbins > 0 && acc && modes && kk && Power && Nmodes && nonzero, "null argument");
    const double Norm = acc[2 * (size_t)nbins];
    int nz = 0;
    for(int i = 0; i < nbins; i++) { // powerspectrum.c:75-89
        if(modes[i] == 0)
            continue;
        double P = acc[i] / modes[i];
        P /= Norm;
        double k = acc[nbins + i] / modes[i];
        k *= 2 * M_PI / BoxSize_in_MPC;
        P *= pow(BoxSize_in_MPC, 3.0);
        Power[nz] = P;
        kk[nz] = k;
        Nmodes[nz] = modes[i];
        nz++;
    }
    *nonzero = nz;
    API_END
}

int mpg_gravpm_get_powerspectrum(mpg_engine *eng, double BoxSize_in_MPC, double *kk, double *Power, int64_t *Nmodes, int *nonzero)
{
    API_BEGIN
    MPG_CHECK(eng && kk && Power && Nmodes && nonzero, "null argument");
    MPG_CHECK(eng->pm.nmesh > 0 && eng->pm.ps_valid, "power spectrum: no PM step has been run with the measurement on");
    MPG_HIP(hipSetDevice(eng->device));
    const size_t nb = (size_t)eng->pm.nmesh;
    std::vector<double> acc(2 * nb + 1);
    std::vector<int64_t> modes(nb);
    MPG_HIP(hipMemcpyAsync(acc.data(), eng->pm.ps_acc.p, (2 * nb + 1) * sizeof(double), hipMemcpyDeviceToHost, eng->stream));
    MPG_HIP(hipMemcpyAsync(modes.data(), eng->pm.ps_modes.p, nb * sizeof(int64_t), hipMemcpyDeviceToHost, eng->stream));
    MPG_HIP(hipStreamSynchronize(eng->stream));
    if(mpg_powerspectrum_sum((int)nb, acc.data(), modes.data(), BoxSize_in_MPC, kk, Power, Nmodes, nonzero) != 0)
        throw Error(g_err);
    API_END
}

int mpg_powerspectrum_save(const char *OutputDir, const char *filename, double Time, double D1, int nonzero, const double *kk,
                           const double *Power, const int64_t *Nmodes)
{
    API_BEGIN
    MPG_CHECK(OutputDir && filename && kk && Power && Nmodes && nonzero >= 0, "null argument");
    char fname[4096];
    if(Time <= 1e-4) // "avoid -0.0000.txt at high z" (powerspectrum.c:101-105)
        snprintf(fname, sizeof(fname), "%s/%s-%0.4e.txt", OutputDir, filename, Time);
    else
        snprintf(fname, sizeof(fname), "%s/%s-%0.4f.txt", OutputDir, filename, Time);
    FILE *fp = fopen(fname, "w");
    MPG_CHECK(fp != nullptr, std::string("Could not open ") + fname + " for writing");
    fprintf(fp, "# in Mpc/h Units \n");
    fprintf(fp, "# D1 = %g \n", D1);
    fprintf(fp, "# k P N P(z=0)\n");
    for(int i = 0; i < nonzero; i++)
        fprintf(fp, "%g %g %ld %g\n", kk[i], Power[i], (long)Nmodes[i], Power[i] / (D1 * D1));
    fclose(fp);
    API_END
}

/* ------------------------------ snapshot / IC wire format (host IO) ------------------------------ */
int mpg_bigfile_block_info(const char *file, const char *block, mpg_bigblock_info *info)
{
    API_BEGIN
    MPG_CHECK(info, "null argument");
    BigBlockInfo b;
    bigfile_block_info(file, block, &b);
    memcpy(info->dtype, b.dtype, 8);
    info->nmemb = b.nmemb;
    info->nfile = b.nfile;
    info->size = b.size;
    API_END
}

int mpg_bigfile_read_block(const char *file, const char *block, int64_t start, int64_t count, const char *want_dtype, void *out)
{
    API_BEGIN
    MPG_CHECK(out || count == 0, "null argument");
    bigfile_read_block(file, block, start, count, want_dtype, out);
    API_END
}

int mpg_bigfile_write_block(const char *file, const char *block, const char *dtype, int nmemb, int nfile, int64_t size, const char *src_dtype,
                            const void *data)
{
    API_BEGIN
    MPG_CHECK(data || size == 0, "null argument");
    bigfile_write_block(file, block, dtype, nmemb, nfile, size, src_dtype, data);
    API_END
}

int mpg_bigfile_get_attr(const char *file, const char *block, const char *name, const char *want_dtype, void *out, int nmemb)
{
    try {
        MPG_CHECK(name && out, "null argument");
        if(bigfile_get_attr(file, block, name, want_dtype, out, nmemb) != 0) {
            g_err = std::string("bigfile: no attribute `") + name + "'";
            return 2;
        }
    }
    catch(const std::exception &e) {
        g_err = e.what();
        return 1;
    }
    g_err.clear();
    return 0;
}

int mpg_bigfile_set_attr(const char *file, const char *block, const char *name, const char *dtype, const void *data, int nmemb)
{
    API_BEGIN
    MPG_CHECK(data || nmemb == 0, "null argument");
    bigfile_set_attr(file, block, name, dtype, data, nmemb);
    API_END
}

/* ------------------------------ hierarchical gravity (timestep.c:239-599) ------------------------------ */

} // extern "C" (helpers below have C++ linkage)

static inline int64_t dti_from_timebin(int bin) { return bin > 0 ? ((int64_t)1 << bin) : 0; } // timebinmgr.h:47-50
static inline bool is_timebin_active(int i, int64_t current) // timestep.c:143-150
{
    if(i <= 0 || current <= 0)
        return true;
    return current % dti_from_timebin(i) == 0;
}

// build_active_sublist on the device; returns the count (one small D2H copy: the reference needs the count on the host as well)
static int64_t hier_sublist(mpg_engine *eng, const int *list, int64_t nlist, const uint8_t *tb, const uint8_t *flags, int maxtimebin,
                            int64_t Ti_Current, int *out)
{
    eng->hier_val.reserve((size_t)nlist + 1);
    eng->hier_keep.reserve((size_t)nlist + 1);
    eng->hier_cnt.reserve(64);
    launch_sublist_flags(list, nlist, tb, flags, maxtimebin, Ti_Current, eng->hier_val.p, eng->hier_keep.p, eng->stream);
    compact_flagged(eng->hier_val.p, eng->hier_keep.p, nlist, out, eng->hier_cnt.p + 60, eng->hier_tmp, eng->stream);
    unsigned long long c = 0;
    MPG_HIP(hipMemcpyAsync(&c, eng->hier_cnt.p + 60, sizeof(c), hipMemcpyDeviceToHost, eng->stream));
    MPG_HIP(hipStreamSynchronize(eng->stream));
    return (int64_t)c;
}

// grav_short_tree_build_tree (timestep.c:281-291): tree of the listed particles, walk for them into `accel`
static void hier_tree_and_walk(mpg_engine *eng, const mpg_hiergrav_arrays *A, const int *list, int64_t nlist, double *accel, double rho0,
                               int HybridNuGrav)
{
    if(list && nlist == 0)
        return; // nothing active on this rank: an empty tree and an empty walk
    if(mpg_dev_force_tree_active_moments(eng, list, list ? nlist : 0, HybridNuGrav) != 0)
        fail(__FILE__, __LINE__, mpg_last_error());
    if(mpg_dev_grav_short_tree(eng, nullptr, A->d_fulltree_accel, A->d_gravpm, list, nlist, accel, A->d_potential, rho0) != 0)
        fail(__FILE__, __LINE__, mpg_last_error());
    // a tree of all particles: grav_short_postprocess also stores the result in P[].FullTreeGravAccel (gravshort.h:47-67)
    if(eng->full_particle_tree && accel != A->d_fulltree_accel)
        MPG_HIP(hipMemcpyAsync(A->d_fulltree_accel, accel, 3 * (size_t)eng->n * sizeof(double), hipMemcpyDeviceToDevice, eng->stream));
}

// apply_hierarchical_grav_kick, timestep.c:238-278
static void hier_kick(mpg_engine *eng, const mpg_hiergrav_arrays *A, const int *list, int64_t nlist, const double *accel,
                      const mpg_drift_kick_times *times, int ti, int largest_active, mpg_gravkick_fn fn, void *ctx)
{
    const int64_t dti = dti_from_timebin(ti);
    double gravkick = fn(ctx, times->Ti_kick[ti], times->Ti_kick[ti] + dti / 2);
    if(ti < largest_active) {
        const int64_t lowerdti = dti_from_timebin(ti + 1);
        gravkick -= fn(ctx, times->Ti_kick[ti + 1], times->Ti_kick[ti + 1] + lowerdti / 2);
    }
    launch_kick_list(list, nlist, A->d_vel, accel ? accel : A->d_fulltree_accel, A->d_flags, gravkick, eng->stream);
}

static int hier_largest_active(const mpg_drift_kick_times *times, int *ti_out)
{
    int ti, largest_active = MPG_TIMEBINS;
    for(ti = MPG_TIMEBINS; ti >= 0; ti--)
        if(is_timebin_active(ti, times->Ti_Current) && dti_from_timebin(ti) <= times->PM_length) {
            largest_active = ti;
            break;
        }
    if(ti_out)
        *ti_out = ti;
    return largest_active;
}

// timebinmgr.c:372-417 on the host
static double host_dloga_interval(const mpg_timeline *tl, int64_t ti)
{
    const int64_t lastsnap = ti >> MPG_TIMEBINS;
    if(lastsnap >= tl->nsync - 1)
        return 0;
    return (tl->loga[lastsnap + 1] - tl->loga[lastsnap]) / (double)(1ull << MPG_TIMEBINS);
}
static double host_loga_from_ti(const mpg_timeline *tl, int64_t ti)
{
    const int64_t lastsnap = ti >> MPG_TIMEBINS;
    MPG_CHECK(lastsnap >= 0 && lastsnap < tl->nsync, "loga_from_ti: Ti_Current beyond the last sync point");
    const int64_t dti = ti & (((int64_t)1 << MPG_TIMEBINS) - 1);
    return tl->loga[lastsnap] + dti * host_dloga_interval(tl, ti);
}
static int64_t host_ti_from_loga(const mpg_timeline *tl, double loga)
{
    int64_t i;
    for(i = 1; i < tl->nsync - 1; i++)
        if(tl->loga[i] > loga)
            break;
    const double logDTime = (tl->loga[i] - tl->loga[i - 1]) / (double)(1ull << MPG_TIMEBINS);
    int64_t ti = (i - 1) << MPG_TIMEBINS;
    ti = (int64_t)((double)ti + (loga - tl->loga[i - 1]) / logDTime);
    return ti;
}

extern "C" {

int mpg_dev_build_active_sublist(mpg_engine *eng, const int *d_active, int64_t NumActiveParticle, const unsigned char *d_tb_grav,
                                 const unsigned char *d_flags, int maxtimebin, int64_t Ti_Current, int *d_out, int64_t *n_out)
{
    API_BEGIN
    MPG_CHECK(eng && d_tb_grav && d_out && n_out && NumActiveParticle >= 0, "null argument");
    MPG_HIP(hipSetDevice(eng->device));
    *n_out = hier_sublist(eng, d_active, NumActiveParticle, d_tb_grav, d_flags, maxtimebin, Ti_Current, d_out);
    API_END
}

__global__ void __launch_bounds__(256) k_set_bh_bins(int64_t n, const uint8_t *__restrict__ type, uint8_t *__restrict__ tb, int bin)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if(i < n && (type[i] & 7) == 5)
        tb[i] = (uint8_t)bin;
}

int mpg_dev_find_hydro_timesteps(mpg_engine *eng, const mpg_hydrostep_arrays *A, const int *d_active, int64_t NumActiveParticle,
                                 const mpg_drift_kick_times *times, const mpg_timeline *timeline, const mpg_timestep_params *par, double CourantFac,
                                 double atime, double hubble, mpg_hydrostep_result *out)
{
    API_BEGIN
    MPG_CHECK(eng && A && times && timeline && par && out, "null argument");
    MPG_CHECK(A->d_tb_hydro, "find_hydro_timesteps: TimeBinHydro is needed");
    MPG_CHECK(!A->d_type || (A->d_hsml && A->d_maxsignalvel), "find_hydro_timesteps: gas needs Hsml and MaxSignalVel");
    MPG_CHECK(timeline->nsync >= 2 && timeline->loga, "find_hydro_timesteps: the timeline needs at least two sync points");
    MPG_HIP(hipSetDevice(eng->device));
    const int64_t nact = d_active ? NumActiveParticle : eng->n;
    // the timeline for the per-particle conversion, and behind it get_dloga_for_bin of every bin (timebinmgr.c:443-447)
    eng->hier_sp.reserve((size_t)timeline->nsync + MPG_TIMEBINS + 2);
    MPG_HIP(hipMemcpyAsync(eng->hier_sp.p, timeline->loga, (size_t)timeline->nsync * sizeof(double), hipMemcpyHostToDevice, eng->stream));
    double bins[MPG_TIMEBINS + 1];
    const double logDTime = host_dloga_interval(timeline, times->Ti_Current);
    for(int b = 0; b <= MPG_TIMEBINS; b++)
        bins[b] = (double)dti_from_timebin(b) * logDTime;
    double *d_bins = eng->hier_sp.p + timeline->nsync;
    MPG_HIP(hipMemcpyAsync(d_bins, bins, sizeof(bins), hipMemcpyHostToDevice, eng->stream));
    HierTimeline T;
    T.sp = eng->hier_sp.p;
    T.nsync = (int)timeline->nsync;
    T.loga_cur = host_loga_from_ti(timeline, times->Ti_Current);
    T.ti0 = host_ti_from_loga(timeline, T.loga_cur);
    T.MinSizeTimestep = par->MinSizeTimestep;
    eng->hier_cnt.reserve(64);
    unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, (unsigned long long)MPG_TIMEBINS};
    MPG_HIP(hipMemcpyAsync(eng->hier_cnt.p, h, sizeof(h), hipMemcpyHostToDevice, eng->stream));
    const double fac3 = pow(atime, 3 * (1 - 5.0 / 3.0) / 2.0);
    launch_find_hydro_timesteps(d_active, nact, A->d_type, A->d_flags, A->d_hsml, A->d_dthsml, A->d_maxsignalvel, A->d_bh_mintimebin, d_bins,
                                A->d_tb_grav, A->d_tb_hydro, atime, hubble, CourantFac, fac3, T, times->PM_length, times->Ti_Current, eng->hier_cnt.p,
                                eng->stream);
    MPG_HIP(hipMemcpyAsync(h, eng->hier_cnt.p, sizeof(h), hipMemcpyDeviceToHost, eng->stream));
    MPG_HIP(hipStreamSynchronize(eng->stream));
    for(int k = 0; k < 5; k++)
        out->ntitype[k] = (int64_t)h[k];
    out->badstepsizecount = (int64_t)h[5];
    out->badtimebins = (int64_t)h[6];
    out->mTimeBin = (int)h[7];
    API_END
}

// find_timesteps (timestep.c:739-849), the step assignment of a run without SplitGravityTimestepsOn (run.c:756): the particle loop on the
// device; the PM step (get_PM_timestep_ti) comes from the caller, the shrink of the PM step onto the longest tree step and
// times->mintimebin / maxtimebin are done here as the reference does (one rank; several ranks reduce `out` first and call
// mpg_find_timesteps_finish)
int mpg_dev_find_timesteps(mpg_engine *eng, const mpg_hydrostep_arrays *A, const double *d_fulltree_accel, const double *d_gravpm,
                           unsigned char *d_tb_grav, const int *d_active, int64_t NumActiveParticle, mpg_drift_kick_times *times,
                           const mpg_timeline *timeline, const mpg_timestep_params *par, double CourantFac, double atime, double hubble,
                           int64_t dti_max_pm, mpg_timestep_result *out)
{
    API_BEGIN
    MPG_CHECK(eng && A && times && timeline && par && out && d_fulltree_accel && d_gravpm && d_tb_grav, "null argument");
    MPG_CHECK(A->d_tb_hydro, "find_timesteps: TimeBinHydro is needed");
    MPG_CHECK(!A->d_type || (A->d_hsml && A->d_maxsignalvel), "find_timesteps: gas needs Hsml and MaxSignalVel");
    MPG_CHECK(timeline->nsync >= 2 && timeline->loga, "find_timesteps: the timeline needs at least two sync points");
    MPG_CHECK(eng->GravitySoftening > 0, "timestep: gravshort_set_softenings has not been called");
    MPG_HIP(hipSetDevice(eng->device));
    const int64_t nact = d_active ? NumActiveParticle : eng->n;
    // is_PM_timestep, timestep.c:153-159; the new PM step, timestep.c:748-755
    MPG_CHECK(times->Ti_Current <= times->PM_start + times->PM_length, "Passed end of PM step!");
    const bool isPM = times->Ti_Current == times->PM_start + times->PM_length;
    int64_t dti_max = times->PM_length;
    if(isPM) {
        dti_max = dti_max_pm;
        times->PM_length = dti_max;
        times->PM_start = times->PM_kick;
    }
    eng->hier_sp.reserve((size_t)timeline->nsync + MPG_TIMEBINS + 2);
    MPG_HIP(hipMemcpyAsync(eng->hier_sp.p, timeline->loga, (size_t)timeline->nsync * sizeof(double), hipMemcpyHostToDevice, eng->stream));
    double bins[MPG_TIMEBINS + 1];
    const double logDTime = host_dloga_interval(timeline, times->Ti_Current);
    for(int b = 0; b <= MPG_TIMEBINS; b++)
        bins[b] = (double)dti_from_timebin(b) * logDTime;
    double *d_bins = eng->hier_sp.p + timeline->nsync;
    MPG_HIP(hipMemcpyAsync(d_bins, bins, sizeof(bins), hipMemcpyHostToDevice, eng->stream));
    HierTimeline T;
    T.sp = eng->hier_sp.p;
    T.nsync = (int)timeline->nsync;
    T.loga_cur = host_loga_from_ti(timeline, times->Ti_Current);
    T.ti0 = host_ti_from_loga(timeline, T.loga_cur);
    T.MinSizeTimestep = par->MinSizeTimestep;
    eng->hier_cnt.reserve(64);
    unsigned long long h[9] = {0, 0, 0, 0, 0, 0, 0, (unsigned long long)MPG_TIMEBINS, 0};
    MPG_HIP(hipMemcpyAsync(eng->hier_cnt.p, h, sizeof(h), hipMemcpyHostToDevice, eng->stream));
    const double fac3 = pow(atime, 3 * (1 - 5.0 / 3.0) / 2.0);
    launch_find_timesteps(d_active, nact, A->d_type, A->d_flags, d_fulltree_accel, d_gravpm, A->d_hsml, A->d_dthsml, A->d_maxsignalvel,
                          A->d_bh_mintimebin, d_bins, d_tb_grav, A->d_tb_hydro, atime, hubble, par->ErrTolIntAccuracy, 2.8 * eng->GravitySoftening, CourantFac,
                          fac3, T, dti_max, times->Ti_Current, eng->hier_cnt.p, eng->stream);
    MPG_HIP(hipMemcpyAsync(h, eng->hier_cnt.p, sizeof(h), hipMemcpyDeviceToHost, eng->stream));
    MPG_HIP(hipStreamSynchronize(eng->stream));
    for(int k = 0; k < 5; k++)
        out->ntitype[k] = (int64_t)h[k];
    out->badstepsizecount = (int64_t)h[5];
    out->badtimebins = (int64_t)h[6];
    out->mTimeBin = (int)h[7];
    out->maxTimeBin = (int)h[8];
    out->isPM = isPM ? 1 : 0;
    API_END
}

// the tail of find_timesteps (timestep.c:825-848) with the all-reduced smallest / largest bin
int mpg_find_timesteps_finish(int mTimeBin, int maxTimeBin, int isPM, mpg_drift_kick_times *times)
{
    API_BEGIN
    MPG_CHECK(times, "null argument");
    if(isPM && times->PM_length > dti_from_timebin(maxTimeBin))
        times->PM_length = dti_from_timebin(maxTimeBin);
    times->mintimebin = mTimeBin;
    times->maxtimebin = maxTimeBin;
    API_END
}

int mpg_dev_hydro_timesteps_finish(mpg_engine *eng, int mTimeBin, int isFirstTimeStep, int64_t n, const unsigned char *d_type,
                                   unsigned char *d_tb_hydro, mpg_drift_kick_times *times)
{
    API_BEGIN
    MPG_CHECK(eng && times, "null argument");
    // all gas of the shortest bin turned into stars: keep the step, or lengthen it if the next bin is active (timestep.c:709-715)
    if(!is_timebin_active(mTimeBin, times->Ti_Current)) {
        mTimeBin = times->mintimebin;
        if(is_timebin_active(mTimeBin + 1, times->Ti_Current))
            mTimeBin++;
    }
    if(isFirstTimeStep && d_type && d_tb_hydro && n > 0) { // set_bh_first_timestep, timestep.c:600-612
        MPG_HIP(hipSetDevice(eng->device));
        hipLaunchKernelGGL(k_set_bh_bins, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, eng->stream, n, d_type, d_tb_hydro, mTimeBin);
        MPG_HIP(hipGetLastError());
    }
    times->mintimebin = mTimeBin;
    if(times->mintimebin > times->mingravtimebin && times->mingravtimebin > 0) // (a dark-matter particle in the shortest bin)
        times->mintimebin = times->mingravtimebin;
    API_END
}

int mpg_dev_hierarchical_gravity_and_timesteps(mpg_engine *eng, const mpg_hiergrav_arrays *A, const int *d_active, int64_t NumActiveParticle,
                                               int64_t NumActiveGravity, mpg_drift_kick_times *times, const mpg_timeline *timeline,
                                               const mpg_timestep_params *par, double atime, double hubble, int64_t dti_max_pm, double rho0,
                                               int HybridNuGrav, mpg_gravkick_fn gravkick, void *gravkick_ctx, int64_t *badstepsizecount)
{
    API_BEGIN
    MPG_CHECK(eng && A && times && timeline && par && gravkick && badstepsizecount, "null argument");
    MPG_CHECK(A->d_vel && A->d_gravpm && A->d_fulltree_accel && A->d_tb_grav, "hierarchical gravity: Vel, GravPM, FullTreeGravAccel and TimeBinGravity are needed");
    MPG_CHECK(timeline->nsync >= 2 && timeline->loga, "hierarchical gravity: the timeline needs at least two sync points");
    MPG_CHECK(eng->GravitySoftening > 0, "timestep: gravshort_set_softenings has not been called");
    MPG_HIP(hipSetDevice(eng->device));
    const int64_t n = eng->n;
    const int64_t nact = d_active ? NumActiveParticle : n;
    // is_PM_timestep, timestep.c:153-159
    MPG_CHECK(times->Ti_Current <= times->PM_start + times->PM_length, "Passed end of PM step!");
    const bool isPM = times->Ti_Current == times->PM_start + times->PM_length;
    int64_t dti_max = times->PM_length;
    if(isPM) { // timestep.c:303-309
        dti_max = dti_max_pm;
        times->PM_length = dti_max;
        times->PM_start = times->PM_kick;
    }
    int largest_active = hier_largest_active(times, nullptr);
    // the gravitationally active particles (timestep.c:323-328)
    eng->hier_list[0].reserve((size_t)nact + 1);
    eng->hier_list[1].reserve((size_t)nact + 1);
    const int *sub = d_active;
    int64_t nsub = nact;
    int cur = 0; // hier_list[cur] is free
    if(!(NumActiveGravity == NumActiveParticle || isPM)) {
        nsub = hier_sublist(eng, d_active, nact, A->d_tb_grav, A->d_flags, largest_active, times->Ti_Current, eng->hier_list[0].p);
        sub = eng->hier_list[0].p;
        cur = 1;
    }
    // the timeline for the per-particle conversion
    eng->hier_sp.reserve((size_t)timeline->nsync + 1);
    MPG_HIP(hipMemcpyAsync(eng->hier_sp.p, timeline->loga, (size_t)timeline->nsync * sizeof(double), hipMemcpyHostToDevice, eng->stream));
    HierTimeline T;
    T.sp = eng->hier_sp.p;
    T.nsync = (int)timeline->nsync;
    T.loga_cur = host_loga_from_ti(timeline, times->Ti_Current);
    T.ti0 = host_ti_from_loga(timeline, T.loga_cur);
    T.MinSizeTimestep = par->MinSizeTimestep;
    const double soft = 2.8 * eng->GravitySoftening; // FORCE_SOFTENING
    // new gravity bins from the acceleration of the longest step (timestep.c:345-370)
    eng->hier_cnt.reserve(64);
    MPG_HIP(hipMemsetAsync(eng->hier_cnt.p, 0, 64 * sizeof(unsigned long long), eng->stream));
    const double *topacc = A->d_stored_accel ? A->d_stored_accel : A->d_fulltree_accel;
    launch_assign_gravity_bins(sub, nsub, topacc, A->d_gravpm, A->d_flags, atime, hubble, par->ErrTolIntAccuracy, soft, T, dti_max, largest_active,
                               A->d_tb_grav, eng->hier_cnt.p, eng->hier_cnt.p + 48, eng->stream);
    unsigned long long hc[49];
    MPG_HIP(hipMemcpyAsync(hc, eng->hier_cnt.p, sizeof(hc), hipMemcpyDeviceToHost, eng->stream));
    MPG_HIP(hipStreamSynchronize(eng->stream));
    int64_t counts[MPG_TIMEBINS + 1];
    for(int b = 0; b <= MPG_TIMEBINS; b++)
        counts[b] = (int64_t)hc[b];
    int64_t bad = (int64_t)hc[48];
    // largest bin with particles (timestep.c:379-385)
    for(int ti = largest_active; ti >= 1; ti--)
        if(counts[ti] > 0) {
            largest_active = ti;
            break;
        }
    // push the top bin down where it holds too few particles (PM steps only, timestep.c:392-413)
    int push_down_bin = largest_active;
    if(isPM) {
        for(int ti = largest_active; ti >= 1; ti--) {
            if(counts[ti] / 3 > counts[ti - 1])
                break;
            push_down_bin = ti - 1;
            counts[ti - 1] += counts[ti];
        }
    }
    MPG_CHECK(push_down_bin != 0, "Bad timestep: every particle wants the shortest bin");
    if(push_down_bin != largest_active) {
        launch_push_down_bins(sub, nsub, push_down_bin, A->d_tb_grav, eng->stream);
        largest_active = push_down_bin;
    }
    times->maxtimebin = largest_active;
    // the kick of the topmost bin (timestep.c:417)
    hier_kick(eng, A, sub, nsub, A->d_stored_accel, times, largest_active, largest_active, gravkick, gravkick_ctx);
    // all lower bins (timestep.c:433-490)
    eng->hier_accel.reserve(3 * (size_t)n + 3);
    MPG_HIP(hipMemsetAsync(eng->hier_cnt.p + 48, 0, sizeof(unsigned long long), eng->stream));
    const int *last = sub;
    int64_t nlast = nsub;
    for(int ti = largest_active - 1; ti > 0; ti--) {
        int *subl = eng->hier_list[cur].p;
        const int64_t ns = hier_sublist(eng, last, nlast, A->d_tb_grav, A->d_flags, ti, times->Ti_Current, subl);
        if(ns == 0) {
            times->mingravtimebin = ti + 1;
            break;
        }
        hier_tree_and_walk(eng, A, subl, ns, eng->hier_accel.p, rho0, HybridNuGrav);
        launch_level_gravity_bins(subl, ns, eng->hier_accel.p, A->d_gravpm, A->d_flags, atime, hubble, par->ErrTolIntAccuracy, soft, T, dti_max, ti,
                                  A->d_tb_grav, eng->hier_cnt.p + 48, eng->stream);
        hier_kick(eng, A, subl, ns, eng->hier_accel.p, times, ti, largest_active, gravkick, gravkick_ctx);
        last = subl;
        nlast = ns;
        cur ^= 1;
    }
    times->mintimebin = times->mingravtimebin;
    unsigned long long b2 = 0;
    MPG_HIP(hipMemcpyAsync(&b2, eng->hier_cnt.p + 48, sizeof(b2), hipMemcpyDeviceToHost, eng->stream));
    MPG_HIP(hipStreamSynchronize(eng->stream));
    *badstepsizecount = bad + (int64_t)b2;
    API_END
}

int mpg_dev_hierarchical_gravity_accelerations(mpg_engine *eng, const mpg_hiergrav_arrays *A, const int *d_active, int64_t NumActiveParticle,
                                               int64_t NumActiveGravity, mpg_drift_kick_times *times, double rho0, int HybridNuGrav,
                                               mpg_gravkick_fn gravkick, void *gravkick_ctx)
{
    API_BEGIN
    MPG_CHECK(eng && A && times && gravkick, "null argument");
    MPG_CHECK(A->d_vel && A->d_gravpm && A->d_fulltree_accel && A->d_tb_grav, "hierarchical gravity: Vel, GravPM, FullTreeGravAccel and TimeBinGravity are needed");
    MPG_HIP(hipSetDevice(eng->device));
    const int64_t n = eng->n;
    const int64_t nact = d_active ? NumActiveParticle : n;
    int ti = 0;
    const int largest_active = hier_largest_active(times, &ti);
    eng->hier_list[0].reserve((size_t)nact + 1);
    eng->hier_list[1].reserve((size_t)nact + 1);
    eng->hier_accel.reserve(3 * (size_t)n + 3);
    const int *last = d_active;
    int64_t nlast = nact, last_grav = NumActiveGravity;
    int cur = 0;
    if(NumActiveGravity != NumActiveParticle) { // some particles are only hydro active (timestep.c:520-524)
        nlast = hier_sublist(eng, d_active, nact, A->d_tb_grav, A->d_flags, ti, times->Ti_Current, eng->hier_list[0].p);
        last = eng->hier_list[0].p;
        last_grav = nlast;
        cur = 1;
    }
    // all currently active particles: into StoredGravAccel (or, without it, a scratch array: only a full tree then leaves a
    // result, in FullTreeGravAccel, as in the reference where grav_short_tree allocates the array itself)
    hier_tree_and_walk(eng, A, last, nlast, A->d_stored_accel ? A->d_stored_accel : eng->hier_accel.p, rho0, HybridNuGrav);
    hier_kick(eng, A, last, nlast, A->d_stored_accel, times, ti, largest_active, gravkick, gravkick_ctx);
    const double *gravaccel = nullptr;
    for(ti = largest_active - 1; ti >= times->mingravtimebin; ti--) {
        int *subl = eng->hier_list[cur].p;
        const int64_t ns = hier_sublist(eng, last, nlast, A->d_tb_grav, A->d_flags, ti, times->Ti_Current, subl);
        if(ns != last_grav) { // (the same particles as one level up: the accelerations are the same, timestep.c:556-564)
            gravaccel = eng->hier_accel.p;
            hier_tree_and_walk(eng, A, subl, ns, eng->hier_accel.p, rho0, HybridNuGrav);
        }
        hier_kick(eng, A, subl, ns, gravaccel ? gravaccel : A->d_stored_accel, times, ti, largest_active, gravkick, gravkick_ctx);
        last = subl;
        nlast = ns;
        last_grav = ns;
        cur ^= 1;
    }
    API_END
}

/* ------------------------------ host (AoS) path ------------------------------ */

// The host <-> device staging of the AoS path is cut into chunks so that packing / unpacking on the host threads overlaps the
// PCIe transfers of the neighbouring chunks (pinned buffers: the copies are asynchronous).
constexpr int HOST_CHUNKS = 8;
// MPG_HOST_TIMING=1: wall-clock marks of the host forms' phases on stderr (a diagnostic)
struct HostClock {
    bool on;
    const char *name;
    double t0;
    std::string line;
    static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    explicit HostClock(const char *n) : on(getenv("MPG_HOST_TIMING") != nullptr), name(n), t0(now()) {}
    void mark(const char *what)
    {
        if(!on)
            return;
        const double t = now();
        char buf[96];
        snprintf(buf, sizeof(buf), " %s %.2f", what, t - t0);
        line += buf;
        t0 = t;
    }
    ~HostClock()
    {
        if(on)
            fprintf(stderr, "HOST_TIMING %s:%s\n", name, line.c_str());
    }
};
static inline void chunk_range(int64_t n, int c, int64_t &lo, int64_t &hi)
{
    lo = n * c / HOST_CHUNKS;
    hi = n * (c + 1) / HOST_CHUNKS;
}

} // extern "C" (a template)
// unpack(lo, hi) runs on the host for each chunk as soon as the device -> host copies issue(lo, hi) queued for it have landed
template <class Issue, class Unpack> static void download_chunks(mpg_engine *eng, int64_t n, Issue issue, Unpack unpack)
{
    for(int c = 0; c < HOST_CHUNKS; c++) {
        int64_t lo, hi;
        chunk_range(n, c, lo, hi);
        if(hi > lo)
            issue(lo, hi);
        if(!eng->chunk_ev[c])
            MPG_HIP(hipEventCreateWithFlags(&eng->chunk_ev[c], hipEventDisableTiming));
        MPG_HIP(hipEventRecord(eng->chunk_ev[c], eng->stream));
    }
    for(int c = 0; c < HOST_CHUNKS; c++) {
        int64_t lo, hi;
        chunk_range(n, c, lo, hi);
        MPG_HIP(hipEventSynchronize(eng->chunk_ev[c]));
        if(hi > lo)
            parallel_for(hi - lo, [=](int64_t a, int64_t b) { unpack(lo + a, lo + b); });
    }
}
extern "C" {

static void stage_particles_body(mpg_engine *eng, const mpg_particle_view *P, double BoxSize, bool on_prefetch_thread);
static void stage_particles(mpg_engine *eng, const mpg_particle_view *P, double BoxSize)
{
    eng->prefetch_join(); // (a prefetch of this epoch ends here; the write-back thread of gravpm_force is waited for only if P[] is read again)
    if(!eng->prefetch_error.empty()) {
        const std::string e = eng->prefetch_error;
        eng->prefetch_error.clear();
        MPG_CHECK(false, "host path: the prefetch of the particle table failed: " + e);
    }
    stage_particles_body(eng, P, BoxSize, false);
}
static void stage_particles_body(mpg_engine *eng, const mpg_particle_view *P, double BoxSize, bool on_prefetch_thread)
{
    MPG_CHECK(P && (P->n == 0 || P->base), "null particle view");
    MPG_CHECK(P->off_pos >= 0 && P->off_mass >= 0, "particle view needs Pos and Mass");
    const int64_t n = P->n;
    if(eng->resident && eng->res_base == P->base) { // resident mode: the table is the device's
        MPG_CHECK(eng->res_n == n && eng->box == BoxSize && eng->d_pos == eng->s_pos.p,
                  "resident mode: the particle table changed size, box or binding (mpg_resident_end / _begin around anything that reorders P[])");
        return;
    }
    // positions, masses and types of this very table are on the device already (mpg_set_particle_epoch)
    if(eng->host_epoch != 0 && eng->staged_epoch == eng->host_epoch && eng->staged_base == P->base && eng->staged_n == n &&
       eng->staged_box == BoxSize && eng->d_pos == eng->s_pos.p)
        return;
    if(!on_prefetch_thread) // (mpg_host_prefetch joined before it started this thread)
        eng->host_join();   // a pending write-back of the last epoch's GravPM touches the records this pass reads
    eng->h_d.reserve(3 * (size_t)n + 1);
    eng->h_f.reserve((size_t)n + 1);
    eng->h_b.reserve((size_t)n + 1);
    eng->s_pos.reserve(3 * (size_t)n + 1);
    eng->s_mass.reserve((size_t)n + 1);
    eng->s_type.reserve((size_t)n + 1);
    const char *b = (const char *)P->base;
    double *hd = eng->h_d.p;
    float *hf = eng->h_f.p;
    uint8_t *hb = eng->h_b.p;
    const mpg_particle_view V = *P;
    // with overlap, the same pass also takes what the epoch's later calls would read from P[] again: Potential (gravpm_force accumulates
    // into it) and FullTreeGravAccel (the walk's opening criterion)
    const bool extras = eng->host_overlap && eng->host_epoch != 0 && V.off_accel >= 0;
    const bool xpot = extras && V.off_potential >= 0;
    double *hacc = nullptr, *hpot = nullptr;
    if(extras) {
        eng->h_acc.reserve(3 * (size_t)n + 1);
        eng->s_prevacc.reserve(3 * (size_t)n + 1);
        hacc = eng->h_acc.p;
        if(xpot) {
            eng->h_gpot.reserve((size_t)n + 1);
            eng->s_pot.reserve((size_t)n + 1);
            hpot = eng->h_gpot.p;
        }
    }
    int any_dead_store[HOST_CHUNKS] = {};
    int *any_dead = any_dead_store;
    for(int c = 0; c < HOST_CHUNKS; c++) {
        int64_t lo, hi;
        chunk_range(n, c, lo, hi);
        if(hi <= lo)
            continue;
        parallel_for(hi - lo, [=](int64_t a0, int64_t a1) {
            for(int64_t i = lo + a0; i < lo + a1; i++) {
                const char *rec = b + i * V.stride;
                const double *pp = (const double *)(rec + V.off_pos);
                hd[3 * i + 0] = pp[0];
                hd[3 * i + 1] = pp[1];
                hd[3 * i + 2] = pp[2];
                hf[i] = *(const float *)(rec + V.off_mass);
                uint8_t ty = V.off_type >= 0 ? (*(const uint8_t *)(rec + V.off_type) & 7) : 1;
                // garbage / swallowed-BH particles never enter the tree (forcetree.c:806): give them type 7 (no mask bit)
                if(V.off_flags >= 0) {
                    const uint8_t fl = *(const uint8_t *)(rec + V.off_flags);
                    if((fl & 1) || ((fl & 2) && ty == 5))
                        ty = 7;
                }
                hb[i] = ty;
                if(ty == 7)
                    any_dead[c] = 1;
                if(hacc) {
                    const double *aa = (const double *)(rec + V.off_accel);
                    hacc[3 * i + 0] = aa[0];
                    hacc[3 * i + 1] = aa[1];
                    hacc[3 * i + 2] = aa[2];
                    if(hpot)
                        hpot[i] = *(const double *)(rec + V.off_potential);
                }
            }
        });
        MPG_HIP(hipMemcpyAsync(eng->s_pos.p + 3 * lo, hd + 3 * lo, 3 * (hi - lo) * sizeof(double), hipMemcpyHostToDevice, eng->stream));
        MPG_HIP(hipMemcpyAsync(eng->s_mass.p + lo, hf + lo, (hi - lo) * sizeof(float), hipMemcpyHostToDevice, eng->stream));
        MPG_HIP(hipMemcpyAsync(eng->s_type.p + lo, hb + lo, (hi - lo) * sizeof(uint8_t), hipMemcpyHostToDevice, eng->stream));
        if(hpot)
            MPG_HIP(hipMemcpyAsync(eng->s_pot.p + lo, hpot + lo, (hi - lo) * sizeof(double), hipMemcpyHostToDevice, eng->stream));
    }
    eng->staged_extra_epoch = extras ? eng->host_epoch : -1;
    eng->gravpm_epoch = -1;
    // garbage and swallowed particles are not deposited and receive no mesh force either (gravpm.c:176-179: region -2): the PM takes a
    // live flag per particle when the table holds any
    bool dead = false;
    for(int c = 0; c < HOST_CHUNKS; c++)
        dead = dead || any_dead[c];
    eng->pm_live = nullptr;
    if(dead) {
        eng->s_live.reserve((size_t)n + 1);
        MPG_HIP(hipStreamSynchronize(eng->stream)); // (the type bytes have left the staging buffer, which now takes the flags)
        parallel_for(n, [=](int64_t lo, int64_t hi) {
            for(int64_t i = lo; i < hi; i++)
                hb[i] = hb[i] != 7;
        });
        MPG_HIP(hipMemcpyAsync(eng->s_live.p, hb, (size_t)n, hipMemcpyHostToDevice, eng->stream));
        eng->pm_live = eng->s_live.p;
    }
    MPG_HIP(hipStreamSynchronize(eng->stream));
    if(hacc) { // (needed by the walk only: on the copy stream it travels while the PM step computes; its staging buffer is its own)
        if(!eng->copy_stream) {
            MPG_HIP(hipStreamCreateWithFlags(&eng->copy_stream, hipStreamNonBlocking));
            MPG_HIP(hipEventCreateWithFlags(&eng->ev_pm_done, hipEventDisableTiming));
        }
        if(!eng->ev_acc_up)
            MPG_HIP(hipEventCreateWithFlags(&eng->ev_acc_up, hipEventDisableTiming));
        MPG_HIP(hipMemcpyAsync(eng->s_prevacc.p, hacc, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice, eng->copy_stream));
        MPG_HIP(hipEventRecord(eng->ev_acc_up, eng->copy_stream));
    }
    eng->n = n;
    eng->d_pos = eng->s_pos.p;
    eng->d_mass = eng->s_mass.p;
    eng->d_type = eng->s_type.p;
    eng->box = BoxSize;
    eng->staged_epoch = eng->host_epoch;
    eng->staged_base = P->base;
    eng->staged_n = n;
    eng->staged_box = BoxSize;
}

int mpg_set_particle_epoch(mpg_engine *eng, int64_t epoch)
{
    API_BEGIN
    MPG_CHECK(eng, "null engine");
    if(epoch != eng->host_epoch)
        eng->host_join(); // (a prefetch reads host_epoch on its own thread)
    eng->host_epoch = epoch;
    API_END
}

// The epoch's one packing pass and its uploads, started EARLY: a caller that knows P[] is final for the step - the end of
// drift_all_particles (drift.c:84-102), where nothing of run.c touches Pos / Mass / FullTreeGravAccel / Potential again before gravpm_force
// (run.c:420-522: domain_maintain reads them; an exchange or a garbage collection declares a new epoch, and this upload is then simply not used)
// - calls this after mpg_set_particle_epoch; the pass runs on a host thread of its own and the first entry point of the epoch joins it
// instead of packing.  Needs the overlap mode (the pass must also take Potential / FullTreeGravAccel); without it, or resident, a no-op.
int mpg_host_prefetch(mpg_engine *eng, const mpg_particle_view *P, double BoxSize)
{
    API_BEGIN
    MPG_CHECK(eng && P && BoxSize > 0, "mpg_host_prefetch: bad argument");
    MPG_HIP(hipSetDevice(eng->device));
    eng->host_join();
    if(!eng->host_overlap || eng->host_epoch == 0 || eng->resident || P->n == 0) {
        mpg_err_slot().clear();
        return 0;
    }
    eng->prefetch_error.clear();
    eng->prefetch_view = *P;
    eng->prefetch_thread = std::thread([eng, BoxSize] {
        try {
            MPG_HIP(hipSetDevice(eng->device));
            stage_particles_body(eng, &eng->prefetch_view, BoxSize, true);
        }
        catch(const std::exception &e) {
            eng->prefetch_error = e.what();
        }
    });
    API_END
}

int mpg_set_host_overlap(mpg_engine *eng, int on)
{
    API_BEGIN
    MPG_CHECK(eng, "null engine");
    eng->host_join();
    eng->host_overlap = on != 0;
    eng->host_slices = on > 1 ? on : 0; // (2 .. 8: that many slices of the walk whatever the size - the tests' way to the sliced path)
    API_END
}

int mpg_host_results_sync(mpg_engine *eng)
{
    API_BEGIN
    MPG_CHECK(eng, "null engine");
    eng->host_join();
    MPG_CHECK(eng->unpack_error.empty(), "host path: the write-back of GravPM failed: " + eng->unpack_error);
    API_END
}

int mpg_gravpm_force(mpg_engine *eng, const mpg_particle_view *P)
{
    API_BEGIN
    MPG_CHECK(eng && P, "null argument");
    MPG_HIP(hipSetDevice(eng->device));
    MPG_CHECK(eng->pm.nmesh > 0, "gravpm_force called before gravpm_init_periodic");
    MPG_CHECK(P->off_gravpm >= 0, "particle view needs GravPM");
    HostClock hc("gravpm_force");
    eng->host_join(); // (the write-back of an earlier call)
    MPG_CHECK(eng->unpack_error.empty(), "host path: the write-back of GravPM failed: " + eng->unpack_error);
    stage_particles(eng, P, eng->pm.box);
    hc.mark("stage");
    const int64_t n = P->n;
    if(eng->resident && eng->res_base == P->base) { // results stay in HBM: GravPM assigned, Potential accumulated (gravpm.c:499-501)
        if(eng->pm_live) // (gravpm.c:88-92 zeroes GravPM of every particle; the readout reaches the live ones)
            MPG_HIP(hipMemsetAsync(eng->r_gravpm.p, 0, 3 * (size_t)n * sizeof(double), eng->stream));
        eng->pm.force(n, eng->d_pos, eng->d_mass, eng->pm_live, eng->r_gravpm.p, eng->r_pot.p, eng->stream, &eng->timer);
        mpg_err_slot().clear();
        return 0;
    }
    eng->s_gravpm.reserve(3 * (size_t)n + 1);
    const bool wantpot = P->off_potential >= 0;
    char *b = (char *)P->base;
    const mpg_particle_view V = *P;
    eng->h_d2.reserve(3 * (size_t)n + 1);
    eng->h_d3.reserve((size_t)n + 1);
    double *hg = eng->h_d2.p, *hp = eng->h_d3.p;
    const bool overlap = eng->host_overlap && eng->host_epoch != 0;
    if(wantpot && !(overlap && eng->staged_extra_epoch == eng->host_epoch)) { // (with overlap the Potential went up with the positions)
        // readout_potential accumulates into P.Potential (gravpm.c:499-501), which is NOT zeroed first
        eng->s_pot.reserve((size_t)n + 1);
        parallel_for(n, [=](int64_t lo, int64_t hi) {
            for(int64_t i = lo; i < hi; i++)
                hp[i] = *(const double *)(b + i * V.stride + V.off_potential);
        });
        MPG_HIP(hipMemcpyAsync(eng->s_pot.p, hp, n * sizeof(double), hipMemcpyHostToDevice, eng->stream));
    }
    if(eng->pm_live)
        MPG_HIP(hipMemsetAsync(eng->s_gravpm.p, 0, 3 * (size_t)n * sizeof(double), eng->stream));
    eng->pm.force(n, eng->d_pos, eng->d_mass, eng->pm_live, eng->s_gravpm.p, wantpot ? eng->s_pot.p : nullptr, eng->stream, &eng->timer);
    eng->gravpm_epoch = eng->host_epoch;
    const double *dg = eng->s_gravpm.p, *dp = eng->s_pot.p;
    if(overlap) {
        // the results leave on a copy stream behind the PM step and are written into P[] by a host thread, chunk by chunk, while this
        // thread returns and queues the tree build and the walk (the walk reads GravPM and the Potential it accumulates onto from the
        // device buffers, which nothing overwrites before the next gravpm_force)
        if(!eng->copy_stream) {
            MPG_HIP(hipStreamCreateWithFlags(&eng->copy_stream, hipStreamNonBlocking));
            MPG_HIP(hipEventCreateWithFlags(&eng->ev_pm_done, hipEventDisableTiming));
        }
        eng->h_gpm.reserve(3 * (size_t)n + 1);
        double *ag = eng->h_gpm.p, *ap = nullptr;
        if(wantpot) {
            eng->h_gpot.reserve((size_t)n + 1);
            ap = eng->h_gpot.p;
        }
        MPG_HIP(hipEventRecord(eng->ev_pm_done, eng->stream));
        MPG_HIP(hipStreamWaitEvent(eng->copy_stream, eng->ev_pm_done, 0));
        for(int c = 0; c < HOST_CHUNKS; c++) {
            int64_t lo, hi;
            chunk_range(n, c, lo, hi);
            if(hi > lo) {
                MPG_HIP(hipMemcpyAsync(ag + 3 * lo, dg + 3 * lo, 3 * (hi - lo) * sizeof(double), hipMemcpyDeviceToHost, eng->copy_stream));
                if(wantpot)
                    MPG_HIP(hipMemcpyAsync(ap + lo, dp + lo, (hi - lo) * sizeof(double), hipMemcpyDeviceToHost, eng->copy_stream));
            }
            if(!eng->gchunk_ev[c])
                MPG_HIP(hipEventCreateWithFlags(&eng->gchunk_ev[c], hipEventDisableTiming));
            MPG_HIP(hipEventRecord(eng->gchunk_ev[c], eng->copy_stream));
        }
        eng->unpack_error.clear();
        const int device = eng->device;
        eng->unpack_thread = std::thread([=] {
            if(hipSetDevice(device) != hipSuccess) {
                eng->unpack_error = "hipSetDevice";
                return;
            }
            for(int c = 0; c < HOST_CHUNKS; c++) {
                int64_t lo, hi;
                chunk_range(n, c, lo, hi);
                if(hipEventSynchronize(eng->gchunk_ev[c]) != hipSuccess) {
                    eng->unpack_error = "hipEventSynchronize";
                    return;
                }
                if(hi > lo)
                    parallel_for(hi - lo, [=](int64_t a0, int64_t a1) {
                        for(int64_t i = lo + a0; i < lo + a1; i++) {
                            double *g = (double *)(b + i * V.stride + V.off_gravpm);
                            g[0] = ag[3 * i + 0];
                            g[1] = ag[3 * i + 1];
                            g[2] = ag[3 * i + 2];
                            if(ap)
                                *(double *)(b + i * V.stride + V.off_potential) = ap[i];
                        }
                    });
            }
        });
        hc.mark("queued");
        mpg_err_slot().clear();
        return 0;
    }
    hipStream_t st = eng->stream;
    download_chunks(
        eng, n,
        [=](int64_t lo, int64_t hi) {
            MPG_HIP(hipMemcpyAsync(hg + 3 * lo, dg + 3 * lo, 3 * (hi - lo) * sizeof(double), hipMemcpyDeviceToHost, st));
            if(wantpot)
                MPG_HIP(hipMemcpyAsync(hp + lo, dp + lo, (hi - lo) * sizeof(double), hipMemcpyDeviceToHost, st));
        },
        [=](int64_t lo, int64_t hi) {
            for(int64_t i = lo; i < hi; i++) {
                double *g = (double *)(b + i * V.stride + V.off_gravpm);
                g[0] = hg[3 * i + 0];
                g[1] = hg[3 * i + 1];
                g[2] = hg[3 * i + 2];
                if(wantpot)
                    *(double *)(b + i * V.stride + V.off_potential) = hp[i];
            }
        });
    API_END
}

int mpg_force_tree_rebuild_mask(mpg_engine *eng, const mpg_particle_view *P, double BoxSize, int mask)
{
    API_BEGIN
    MPG_CHECK(eng && P, "null argument");
    MPG_HIP(hipSetDevice(eng->device));
    stage_particles(eng, P, BoxSize);
    wait_for_leaf_blocks(eng, eng->stream);
    eng->tree.build(eng->n, eng->d_pos, eng->d_mass, eng->d_type, mask, BoxSize, eng->stream, &eng->timer);
    eng->tree.calc_moments(nullptr, eng->stream, &eng->timer);
    eng->tree_allocated = true;
    eng->tree_mask = mask;
    eng->full_particle_tree = (eng->tree.npart == eng->n) || mask == 63;
    API_END
}

int mpg_force_tree_full(mpg_engine *eng, const mpg_particle_view *P, double BoxSize)
{
    return mpg_force_tree_rebuild_mask(eng, P, BoxSize, 63 /* ALLMASK, forcetree.h:22 */);
}

int mpg_force_tree_free(mpg_engine *eng)
{
    API_BEGIN
    MPG_CHECK(eng, "null engine");
    eng->tree_allocated = false;
    eng->full_particle_tree = false;
    eng->tree.has_moments = false;
    API_END
}

// the results of the targets [lo, hi) of the tree order, compacted in that order for a contiguous copy to the host
__global__ void __launch_bounds__(256) k_gather_results(int64_t lo, int64_t hi, const int *__restrict__ order, const double *__restrict__ acc,
                                                        const double *__restrict__ pot, double *__restrict__ acc_t, double *__restrict__ pot_t)
{
    const int64_t j = lo + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if(j >= hi)
        return;
    const int64_t i = order[j];
    acc_t[3 * j + 0] = acc[3 * i + 0];
    acc_t[3 * j + 1] = acc[3 * i + 1];
    acc_t[3 * j + 2] = acc[3 * i + 2];
    if(pot_t)
        pot_t[j] = pot[i];
}

int mpg_grav_short_tree(mpg_engine *eng, const mpg_particle_view *P, const int *ActiveParticle, int64_t NumActiveParticle,
                        double (*AccelStore)[3], double rho0)
{
    API_BEGIN
    MPG_CHECK(eng && P, "null argument");
    MPG_HIP(hipSetDevice(eng->device));
    MPG_CHECK(eng->tree_allocated && eng->tree.has_moments, "Gravtree called before tree moments computed!");
    MPG_CHECK(P->n == eng->n, "grav_short_tree: particle table changed size since the tree was built");
    MPG_CHECK(P->off_accel >= 0 && P->off_gravpm >= 0, "particle view needs FullTreeGravAccel and GravPM");
    const int64_t n = P->n;
    if(eng->resident && eng->res_base == P->base) {
        // OldAcc from the resident FullTreeGravAccel + GravPM (grav_get_abs_accel, gravshort.h:70-80), results into the resident
        // FullTreeGravAccel / Potential in place: a target's old value is read before its new one is written, and no walk reads
        // another target's acceleration.  AccelStore (host) receives a copy when given (timestep.c:454-456).
        const int *d_act = nullptr;
        if(ActiveParticle) {
            eng->s_active.reserve((size_t)NumActiveParticle + 1);
            MPG_HIP(hipMemcpyAsync(eng->s_active.p, ActiveParticle, NumActiveParticle * sizeof(int), hipMemcpyHostToDevice, eng->stream));
            d_act = eng->s_active.p;
        }
        const bool full = eng->full_particle_tree;
        double *out = eng->r_accel.p;
        if(!full) { // (a tree of a subset: results go to AccelStore only, gravshort.h:54-66)
            eng->s_accel.reserve(3 * (size_t)n + 1);
            MPG_HIP(hipMemsetAsync(eng->s_accel.p, 0, 3 * n * sizeof(double), eng->stream));
            out = eng->s_accel.p;
        }
        if(mpg_dev_grav_short_tree(eng, nullptr, eng->r_accel.p, eng->r_gravpm.p, d_act, NumActiveParticle, out, full ? eng->r_pot.p : nullptr, rho0))
            throw Error(g_err);
        if(AccelStore) {
            eng->h_d2.reserve(3 * (size_t)n + 1);
            double *ha = eng->h_d2.p;
            MPG_HIP(hipMemcpyAsync(ha, out, 3 * n * sizeof(double), hipMemcpyDeviceToHost, eng->stream));
            MPG_HIP(hipStreamSynchronize(eng->stream));
            const int64_t m = ActiveParticle ? NumActiveParticle : n;
            parallel_for(m, [=](int64_t lo, int64_t hi) {
                for(int64_t k = lo; k < hi; k++) {
                    const int64_t i = ActiveParticle ? ActiveParticle[k] : k;
                    AccelStore[i][0] = ha[3 * i + 0];
                    AccelStore[i][1] = ha[3 * i + 1];
                    AccelStore[i][2] = ha[3 * i + 2];
                }
            });
        }
        mpg_err_slot().clear();
        return 0;
    }
    const char *b = (const char *)P->base;
    const mpg_particle_view V = *P;
    // OldAcc = |FullTreeGravAccel + GravPM| / G (grav_short_copy, gravshort.h:82-86).  With overlap, and when this epoch's first call
    // uploaded FullTreeGravAccel and this epoch's gravpm_force left GravPM on the device, it is taken there (k_oldacc / the list kernel:
    // the same arithmetic); otherwise from P[] on the host.
    HostClock hc("grav_short_tree");
    const bool dev_old = eng->host_overlap && eng->host_epoch != 0 && eng->staged_extra_epoch == eng->host_epoch &&
                         eng->gravpm_epoch == eng->host_epoch && eng->staged_base == P->base && eng->staged_n == n;
    eng->s_accel.reserve(3 * (size_t)n + 1);
    eng->s_pot.reserve((size_t)n + 1);
    if(!dev_old) {
        eng->host_join(); // (GravPM is read from P[])
        eng->h_d3.reserve((size_t)n + 1);
        double *old = eng->h_d3.p;
        const double G = eng->pm.G;
        parallel_for(n, [=](int64_t lo, int64_t hi) {
            for(int64_t i = lo; i < hi; i++) {
                const double *a = (const double *)(b + i * V.stride + V.off_accel);
                const double *g = (const double *)(b + i * V.stride + V.off_gravpm);
                double s2 = 0;
                for(int j = 0; j < 3; j++) {
                    const double ax = a[j] + g[j];
                    s2 += ax * ax;
                }
                old[i] = sqrt(s2) / G;
            }
        });
        eng->s_old.reserve((size_t)n + 1);
        MPG_HIP(hipMemcpyAsync(eng->s_old.p, old, n * sizeof(double), hipMemcpyHostToDevice, eng->stream));
    }
    const int *d_act = nullptr;
    if(ActiveParticle) {
        eng->s_active.reserve((size_t)NumActiveParticle + 1);
        MPG_HIP(hipMemcpyAsync(eng->s_active.p, ActiveParticle, NumActiveParticle * sizeof(int), hipMemcpyHostToDevice, eng->stream));
        d_act = eng->s_active.p;
    }
    const bool full = eng->full_particle_tree;
    const bool wantpot = full && P->off_potential >= 0;
    MPG_HIP(hipMemsetAsync(eng->s_accel.p, 0, 3 * n * sizeof(double), eng->stream));
    if(dev_old && eng->ev_acc_up) // (FullTreeGravAccel went up on the copy stream)
        MPG_HIP(hipStreamWaitEvent(eng->stream, eng->ev_acc_up, 0));
    // ---- with overlap, all particles active, a tree of all of them: the walk in SLICES of the tree order, the results of slice k copied
    // down (compacted in tree order: one contiguous copy) and written into P[] by a host thread while slice k + 1 is walked.  The slices
    // are cut at multiples of 8 targets, i.e. between the waves of the list kernel: every target's lists, and with them its sums, are
    // those of the unsliced walk bit for bit.  What stays on the critical path is the last slice's copy and write-back.
    static const int nslices_env = getenv("MPG_HOST_WALK_SLICES") ? atoi(getenv("MPG_HOST_WALK_SLICES")) : 5;
    const int64_t npart = eng->tree.npart;
    const int nslices = eng->host_slices > 1 ? eng->host_slices : (npart >= (1 << 20) ? nslices_env : 1); // (small walks: not worth the calls)
    if(eng->host_overlap && dev_old && !ActiveParticle && !AccelStore && full && eng->copy_stream && nslices > 1 && npart >= 8 * nslices) {
        const int S = nslices < 8 ? nslices : 8;
        const int *d_order = (const int *)eng->tree.idx_b.p; // tree slot -> particle
        eng->h_order.reserve((size_t)npart + 1);
        eng->s_acc_t.reserve(3 * (size_t)npart + 1);
        eng->h_acc_t.reserve(3 * (size_t)npart + 1);
        if(wantpot) {
            eng->s_pot_t.reserve((size_t)npart + 1);
            eng->h_pot_t.reserve((size_t)npart + 1);
            eng->s_pot2.reserve((size_t)n + 1);
        }
        int *h_order = eng->h_order.p;
        double *hat = eng->h_acc_t.p, *hpt = wantpot ? eng->h_pot_t.p : nullptr;
        double *dat = eng->s_acc_t.p, *dpt = wantpot ? eng->s_pot_t.p : nullptr;
        for(int k = 0; k <= S; k++)
            if(!eng->slice_ev[k])
                MPG_HIP(hipEventCreateWithFlags(&eng->slice_ev[k], hipEventDisableTiming));
        // the tree order for the host thread (the tree is complete: force_tree_full waited for it)
        MPG_HIP(hipMemcpyAsync(h_order, d_order, (size_t)npart * sizeof(int), hipMemcpyDeviceToHost, eng->copy_stream));
        // (what nothing hides is the write-back that is still running when the last walk ends.  The writer needs about half as long for a slice's
        // results as the walk of that slice took, so slice k + 1 may be about half of slice k and still cover it: the slices shrink geometrically,
        // MPG_HOST_SLICE_RATIO per step, and the last - the one nothing covers - is the smallest.  1 = equal slices)
        static const double ratio_env = getenv("MPG_HOST_SLICE_RATIO") ? atof(getenv("MPG_HOST_SLICE_RATIO")) : 0.55;
        const double ratio = (ratio_env > 0.05 && ratio_env < 1.0) ? ratio_env : 1.0;
        int64_t cut[9];
        {
            double w = 1.0, tot = 0.0, acc = 0.0;
            for(int k = 0; k < S; k++, w *= ratio)
                tot += w;
            w = 1.0;
            for(int k = 0; k <= S; k++) {
                cut[k] = k == S ? npart : ((int64_t)((double)npart * (acc / tot)) & ~(int64_t)7);
                acc += w;
                w *= ratio;
            }
        }
        char *wbs = (char *)P->base;
        std::string therr;
        std::thread writer;
        std::atomic<int> issued{0}; // slices whose copies are queued and whose event is recorded (an event not yet recorded "is complete")
        std::atomic<bool> abandon{false};
        // (an error thrown below while the writer runs: tell it to stop and wait for it - a joinable std::thread must not be destroyed)
        struct JoinOnExit {
            std::thread &t;
            std::atomic<bool> &stop;
            ~JoinOnExit()
            {
                if(t.joinable()) {
                    stop = true;
                    t.join();
                }
            }
        } join_on_exit{writer, abandon};
        for(int k = 0; k < S; k++) {
            const int64_t lo = cut[k], hi = cut[k + 1];
            if(hi > lo) {
                if(mpg_dev_grav_short_tree(eng, nullptr, eng->s_prevacc.p, eng->s_gravpm.p, d_order + lo, hi - lo, eng->s_accel.p,
                                           wantpot ? eng->s_pot2.p : nullptr, rho0))
                    throw Error(g_err);
                hipLaunchKernelGGL(k_gather_results, dim3((unsigned)((hi - lo + 255) / 256)), dim3(256), 0, eng->stream, lo, hi, d_order, eng->s_accel.p,
                                   wantpot ? eng->s_pot2.p : nullptr, dat, dpt);
            }
            MPG_HIP(hipEventRecord(eng->ev_pm_done, eng->stream)); // (re-used as "slice k gathered")
            MPG_HIP(hipStreamWaitEvent(eng->copy_stream, eng->ev_pm_done, 0));
            if(hi > lo) {
                MPG_HIP(hipMemcpyAsync(hat + 3 * lo, dat + 3 * lo, 3 * (size_t)(hi - lo) * sizeof(double), hipMemcpyDeviceToHost, eng->copy_stream));
                if(wantpot)
                    MPG_HIP(hipMemcpyAsync(hpt + lo, dpt + lo, (size_t)(hi - lo) * sizeof(double), hipMemcpyDeviceToHost, eng->copy_stream));
            }
            MPG_HIP(hipEventRecord(eng->slice_ev[k], eng->copy_stream));
            issued = k + 1;
            if(k == 0) {
                // (the PM step's GravPM / Potential must be in P[] before the tree's Potential goes over it)
                eng->host_join();
                MPG_CHECK(eng->unpack_error.empty(), "host path: the write-back of GravPM failed: " + eng->unpack_error);
                const int device = eng->device;
                mpg_engine *e = eng;
                writer = std::thread([=, &therr, &issued, &abandon] {
                    if(hipSetDevice(device) != hipSuccess) {
                        therr = "hipSetDevice";
                        return;
                    }
                    for(int q = 0; q < S; q++) {
                        while(issued.load() <= q) { // (the main thread is inside the walk of slice q)
                            if(abandon.load())
                                return;
                            std::this_thread::sleep_for(std::chrono::microseconds(50));
                        }
                        if(hipEventSynchronize(e->slice_ev[q]) != hipSuccess) {
                            therr = "hipEventSynchronize";
                            return;
                        }
                        const int64_t a = cut[q], b = cut[q + 1];
                        if(b > a)
                            parallel_for(b - a, [=](int64_t j0, int64_t j1) {
                                for(int64_t j = a + j0; j < a + j1; j++) {
                                    const int64_t i = h_order[j];
                                    double *acc = (double *)(wbs + i * V.stride + V.off_accel);
                                    acc[0] = hat[3 * j + 0];
                                    acc[1] = hat[3 * j + 1];
                                    acc[2] = hat[3 * j + 2];
                                    if(hpt)
                                        *(double *)(wbs + i * V.stride + V.off_potential) = hpt[j];
                                }
                            });
                    }
                });
            }
        }
        hc.mark("walk (sliced)");
        writer.join();
        MPG_CHECK(therr.empty(), "host path: the write-back of the walk's results failed: " + therr);
        hc.mark("last slice down");
        //  P[].FullTreeGravAccel now holds THIS walk's result: the copy staged at the epoch's first upload is no longer
        // what grav_short_copy (gravshort.h:82-86) would read.  A second walk of the same epoch takes OldAcc from P[] again.
        eng->staged_extra_epoch = -1;
        mpg_err_slot().clear();
        return 0;
    }
    // (with overlap the PM step's Potential may still be on its way down from s_pot: the tree's goes to a buffer of its own)
    double *d_treepot = eng->s_pot.p;
    if(eng->host_overlap && wantpot) {
        eng->s_pot2.reserve((size_t)n + 1);
        d_treepot = eng->s_pot2.p;
    }
    int rc = dev_old ? mpg_dev_grav_short_tree(eng, nullptr, eng->s_prevacc.p, eng->s_gravpm.p, d_act, NumActiveParticle, eng->s_accel.p,
                                               wantpot ? d_treepot : nullptr, rho0)
                     : mpg_dev_grav_short_tree(eng, eng->s_old.p, nullptr, nullptr, d_act, NumActiveParticle, eng->s_accel.p,
                                               wantpot ? d_treepot : nullptr, rho0);
    if(rc)
        throw Error(g_err);
    hc.mark("walk");
    // (the PM step's GravPM / Potential must be in P[] before this call's Potential - the tree's, gravshort.h:94-95 - goes over it)
    eng->host_join();
    hc.mark("join");
    MPG_CHECK(eng->unpack_error.empty(), "host path: the write-back of GravPM failed: " + eng->unpack_error);
    eng->h_d2.reserve(3 * (size_t)n + 1);
    eng->h_d3.reserve((size_t)n + 1);
    double *ha = eng->h_d2.p, *hp = eng->h_d3.p; // (OldAcc has been uploaded: its staging buffer is free again)
    char *wb = (char *)P->base;
    // (garbage / swallowed particles are no walk targets, treewalk.c:234: their fields stay as they are.  eng->h_b holds the live flags
    // of the staged table whenever it has dead records: stage_particles)
    const uint8_t *liveflag = eng->pm_live ? eng->h_b.p : nullptr;
    auto put = [=](int64_t i) {
        if(liveflag && !liveflag[i])
            return;
        if(AccelStore) {
            AccelStore[i][0] = ha[3 * i + 0];
            AccelStore[i][1] = ha[3 * i + 1];
            AccelStore[i][2] = ha[3 * i + 2];
        }
        if(full) { // gravshort.h:54-66
            double *a = (double *)(wb + i * V.stride + V.off_accel);
            a[0] = ha[3 * i + 0];
            a[1] = ha[3 * i + 1];
            a[2] = ha[3 * i + 2];
            if(wantpot)
                *(double *)(wb + i * V.stride + V.off_potential) = hp[i];
        }
    };
    const double *da = eng->s_accel.p, *dpot = d_treepot;
    hipStream_t st = eng->stream;
    if(!ActiveParticle) // all particles: unpack chunk by chunk while the later chunks are still on the bus
        download_chunks(
            eng, n,
            [=](int64_t lo, int64_t hi) {
                MPG_HIP(hipMemcpyAsync(ha + 3 * lo, da + 3 * lo, 3 * (hi - lo) * sizeof(double), hipMemcpyDeviceToHost, st));
                if(wantpot)
                    MPG_HIP(hipMemcpyAsync(hp + lo, dpot + lo, (hi - lo) * sizeof(double), hipMemcpyDeviceToHost, st));
            },
            [=](int64_t lo, int64_t hi) {
                for(int64_t i = lo; i < hi; i++)
                    put(i);
            });
    else {
        MPG_HIP(hipMemcpyAsync(ha, da, 3 * n * sizeof(double), hipMemcpyDeviceToHost, st));
        if(wantpot)
            MPG_HIP(hipMemcpyAsync(hp, dpot, n * sizeof(double), hipMemcpyDeviceToHost, st));
        MPG_HIP(hipStreamSynchronize(st));
        parallel_for(NumActiveParticle, [=](int64_t lo, int64_t hi) {
            for(int64_t k = lo; k < hi; k++)
                put(ActiveParticle[k]);
        });
    }
    hc.mark("download+unpack");
    if(full) // (as above: a walk wrote P[].FullTreeGravAccel; the staged copy of the old one must not open nodes for a second walk of the epoch)
        eng->staged_extra_epoch = -1;
    API_END
}

/* ---- device-resident drop-in mode ------------------------------------------------------------------------------------------
 * The host-pointer calls above move Pos / Mass up and GravPM / FullTreeGravAccel / Potential down on every call because the caller
 * may have changed P[] in between: at 256^3 that is half of a step (bench.py host_path).  A caller that lets the engine integrate -
 * mpg_dev_drift_all_particles, mpg_dev_apply_pm_half_kick, mpg_dev_apply_half_kick on the arrays of mpg_resident_arrays - declares
 * the table resident: one upload, then gravpm_force / force_tree_* / grav_short_tree on the same mpg_particle_view run on the device
 * copies and leave their results there; the host asks for the columns its other modules read (mpg_resident_fetch) and hands back
 * what they changed (mpg_resident_push).  Anything that reorders or resizes P[] (domain exchange, garbage collection) goes between
 * mpg_resident_end and a new mpg_resident_begin. */
namespace {
// one column of the AoS table <-> a device array of w doubles per particle
void column_to_device(mpg_engine *eng, const mpg_particle_view &V, int64_t off, int w, double *dev)
{
    const int64_t n = V.n;
    eng->h_d.reserve((size_t)w * n + 1);
    double *h = eng->h_d.p;
    const char *b = (const char *)V.base;
    const int64_t stride = V.stride;
    for(int c = 0; c < HOST_CHUNKS; c++) {
        int64_t lo, hi;
        chunk_range(n, c, lo, hi);
        if(hi <= lo)
            continue;
        parallel_for(hi - lo, [=](int64_t a0, int64_t a1) {
            for(int64_t i = lo + a0; i < lo + a1; i++)
                for(int k = 0; k < w; k++)
                    h[w * i + k] = ((const double *)(b + i * stride + off))[k];
        });
        MPG_HIP(hipMemcpyAsync(dev + w * lo, h + w * lo, (size_t)w * (hi - lo) * sizeof(double), hipMemcpyHostToDevice, eng->stream));
    }
    MPG_HIP(hipStreamSynchronize(eng->stream));
}

void column_to_host(mpg_engine *eng, const mpg_particle_view &V, int64_t off, int w, const double *dev)
{
    const int64_t n = V.n;
    eng->h_d.reserve((size_t)w * n + 1);
    double *h = eng->h_d.p;
    char *b = (char *)V.base;
    const int64_t stride = V.stride;
    hipStream_t st = eng->stream;
    download_chunks(
        eng, n, [=](int64_t lo, int64_t hi) { MPG_HIP(hipMemcpyAsync(h + w * lo, dev + w * lo, (size_t)w * (hi - lo) * sizeof(double), hipMemcpyDeviceToHost, st)); },
        [=](int64_t lo, int64_t hi) {
            for(int64_t i = lo; i < hi; i++)
                for(int k = 0; k < w; k++)
                    ((double *)(b + i * stride + off))[k] = h[w * i + k];
        });
}

void resident_check(mpg_engine *eng, const mpg_particle_view *P)
{
    MPG_CHECK(eng && P, "null argument");
    MPG_CHECK(eng->resident && eng->res_base == P->base && eng->res_n == P->n, "not the resident particle table (mpg_resident_begin first)");
    MPG_HIP(hipSetDevice(eng->device));
}
} // namespace

int mpg_resident_begin(mpg_engine *eng, const mpg_particle_view *P, double BoxSize)
{
    API_BEGIN
    MPG_CHECK(eng && P, "null argument");
    MPG_CHECK(P->off_accel >= 0 && P->off_gravpm >= 0 && P->off_potential >= 0, "resident mode needs FullTreeGravAccel, GravPM and Potential in the view");
    MPG_HIP(hipSetDevice(eng->device));
    eng->host_join(); // (a prefetch in flight writes the staging state set below)
    eng->resident = false;
    eng->staged_epoch = -1; // (force the upload whatever epoch the caller declared)
    stage_particles(eng, P, BoxSize);
    const size_t n = (size_t)P->n;
    eng->r_accel.reserve(3 * n + 3);
    eng->r_gravpm.reserve(3 * n + 3);
    eng->r_pot.reserve(n + 1);
    column_to_device(eng, *P, P->off_accel, 3, eng->r_accel.p);
    column_to_device(eng, *P, P->off_gravpm, 3, eng->r_gravpm.p);
    column_to_device(eng, *P, P->off_potential, 1, eng->r_pot.p);
    eng->res_has_vel = P->off_vel >= 0;
    if(eng->res_has_vel) {
        eng->r_vel.reserve(3 * n + 3);
        column_to_device(eng, *P, P->off_vel, 3, eng->r_vel.p);
    }
    eng->resident = true;
    eng->res_base = P->base;
    eng->res_n = P->n;
    API_END
}

int mpg_resident_arrays(mpg_engine *eng, mpg_resident_view *out)
{
    API_BEGIN
    MPG_CHECK(eng && out && eng->resident, "mpg_resident_arrays: no resident table");
    out->n = eng->res_n;
    out->d_pos = eng->s_pos.p;
    out->d_mass = eng->s_mass.p;
    out->d_type = eng->s_type.p;
    out->d_vel = eng->res_has_vel ? eng->r_vel.p : nullptr; // (a buffer left by an earlier session is not this table's Vel)
    out->d_fulltree_accel = eng->r_accel.p;
    out->d_gravpm = eng->r_gravpm.p;
    out->d_potential = eng->r_pot.p;
    API_END
}

int mpg_resident_fetch(mpg_engine *eng, const mpg_particle_view *P, unsigned fields)
{
    API_BEGIN
    resident_check(eng, P);
    if(fields & MPG_FIELD_POS)
        column_to_host(eng, *P, P->off_pos, 3, eng->s_pos.p);
    if((fields & MPG_FIELD_VEL) && P->off_vel >= 0 && eng->res_has_vel)
        column_to_host(eng, *P, P->off_vel, 3, eng->r_vel.p);
    if(fields & MPG_FIELD_ACCEL)
        column_to_host(eng, *P, P->off_accel, 3, eng->r_accel.p);
    if(fields & MPG_FIELD_GRAVPM)
        column_to_host(eng, *P, P->off_gravpm, 3, eng->r_gravpm.p);
    if(fields & MPG_FIELD_POTENTIAL)
        column_to_host(eng, *P, P->off_potential, 1, eng->r_pot.p);
    API_END
}

int mpg_resident_push(mpg_engine *eng, const mpg_particle_view *P, unsigned fields)
{
    API_BEGIN
    resident_check(eng, P);
    if(fields & MPG_FIELD_POS) {
        column_to_device(eng, *P, P->off_pos, 3, eng->s_pos.p);
        eng->pm_queued = false;
    }
    if((fields & MPG_FIELD_VEL) && P->off_vel >= 0) {
        eng->r_vel.reserve(3 * (size_t)P->n + 3);
        column_to_device(eng, *P, P->off_vel, 3, eng->r_vel.p);
        eng->res_has_vel = true;
    }
    if(fields & MPG_FIELD_ACCEL)
        column_to_device(eng, *P, P->off_accel, 3, eng->r_accel.p);
    if(fields & MPG_FIELD_GRAVPM)
        column_to_device(eng, *P, P->off_gravpm, 3, eng->r_gravpm.p);
    if(fields & MPG_FIELD_POTENTIAL)
        column_to_device(eng, *P, P->off_potential, 1, eng->r_pot.p);
    API_END
}

int mpg_resident_end(mpg_engine *eng, const mpg_particle_view *P)
{
    API_BEGIN
    resident_check(eng, P);
    if(mpg_resident_fetch(eng, P, MPG_FIELD_POS | MPG_FIELD_VEL | MPG_FIELD_ACCEL | MPG_FIELD_GRAVPM | MPG_FIELD_POTENTIAL))
        throw Error(g_err);
    MPG_CHECK(!eng->sph_resident, "mpg_resident_end: the gas arrays are still resident (mpg_resident_sph_end first)");
    eng->resident = false;
    eng->res_has_vel = false;
    eng->res_base = nullptr;
    eng->res_n = -1;
    eng->staged_epoch = -1;
    API_END
}

int mpg_grav_short_pair(mpg_engine *eng, const mpg_particle_view *P, const int *ActiveParticle, int64_t NumActiveParticle, double Rcut, double rho0)
{
    API_BEGIN
    MPG_CHECK(eng && P, "null argument");
    MPG_HIP(hipSetDevice(eng->device));
    MPG_CHECK(eng->tree_allocated, "grav_short_pair: no tree");
    MPG_CHECK(P->n == eng->n, "grav_short_pair: particle table changed size since the tree was built");
    MPG_CHECK(P->off_accel >= 0, "particle view needs FullTreeGravAccel");
    const int64_t n = P->n;
    eng->s_accel.reserve(3 * (size_t)n + 1);
    eng->s_pot.reserve((size_t)n + 1);
    const int *d_act = nullptr;
    if(ActiveParticle) {
        eng->s_active.reserve((size_t)NumActiveParticle + 1);
        MPG_HIP(hipMemcpyAsync(eng->s_active.p, ActiveParticle, NumActiveParticle * sizeof(int), hipMemcpyHostToDevice, eng->stream));
        d_act = eng->s_active.p;
    }
    const bool full = eng->full_particle_tree;
    const bool wantpot = full && P->off_potential >= 0;
    MPG_HIP(hipMemsetAsync(eng->s_accel.p, 0, 3 * n * sizeof(double), eng->stream));
    if(mpg_dev_grav_short_pair(eng, d_act, NumActiveParticle, Rcut, eng->s_accel.p, wantpot ? eng->s_pot.p : nullptr, rho0))
        throw Error(g_err);
    eng->h_d2.reserve(3 * (size_t)n + 1);
    eng->h_d3.reserve((size_t)n + 1);
    double *ha = eng->h_d2.p, *hp = eng->h_d3.p;
    MPG_HIP(hipMemcpyAsync(ha, eng->s_accel.p, 3 * n * sizeof(double), hipMemcpyDeviceToHost, eng->stream));
    if(wantpot)
        MPG_HIP(hipMemcpyAsync(hp, eng->s_pot.p, n * sizeof(double), hipMemcpyDeviceToHost, eng->stream));
    MPG_HIP(hipStreamSynchronize(eng->stream));
    if(full) { // the pair-wise result reaches the caller only through P (gravshort.h:54-66): priv.Accel is freed on return
        const int64_t nt = ActiveParticle ? NumActiveParticle : n;
        char *wb = (char *)P->base;
        const mpg_particle_view V = *P;
        parallel_for(nt, [=](int64_t lo, int64_t hi) {
            for(int64_t k = lo; k < hi; k++) {
                const int64_t i = ActiveParticle ? ActiveParticle[k] : k;
                double *a = (double *)(wb + i * V.stride + V.off_accel);
                a[0] = ha[3 * i + 0];
                a[1] = ha[3 * i + 1];
                a[2] = ha[3 * i + 2];
                if(wantpot)
                    *(double *)(wb + i * V.stride + V.off_potential) = hp[i];
            }
        });
    }
    API_END
}

/* ------------------------------ SPH ------------------------------ */

int mpg_set_densitypar(mpg_engine *eng, const mpg_density_params *dp)
{
    API_BEGIN
    MPG_CHECK(eng && dp, "null argument");
    MPG_CHECK(dp->DensityKernelType == 1 || dp->DensityKernelType == 2 || dp->DensityKernelType == 4, "Density Kernel type is unknown");
    eng->denspar = *dp;
    API_END
}

int mpg_set_hydropar(mpg_engine *eng, const mpg_hydro_params *hp)
{
    API_BEGIN
    MPG_CHECK(eng && hp, "null argument");
    eng->hydropar = *hp;
    API_END
}

double mpg_get_numngb(mpg_engine *eng) { return eng ? sph_desnumngb(eng->denspar) : 0.0; }

static SphView make_sph_view(mpg_engine *eng, const mpg_sph_arrays *A)
{
    MPG_CHECK(A, "null SPH arrays");
    MPG_CHECK(A->hsml && A->vel && A->entropy && A->density && A->dhsmlegyfac && A->divvel && A->curlvel,
              "SPH arrays: hsml, vel, entropy, density, dhsmlegyfac, divvel, curlvel are required");
    SphView v{};
    v.pos = eng->d_pos;
    v.mass = eng->d_mass;
    v.type = eng->d_type;
    v.hsml = A->hsml;
    v.dthsml = A->dthsml;
    v.vel = A->vel;
    v.gacc = A->gacc;
    v.gpm = A->gpm;
    v.hydroacc_in = A->hydroacc_in;
    v.tb_hydro = A->tb_hydro;
    v.tb_grav = A->tb_grav;
    v.entropy = A->entropy;
    v.dtentropy_in = A->dtentropy_in;
    v.density = A->density;
    v.egywtdensity = A->egywtdensity;
    v.dhsmlegyfac = A->dhsmlegyfac;
    v.divvel = A->divvel;
    v.curlvel = A->curlvel;
    v.gradrho = A->gradrho;
    v.hydroacc_out = A->hydroacc_out;
    v.dtentropy_out = A->dtentropy_out;
    v.maxsignalvel = A->maxsignalvel;
    return v;
}

int mpg_dev_force_tree_rebuild_mask(mpg_engine *eng, int mask, int with_moments)
{
    API_BEGIN
    MPG_CHECK(eng, "null engine");
    MPG_HIP(hipSetDevice(eng->device));
    wait_for_leaf_blocks(eng, eng->stream);
    eng->tree.build(eng->n, eng->d_pos, eng->d_mass, eng->d_type, mask, eng->box, eng->stream, &eng->timer);
    if(with_moments)
        eng->tree.calc_moments(nullptr, eng->stream, &eng->timer);
    eng->tree_allocated = true;
    eng->tree_mask = mask;
    eng->full_particle_tree = (mask == 63) || (eng->tree.npart == eng->n);
    eng->sph.hmax_pending = false;
    API_END
}

__global__ void __launch_bounds__(256) k_mark_included(int64_t nact, const int *__restrict__ active, uint8_t *__restrict__ flags)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k < nact)
        flags[active[k]] = 1;
}

int mpg_dev_force_tree_active_moments(mpg_engine *eng, const int *d_active, int64_t nactive, int HybridNuTracer)
{
    API_BEGIN
    MPG_CHECK(eng && (d_active || nactive == 0), "null argument");
    MPG_HIP(hipSetDevice(eng->device));
    const int mask = HybridNuTracer ? (1 + 2 + 16 + 32) : 63; // GASMASK + DMMASK + STARMASK + BHMASK, or ALLMASK (forcetree.c:139-141)
    const uint8_t *incl = nullptr;
    if(d_active) {
        eng->tree_incl.reserve((size_t)eng->n + 1);
        MPG_HIP(hipMemsetAsync(eng->tree_incl.p, 0, (size_t)eng->n, eng->stream));
        if(nactive > 0)
            hipLaunchKernelGGL(k_mark_included, dim3((unsigned)((nactive + 255) / 256)), dim3(256), 0, eng->stream, nactive, d_active,
                               eng->tree_incl.p);
        incl = eng->tree_incl.p;
    }
    wait_for_leaf_blocks(eng, eng->stream);
    eng->tree.build(eng->n, eng->d_pos, eng->d_mass, eng->d_type, mask, eng->box, eng->stream, &eng->timer, incl);
    eng->tree.calc_moments(nullptr, eng->stream, &eng->timer);
    eng->tree_allocated = true;
    eng->tree_mask = mask;
    eng->full_particle_tree = !d_active && eng->tree.npart == eng->n; // forcetree.c:146-147
    eng->sph.hmax_pending = false;
    API_END
}

int mpg_force_tree_active_moments(mpg_engine *eng, const mpg_particle_view *P, double BoxSize, const int *ActiveParticle,
                                  int64_t NumActiveParticle, int HybridNuTracer)
{
    API_BEGIN
    MPG_CHECK(eng && P, "null argument");
    MPG_HIP(hipSetDevice(eng->device));
    stage_particles(eng, P, BoxSize);
    const int *d_act = nullptr;
    if(ActiveParticle) {
        eng->s_active.reserve((size_t)NumActiveParticle + 1);
        MPG_HIP(hipMemcpyAsync(eng->s_active.p, ActiveParticle, NumActiveParticle * sizeof(int), hipMemcpyHostToDevice, eng->stream));
        d_act = eng->s_active.p;
    }
    if(mpg_dev_force_tree_active_moments(eng, d_act, NumActiveParticle, HybridNuTracer))
        throw Error(g_err);
    API_END
}

int mpg_dev_set_init_hsml(mpg_engine *eng, const mpg_sph_arrays *A, double MeanGasSeparation)
{
    API_BEGIN
    MPG_CHECK(eng && eng->tree_allocated, "set_init_hsml: no tree");
    MPG_HIP(hipSetDevice(eng->device));
    const SphView v = make_sph_view(eng, A);
    eng->sph.set_init_hsml(eng->tree, v, eng->denspar, MeanGasSeparation, eng->stream);
    API_END
}

int mpg_dev_density(mpg_engine *eng, const mpg_sph_arrays *A, const mpg_sph_times *T, const int *d_active, int64_t nactive,
                    int update_hsml, int DoEgyDensity, int BlackHoleOn)
{
    API_BEGIN
    MPG_CHECK(eng && T, "null argument");
    MPG_CHECK(eng->tree_allocated, "density: no tree (force_tree_rebuild_mask first)");
    MPG_CHECK((eng->tree_mask & 1) != 0, "density: the tree does not contain gas (GASMASK)"); // treewalk.c:942-943
    MPG_CHECK(!DoEgyDensity || A->egywtdensity, "density: DoEgyDensity needs the egywtdensity array");
    MPG_HIP(hipSetDevice(eng->device));
    const SphView v = make_sph_view(eng, A);
    eng->sph.density(eng->tree, v, *T, eng->denspar, 2.8 * eng->GravitySoftening, d_active, nactive, eng->n, update_hsml, DoEgyDensity,
                     BlackHoleOn, (eng->tree_mask & 32) != 0, eng->stream);
    API_END
}

int mpg_dev_force_tree_calc_hmax(mpg_engine *eng)
{
    API_BEGIN
    MPG_CHECK(eng && eng->tree_allocated, "force_tree_calc_moments: no tree");
    MPG_HIP(hipSetDevice(eng->device));
    eng->sph.calc_hmax(eng->tree, eng->stream);
    API_END
}

int mpg_dev_force_update_hmax(mpg_engine *eng, const double *d_hsml)
{
    API_BEGIN
    MPG_CHECK(eng && d_hsml && eng->tree_allocated, "force_update_hmax: no tree or no Hsml");
    MPG_HIP(hipSetDevice(eng->device));
    SphView v{};
    v.type = eng->d_type;
    v.hsml = const_cast<double *>(d_hsml);
    eng->sph.hsml_view = v;
    eng->sph.hmax_pending = true;
    eng->sph.calc_hmax(eng->tree, eng->stream);
    API_END
}

int mpg_dev_hydro_force(mpg_engine *eng, const mpg_sph_arrays *A, const mpg_sph_times *T, const int *d_active, int64_t nactive)
{
    API_BEGIN
    MPG_CHECK(eng && T, "null argument");
    MPG_CHECK(eng->tree_allocated, "hydro_force: no tree");
    MPG_CHECK(A->hydroacc_out && A->dtentropy_out && A->maxsignalvel, "hydro_force: output arrays are required");
    MPG_CHECK(!eng->hydropar.DensityIndependentSphOn || A->egywtdensity, "hydro_force: pressure-entropy SPH needs egywtdensity");
    MPG_HIP(hipSetDevice(eng->device));
    const SphView v = make_sph_view(eng, A);
    eng->sph.hydro_force(eng->tree, v, *T, eng->denspar, eng->hydropar, d_active, nactive, eng->n, eng->stream);
    API_END
}

/* host-pointer SPH path: stage every array of mpg_sph_arrays in HBM (doubles: hsml, dthsml, vel[3], gacc[3], gpm[3],
 * hydroacc_in[3], entropy, dtentropy_in, density, egywtdensity, dhsmlegyfac, divvel, curlvel, gradrho[3], hydroacc_out[3],
 * dtentropy_out, maxsignalvel; bytes: tb_hydro, tb_grav) */
namespace {
struct SphField {
    int idx;      // slot in h_sph (or h_sph_u8 when width == 0)
    int width;    // doubles per particle; 0 = uint8
    bool in, out; // copied to / from the device
};
// order = field order of mpg_sph_arrays
const SphField SPH_FIELDS[19] = {{0, 1, true, true},  {1, 1, false, true}, {2, 3, true, false},  {3, 3, true, false}, {4, 3, true, false},
                                 {5, 3, true, false}, {0, 0, true, false}, {1, 0, true, false},  {6, 1, true, false}, {7, 1, true, false},
                                 {8, 1, true, true},  {9, 1, true, true},  {10, 1, true, true},  {11, 1, true, true}, {12, 1, true, true},
                                 {13, 3, false, true}, {14, 3, false, true}, {15, 1, false, true}, {16, 1, false, true}};

void stage_sph(mpg_engine *eng, const mpg_sph_arrays *host, mpg_sph_arrays *dev, int64_t n)
{
    // (the staging buffers ARE the resident copies of a gas run: another array set must wait for mpg_resident_sph_end)
    MPG_CHECK(!eng->sph_resident, "SPH host call with arrays other than the resident gas run's (mpg_resident_sph_end first)");
    void *const *hp = (void *const *)host;
    void **dp = (void **)dev;
    for(int f = 0; f < 19; f++) {
        dp[f] = nullptr;
        if(!hp[f])
            continue;
        const SphField &F = SPH_FIELDS[f];
        if(F.width == 0) {
            eng->h_sph_u8[F.idx].reserve((size_t)n + 1);
            dp[f] = eng->h_sph_u8[F.idx].p;
            MPG_HIP(hipMemcpyAsync(dp[f], hp[f], (size_t)n, hipMemcpyHostToDevice, eng->stream));
        }
        else {
            eng->h_sph[F.idx].reserve((size_t)n * F.width + 1);
            dp[f] = eng->h_sph[F.idx].p;
            if(F.in)
                MPG_HIP(hipMemcpyAsync(dp[f], hp[f], (size_t)n * F.width * sizeof(double), hipMemcpyHostToDevice, eng->stream));
            else
                MPG_HIP(hipMemsetAsync(dp[f], 0, (size_t)n * F.width * sizeof(double), eng->stream));
        }
    }
}

void unstage_sph(mpg_engine *eng, const mpg_sph_arrays *host, const mpg_sph_arrays *dev, int64_t n, bool hydro)
{
    void *const *hp = (void *const *)host;
    void *const *dp = (void *const *)dev;
    for(int f = 0; f < 19; f++) {
        const SphField &F = SPH_FIELDS[f];
        if(!hp[f] || !F.out || F.width == 0)
            continue;
        const bool hydro_field = (f >= 16);          // hydroacc_out, dtentropy_out, maxsignalvel
        if(hydro != hydro_field)
            continue;
        MPG_HIP(hipMemcpyAsync(hp[f], dp[f], (size_t)n * F.width * sizeof(double), hipMemcpyDeviceToHost, eng->stream));
    }
    MPG_HIP(hipStreamSynchronize(eng->stream));
}
// the host arrays of a resident gas run (mpg_resident_sph_begin): their device copies are current, nothing is staged
bool sph_is_resident(mpg_engine *eng, const mpg_particle_view *P, const mpg_sph_arrays *A)
{
    return eng->sph_resident && eng->resident && eng->res_base == P->base && A->hsml == eng->res_sph_host.hsml;
}
} // namespace

int mpg_set_init_hsml(mpg_engine *eng, const mpg_particle_view *P, double BoxSize, const mpg_sph_arrays *A, double MeanGasSeparation)
{
    API_BEGIN
    MPG_CHECK(eng && P && A, "null argument");
    MPG_HIP(hipSetDevice(eng->device));
    stage_particles(eng, P, BoxSize);
    mpg_sph_arrays d;
    stage_sph(eng, A, &d, P->n);
    // the reference calls it on the GAS+BH tree with moments (init.c:485-511, test_density.c:86-87)
    if(mpg_dev_force_tree_rebuild_mask(eng, 1 + 32, 1) || mpg_dev_set_init_hsml(eng, &d, MeanGasSeparation))
        throw Error(g_err);
    MPG_HIP(hipMemcpyAsync(A->hsml, d.hsml, P->n * sizeof(double), hipMemcpyDeviceToHost, eng->stream));
    MPG_HIP(hipStreamSynchronize(eng->stream));
    API_END
}

int mpg_density(mpg_engine *eng, const mpg_particle_view *P, double BoxSize, const mpg_sph_arrays *A, const mpg_sph_times *T,
                const int *ActiveParticle, int64_t NumActiveParticle, int update_hsml, int DoEgyDensity, int BlackHoleOn)
{
    API_BEGIN
    MPG_CHECK(eng && P && A && T, "null argument");
    MPG_HIP(hipSetDevice(eng->device));
    stage_particles(eng, P, BoxSize);
    mpg_sph_arrays d;
    const bool res = sph_is_resident(eng, P, A); // a resident gas run: the arrays are in HBM already and stay there
    if(res)
        d = eng->res_sph_dev;
    else
        stage_sph(eng, A, &d, P->n);
    const int *d_act = nullptr;
    if(ActiveParticle) {
        eng->s_active.reserve((size_t)NumActiveParticle + 1);
        MPG_HIP(hipMemcpyAsync(eng->s_active.p, ActiveParticle, NumActiveParticle * sizeof(int), hipMemcpyHostToDevice, eng->stream));
        d_act = eng->s_active.p;
    }
    if(mpg_dev_force_tree_rebuild_mask(eng, 1 /* GASMASK */, 0) ||
       mpg_dev_density(eng, &d, T, d_act, NumActiveParticle, update_hsml, DoEgyDensity, BlackHoleOn) ||
       (update_hsml && mpg_dev_force_tree_calc_hmax(eng)))
        throw Error(g_err);
    if(!res)
        unstage_sph(eng, A, &d, P->n, false);
    API_END
}

int mpg_hydro_force(mpg_engine *eng, const mpg_particle_view *P, const mpg_sph_arrays *A, const mpg_sph_times *T,
                    const int *ActiveParticle, int64_t NumActiveParticle)
{
    API_BEGIN
    MPG_CHECK(eng && P && A && T, "null argument");
    MPG_HIP(hipSetDevice(eng->device));
    MPG_CHECK(eng->tree_allocated && eng->tree.has_hmax, "Hydro called before hmax computed"); // hydra.c:172-173
    MPG_CHECK(P->n == eng->n, "hydro_force: particle table changed size since density()");
    mpg_sph_arrays d;
    const bool res = sph_is_resident(eng, P, A);
    if(res)
        d = eng->res_sph_dev;
    else
        stage_sph(eng, A, &d, P->n);
    const int *d_act = nullptr;
    if(ActiveParticle) {
        eng->s_active.reserve((size_t)NumActiveParticle + 1);
        MPG_HIP(hipMemcpyAsync(eng->s_active.p, ActiveParticle, NumActiveParticle * sizeof(int), hipMemcpyHostToDevice, eng->stream));
        d_act = eng->s_active.p;
    }
    if(mpg_dev_hydro_force(eng, &d, T, d_act, NumActiveParticle))
        throw Error(g_err);
    if(!res)
        unstage_sph(eng, A, &d, P->n, true);
    API_END
}

/* ---- a resident gas run: the SPH arrays stay in HBM between the calls, the integrator runs there (include/mpgadget_hip.h) ---- */
namespace {
__global__ void __launch_bounds__(256) k_flags_from_type(int64_t n, const uint8_t *__restrict__ type, uint8_t *__restrict__ flags)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if(i < n)
        flags[i] = (type[i] & 7) == 7 ? 1 : 0; // (stage_particles gave garbage and swallowed particles type 7)
}
void resident_sph_check(mpg_engine *eng, const mpg_particle_view *P)
{
    resident_check(eng, P);
    MPG_CHECK(eng->sph_resident, "no resident gas arrays (mpg_resident_sph_begin first)");
}
} // namespace

int mpg_resident_sph_begin(mpg_engine *eng, const mpg_particle_view *P, const mpg_sph_arrays *A)
{
    API_BEGIN
    MPG_CHECK(eng && P && A, "null argument");
    resident_check(eng, P);
    MPG_CHECK(eng->res_has_vel, "mpg_resident_sph_begin: the resident table has no Vel column (the view's off_vel)");
    MPG_CHECK(A->hsml && A->entropy && A->density && A->dhsmlegyfac && A->divvel && A->curlvel && A->hydroacc_out && A->dtentropy_out &&
                  A->maxsignalvel,
              "mpg_resident_sph_begin: hsml, entropy, density, dhsmlegyfac, divvel, curlvel, hydroacc_out, dtentropy_out and maxsignalvel are required");
    const int64_t n = P->n;
    mpg_sph_arrays d;
    stage_sph(eng, A, &d, n);
    // the arrays stage_sph only clears hold state of the previous step that the predictions and the drift read: DtHsml, HydroAccel, DtEntropy
    void *const *hp = (void *const *)A;
    void **dp = (void **)&d;
    for(int f = 0; f < 19; f++) {
        const SphField &F = SPH_FIELDS[f];
        if(hp[f] && F.width > 0 && !F.in)
            MPG_HIP(hipMemcpyAsync(dp[f], hp[f], (size_t)n * F.width * sizeof(double), hipMemcpyHostToDevice, eng->stream));
    }
    // one field each in the reference: SphP.HydroAccel and SphP.DtEntropy are what the next step's predictions read; P[].Vel,
    // FullTreeGravAccel and GravPM are the resident table's
    d.hydroacc_in = d.hydroacc_out;
    d.dtentropy_in = d.dtentropy_out;
    d.vel = eng->r_vel.p;
    d.gacc = eng->r_accel.p;
    d.gpm = eng->r_gravpm.p;
    eng->r_flags.reserve((size_t)n + 1);
    if(n > 0)
        hipLaunchKernelGGL(k_flags_from_type, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, eng->stream, n, eng->s_type.p, eng->r_flags.p);
    MPG_HIP(hipStreamSynchronize(eng->stream));
    eng->res_sph_host = *A;
    eng->res_sph_dev = d;
    eng->sph_resident = true;
    API_END
}

int mpg_resident_sph_arrays(mpg_engine *eng, mpg_sph_arrays *out)
{
    API_BEGIN
    MPG_CHECK(eng && out && eng->sph_resident, "mpg_resident_sph_arrays: no resident gas arrays");
    *out = eng->res_sph_dev;
    API_END
}

int mpg_resident_sph_end(mpg_engine *eng, const mpg_sph_arrays *A)
{
    API_BEGIN
    MPG_CHECK(eng && A && eng->sph_resident, "mpg_resident_sph_end: no resident gas arrays");
    MPG_CHECK(A->hsml == eng->res_sph_host.hsml, "mpg_resident_sph_end: not the arrays mpg_resident_sph_begin took");
    MPG_HIP(hipSetDevice(eng->device));
    const int64_t n = eng->res_n;
    void *const *hp = (void *const *)A;
    void *const *dp = (void *const *)&eng->res_sph_dev;
    for(int f = 0; f < 19; f++) {
        const SphField &F = SPH_FIELDS[f];
        // everything the device may have changed: the outputs, Entropy (kicks), the time bins; not the aliases of the table's columns
        const bool table_alias = (f >= 2 && f <= 4), pred_alias = (f == 5 || f == 9);
        if(!hp[f] || table_alias || pred_alias)
            continue;
        const size_t bytes = F.width == 0 ? (size_t)n : (size_t)n * F.width * sizeof(double);
        MPG_HIP(hipMemcpyAsync(hp[f], dp[f], bytes, hipMemcpyDeviceToHost, eng->stream));
    }
    MPG_HIP(hipStreamSynchronize(eng->stream));
    eng->sph_resident = false;
    API_END
}

// the resident time bins into host arrays (n bytes each; either may be NULL): build_active_particles (timestep.c:1333-1420) reads
// P[].TimeBinHydro / TimeBinGravity on the host at the top of every step
int mpg_resident_fetch_timebins(mpg_engine *eng, unsigned char *tb_hydro, unsigned char *tb_grav)
{
    API_BEGIN
    MPG_CHECK(eng && eng->sph_resident, "mpg_resident_fetch_timebins: no resident gas arrays");
    MPG_HIP(hipSetDevice(eng->device));
    const int64_t n = eng->res_n;
    const mpg_sph_arrays &d = eng->res_sph_dev;
    MPG_CHECK((!tb_hydro || d.tb_hydro) && (!tb_grav || d.tb_grav), "mpg_resident_fetch_timebins: the resident arrays have no such bins");
    if(tb_hydro)
        MPG_HIP(hipMemcpyAsync(tb_hydro, d.tb_hydro, (size_t)n, hipMemcpyDeviceToHost, eng->stream));
    if(tb_grav)
        MPG_HIP(hipMemcpyAsync(tb_grav, d.tb_grav, (size_t)n, hipMemcpyDeviceToHost, eng->stream));
    MPG_HIP(hipStreamSynchronize(eng->stream));
    API_END
}

int mpg_resident_drift_all_particles(mpg_engine *eng, const mpg_particle_view *P, double ddrift, const double random_shift[3])
{
    API_BEGIN
    MPG_CHECK(eng && P && random_shift, "null argument");
    resident_check(eng, P);
    MPG_CHECK(eng->res_has_vel, "resident drift: the resident table has no Vel column");
    const bool gas = eng->sph_resident;
    eng->r_flags.reserve((size_t)P->n + 1);
    if(!gas && P->n > 0)
        hipLaunchKernelGGL(k_flags_from_type, dim3((unsigned)((P->n + 255) / 256)), dim3(256), 0, eng->stream, P->n, eng->s_type.p, eng->r_flags.p);
    if(mpg_dev_drift_all_particles(eng, P->n, eng->s_pos.p, eng->r_vel.p, eng->s_type.p, eng->r_flags.p, gas ? eng->res_sph_dev.hsml : nullptr,
                                   gas ? eng->res_sph_dev.dthsml : nullptr, ddrift, eng->box, random_shift))
        throw Error(g_err);
    API_END
}

int mpg_resident_apply_pm_half_kick(mpg_engine *eng, const mpg_particle_view *P, double Fgravkick)
{
    API_BEGIN
    MPG_CHECK(eng && P, "null argument");
    resident_check(eng, P);
    MPG_CHECK(eng->res_has_vel, "resident kick: the resident table has no Vel column");
    eng->r_flags.reserve((size_t)P->n + 1);
    if(!eng->sph_resident && P->n > 0)
        hipLaunchKernelGGL(k_flags_from_type, dim3((unsigned)((P->n + 255) / 256)), dim3(256), 0, eng->stream, P->n, eng->s_type.p, eng->r_flags.p);
    if(mpg_dev_apply_pm_half_kick(eng, P->n, eng->r_vel.p, eng->r_gravpm.p, eng->r_flags.p, Fgravkick))
        throw Error(g_err);
    API_END
}

int mpg_resident_apply_half_kick(mpg_engine *eng, const mpg_particle_view *P, const int *ActiveParticle, int64_t NumActiveParticle,
                                 const mpg_kick_factors *K)
{
    API_BEGIN
    MPG_CHECK(eng && P && K, "null argument");
    resident_sph_check(eng, P);
    const mpg_sph_arrays &d = eng->res_sph_dev;
    const int *d_act = nullptr;
    if(ActiveParticle) {
        eng->s_active.reserve((size_t)NumActiveParticle + 1);
        MPG_HIP(hipMemcpyAsync(eng->s_active.p, ActiveParticle, NumActiveParticle * sizeof(int), hipMemcpyHostToDevice, eng->stream));
        d_act = eng->s_active.p;
    }
    if(mpg_dev_apply_half_kick(eng, P->n, d_act, NumActiveParticle, eng->r_vel.p, eng->r_accel.p, eng->s_type.p, eng->r_flags.p, d.tb_grav, d.tb_hydro,
                               d.hydroacc_out, (double *)d.entropy, d.dtentropy_out, K))
        throw Error(g_err);
    API_END
}

int mpg_resident_find_hydro_timesteps(mpg_engine *eng, const mpg_particle_view *P, const int *ActiveParticle, int64_t NumActiveParticle,
                                      mpg_drift_kick_times *times, const mpg_timeline *timeline, const mpg_timestep_params *par, double CourantFac,
                                      double atime, double hubble, int isFirstTimeStep, mpg_hydrostep_result *out)
{
    API_BEGIN
    MPG_CHECK(eng && P && times && out, "null argument");
    resident_sph_check(eng, P);
    const mpg_sph_arrays &d = eng->res_sph_dev;
    MPG_CHECK(d.tb_hydro, "resident find_hydro_timesteps: the gas arrays have no TimeBinHydro");
    const int *d_act = nullptr;
    if(ActiveParticle) {
        eng->s_active.reserve((size_t)NumActiveParticle + 1);
        MPG_HIP(hipMemcpyAsync(eng->s_active.p, ActiveParticle, NumActiveParticle * sizeof(int), hipMemcpyHostToDevice, eng->stream));
        d_act = eng->s_active.p;
    }
    mpg_hydrostep_arrays H{};
    H.d_type = eng->s_type.p;
    H.d_flags = eng->r_flags.p;
    H.d_hsml = d.hsml;
    H.d_dthsml = d.dthsml;
    H.d_maxsignalvel = d.maxsignalvel;
    H.d_tb_grav = d.tb_grav;
    H.d_tb_hydro = (unsigned char *)d.tb_hydro;
    if(mpg_dev_find_hydro_timesteps(eng, &H, d_act, NumActiveParticle, times, timeline, par, CourantFac, atime, hubble, out) ||
       mpg_dev_hydro_timesteps_finish(eng, out->mTimeBin, isFirstTimeStep, P->n, eng->s_type.p, (unsigned char *)d.tb_hydro, times))
        throw Error(g_err);
    API_END
}

int mpg_resident_find_timesteps(mpg_engine *eng, const mpg_particle_view *P, const int *ActiveParticle, int64_t NumActiveParticle,
                                mpg_drift_kick_times *times, const mpg_timeline *timeline, const mpg_timestep_params *par, double CourantFac,
                                double atime, double hubble, int64_t dti_max_pm, mpg_timestep_result *out)
{
    API_BEGIN
    MPG_CHECK(eng && P && times && out, "null argument");
    resident_sph_check(eng, P);
    const mpg_sph_arrays &d = eng->res_sph_dev;
    MPG_CHECK(d.tb_hydro && d.tb_grav, "resident find_timesteps: the gas arrays have no time bins");
    const int *d_act = nullptr;
    if(ActiveParticle) {
        eng->s_active.reserve((size_t)NumActiveParticle + 1);
        MPG_HIP(hipMemcpyAsync(eng->s_active.p, ActiveParticle, NumActiveParticle * sizeof(int), hipMemcpyHostToDevice, eng->stream));
        d_act = eng->s_active.p;
    }
    mpg_hydrostep_arrays H{};
    H.d_type = eng->s_type.p;
    H.d_flags = eng->r_flags.p;
    H.d_hsml = d.hsml;
    H.d_dthsml = d.dthsml;
    H.d_maxsignalvel = d.maxsignalvel;
    H.d_tb_grav = d.tb_grav;
    H.d_tb_hydro = (unsigned char *)d.tb_hydro;
    if(mpg_dev_find_timesteps(eng, &H, eng->r_accel.p, eng->r_gravpm.p, (unsigned char *)d.tb_grav, d_act, NumActiveParticle, times, timeline, par,
                              CourantFac, atime, hubble, dti_max_pm, out) ||
       mpg_find_timesteps_finish(out->mTimeBin, out->maxTimeBin, out->isPM, times))
        throw Error(g_err);
    API_END
}

int mpg_sph_get_stats(mpg_engine *eng, int64_t stats[4])
{
    API_BEGIN
    MPG_CHECK(eng && stats, "null argument");
    stats[0] = eng->sph.last_iterations;
    stats[1] = eng->sph.last_targets;
    stats[2] = eng->sph.last_interactions;
    stats[3] = eng->sph.last_candidates;
    API_END
}

/* ------------------------------ introspection ------------------------------ */

int mpg_tree_get_stats(mpg_engine *eng, mpg_tree_stats *st)
{
    API_BEGIN
    MPG_CHECK(eng && st, "null argument");
    MPG_CHECK(eng->tree_allocated, "no tree");
    MPG_HIP(hipSetDevice(eng->device));
    const TreeBuilder &t = eng->tree;
    st->NumParticles = t.npart;
    st->numnodes = t.nnodes;
    st->maxlevel = t.maxlevel;
    std::vector<NodeLink> lk(t.nnodes);
    MPG_HIP(hipStreamSynchronize(eng->stream));
    MPG_HIP(hipMemcpy(lk.data(), t.link.p, t.nnodes * sizeof(NodeLink), hipMemcpyDeviceToHost));
    int64_t nl = 0;
    for(auto &l : lk)
        nl += l.pcount > 0;
    st->numleaves = nl;
    Src4 root{};
    if(t.has_moments)
        MPG_HIP(hipMemcpy(&root, t.src.p + t.npart, sizeof(Src4), hipMemcpyDeviceToHost));
    st->root_mass = root.m;
    st->root_cofm[0] = root.x;
    st->root_cofm[1] = root.y;
    st->root_cofm[2] = root.z;
    st->root_hmax = 0;
    if(t.has_hmax)
        MPG_HIP(hipMemcpy(&st->root_hmax, t.hmax.p, sizeof(double), hipMemcpyDeviceToHost));
    API_END
}

int mpg_tree_export(mpg_engine *eng, int32_t *level, double *center, double *len, double *cofm, double *mass, double *hmax,
                    int32_t *sibling, int32_t *pstart, int32_t *pcount)
{
    API_BEGIN
    MPG_CHECK(eng, "null engine");
    MPG_CHECK(eng->tree_allocated, "no tree");
    MPG_HIP(hipSetDevice(eng->device));
    const TreeBuilder &t = eng->tree;
    MPG_HIP(hipStreamSynchronize(eng->stream));
    std::vector<NodeLink> lk(t.nnodes);
    std::vector<NodeGeo> g(t.nnodes);
    std::vector<Src4> m(t.nnodes);
    MPG_HIP(hipMemcpy(lk.data(), t.link.p, t.nnodes * sizeof(NodeLink), hipMemcpyDeviceToHost));
    MPG_HIP(hipMemcpy(g.data(), t.geo.p, t.nnodes * sizeof(NodeGeo), hipMemcpyDeviceToHost));
    if(t.has_moments)
        MPG_HIP(hipMemcpy(m.data(), t.src.p + t.npart, t.nnodes * sizeof(Src4), hipMemcpyDeviceToHost));
    std::vector<double> hm;
    if(hmax && t.has_hmax) {
        hm.resize(t.nnodes);
        MPG_HIP(hipMemcpy(hm.data(), t.hmax.p, t.nnodes * sizeof(double), hipMemcpyDeviceToHost));
    }
    for(int64_t j = 0; j < t.nnodes; j++) {
        if(level)
            level[j] = lk[j].level;
        if(sibling)
            sibling[j] = lk[j].sibling;
        if(pstart)
            pstart[j] = lk[j].pstart;
        if(pcount)
            pcount[j] = lk[j].pcount;
        if(center) {
            center[3 * j + 0] = g[j].cx;
            center[3 * j + 1] = g[j].cy;
            center[3 * j + 2] = g[j].cz;
        }
        if(len)
            len[j] = g[j].len;
        if(cofm) {
            cofm[3 * j + 0] = m[j].x;
            cofm[3 * j + 1] = m[j].y;
            cofm[3 * j + 2] = m[j].z;
        }
        if(mass)
            mass[j] = m[j].m;
        if(hmax)
            hmax[j] = hm.empty() ? 0.0 : hm[j];
    }
    API_END
}

int mpg_tree_export_order(mpg_engine *eng, int32_t *order)
{
    API_BEGIN
    MPG_CHECK(eng && order, "null argument");
    MPG_CHECK(eng->tree_allocated, "no tree");
    MPG_HIP(hipSetDevice(eng->device));
    MPG_HIP(hipStreamSynchronize(eng->stream));
    MPG_HIP(hipMemcpy(order, eng->tree.idx_b.p, eng->tree.npart * sizeof(int32_t), hipMemcpyDeviceToHost));
    API_END
}

int mpg_walk_get_counters(mpg_engine *eng, int64_t counters[10])
{
    API_BEGIN
    MPG_CHECK(eng && counters, "null argument");
    MPG_HIP(hipSetDevice(eng->device));
    MPG_HIP(hipStreamSynchronize(eng->stream));
    unsigned long long c[16] = {0};
    MPG_HIP(hipMemcpy(c, eng->counters.p, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    counters[0] = (int64_t)c[0];
    counters[1] = (int64_t)c[1];
    counters[2] = (int64_t)c[2];
    counters[3] = eng->last_targets;
    for(int k = 0; k < 6; k++)
        counters[4 + k] = (int64_t)c[3 + k];
    API_END
}

int mpg_walk_get_f32_stats(mpg_engine *eng, int64_t out[2])
{
    API_BEGIN
    MPG_CHECK(eng && out, "null argument");
    MPG_HIP(hipSetDevice(eng->device));
    MPG_HIP(hipStreamSynchronize(eng->stream));
    unsigned long long c[16] = {0};
    MPG_HIP(hipMemcpy(c, eng->counters.p, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    out[0] = (int64_t)c[12];
    out[1] = (int64_t)c[13];
    API_END
}

int mpg_get_phase_times(mpg_engine *eng, mpg_phase_times *t)
{
    API_BEGIN
    MPG_CHECK(eng && t, "null argument");
    *t = eng->timer.t;
    API_END
}

int mpg_set_instrumentation(mpg_engine *eng, int timing, int counters)
{
    API_BEGIN
    MPG_CHECK(eng, "null engine");
    eng->timer.enabled = timing != 0;
    eng->count = counters != 0;
    API_END
}

int mpg_walk_events_collect2(mpg_engine *eng, double *total_ms, int *count, double *lists_ms, double *eval_ms, int *count_split)
{
    API_BEGIN
    MPG_CHECK(eng && total_ms && count, "null argument");
    MPG_HIP(hipSetDevice(eng->device));
    MPG_HIP(hipStreamSynchronize(eng->stream));
    double tot = 0, tl = 0, te = 0;
    int c = 0, cs = 0;
    for(size_t k = 0; k < eng->walk_events.size(); k++) {
        auto &ev = eng->walk_events[k];
        float ms = 0;
        MPG_HIP(hipEventSynchronize(ev.second));
        MPG_HIP(hipEventElapsedTime(&ms, ev.first, ev.second));
        tot += ms;
        c++;
        if(hipEvent_t mid = k < eng->walk_mid.size() ? eng->walk_mid[k] : nullptr) {
            float a = 0, b = 0;
            MPG_HIP(hipEventElapsedTime(&a, ev.first, mid));
            MPG_HIP(hipEventElapsedTime(&b, mid, ev.second));
            tl += a;
            te += b;
            cs++;
            eng->free_mid.push_back(mid);
        }
        eng->free_events.push_back(ev);
    }
    eng->walk_events.clear();
    eng->walk_mid.clear();
    *total_ms = tot;
    *count = c;
    if(lists_ms)
        *lists_ms = tl;
    if(eval_ms)
        *eval_ms = te;
    if(count_split)
        *count_split = cs;
    API_END
}

int mpg_walk_events_collect(mpg_engine *eng, double *total_ms, int *count)
{
    return mpg_walk_events_collect2(eng, total_ms, count, nullptr, nullptr, nullptr);
}

const int *mpg_dev_tree_order(mpg_engine *eng) { return (eng && eng->tree_allocated) ? (const int *)eng->tree.idx_b.p : nullptr; }

/* tuning knob used by bench/tests: minimum number of walking lanes that keeps the node phase going */
int mpg_set_walk_split_mode(mpg_engine *eng, int overlap, int chunks_per_wave)
{
    API_BEGIN
    MPG_CHECK(eng && chunks_per_wave >= 0 && chunks_per_wave <= 1024, "bad argument");
    eng->w3.split_overlap = overlap != 0;
    eng->w3.split_chunks_per_wave = chunks_per_wave;
    API_END
}

int mpg_set_walk_offsets64(mpg_engine *eng, int on)
{
    API_BEGIN
    MPG_CHECK(eng, "null engine");
    eng->w3.split_offsets64 = on != 0;
    API_END
}

int mpg_set_walk_threshold(mpg_engine *eng, int thresh)
{
    API_BEGIN
    MPG_CHECK(eng, "null engine");
    eng->walk_thresh = thresh;
    API_END
}

int mpg_set_walk_list_capacity(mpg_engine *eng, int cap)
{
    API_BEGIN
    MPG_CHECK(eng && cap >= 16 && cap <= 65536, "walk list capacity must be in [16, 65536]");
    eng->w3.cap = cap;
    eng->w3.split_cap = (cap + 7) / 8 * 8;
    API_END
}

int mpg_get_walk_choice(mpg_engine *eng, int *variant, int *list_capacity, unsigned *last_overflow)
{
    API_BEGIN
    MPG_CHECK(eng && variant, "null argument");
    *variant = eng->walk_variant ? eng->walk_variant : eng->walk_choice;
    if(list_capacity)
        *list_capacity = eng->w3.split_cap;
    if(last_overflow)
        *last_overflow = eng->w3.split_last_overflow;
    API_END
}

int mpg_set_walk_variant(mpg_engine *eng, int variant)
{
    API_BEGIN
    MPG_CHECK(eng && (variant == 0 || variant == 1 || variant == 4 || variant == 6), "walk variant must be 0 (auto), 1, 4 or 6");
    eng->walk_variant = variant;
    eng->walk_choice = 0;
    API_END
}

} // extern "C"
