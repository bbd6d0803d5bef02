// engine_internal.h -- the engine object behind the C-ABI handle (include/mpgadget_hip.h), shared by engine.hip and dist.hip
#pragma once
#include "../../include/mpgadget_hip.h"
#include "grav_walk.h"
#include "mpg_common.h"
#include "pm.h"
#include "sph.h"
#include "timestep.h"
#include "peano.h"
#include "domain.h"
#include "fof.h"
#include "snapshot_io.h"
#include "tree_build.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

using namespace mpg;

std::string &mpg_err_slot(); // thread-local text of the last error (mpg_last_error)

// ---- host-side staging helpers of the AoS (host pointer) path -------------------------------------------------------------
// Pinned, growable host buffer: transfers from / to pageable std::vector memory run at a fraction of the PCIe rate.
template <typename T> struct HostBuf {
    T *p = nullptr;
    size_t cap = 0;
    void reserve(size_t n)
    {
        if(n <= cap)
            return;
        release();
        const size_t want = n + n / 16 + 64;
        MPG_HIP(hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault));
        cap = want;
    }
    void release()
    {
        if(p)
            (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    ~HostBuf() { release(); }
    HostBuf() = default;
    HostBuf(const HostBuf &) = delete;
    HostBuf &operator=(const HostBuf &) = delete;
};

// f(lo, hi) over [0, n) on up to 32 host threads: packing 160-byte records into arrays (and back) is memory-bound and one
// thread moves ~2 GB/s of them; the reference's callers have the cores of the rank idle while the GPU works anyway.
// The threads are persistent (round 5): a pass over the table is cut into 8 chunks that overlap the PCIe transfers, i.e. 8 calls, and
// creating 32 threads per call cost 0.3 - 0.5 ms of each (three passes per step on the critical path of the host forms).  A second
// caller that finds the pool busy (the write-back thread of mpg_gravpm_force beside the main thread) starts its own threads as before.
class HostPool {
    std::vector<std::thread> th;
    std::mutex m, busy;
    std::condition_variable cv_work, cv_done;
    const std::function<void(int64_t, int64_t)> *job = nullptr;
    int64_t n = 0, chunk = 0;
    unsigned gen = 0, pending = 0;
    bool stop = false;
    void worker(unsigned t)
    {
        unsigned seen = 0;
        for(;;) {
            const std::function<void(int64_t, int64_t)> *f;
            int64_t lo, hi;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_work.wait(lk, [&] { return stop || gen != seen; });
                if(stop)
                    return;
                seen = gen;
                f = job;
                lo = (int64_t)t * chunk;
                hi = lo + chunk < n ? lo + chunk : n;
            }
            if(lo < hi)
                (*f)(lo, hi);
            {
                std::lock_guard<std::mutex> lk(m);
                if(--pending == 0)
                    cv_done.notify_all();
            }
        }
    }

  public:
    explicit HostPool(unsigned T)
    {
        for(unsigned t = 0; t < T; t++)
            th.emplace_back([this, t] { worker(t); });
    }
    ~HostPool()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv_work.notify_all();
        for(auto &x : th)
            x.join();
    }
    unsigned size() const { return (unsigned)th.size(); }
    // false: the pool is in use by another caller
    bool run(int64_t count, const std::function<void(int64_t, int64_t)> &f)
    {
        std::unique_lock<std::mutex> one(busy, std::try_to_lock);
        if(!one.owns_lock())
            return false;
        std::unique_lock<std::mutex> lk(m);
        job = &f;
        n = count;
        chunk = (count + size() - 1) / size();
        pending = size();
        gen++;
        cv_work.notify_all();
        cv_done.wait(lk, [&] { return pending == 0; });
        return true;
    }
};

inline HostPool &host_pool(unsigned T)
{
    static HostPool pool(T); // ONE pool per process (not one per instantiation of parallel_for); lives until the process ends
    return pool;
}

template <class F> inline void parallel_for(int64_t n, F f)
{
    static const unsigned cap = getenv("MPG_HOST_THREADS") ? (unsigned)atoi(getenv("MPG_HOST_THREADS")) : 32u;
    unsigned T = std::thread::hardware_concurrency();
    if(T > cap)
        T = cap;
    if(T < 2 || n < 131072) {
        f((int64_t)0, n);
        return;
    }
    static const bool use_pool = getenv("MPG_HOST_NO_POOL") == nullptr;
    if(use_pool) {
        const std::function<void(int64_t, int64_t)> fn = [&f](int64_t lo, int64_t hi) { f(lo, hi); };
        if(host_pool(T).run(n, fn))
            return;
    }
    const int64_t chunk = (n + T - 1) / T;
    std::vector<std::thread> th;
    th.reserve(T);
    for(unsigned t = 0; t < T; t++) {
        const int64_t lo = (int64_t)t * chunk, hi = lo + chunk < n ? lo + chunk : n;
        if(lo >= hi)
            break;
        th.emplace_back([=] { f(lo, hi); });
    }
    for(auto &x : th)
        x.join();
}

struct mpg_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // the tree build of a step has no data dependence on the step's PM force (both read the bound positions): when a PM
    // force has just been queued, force_tree_build runs on this second stream next to it (engine-internal; MPG_NO_TREE_OVERLAP=1
    // keeps everything on one stream)
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_inputs = nullptr, ev_tree_done = nullptr, ev_pad_done = nullptr;
    bool pad_pending = false; // the leaf blocks of the current tree were queued on aux_stream behind ev_tree_done (ev_pad_done)
    bool pm_queued = false;
    // module state (static variables of gravshort-tree.c:30-32, gravity.c:20, forcetree.c:30-37)
    mpg_gravshort_tree_params treepar{0.002, 0.175, 0.9, 2, 6.0, 1.0 / 30.};
    double GravitySoftening = 0;
    double TreeAllocFactor = 0.9;
    bool have_tab = false;
    double tab_dx = 0.02935420743639786;
    DevBuf<float> tab_force, tab_pot;
    // subsystems
    PMesh pm;
    TreeBuilder tree;
    bool tree_allocated = false;
    bool full_particle_tree = false;
    int tree_mask = 63;
    EventTimer timer;
    bool count = false;
    int walk_thresh = 16;
    // 1: lane-per-target while-while kernel (grav_walk.hip); 4: group-cooperative list kernel (grav_walk_coop.hip); 6: two-kernel walk
    // (grav_walk_split.hip); 0: 6 for large target sets, 1 for small ones
    int walk_variant = 0;
    int walk_choice = 0; // the kernel the default policy used last (0: no walk yet)
    int walks_since_tune = 0;
    WalkScratch w3;
    DevBuf<unsigned long long> counters;
    int64_t last_targets = 0;
    float *d_walk_cost = nullptr; // per-target work of the walks, caller order (mpg_dev_set_walk_cost; null: not recorded)
    // un-synchronised HIP event pairs around every walk launch (collected by mpg_walk_events_collect)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> walk_events, free_events;
    // ... and, per walk, the event between the two kernels of a one-slice two-kernel walk (null otherwise)
    std::vector<hipEvent_t> walk_mid, free_mid;
    // SPH module state (static variables of density.c:20, hydra.c:26-34)
    mpg_density_params denspar{1.0, 2.0, 2.0, 99999., 2 /* quintic */, 0.006};
    mpg_hydro_params hydropar{1, 100.0, 0.75};
    SphEngine sph;
    // bound device particles (caller order)
    int64_t n = 0;
    const double *d_pos = nullptr;
    const float *d_mass = nullptr;
    const uint8_t *d_type = nullptr;
    double box = 0;
    // staging for the host SPH path: one device buffer per mpg_sph_arrays field
    DevBuf<double> h_sph[19];
    DevBuf<uint8_t> h_sph_u8[2];
    // staging for the host (AoS) path
    DevBuf<double> s_pos, s_accel, s_gravpm, s_pot, s_prev, s_old;
    DevBuf<double> w_old; // OldAcc of a walk that writes over its own opening input (mpg_dev_grav_short_tree)
    DevBuf<float> s_mass;
    DevBuf<uint8_t> s_type, s_live;
    const uint8_t *pm_live = nullptr; // host path: 0 for garbage / swallowed particles when the staged table holds any (else null)
    DevBuf<int> s_active;
    DevBuf<unsigned> ts_flag;
    DevBuf<uint8_t> tree_incl; // particles included in an active-particle tree
    // hierarchical gravity (timestep.c:239-599): active sublists (ping-pong), the per-level acceleration array, scratch
    DevBuf<int> hier_list[2], hier_val;
    DevBuf<uint8_t> hier_keep;
    DevBuf<double> hier_accel, hier_sp;
    DevBuf<unsigned long long> hier_cnt;
    DevBuf<char> hier_tmp;
    PeanoScratch peano;
    DomainScratch domain;
    FofEngine fof;
    HostBuf<double> h_d, h_d2, h_d3; // pinned staging: positions / 3-vectors, scalars
    HostBuf<float> h_f;
    HostBuf<uint8_t> h_b;
    // host path: what is staged (mpg_set_particle_epoch) and the events of the chunked downloads
    int64_t host_epoch = 0, staged_epoch = 0, staged_n = -1;
    const void *staged_base = nullptr;
    // device-resident drop-in mode (mpg_resident_begin): the table at res_base lives in s_pos / s_mass / s_type and r_*; the host calls on
    // that table move no particle data
    bool resident = false, res_has_vel = false;
    const void *res_base = nullptr;
    int64_t res_n = -1;
    DevBuf<double> r_vel, r_accel, r_gravpm, r_pot;
    // ... and a gas run's SPH arrays (mpg_resident_sph_begin): the host set they were taken from, their device copies (h_sph / h_sph_u8,
    // with vel / gacc / gpm aliasing r_vel / r_accel / r_gravpm) and the garbage flags of the resident table for the integrator kernels
    bool sph_resident = false;
    mpg_sph_arrays res_sph_host{}, res_sph_dev{};
    DevBuf<uint8_t> r_flags;
    double staged_box = 0;
    hipEvent_t chunk_ev[8] = {};
    // Host path with overlap (mpg_set_host_overlap): within one particle-table epoch Pos / Mass / Type, Potential AND the
    // previous FullTreeGravAccel go up in ONE pass over the records (the walk then takes OldAcc on the device from the uploaded
    // acceleration and the device's GravPM: no second host pass), and the results of gravpm_force travel down and into P[] on a copy
    // stream and a host thread while the tree build and the walk run.  host_join() waits for that thread.
    bool host_overlap = false;
    int host_slices = 0;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_pm_done = nullptr, ev_acc_up = nullptr, gchunk_ev[8] = {};
    std::thread unpack_thread;
    std::string unpack_error;
    HostBuf<double> h_acc, h_gpm, h_gpot;
    DevBuf<double> s_prevacc, s_pot2, s_acc_t, s_pot_t; // ... the walk's results compacted in tree order, slice by slice
    HostBuf<double> h_acc_t, h_pot_t;
    HostBuf<int> h_order;
    hipEvent_t slice_ev[9] = {};
    int64_t staged_extra_epoch = -1; // the epoch whose Potential / FullTreeGravAccel are staged (s_pot, s_prevacc)
    int64_t gravpm_epoch = -1;       // the epoch whose GravPM sits in s_gravpm
    // mpg_host_prefetch (round 6): the epoch's packing pass + uploads on a host thread of their own, started by the caller as soon as P[] is
    // final for the step (the end of drift_all_particles) and joined by the first entry point that needs the staged columns
    std::thread prefetch_thread;
    std::string prefetch_error;
    mpg_particle_view prefetch_view{};
    void prefetch_join()
    {
        if(prefetch_thread.joinable())
            prefetch_thread.join();
    }
    void host_join()
    {
        if(unpack_thread.joinable())
            unpack_thread.join();
        prefetch_join();
    }
};

extern "C" void engine_tree_build_on(mpg_engine *eng, int mask, hipStream_t st); // engine.hip

#define API_BEGIN try {
#define API_END_NORETURN             \
    }                                \
    catch(const std::exception &e) { \
        mpg_err_slot() = e.what();   \
        return 1;                    \
    }
#define API_END                      \
    }                                \
    catch(const std::exception &e) { \
        mpg_err_slot() = e.what();   \
        return 1;                    \
    }                                \
    mpg_err_slot().clear();          \
    return 0;

