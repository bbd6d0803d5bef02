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
    unsigned long long *counters = nullptr; // [8] pp interactions, nodes visited, nodes used, burst statistics (COUNT builds only)
    int list_prio = 0, eval_prio = 0;  // split walk: s_setprio of the two kernels' waves (experiment knobs MPG_LIST_PRIO / MPG_EVAL_PRIO)
    float *cost = nullptr;             // caller order [n] or null: per-target work of this walk (split walk: 8 per leaf entry + 1 per node
                                       // used + 8 per traversal step), the load measure of the domain decomposition (domain.c:611)
};

void launch_grav_walk(const TreeView &tv, const GravParams &gp, const WalkIO &io, bool want_pot, bool count, bool fastwrap, int thresh,
                      hipStream_t st);

// group-cooperative list-form walk (grav_walk_coop.hip): the default.  Persistent kernel; every group of 8 lanes owns
// one target and keeps its interaction lists in a per-wave scratch area.
struct WalkScratch {
    DevBuf<int2> list;      // [resident waves][8 groups][cap]
    DevBuf<unsigned> ctr;   // [16] scratch words; [8] = device error flag (runaway-loop guard)
    int cap = 512;          // list entries per target before the group drains its lists
    int num_cu = 0;
    // two-kernel walk (grav_walk_split.hip)
    DevBuf<unsigned> split_lists; // [slice/8][cap*8] interleaved per-target lists
    DevBuf<int2> split_counts;    // [slice] {leaf entries | wrapped << 30, node entries} or {-1, 0}: overflowed
    DevBuf<int> split_ovf;        // caller indices of overflowed targets
    int split_cap = 512;          // list entries per target (multiple of 8)
    int split_slice = 1 << 25;    // most targets per list-construction / evaluation kernel pair.  Measured at 256^3 Zel'dovich
                                  // (profiles/r02b_walk_knobs.txt): slices of 2^21 / 2^22 / 2^23 targets with the two kernels overlapped on
                                  // two streams 113.8 / 110.6 / 105.9 ms per step, the same slices one after the other 104.6 (2^22) /
                                  // 102.3 (2^23), ONE slice 101.5: every kernel ends with a tail in which its slowest waves hold the
                                  // chip, and two kernels sharing the chip slow each other by more than their mix gains
    size_t split_bytes = 80ull << 30; // list area (bytes) that bounds the slice: slice * cap * 4 <= split_bytes (288 GB of HBM)
    unsigned split_last_overflow = 0, split_last_maxlen = 0;
    bool split_overlap = false;       // true: build the lists of slice k+1 on a second stream while slice k is evaluated (two list areas);
                                      // the default of round 1, measured slower than serial slices in round 2 (see split_slice)
    bool split_offsets64 = false;     // take the 64-bit-offset variants of the kernels whatever the array sizes (tests: the forms a 512^3 tree needs)
    int split_chunks_per_wave = 1;    // 0: persistent grids; > 0: chunks of 8 targets per wave.  Blocks that are dispatched in tree order keep the
                                      // resident waves on neighbouring targets (256^3, walk ms at 1 / 2 / 4 / 8 / 32 / 128 chunks and persistent:
                                      // 62.3 / 62.7; 58.8 / 59.6 / 60.8 / 64.5 / 69.6 / 74.2 on two boxes)
    hipEvent_t ev_mid = nullptr;     // (not owned) recorded between the list kernel and the evaluation kernel of a one-slice walk (bench: the two kernels' times)
    bool mid_recorded = false;
    hipEvent_t ev_before_eval = nullptr; // (not owned) recorded when the tree's leaf blocks are complete: the evaluation kernel waits for it, the list kernel does not need them
    hipStream_t split_stream = nullptr;
    hipEvent_t ev_lists[2] = {nullptr, nullptr}, ev_eval[2] = {nullptr, nullptr}, ev_begin = nullptr;
    ~WalkScratch()
    {
        for(hipEvent_t e : {ev_lists[0], ev_lists[1], ev_eval[0], ev_eval[1], ev_begin})
            if(e)
                (void)hipEventDestroy(e);
        if(split_stream)
            (void)hipStreamDestroy(split_stream);
    }
};
// fastwrap: the minimum-image wrap may be hoisted out of the pair loop (decided by the caller from Rcut, Box, leaf sizes)
void launch_grav_walk_coop(const TreeView &tv, const GravParams &gp, const WalkIO &io, bool want_pot, bool count, bool fastwrap,
                           WalkScratch &ws, hipStream_t st);
// two-kernel walk (grav_walk_split.hip): list construction, then evaluation; overflowing targets fall back to launch_grav_walk
void launch_grav_walk_split(const TreeView &tv, const GravParams &gp, const WalkIO &io, bool want_pot, bool count, bool fastwrap, int thresh,
                            WalkScratch &ws, hipStream_t st);
// grav_short_pair (grav_pair_walk.hip): exact pair-wise short-range force within the sphere of radius rcut_abs
void launch_grav_short_pair(const TreeView &tv, const GravParams &gp, const WalkIO &io, double rcut_abs, bool want_pot, unsigned *d_err,
                            hipStream_t st);
// returns the device error flag of the last cooperative walk (0 = ok); synchronises the stream
unsigned walk_coop_error(WalkScratch &ws, hipStream_t st);

} // namespace mpg
