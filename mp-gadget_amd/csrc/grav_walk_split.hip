// grav_walk_split.hip -- short-range gravity walk as two kernels: list construction, then evaluation (variant 6).
//
// Per-target semantics are the reference's (force_treeev_shortrange, gravshort-tree.c:253-379): every node a target
// visits is discarded, used unopened or opened by exactly the reference's tests for THAT target.  The reference itself
// separates the two activities -- it collects the particles of opened leaves in `ngblist` and evaluates them afterwards
// (gravshort-tree.c:346-374) -- and so does this file, at kernel granularity:
//
//   k_walk_lists8  one traversal per wave of 8 targets (tree-order neighbours): the 64 lanes hold 64 different pending nodes of the
//                  union of the 8 walks, the wave loops over its targets and every lane applies the reference's two tests to its
//                  node for that target.  Written per target to HBM: the opened leaves (4-byte entries: first particle << 3 |
//                  count-1) and the nodes used unopened (4-byte level-order indices).  No force arithmetic, no window tables.
//   k_walk_eval    8 lanes per target stream the target's lists: for a leaf entry lane s evaluates source s (one
//                  coalesced 256-byte read per group), node entries are taken 8 at a time.  No traversal state: the
//                  kernel is a pure fp64 pair loop fed by sequential list reads.
//
// List layout: target t of chunk u (the 8 targets of a wave) owns lists[(u * 8 + t) * cap ...]: leaf entries from 0 up, node
// entries from cap - 1 down; the 64 lanes of an append write one contiguous run.  A target whose lists would exceed `cap` is put on
// an overflow list and handled afterwards by the cooperative kernel (grav_walk_coop.hip), so `cap` bounds memory, not
// correctness.  Targets are processed in slices of `slice` targets so the list area stays bounded (cap * 4 B each).
//
// Periodic wrap: with FASTWRAP (box large against Rcut and the leaves) a source range shares the periodic image of its
// node; k_walk_lists8 records per wave whether ANY of its entries lies on a wrapped image.  Waves without (all but a
// surface layer Rcut thick) evaluate with plain differences, which is bit-identical to NEAREST() there; the others take
// NEAREST() per pair as partmanager.h:99 does.
// (Rounds 1-2 built the lists with 8 lanes per target - k_walk_lists, k_walk_lists2: DESIGN.md 3.2 - retired in round 3.)
#include "grav_walk.h"
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <type_traits>

namespace mpg {

namespace {

#ifndef MPG_EVAL_BLOCKS
// resident 256-thread blocks per CU the evaluation kernel is compiled for.  4 (128 VGPRs) since round 4: the instantiation with the
// potential spills 8 registers there (50 at 6 blocks / 80 VGPRs: 15 GB of scratch writes per walk at 256^3) and the walk is 1.5 ms faster
// (69.2 -> 67.7 ms); round 3 had measured 4, 5 and 6 blocks as equal before the list kernel changed
#define MPG_EVAL_BLOCKS 4
#endif
#ifndef MPG_LISTS_PREFETCH_MIN
#define MPG_LISTS_PREFETCH_MIN 64
#endif
#ifndef MPG_LIST_BLOCKS
#define MPG_LIST_BLOCKS 6 // resident 256-thread blocks per CU the list kernel is compiled for
#endif
#ifndef MPG_EVAL_BLOCKS_LONG
// ... and where the lists are long (capacity >= 4096: a clustered set).  6 (80 registers) until round 5, when "it waits for memory and the
// waves count" held; with every source record requested two evaluations ahead the 4-block build wins there too (256^3 clustered set:
// 125.0 -> 118.7 ms per step; 5 blocks 121.9)
#define MPG_EVAL_BLOCKS_LONG 4
#endif


#ifdef MPG_LEAF_HIST
__device__ unsigned long long g_leaf_hist[16];
#endif

__device__ __forceinline__ double rsqrt_nr(double x)
{
    // v_rsq_f64 + one cubic Newton step -> full double precision; x > 0
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * y), y, 1.0);
    return fma(y * e, fma(e, 0.375, 0.5), y);
}

// Softened branch of apply_accn_to_output, gravshort-tree.c:168-185 (Gadget-2 spline, constants as truncated there).
// Rare (close encounters; the self interaction takes the literals in pair_force).  Its 13 constants must not occupy registers in the
// pair loop, so they are produced INSIDE the branch.  Rounds 1-4 fetched them from constant memory through a laundered pointer, which
// hipcc turned into flat_load + s_waitcnt vmcnt(0): two (inner spline branch) to five (outer) serialised memory round trips per
// softened pair step, each wait also draining the prefetched source records - a softened step cost ~10 ordinary ones, and a clustered
// set has them.  Now every constant is two s_mov_b32 of literals into a scalar register pair, emitted by a volatile asm at the point
// of use (volatile: not hoisted out of the branch): no memory access, no wait.  (MPG_SPLINE_CONSTMEM restores the loads.)
// The divisions are v_rcp_f64 + Newton steps (<= 1 ulp from the quotient) instead of the 40-instruction IEEE expansion.
#ifdef MPG_SPLINE_CONSTMEM
__constant__ double SPLINE_C[13] = {10.666666666667, 32.0, 38.4,  -2.8, 5.333333333333, 6.4, 9.6,
                                    21.333333333333, 48.0, 0.066666666667, -3.2, -16.0, 2.133333333333};
#endif

template <unsigned long long BITS>
__device__ __forceinline__ double scalar_literal()
{
    unsigned lo, hi;
    asm volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=s"(lo), "=s"(hi) : "i"((unsigned)(BITS & 0xffffffffull)), "i"((unsigned)(BITS >> 32)));
    return __hiloint2double((int)hi, (int)lo);
}
#define MPG_K(x) scalar_literal<__builtin_bit_cast(unsigned long long, (double)(x))>()

__device__ __forceinline__ double rcp_nr(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    return fma(fma(-x, y, 1.0), y, y);
}

__device__ __forceinline__ void softened_pair(const double r, const double m, const double hinv, const double h3inv, double &fac, double &facpot)
{
    const double u = r * hinv;
    double wpk;
#ifdef MPG_SPLINE_CONSTMEM
    const double *c = SPLINE_C;
    asm volatile("" : "+s"(c));
    if(u < 0.5) {
        fac = m * h3inv * (c[0] + u * u * (c[1] * u - c[2]));
        wpk = c[3] + u * u * (c[4] + u * u * (c[5] * u - c[6]));
    }
    else {
        const double iu = rcp_nr(u);
        fac = m * h3inv * (c[7] - c[8] * u + c[2] * u * u - c[0] * u * u * u - c[9] * (iu * iu * iu));
        wpk = c[10] + c[9] * iu + u * u * (c[0] + u * (c[11] + u * (c[6] - c[12] * u)));
    }
#else
    if(u < 0.5) {
        fac = m * h3inv * (MPG_K(10.666666666667) + u * u * (MPG_K(32.0) * u - MPG_K(38.4)));
        wpk = MPG_K(-2.8) + u * u * (MPG_K(5.333333333333) + u * u * (MPG_K(6.4) * u - MPG_K(9.6)));
    }
    else {
        const double iu = rcp_nr(u);
        fac = m * h3inv * (MPG_K(21.333333333333) - MPG_K(48.0) * u + MPG_K(38.4) * u * u - MPG_K(10.666666666667) * u * u * u -
                           MPG_K(0.066666666667) * (iu * iu * iu));
        wpk = MPG_K(-3.2) + MPG_K(0.066666666667) * iu +
              u * u * (MPG_K(10.666666666667) + u * (MPG_K(-16.0) + u * (MPG_K(9.6) - MPG_K(2.133333333333) * u)));
    }
#endif
    facpot = m * hinv * wpk;
}

// apply_accn_to_output, gravshort-tree.c:158-193, in three stages so that the evaluation loop can run TWO pairs side by side (their chains
// of dependent fp64 operations are what the kernel waits for: with 2 / 3 / 4 resident waves per SIMD it takes 52.8 / 41.3 / 36.1 ms at 256^3,
// i.e. 18 ms of issue + 69 ms / waves of exposed latency): the common arithmetic up to the softening test, the rare softened branch (taken
// once for both pairs), the window table and the sums.
struct PairTmp {
    double dx, dy, dz, r2, rinv, r, fac, facpot, m;
};

__device__ __forceinline__ void pair_pre(const Src4 s, const double dx, const double dy, const double dz, PairTmp &t)
{
    t.dx = dx;
    t.dy = dy;
    t.dz = dz;
    t.m = s.m;
    t.r2 = dx * dx + dy * dy + dz * dz;
    // (no clamp of r2 on the common path: for r2 = 0 - the self interaction - and below ~1e-300 rinv is not finite, and the softened
    // branch, which such a pair always takes and which overwrites fac and facpot, puts r = 0 in its place)
    t.rinv = rsqrt_nr(t.r2);
    t.r = t.r2 * t.rinv;
    const double mr = s.m * t.rinv;
    t.fac = mr * t.rinv * t.rinv;
    t.facpot = -mr;
}

// rare (the self interaction, close encounters): kept out of line so that its divisions and constants do not occupy registers in the pair loop
__device__ __forceinline__ void pair_soft(const GravParams &gp, PairTmp &t)
{
#ifndef MPG_NO_SELF_FAST
    if(t.r2 == 0.0) { // the self interaction, once per target and so in ~6 % of a wave's pair steps: the spline's inner branch at u = 0
        t.r = 0.0;    // (bit for bit: c0 + 0 and c3 + 0), from literals - softened_pair's constant fetches are dependent vector loads whose
        t.fac = t.m * gp.h3inv * 10.666666666667; // waits also drain the prefetched source records
        t.facpot = t.m * gp.hinv * -2.8;
    }
    else
#endif
    {
        if(!(t.rinv < 1e150))
            t.r = 0.0;
        softened_pair(t.r, t.m, gp.hinv, gp.h3inv, t.fac, t.facpot);
    }
}

template <bool POT>
__device__ __forceinline__ void pair_post(const PairTmp &t, const GravParams &gp, const double *__restrict__ wtab, double &ax, double &ay, double &az,
                                          double &pot)
{
    // r / cellsize / dx, gravity.c:57-58.  tabindex >= NTAB-1 contributes nothing (gravity.c:60-61): the clamp lands on
    // the table's last row, which holds zeros
    const double ti = t.r * gp.inv_cell_dx;
    // row t holds {T[t], T[t+1] - T[t]} (the difference of two floats is exact in double): T[t] + (i - t) (T[t+1] - T[t]) is the
    // interpolation of gravity.c:63 up to one rounding.  (i - t) = fract(i) exactly for i >= 0.
    const int ix = min((int)ti, NTAB - 1);
    const double w1 = __builtin_amdgcn_fract(ti);
    // two tables of 16-byte rows, {T, dT} of the force, then (with POT) {T, dT} of the potential NTAB rows further on: both reads are
    // issued together.  (One 32-byte row holding both put every force read on one half of the LDS banks and every potential read on
    // the other half; with 16-byte rows the random rows of a wave's 64 lanes spread over all banks.  Round 3 also measured rows of four
    // floats {F[t], F[t+1], P[t], P[t+1]} - one 16-byte read per pair, six more conversions: no change.  Round 5: rows {T[t] - t dT, dT}
    // evaluated without the fraction - one instruction less, no change.)
    const double *__restrict__ row = wtab + ix * 2;
    const double2 f = *(const double2 *)row;
    double2 p = f;
    if(POT)
        p = *(const double2 *)(row + 2 * NTAB);
    const double fac = t.fac * fma(w1, f.y, f.x);
    ax = fma(t.dx, fac, ax);
    ay = fma(t.dy, fac, ay);
    az = fma(t.dz, fac, az);
    if(POT)
        pot = fma(t.facpot, fma(w1, p.y, p.x), pot);
}

template <bool POT>
__device__ __forceinline__ void pair_force(const Src4 s, const double dx, const double dy, const double dz, const GravParams &gp,
                                           const double *__restrict__ wtab, double &ax, double &ay,
                                           double &az, double &pot)
{
    PairTmp t;
    pair_pre(s, dx, dy, dz, t);
    if(t.r2 < gp.h2)
        pair_soft(gp, t);
    pair_post<POT>(t, gp, wtab, ax, ay, az, pot);
}

// element i of a device array.  O32: the byte offset fits 32 bits (decided on the host), which lets the load use the
// scalar-base + 32-bit-offset addressing mode instead of 64-bit vector address arithmetic
template <bool O32, typename T>
__device__ __forceinline__ T ld(const T *__restrict__ base, const unsigned i)
{
    if(O32)
        return *(const T *)((const char *)base + (size_t)(unsigned)(i * (unsigned)sizeof(T)));
    return base[i];
}

__device__ __forceinline__ void st32(unsigned *__restrict__ base, const unsigned i, const unsigned v)
{
    *(unsigned *)((char *)base + (size_t)(unsigned)(i * 4u)) = v; // offsets inside one chunk's list area: always < 2^32 bytes
}

__device__ __forceinline__ bool any_lane(const bool b) { return __builtin_amdgcn_ballot_w64(b) != 0; }

// wave priority for instruction arbitration between the two kernels when they share the CUs (s_setprio takes an immediate)
__device__ __forceinline__ void set_wave_prio(const int p)
{
    if(p == 1)
        __builtin_amdgcn_s_setprio(1);
    else if(p == 2)
        __builtin_amdgcn_s_setprio(2);
    else if(p >= 3)
        __builtin_amdgcn_s_setprio(3);
}

__device__ __forceinline__ double nearest_img(double d, double box, double invbox) { return fma(-rint(d * invbox), box, d); }

// chunks of 8 targets (one per 8-lane group) of the slice; XCD x (= blockIdx % 8, where the hardware places this block) owns
// a contiguous part of the tree-ordered targets and its waves take chunks round-robin, so that each private L2 serves one
// region of the tree and concurrently running waves work on neighbouring targets
struct ChunkIter {
    unsigned lo, hi, first, stride;
    __device__ ChunkIter(unsigned nchunks)
    {
        const unsigned xcd = blockIdx.x & 7;
        const unsigned waves_per_block = blockDim.x >> 6;
        first = (blockIdx.x >> 3) * waves_per_block + (threadIdx.x >> 6);
        stride = (gridDim.x >> 3) * waves_per_block;
        lo = (unsigned)(((uint64_t)nchunks * xcd) >> 3);
        hi = (unsigned)(((uint64_t)nchunks * (xcd + 1)) >> 3);
    }
};

// ctl words: [0] number of overflowed targets, [1] error flag (loop guard / frontier), [2] longest list seen
// counters (COUNT builds): [0] pair interactions [1] nodes visited [2] nodes used unopened [3] frontier pops [4] nodes popped
//                          [5] leaf entries written [6] node entries written
//
// MODE of the node tests: 0 NEAREST() per quantity (small boxes); 1 the node's periodic image k = rint((c - p)/Box) is applied to the
// target (FASTWRAP); 2 plain differences, for targets farther than Rcut + Box/500 from every face of the box.  MODE 2 is exact: a
// node that is not discarded has |c - p| <= Rcut + len/2 per axis on its nearest image (its centre of mass lies inside it), which for
// such a target is the unwrapped image; and a node that is discarded with nearest-image distances is discarded with the (larger or
// equal) unwrapped ones.  In modes 1 and 2 the centre of mass of a node can sit on another image than its centre only if
// Rcut + len >= Box/2 (the root and its children): those (`special`) take NEAREST() for both, exactly as gravshort-tree.c:299-300 does.
//
// The same tests with every comparison taken as a lane mask (k_walk_lists8).  The Barnes-Hut switch is folded into aold by the caller
// (aold = +inf: "mass l^2 > r^4 aold" is false for every r, NaN at r = 0 included).
template <int MODE>
__device__ __forceinline__ void node_test_masks(const GravParams &gp, const NodeGeo &g, const Src4 &mom, const bool special,
                                                const bool any_special /* wave-uniform */, const double eff, const double l2, const double inside,
                                                const double ml2, const double px, const double py, const double pz, const double aold,
                                                unsigned long long &m_discard, unsigned long long &m_open, unsigned long long &m_wrap)
{
    double dx, dy, dz, cmax;
    m_wrap = 0ull;
    if(MODE == 0) {
        cmax = fmax(fmax(fabs(nearest_img(g.cx - px, gp.box, gp.invbox)), fabs(nearest_img(g.cy - py, gp.box, gp.invbox))),
                    fabs(nearest_img(g.cz - pz, gp.box, gp.invbox)));
        dx = nearest_img(mom.x - px, gp.box, gp.invbox);
        dy = nearest_img(mom.y - py, gp.box, gp.invbox);
        dz = nearest_img(mom.z - pz, gp.box, gp.invbox);
    }
    else {
        if(MODE == 1) {
            const double kx = rint((g.cx - px) * gp.invbox);
            const double ky = rint((g.cy - py) * gp.invbox);
            const double kz = rint((g.cz - pz) * gp.invbox);
            const double qx = fma(kx, gp.box, px), qy = fma(ky, gp.box, py), qz = fma(kz, gp.box, pz);
            cmax = fmax(fmax(fabs(g.cx - qx), fabs(g.cy - qy)), fabs(g.cz - qz));
            dx = mom.x - qx;
            dy = mom.y - qy;
            dz = mom.z - qz;
            m_wrap = __builtin_amdgcn_ballot_w64(kx != 0.0) | __builtin_amdgcn_ballot_w64(ky != 0.0) | __builtin_amdgcn_ballot_w64(kz != 0.0);
        }
        else {
            cmax = fmax(fmax(fabs(g.cx - px), fabs(g.cy - py)), fabs(g.cz - pz));
            dx = mom.x - px;
            dy = mom.y - py;
            dz = mom.z - pz;
        }
        if(any_special) { // (the root and its children: exact images for the centre and the centre of mass, see above)
            const double jx = rint((mom.x - px) * gp.invbox), jy = rint((mom.y - py) * gp.invbox), jz = rint((mom.z - pz) * gp.invbox);
            dx = special ? fma(-jx, gp.box, mom.x - px) : dx;
            dy = special ? fma(-jy, gp.box, mom.y - py) : dy;
            dz = special ? fma(-jz, gp.box, mom.z - pz) : dz;
            const unsigned long long m_sp = __builtin_amdgcn_ballot_w64(special);
            m_wrap |= m_sp & (__builtin_amdgcn_ballot_w64(jx != 0.0) | __builtin_amdgcn_ballot_w64(jy != 0.0) | __builtin_amdgcn_ballot_w64(jz != 0.0));
            if(MODE == 2) {
                const double cm = fmax(fmax(fabs(nearest_img(g.cx - px, gp.box, gp.invbox)), fabs(nearest_img(g.cy - py, gp.box, gp.invbox))),
                                       fabs(nearest_img(g.cz - pz, gp.box, gp.invbox)));
                cmax = special ? cm : cmax;
            }
        }
    }
    const double r2 = dx * dx + dy * dy + dz * dz;
    // shall_we_discard_node / shall_we_open_node, gravshort-tree.c:198-241
    m_discard = __builtin_amdgcn_ballot_w64(r2 > gp.rcut2) & __builtin_amdgcn_ballot_w64(cmax > eff);
    m_open = __builtin_amdgcn_ballot_w64(ml2 > r2 * r2 * aold) | __builtin_amdgcn_ballot_w64(l2 > r2 * gp.bhangle2) | __builtin_amdgcn_ballot_w64(cmax < inside);
}

// ---- list construction with ONE traversal per wave: 8 targets share a frontier, one node per lane ---------------------------------
// The 64 lanes of a wave hold 64 DIFFERENT pending nodes of the union of the walks of the wave's 8 targets (tree-order neighbours:
// leaf-mates mostly), popped from a wave-shared frontier in LDS whose entries carry the mask of the targets that reached the node.
// The wave then loops over its targets: target t's position and opening parameter are wave-uniform, every lane applies the
// reference's two tests to ITS node for target t (gravshort-tree.c:198-241; node_test_masks above), and the outcomes are lane
// masks: the entries of target t are appended by all lanes at once (position = count + v_mbcnt of the mask), the counts advance by
// s_bcnt1, the push mask of a lane collects the targets that open its node.  After the 8 targets the opened internal nodes' children
// go back to the frontier (one entry per child: a prefix sum over the lanes by four ballots).
// Per target the set of nodes tested and the outcome of every test are exactly those of its own walk (a node reaches the frontier
// with bit t set iff target t opened its parent); the ORDER of a target's list entries depends on its 7 wave-mates.
#ifndef MPG_QCAP
#define MPG_QCAP 1024
#endif
constexpr int QCAP = MPG_QCAP; // frontier entries per wave (node index 4 B + target mask 1 B); a wave that would exceed it hands its targets to the fallback

__device__ __forceinline__ unsigned mbcnt64(const unsigned long long b)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
}


// ---- Round 6: the node tests of a target pass pre-classified in fp32 (MODE 2 waves), fp64 only where fp32 cannot decide ------------------
// 19 of the ~30 vector instructions of a target pass are the fp64 arithmetic and compares of the reference's two tests (node_test_masks), and
// tools/valu_rates.hip measures fp64 add / mul / fma at 4.8 - 5.0 cycles per wave instruction against 2.4 for fp32, and a compare into a scalar
// register pair at ~7 (fp64) / ~6 (fp32).  The fast path below takes the SAME decisions from fp32 arithmetic with proven error margins:
//   * coordinates relative to a wave-local origin O (the first valid target): C = fl32(centre - O), S = fl32(cofm - O) once per popped node
//     (amortised over the up to 8 targets that test it), P_t = fl32(p_t - O) once per wave;  D >= |p_t - O| per axis bounds the spread;
//   * every predicate as a NORMALISED slack n = (x - threshold) / E with E an upper bound on twice the absolute error of (x - threshold):
//       n1 = (r2 - rcut2) / E1            E1 = 24u rcut2 + 20u D rcut                       (u = 2^-24;  r2 error <= 7u r2 + 7u D r)
//       n2 = (cmax - eff) / E23           E23 = 12u (len + D + rcut) >= 12u eff + 4u D      (cmax error <= 2u cmax + 2u D)
//       n3 = (cmax - inside) / E23
//       n4 = (r2 / T - 1) / d4            T = max(l2 / theta2, sqrt(m l2 / aold)),  d4 = 64u + 32u D theta / len
//     ("m l2 > r2 r2 aold" <=> r2 < sqrt(m l2) / sqrt(aold): no r2^2 in fp32; aold = +inf - the Barnes-Hut switch - and aold = 0 come out right);
//     a slack with |n| > 1 has the sign of the exact quantity (derivation: DESIGN.md 3.2, round 6);
//   * discard <=> min(n1, n2) > 0, open <=> min(n3, n4) < 0, and the pass is AMBIGUOUS for a lane iff min(|min(n1, n2)|, |min(n3, n4)|) <= 1
//     (a superset of the lanes where some deciding slack is within its margin): three compares instead of five;
//   * a pass with an ambiguous active lane, a node or target outside the range fp32 handles (m l2 or aold beyond 1e-30 .. 1e30) re-reads the node
//     and runs node_test_masks in fp64.  Decisions are therefore exactly the reference's; COUNT builds evaluate both and raise ctl[1] = 3 on any
//     difference, and count the ambiguous passes (counters[12]; [13]: the waves that ran the fp32 form).
struct F32Wave {
    double ox, oy, oz;   // the origin (wave-uniform)
    float w1, c1;        // n1 = fma(r2, w1, c1)
    float dr;            // D + rcut
    float k4;            // 32u D theta
    unsigned tweird;     // targets whose aold fp32 cannot carry: always fp64
    const float *s_tgtf; // LDS: [t][4] = P_t, sqrt(aold_t)
};
constexpr float F32_U = 5.9604644775390625e-8f; // 2^-24

// the 8 targets of a wave: list lengths and counters are wave-uniform (scalar registers); positions and opening parameters sit in
// LDS (s_tgt: [t][4] doubles) and are read back with a wave-uniform address per target - as scalars they overflowed the SGPR file
// (8 x 8 registers) and every use cost a v_readlane
struct WaveTargets {
    int nleaf[8], nnode[8];
    unsigned c_vis[8], c_used[8];
};

// returns false on an internal error (loop guard)
template <bool COUNT, int MODE, bool O32, bool F32 = false>
__device__ __forceinline__ bool walk_wave8(const TreeView &tv, const GravParams &gp, unsigned *__restrict__ Lw, unsigned *__restrict__ q_node,
                                           unsigned char *__restrict__ q_mask, const double *__restrict__ s_tgt, const int cap, const int lane,
                                           const unsigned live0, WaveTargets &T, unsigned &overflowed, bool &wrapped, unsigned (&c_pp)[8],
                                           const unsigned guard_max, unsigned *__restrict__ ctl, unsigned &st_a, unsigned &st_al,
                                           const F32Wave &W = F32Wave(), unsigned *n_amb = nullptr)
{
    static_assert(!F32 || MODE == 2, "the fp32 pre-classification is written for plain differences");
    unsigned live = live0; // targets still walking (wave-uniform)
    int sp = 0;            // frontier entries (wave-uniform)
    unsigned guard = 0;
    unsigned long long wmask = 0;
    int maxused = 0; // longest pair of lists among the wave's targets (wave-uniform)

    // the children of the nodes a pass opened: one frontier entry per child, mask = the targets that opened the parent.  Returns false
    // if the frontier would overflow (a pathological tree): every target still walking goes to the fallback.
    auto push_children = [&](const bool pushing, const NodeLinkB &lk, const unsigned openmask) -> bool {
        const unsigned long long bp = __builtin_amdgcn_ballot_w64(pushing);
        if(bp == 0ull)
            return true;
        const unsigned nm1 = (unsigned)(lk.nchild - 1); // 0..7
        const unsigned long long b0 = __builtin_amdgcn_ballot_w64(pushing && (nm1 & 1u)), b1 = __builtin_amdgcn_ballot_w64(pushing && (nm1 & 2u)),
                                 b2 = __builtin_amdgcn_ballot_w64(pushing && (nm1 & 4u));
        const int total = __builtin_popcountll(bp) + __builtin_popcountll(b0) + 2 * __builtin_popcountll(b1) + 4 * __builtin_popcountll(b2);
        if(sp + total > QCAP) {
            overflowed |= live;
            live = 0;
            return false;
        }
        if(pushing) {
            const unsigned at = (unsigned)sp + mbcnt64(bp) + mbcnt64(b0) + 2u * mbcnt64(b1) + 4u * mbcnt64(b2);
#pragma unroll
            for(unsigned c = 0; c < 8; c++)
                if(c <= nm1) {
                    q_node[at + c] = (unsigned)lk.firstchild + c;
                    q_mask[at + c] = (unsigned char)openmask;
                }
        }
        sp += total;
        return true;
    };

    // ---- the top of the tree: the root, then its children, then - as long as the wave's targets open ONE node of a level - that node's
    // children, with ONE pass per level over (node, target) pairs: lane = 8 * target + node.  (Popped from the frontier like every
    // other node, the root filled 1 lane and its children 8 of a pop that costs 8 target passes whatever it holds: 2 to 3 of the ~21
    // pops of a wave at 256^3.  Written as a branch of the main loop instead - any pop of <= 8 nodes - the same code made hipcc spill
    // 36 vector registers in the passes per target: 79 ms per walk against 70.)  The root and its children are also the only nodes that
    // can need exact periodic images for the centre and the centre of mass separately (`special`: Rcut + len >= Box / 2 needs len >=
    // Box / 2, FASTWRAP has Rcut < 0.2 Box), so the passes of the main loop below do without that branch.
    {
        const int tc = lane & 7, tt = lane >> 3;
        unsigned tmask = live; // targets that reach the nodes of this level (wave-uniform)
        unsigned first = 0u;
        int nch = 1;
        for(int level = 0; tmask != 0u; level++) {
            if(level > 64) { // (a corrupt tree: the loop guard of the main loop, for the chain of single opened nodes)
                if(lane == 0)
                    atomicExch(&ctl[1], 1u);
                return false;
            }
            const bool valid = tc < nch && ((tmask >> tt) & 1u);
            const unsigned my = first + (unsigned)(tc < nch ? tc : 0);
            const NodeGeo g = ld<O32>(tv.geoB, my);
            const Src4 mom = ld<O32>(tv.momB, my);
            const NodeLinkB lk = ld<O32>(tv.linkB, my);
            const double4 tg = *(const double4 *)(s_tgt + 4 * tt);
            const double eff = fma(0.5, g.len, gp.rcut);
            const double l2 = g.len * g.len;
            const bool special = MODE != 0 && valid && level < 2;
            unsigned long long m_discard, m_open, m_wrap;
            node_test_masks<MODE>(gp, g, mom, special, MODE != 0 && level < 2, eff, l2, 0.6 * g.len, mom.m * l2, tg.x, tg.y, tg.z, tg.w, m_discard,
                                  m_open, m_wrap);
            const unsigned long long m_valid = __builtin_amdgcn_ballot_w64(valid);
            const unsigned long long m_leafnode = __builtin_amdgcn_ballot_w64(lk.pcount > 0);
            const unsigned long long m_intnode = ~m_leafnode & __builtin_amdgcn_ballot_w64(lk.nchild > 0);
            const unsigned long long keep = m_valid & ~m_discard;
            const unsigned long long bn = keep & ~m_open, bl = keep & m_open & m_leafnode;
            unsigned long long bpush = keep & m_open & m_intnode;
            if(COUNT) {
                st_a++;
                st_al += (unsigned)nch;
            }
            if(COUNT || (bn | bl) != 0ull) { // (list entries this high in the tree: small trees, or coarse nodes far from the targets)
                const unsigned ent_val = my; // a leaf entry = the leaf's level-order node number (its block of 8 source records: tv.srcL)
#pragma unroll
                for(int t = 0; t < 8; t++) {
                    const unsigned long long grp = 0xffull << (8 * t);
                    const unsigned long long bl_t = bl & grp, bn_t = bn & grp;
                    const int kl = __builtin_popcountll(bl_t), kn = __builtin_popcountll(bn_t);
                    if(T.nleaf[t] + T.nnode[t] + kl + kn > cap) { // the lists of target t are full: the fallback kernel walks it again
                        overflowed |= 1u << t;
                        live &= ~(1u << t);
                        bpush &= ~grp;
                        continue;
                    }
                    unsigned *__restrict__ Lt = Lw + (unsigned)(t * cap);
                    if(__builtin_amdgcn_inverse_ballot_w64(bl_t))
                        st32(Lt, __builtin_amdgcn_mbcnt_hi((unsigned)(bl_t >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bl_t, (unsigned)T.nleaf[t])), ent_val);
                    if(__builtin_amdgcn_inverse_ballot_w64(bn_t))
                        st32(Lt, (unsigned)(cap - 1 - T.nnode[t]) - mbcnt64(bn_t), my);
                    T.nleaf[t] += kl;
                    T.nnode[t] += kn;
                    maxused = max(maxused, T.nleaf[t] + T.nnode[t]);
                    if(COUNT) {
                        T.c_vis[t] += (unsigned)__builtin_popcountll(m_valid & grp);
                        T.c_used[t] += (unsigned)kn;
                        c_pp[t] += __builtin_amdgcn_inverse_ballot_w64(bl_t) ? (unsigned)lk.pcount : 0u;
                    }
                }
            }
            if(MODE != 0)
                wmask |= m_wrap & (bl | bn);
            // the targets that opened node tc (the same in every lane with this tc); lanes 0..7 - target group 0 - hold one lane per node
            unsigned om = 0;
#pragma unroll
            for(int t = 0; t < 8; t++)
                om |= (unsigned)((bpush >> (8 * t + tc)) & 1ull) << t;
            const unsigned long long m_opened = __builtin_amdgcn_ballot_w64(lane < 8 && om != 0u);
            if(__builtin_popcountll(m_opened) == 1) { // one node opened: its children are the next level
                const int src = __builtin_ctzll(m_opened);
                tmask = (unsigned)__builtin_amdgcn_readlane((int)om, src);
                first = (unsigned)__builtin_amdgcn_readlane(lk.firstchild, src);
                nch = __builtin_amdgcn_readlane(lk.nchild, src);
            }
            else {
                push_children(lane < 8 && om != 0u, lk, om);
                tmask = 0u;
            }
        }
    }
#ifdef MPG_LISTS_PREFETCH
    // Round 5 experiment, measured and left off (29.6 ms at 5 resident blocks, 34 at 6 with 37 spilled registers, against 25.3 ms without it;
    // 20 more registers for the second batch cost a resident wave, which is worth what the prefetch hides; 7 blocks of the plain form: 25.7).
    // Two batches of nodes in flight: the batch BELOW the one being tested is taken off the frontier and its 80 bytes per lane
    // requested before the 8 passes over the current batch, which then hide that gather (the kernel is latency-bound: with 2 / 4 / 6 resident
    // waves per SIMD it takes 55.5 / 33.1 / 25.7 ms at 256^3, i.e. 11 ms of issue + 89 ms / waves).  The batch below does not depend on
    // what the current batch pushes, so the set of (node, target) tests is unchanged; the frontier is a batch deeper at most, and the order
    // of a target's list entries changes (it already depended on its wave-mates).  When nothing waits below, the next batch is what the
    // current one pushed, as before.
    struct Batch {
        unsigned my, qmask;
        int n;
        NodeGeo g;
        Src4 mom;
        NodeLinkB lk;
    };
    auto pop = [&](Batch &b) {
        __builtin_amdgcn_wave_barrier();
        b.n = sp < 64 ? sp : 64;
        // (idle lanes read entry 0, which always holds a node index: no exec-mask regions around the two reads)
        const int qi = lane < b.n ? sp - 1 - lane : 0;
        b.my = q_node[qi];
        b.qmask = lane < b.n ? (unsigned)q_mask[qi] : 0u;
        sp -= b.n;
        b.g = ld<O32>(tv.geoB, b.my);
        b.mom = ld<O32>(tv.momB, b.my);
        b.lk = ld<O32>(tv.linkB, b.my);
    };
    Batch cur, nxt;
    pop(cur);
    while(cur.n > 0 && live) {
        if(++guard > guard_max) {
            if(lane == 0)
                atomicExch(&ctl[1], 1u);
            return false;
        }
        nxt.n = 0;
        if(sp >= MPG_LISTS_PREFETCH_MIN) // (a FULL batch waits below: a part of one would be tested with idle lanes, where the pure LIFO tops it up
            pop(nxt);                    // with what the current batch pushes)
        const int n = cur.n;
        const unsigned my = cur.my;
        const unsigned mask = cur.qmask & live;
        const NodeGeo g = cur.g;
        const Src4 mom = cur.mom;
        const NodeLinkB lk = cur.lk;
#else
    while(sp > 0 && live) {
        if(++guard > guard_max) {
            if(lane == 0)
                atomicExch(&ctl[1], 1u);
            return false;
        }
        __builtin_amdgcn_wave_barrier();
        const int n = sp < 64 ? sp : 64;
        const bool valid = lane < n;
        // (idle lanes read entry 0, which always holds a node index: no exec-mask regions around the two reads)
        const int qi = valid ? sp - 1 - lane : 0;
        const unsigned my = q_node[qi];
        const unsigned mask = valid ? ((unsigned)q_mask[qi] & live) : 0u;
        sp -= n;
        const NodeGeo g = ld<O32>(tv.geoB, my);
        const Src4 mom = ld<O32>(tv.momB, my);
        const NodeLinkB lk = ld<O32>(tv.linkB, my);
#endif
        // fp64 operands of the tests (F32: only the fall-back and the COUNT builds' cross-check use them, from a second read of the node)
        double eff = 0, l2 = 0, inside = 0, ml2 = 0;
        // F32: the node relative to the wave's origin and the normalising factors of its slacks (see F32Wave)
        float Cx = 0, Cy = 0, Cz = 0, Sx = 0, Sy = 0, Sz = 0, rE = 0, c2 = 0, c3 = 0, wA = 0, wB = 0, c4 = 0;
        unsigned long long m_nweird = 0ull;
        if constexpr(F32) {
            Cx = (float)(g.cx - W.ox);
            Cy = (float)(g.cy - W.oy);
            Cz = (float)(g.cz - W.oz);
            Sx = (float)(mom.x - W.ox);
            Sy = (float)(mom.y - W.oy);
            Sz = (float)(mom.z - W.oz);
            const float lenf = (float)g.len, mf = (float)mom.m;
            const float l2f = lenf * lenf, ml2f = mf * l2f;
            rE = __builtin_amdgcn_rcpf((12.0f * F32_U) * (lenf + W.dr));
            c2 = -fmaf(0.5f, lenf, (float)gp.rcut) * rE;
            c3 = -(0.6f * lenf) * rE;
            // wA = (theta2 / l2) / d4 with d4 = 64u + 32u D theta / len, i.e. theta2 / (len (64u len + 32u D theta));  1 / d4 = wA l2 / theta2
            wA = (float)gp.bhangle2 * __builtin_amdgcn_rcpf(lenf * fmaf(64.0f * F32_U, lenf, W.k4));
            const float g4 = wA * l2f * (float)(1.0 / gp.bhangle2);
            wB = __builtin_amdgcn_rsqf(ml2f) * g4;
            c4 = -g4;
            m_nweird = __builtin_amdgcn_ballot_w64(!(ml2f > 1e-30f && ml2f < 1e30f && lenf > 1e-12f && lenf < 1e12f));
        }
        else {
            eff = fma(0.5, g.len, gp.rcut);
            l2 = g.len * g.len;
            inside = 0.6 * g.len;
            ml2 = mom.m * l2;
        }
        const unsigned ent_val = my; // a leaf entry = the leaf's level-order node number (its block of 8 source records: tv.srcL)
        unsigned openmask = 0;
        if(COUNT) {
            st_a++;
            st_al += (unsigned)n;
        }
        // One pass over the wave's targets.  The outcome of every comparison is taken as a 64-bit LANE MASK (the ballot of a bare
        // comparison is the comparison's own result register: no instruction) and the reference's boolean expressions are evaluated
        // once per wave on those masks by the scalar unit; a mask comes back as the predicate of a store through inverse_ballot (it
        // becomes the exec mask: no vector instruction either).  Written with per-lane booleans, hipcc materialised every && / || as
        // v_cndmask / v_and chains: 45 - 50 vector instructions per pass where this form needs ~35.
        // A target gains at most 64 entries per pass (one per lane): only when some list is that close to its capacity are the
        // appends CHECKED one by one.
        const unsigned long long m_leafnode = __builtin_amdgcn_ballot_w64(lk.pcount > 0);
        const unsigned long long m_intnode = ~m_leafnode & __builtin_amdgcn_ballot_w64(lk.nchild > 0);
#ifndef MPG_NO_SINGLES_AS_NODES
        // An OPENED leaf of one particle is listed with the nodes (round 4): its moment record is that particle (centre of mass = its position
        // to a rounding, the same softening: apply_accn_to_output treats both alike), and there it shares an evaluation step with 7 other
        // sources instead of taking one alone - 201 M of the 1916 M opened leaves per walk at 256^3 (a cell split at its 9th particle leaves
        // children of one or two).  The decision stays the reference's and is counted as such (a pair interaction, not a node used).
        const unsigned long long m_single = __builtin_amdgcn_ballot_w64(lk.pcount == 1);
#else
        const unsigned long long m_single = 0ull;
#endif
        auto pass = [&](auto checked_tag) {
            constexpr bool CHECKED = decltype(checked_tag)::value;
#ifdef MPG_LISTS_TGT_PREFETCH
            // experiment: target t + 1's record is requested before target t's pass (the LDS read's latency under the pass before it)
            float4 tf_pre[9];
            if constexpr(F32) {
                unsigned o0 = 0u;
                asm volatile("" : "+v"(o0));
                tf_pre[0] = *(const float4 *)(W.s_tgtf + o0);
            }
#endif
#pragma unroll
            for(int t = 0; t < 8; t++) {
#ifdef MPG_LISTS_TGT_PREFETCH
                if constexpr(F32) {
                    if(t < 7) {
                        unsigned on = 4u * (t + 1);
                        asm volatile("" : "+v"(on));
                        tf_pre[t + 1] = *(const float4 *)(W.s_tgtf + on);
                    }
                }
#endif
                // (no lane for a target that overflowed, is absent or did not open the parent; measured: 28 % of the passes over a
                // target find no lane with its bit - the entries popped late in a walk belong to few of the 8 targets)
                const unsigned long long m_act = __builtin_amdgcn_ballot_w64((mask & (1u << t)) != 0u);
                if(m_act == 0ull)
                    continue;
                // (one address for the wave: a broadcast read.  The empty asm hides from hipcc that the 8 reads are the same in every
                // pass of the walk: hoisted out of the loop they would hold 64 registers)
                unsigned ot = 4u * t;
                asm volatile("" : "+v"(ot));
                unsigned long long m_discard, m_open, m_wrap;
                if constexpr(F32) {
#ifdef MPG_LISTS_TGT_PREFETCH
                    const float4 tf = tf_pre[t];
#else
                    const float4 tf = *(const float4 *)(W.s_tgtf + ot);
#endif
                    const float cdx = Cx - tf.x, cdy = Cy - tf.y, cdz = Cz - tf.z;
                    const float dx = Sx - tf.x, dy = Sy - tf.y, dz = Sz - tf.z;
                    const float cmax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(cdx), __builtin_fabsf(cdy)), __builtin_fabsf(cdz));
                    const float r2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                    float wT; // min(wA, wB sqrt(aold)) - written out: for fminf hipcc first canonicalises wA (a v_max_f32 x, x per pass)
                    asm("v_min_f32 %0, %1, %2" : "=v"(wT) : "v"(wA), "v"(wB * tf.w));
                    const float n1 = fmaf(r2, W.w1, W.c1), n2 = fmaf(cmax, rE, c2), n3 = fmaf(cmax, rE, c3), n4 = fmaf(r2, wT, c4);
                    const float dmin = __builtin_fminf(n1, n2), omin = __builtin_fminf(n3, n4);
                    const float amb = __builtin_fminf(__builtin_fabsf(dmin), __builtin_fabsf(omin));
                    m_discard = __builtin_amdgcn_ballot_w64(dmin > 0.0f);
                    m_open = __builtin_amdgcn_ballot_w64(omin < 0.0f);
                    m_wrap = 0ull;
                    const unsigned long long m_amb = (__builtin_amdgcn_ballot_w64(!(amb > 1.0f)) | m_nweird) & m_act;
                    const bool redo = m_amb != 0ull || ((W.tweird >> t) & 1u);
#ifdef MPG_F32_NOEXPECT
                    if(COUNT || redo) { // (rare: the node again, and the reference's arithmetic)
#else
                    if(COUNT || __builtin_expect(redo, 0)) { // (rare - out of line: the node again, and the reference's arithmetic)
#endif
                        unsigned my2 = my;
                        asm volatile("" : "+v"(my2));
                        const NodeGeo g2 = ld<O32>(tv.geoB, my2);
                        const Src4 mom2 = ld<O32>(tv.momB, my2);
                        const double4 tg = *(const double4 *)(s_tgt + ot);
                        const double l2d = g2.len * g2.len;
                        unsigned long long e_discard, e_open, e_wrap;
                        node_test_masks<MODE>(gp, g2, mom2, false, false, fma(0.5, g2.len, gp.rcut), l2d, 0.6 * g2.len, mom2.m * l2d, tg.x, tg.y, tg.z,
                                              tg.w, e_discard, e_open, e_wrap);
                        if(COUNT && !redo) {
                            // (the open decision of a discarded node is never used: compared on the kept lanes only)
                            if((((m_discard ^ e_discard) | ((m_open ^ e_open) & ~e_discard)) & m_act) != 0ull && lane == 0)
                                atomicExch(&ctl[1], 3u);
                        }
                        if(COUNT && redo && n_amb)
                            (*n_amb)++;
                        m_discard = e_discard;
                        m_open = e_open;
                    }
                }
                else {
                    const double4 tg = *(const double4 *)(s_tgt + ot);
                    node_test_masks<MODE>(gp, g, mom, false, false, eff, l2, inside, ml2, tg.x, tg.y, tg.z, tg.w, m_discard, m_open, m_wrap);
                }
                const unsigned long long keep = m_act & ~m_discard;
                const unsigned long long bn0 = keep & ~m_open;              // used unopened
                const unsigned long long bl0 = keep & m_open & m_leafnode;  // opened leaves
                const unsigned long long bn = bn0 | (bl0 & m_single);       // ... entries of the node list (with the opened one-particle leaves)
                const unsigned long long bl = bl0 & ~m_single;              // ... entries of the leaf list
                const unsigned long long bpush = keep & m_open & m_intnode;
                const int kl = __builtin_popcountll(bl), kn = __builtin_popcountll(bn);
                if(CHECKED && T.nleaf[t] + T.nnode[t] + kl + kn > cap) { // the lists of target t are full: the fallback kernel walks it again
                    overflowed |= 1u << t;
                    live &= ~(1u << t);
                    continue;
                }
                unsigned *__restrict__ Lt = Lw + (unsigned)(t * cap); // (wave-uniform base)
                if(__builtin_amdgcn_inverse_ballot_w64(bl)) // position = entries so far + set bits below this lane (v_mbcnt accumulates onto its last operand)
                    st32(Lt, __builtin_amdgcn_mbcnt_hi((unsigned)(bl >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bl, (unsigned)T.nleaf[t])), ent_val);
                if(__builtin_amdgcn_inverse_ballot_w64(bn))
                    st32(Lt, (unsigned)(cap - 1 - T.nnode[t]) - mbcnt64(bn), my);
                T.nleaf[t] += kl;
                T.nnode[t] += kn;
                maxused = max(maxused, T.nleaf[t] + T.nnode[t]);
                openmask |= __builtin_amdgcn_inverse_ballot_w64(bpush) ? (1u << t) : 0u;
                // an entry on a wrapped periodic image: MODE 2 can meet one only among the root and its children (the top passes above)
                if(MODE == 1)
                    wmask |= m_wrap & (bl | bn);
                if(COUNT) {
                    T.c_vis[t] += (unsigned)__builtin_popcountll(m_act);
                    T.c_used[t] += (unsigned)__builtin_popcountll(bn0);
                    c_pp[t] += __builtin_amdgcn_inverse_ballot_w64(bl0) ? (unsigned)lk.pcount : 0u;
#ifdef MPG_LEAF_HIST // experiment: opened leaves by particle count
                    if(__builtin_amdgcn_inverse_ballot_w64(bl))
                        atomicAdd(&g_leaf_hist[lk.pcount], 1ull);
#endif
                }
                // (the 8 targets' tests are independent: left alone, hipcc interleaves them and runs out of registers)
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if(maxused + 64 > cap)
            pass(std::true_type{});
        else
            pass(std::false_type{});
        if(!push_children(openmask != 0u, lk, openmask))
            break;
#ifdef MPG_LISTS_PREFETCH
        if(nxt.n == 0 && sp > 0)
            pop(nxt); // nothing waited below: on with what this batch pushed
        cur = nxt;
#endif
    }
    wrapped = wmask != 0ull;
    return true;
}

// one wave = one chunk of k_walk_eval (8 consecutive targets)
template <bool COUNT, bool FASTWRAP, bool O32, bool F32>
__global__ void __launch_bounds__(256, MPG_LIST_BLOCKS) k_walk_lists8(const TreeView tv, const GravParams gp, const WalkIO io, unsigned *__restrict__ lists,
                                                      int2 *__restrict__ counts, const int cap, const int64_t slot0, const int64_t nslots,
                                                      unsigned *__restrict__ ctl, int *__restrict__ ovf)
{
    __shared__ unsigned s_qnode[4 * QCAP];
    __shared__ unsigned char s_qmask[4 * QCAP];
    __shared__ __attribute__((aligned(32))) double s_tgt4[4 * 8 * 4];
    __shared__ __attribute__((aligned(16))) float s_tgtf4[F32 ? 4 * 8 * 4 : 4];
#ifdef MPG_EXP_LDSPAD_LISTS // timing experiment: fewer resident blocks per CU with the same code (bytes of unused LDS)
    __shared__ unsigned s_pad[MPG_EXP_LDSPAD_LISTS / 4];
    if(gp.box < 0)
        s_pad[threadIdx.x] = 1u, s_qnode[0] = s_pad[(threadIdx.x + 1) & 255];
#endif
    set_wave_prio(io.list_prio);
    const int lane = threadIdx.x & 63;
    unsigned *q_node = s_qnode + (threadIdx.x >> 6) * QCAP;
    unsigned char *q_mask = s_qmask + (threadIdx.x >> 6) * QCAP;
    double *s_tgt = s_tgt4 + (threadIdx.x >> 6) * 32;
    const unsigned nchunks = (unsigned)((nslots + 7) / 8);
    const ChunkIter it(nchunks);
    const unsigned guard_max = (unsigned)min((long long)(8ll * (tv.nnodes + 1024)), 0x7fffffffll);
    const double face = gp.rcut + 0.002 * gp.box;
    unsigned long long n_pp = 0, n_vis = 0, n_used = 0, n_le = 0, n_se = 0;
    unsigned st_a = 0, st_al = 0, n_amb = 0, n_f32w = 0;
    float *s_tgtf = s_tgtf4 + (F32 ? (threadIdx.x >> 6) * 32 : 0);

    for(unsigned chunk = it.lo + it.first; chunk < it.hi; chunk += it.stride) {
        // lane t < 8 fetches target t; the values are then made wave-uniform
        const int64_t rel = (int64_t)chunk * 8 + (lane & 7);
        const bool tvalid = lane < 8 && rel < nslots;
        int ci = -1;
        double vx = 0, vy = 0, vz = 0, vaold = 0;
        if(tvalid) {
            const int64_t slot = slot0 + rel;
            ci = io.targets ? io.targets[slot] : tv.order[slot];
            vx = io.pos[3 * (int64_t)ci + 0];
            vy = io.pos[3 * (int64_t)ci + 1];
            vz = io.pos[3 * (int64_t)ci + 2];
            double old = 0;
            if(io.oldacc)
                old = io.oldacc[ci];
            else if(io.prev_accel) { // grav_get_abs_accel, gravshort.h:70-80
                double s2 = 0;
                for(int j = 0; j < 3; j++) {
                    const double a = io.prev_accel[3 * (int64_t)ci + j] + (io.gravpm ? io.gravpm[3 * (int64_t)ci + j] : 0.0);
                    s2 += a * a;
                }
                old = sqrt(s2) / gp.G;
            }
            // (Barnes-Hut walk: the relative criterion "mass l^2 > r^4 aold" must never hold)
            vaold = gp.use_bh ? __builtin_inf() : gp.errtol * old;
        }
        const bool near_face = tvalid && (fmin(fmin(vx, vy), vz) < face || fmax(fmax(vx, vy), vz) > gp.box - face);
        const unsigned live0 = (unsigned)(__builtin_amdgcn_ballot_w64(tvalid) & 0xffull);
        WaveTargets T;
        unsigned c_pp[8];
        __builtin_amdgcn_wave_barrier(); // (the previous chunk's reads of s_tgt are done: LDS operations of a wave complete in order)
        if(lane < 8)
            *(double4 *)(s_tgt + 4 * lane) = make_double4(vx, vy, vz, vaold);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for(int t = 0; t < 8; t++) {
            T.nleaf[t] = T.nnode[t] = 0;
            T.c_vis[t] = T.c_used[t] = 0u;
            c_pp[t] = 0u;
        }
        unsigned *__restrict__ L = lists + (size_t)__builtin_amdgcn_readfirstlane((int)chunk) * (size_t)cap * 8;
        unsigned overflowed = 0;
        bool wrapped = false; // (one flag for the wave: k_walk_eval takes NEAREST() for a whole chunk anyway)
        bool ok;
        if(FASTWRAP) {
            if(!any_lane(near_face)) {
                if constexpr(F32) {
                    // the wave's origin = its first valid target; D = the largest |p_t - O| of an axis; the targets relative to O and
                    // sqrt(aold) as floats in LDS (see F32Wave)
                    F32Wave W;
                    const int l0 = live0 ? __builtin_ctz(live0) : 0;
                    W.ox = __shfl(vx, l0);
                    W.oy = __shfl(vy, l0);
                    W.oz = __shfl(vz, l0);
                    double dmax = tvalid ? fmax(fmax(fabs(vx - W.ox), fabs(vy - W.oy)), fabs(vz - W.oz)) : 0.0;
                    for(int off = 1; off < 8; off <<= 1)
                        dmax = fmax(dmax, __shfl_xor(dmax, off));
                    const double D = __shfl(dmax, 0) * 1.000001;
                    const double E1 = 24.0 * (double)F32_U * gp.rcut2 + 20.0 * (double)F32_U * D * gp.rcut;
                    W.w1 = (float)(1.0 / E1);
                    W.c1 = -(float)gp.rcut2 * W.w1;
                    W.dr = (float)((D + gp.rcut) * 1.000001);
                    W.k4 = (float)(32.0 * (double)F32_U * D * sqrt(gp.bhangle2) * 1.000001);
                    // aold that fp32 carries: 0, +inf (the Barnes-Hut switch) or 1e-30 .. 1e30
                    const bool tw = tvalid && !(vaold == 0.0 || vaold == __builtin_inf() || (vaold > 1e-30 && vaold < 1e30));
                    W.tweird = (unsigned)(__builtin_amdgcn_ballot_w64(tw) & 0xffull);
                    W.s_tgtf = s_tgtf;
                    __builtin_amdgcn_wave_barrier();
                    if(lane < 8)
                        *(float4 *)(s_tgtf + 4 * lane) = make_float4((float)(vx - W.ox), (float)(vy - W.oy), (float)(vz - W.oz), (float)sqrt(vaold));
                    __builtin_amdgcn_wave_barrier();
                    n_f32w++;
                    ok = walk_wave8<COUNT, 2, O32, true>(tv, gp, L, q_node, q_mask, s_tgt, cap, lane, live0, T, overflowed, wrapped, c_pp, guard_max, ctl, st_a,
                                                         st_al, W, &n_amb);
                }
                else
                    ok = walk_wave8<COUNT, 2, O32>(tv, gp, L, q_node, q_mask, s_tgt, cap, lane, live0, T, overflowed, wrapped, c_pp, guard_max, ctl, st_a, st_al);
            }
            else
                ok = walk_wave8<COUNT, 1, O32>(tv, gp, L, q_node, q_mask, s_tgt, cap, lane, live0, T, overflowed, wrapped, c_pp, guard_max, ctl, st_a, st_al);
        }
        else
            ok = walk_wave8<COUNT, 0, O32>(tv, gp, L, q_node, q_mask, s_tgt, cap, lane, live0, T, overflowed, wrapped, c_pp, guard_max, ctl, st_a, st_al);
        if(!ok)
            return;
        int nl = 0, nn = 0;
#pragma unroll
        for(int t = 0; t < 8; t++)
            if(lane == t) {
                nl = T.nleaf[t];
                nn = T.nnode[t];
            }
        if(tvalid) {
            const bool overflow = (overflowed >> lane) & 1u;
            // the work this target causes in the two kernels: 8 lanes per leaf entry and 1 per node entry in the evaluation, and about
            // as many node tests as it has entries in the list construction (the measure only has to be proportional to the time
            // spent: domain.c:611)
            if(io.cost)
                io.cost[ci] = (float)(8 * (overflow ? cap : nl) + nn + 3 * (nl + nn));
            if(overflow) {
                counts[rel] = make_int2(-1, 0);
                ovf[atomicAdd(&ctl[0], 1u)] = ci;
            }
            else {
                counts[rel] = make_int2(nl | (wrapped ? (1 << 30) : 0), nn);
                if((unsigned)(nl + nn) > ctl[2])
                    atomicMax(&ctl[2], (unsigned)(nl + nn));
            }
        }
        if(COUNT) {
#pragma unroll
            for(int t = 0; t < 8; t++)
                if(!((overflowed >> t) & 1u)) { // an overflowed target is walked again, and counted, by the fallback kernel
                    n_pp += c_pp[t]; // (per lane; summed over the wave below)
                    if(lane == 0) {
                        n_vis += T.c_vis[t];
                        n_used += T.c_used[t];
                        n_le += (unsigned)T.nleaf[t];
                        n_se += (unsigned)T.nnode[t];
                    }
                }
        }
    }
    if(COUNT) {
        unsigned long long c0 = n_pp, c1 = n_vis, c2 = n_used, c3 = lane == 0 ? st_a : 0u, c4 = lane == 0 ? st_al : 0u;
        for(int off = 32; off > 0; off >>= 1) {
            c0 += __shfl_down(c0, off);
            c1 += __shfl_down(c1, off);
            c2 += __shfl_down(c2, off);
            c3 += __shfl_down(c3, off);
            c4 += __shfl_down(c4, off);
        }
        if(lane == 0) {
            atomicAdd(&io.counters[5], n_le); // list entries written: leaves, nodes (lane 0 holds the wave's sums)
            atomicAdd(&io.counters[6], n_se);
            atomicAdd(&io.counters[0], c0);
            atomicAdd(&io.counters[1], c1);
            atomicAdd(&io.counters[2], c2);
            atomicAdd(&io.counters[3], c3);
            atomicAdd(&io.counters[4], c4);
            if(F32 && n_amb)
                atomicAdd(&io.counters[12], (unsigned long long)n_amb); // target passes that fell back to fp64 (lane 0 counts: wave-uniform)
            if(F32 && n_f32w)
                atomicAdd(&io.counters[13], (unsigned long long)n_f32w); // waves (chunks of 8 targets) whose main loop ran the fp32 tests
        }
    }
}

// The two list loops of one group (8 lanes, lane s <-> source s of a leaf entry / entry r0 + s of the node list).
// WRAP: take NEAREST() per pair (partmanager.h:99); otherwise plain differences (bit-identical where no image is wrapped).
//
// Leaf entries are fetched 8 per group at a time (one coalesced 256-byte read per wave: lane s takes entry e0 + s), one batch ahead of
// their use, and staged in a per-group ring of 16 words in LDS; the pair loop reads the two entries of its next steps with ONE
// ds_read2_b32 at a wave-uniform offset (all 8 lanes of a group read the same word: a broadcast).  Rounds 1-4 kept the batch in a
// register and broadcast entry J with ds_bpermute per pair: v_and_or + v_cndmask + v_lshlrev + ds_bpermute + s_waitcnt lgkmcnt(0) in
// front of EVERY source load - 3 vector and 1 LDS instruction more per pair step than this form, with the LDS latency exposed.
// Sources are requested two pair evaluations ahead of their use.  The scheduling barriers and the empty asm keep hipcc from
// interleaving or sinking the (independent) pair evaluations, which would triple the live registers.  A lane without a
// source (short leaf, list exhausted) reads a zero-mass padding record behind the tree's source array instead: its pair
// evaluates to exactly zero, so the accumulators are updated unconditionally (a conditional update makes hipcc keep a
// renamed copy of the four accumulators per unrolled stage).
// The lists of group g start at L + g * cap (layout: top of this file).
constexpr int RING_STRIDE = 17; // words per group: 16 used; the odd stride puts the 8 groups' words on different banks

template <bool POT, bool WRAP, bool O32>
__device__ __forceinline__ void eval_lists(const TreeView &tv, const GravParams &gp, const unsigned *__restrict__ L, const int cap, const int nleaf,
                                           const int nnode, const int s, const int gshift, const double px,
                                           const double py, const double pz, const double *__restrict__ s_wtab, unsigned *__restrict__ ring_g,
                                           double &ax, double &ay, double &az, double &pot)
{
    const unsigned empty = (unsigned)tv.nnodes; // the block of zero-mass records behind the last node's (TreeBuilder::ensure_leaf_pad)
    // source s of leaf entry EJ: record s of block EJ of tv.srcL - a leaf's particles filled up to 8 with zero-mass records, so that neither
    // the leaf's count nor a select for the lanes beyond it is needed (rounds 3-5 read tv.src[first + s] with the count in the entry's low
    // bits: v_and, v_and, v_lshl_add, v_cmp, v_cndmask per pair step where this form is one v_lshl_add)
    const unsigned s32 = (unsigned)s * 32u;
#define MPG_LOAD_REAL(EJ, SV)                                                                   \
    {                                                                                           \
        const unsigned ej_ = (EJ);                                                              \
        if(O32)                                                                                 \
            SV = *(const Src4 *)((const char *)tv.srcL + (size_t)((ej_ << 8) + s32));           \
        else                                                                                    \
            SV = tv.srcL[(size_t)ej_ * 8 + (size_t)s];                                          \
    }
#ifdef MPG_EXP_NOLOAD // timing experiment (wrong results): the leaf loop keeps the records it started with - no load, no other instruction
#define MPG_LOAD(EJ, SV) asm volatile("" : "+v"(SV.x), "+v"(SV.y), "+v"(SV.z), "+v"(SV.m) : "v"(EJ))
#else
#define MPG_LOAD(EJ, SV) MPG_LOAD_REAL(EJ, SV)
#endif
#define MPG_EVAL(SV)                                                              \
    {                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                        \
        double dx_ = SV.x - px, dy_ = SV.y - py, dz_ = SV.z - pz;                 \
        if(WRAP) {                                                                \
            dx_ = nearest_img(dx_, gp.box, gp.invbox);                            \
            dy_ = nearest_img(dy_, gp.box, gp.invbox);                            \
            dz_ = nearest_img(dz_, gp.box, gp.invbox);                            \
        }                                                                         \
        pair_force<POT>(SV, dx_, dy_, dz_, gp, s_wtab, ax, ay, az, pot);           \
        asm volatile("" : "+v"(ax), "+v"(ay), "+v"(az), "+v"(pot));               \
        __builtin_amdgcn_sched_barrier(0);                                        \
    }
    {
        // two source buffers (A, B) used alternately: while one pair is evaluated the other buffer's load is in flight.  The
        // stage loop is deliberately not unrolled beyond that: every unrolled stage carries its own copy of the (rare)
        // softened branch, and those copies are what drives register pressure and code size.
        // index of leaf entry e0 + s (e0 a multiple of 8)
        const unsigned ls = (unsigned)((gshift >> 3) * cap + s);
#define MPG_LEAF_AT(E0) (ls + (unsigned)(E0))
        // batch 0 into ring slot 0, batch 1 requested.  (LDS operations of one wave complete in order; the wave barriers only keep
        // hipcc from moving the ring's reads over its writes.)
        __builtin_amdgcn_wave_barrier(); // (the previous chunk's reads of the ring are done)
        ring_g[s] = (s < nleaf) ? ld<true>(L, MPG_LEAF_AT(0)) : empty;
        unsigned ent_n = (8 + s < nleaf) ? ld<true>(L, MPG_LEAF_AT(8)) : empty;
        __builtin_amdgcn_wave_barrier();
        unsigned eb = ring_g[1];
        Src4 A, B;
        MPG_LOAD_REAL(ring_g[0], A);
#if defined(MPG_EXP_NOLOAD) || !defined(MPG_EVAL_PF2)
        B = A;
#endif
        // ONE loop over pairs of entries up to the longest leaf list of the wave's 8 targets (a loop over batches around a loop over
        // the pairs of a batch made hipcc copy the four accumulators out and back at every batch: 8 v_mov_b64 per 8 pair steps; and
        // whole batches ran up to 6 pair steps beyond the longest list)
        int wmax = nleaf;
        for(int off = 8; off < 64; off <<= 1)
            wmax = max(wmax, __shfl_xor(wmax, off));
        wmax = __builtin_amdgcn_readfirstlane(wmax);
#ifdef MPG_EVAL_ILP2
        // Round 5 experiment (measured: 38.2 against 35.9 ms per evaluation at 256^3, removed from the default): the two pairs of a trip
        // side by side up to the softening test, one branch for both, then their table parts side by side.  The kernel's time falls with
        // the resident waves (2 / 3 / 4 per SIMD: 52.8 / 41.3 / 36.1 ms), but what the waves hide is the latency of the source loads, not of
        // the dependent arithmetic: here both records of a trip are needed at its start and are requested only after the previous
        // trip's first half, i.e. closer to their use than in the alternating form below.
        MPG_LOAD(eb, B);
        const double h2 = gp.h2;
#pragma unroll 1
        for(int e = 0; e < wmax; e += 2) {
            if((e & 7) == 0) {
                ring_g[((e + 8) & 8) + s] = ent_n;
                ent_n = (e + 16 + s < nleaf) ? ld<true>(L, MPG_LEAF_AT(e + 16)) : empty;
                __builtin_amdgcn_wave_barrier();
            }
            const unsigned *__restrict__ rr = ring_g + ((e + 2) & 15);
            const unsigned ea = rr[0];
            eb = rr[1];
            __builtin_amdgcn_sched_barrier(0);
            PairTmp ta, tb;
            {
                double dx_ = A.x - px, dy_ = A.y - py, dz_ = A.z - pz;
                double ex_ = B.x - px, ey_ = B.y - py, ez_ = B.z - pz;
                if(WRAP) {
                    dx_ = nearest_img(dx_, gp.box, gp.invbox);
                    dy_ = nearest_img(dy_, gp.box, gp.invbox);
                    dz_ = nearest_img(dz_, gp.box, gp.invbox);
                    ex_ = nearest_img(ex_, gp.box, gp.invbox);
                    ey_ = nearest_img(ey_, gp.box, gp.invbox);
                    ez_ = nearest_img(ez_, gp.box, gp.invbox);
                }
                pair_pre(A, dx_, dy_, dz_, ta);
                pair_pre(B, ex_, ey_, ez_, tb);
            }
            MPG_LOAD(ea, A);
            MPG_LOAD(eb, B);
            if((ta.r2 < h2) | (tb.r2 < h2)) { // rare: one branch for both pairs
                if(ta.r2 < h2)
                    pair_soft(gp, ta);
                if(tb.r2 < h2)
                    pair_soft(gp, tb);
            }
            pair_post<POT>(ta, gp, s_wtab, ax, ay, az, pot);
            pair_post<POT>(tb, gp, s_wtab, ax, ay, az, pot);
            asm volatile("" : "+v"(ax), "+v"(ay), "+v"(az), "+v"(pot));
            __builtin_amdgcn_sched_barrier(0);
        }
#elif !defined(MPG_EVAL_PF2)
        // three source buffers: every record is requested TWO pair evaluations ahead of its use (rounds 1-5 alternated two buffers, one
        // evaluation ahead - kept under MPG_EVAL_PF2: 36.1 - 36.4 against 34.6 ms per evaluation at 256^3 on one box; the kernel's time falls
        // with the resident waves - 2 / 3 / 4 per SIMD: 52.8 / 41.3 / 36.1 ms - and what the waves hide is the latency of these loads:
        // without them, MPG_EXP_NOLOAD, the evaluation of a fixed set of lists takes 20.3 instead of 24.5 ms)
        Src4 Cq = A;
        MPG_LOAD(eb, B);
        int staged = 1; // batches of 8 entries written to the ring so far (batch `staged` is in flight in ent_n)
#pragma unroll 1
        for(int e = 0; e < wmax; e += 3) {
            if(e + 4 >= 8 * staged) { // (wave-uniform) this trip reads into the next batch: write it over the one before the current
                ring_g[(staged & 1) * 8 + s] = ent_n;
                ent_n = ((staged + 1) * 8 + s < nleaf) ? ld<true>(L, MPG_LEAF_AT((staged + 1) * 8)) : empty;
                staged++;
                __builtin_amdgcn_wave_barrier();
            }
            const unsigned e2 = ring_g[(e + 2) & 15], e3 = ring_g[(e + 3) & 15], e4 = ring_g[(e + 4) & 15];
            MPG_LOAD(e2, Cq);
            MPG_EVAL(A);
            MPG_LOAD(e3, A);
            MPG_EVAL(B);
            MPG_LOAD(e4, B);
            MPG_EVAL(Cq);
        }
#else
#pragma unroll 1
        for(int e = 0; e < wmax; e += 2) {
            if((e & 7) == 0) { // (wave-uniform) the next batch into the other half of the ring - its first entry is read in the
                               // last stage of this batch -, the one after that requested
                ring_g[((e + 8) & 8) + s] = ent_n;
                ent_n = (e + 16 + s < nleaf) ? ld<true>(L, MPG_LEAF_AT(e + 16)) : empty;
                __builtin_amdgcn_wave_barrier();
            }
            MPG_LOAD(eb, B);
            // entries e + 2 (the next A) and e + 3 (the next B): one ds_read2_b32, back before the evaluation of A ends
            const unsigned *__restrict__ rr = ring_g + ((e + 2) & 15);
            const unsigned ea = rr[0];
            eb = rr[1];
            MPG_EVAL(A);
            MPG_LOAD(ea, A);
            MPG_EVAL(B);
        }
#endif
    }
    // ---- node entries (level-order indices): lane s takes entry r0 + s; entries two batches ahead, moments one
    if(any_lane(nnode > 0)) {
        const unsigned NONE = (unsigned)tv.nnodes; // a zero-mass padding record behind the moments (TreeBuilder::make_level_order)
        // index of node entry r0 + s counted from the top of the list (r0 a multiple of 8)
        const unsigned top = (unsigned)((gshift >> 3) * cap + cap - 1 - s);
#define MPG_NODE_AT(R0) (top - (unsigned)(R0))
#ifdef MPG_EVAL_PF2
        unsigned ne = (s < nnode) ? ld<true>(L, MPG_NODE_AT(0)) : NONE;
        unsigned ne_n = (8 + s < nnode) ? ld<true>(L, MPG_NODE_AT(8)) : NONE;
        Src4 sc = ld<O32>(tv.momB, ne);
        for(int r0 = 0;; r0 += 8) {
            if(!any_lane(r0 < nnode))
                break;
            ne = ne_n;
            ne_n = (r0 + 16 + s < nnode) ? ld<true>(L, MPG_NODE_AT(r0 + 16)) : NONE;
            const Src4 sc_n = ld<O32>(tv.momB, ne);
            MPG_EVAL(sc);
            sc = sc_n;
        }
#else
        // three moment buffers in rotation (no register copies): the moments of a batch are requested two evaluations ahead of their use,
        // its entries three evaluations before that
#define MPG_NODE_ENT(R0) (((R0) + s < nnode) ? ld<true>(L, MPG_NODE_AT(R0)) : NONE)
        Src4 Sa = ld<O32>(tv.momB, MPG_NODE_ENT(0));
        Src4 Sb = ld<O32>(tv.momB, MPG_NODE_ENT(8));
        Src4 Sc;
        unsigned nx = MPG_NODE_ENT(16), ny = MPG_NODE_ENT(24), nz = MPG_NODE_ENT(32);
#pragma unroll 1
        for(int r0 = 0;; r0 += 24) {
            if(!any_lane(r0 < nnode))
                break;
            Sc = ld<O32>(tv.momB, nx);
            nx = MPG_NODE_ENT(r0 + 40);
            MPG_EVAL(Sa);
            if(!any_lane(r0 + 8 < nnode))
                break;
            Sa = ld<O32>(tv.momB, ny);
            ny = MPG_NODE_ENT(r0 + 48);
            MPG_EVAL(Sb);
            if(!any_lane(r0 + 16 < nnode))
                break;
            Sb = ld<O32>(tv.momB, nz);
            nz = MPG_NODE_ENT(r0 + 56);
            MPG_EVAL(Sc);
        }
#undef MPG_NODE_ENT
#endif
    }
#undef MPG_LOAD
#undef MPG_LOAD_REAL
#undef MPG_EVAL
#undef MPG_LEAF_AT
#undef MPG_NODE_AT
}

// grav_short_postprocess for the potential, gravshort.h:88-96.  Out of line: inlined, the constants of pow()'s expansion were hoisted out of
// the loop over the wave's chunks and cost the evaluation loops two registers (spilled: 1 KB of scratch per wave)
__device__ __attribute__((noinline)) double potential_postprocess(double pot, const double m, const double h, const double cbrtrho0, const double G)
{
    pot += m / (h / 2.8);
    pot -= 2.8372975 * pow(m, 2.0 / 3) * cbrtrho0;
    return pot * G;
}

template <bool POT, bool FASTWRAP, bool O32, int BLK>
__global__ void __launch_bounds__(256, BLK) k_walk_eval(const TreeView tv, const GravParams gp, const WalkIO io, const unsigned *__restrict__ lists,
                                                    const int2 *__restrict__ counts, const int cap, const int64_t slot0, const int64_t nslots)
{
    constexpr int ROW = POT ? 4 : 2;
    __shared__ __attribute__((aligned(16))) double s_wtab[NTAB * ROW];
    __shared__ unsigned s_ring[4 * 8 * RING_STRIDE]; // per wave and group: two batches of leaf entries (eval_lists)
#ifdef MPG_EXP_LDSPAD // timing experiment: fewer resident blocks per CU with the same code (bytes of unused LDS)
    __shared__ unsigned s_pad[MPG_EXP_LDSPAD / 4];
    if(gp.box < 0)
        s_pad[threadIdx.x] = 1u, s_ring[0] = s_pad[(threadIdx.x + 1) & 255];
#endif
    set_wave_prio(io.eval_prio);
    for(int i = threadIdx.x; i < NTAB; i += blockDim.x) {
        const bool last = i == NTAB - 1; // the row the clamp lands on: zeros
        s_wtab[i * 2 + 0] = last ? 0.0 : (double)io.tab_force[i];
        s_wtab[i * 2 + 1] = last ? 0.0 : (double)io.tab_force[i + 1] - (double)io.tab_force[i];
        if(POT) {
            s_wtab[2 * NTAB + i * 2 + 0] = last ? 0.0 : (double)io.tab_pot[i];
            s_wtab[2 * NTAB + i * 2 + 1] = last ? 0.0 : (double)io.tab_pot[i + 1] - (double)io.tab_pot[i];
        }
#ifdef MPG_TAB_AFFINE
        s_wtab[i * 2 + 0] -= (double)i * s_wtab[i * 2 + 1];
        if(POT)
            s_wtab[2 * NTAB + i * 2 + 0] -= (double)i * s_wtab[2 * NTAB + i * 2 + 1];
#endif
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int grp = lane >> 3, s = lane & 7;
    const int gshift = grp * 8;
    unsigned *__restrict__ ring_g = s_ring + ((threadIdx.x >> 6) * 8 + grp) * RING_STRIDE;
    const unsigned nchunks = (unsigned)((nslots + 7) / 8);
    const ChunkIter it(nchunks);

    for(unsigned chunk = it.lo + it.first; chunk < it.hi; chunk += it.stride) {
        const int64_t rel = (int64_t)chunk * 8 + grp;
        bool valid = rel < nslots;
        const int64_t slot = slot0 + rel;
        int ci = -1;
        double px = 0, py = 0, pz = 0;
        int nleaf = 0, nnode = 0;
        bool wrapped = false;
        if(valid) {
            const int2 c = counts[rel];
            if(c.x < 0)
                valid = false; // overflowed: left to the fallback kernel
            else {
                nleaf = c.x & 0x3fffffff;
                wrapped = (c.x >> 30) & 1;
                nnode = c.y;
                ci = io.targets ? io.targets[slot] : tv.order[slot];
                px = io.pos[3 * (int64_t)ci + 0];
                py = io.pos[3 * (int64_t)ci + 1];
                pz = io.pos[3 * (int64_t)ci + 2];
            }
        }
        const unsigned *__restrict__ L = lists + (size_t)__builtin_amdgcn_readfirstlane((int)chunk) * (size_t)cap * 8; // wave-uniform
        double ax = 0, ay = 0, az = 0, pot = 0;
        if(!FASTWRAP || any_lane(wrapped)) // a target on a wrapped image in this wave: NEAREST() per pair for all 8
            eval_lists<POT, true, O32>(tv, gp, L, cap, nleaf, nnode, s, gshift, px, py, pz, s_wtab, ring_g, ax, ay, az, pot);
        else
            eval_lists<POT, false, O32>(tv, gp, L, cap, nleaf, nnode, s, gshift, px, py, pz, s_wtab, ring_g, ax, ay, az, pot);
        // reduce the partial sums over the 8 lanes of the group
        for(int off = 1; off < 8; off <<= 1) {
            ax += __shfl_xor(ax, off);
            ay += __shfl_xor(ay, off);
            az += __shfl_xor(az, off);
            if(POT)
                pot += __shfl_xor(pot, off);
        }
        if(valid && s == 0) {
            // grav_short_reduce (assign) + grav_short_postprocess, gravshort.h:47-67,88-96
            io.accel[3 * (int64_t)ci + 0] = ax * gp.G;
            io.accel[3 * (int64_t)ci + 1] = ay * gp.G;
            io.accel[3 * (int64_t)ci + 2] = az * gp.G;
            if(POT && io.potential)
                io.potential[ci] = potential_postprocess(pot, (double)io.mass[ci], gp.h, gp.cbrtrho0, gp.G);
        }
    }
}

// Grid of a kernel over `nchunks` chunks of 8 targets.  chunks_per_wave == 0: persistent (one block per resident slot, every wave
// loops over many chunks); otherwise every wave takes about that many chunks, so that blocks of the list-construction kernel
// of one slice and of the evaluation kernel of the previous slice, launched on two streams, share the CUs.
int grid_blocks(WalkScratch &ws, const void *kern, int64_t nchunks, int chunks_per_wave)
{
    if(ws.num_cu == 0) {
        int dev = 0;
        MPG_HIP(hipGetDevice(&dev));
        MPG_HIP(hipDeviceGetAttribute(&ws.num_cu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    const int64_t need = (nchunks + 3) / 4;
    int64_t nblocks;
    if(chunks_per_wave > 0)
        nblocks = (need + chunks_per_wave - 1) / chunks_per_wave;
    else {
        int occ = 0;
        MPG_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, 0));
        if(occ < 1)
            occ = 1;
        if(occ > 8)
            occ = 8;
        nblocks = (int64_t)ws.num_cu * occ;
        if(nblocks > need)
            nblocks = need;
    }
    return (int)((nblocks + 7) / 8 * 8);
}

template <bool POT, bool COUNT, bool FASTWRAP, bool O32>
void launch_split_t(const TreeView &tv, const GravParams &gp, const WalkIO &io, WalkScratch &ws, hipStream_t st)
{
    // fp32 pre-classification of the node tests (round 6): MODE 2 waves of a FASTWRAP walk whose constants fp32 carries.  Built, parity-tested
    // (decisions identical, 1e-4 of the passes fall back to fp64) and measured SLOWER than the fp64 tests on gfx950 (list kernel 26.8 -> 28.4 ms
    // at 256^3: only v_add / v_mul / v_fma_f32 issue at twice the fp64 rate, the min / max / compare half of the fp32 pass costs what fp64 costs,
    // and every popped node pays ~25 more instructions for the conversion - DESIGN.md 3.2, profiles/r06a_lists_f32/): OFF unless MPG_LISTS_F32=1
    // (read per launch: the tests switch it)
    const char *f32_e = getenv("MPG_LISTS_F32");
    const bool f32_env = f32_e && f32_e[0] == '1';
    const bool f32 = FASTWRAP && f32_env && gp.bhangle2 > 1e-6 && gp.bhangle2 < 1e6 && gp.rcut > 1e-12 && gp.rcut < 1e12 && gp.box < 1e12;
    auto kl = (FASTWRAP && f32) ? k_walk_lists8<COUNT, FASTWRAP, O32, FASTWRAP> : k_walk_lists8<COUNT, FASTWRAP, O32, false>;
    // 4 resident blocks per CU (128 registers, hardly a spill) where the lists are short - the evaluation is then bound by instruction issue -
    // and 6 (80 registers) where they are long (a clustered set: the list capacity has grown to >= 4096 entries), where it waits for memory and
    // the waves count: 256^3 clustered 148 -> 127 ms per walk with 6, Zel'dovich 69.2 -> 67.7 with 4
    auto ke = ws.split_cap >= 4096 ? k_walk_eval<POT, FASTWRAP, O32, MPG_EVAL_BLOCKS_LONG> : k_walk_eval<POT, FASTWRAP, O32, MPG_EVAL_BLOCKS>;
    if(const char *e = getenv("MPG_LIST_CAP")) // experiment knob
        ws.split_cap = atoi(e) / 8 * 8;
    const int cap = ws.split_cap;
    // targets per kernel pair: bounded by the list area (split_bytes), at least 64 Ki so that a launch still fills the chip.
    // Long lists (a clustered set: capacity >= 4096 entries = 16 - 32 KiB per target, of which most targets use a small part): the list area of
    // a slice is kept to 16 GiB - the used parts of the lists are then islands in a region the next slice writes again, and a smaller region
    // walks faster (round 6, 256^3 clustered set, walk ms at slices of 5.6 M / 2.6 M (80 GiB) / 1 M / 512 Ki / 256 Ki / 128 Ki / 64 Ki targets:
    // 112.0 / 101.2 / 93.3 / 91.2 / 93.6 / 97.3 / 105.7); with short lists (capacity 1024: 4 KiB per target, a third of it used) ONE slice
    // stays best (59.8 ms against 60.5 / 62.3 / 62.1 / 63.0 / 63.7 with 2 / 4 / 8 / 16 / 32 slices).
    static const size_t long_bytes = getenv("MPG_SPLIT_BYTES_LONG") ? (size_t)atoll(getenv("MPG_SPLIT_BYTES_LONG")) : ((size_t)16 << 30);
    const size_t area = cap >= 4096 ? (ws.split_bytes < long_bytes ? ws.split_bytes : long_bytes) : ws.split_bytes;
    int64_t slice = (int64_t)(area / ((size_t)cap * sizeof(unsigned))) / 2048 * 2048;
    if(slice < 65536)
        slice = 65536;
    if(slice > ws.split_slice)
        slice = ws.split_slice;
    if(const char *e = getenv("MPG_SPLIT_SLICE")) // experiment knob
        slice = atoll(e);
    const int64_t nmax = io.ntargets < slice ? io.ntargets : slice;
    const int64_t nslices = (io.ntargets + slice - 1) / slice;
    static const int ov_env = getenv("MPG_SPLIT_OVERLAP") ? (getenv("MPG_SPLIT_OVERLAP")[0] == '0' ? 0 : 1) : -1; // experiment knob
    const bool overlap = (ov_env >= 0 ? ov_env != 0 : ws.split_overlap) && nslices > 1;
    const size_t lists_sz = (size_t)((nmax + 15) / 16) * 16 * (size_t)cap, counts_sz = (size_t)nmax + 8;
    ws.split_lists.reserve(lists_sz * (overlap ? 2 : 1));
    ws.split_counts.reserve(counts_sz * (overlap ? 2 : 1));
    ws.split_ovf.reserve((size_t)io.ntargets);
    if(overlap && !ws.split_stream) {
        MPG_HIP(hipStreamCreateWithFlags(&ws.split_stream, hipStreamNonBlocking));
        for(int k = 0; k < 2; k++) {
            MPG_HIP(hipEventCreateWithFlags(&ws.ev_lists[k], hipEventDisableTiming));
            MPG_HIP(hipEventCreateWithFlags(&ws.ev_eval[k], hipEventDisableTiming));
        }
        MPG_HIP(hipEventCreateWithFlags(&ws.ev_begin, hipEventDisableTiming));
    }
    static const int cpw_env = getenv("MPG_SPLIT_CPW") ? atoi(getenv("MPG_SPLIT_CPW")) : -1; // experiment knob
    const int cpw = cpw_env >= 0 ? cpw_env : ws.split_chunks_per_wave;
    hipStream_t sl = overlap ? ws.split_stream : st; // list construction runs one slice ahead of the evaluation
    if(overlap) {
        MPG_HIP(hipEventRecord(ws.ev_begin, st)); // tree, counters and the control words are ready
        MPG_HIP(hipStreamWaitEvent(sl, ws.ev_begin, 0));
    }
    int64_t i = 0;
    for(int64_t s0 = 0; s0 < io.ntargets; s0 += slice, i++) {
        const int64_t ns = (io.ntargets - s0 < slice) ? io.ntargets - s0 : slice;
        const int64_t nchunks = (ns + 7) / 8;
        const int b = overlap ? (int)(i & 1) : 0;
        unsigned *lists = ws.split_lists.p + (size_t)b * lists_sz;
        int2 *counts = ws.split_counts.p + (size_t)b * counts_sz;
        if(overlap && i >= 2)
            MPG_HIP(hipStreamWaitEvent(sl, ws.ev_eval[b], 0)); // the evaluation that read this buffer two slices ago is done
        // MPG_SPLIT_TIME=1: the two kernels timed one by one with HIP events (a diagnostic: two host waits per slice)
        static const bool split_time = getenv("MPG_SPLIT_TIME") != nullptr;
        hipEvent_t te[3] = {nullptr, nullptr, nullptr};
        if(split_time) {
            for(auto &e : te)
                MPG_HIP(hipEventCreate(&e));
            MPG_HIP(hipEventRecord(te[0], sl));
        }
        // (a counting walk ends every wave with 7 atomic adds on the same words: fewer, longer-lived waves there)
        hipLaunchKernelGGL(kl, dim3((unsigned)grid_blocks(ws, (const void *)kl, nchunks, (COUNT && cpw > 0 && cpw < 8) ? 8 : cpw)), dim3(256), 0, sl, tv, gp, io,
                           lists, counts, cap, s0, ns, ws.ctr.p, ws.split_ovf.p);
        if(split_time)
            MPG_HIP(hipEventRecord(te[1], sl));
        if(overlap) {
            MPG_HIP(hipEventRecord(ws.ev_lists[b], sl));
            MPG_HIP(hipStreamWaitEvent(st, ws.ev_lists[b], 0));
        }
        if(ws.ev_mid && nslices == 1 && !overlap) {
            MPG_HIP(hipEventRecord(ws.ev_mid, st));
            ws.mid_recorded = true;
        }
        if(ws.ev_before_eval && i == 0)
            MPG_HIP(hipStreamWaitEvent(st, ws.ev_before_eval, 0)); // the leaf blocks, made on the tree's stream beside the list kernel
        hipLaunchKernelGGL(ke, dim3((unsigned)grid_blocks(ws, (const void *)ke, nchunks, cpw)), dim3(256), 0, st, tv, gp, io, lists, counts, cap, s0,
                           ns);
        if(overlap)
            MPG_HIP(hipEventRecord(ws.ev_eval[b], st));
        if(split_time) {
            MPG_HIP(hipEventRecord(te[2], st));
            MPG_HIP(hipEventSynchronize(te[2]));
            float ml = 0, me = 0;
            MPG_HIP(hipEventElapsedTime(&ml, te[0], te[1]));
            MPG_HIP(hipEventElapsedTime(&me, te[1], te[2]));
            fprintf(stderr, "SPLIT_TIME targets %lld lists %.3f ms eval %.3f ms\n", (long long)ns, ml, me);
            for(auto &e : te)
                MPG_HIP(hipEventDestroy(e));
        }
    }
    MPG_HIP(hipGetLastError());
}

} // namespace

void launch_grav_walk_split(const TreeView &tv, const GravParams &gp, const WalkIO &io, bool want_pot, bool count, bool fastwrap, int thresh,
                            WalkScratch &ws, hipStream_t st)
{
    if(io.ntargets == 0)
        return;
    MPG_CHECK(tv.npart < (1ll << 29), "split walk: more than 2^29 particles in one tree");
    ws.ctr.reserve(16);
    // 32-bit byte offsets into the source and node arrays (32-byte records, padding included)?
    MPG_CHECK(tv.srcL != nullptr, "split walk: the tree carries no padded leaf sources (TreeBuilder::ensure_leaf_pad)");
    const bool o32 = !ws.split_offsets64 && (tv.npart + tv.nnodes + 64) * 32 < (1ll << 32) && (tv.nnodes + 2) * 256 < (1ll << 32);
#define MPG_WS(P, C)                                                \
    do {                                                            \
        if(fastwrap) {                                              \
            if(o32)                                                 \
                launch_split_t<P, C, true, true>(tv, gp, io, ws, st);  \
            else                                                    \
                launch_split_t<P, C, true, false>(tv, gp, io, ws, st); \
        }                                                           \
        else                                                        \
            launch_split_t<P, C, false, false>(tv, gp, io, ws, st); \
    } while(0)
    unsigned ctl[4] = {0, 0, 0, 0};
    for(;;) {
        MPG_HIP(hipMemsetAsync(ws.ctr.p, 0, 16 * sizeof(unsigned), st));
        if(want_pot) {
            if(count)
                MPG_WS(true, true);
            else
                MPG_WS(true, false);
        }
        else {
            if(count)
                MPG_WS(false, true);
            else
                MPG_WS(false, false);
        }
        MPG_HIP(hipMemcpyAsync(ctl, ws.ctr.p, sizeof(ctl), hipMemcpyDeviceToHost, st));
        MPG_HIP(hipStreamSynchronize(st));
        MPG_CHECK(ctl[1] == 0, "short-range walk (list construction) aborted by its loop guard (corrupt tree?)");
#ifdef MPG_LEAF_HIST
        if(count) {
            unsigned long long h[16];
            MPG_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_leaf_hist), sizeof(h)));
            fprintf(stderr, "LEAF_HIST targets %lld opened leaves by count 1..8:", (long long)io.ntargets);
            for(int k = 1; k <= 8; k++)
                fprintf(stderr, " %llu", h[k]);
            fprintf(stderr, "\n");
            memset(h, 0, sizeof(h));
            MPG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_leaf_hist), h, sizeof(h)));
        }
#endif
        // More than a fifth of the targets did not fit their lists (the first walk of a clustered set with the initial capacity):
        // walking them all with the fallback kernel costs seconds (256^3 clustered set: 3.3 s), a second pass of this kernel with
        // four times the capacity a fraction of one.  (Not when the interaction counters are on: they would count twice.)
        if(!count && (int64_t)ctl[0] * 5 > io.ntargets && ws.split_cap < 8192) {
            ws.split_cap = ws.split_cap * 4 < 8192 ? ws.split_cap * 4 : 8192;
            // (the event between the two kernels was recorded in the pass that is now repeated: a lists / eval split taken from it would
            // count a whole first pass as list construction - no split for this walk)
            ws.mid_recorded = false;
            ws.ev_mid = nullptr;
            continue;
        }
        break;
    }
#undef MPG_WS
    ws.split_last_overflow = ctl[0];
    ws.split_last_maxlen = ctl[2];
    if(ctl[0] > 0) {
        WalkIO io2 = io;
        io2.targets = ws.split_ovf.p;
        io2.ntargets = ctl[0];
        // The targets that overflow are the heaviest ones (a dense clump's core: tens of thousands of entries each).  The
        // cooperative kernel drains its lists in place and keeps 8 lanes busy per target; the lane-per-target kernel took
        // 115 ms for the 20 000 such targets of the 128^3 clustered test set, this takes a fraction of that.
        if(getenv("MPG_SPLIT_FALLBACK_LANE"))
            launch_grav_walk(tv, gp, io2, want_pot, count, fastwrap, thresh, st);
        else {
            launch_grav_walk_coop(tv, gp, io2, want_pot, count, fastwrap, ws, st);
            MPG_CHECK(walk_coop_error(ws, st) == 0, "short-range walk (fallback for long lists) aborted by its loop guard");
        }
        if((int64_t)ctl[0] * 50 > io.ntargets && ws.split_cap < 8192) {
            ws.split_cap *= 2; // more than 2 % of the targets overflowed: give the next walk longer lists
            if((int64_t)ctl[0] * 5 > io.ntargets && ws.split_cap < 8192)
                ws.split_cap *= 2; // ... much longer if it was more than 20 % (a clustered set: the fallback is what costs then)
        }
    }
}

} // namespace mpg
