// dist.hip -- the force step on several ranks (one process per GPU) behind the caller's communicator: mpg_dist_* of
// include/mpgadget_hip.h.
//
// The reference's gravpm_force / force_tree_full / grav_short_tree are collective over MPI_COMM_WORLD: particles live on the
// rank that owns their Peano-Hilbert TopLeaf (domain.c:153-258), petapm ships region meshes to 2-D pencils and back
// (petapm.c:584-885), the force tree hangs local sub-trees under a replicated top-tree whose remote leaves are pseudo nodes
// (forcetree.c:654-723, 1145-1284) and the walk exports targets to the owners of the pseudo nodes they open
// (treewalk.c:325-793).  Here, MI355X-first (288 GB per GPU, xGMI point-to-point links):
//
//   PM    {Pos, Mass} of every particle (32 B) goes to the rank(s) owning the x-planes its CIC cloud touches - at Nmesh = 2 N^(1/3)
//         a particle is an eighth of the mesh cells it deposits on, so shipping particles beats shipping region meshes -, the
//         slab solver of pm.hip runs on what arrived (two all-to-all transposes, five neighbour planes), and {GravPM, Potential}
//         (32 B) returns along the same lists.
//   tree  a rank imports the particles of every level-La tree cell within `margin` (>= Rcut) of its TopLeaves: WHOLE cells, so
//         that every node at level >= La a target can reach is complete and identical to the global tree's; cells that are not
//         imported lie more than Rcut from every own target, where the walk discards on geometry alone.  Nodes above level La
//         hold remote particles: they are kept internal (TreeBuilder::force_internal_above) and their moments are the
//         all-reduced sums of the ranks' own particles - the counterpart of the replicated top-tree.  Ghost import replaces the
//         target export: same interaction sets, no return trip, and the walk kernels are the single-GPU ones.
//
// The library links neither MPI nor RCCL; the collectives are the caller's (mpg_comm).  All ordering is on the engine's stream;
// a callback is entered with the stream idle.
#include "engine_internal.h"
#include "peano_tables.h"
#include <algorithm>
#include <chrono>
#include <thread>
#include <rocprim/rocprim.hpp>

namespace {

constexpr int PH_BITS = 21;
const unsigned char H_SUBPIX[MPG_PEANO_NSTATES][8] = MPG_PEANO_SUBPIX;
const unsigned char H_NEXT[MPG_PEANO_NSTATES][8] = MPG_PEANO_NEXT;

struct alignas(16) PRow { // a particle on the wire
    double x, y, z;
    float m;
    unsigned pad;
};
struct alignas(16) RRow { // a PM result on the wire
    double gx, gy, gz, pot;
};
static_assert(sizeof(PRow) == 32 && sizeof(RRow) == 32, "wire rows are 32 bytes");

inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }

__device__ __forceinline__ int wrapi(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }

// slab owner(s) of a particle's CIC cloud: the planes floor(x / cellsize) and that + 1 (periodic), P planes per rank
__device__ __forceinline__ void pm_owners(double x, double cellsize, int nmesh, int P, int &o0, int &o1)
{
    const int ix = (int)floor(x / cellsize);
    o0 = wrapi(ix, nmesh) / P;
    o1 = wrapi(ix + 1, nmesh) / P;
}

// skip (may be null): garbage and swallowed particles take no part in the force (gravpm.c:176-179, forcetree.c:806): they go nowhere
__global__ void __launch_bounds__(256) k_pm_mask(int64_t n, const double *__restrict__ pos, double cellsize, int nmesh, int P,
                                                 const unsigned char *__restrict__ skip, unsigned long long *__restrict__ mask)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    int o0, o1;
    pm_owners(pos[3 * i], cellsize, nmesh, P, o0, o1);
    mask[i] = (skip && skip[i]) ? 0ull : ((1ull << o0) | (1ull << o1));
}

// level-La tree cell of a position by the reference's own floating-point descent (forcetree.c get_subnode, as k_keys of
// tree_build.hip replays it), so that "cell" here is bit for bit the cell of the tree; cells numbered (ix * nc + iy) * nc + iz
__device__ __forceinline__ unsigned tree_cell(double x, double y, double z, double box, int La)
{
    double cx = box / 2., cy = box / 2., cz = box / 2.;
    double len = box * 1.001;
    unsigned ix = 0, iy = 0, iz = 0;
    for(int l = 0; l < La; l++) {
        const double q = 0.25 * len;
        const int bx = x > cx, by = y > cy, bz = z > cz;
        ix = 2 * ix + bx;
        iy = 2 * iy + by;
        iz = 2 * iz + bz;
        cx += bx ? q : -q;
        cy += by ? q : -q;
        cz += bz ? q : -q;
        len *= 0.5;
    }
    return ((ix << La) | iy) << La | iz;
}

__global__ void __launch_bounds__(256) k_need_mask(int64_t n, const double *__restrict__ pos, double box, int La,
                                                   const unsigned long long *__restrict__ need, int me, const unsigned char *__restrict__ skip,
                                                   unsigned long long *__restrict__ mask)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    mask[i] = (skip && skip[i]) ? 0ull : (need[tree_cell(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], box, La)] & ~(1ull << me));
}

// types of the local set [own | ghosts] when the own rows hold garbage: 7 (no bit of any tree mask) for those, 1 otherwise
__global__ void __launch_bounds__(256) k_local_types(int64_t nl, int64_t n_own, const unsigned char *__restrict__ skip, uint8_t *__restrict__ type)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i < nl)
        type[i] = (i < n_own && skip[i]) ? 7 : 1;
}

__global__ void __launch_bounds__(256) k_count_listed(int64_t n, const int *__restrict__ list, const unsigned char *__restrict__ flags,
                                                       unsigned long long *__restrict__ count)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool hit = k < n && flags[list[k]] != 0;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
    if(m != 0ull && (threadIdx.x & 63) == 0)
        atomicAdd(count, (unsigned long long)__builtin_popcountll(m));
}

__global__ void __launch_bounds__(256) k_count_nonzero(int64_t n, const unsigned char *__restrict__ b, unsigned long long *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long m = __ballot(i < n && b[i] != 0);
    if((threadIdx.x & 63) == 0 && m)
        atomicAdd(out, (unsigned long long)__popcll(m));
}

// ---- send lists for all destinations in two passes (a stable multi-way split).  A row may go to several destinations (one bit per
// rank in its mask); within a destination the rows keep their order, so the lists - and with them the order of the ghosts in the
// receivers' arrays - do not depend on scheduling.  Tiles of 1024 consecutive rows: pass 1 counts the rows of every tile per
// destination (tilecnt[d][tile]: an exclusive scan over that array, destinations back to back, IS the start of (d, tile) in the
// concatenated lists) and adds the totals up (one global atomic per tile and destination: per WAVE and destination on the same 8
// words it took 1.76 ms for 16.8 M rows); pass 2 ranks the rows of a tile by ballots and wave totals in LDS and writes the indices.
// (Round 2: one rocprim::select pass over all rows PER DESTINATION.)
constexpr int SPLIT_TILE = 1024;
__global__ void __launch_bounds__(256) k_split_count(int64_t n, const unsigned long long *__restrict__ mask, int nt, int64_t ntiles,
                                                     unsigned *__restrict__ tilecnt, unsigned long long *__restrict__ counts)
{
    __shared__ unsigned s_cnt[64];
    if(threadIdx.x < 64)
        s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * SPLIT_TILE;
    for(int j = 0; j < SPLIT_TILE / 256; j++) {
        const int64_t i = base + j * 256 + threadIdx.x;
        const unsigned long long m = i < n ? mask[i] : 0ull;
        for(int d = 0; d < nt; d++) {
            const unsigned long long b = __builtin_amdgcn_ballot_w64((m >> d) & 1ull);
            if(b && (threadIdx.x & 63) == 0)
                atomicAdd(&s_cnt[d], (unsigned)__popcll(b));
        }
    }
    __syncthreads();
    if(threadIdx.x < nt) {
        const unsigned c = s_cnt[threadIdx.x];
        tilecnt[(int64_t)threadIdx.x * ntiles + blockIdx.x] = c;
        if(c)
            atomicAdd(&counts[threadIdx.x], (unsigned long long)c);
    }
}

__global__ void __launch_bounds__(256) k_split_scatter(int64_t n, const unsigned long long *__restrict__ mask, int nt, int64_t ntiles,
                                                       const unsigned *__restrict__ tilepos, int *__restrict__ idx)
{
    __shared__ unsigned s_base[64], s_w[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if(threadIdx.x < nt)
        s_base[threadIdx.x] = tilepos[(int64_t)threadIdx.x * ntiles + blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * SPLIT_TILE;
    for(int j = 0; j < SPLIT_TILE / 256; j++) {
        const int64_t i = base + j * 256 + threadIdx.x;
        const unsigned long long m = i < n ? mask[i] : 0ull;
        for(int d = 0; d < nt; d++) {
            const unsigned long long b = __builtin_amdgcn_ballot_w64((m >> d) & 1ull);
            if(lane == 0)
                s_w[wave][d] = (unsigned)__popcll(b);
        }
        __syncthreads();
        for(int d = 0; d < nt; d++) {
            const unsigned long long b = __builtin_amdgcn_ballot_w64((m >> d) & 1ull);
            if((m >> d) & 1ull) {
                unsigned pos = s_base[d] + (unsigned)__popcll(b & ((1ull << lane) - 1ull));
                for(int w = 0; w < wave; w++)
                    pos += s_w[w][d];
                idx[pos] = (int)i;
            }
        }
        __syncthreads();
        if(threadIdx.x < nt)
            s_base[threadIdx.x] += s_w[0][threadIdx.x] + s_w[1][threadIdx.x] + s_w[2][threadIdx.x] + s_w[3][threadIdx.x];
        __syncthreads();
    }
}

struct BitOf {
    int d;
    __host__ __device__ bool operator()(const unsigned long long &m) const { return (m >> d) & 1ull; }
};

__global__ void __launch_bounds__(256) k_pack_particles(int64_t ns, const int *__restrict__ idx, const double *__restrict__ pos,
                                                        const float *__restrict__ mass, PRow *__restrict__ rows)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= ns)
        return;
    const int64_t i = idx[k];
    PRow r;
    r.x = pos[3 * i];
    r.y = pos[3 * i + 1];
    r.z = pos[3 * i + 2];
    r.m = mass[i];
    r.pad = 0u;
    rows[k] = r;
}

__global__ void __launch_bounds__(256) k_unpack_particles(int64_t nr, const PRow *__restrict__ rows, double *__restrict__ pos,
                                                          float *__restrict__ mass)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= nr)
        return;
    const PRow r = rows[k];
    pos[3 * k] = r.x;
    pos[3 * k + 1] = r.y;
    pos[3 * k + 2] = r.z;
    mass[k] = r.m;
}

// the particles a slab rank received whose BASE cell lies in its planes: the ones it reads the mesh out for
__global__ void __launch_bounds__(256) k_flag_slab_targets(int64_t n, const double *__restrict__ pos, double cellsize, int nmesh, int P, int me,
                                                           unsigned char *__restrict__ flag)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    int o0, o1;
    pm_owners(pos[3 * i], cellsize, nmesh, P, o0, o1);
    flag[i] = o0 == me;
}

__global__ void __launch_bounds__(256) k_pack_results(int64_t n, const double *__restrict__ gravpm, const double *__restrict__ pot,
                                                      RRow *__restrict__ rows)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= n)
        return;
    RRow r;
    r.gx = gravpm[3 * k];
    r.gy = gravpm[3 * k + 1];
    r.gz = gravpm[3 * k + 2];
    r.pot = pot[k];
    rows[k] = r;
}

// rows come back in the order of the send list; the row from the owner of the particle's BASE cell carries the result
__global__ void __launch_bounds__(256) k_scatter_results(int64_t ns, const int *__restrict__ idx, const RRow *__restrict__ rows,
                                                         const long long *__restrict__ sdsp, int nt, const double *__restrict__ pos,
                                                         double cellsize, int nmesh, int P, double *__restrict__ gravpm,
                                                         double *__restrict__ pot)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= ns)
        return;
    int d = 0;
    while(d + 1 < nt && k >= sdsp[d + 1])
        d++;
    const int64_t i = idx[k];
    int o0, o1;
    pm_owners(pos[3 * i], cellsize, nmesh, P, o0, o1);
    if(o0 != d)
        return;
    const RRow r = rows[k];
    gravpm[3 * i] = r.gx;
    gravpm[3 * i + 1] = r.gy;
    gravpm[3 * i + 2] = r.gz;
    if(pot)
        pot[i] += r.pot; // readout_potential accumulates (gravpm.c:499-501)
}

// flag[list[i]] = 1; an index outside [0, n) raises *err
__global__ void __launch_bounds__(256) k_flag_list(int64_t m, const int *__restrict__ list, int n, uint8_t *__restrict__ flag,
                                                   unsigned *__restrict__ err)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= m)
        return;
    const int j = list[i];
    if(j < 0 || j >= n)
        atomicOr(err, 1u);
    else
        flag[j] = 1;
}

struct IsFlagged {
    const uint8_t *flag;
    __host__ __device__ bool operator()(const int &i) const { return flag[i] != 0; }
};

// own particles of the local tree, in tree order (caller index < n_own)
struct IsOwn {
    int n_own;
    __host__ __device__ bool operator()(const unsigned &ci) const { return (int)ci < n_own; }
};

// number of OWN particles per level-(La-1) cell (sorted keys: mostly one cell per wave)
__global__ void __launch_bounds__(256) k_top_counts(int64_t npart, const uint64_t *__restrict__ keys, const uint32_t *__restrict__ order,
                                                    int64_t n_own, int shift, double *__restrict__ out)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool own = k < npart && (int64_t)order[k] < n_own;
    const unsigned long long cell = own ? (unsigned long long)(keys[k] >> shift) : ~0ull;
    const unsigned long long c0 = __shfl(cell, 0);
    const unsigned long long nown = __popcll(__ballot(own)); // (taken by all lanes: inside the branch below only lane 0 would vote)
    if(__ballot(cell != c0) == 0) {
        if(c0 != ~0ull && (threadIdx.x & 63) == 0)
            unsafeAtomicAdd(&out[c0], (double)nown);
    }
    else if(own)
        unsafeAtomicAdd(&out[cell], 1.0);
}

// ---- SPH columns of the ghosts: 16 words per row before the density loop, 6 after it
struct SphIn { // own arrays (device, may be null)
    const uint8_t *type, *tbh, *tbg;
    const double *hsml, *vel, *entropy, *gacc, *gpm, *hin, *dte;
};

__global__ void __launch_bounds__(256) k_pack_sph_in(int64_t ns, const int *__restrict__ idx, SphIn a, double *__restrict__ rows)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= ns)
        return;
    const int64_t i = idx[k];
    double *r = rows + 16 * k;
    r[0] = a.hsml[i];
    for(int j = 0; j < 3; j++) {
        r[1 + j] = a.vel ? a.vel[3 * i + j] : 0.0;
        r[5 + j] = a.gacc ? a.gacc[3 * i + j] : 0.0;
        r[8 + j] = a.gpm ? a.gpm[3 * i + j] : 0.0;
        r[11 + j] = a.hin ? a.hin[3 * i + j] : 0.0;
    }
    r[4] = a.entropy ? a.entropy[i] : 0.0;
    r[14] = a.dte ? a.dte[i] : 0.0;
    const unsigned long long w = (unsigned long long)(a.type ? a.type[i] : 0) | ((unsigned long long)(a.tbh ? a.tbh[i] : 0) << 8) |
                                 ((unsigned long long)(a.tbg ? a.tbg[i] : 0) << 16);
    r[15] = __longlong_as_double((long long)w);
}

struct SphLocal { // arrays over [own | ghosts]
    uint8_t *type, *tbh, *tbg;
    double *hsml, *vel, *entropy, *gacc, *gpm, *hin, *dte;
};

__global__ void __launch_bounds__(256) k_unpack_sph_in(int64_t nr, const double *__restrict__ rows, SphLocal a, int64_t off)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= nr)
        return;
    const double *r = rows + 16 * k;
    const int64_t i = off + k;
    a.hsml[i] = r[0];
    for(int j = 0; j < 3; j++) {
        a.vel[3 * i + j] = r[1 + j];
        a.gacc[3 * i + j] = r[5 + j];
        a.gpm[3 * i + j] = r[8 + j];
        a.hin[3 * i + j] = r[11 + j];
    }
    a.entropy[i] = r[4];
    a.dte[i] = r[14];
    const unsigned long long w = (unsigned long long)__double_as_longlong(r[15]);
    a.type[i] = (uint8_t)(w & 255);
    a.tbh[i] = (uint8_t)((w >> 8) & 255);
    a.tbg[i] = (uint8_t)((w >> 16) & 255);
}

// own rows of the local inputs (absent optional inputs read as zero)
__global__ void __launch_bounds__(256) k_fill_sph_own(int64_t n, SphIn a, SphLocal l)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    l.hsml[i] = a.hsml[i];
    for(int j = 0; j < 3; j++) {
        l.vel[3 * i + j] = a.vel ? a.vel[3 * i + j] : 0.0;
        l.gacc[3 * i + j] = a.gacc ? a.gacc[3 * i + j] : 0.0;
        l.gpm[3 * i + j] = a.gpm ? a.gpm[3 * i + j] : 0.0;
        l.hin[3 * i + j] = a.hin ? a.hin[3 * i + j] : 0.0;
    }
    l.entropy[i] = a.entropy ? a.entropy[i] : 0.0;
    l.dte[i] = a.dte ? a.dte[i] : 0.0;
    l.type[i] = a.type ? a.type[i] : (uint8_t)0;
    l.tbh[i] = a.tbh ? a.tbh[i] : (uint8_t)0;
    l.tbg[i] = a.tbg ? a.tbg[i] : (uint8_t)0;
}

// the six fields the hydro loop reads of its neighbours, of the own particles that are somebody's ghosts
__global__ void __launch_bounds__(256) k_pack_sph_mid(int64_t ns, const int *__restrict__ idx, const double *hsml, const double *density,
                                                      const double *egy, const double *dhsml, const double *divvel, const double *curlvel,
                                                      double *__restrict__ rows)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= ns)
        return;
    const int64_t i = idx[k];
    double *r = rows + 6 * k;
    r[0] = hsml[i];
    r[1] = density[i];
    r[2] = egy ? egy[i] : 0.0;
    r[3] = dhsml[i];
    r[4] = divvel[i];
    r[5] = curlvel[i];
}

__global__ void __launch_bounds__(256) k_unpack_sph_mid(int64_t nr, const double *__restrict__ rows, int64_t off, double *hsml, double *density,
                                                        double *egy, double *dhsml, double *divvel, double *curlvel)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= nr)
        return;
    const double *r = rows + 6 * k;
    const int64_t i = off + k;
    hsml[i] = r[0];
    density[i] = r[1];
    if(egy)
        egy[i] = r[2];
    dhsml[i] = r[3];
    divvel[i] = r[4];
    curlvel[i] = r[5];
}

// density_haswork (density.c:521-530): gas and black holes (bh = 1; swallowed ones carry type 7 here, like garbage); hydro_haswork: gas
struct IsOwnGas {
    const uint8_t *type;
    int bh;
    __host__ __device__ bool operator()(const int &i) const { return type[i] == 0 || (bh && type[i] == 5); }
};
struct IsActiveGas {
    const uint8_t *type, *flag;
    int bh;
    __host__ __device__ bool operator()(const int &i) const { return (type[i] == 0 || (bh && type[i] == 5)) && flag[i] != 0; }
};

__global__ void __launch_bounds__(256) k_max_gas_hsml(int64_t n, const uint8_t *__restrict__ type, const double *__restrict__ hsml, const int bh,
                                                      unsigned long long *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double h = (i < n && (type[i] == 0 || (bh && type[i] == 5))) ? hsml[i] : 0.0;
    for(int off = 32; off > 0; off >>= 1)
        h = fmax(h, __shfl_down(h, off));
    if((threadIdx.x & 63) == 0 && h > 0)
        atomicMax(out, (unsigned long long)__double_as_longlong(h)); // (positive doubles order like their bit patterns)
}

// ---- domain decomposition: destination masks, generic columns on the wire, cost per TopLeaf
__global__ void __launch_bounds__(256) k_task_mask(int64_t n, const int *__restrict__ task, unsigned long long *__restrict__ mask)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        mask[i] = task[i] >= 0 ? (1ull << task[i]) : 0ull; // garbage (-1) goes nowhere: domain_exchange drops it
}

struct Cols {
    const char *src[16];
    char *dst[16];
    int w[16], off[16]; // bytes of a column, its offset in the wire row (8-byte aligned)
    int n, row;
};

__device__ __forceinline__ void copy_bytes(char *d, const char *s, int w)
{
    if((w & 7) == 0)
        for(int k = 0; k < w; k += 8)
            *(unsigned long long *)(d + k) = *(const unsigned long long *)(s + k);
    else if((w & 3) == 0)
        for(int k = 0; k < w; k += 4)
            *(unsigned *)(d + k) = *(const unsigned *)(s + k);
    else
        for(int k = 0; k < w; k++)
            d[k] = s[k];
}

__global__ void __launch_bounds__(256) k_pack_cols(int64_t ns, const int *__restrict__ idx, Cols c, char *__restrict__ rows)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= ns)
        return;
    const int64_t i = idx[k];
    for(int j = 0; j < c.n; j++)
        copy_bytes(rows + k * c.row + c.off[j], c.src[j] + i * c.w[j], c.w[j]);
}

__global__ void __launch_bounds__(256) k_unpack_cols(int64_t nr, const char *__restrict__ rows, Cols c)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= nr)
        return;
    for(int j = 0; j < c.n; j++)
        copy_bytes(c.dst[j] + k * c.w[j], rows + k * c.row + c.off[j], c.w[j]);
}

// work per TopLeaf: block-local sums in LDS, one atomic per block and occupied leaf
__global__ void __launch_bounds__(256) k_leaf_cost(int64_t n, const int *__restrict__ topleaf, const float *__restrict__ cost, int nleaves,
                                                   double *__restrict__ out)
{
    extern __shared__ double s_sum[];
    for(int l = threadIdx.x; l < nleaves; l += blockDim.x)
        s_sum[l] = 0;
    __syncthreads();
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int l = topleaf[i];
        if(l >= 0 && l < nleaves)
            atomicAdd(&s_sum[l], (double)cost[i]);
    }
    __syncthreads();
    for(int l = threadIdx.x; l < nleaves; l += blockDim.x)
        if(s_sum[l] != 0)
            unsafeAtomicAdd(&out[l], s_sum[l]);
}

// ---- friends-of-friends over ranks: ghost columns, label exchange, label maps, group numbers
struct alignas(16) IdRow {
    unsigned long long id;
    unsigned long long type;
};

__global__ void __launch_bounds__(256) k_pack_idtype(int64_t ns, const int *__restrict__ idx, const unsigned long long *__restrict__ id,
                                                     const uint8_t *__restrict__ type, IdRow *__restrict__ rows)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= ns)
        return;
    const int64_t i = idx[k];
    rows[k].id = id[i];
    rows[k].type = type ? type[i] : 1ull;
}

__global__ void __launch_bounds__(256) k_unpack_idtype(int64_t nr, const IdRow *__restrict__ rows, unsigned long long *__restrict__ id,
                                                       uint8_t *__restrict__ type)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= nr)
        return;
    id[k] = rows[k].id;
    type[k] = (uint8_t)rows[k].type;
}

__global__ void __launch_bounds__(256) k_fill_type(int64_t n, const uint8_t *__restrict__ type, uint8_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        out[i] = type ? type[i] : (uint8_t)1;
}

__global__ void __launch_bounds__(256) k_gather_u64(int64_t ns, const int *__restrict__ idx, const unsigned long long *__restrict__ src,
                                                    unsigned long long *__restrict__ out)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k < ns)
        out[k] = src[idx[k]];
}

// a ghost whose owner knows a smaller label than this rank's component of it: (this rank's label, the owner's) is a relabelling
__global__ void __launch_bounds__(256) k_label_pairs(int64_t nr, const unsigned long long *__restrict__ mine, const unsigned long long *__restrict__ ext,
                                                     unsigned long long *__restrict__ keys, unsigned long long *__restrict__ vals,
                                                     uint8_t *__restrict__ flag)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= nr)
        return;
    keys[k] = mine[k];
    vals[k] = ext[k];
    flag[k] = ext[k] < mine[k];
}

// label[i] <- map(label[i]) where the (sorted, unique) keys hold it
__global__ void __launch_bounds__(256) k_apply_label_map(int64_t n, unsigned long long *__restrict__ label, const unsigned long long *__restrict__ keys,
                                                         const unsigned long long *__restrict__ vals, int64_t m)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    const unsigned long long l = label[i];
    int64_t lo = 0, hi = m;
    while(lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if(keys[mid] < l)
            lo = mid + 1;
        else
            hi = mid;
    }
    if(lo < m && keys[lo] == l)
        label[i] = vals[lo];
}

// P[].GrNr from the table of all groups (MinID ascending)
__global__ void __launch_bounds__(256) k_lookup_grnr(int64_t n, const unsigned long long *__restrict__ label, const unsigned long long *__restrict__ minid,
                                                     const long long *__restrict__ grnr, int64_t ng, long long *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    const unsigned long long l = label[i];
    int64_t lo = 0, hi = ng;
    while(lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if(minid[mid] < l)
            lo = mid + 1;
        else
            hi = mid;
    }
    out[i] = (lo < ng && minid[lo] == l) ? grnr[lo] : -1ll;
}

struct Plan { // one personalised exchange: who gets which of my rows, and what I get
    DevBuf<int> idx;
    DevBuf<long long> d_sdsp;
    HostBuf<long long> h_sdsp;
    std::vector<int64_t> scnt, rcnt, sdsp, rdsp; // rows
    int64_t nsend = 0, nrecv = 0;
};

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

} // namespace

struct mpg_dist {
    mpg_engine *eng = nullptr;
    mpg_comm comm{};
    int me = 0, nt = 1;
    // the communicator runs device-buffer collectives as work on the engine's stream (mpg_comm.bind_stream: RCCL): no host
    // synchronisation before or after them
    bool stream_ordered = false;
    hipStream_t bound_stream = nullptr;
    // domain
    double box = 0, margin = 0;
    int La = 0;
    bool have_domain = false;
    DevBuf<unsigned long long> need; // [8^La] ranks that need the cell
    // work
    DevBuf<unsigned long long> mask, cnt;
    DevBuf<unsigned> tilecnt, tilepos; // build_plan: rows per (destination, tile of 1024 rows) and their exclusive scan
    DevBuf<char> tmp;
    DevBuf<char> sendbuf, recvbuf;
    HostBuf<char> hsend, hrecv;
    Plan pm, ghost;
    // PM side
    DevBuf<double> spos, sgrav, spot;
    DevBuf<float> smass;
    DevBuf<unsigned char> sflag;
    DevBuf<int> starg;
    DevBuf<unsigned long long> scount;
    DevBuf<double> sendA, recvA, sendB, recvB, gsend, grecv;
    int64_t per_peer = 0, plane = 0;
    bool slab_ready = false;
    // tree side
    hipStream_t tree_stream = nullptr; // the local tree build beside the PM step (mpg_dist_gravity_step)
    DevBuf<double> lpos, top;
    DevBuf<float> lmass;
    DevBuf<int> targets, act_targets;
    DevBuf<uint8_t> actflag;
    DevBuf<unsigned> err;
    HostBuf<double> htop;
    int64_t ntarg = 0, n_own_tree = -1;
    // garbage / swallowed particles among the own rows (mpg_dist_dev_set_garbage): flags over n_skip_rows rows, n_skip of them set
    const unsigned char *d_skip = nullptr;
    DevBuf<unsigned char> skip_own; // (the library's copy of the caller's flags)
    int64_t n_skip = 0, n_skip_rows = -1;
    DevBuf<uint8_t> ltype, o_skip;
    const unsigned char *skip_for(int64_t n) const { return (n_skip > 0 && n == n_skip_rows) ? d_skip : nullptr; }
    HostBuf<unsigned long long> chk; // [0] own particles found in the local tree, [1] the two flags of the global top (pinned: read back without a wait)
    bool chk_pending = false;
    bool grav_tree_valid = false; // the engine's tree is the gravity tree of mpg_dist_dev_force_tree_build (the SPH loops and FOF replace it)
    double last_hmax = 0;         // largest smoothing length over all ranks after the last density loop
    int blackholes = 0;           // BlackHoleOn of density(): the own non-swallowed black holes are targets of the density loop too
    int64_t dom_max_part = 0;     // PartManager->MaxPart of domain_check_memory_bound (0: no bound)
    DevBuf<float> cost;
    // SPH loops on the local set: inputs and outputs over [own | ghosts]
    DevBuf<uint8_t> s_type, s_tbh, s_tbg;
    DevBuf<double> s_in[7];   // hsml, vel, entropy, gacc, gpm, hydroacc_in, dtentropy_in
    DevBuf<double> s_out[11]; // dthsml, density, egywtdensity, dhsmlegyfac, divvel, curlvel, gradrho, hydroacc_out, dtentropy_out, maxsignalvel
    DevBuf<int> gas;
    int64_t ngas = 0, sph_nl = -1, sph_n_own = -1;
    // domain decomposition (mpg_dist_domain_decompose): the global tree, its leaves, where every particle goes
    std::vector<mpg_topnode> dom_tree;
    std::vector<int> dom_leaf_task, dom_leaf_topnode, dom_start, dom_end;
    std::vector<int64_t> dom_leaf_count, dom_send_counts;
    int dom_size = 0, dom_nleaves = 0, dom_policy = 0;
    double dom_alloc_factor = 0.5;
    DevBuf<int> dom_topleaf, dom_task;
    DevBuf<double> dom_cost;
    int64_t dom_n = -1;
    Plan dom_plan;
    DevBuf<char> dom_out[16];
    // friends-of-friends over ranks
    DevBuf<unsigned long long> f_id, f_ext, f_keys, f_vals, f_keys2, f_vals2, f_ukeys, f_uvals, f_glab, f_minid;
    DevBuf<uint8_t> f_type, f_flag;
    DevBuf<long long> f_grnr_tab, f_pgrnr;
    struct GroupRec { // one (part of a) group: struct BaseGroup / Group (fof.h:14-49) with raw sums about FirstPos
        unsigned long long MinID;
        long long Length;
        int LenType[6];
        float FirstPos[3];
        int src;
        double acc[27]; // Mass, MassType[6], sum m x[3], sum m v[3], sum m (rel x v)[3], sum m rel rel^T [9]
    };
    std::vector<GroupRec> f_groups; // the complete groups this rank holds (MinID % NTask == ThisTask), finished
    std::vector<long long> f_group_grnr;
    int64_t f_total = 0;
    int64_t stats[8] = {};
    double times[8] = {};
    // host (drop-in) path: the rank's P[] staged on the device
    DevBuf<double> o_pos, o_gravpm, o_pot, o_acc, o_prev;
    DevBuf<float> o_mass;
    std::vector<double> hbuf;
    std::vector<float> hbuf_f;
    std::vector<uint8_t> hbuf_b;
    int64_t o_n = -1;
    DevBuf<int> o_act;        // ActiveParticle of the host drop-in walk
    DevBuf<double> o_sph[17]; // the double-valued fields of mpg_sph_arrays over the own particles (host SPH path)
    DevBuf<uint8_t> o_u8[3];  // type, tb_hydro, tb_grav
};

namespace {

void sync(mpg_dist *d) { MPG_HIP(hipStreamSynchronize(d->eng->stream)); }

void cb(int rc, const char *what) { MPG_CHECK(rc == 0, std::string("mpg_comm callback failed: ") + what); }

// a stream-ordered communicator follows the engine's stream (mpg_use_stream may have replaced it since mpg_dist_create)
void follow_stream(mpg_dist *d)
{
    if(d->stream_ordered && d->bound_stream != d->eng->stream) {
        cb(d->comm.bind_stream(d->comm.ctx, (void *)d->eng->stream), "bind_stream");
        d->bound_stream = d->eng->stream;
    }
}

// alltoallv of bytes between device buffers (staged through pinned host memory unless the communicator takes device pointers)
void a2av(mpg_dist *d, const void *dsend, const std::vector<int64_t> &sb, const std::vector<int64_t> &sd, void *drecv,
          const std::vector<int64_t> &rb, const std::vector<int64_t> &rd, int64_t stot, int64_t rtot)
{
    hipStream_t st = d->eng->stream;
    d->stats[4] += stot;
    if(d->nt == 1 && !d->comm.alltoallv) {
        if(stot > 0)
            MPG_HIP(hipMemcpyAsync(drecv, dsend, (size_t)stot, hipMemcpyDeviceToDevice, st));
        return;
    }
    MPG_CHECK(d->comm.alltoallv, "mpg_comm: alltoallv callback missing");
    if(d->comm.device_buffers) {
        follow_stream(d);
        if(!d->stream_ordered)
            sync(d); // (a blocking communicator works outside the stream: what it sends must be complete)
        cb(d->comm.alltoallv(d->comm.ctx, dsend, sb.data(), sd.data(), drecv, rb.data(), rd.data(), 1), "alltoallv");
        return;
    }
    d->hsend.reserve((size_t)stot + 16);
    d->hrecv.reserve((size_t)rtot + 16);
    if(stot > 0)
        MPG_HIP(hipMemcpyAsync(d->hsend.p, dsend, (size_t)stot, hipMemcpyDeviceToHost, st));
    sync(d);
    cb(d->comm.alltoallv(d->comm.ctx, d->hsend.p, sb.data(), sd.data(), d->hrecv.p, rb.data(), rd.data(), 0), "alltoallv");
    if(rtot > 0)
        MPG_HIP(hipMemcpyAsync(drecv, d->hrecv.p, (size_t)rtot, hipMemcpyHostToDevice, st));
    sync(d); // (hrecv is reused by the next exchange)
}

// uniform all-to-all of `bytes_per_peer` per rank pair (the transposes of the slab PM)
void a2a_uniform(mpg_dist *d, const void *dsend, void *drecv, int64_t bytes_per_peer)
{
    std::vector<int64_t> b((size_t)d->nt, bytes_per_peer), dsp((size_t)d->nt);
    for(int r = 0; r < d->nt; r++)
        dsp[r] = (int64_t)r * bytes_per_peer;
    a2av(d, dsend, b, dsp, drecv, b, dsp, bytes_per_peer * d->nt, bytes_per_peer * d->nt);
    d->stats[4] -= bytes_per_peer * d->nt;
    d->stats[5] += bytes_per_peer * d->nt;
}

void allreduce_host_f64(mpg_dist *d, double *h, int64_t n, int op)
{
    if(d->nt == 1 && !d->comm.allreduce)
        return;
    MPG_CHECK(d->comm.allreduce, "mpg_comm: allreduce callback missing");
    cb(d->comm.allreduce(d->comm.ctx, h, n, 0, op, 0), "allreduce");
}

// send lists from per-particle destination masks; the counts travel through alltoall_i64
void build_plan(mpg_dist *d, Plan &pl, int64_t n, const unsigned long long *mask)
{
    hipStream_t st = d->eng->stream;
    const int nt = d->nt;
    d->cnt.reserve(64);
    MPG_HIP(hipMemsetAsync(d->cnt.p, 0, 64 * sizeof(unsigned long long), st));
    const int64_t ntiles = (n + SPLIT_TILE - 1) / SPLIT_TILE;
    d->tilecnt.reserve((size_t)nt * ntiles + 1);
    d->tilepos.reserve((size_t)nt * ntiles + 1);
    if(n > 0)
        hipLaunchKernelGGL(k_split_count, dim3((unsigned)ntiles), dim3(256), 0, st, n, mask, nt, ntiles, d->tilecnt.p, d->cnt.p);
    unsigned long long hc[64];
    MPG_HIP(hipMemcpyAsync(hc, d->cnt.p, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    if(n > 0) { // (queued behind the count pass, before the host looks at the totals)
        size_t tb = 0;
        MPG_HIP(rocprim::exclusive_scan(nullptr, tb, d->tilecnt.p, d->tilepos.p, 0u, (size_t)nt * ntiles, rocprim::plus<unsigned>(), st));
        d->tmp.reserve(tb + 16);
        MPG_HIP(rocprim::exclusive_scan((void *)d->tmp.p, tb, d->tilecnt.p, d->tilepos.p, 0u, (size_t)nt * ntiles, rocprim::plus<unsigned>(), st));
    }
    sync(d);
    pl.scnt.assign(nt, 0);
    pl.sdsp.assign(nt + 1, 0);
    for(int r = 0; r < nt; r++) {
        pl.scnt[r] = (int64_t)hc[r];
        pl.sdsp[r + 1] = pl.sdsp[r] + pl.scnt[r];
    }
    pl.nsend = pl.sdsp[nt];
    MPG_CHECK(pl.nsend < (1ll << 31), "mpg_dist: more than 2^31 rows in one exchange");
    pl.idx.reserve((size_t)pl.nsend + 1);
    pl.d_sdsp.reserve((size_t)nt + 1);
    pl.h_sdsp.reserve((size_t)nt + 1); // (pinned, owned by the plan: the copy needs no wait; the next build_plan of this plan starts with one)
    for(int r = 0; r <= nt; r++)
        pl.h_sdsp.p[r] = (long long)pl.sdsp[r];
    MPG_HIP(hipMemcpyAsync(pl.d_sdsp.p, pl.h_sdsp.p, (nt + 1) * sizeof(long long), hipMemcpyHostToDevice, st));
    if(n > 0 && pl.nsend > 0)
        hipLaunchKernelGGL(k_split_scatter, dim3((unsigned)ntiles), dim3(256), 0, st, n, mask, nt, ntiles, d->tilepos.p, pl.idx.p);
    pl.rcnt.assign(nt, 0);
    if(nt == 1 && !d->comm.alltoall_i64)
        pl.rcnt[0] = pl.scnt[0];
    else {
        MPG_CHECK(d->comm.alltoall_i64, "mpg_comm: alltoall_i64 callback missing");
        cb(d->comm.alltoall_i64(d->comm.ctx, pl.scnt.data(), pl.rcnt.data()), "alltoall_i64");
    }
    pl.rdsp.assign(nt + 1, 0);
    for(int r = 0; r < nt; r++)
        pl.rdsp[r + 1] = pl.rdsp[r] + pl.rcnt[r];
    pl.nrecv = pl.rdsp[nt];
}

// rows of `rb_` bytes each along a plan; reverse: the answers travel back (what I received, I return; what I sent, I get back)
void exchange_rows(mpg_dist *d, const Plan &pl, const void *dsend, void *drecv, bool reverse, int64_t rb_)
{
    const int nt = d->nt;
    std::vector<int64_t> sb(nt), sd(nt), rb(nt), rd(nt);
    for(int r = 0; r < nt; r++) {
        sb[r] = rb_ * (reverse ? pl.rcnt[r] : pl.scnt[r]);
        sd[r] = rb_ * (reverse ? pl.rdsp[r] : pl.sdsp[r]);
        rb[r] = rb_ * (reverse ? pl.scnt[r] : pl.rcnt[r]);
        rd[r] = rb_ * (reverse ? pl.sdsp[r] : pl.rdsp[r]);
    }
    a2av(d, dsend, sb, sd, drecv, rb, rd, rb_ * (reverse ? pl.nrecv : pl.nsend), rb_ * (reverse ? pl.nsend : pl.nrecv));
}
void exchange_rows32(mpg_dist *d, const Plan &pl, const void *dsend, void *drecv, bool reverse) { exchange_rows(d, pl, dsend, drecv, reverse, 32); }

// cube (integer coordinates at its level) of a TopNode: the prefix of its Peano-Hilbert key walked through the curve's state
// machine backwards (digit -> octant)
void topnode_cube(const mpg_topnode &tn, int &level, unsigned &x, unsigned &y, unsigned &z)
{
    level = (3 * PH_BITS - tn.Shift) / 3;
    x = y = z = 0;
    unsigned state = 0;
    for(int l = 0; l < level; l++) {
        const unsigned digit = (unsigned)((tn.StartKey >> (3 * (PH_BITS - 1 - l))) & 7ull);
        unsigned pix = 0;
        while(pix < 8 && H_SUBPIX[state][pix] != digit)
            pix++;
        MPG_CHECK(pix < 8, "mpg_dist_set_domain: corrupt Peano-Hilbert key");
        x = 2 * x + ((pix >> 2) & 1u);
        y = 2 * y + ((pix >> 1) & 1u);
        z = 2 * z + (pix & 1u);
        state = H_NEXT[state][pix];
    }
}

// cells (of nc across the 1.001-Box root cell starting at lo0) that the interval [a, b] touches, periodic with period Box
void cells_of_interval(double a, double b, double box, double lo0, double w, int nc, std::vector<char> &sel)
{
    sel.assign((size_t)nc, 0);
    if(b - a >= box) {
        std::fill(sel.begin(), sel.end(), 1);
        return;
    }
    for(int k = -1; k <= 1; k++) {
        const double aa = a + k * box, bb = b + k * box;
        int k0 = (int)floor((aa - lo0) / w), k1 = (int)floor((bb - lo0) / w);
        if(k1 < 0 || k0 >= nc)
            continue;
        k0 = std::max(k0, 0);
        k1 = std::min(k1, nc - 1);
        for(int c = k0; c <= k1; c++)
            sel[(size_t)c] = 1;
    }
}

void pm_slab_setup(mpg_dist *d)
{
    mpg_engine *e = d->eng;
    MPG_CHECK(e->pm.nmesh > 0, "mpg_dist: gravpm_init_periodic first");
    MPG_CHECK(e->pm.nmesh % d->nt == 0, "mpg_dist: Nmesh must be a multiple of the number of ranks");
    e->pm.slab_init(d->me, d->nt);
    d->per_peer = (int64_t)e->pm.slab_cplx_per_peer();
    d->plane = (int64_t)e->pm.nmesh * e->pm.nmesh;
    const size_t tr = (size_t)2 * d->per_peer * d->nt;
    d->sendA.reserve(tr);
    d->recvA.reserve(tr);
    d->sendB.reserve(tr);
    d->recvB.reserve(tr);
    d->gsend.reserve((size_t)5 * d->plane);
    d->grecv.reserve((size_t)5 * d->plane);
    d->slab_ready = true;
}

// gravpm_force over the ranks: ship particles to their slabs, solve, return the results
void pm_step(mpg_dist *d, int64_t n, const double *pos, const float *mass, double *gravpm, double *pot)
{
    mpg_engine *e = d->eng;
    hipStream_t st = e->stream;
    PMesh &pm = e->pm;
    if(!d->slab_ready || pm.slab.rank != d->me || pm.slab.world != d->nt || !pm.slab.ready)
        pm_slab_setup(d);
    const int nmesh = pm.nmesh, P = nmesh / d->nt;
    d->mask.reserve((size_t)n + 1);
    if(n > 0)
        hipLaunchKernelGGL(k_pm_mask, dim3(nblk(n)), dim3(256), 0, st, n, pos, pm.cellsize, nmesh, P, d->skip_for(n), d->mask.p);
    build_plan(d, d->pm, n, d->mask.p);
    Plan &pl = d->pm;
    d->sendbuf.reserve((size_t)32 * pl.nsend + 32);
    d->recvbuf.reserve((size_t)32 * pl.nrecv + 32);
    if(pl.nsend > 0)
        hipLaunchKernelGGL(k_pack_particles, dim3(nblk(pl.nsend)), dim3(256), 0, st, pl.nsend, pl.idx.p, pos, mass, (PRow *)d->sendbuf.p);
    exchange_rows32(d, pl, d->sendbuf.p, d->recvbuf.p, false);
    const int64_t nr = pl.nrecv;
    d->spos.reserve((size_t)3 * nr + 3);
    d->smass.reserve((size_t)nr + 1);
    d->sgrav.reserve((size_t)3 * nr + 3);
    d->spot.reserve((size_t)nr + 1);
    d->sflag.reserve((size_t)nr + 1);
    d->starg.reserve((size_t)nr + 1);
    d->scount.reserve(4);
    if(nr > 0)
        hipLaunchKernelGGL(k_unpack_particles, dim3(nblk(nr)), dim3(256), 0, st, nr, (const PRow *)d->recvbuf.p, d->spos.p, d->smass.p);
    // local stages of the slab solver with the two transposes and the neighbour planes in between (pm.hip, "slab-decomposed form")
    pm.slab_forward_a(nr, d->spos.p, d->smass.p, d->sendA.p, st);
    a2a_uniform(d, d->sendA.p, d->recvA.p, 16 * d->per_peer);
    pm.slab_forward_b(d->recvA.p, d->sendB.p, st);
    a2a_uniform(d, d->sendB.p, d->recvB.p, 16 * d->per_peer);
    pm.slab_inverse_c(d->recvB.p, d->gsend.p, st);
    {
        // the first 3 planes go to the previous rank (its upper ghosts), the last 2 to the next rank (its lower ghosts);
        // received: from the next rank its first 3, from the previous rank its last 2.  One alltoallv; with 1 or 2 ranks both
        // neighbours are the same rank, which then gets [first 3 | last 2] in one block and sends the same
        const int nt = d->nt, prev = (d->me + nt - 1) % nt, next = (d->me + 1) % nt;
        const int64_t pb = d->plane * 8;
        std::vector<int64_t> sb(nt, 0), sd(nt, 0), rb(nt, 0), rd(nt, 0);
        if(prev == next) { // (nt <= 2)
            sb[prev] = 5 * pb;
            rb[prev] = 5 * pb;
        }
        else {
            sb[prev] = 3 * pb;
            sd[prev] = 0;
            sb[next] = 2 * pb;
            sd[next] = 3 * pb;
            rb[next] = 3 * pb;
            rd[next] = 0;
            rb[prev] = 2 * pb;
            rd[prev] = 3 * pb;
        }
        a2av(d, d->gsend.p, sb, sd, d->grecv.p, rb, rd, 5 * pb, 5 * pb);
    }
    // the received particles whose base cell is mine are read out (the kernel tells: no target list, no count read back - round 4)
    if(nr > 0) {
        MPG_HIP(hipMemsetAsync(d->sgrav.p, 0, (size_t)3 * nr * sizeof(double), st));
        MPG_HIP(hipMemsetAsync(d->spot.p, 0, (size_t)nr * sizeof(double), st));
    }
    static const bool stencil_rows = !(getenv("MPG_PM_STENCIL") && getenv("MPG_PM_STENCIL")[0] == '0');
    if(stencil_rows)
        pm.slab_readout_rows(d->grecv.p, nr, d->spos.p, d->sgrav.p, d->spot.p, st);
    else {
        int64_t ntarg = 0;
        if(nr > 0) {
            hipLaunchKernelGGL(k_flag_slab_targets, dim3(nblk(nr)), dim3(256), 0, st, nr, d->spos.p, pm.cellsize, nmesh, P, d->me, d->sflag.p);
            rocprim::counting_iterator<int> iota(0);
            size_t tb = 0;
            MPG_HIP(rocprim::select(nullptr, tb, iota, d->sflag.p, d->starg.p, d->scount.p, (size_t)nr, st));
            d->tmp.reserve(tb + 16);
            MPG_HIP(rocprim::select((void *)d->tmp.p, tb, iota, d->sflag.p, d->starg.p, d->scount.p, (size_t)nr, st));
            unsigned long long c = 0;
            MPG_HIP(hipMemcpyAsync(&c, d->scount.p, sizeof(c), hipMemcpyDeviceToHost, st));
            sync(d);
            ntarg = (int64_t)c;
        }
        pm.slab_readout(d->grecv.p, d->starg.p, ntarg, d->spos.p, d->sgrav.p, d->spot.p, st);
    }
    // results back along the same lists
    if(nr > 0)
        hipLaunchKernelGGL(k_pack_results, dim3(nblk(nr)), dim3(256), 0, st, nr, d->sgrav.p, d->spot.p, (RRow *)d->recvbuf.p);
    exchange_rows32(d, pl, d->recvbuf.p, d->sendbuf.p, true);
    if(pl.nsend > 0)
        hipLaunchKernelGGL(k_scatter_results, dim3(nblk(pl.nsend)), dim3(256), 0, st, pl.nsend, pl.idx.p, (const RRow *)d->sendbuf.p, pl.d_sdsp.p, d->nt,
                           pos, pm.cellsize, nmesh, P, gravpm, pot);
    MPG_HIP(hipGetLastError());
    d->stats[1] = pl.nsend;
}

// own particles followed by the ghosts of every needed cell: d->lpos / d->lmass; returns the local count
int64_t import_ghosts(mpg_dist *d, int64_t n, const double *pos, const float *mass)
{
    hipStream_t st = d->eng->stream;
    d->mask.reserve((size_t)n + 1);
    if(n > 0)
        hipLaunchKernelGGL(k_need_mask, dim3(nblk(n)), dim3(256), 0, st, n, pos, d->box, d->La, d->need.p, d->me, d->skip_for(n), d->mask.p);
    build_plan(d, d->ghost, n, d->mask.p);
    Plan &pl = d->ghost;
    d->sendbuf.reserve((size_t)32 * pl.nsend + 32);
    d->recvbuf.reserve((size_t)32 * pl.nrecv + 32);
    if(pl.nsend > 0)
        hipLaunchKernelGGL(k_pack_particles, dim3(nblk(pl.nsend)), dim3(256), 0, st, pl.nsend, pl.idx.p, pos, mass, (PRow *)d->sendbuf.p);
    exchange_rows32(d, pl, d->sendbuf.p, d->recvbuf.p, false);
    const int64_t nl = n + pl.nrecv;
    d->lpos.reserve((size_t)3 * nl + 3);
    d->lmass.reserve((size_t)nl + 1);
    if(n > 0) {
        MPG_HIP(hipMemcpyAsync(d->lpos.p, pos, (size_t)3 * n * sizeof(double), hipMemcpyDeviceToDevice, st));
        MPG_HIP(hipMemcpyAsync(d->lmass.p, mass, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    if(pl.nrecv > 0)
        hipLaunchKernelGGL(k_unpack_particles, dim3(nblk(pl.nrecv)), dim3(256), 0, st, pl.nrecv, (const PRow *)d->recvbuf.p, d->lpos.p + 3 * n,
                           d->lmass.p + n);
    MPG_HIP(hipGetLastError());
    d->stats[0] = pl.nrecv;
    d->stats[2] = nl;
    return nl;
}

// moments of the nodes above level La from the sums over all ranks (tree_build.hip, "the top of the tree from global sums")
// levels La-2 .. 0 of the global top from the all-reduced sums of level La-1 (4 doubles per cell, children consecutive in octant-path
// numbering), laid out level 0 first as k_top_set reads them; the all-reduced particle counts are checked on the way: a cell above
// the decomposition level is internal on every rank whatever it holds locally, and the global tree agrees only if it holds more than
// NMAXCHILD particles in all.  One block: the levels depend on each other and the whole top is a few thousand cells.
__global__ void __launch_bounds__(1024) k_top_levels(int La, const double *__restrict__ fine, const double *__restrict__ cnt, double *__restrict__ all,
                                                     unsigned *__restrict__ err)
{
    const size_t nfine = (size_t)1 << (3 * (La - 1));
    size_t off = 0;
    for(int l = 0; l < La - 1; l++)
        off += (size_t)1 << (3 * l);
    for(size_t c = threadIdx.x; c < nfine; c += blockDim.x) {
        const double k = cnt[c];
        if(!(k == 0 || k > NMAXCHILD))
            atomicOr(err, 1u);
        for(int j = 0; j < 4; j++)
            all[(off + c) * 4 + j] = fine[c * 4 + j];
    }
    for(int l = La - 2; l >= 0; l--) {
        __syncthreads();
        const size_t nc = (size_t)1 << (3 * l), coff = off;
        off -= nc;
        for(size_t c = threadIdx.x; c < nc; c += blockDim.x)
            for(int j = 0; j < 4; j++) {
                double sum = 0.0; // (children added in octant order, as the host loop of rounds 2-3 did: same bits)
                for(int k = 0; k < 8; k++)
                    sum += all[(coff + 8 * c + k) * 4 + j];
                all[(off + c) * 4 + j] = sum;
            }
    }
}

void global_top(mpg_dist *d, int64_t n_own)
{
    mpg_engine *e = d->eng;
    hipStream_t st = e->stream;
    const int La = d->La;
    const size_t nfine = (size_t)1 << (3 * (La - 1));
    size_t ntot = 0;
    for(int l = 0; l < La; l++)
        ntot += (size_t)1 << (3 * l);
    // [nfine * 4 sums | nfine counts | ntot * 4 levels]
    d->top.reserve(nfine * 5 + ntot * 4 + 8);
    e->tree.top_partial(La, n_own, d->top.p, st);
    double *cnt = d->top.p + nfine * 4, *all = d->top.p + nfine * 5;
    MPG_HIP(hipMemsetAsync(cnt, 0, nfine * sizeof(double), st));
    if(e->tree.npart > 0)
        hipLaunchKernelGGL(k_top_counts, dim3(nblk(e->tree.npart)), dim3(256), 0, st, e->tree.npart, e->tree.keys_b.p, e->tree.idx_b.p, n_own,
                           3 * (MAXLEVEL - (La - 1)), cnt);
    follow_stream(d);
    if(d->stream_ordered) // (RCCL: the all-reduce is the next piece of work on the stream)
        cb(d->comm.allreduce(d->comm.ctx, d->top.p, (int64_t)(nfine * 5), 0, 0, 1), "allreduce");
    else if(!(d->nt == 1 && !d->comm.allreduce)) {
        d->htop.reserve(nfine * 5 + 8);
        MPG_HIP(hipMemcpyAsync(d->htop.p, d->top.p, nfine * 5 * sizeof(double), hipMemcpyDeviceToHost, st));
        sync(d);
        allreduce_host_f64(d, d->htop.p, (int64_t)(nfine * 5), 0);
        MPG_HIP(hipMemcpyAsync(d->top.p, d->htop.p, nfine * 5 * sizeof(double), hipMemcpyHostToDevice, st));
    }
    // the checks of this phase raise bits of d->err[1..2]; tree_finish reads them with the target count (no wait here)
    d->err.reserve(8);
    MPG_HIP(hipMemsetAsync(d->err.p + 1, 0, 2 * sizeof(unsigned), st));
    hipLaunchKernelGGL(k_top_levels, dim3(1), dim3(1024), 0, st, La, (const double *)d->top.p, (const double *)cnt, all, d->err.p + 1);
    e->tree.top_set(La, all, st, (int *)(d->err.p + 2));
}

} // namespace

extern "C" {

int mpg_dist_create(mpg_dist **out, mpg_engine *eng, const mpg_comm *comm)
{
    API_BEGIN
    MPG_CHECK(out && eng && comm, "null argument");
    MPG_CHECK(comm->NTask >= 1 && comm->NTask <= 64 && comm->ThisTask >= 0 && comm->ThisTask < comm->NTask, "mpg_dist_create: bad ThisTask / NTask");
    mpg_dist *d = new mpg_dist();
    d->eng = eng;
    d->comm = *comm;
    d->me = comm->ThisTask;
    d->nt = comm->NTask;
    if(comm->bind_stream && comm->device_buffers) {
        if(comm->bind_stream(comm->ctx, (void *)eng->stream) != 0) {
            delete d;
            throw Error("mpg_comm: bind_stream failed");
        }
        d->stream_ordered = true;
        d->bound_stream = eng->stream;
    }
    *out = d;
    API_END
}

void mpg_dist_destroy(mpg_dist *d)
{
    if(!d)
        return;
    if(d->eng) {
        (void)hipSetDevice(d->eng->device);
        (void)hipStreamSynchronize(d->eng->stream);
        d->eng->tree.force_internal_above = 0;
        if(d->tree_stream)
            (void)hipStreamDestroy(d->tree_stream);
    }
    delete d;
}

int mpg_dist_set_domain(mpg_dist *d, double BoxSize, const mpg_topnode *TopNodes, int NTopNodes, const int *leaf_task, int NTopLeaves,
                        double margin, int La)
{
    API_BEGIN
    MPG_CHECK(d && TopNodes && leaf_task && NTopNodes > 0 && NTopLeaves > 0, "null argument");
    MPG_CHECK(BoxSize > 0 && margin > 0, "mpg_dist_set_domain: BoxSize and margin must be positive");
    MPG_HIP(hipSetDevice(d->eng->device));
    if(La <= 0) // cells [margin, 2 margin) wide: little over-import, few levels above the decomposition level
        La = (int)floor(log2(1.001 * BoxSize / margin));
    La = std::max(1, std::min(7, La));
    const int nc = 1 << La;
    const double w = 1.001 * BoxSize / nc, lo0 = BoxSize / 2. - 0.5 * 1.001 * BoxSize;
    // A node above level La that holds no particle of a needed cell is absent from the local tree.  That is harmless only if the
    // global walk could not have drawn a force from it: a node used unopened has len^2 / r^2 <= MaxBHOpeningAngle^2 <= 1, i.e. lies
    // at r >= len >= 2 w >= 2 margin from the target, beyond the short-range window's table (15 mesh cells, gravity.c:57-61)
    // whenever margin >= Rcut = 9 cells - so cells must not be narrower than the margin
    MPG_CHECK(w >= margin || La == 1, "mpg_dist_set_domain: cells of level La are narrower than the margin");
    const double grow = margin + 1e-9 * BoxSize; // (ownership is by integer key, cells by floating-point descent)
    std::vector<unsigned long long> need((size_t)nc * nc * nc, 0ull);
    std::vector<char> sx, sy, sz;
    int seen = 0;
    for(int i = 0; i < NTopNodes; i++) {
        const mpg_topnode &tn = TopNodes[i];
        if(tn.Daughter >= 0)
            continue;
        MPG_CHECK(tn.Leaf >= 0 && tn.Leaf < NTopLeaves, "mpg_dist_set_domain: TopNode without a valid Leaf");
        const int task = leaf_task[tn.Leaf];
        MPG_CHECK(task >= 0 && task < d->nt, "mpg_dist_set_domain: Task of a TopLeaf out of range");
        seen++;
        int level;
        unsigned x, y, z;
        topnode_cube(tn, level, x, y, z);
        const double s = 1.001 * BoxSize / (double)(1u << level);
        cells_of_interval(lo0 + x * s - grow, lo0 + (x + 1) * s + grow, BoxSize, lo0, w, nc, sx);
        cells_of_interval(lo0 + y * s - grow, lo0 + (y + 1) * s + grow, BoxSize, lo0, w, nc, sy);
        cells_of_interval(lo0 + z * s - grow, lo0 + (z + 1) * s + grow, BoxSize, lo0, w, nc, sz);
        const unsigned long long bit = 1ull << task;
        for(int a = 0; a < nc; a++) {
            if(!sx[a])
                continue;
            for(int b = 0; b < nc; b++) {
                if(!sy[b])
                    continue;
                unsigned long long *row = need.data() + ((size_t)a * nc + b) * nc;
                for(int c = 0; c < nc; c++)
                    if(sz[c])
                        row[c] |= bit;
            }
        }
    }
    MPG_CHECK(seen == NTopLeaves, "mpg_dist_set_domain: the TopNodes do not hold NTopLeaves leaves");
    d->need.reserve(need.size());
    MPG_HIP(hipMemcpy(d->need.p, need.data(), need.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
    d->box = BoxSize;
    d->margin = margin;
    d->La = La;
    d->have_domain = true;
    d->stats[3] = La;
    API_END
}

int mpg_dist_dev_set_garbage(mpg_dist *d, int64_t n_own, const unsigned char *d_garbage)
{
    API_BEGIN
    MPG_CHECK(d && n_own >= 0, "null argument");
    MPG_HIP(hipSetDevice(d->eng->device));
    d->d_skip = nullptr;
    d->n_skip = 0;
    d->n_skip_rows = -1;
    if(d_garbage && n_own > 0) {
        hipStream_t st = d->eng->stream;
        d->scount.reserve(4);
        MPG_HIP(hipMemsetAsync(d->scount.p, 0, sizeof(unsigned long long), st));
        hipLaunchKernelGGL(k_count_nonzero, dim3(nblk(n_own)), dim3(256), 0, st, n_own, d_garbage, d->scount.p);
        unsigned long long c = 0;
        MPG_HIP(hipMemcpyAsync(&c, d->scount.p, sizeof(c), hipMemcpyDeviceToHost, st));
        sync(d);
        if(c > 0) {
            // a COPY of the flags (the caller's pointer was kept "until the next call" and matched to a table by its row
            // count alone - a caller moving to another table of the same size without calling this again would have had freed or stale
            // flags applied to it); they stay in force until the next call of this function, as documented
            d->skip_own.reserve((size_t)n_own + 1);
            MPG_HIP(hipMemcpyAsync(d->skip_own.p, d_garbage, (size_t)n_own, hipMemcpyDeviceToDevice, st));
            d->d_skip = d->skip_own.p;
            d->n_skip = (int64_t)c;
            d->n_skip_rows = n_own;
        }
    }
    API_END
}

int mpg_dist_dev_gravpm_force(mpg_dist *d, int64_t n_own, const double *d_pos, const float *d_mass, double *d_gravpm, double *d_potential)
{
    API_BEGIN
    MPG_CHECK(d && d_gravpm && (n_own == 0 || (d_pos && d_mass)), "null argument");
    MPG_CHECK(d->have_domain, "mpg_dist: mpg_dist_set_domain first");
    mpg_engine *e = d->eng;
    MPG_HIP(hipSetDevice(e->device));
    MPG_CHECK(n_own < (1ll << 31), "mpg_dist: too many particles on one rank");
    MPG_CHECK(e->pm.box == d->box, "mpg_dist: BoxSize of the mesh differs from the domain's");
    d->stats[4] = d->stats[5] = 0;
    if(d->skip_for(n_own)) // (gravpm.c:88-92 zeroes GravPM of every particle; the readout then reaches the live ones only)
        MPG_HIP(hipMemsetAsync(d_gravpm, 0, (size_t)3 * n_own * sizeof(double), e->stream));
    sync(d);
    const double t0 = now_ms();
    pm_step(d, n_own, d_pos, d_mass, d_gravpm, d_potential);
    sync(d);
    d->times[0] = now_ms() - t0;
    API_END
}

// The three phases of the local tree: ghosts (collective), build (no collective: may run on another stream beside the PM), global top
// + targets (collective).
static void tree_build_local(mpg_dist *d, int64_t n_own, int64_t nl, hipStream_t st)
{
    mpg_engine *e = d->eng;
    const uint8_t *ltype = nullptr;
    if(d->skip_for(n_own)) { // garbage among the own rows: type 7 keeps them out of the tree (forcetree.c:806)
        d->ltype.reserve((size_t)nl + 1);
        hipLaunchKernelGGL(k_local_types, dim3(nblk(nl)), dim3(256), 0, st, nl, n_own, d->d_skip, d->ltype.p);
        ltype = d->ltype.p;
    }
    MPG_CHECK(mpg_dev_bind_particles(e, nl, d->lpos.p, d->lmass.p, ltype, d->box) == 0, mpg_last_error());
    e->tree.force_internal_above = d->La;
    try {
        engine_tree_build_on(e, 63, st);
    }
    catch(...) {
        e->tree.force_internal_above = 0;
        throw;
    }
    e->tree.force_internal_above = 0;
}

// the deferred checks of tree_finish; call after a synchronisation of the engine's stream
static void tree_checks(mpg_dist *d)
{
    if(!d->chk_pending)
        return;
    d->chk_pending = false;
    const unsigned *ef = (const unsigned *)&d->chk.p[1];
    MPG_CHECK(ef[0] == 0, "mpg_dist: a cell above the decomposition level holds <= 8 particles in all (use a coarser level La)");
    MPG_CHECK(ef[1] == 0, "domain decomposition: a cell above the decomposition level holds <= 8 local particles (use a coarser level)");
    MPG_CHECK((int64_t)d->chk.p[0] == d->ntarg, "mpg_dist: own particles missing from the local tree");
}

static void tree_finish(mpg_dist *d, int64_t n_own, int64_t nl)
{
    mpg_engine *e = d->eng;
    hipStream_t st = e->stream;
    global_top(d, n_own);
    // own particles in tree order: the walk's targets
    d->targets.reserve((size_t)nl + 1);
    d->scount.reserve(4);
    d->ntarg = 0;
    d->n_own_tree = n_own;
    if(nl > 0) {
        size_t tb = 0;
        MPG_HIP(rocprim::select(nullptr, tb, e->tree.idx_b.p, (int *)d->targets.p, d->scount.p, (size_t)e->tree.npart, IsOwn{(int)n_own}, st));
        d->tmp.reserve(tb + 16);
        MPG_HIP(rocprim::select((void *)d->tmp.p, tb, e->tree.idx_b.p, (int *)d->targets.p, d->scount.p, (size_t)e->tree.npart, IsOwn{(int)n_own}, st));
        // every own live particle is in the local tree, so the number of targets is known; the count the selection found and the flags of
        // the global top travel to pinned memory behind it and are checked at the next point the host waits anyway (tree_checks)
        d->ntarg = n_own - (d->skip_for(n_own) ? d->n_skip : 0);
        d->chk.reserve(4);
        d->chk.p[0] = ~0ull;
        MPG_HIP(hipMemcpyAsync(&d->chk.p[0], d->scount.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    }
    else {
        d->chk.reserve(4);
        d->chk.p[0] = 0;
    }
    d->chk.p[1] = 0;
    MPG_HIP(hipMemcpyAsync(&d->chk.p[1], d->err.p + 1, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    d->chk_pending = true;
    d->grav_tree_valid = true;
}

int mpg_dist_dev_force_tree_build(mpg_dist *d, int64_t n_own, const double *d_pos, const float *d_mass)
{
    API_BEGIN
    MPG_CHECK(d && (n_own == 0 || (d_pos && d_mass)), "null argument");
    MPG_CHECK(d->have_domain, "mpg_dist: mpg_dist_set_domain first");
    mpg_engine *e = d->eng;
    MPG_HIP(hipSetDevice(e->device));
    MPG_CHECK(n_own < (1ll << 31), "mpg_dist: too many particles on one rank");
    sync(d);
    const double t1 = now_ms();
    // ---- ghosts, local tree, global top
    const int64_t nl = import_ghosts(d, n_own, d_pos, d_mass);
    sync(d);
    const double t2 = now_ms();
    d->times[1] = t2 - t1;
    tree_build_local(d, n_own, nl, e->stream);
    tree_finish(d, n_own, nl);
    sync(d); // (this entry point stands alone - the SPH loops and FOF follow it as well as the walk: its checks are made here)
    tree_checks(d);
    d->times[2] = now_ms() - t2;
    d->times[4] = 0;
    API_END
}

int mpg_dist_dev_grav_short_tree(mpg_dist *d, const double *d_oldacc, const double *d_prev_accel, const double *d_gravpm, double *d_accel,
                                 double *d_potential, double rho0)
{
    return mpg_dist_dev_grav_short_tree_active(d, nullptr, 0, d_oldacc, d_prev_accel, d_gravpm, d_accel, d_potential, rho0);
}

int mpg_dist_dev_grav_short_tree_active(mpg_dist *d, const int *d_active, int64_t nactive, const double *d_oldacc, const double *d_prev_accel,
                                        const double *d_gravpm, double *d_accel, double *d_potential, double rho0)
{
    API_BEGIN
    MPG_CHECK(d && d_accel, "null argument");
    MPG_CHECK(d->n_own_tree >= 0, "mpg_dist_dev_grav_short_tree: mpg_dist_dev_force_tree_build first");
    MPG_CHECK(d->grav_tree_valid, "mpg_dist_dev_grav_short_tree: the local tree was replaced since mpg_dist_dev_force_tree_build (the density loop "
                                  "builds a gas tree, FOF a tree of the primary types): build the gravity tree again");
    MPG_CHECK(nactive >= 0 && nactive <= d->n_own_tree, "mpg_dist_dev_grav_short_tree_active: bad number of active particles");
    mpg_engine *e = d->eng;
    MPG_HIP(hipSetDevice(e->device));
    sync(d);
    tree_checks(d);
    const double t3 = now_ms();
    // the walk's targets: the own particles in tree order, or those of them the caller lists as active (ActiveParticle of the
    // sub-steps, run.c:392-470: the tree holds every particle, a subset is walked)
    const int *targets = d->targets.p;
    int64_t ntarg = d->ntarg;
    if(d_active) {
        hipStream_t st = e->stream;
        ntarg = 0;
        if(nactive > 0 && d->ntarg > 0) {
            d->actflag.reserve((size_t)d->n_own_tree + 1);
            d->act_targets.reserve((size_t)d->n_own_tree + 1);
            d->err.reserve(4);
            MPG_HIP(hipMemsetAsync(d->err.p, 0, sizeof(unsigned), st));
            MPG_HIP(hipMemsetAsync(d->actflag.p, 0, (size_t)d->n_own_tree, st));
            hipLaunchKernelGGL(k_flag_list, dim3(nblk(nactive)), dim3(256), 0, st, nactive, d_active, (int)d->n_own_tree, d->actflag.p, d->err.p);
            size_t tb = 0;
            MPG_HIP(rocprim::select(nullptr, tb, d->targets.p, d->act_targets.p, d->scount.p, (size_t)d->ntarg, IsFlagged{d->actflag.p}, st));
            d->tmp.reserve(tb + 16);
            MPG_HIP(rocprim::select((void *)d->tmp.p, tb, d->targets.p, d->act_targets.p, d->scount.p, (size_t)d->ntarg, IsFlagged{d->actflag.p}, st));
            unsigned long long c = 0;
            unsigned bad = 0;
            MPG_HIP(hipMemcpyAsync(&c, d->scount.p, sizeof(c), hipMemcpyDeviceToHost, st));
            MPG_HIP(hipMemcpyAsync(&bad, d->err.p, sizeof(bad), hipMemcpyDeviceToHost, st));
            sync(d);
            MPG_CHECK(bad == 0, "mpg_dist_dev_grav_short_tree_active: an active index is not an own particle");
            // (garbage on the active list is skipped in place, treewalk.c:234: it is not in the tree, hence not among the targets.  The
            // garbage entries ON the list are counted, so that a duplicate cannot hide behind them: ADVICE round 4)
            int64_t ngarb = 0;
            if(d->skip_for(d->n_own_tree)) {
                MPG_HIP(hipMemsetAsync(d->scount.p, 0, sizeof(unsigned long long), st));
                hipLaunchKernelGGL(k_count_listed, dim3(nblk(nactive)), dim3(256), 0, st, nactive, d_active, d->d_skip, d->scount.p);
                unsigned long long g = 0;
                MPG_HIP(hipMemcpyAsync(&g, d->scount.p, sizeof(g), hipMemcpyDeviceToHost, st));
                sync(d);
                ngarb = (int64_t)g;
            }
            MPG_CHECK((int64_t)c + ngarb == nactive, "mpg_dist_dev_grav_short_tree_active: the active list holds duplicates");
            ntarg = (int64_t)c;
        }
        targets = d->act_targets.p;
    }
    if(ntarg > 0) {
        d->cost.reserve((size_t)d->n_own_tree + 1);
        MPG_HIP(hipMemsetAsync(d->cost.p, 0, (size_t)d->n_own_tree * sizeof(float), e->stream));
        float *keep = e->d_walk_cost;
        const int keep_variant = e->walk_variant;
        e->d_walk_cost = d->cost.p;
        if(e->walk_variant == 0) // the two-kernel walk whatever the target count: it is the one that records the work per target
            e->walk_variant = 6;
        const int rc = mpg_dev_grav_short_tree(e, d_oldacc, d_prev_accel, d_gravpm, targets, ntarg, d_accel, d_potential, rho0);
        e->d_walk_cost = keep;
        e->walk_variant = keep_variant;
        MPG_CHECK(rc == 0, mpg_last_error());
    }
    sync(d);
    d->times[3] = now_ms() - t3;
    API_END
}

int mpg_dist_gravity_step(mpg_dist *d, int64_t n_own, const double *d_pos, const float *d_mass, const double *d_oldacc,
                          const double *d_prev_accel, double *d_accel, double *d_gravpm, double *d_potential, double rho0)
{
    if(d && d_potential && n_own > 0) // (readout_potential accumulates; the walk then assigns the tree's, gravshort.h:94-95)
        if(hipMemsetAsync(d_potential, 0, (size_t)n_own * sizeof(double), d->eng->stream) != hipSuccess)
            return 1;
    static const bool no_overlap = getenv("MPG_DIST_NO_OVERLAP") != nullptr;
    if(no_overlap) {
        if(int rc = mpg_dist_dev_gravpm_force(d, n_own, d_pos, d_mass, d_gravpm, d_potential))
            return rc;
        if(int rc = mpg_dist_dev_force_tree_build(d, n_own, d_pos, d_mass))
            return rc;
        return mpg_dist_dev_grav_short_tree(d, d_oldacc, d_prev_accel, d_gravpm, d_accel, d_potential, rho0);
    }
    // The local tree does not depend on the PM force (run.c:522-546 runs them one after the other; both only read the positions): the
    // ghosts are imported first, then the tree of own + ghost particles is built by a second host thread on a second stream WHILE this
    // thread runs the PM step - whose phases end in collectives the host waits for, so the tree's kernels fill the gaps (one rank of a
    // 256^3-per-GPU run: 5 ms of tree build beside 18 ms of PM).  Every collective stays on this thread, in the same order on every rank.
    {
        API_BEGIN
        MPG_CHECK(d && d_gravpm && d_accel && (n_own == 0 || (d_pos && d_mass)), "null argument");
        MPG_CHECK(d->have_domain, "mpg_dist: mpg_dist_set_domain first");
        mpg_engine *e = d->eng;
        MPG_HIP(hipSetDevice(e->device));
        MPG_CHECK(n_own < (1ll << 31), "mpg_dist: too many particles on one rank");
        MPG_CHECK(e->pm.box == d->box, "mpg_dist: BoxSize of the mesh differs from the domain's");
        d->stats[4] = d->stats[5] = 0;
        if(d->skip_for(n_own))
            MPG_HIP(hipMemsetAsync(d_gravpm, 0, (size_t)3 * n_own * sizeof(double), e->stream));
        sync(d);
        const double t0 = now_ms();
        const int64_t nl = import_ghosts(d, n_own, d_pos, d_mass);
        sync(d); // (d->lpos / d->lmass are complete: the other stream may read them)
        const double t1 = now_ms();
        d->times[1] = t1 - t0;
        if(!d->tree_stream)
            MPG_HIP(hipStreamCreateWithFlags(&d->tree_stream, hipStreamNonBlocking));
        std::string werr;
        double t_tree = 0;
        std::thread worker([&]() {
            try {
                MPG_HIP(hipSetDevice(e->device));
                const double ta = now_ms();
                tree_build_local(d, n_own, nl, d->tree_stream);
                MPG_HIP(hipStreamSynchronize(d->tree_stream));
                t_tree = now_ms() - ta;
            }
            catch(const std::exception &ex) {
                werr = ex.what();
                if(werr.empty())
                    werr = "tree build failed";
            }
        });
        std::string perr;
        try {
            pm_step(d, n_own, d_pos, d_mass, d_gravpm, d_potential);
            sync(d);
        }
        catch(const std::exception &ex) {
            perr = ex.what();
        }
        worker.join();
        // (both failures are reported; ADVICE round 3.  The invariant the two threads rely on: tree_build_local touches the engine's tree,
        // its particle binding (eng->n, d_pos, tree_mask) and d->tree_stream only, pm_step touches the mesh, the slab buffers, the
        // communicator and eng->stream only, and neither reads what the other writes before the join above.)
        MPG_CHECK(perr.empty() && werr.empty(), perr.empty() ? werr : (werr.empty() ? perr : perr + " | and the tree build beside it: " + werr));
        const double t2 = now_ms();
        d->times[0] = t2 - t1; // PM with the tree build beside it
        d->times[4] = t_tree;  // ... of which the tree build took this long on its stream
        tree_finish(d, n_own, nl);
        d->times[2] = now_ms() - t2;
        API_END_NORETURN
    }
    return mpg_dist_dev_grav_short_tree(d, d_oldacc, d_prev_accel, d_gravpm, d_accel, d_potential, rho0);
}

} // extern "C"

/* ---- the drop-in (host pointer) forms: the rank's P[] in host memory, results written back into it ------------------------ */
namespace {
// Pos / Mass of the rank's particle table onto the device (d->o_pos, d->o_mass)
void stage_own(mpg_dist *d, const mpg_particle_view *P)
{
    MPG_CHECK(P && (P->n == 0 || P->base), "null particle view");
    MPG_CHECK(P->off_pos >= 0 && P->off_mass >= 0, "particle view needs Pos and Mass");
    const int64_t n = P->n;
    hipStream_t st = d->eng->stream;
    d->hbuf.resize(3 * (size_t)n + 3);
    d->hbuf_f.resize((size_t)n + 1);
    const mpg_particle_view V = *P;
    const char *b = (const char *)P->base;
    double *hd = d->hbuf.data();
    float *hf = d->hbuf_f.data();
    // IsGarbage (bit 0) and Swallowed (bit 1) of the flag byte (partmanager.h:24-44): such particles stay where they are in P[] and are
    // skipped in place by every loop of this path (treewalk.c:234, forcetree.c:806, gravpm.c:176-179) - after star formation and black
    // hole mergers every sub-step sees some until the next domain_decompose_full collects them
    d->hbuf_b.resize((size_t)n + 1);
    uint8_t *hb = d->hbuf_b.data();
    std::vector<int> bad(64, 0);
    parallel_for(n, [=, &bad](int64_t lo, int64_t hi) {
        int any = 0;
        for(int64_t i = lo; i < hi; i++) {
            const char *rec = b + i * V.stride;
            const double *pp = (const double *)(rec + V.off_pos);
            hd[3 * i] = pp[0];
            hd[3 * i + 1] = pp[1];
            hd[3 * i + 2] = pp[2];
            hf[i] = *(const float *)(rec + V.off_mass);
            hb[i] = (V.off_flags >= 0 && (*(const uint8_t *)(rec + V.off_flags) & 3)) ? 1 : 0;
            any |= hb[i];
        }
        if(any)
            bad[0] = 1;
    });
    if(bad[0]) {
        d->o_skip.reserve((size_t)n + 1);
        MPG_HIP(hipMemcpyAsync(d->o_skip.p, hb, (size_t)n, hipMemcpyHostToDevice, st));
        MPG_CHECK(mpg_dist_dev_set_garbage(d, n, d->o_skip.p) == 0, mpg_last_error());
    }
    else
        MPG_CHECK(mpg_dist_dev_set_garbage(d, n, nullptr) == 0, mpg_last_error());
    d->o_pos.reserve(3 * (size_t)n + 3);
    d->o_mass.reserve((size_t)n + 1);
    d->o_gravpm.reserve(3 * (size_t)n + 3);
    d->o_pot.reserve((size_t)n + 1);
    d->o_acc.reserve(3 * (size_t)n + 3);
    d->o_prev.reserve(3 * (size_t)n + 3);
    if(n > 0) {
        MPG_HIP(hipMemcpyAsync(d->o_pos.p, hd, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
        MPG_HIP(hipMemcpyAsync(d->o_mass.p, hf, (size_t)n * sizeof(float), hipMemcpyHostToDevice, st));
    }
    sync(d);
    d->o_n = n;
}

// one 3-vector (or scalar) column of P[] <-> a device array
void column_up(mpg_dist *d, const mpg_particle_view *P, int64_t off, int w, double *dev)
{
    const int64_t n = P->n;
    d->hbuf.resize((size_t)w * n + 3);
    const mpg_particle_view V = *P;
    const char *b = (const char *)P->base;
    double *hd = d->hbuf.data();
    parallel_for(n, [=](int64_t lo, int64_t hi) {
        for(int64_t i = lo; i < hi; i++)
            for(int k = 0; k < w; k++)
                hd[w * i + k] = ((const double *)(b + i * V.stride + off))[k];
    });
    if(n > 0)
        MPG_HIP(hipMemcpyAsync(dev, hd, (size_t)w * n * sizeof(double), hipMemcpyHostToDevice, d->eng->stream));
    sync(d);
}

template <class F> void column_down(mpg_dist *d, int64_t n, int w, const double *dev, F put)
{
    d->hbuf.resize((size_t)w * n + 3);
    double *hd = d->hbuf.data();
    if(n > 0)
        MPG_HIP(hipMemcpyAsync(hd, dev, (size_t)w * n * sizeof(double), hipMemcpyDeviceToHost, d->eng->stream));
    sync(d);
    parallel_for(n, [=](int64_t lo, int64_t hi) {
        for(int64_t i = lo; i < hi; i++)
            put(i, hd + (size_t)w * i);
    });
}
} // namespace

extern "C" {

int mpg_dist_gravpm_force(mpg_dist *d, const mpg_particle_view *P)
{
    API_BEGIN
    MPG_CHECK(d && P, "null argument");
    MPG_CHECK(P->off_gravpm >= 0, "particle view needs GravPM");
    MPG_HIP(hipSetDevice(d->eng->device));
    stage_own(d, P);
    const int64_t n = P->n;
    if(n > 0)
        MPG_HIP(hipMemsetAsync(d->o_pot.p, 0, (size_t)n * sizeof(double), d->eng->stream));
    MPG_CHECK(mpg_dist_dev_gravpm_force(d, n, d->o_pos.p, d->o_mass.p, d->o_gravpm.p, d->o_pot.p) == 0, mpg_last_error());
    const mpg_particle_view V = *P;
    char *b = (char *)P->base;
    column_down(d, n, 3, d->o_gravpm.p, [=](int64_t i, const double *v) {
        double *g = (double *)(b + i * V.stride + V.off_gravpm);
        g[0] = v[0];
        g[1] = v[1];
        g[2] = v[2];
    });
    if(P->off_potential >= 0)
        column_down(d, n, 1, d->o_pot.p, [=](int64_t i, const double *v) { *(double *)(b + i * V.stride + V.off_potential) += v[0]; });
    API_END
}

int mpg_dist_force_tree_full(mpg_dist *d, const mpg_particle_view *P)
{
    API_BEGIN
    MPG_CHECK(d && P, "null argument");
    MPG_HIP(hipSetDevice(d->eng->device));
    stage_own(d, P);
    MPG_CHECK(mpg_dist_dev_force_tree_build(d, P->n, d->o_pos.p, d->o_mass.p) == 0, mpg_last_error());
    API_END
}

int mpg_dist_grav_short_tree(mpg_dist *d, const mpg_particle_view *P, double (*AccelStore)[3], double rho0)
{
    return mpg_dist_grav_short_tree_active(d, P, nullptr, 0, AccelStore, rho0);
}

int mpg_dist_grav_short_tree_active(mpg_dist *d, const mpg_particle_view *P, const int *ActiveParticle, int64_t NumActiveParticle,
                                    double (*AccelStore)[3], double rho0)
{
    API_BEGIN
    MPG_CHECK(d && P, "null argument");
    MPG_CHECK(P->off_accel >= 0 && P->off_gravpm >= 0, "particle view needs FullTreeGravAccel and GravPM");
    MPG_CHECK(d->o_n == P->n && d->n_own_tree == P->n, "mpg_dist_grav_short_tree: call mpg_dist_force_tree_full on this table first");
    MPG_CHECK(!ActiveParticle || (NumActiveParticle >= 0 && NumActiveParticle <= P->n), "bad NumActiveParticle");
    MPG_HIP(hipSetDevice(d->eng->device));
    const int64_t n = P->n;
    // OldAcc = |FullTreeGravAccel + GravPM| / G of the table as it stands (grav_get_abs_accel, gravshort.h:70-80)
    column_up(d, P, P->off_accel, 3, d->o_prev.p);
    column_up(d, P, P->off_gravpm, 3, d->o_gravpm.p);
    const int *d_act = nullptr;
    if(ActiveParticle) {
        d->o_act.reserve((size_t)NumActiveParticle + 1);
        if(NumActiveParticle > 0)
            MPG_HIP(hipMemcpyAsync(d->o_act.p, ActiveParticle, (size_t)NumActiveParticle * sizeof(int), hipMemcpyHostToDevice, d->eng->stream));
        d_act = d->o_act.p;
    }
    const bool pot = P->off_potential >= 0;
    MPG_CHECK(mpg_dist_dev_grav_short_tree_active(d, d_act, NumActiveParticle, nullptr, d->o_prev.p, d->o_gravpm.p, d->o_acc.p,
                                                  pot ? d->o_pot.p : nullptr, rho0) == 0,
              mpg_last_error());
    // results of the walked particles into the table: P[i].FullTreeGravAccel (full particle tree, gravshort.h:57-62), AccelStore[i],
    // P[i].Potential
    const mpg_particle_view V = *P;
    char *b = (char *)P->base;
    d->hbuf.resize(4 * (size_t)n + 4);
    double *ha = d->hbuf.data(), *hp = ha + 3 * (size_t)n;
    if(n > 0) {
        MPG_HIP(hipMemcpyAsync(ha, d->o_acc.p, 3 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost, d->eng->stream));
        if(pot)
            MPG_HIP(hipMemcpyAsync(hp, d->o_pot.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, d->eng->stream));
    }
    sync(d);
    const int64_t m = ActiveParticle ? NumActiveParticle : n;
    const uint8_t *dead = (d->n_skip > 0 && (int64_t)d->hbuf_b.size() > n) ? d->hbuf_b.data() : nullptr; // (flags of this table: stage_own)
    parallel_for(m, [=](int64_t lo, int64_t hi) {
        for(int64_t k = lo; k < hi; k++) {
            const int64_t i = ActiveParticle ? ActiveParticle[k] : k;
            if(dead && dead[i])
                continue; // garbage / swallowed: not walked, nothing to write (treewalk.c:234)
            double *a = (double *)(b + i * V.stride + V.off_accel);
            for(int j = 0; j < 3; j++) {
                a[j] = ha[3 * i + j];
                if(AccelStore)
                    AccelStore[i][j] = ha[3 * i + j];
            }
            if(pot)
                *(double *)(b + i * V.stride + V.off_potential) = hp[i];
        }
    });
    API_END
}

/* ---- density() and hydro_force() for the rank's own gas (density.h:42, hydra.h) on the particle set of the last
 * mpg_dist_dev_force_tree_build: own particles + the ghosts of every tree cell within the domain margin, which must cover the largest
 * smoothing length (checked).  The ghosts' inputs arrive along the ghost plan before the density loop; their Hsml, Density,
 * EgyWtDensity, DhsmlEgyDensityFactor, DivVel, CurlVel are refreshed from their owners before the hydro loop (the search of
 * hydro_force is symmetric in the two smoothing lengths, treewalk.c:1015-1042). */
int mpg_dist_dev_density(mpg_dist *d, int64_t n_own, const uint8_t *d_type, const mpg_sph_arrays *A, const mpg_sph_times *T, int update_hsml,
                         int DoEgyDensity)
{
    return mpg_dist_dev_density_active(d, n_own, d_type, A, T, nullptr, 0, update_hsml, DoEgyDensity);
}

// the loop's targets: the own gas (d_active == null) or the own gas among the listed particles, ascending
static void sph_targets(mpg_dist *d, int64_t n_own, const int *d_active, int64_t nactive, const int bh)
{
    hipStream_t st = d->eng->stream;
    d->gas.reserve((size_t)n_own + 1);
    d->scount.reserve(4);
    d->ngas = 0;
    if(n_own == 0 || (d_active && nactive == 0))
        return;
    rocprim::counting_iterator<int> iota(0);
    size_t tb = 0;
    unsigned bad = 0;
    if(d_active) {
        MPG_CHECK(nactive > 0 && nactive <= n_own, "mpg_dist SPH loop: bad number of active particles");
        d->actflag.reserve((size_t)n_own + 1);
        d->err.reserve(4);
        MPG_HIP(hipMemsetAsync(d->err.p, 0, sizeof(unsigned), st));
        MPG_HIP(hipMemsetAsync(d->actflag.p, 0, (size_t)n_own, st));
        hipLaunchKernelGGL(k_flag_list, dim3(nblk(nactive)), dim3(256), 0, st, nactive, d_active, (int)n_own, d->actflag.p, d->err.p);
        const IsActiveGas pred{d->s_type.p, d->actflag.p, bh};
        MPG_HIP(rocprim::select(nullptr, tb, iota, d->gas.p, d->scount.p, (size_t)n_own, pred, st));
        d->tmp.reserve(tb + 16);
        MPG_HIP(rocprim::select((void *)d->tmp.p, tb, iota, d->gas.p, d->scount.p, (size_t)n_own, pred, st));
        MPG_HIP(hipMemcpyAsync(&bad, d->err.p, sizeof(bad), hipMemcpyDeviceToHost, st));
    }
    else {
        MPG_HIP(rocprim::select(nullptr, tb, iota, d->gas.p, d->scount.p, (size_t)n_own, IsOwnGas{d->s_type.p, bh}, st));
        d->tmp.reserve(tb + 16);
        MPG_HIP(rocprim::select((void *)d->tmp.p, tb, iota, d->gas.p, d->scount.p, (size_t)n_own, IsOwnGas{d->s_type.p, bh}, st));
    }
    unsigned long long c = 0;
    MPG_HIP(hipMemcpyAsync(&c, d->scount.p, sizeof(c), hipMemcpyDeviceToHost, st));
    sync(d);
    MPG_CHECK(bad == 0, "mpg_dist SPH loop: an active index is not an own particle");
    d->ngas = (int64_t)c;
}

int mpg_dist_dev_density_active(mpg_dist *d, int64_t n_own, const uint8_t *d_type, const mpg_sph_arrays *A, const mpg_sph_times *T,
                                const int *d_active, int64_t nactive, int update_hsml, int DoEgyDensity)
{
    API_BEGIN
    MPG_CHECK(d && A && T && A->hsml && A->density && A->dhsmlegyfac && A->divvel && A->curlvel, "null argument");
    MPG_CHECK(d->n_own_tree == n_own, "mpg_dist_dev_density: mpg_dist_dev_force_tree_build of this particle set first");
    mpg_engine *e = d->eng;
    MPG_HIP(hipSetDevice(e->device));
    hipStream_t st = e->stream;
    const Plan &pl = d->ghost;
    const int64_t nl = n_own + pl.nrecv;
    d->s_type.reserve((size_t)nl + 1);
    d->s_tbh.reserve((size_t)nl + 1);
    d->s_tbg.reserve((size_t)nl + 1);
    const int w_in[7] = {1, 3, 1, 3, 3, 3, 1}, w_out[11] = {1, 1, 1, 1, 1, 1, 3, 3, 1, 1, 0};
    for(int k = 0; k < 7; k++)
        d->s_in[k].reserve((size_t)w_in[k] * nl + 3);
    for(int k = 0; k < 10; k++) {
        d->s_out[k].reserve((size_t)w_out[k] * nl + 3);
        MPG_HIP(hipMemsetAsync(d->s_out[k].p, 0, (size_t)w_out[k] * nl * sizeof(double), st));
    }
    const SphIn in{d_type, A->tb_hydro, A->tb_grav, A->hsml, A->vel, A->entropy, A->gacc, A->gpm, A->hydroacc_in, A->dtentropy_in};
    const SphLocal loc{d->s_type.p, d->s_tbh.p, d->s_tbg.p, d->s_in[0].p, d->s_in[1].p, d->s_in[2].p, d->s_in[3].p, d->s_in[4].p, d->s_in[5].p,
                       d->s_in[6].p};
    if(n_own > 0)
        hipLaunchKernelGGL(k_fill_sph_own, dim3(nblk(n_own)), dim3(256), 0, st, n_own, in, loc);
    d->sendbuf.reserve((size_t)128 * pl.nsend + 128);
    d->recvbuf.reserve((size_t)128 * pl.nrecv + 128);
    if(pl.nsend > 0)
        hipLaunchKernelGGL(k_pack_sph_in, dim3(nblk(pl.nsend)), dim3(256), 0, st, pl.nsend, pl.idx.p, in, (double *)d->sendbuf.p);
    exchange_rows(d, pl, d->sendbuf.p, d->recvbuf.p, false, 128);
    if(pl.nrecv > 0)
        hipLaunchKernelGGL(k_unpack_sph_in, dim3(nblk(pl.nrecv)), dim3(256), 0, st, pl.nrecv, (const double *)d->recvbuf.p, loc, n_own);
    // the gas tree of the local set (force_tree_rebuild_mask(GASMASK), run.c:466); it REPLACES the gravity tree in the engine
    d->grav_tree_valid = false;
    MPG_CHECK(mpg_dev_bind_particles(e, nl, d->lpos.p, d->lmass.p, d->s_type.p, d->box) == 0, mpg_last_error());
    MPG_CHECK(mpg_dev_force_tree_rebuild_mask(e, 1, 0) == 0, mpg_last_error());
    sph_targets(d, n_own, d_active, nactive, 1); // (black holes are density targets whatever BlackHoleOn says: density_haswork)
    if(d_active && n_own > 0) {
        // a sub-step: the inactive own particles keep the results of their last density loop, which the hydro loop of this sub-step
        // reads and their owners hand to the ghosts (the reference leaves SphP of inactive particles alone)
        auto in = [&](double *dst, const double *src, int w) {
            if(src)
                MPG_HIP(hipMemcpyAsync(dst, src, (size_t)w * n_own * sizeof(double), hipMemcpyDeviceToDevice, st));
        };
        in(d->s_out[0].p, A->dthsml, 1);
        in(d->s_out[1].p, A->density, 1);
        in(d->s_out[2].p, A->egywtdensity, 1);
        in(d->s_out[3].p, A->dhsmlegyfac, 1);
        in(d->s_out[4].p, A->divvel, 1);
        in(d->s_out[5].p, A->curlvel, 1);
        in(d->s_out[6].p, A->gradrho, 3);
    }
    mpg_sph_arrays L;
    memset(&L, 0, sizeof(L));
    L.hsml = d->s_in[0].p;
    L.vel = d->s_in[1].p;
    L.entropy = d->s_in[2].p;
    L.gacc = d->s_in[3].p;
    L.gpm = d->s_in[4].p;
    L.hydroacc_in = d->s_in[5].p;
    L.dtentropy_in = d->s_in[6].p;
    L.tb_hydro = d->s_tbh.p;
    L.tb_grav = d->s_tbg.p;
    L.dthsml = d->s_out[0].p;
    L.density = d->s_out[1].p;
    L.egywtdensity = d->s_out[2].p;
    L.dhsmlegyfac = d->s_out[3].p;
    L.divvel = d->s_out[4].p;
    L.curlvel = d->s_out[5].p;
    L.gradrho = A->gradrho ? d->s_out[6].p : nullptr;
    L.hydroacc_out = d->s_out[7].p;
    L.dtentropy_out = d->s_out[8].p;
    L.maxsignalvel = d->s_out[9].p;
    MPG_CHECK(mpg_dev_density(e, &L, T, d->gas.p, d->ngas, update_hsml, DoEgyDensity, d->blackholes) == 0, mpg_last_error());
    // every neighbour within a smoothing length must be local: the largest one against the domain margin
    d->scount.reserve(4);
    MPG_HIP(hipMemsetAsync(d->scount.p, 0, sizeof(unsigned long long), st));
    if(n_own > 0)
        hipLaunchKernelGGL(k_max_gas_hsml, dim3(nblk(n_own)), dim3(256), 0, st, n_own, d->s_type.p, d->s_in[0].p, 1, d->scount.p);
    unsigned long long hb = 0;
    MPG_HIP(hipMemcpyAsync(&hb, d->scount.p, sizeof(hb), hipMemcpyDeviceToHost, st));
    sync(d);
    double hmax;
    memcpy(&hmax, &hb, sizeof(double));
    allreduce_host_f64(d, &hmax, 1, 1);
    d->last_hmax = hmax;
    MPG_CHECK(hmax <= d->margin, "mpg_dist_dev_density: the largest smoothing length exceeds the domain margin (mpg_dist_set_domain with a larger one)");
    // own rows out
    auto out = [&](double *dst, const double *src, int w) {
        if(dst && n_own > 0)
            MPG_HIP(hipMemcpyAsync(dst, src, (size_t)w * n_own * sizeof(double), hipMemcpyDeviceToDevice, st));
    };
    out(A->hsml, L.hsml, 1);
    out(A->dthsml, L.dthsml, 1);
    out(A->density, L.density, 1);
    out(A->egywtdensity, L.egywtdensity, 1);
    out(A->dhsmlegyfac, L.dhsmlegyfac, 1);
    out(A->divvel, L.divvel, 1);
    out(A->curlvel, L.curlvel, 1);
    out(A->gradrho, L.gradrho, 3);
    d->sph_nl = nl;
    d->sph_n_own = n_own;
    API_END
}

int mpg_dist_dev_hydro_force(mpg_dist *d, int64_t n_own, const mpg_sph_arrays *A, const mpg_sph_times *T)
{
    return mpg_dist_dev_hydro_force_active(d, n_own, A, T, nullptr, 0);
}

int mpg_dist_dev_hydro_force_active(mpg_dist *d, int64_t n_own, const mpg_sph_arrays *A, const mpg_sph_times *T, const int *d_active,
                                    int64_t nactive)
{
    API_BEGIN
    MPG_CHECK(d && A && T && A->hydroacc_out && A->dtentropy_out && A->maxsignalvel, "null argument");
    MPG_CHECK(d->sph_n_own == n_own && d->sph_nl >= n_own, "mpg_dist_dev_hydro_force: mpg_dist_dev_density of this particle set first");
    mpg_engine *e = d->eng;
    MPG_HIP(hipSetDevice(e->device));
    hipStream_t st = e->stream;
    const Plan &pl = d->ghost;
    // the ghosts' density-loop results from their owners
    d->sendbuf.reserve((size_t)48 * pl.nsend + 48);
    d->recvbuf.reserve((size_t)48 * pl.nrecv + 48);
    double *hs = d->s_in[0].p, *de = d->s_out[1].p, *eg = d->s_out[2].p, *dh = d->s_out[3].p, *dv = d->s_out[4].p, *cv = d->s_out[5].p;
    if(pl.nsend > 0)
        hipLaunchKernelGGL(k_pack_sph_mid, dim3(nblk(pl.nsend)), dim3(256), 0, st, pl.nsend, pl.idx.p, hs, de, eg, dh, dv, cv, (double *)d->sendbuf.p);
    exchange_rows(d, pl, d->sendbuf.p, d->recvbuf.p, false, 48);
    if(pl.nrecv > 0)
        hipLaunchKernelGGL(k_unpack_sph_mid, dim3(nblk(pl.nrecv)), dim3(256), 0, st, pl.nrecv, (const double *)d->recvbuf.p, n_own, hs, de, eg, dh, dv, cv);
    MPG_CHECK(mpg_dev_force_tree_calc_hmax(e) == 0, mpg_last_error()); // force_tree_calc_moments of the gas tree, run.c:477
    mpg_sph_arrays L;
    memset(&L, 0, sizeof(L));
    L.hsml = hs;
    L.vel = d->s_in[1].p;
    L.entropy = d->s_in[2].p;
    L.gacc = d->s_in[3].p;
    L.gpm = d->s_in[4].p;
    L.hydroacc_in = d->s_in[5].p;
    L.dtentropy_in = d->s_in[6].p;
    L.tb_hydro = d->s_tbh.p;
    L.tb_grav = d->s_tbg.p;
    L.dthsml = d->s_out[0].p;
    L.density = de;
    L.egywtdensity = eg;
    L.dhsmlegyfac = dh;
    L.divvel = dv;
    L.curlvel = cv;
    L.hydroacc_out = d->s_out[7].p;
    L.dtentropy_out = d->s_out[8].p;
    L.maxsignalvel = d->s_out[9].p;
    sph_targets(d, n_own, d_active, nactive, 0); // (hydro_force: gas only, hydra.c:193)
    if(d_active && n_own > 0) { // (inactive particles keep their HydroAccel / DtEntropy / MaxSignalVel)
        MPG_HIP(hipMemcpyAsync(L.hydroacc_out, A->hydroacc_out, (size_t)3 * n_own * sizeof(double), hipMemcpyDeviceToDevice, st));
        MPG_HIP(hipMemcpyAsync(L.dtentropy_out, A->dtentropy_out, (size_t)n_own * sizeof(double), hipMemcpyDeviceToDevice, st));
        MPG_HIP(hipMemcpyAsync(L.maxsignalvel, A->maxsignalvel, (size_t)n_own * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    MPG_CHECK(mpg_dev_hydro_force(e, &L, T, d->gas.p, d->ngas) == 0, mpg_last_error());
    if(n_own > 0) {
        MPG_HIP(hipMemcpyAsync(A->hydroacc_out, L.hydroacc_out, (size_t)3 * n_own * sizeof(double), hipMemcpyDeviceToDevice, st));
        MPG_HIP(hipMemcpyAsync(A->dtentropy_out, L.dtentropy_out, (size_t)n_own * sizeof(double), hipMemcpyDeviceToDevice, st));
        MPG_HIP(hipMemcpyAsync(A->maxsignalvel, L.maxsignalvel, (size_t)n_own * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    sync(d);
    API_END
}

const float *mpg_dist_walk_cost(mpg_dist *d) { return d ? d->cost.p : nullptr; }

int mpg_dist_get_stats(mpg_dist *d, int64_t stats[8])
{
    API_BEGIN
    MPG_CHECK(d && stats, "null argument");
    for(int i = 0; i < 8; i++)
        stats[i] = d->stats[i];
    API_END
}

int mpg_dist_get_times(mpg_dist *d, double ms[8])
{
    API_BEGIN
    MPG_CHECK(d && ms, "null argument");
    for(int i = 0; i < 8; i++)
        ms[i] = d->times[i];
    API_END
}

int mpg_dev_force_tree_set_min_leaf_level(mpg_engine *eng, int level)
{
    API_BEGIN
    MPG_CHECK(eng && level >= 0 && level <= 8, "mpg_dev_force_tree_set_min_leaf_level: level must be in [0, 8]");
    eng->tree.force_internal_above = level;
    API_END
}

} // extern "C"

/* ---- domain_decompose_full (domain.c:153-258) and domain_exchange (exchange.c) over the caller's communicator ----------------
 * The device passes (domain_sample, domain_topleaves) and the host arithmetic on the top-level tree (toptree_*) are domain.hip's;
 * here is the sequence the reference's MPI code runs them in.  The pairwise hand-over of trees (domain.c:1206-1259) is done on an
 * all-gather: every rank receives all local trees and folds them in the pairwise order itself - the same tree on every rank
 * without a broadcast. */
namespace {

void allreduce_i64(mpg_dist *d, int64_t *v, int64_t n, int op)
{
    if(d->nt == 1 && !d->comm.allreduce)
        return;
    MPG_CHECK(d->comm.allreduce, "mpg_comm: allreduce callback missing");
    cb(d->comm.allreduce(d->comm.ctx, v, n, 1, op, 0), "allreduce");
}

bool any_rank(mpg_dist *d, bool f)
{
    int64_t v = f ? 1 : 0;
    allreduce_i64(d, &v, 1, 0);
    return v > 0;
}

// every rank's block of host bytes (padded to 8): out[r] = the block of rank r
void allgather_host(mpg_dist *d, const void *buf, int64_t nbytes, std::vector<std::vector<char>> &out)
{
    const int nt = d->nt;
    out.assign(nt, {});
    if(nt == 1) {
        out[0].assign((const char *)buf, (const char *)buf + nbytes);
        return;
    }
    const int64_t pad = (nbytes + 7) / 8 * 8;
    std::vector<int64_t> sc(nt, pad), rc(nt, 0), sd(nt, 0), rd(nt + 1, 0);
    cb(d->comm.alltoall_i64(d->comm.ctx, sc.data(), rc.data()), "alltoall_i64");
    for(int r = 0; r < nt; r++)
        rd[r + 1] = rd[r] + rc[r];
    std::vector<char> sb((size_t)pad + 8, 0), rb((size_t)rd[nt] + 8);
    memcpy(sb.data(), buf, (size_t)nbytes);
    std::vector<int64_t> rdv(rd.begin(), rd.begin() + nt);
    cb(d->comm.alltoallv(d->comm.ctx, sb.data(), sc.data(), sd.data(), rb.data(), rc.data(), rdv.data(), 0), "alltoallv");
    for(int r = 0; r < nt; r++)
        out[r].assign(rb.begin() + rd[r], rb.begin() + rd[r] + rc[r]); // (padded: the receiver knows the true sizes from the data)
}

struct Policy { // DomainDecompositionPolicy as domain_policies_init fills it (domain.c:351-375)
    int PreSort, SubSampleDistance, NTopLeaves;
    Policy(int i, int ntask, int overdecomp)
    {
        PreSort = i >= 2 ? 1 : 0;
        int dd = 256;
        for(int k = 1; k <= i; k++)
            dd = (k > 4 && dd > 2) ? dd / 2 : 256;
        SubSampleDistance = dd;
        NTopLeaves = overdecomp * ntask * (i + 1);
    }
};

// domain_determine_global_toptree (domain.c:1280-1341): false when the top nodes ran out
bool global_toptree(mpg_dist *d, int64_t n, const double *pos, const uint8_t *garbage, double box, const Policy &pol, int global_sorting, int maxn,
                    std::vector<TopNode> &tree, int &size)
{
    mpg_engine *e = d->eng;
    const int nt = d->nt;
    const int64_t cap = n / pol.SubSampleDistance + 2;
    std::vector<uint64_t> keys((size_t)cap);
    int64_t ns = domain_sample(n, pos, garbage, box, pol.PreSort, pol.SubSampleDistance, keys.data(), cap, e->domain, e->stream);
    keys.resize((size_t)ns);
    if(global_sorting && nt > 1) { // mpsort_mpi (domain.c:1076-1077): sorted over all ranks, every rank keeps as many as it had
        struct Hdr {
            int64_t n;
        } h{ns};
        std::vector<char> blk(sizeof(Hdr) + (size_t)ns * 8);
        memcpy(blk.data(), &h, sizeof(h));
        if(ns)
            memcpy(blk.data() + sizeof(h), keys.data(), (size_t)ns * 8);
        std::vector<std::vector<char>> all;
        allgather_host(d, blk.data(), (int64_t)blk.size(), all);
        std::vector<uint64_t> allk;
        int64_t off = 0;
        for(int r = 0; r < nt; r++) {
            Hdr hr;
            memcpy(&hr, all[r].data(), sizeof(hr));
            const uint64_t *k = (const uint64_t *)(all[r].data() + sizeof(hr));
            if(r < d->me)
                off += hr.n;
            allk.insert(allk.end(), k, k + hr.n);
        }
        std::stable_sort(allk.begin(), allk.end());
        std::copy(allk.begin() + off, allk.begin() + off + ns, keys.begin());
    }
    tree.assign((size_t)maxn + 8, TopNode{});
    size = 0;
    bool ok = toptree_local_refine(keys.data(), nullptr, ns, tree.data(), &size, maxn);
    if(any_rank(d, !ok))
        return false;
    int64_t tot[2] = {tree[0].Cost, tree[0].Count};
    allreduce_i64(d, tot, 2, 0);
    const int64_t costlimit = tot[0] / pol.NTopLeaves, countlimit = tot[1] / pol.NTopLeaves;
    toptree_truncate(tree.data(), &size, countlimit, costlimit);
    bool err = false;
    if(nt > 1) {
        std::vector<char> blk(8 + (size_t)size * sizeof(TopNode));
        const int64_t sz = size;
        memcpy(blk.data(), &sz, 8);
        memcpy(blk.data() + 8, tree.data(), (size_t)size * sizeof(TopNode));
        std::vector<std::vector<char>> all;
        allgather_host(d, blk.data(), (int64_t)blk.size(), all);
        // the pairwise combination of domain.c:1206-1259, replayed on every rank: tree r absorbs tree r + sep, sep = 1, 2, 4 ...
        std::vector<std::vector<TopNode>> T((size_t)nt);
        std::vector<int> S((size_t)nt);
        for(int r = 0; r < nt; r++) {
            int64_t z;
            memcpy(&z, all[r].data(), 8);
            S[r] = (int)z;
            T[r].assign((size_t)maxn + 8, TopNode{});
            memcpy(T[r].data(), all[r].data() + 8, (size_t)z * sizeof(TopNode));
        }
        for(int sep = 1; sep < nt; sep *= 2)
            for(int r = 0; r + sep < nt; r += 2 * sep)
                if(!toptree_merge(T[r].data(), &S[r], T[r + sep].data(), S[r + sep], maxn))
                    err = true;
        tree.swap(T[0]);
        size = S[0];
        if(size >= maxn)
            err = true;
    }
    if(any_rank(d, err))
        return false;
    ok = toptree_global_refine(tree.data(), &size, maxn, countlimit, costlimit);
    return !any_rank(d, !ok);
}

} // namespace

extern "C" {

int mpg_dist_domain_decompose(mpg_dist *d, int64_t n, const double *d_pos, const unsigned char *d_garbage, double BoxSize,
                              int DomainOverDecompositionFactor, int DomainUseGlobalSorting, const float *d_cost, int *NTopNodes,
                              int *NTopLeaves)
{
    API_BEGIN
    MPG_CHECK(d && n >= 0 && (n == 0 || d_pos) && BoxSize > 0 && DomainOverDecompositionFactor >= 1, "mpg_dist_domain_decompose: bad argument");
    mpg_engine *e = d->eng;
    MPG_HIP(hipSetDevice(e->device));
    hipStream_t st = e->stream;
    const int nt = d->nt;
    for(int i = d->dom_policy; i < 16; i++) { // NPOLICY, domain.c:48
        const Policy pol(i, nt, DomainOverDecompositionFactor);
        std::vector<TopNode> tree;
        int size = 0;
        for(;;) { // domain.c:180-195: more top nodes until they suffice
            const int maxn = std::max((int)(d->dom_alloc_factor * (double)(n + 1)), 1);
            if(global_toptree(d, n, d_pos, d_garbage, BoxSize, pol, DomainUseGlobalSorting, maxn, tree, size))
                break;
            d->dom_alloc_factor *= 1.2;
            MPG_CHECK(d->dom_alloc_factor <= 10, "TopNodeAllocFactor unreasonably large");
        }
        std::vector<int> leaf_topnode((size_t)size);
        const int nleaves = toptree_create_leaves(tree.data(), size, leaf_topnode.data());
        // domain_balance (domain.c:481-500): particles per leaf over all ranks
        std::vector<int64_t> counts((size_t)nleaves, 0);
        domain_topleaves(n, d_pos, d_garbage, BoxSize, tree.data(), size, nleaves, nullptr, nt, nullptr, nullptr, counts.data(), nullptr, e->domain, st);
        allreduce_i64(d, counts.data(), nleaves, 0);
        std::vector<int64_t> leaf_cost = counts;
        d->dom_topleaf.reserve((size_t)n + 1);
        d->dom_task.reserve((size_t)n + 1);
        if(d_cost) { // the work of every TopLeaf: its particles' costs, over all ranks (leaves are still in key order here)
            std::vector<int> zero((size_t)nleaves, 0);
            domain_topleaves(n, d_pos, d_garbage, BoxSize, tree.data(), size, nleaves, zero.data(), nt, d->dom_topleaf.p, d->dom_task.p, nullptr,
                             nullptr, e->domain, st);
            d->dom_cost.reserve((size_t)nleaves + 1);
            MPG_HIP(hipMemsetAsync(d->dom_cost.p, 0, (size_t)nleaves * sizeof(double), st));
            MPG_CHECK((size_t)nleaves * sizeof(double) <= 60000, "mpg_dist_domain_decompose: too many TopLeaves for the cost pass");
            if(n > 0)
                hipLaunchKernelGGL(k_leaf_cost, dim3(std::min<unsigned>(nblk(n), 2048u)), dim3(256), (size_t)nleaves * sizeof(double), st, n,
                                   d->dom_topleaf.p, d_cost, nleaves, d->dom_cost.p);
            std::vector<double> lc((size_t)nleaves);
            MPG_HIP(hipMemcpyAsync(lc.data(), d->dom_cost.p, (size_t)nleaves * sizeof(double), hipMemcpyDeviceToHost, st));
            sync(d);
            allreduce_host_f64(d, lc.data(), nleaves, 0);
            for(int l = 0; l < nleaves; l++)
                leaf_cost[l] = std::max<int64_t>((int64_t)llround(lc[l]), 1);
        }
        std::vector<int> leaf_task((size_t)nleaves), start((size_t)nt), end((size_t)nt);
        toptree_assign_balanced(tree.data(), size, leaf_topnode.data(), nleaves, leaf_cost.data(), nt, 1, leaf_task.data(), start.data(), end.data());
        // the assignment renumbers the leaves by (Task, Key): TopLeaf and destination of every particle, counts in the final order
        std::vector<int64_t> fcounts((size_t)nleaves, 0), tcounts((size_t)nt, 0);
        domain_topleaves(n, d_pos, d_garbage, BoxSize, tree.data(), size, nleaves, leaf_task.data(), nt, d->dom_topleaf.p, d->dom_task.p, fcounts.data(),
                         tcounts.data(), e->domain, st);
        sync(d);
        allreduce_i64(d, fcounts.data(), nleaves, 0);
        if(d->dom_max_part > 0 && i < 15) {
            // domain_check_memory_bound (domain.c:378-424): a task that would hold more than MaxPart particles sends the loop to the
            // next policy ("Still try an exchange if this is the last policy", domain.c:199-201)
            std::vector<int64_t> load = tcounts;
            allreduce_i64(d, load.data(), nt, 0);
            if(*std::max_element(load.begin(), load.end()) > d->dom_max_part)
                continue;
        }
        d->dom_policy = i;
        d->dom_tree.assign((const mpg_topnode *)tree.data(), (const mpg_topnode *)tree.data() + size);
        d->dom_size = size;
        d->dom_nleaves = nleaves;
        d->dom_leaf_task = leaf_task;
        d->dom_leaf_topnode.assign(leaf_topnode.begin(), leaf_topnode.begin() + nleaves);
        d->dom_start = start;
        d->dom_end = end;
        d->dom_leaf_count = fcounts;
        d->dom_send_counts = tcounts;
        d->dom_n = n;
        if(NTopNodes)
            *NTopNodes = size;
        if(NTopLeaves)
            *NTopLeaves = nleaves;
        break;
    }
    API_END
}

/* domain_maintain (domain.c:262-319): the decomposition is kept, only P[].TopLeaf and the destination tasks are found again for the
 * (drifted) positions; mpg_dist_domain_exchange then moves the particles that left their owner's TopLeaves */
int mpg_dist_domain_maintain(mpg_dist *d, int64_t n, const double *d_pos, const unsigned char *d_garbage, double BoxSize, int64_t *n_leaving)
{
    API_BEGIN
    MPG_CHECK(d && d->dom_size > 0 && n >= 0 && (n == 0 || d_pos), "mpg_dist_domain_maintain: no decomposition / bad argument");
    mpg_engine *e = d->eng;
    MPG_HIP(hipSetDevice(e->device));
    d->dom_topleaf.reserve((size_t)n + 1);
    d->dom_task.reserve((size_t)n + 1);
    std::vector<int64_t> fcounts((size_t)d->dom_nleaves, 0), tcounts((size_t)d->nt, 0);
    domain_topleaves(n, d_pos, d_garbage, BoxSize, (const TopNode *)d->dom_tree.data(), d->dom_size, d->dom_nleaves, d->dom_leaf_task.data(), d->nt,
                     d->dom_topleaf.p, d->dom_task.p, fcounts.data(), tcounts.data(), e->domain, e->stream);
    sync(d);
    d->dom_send_counts = tcounts;
    d->dom_n = n;
    if(n_leaving) {
        int64_t stay = tcounts[(size_t)d->me], live = 0;
        for(int64_t c : tcounts)
            live += c;
        *n_leaving = live - stay;
    }
    API_END
}

int mpg_dist_domain_get(mpg_dist *d, mpg_topnode *TopNodes, int *leaf_task, int *StartLeaf, int *EndLeaf, int64_t *TopLeafCount)
{
    API_BEGIN
    MPG_CHECK(d && d->dom_size > 0, "mpg_dist_domain_get: no decomposition");
    if(TopNodes)
        std::copy(d->dom_tree.begin(), d->dom_tree.end(), TopNodes);
    if(leaf_task)
        std::copy(d->dom_leaf_task.begin(), d->dom_leaf_task.end(), leaf_task);
    if(StartLeaf)
        std::copy(d->dom_start.begin(), d->dom_start.end(), StartLeaf);
    if(EndLeaf)
        std::copy(d->dom_end.begin(), d->dom_end.end(), EndLeaf);
    if(TopLeafCount)
        std::copy(d->dom_leaf_count.begin(), d->dom_leaf_count.end(), TopLeafCount);
    API_END
}

int mpg_dist_domain_exchange(mpg_dist *d, int64_t n, int ncols, const void *const *d_cols, const int *col_bytes, int64_t *n_new, void **d_new_cols)
{
    API_BEGIN
    MPG_CHECK(d && n_new && d_new_cols && d_cols && col_bytes && ncols >= 1 && ncols <= 16, "mpg_dist_domain_exchange: bad argument");
    MPG_CHECK(d->dom_n == n, "mpg_dist_domain_exchange: mpg_dist_domain_decompose of this particle set first");
    mpg_engine *e = d->eng;
    MPG_HIP(hipSetDevice(e->device));
    hipStream_t st = e->stream;
    d->mask.reserve((size_t)n + 1);
    if(n > 0)
        hipLaunchKernelGGL(k_task_mask, dim3(nblk(n)), dim3(256), 0, st, n, d->dom_task.p, d->mask.p);
    build_plan(d, d->dom_plan, n, d->mask.p);
    Plan &pl = d->dom_plan;
    // Invariant (round 6; the one multi-rank failure of round 4 was its Python twin, PeanoDomain.exchange: "send counts and live rows unequal
    // on one rank"): the rows per destination counted from the per-particle task array by the plan's split pass must be the per-task counts the
    // decomposition's k_topleaf pass summed over the TopLeaves - two kernels, two reductions, one truth.  A difference names the rank, both
    // count vectors and the particle number instead of surfacing as a short exchange somewhere downstream.
    {
        bool same = (int)d->dom_send_counts.size() == d->nt;
        for(int r = 0; same && r < d->nt; r++)
            same = pl.scnt[(size_t)r] == d->dom_send_counts[(size_t)r];
        if(!same) {
            std::string a, b;
            for(int r = 0; r < d->nt; r++) {
                a += " " + std::to_string(r < (int)d->dom_send_counts.size() ? d->dom_send_counts[(size_t)r] : -1);
                b += " " + std::to_string(pl.scnt[(size_t)r]);
            }
            MPG_CHECK(false, "mpg_dist_domain_exchange: rank " + std::to_string(d->me) + " of " + std::to_string(d->nt) + ", " + std::to_string(n) +
                                 " particles: send counts of the decomposition [" + a + " ] differ from the rows per task of its task array [" + b + " ]");
        }
    }
    Cols c;
    memset(&c, 0, sizeof(c));
    c.n = ncols;
    int row = 0;
    for(int j = 0; j < ncols; j++) {
        MPG_CHECK(col_bytes[j] > 0 && d_cols[j], "mpg_dist_domain_exchange: bad column");
        c.src[j] = (const char *)d_cols[j];
        c.w[j] = col_bytes[j];
        c.off[j] = row;
        row += (col_bytes[j] + 7) / 8 * 8;
    }
    c.row = row;
    d->sendbuf.reserve((size_t)row * pl.nsend + 64);
    d->recvbuf.reserve((size_t)row * pl.nrecv + 64);
    if(pl.nsend > 0)
        hipLaunchKernelGGL(k_pack_cols, dim3(nblk(pl.nsend)), dim3(256), 0, st, pl.nsend, pl.idx.p, c, d->sendbuf.p);
    exchange_rows(d, pl, d->sendbuf.p, d->recvbuf.p, false, row);
    for(int j = 0; j < ncols; j++) {
        d->dom_out[j].reserve((size_t)col_bytes[j] * pl.nrecv + 64);
        c.dst[j] = d->dom_out[j].p;
        d_new_cols[j] = d->dom_out[j].p;
    }
    if(pl.nrecv > 0)
        hipLaunchKernelGGL(k_unpack_cols, dim3(nblk(pl.nrecv)), dim3(256), 0, st, pl.nrecv, (const char *)d->recvbuf.p, c);
    MPG_HIP(hipGetLastError());
    sync(d);
    *n_new = pl.nrecv;
    d->dom_n = -1; // the particle set has changed
    API_END
}

/* the decomposition just made becomes the domain of the force step (mpg_dist_set_domain with its own tree) */
int mpg_dist_use_decomposition(mpg_dist *d, double BoxSize, double margin, int La)
{
    if(!d || d->dom_size <= 0)
        return 1;
    return mpg_dist_set_domain(d, BoxSize, d->dom_tree.data(), d->dom_size, d->dom_leaf_task.data(), d->dom_nleaves, margin, La);
}

} // extern "C"

/* ---- fof_fof (fof.c:157-253) with the particles on their Peano-Hilbert owners ---------------------------------------------------
 * The reference labels every particle with the smallest ID it is linked to by repeated tree walks with exports until no label
 * changes anywhere, then ships parts of groups to one task per group and numbers the groups globally.  Here: ghosts within the
 * domain margin (>= the linking length) make every link of an own particle local; the components of the local set are found as
 * on one GPU (fof.hip); the labels of the ghosts are then compared with their owners' until nothing changes (a group that crosses
 * k domain boundaries needs about k rounds of one 8-byte-per-ghost exchange); the parts of the groups found among the OWN
 * particles go to rank MinID % NTask, which adds them up (fof_reduce_groups), drops groups below FOFHaloMinLength and finishes
 * their properties; one all-gather of (Length, MinID) numbers the groups as fof_assign_grnr does (length descending). */
namespace {

using GroupRec = mpg_dist::GroupRec;

inline double nearest(double x, double box) { return x > 0.5 * box ? x - box : (x < -0.5 * box ? x + box : x); }

// b (a part of the same group, sums about its own FirstPos) into a
void add_group_part(GroupRec &a, const GroupRec &b, double box)
{
    double s[3], mrel[3]; // shift of b's reference point into a's frame; sum m rel of b about b's FirstPos
    const double Mb = b.acc[0];
    for(int k = 0; k < 3; k++) {
        s[k] = nearest((double)b.FirstPos[k] - (double)a.FirstPos[k], box);
        mrel[k] = b.acc[7 + k] - Mb * (double)b.FirstPos[k];
    }
    a.Length += b.Length;
    for(int t = 0; t < 6; t++)
        a.LenType[t] += b.LenType[t];
    for(int c = 0; c < 7; c++)
        a.acc[c] += b.acc[c];
    const double *mv = b.acc + 10;
    const double sxmv[3] = {s[1] * mv[2] - s[2] * mv[1], s[2] * mv[0] - s[0] * mv[2], s[0] * mv[1] - s[1] * mv[0]};
    for(int k = 0; k < 3; k++) {
        a.acc[7 + k] += mrel[k] + Mb * (s[k] + (double)a.FirstPos[k]); // sum m (rel_b + s + FirstPos_a)
        a.acc[10 + k] += mv[k];
        a.acc[13 + k] += b.acc[13 + k] + sxmv[k];
        for(int e = 0; e < 3; e++)
            a.acc[16 + 3 * k + e] += b.acc[16 + 3 * k + e] + s[k] * mrel[e] + mrel[k] * s[e] + Mb * s[k] * s[e];
    }
}

// fof_finish_group_properties (fof.c:705-755), as k_fof_finish of fof.hip
void finish_group(GroupRec &g, double box)
{
    double *a = g.acc;
    const double M = a[0];
    double cm[3], rel[3], vcm[3];
    for(int d = 0; d < 3; d++) {
        a[10 + d] /= M;
        vcm[d] = a[10 + d];
        cm[d] = a[7 + d] / M;
        rel[d] = nearest(cm[d] - (double)g.FirstPos[d], box);
        while(cm[d] >= box)
            cm[d] -= box;
        while(cm[d] < 0)
            cm[d] += box;
        a[7 + d] = cm[d];
    }
    const double jcm[3] = {rel[1] * vcm[2] - rel[2] * vcm[1], rel[2] * vcm[0] - rel[0] * vcm[2], rel[0] * vcm[1] - rel[1] * vcm[0]};
    for(int d = 0; d < 3; d++)
        a[13 + d] -= jcm[d] * M;
    for(int d = 0; d < 3; d++)
        for(int e = 0; e < 3; e++)
            a[16 + 3 * d + e] -= M * rel[d] * rel[e];
}

// host alltoallv of records: out = what the other ranks sent here, in source-rank order
template <class T> void exchange_host_records(mpg_dist *d, const std::vector<std::vector<T>> &to, std::vector<T> &out)
{
    static_assert(sizeof(T) % 8 == 0, "records travel in 8-byte units");
    const int nt = d->nt;
    out.clear();
    if(nt == 1) {
        out = to[0];
        return;
    }
    std::vector<int64_t> sc(nt), rc(nt), sd(nt + 1, 0), rd(nt + 1, 0);
    for(int r = 0; r < nt; r++) {
        sc[r] = (int64_t)(to[r].size() * sizeof(T));
        sd[r + 1] = sd[r] + sc[r];
    }
    cb(d->comm.alltoall_i64(d->comm.ctx, sc.data(), rc.data()), "alltoall_i64");
    for(int r = 0; r < nt; r++)
        rd[r + 1] = rd[r] + rc[r];
    std::vector<char> sb((size_t)sd[nt] + 8), rb((size_t)rd[nt] + 8);
    for(int r = 0; r < nt; r++)
        if(sc[r])
            memcpy(sb.data() + sd[r], to[r].data(), (size_t)sc[r]);
    std::vector<int64_t> sdv(sd.begin(), sd.begin() + nt), rdv(rd.begin(), rd.begin() + nt);
    cb(d->comm.alltoallv(d->comm.ctx, sb.data(), sc.data(), sdv.data(), rb.data(), rc.data(), rdv.data(), 0), "alltoallv");
    out.resize((size_t)rd[nt] / sizeof(T));
    if(rd[nt])
        memcpy(out.data(), rb.data(), (size_t)rd[nt]);
}

} // namespace

extern "C" {

int mpg_dist_dev_fof_fof(mpg_dist *d, int64_t n_own, const double *d_pos, const float *d_mass, const uint8_t *d_type, const uint64_t *d_id,
                         const double *d_vel, const mpg_fof_params *par, int64_t *d_grnr, int64_t *ngroups_total, int64_t *ngroups_here)
{
    API_BEGIN
    MPG_CHECK(d && par && (n_own == 0 || (d_pos && d_mass && d_id)), "mpg_dist_dev_fof_fof: null argument");
    MPG_CHECK(d->have_domain, "mpg_dist_dev_fof_fof: mpg_dist_set_domain first");
    MPG_CHECK(par->FOFHaloComovingLinkingLength > 0 && par->FOFHaloMinLength >= 1, "fof_fof: bad parameters");
    MPG_CHECK((par->FOFPrimaryLinkTypes & par->FOFSecondaryLinkTypes) == 0, "fof_fof: primary and secondary link types must be disjoint");
    // primary links reach one linking length; the doubling search of the secondary attachment (0.4 LL, doubled while below 4 LL:
    // fof.c:1235-1239, 1286) ends at 6.4 linking lengths
    MPG_CHECK((par->FOFSecondaryLinkTypes ? 6.4 : 1.0) * par->FOFHaloComovingLinkingLength <= d->margin,
              "mpg_dist_dev_fof_fof: the domain margin must cover the linking length (6.4 linking lengths with secondary link types)");
    mpg_engine *e = d->eng;
    MPG_HIP(hipSetDevice(e->device));
    hipStream_t st = e->stream;
    const int nt = d->nt;
    // ---- ghosts with their IDs and types; the tree of the primary types over the local set
    const int64_t nl = import_ghosts(d, n_own, d_pos, d_mass);
    const Plan &pl = d->ghost;
    d->f_id.reserve((size_t)nl + 1);
    d->f_type.reserve((size_t)nl + 1);
    if(n_own > 0) {
        MPG_HIP(hipMemcpyAsync(d->f_id.p, d_id, (size_t)n_own * 8, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_fill_type, dim3(nblk(n_own)), dim3(256), 0, st, n_own, d_type, d->f_type.p);
    }
    d->sendbuf.reserve((size_t)16 * pl.nsend + 16);
    d->recvbuf.reserve((size_t)16 * pl.nrecv + 16);
    if(pl.nsend > 0)
        hipLaunchKernelGGL(k_pack_idtype, dim3(nblk(pl.nsend)), dim3(256), 0, st, pl.nsend, pl.idx.p, (const unsigned long long *)d_id, d_type,
                           (IdRow *)d->sendbuf.p);
    exchange_rows(d, pl, d->sendbuf.p, d->recvbuf.p, false, 16);
    if(pl.nrecv > 0)
        hipLaunchKernelGGL(k_unpack_idtype, dim3(nblk(pl.nrecv)), dim3(256), 0, st, pl.nrecv, (const IdRow *)d->recvbuf.p, d->f_id.p + n_own,
                           d->f_type.p + n_own);
    d->grav_tree_valid = false; // (the tree of the primary types replaces the gravity tree in the engine)
    MPG_CHECK(mpg_dev_bind_particles(e, nl, d->lpos.p, d->lmass.p, d->f_type.p, d->box) == 0, mpg_last_error());
    e->tree.build(nl, d->lpos.p, d->lmass.p, d->f_type.p, par->FOFPrimaryLinkTypes, d->box, st, &e->timer, nullptr);
    e->tree_allocated = true;
    e->tree_mask = par->FOFPrimaryLinkTypes;
    e->full_particle_tree = false;
    e->sph.hmax_pending = false;
    FofInput in;
    in.n = nl;
    in.pos = d->lpos.p;
    in.vel = nullptr;
    in.mass = d->lmass.p;
    in.type = d->f_type.p;
    in.flags = nullptr;
    in.id = d->f_id.p;
    in.hsml = nullptr;
    in.box = d->box;
    in.LL = par->FOFHaloComovingLinkingLength;
    in.minlen = par->FOFHaloMinLength;
    in.secondary_mask = par->FOFSecondaryLinkTypes;
    FofEngine &F = e->fof;
    F.compute_labels(e->tree, in, st);
    // ---- the labels of the ghosts against their owners', until no rank changes any
    const int64_t ng_ = pl.nrecv;
    d->f_ext.reserve((size_t)ng_ + 1);
    d->f_keys.reserve((size_t)ng_ + 1);
    d->f_vals.reserve((size_t)ng_ + 1);
    d->f_keys2.reserve((size_t)ng_ + 1);
    d->f_vals2.reserve((size_t)ng_ + 1);
    d->f_ukeys.reserve((size_t)ng_ + 1);
    d->f_uvals.reserve((size_t)ng_ + 1);
    d->f_flag.reserve((size_t)ng_ + 1);
    d->scount.reserve(4);
    int rounds = 0;
    for(;; rounds++) {
        MPG_CHECK(rounds < 10000, "mpg_dist_dev_fof_fof: the labels do not converge");
        d->sendbuf.reserve((size_t)8 * pl.nsend + 8);
        if(pl.nsend > 0)
            hipLaunchKernelGGL(k_gather_u64, dim3(nblk(pl.nsend)), dim3(256), 0, st, pl.nsend, pl.idx.p, F.label.p, (unsigned long long *)d->sendbuf.p);
        exchange_rows(d, pl, d->sendbuf.p, d->f_ext.p, false, 8);
        int64_t changed = 0;
        if(ng_ > 0) {
            hipLaunchKernelGGL(k_label_pairs, dim3(nblk(ng_)), dim3(256), 0, st, ng_, F.label.p + n_own, d->f_ext.p, d->f_keys.p, d->f_vals.p, d->f_flag.p);
            size_t tb = 0, tb2 = 0;
            MPG_HIP(rocprim::select(nullptr, tb, d->f_keys.p, d->f_flag.p, d->f_keys2.p, d->scount.p, (size_t)ng_, st));
            d->tmp.reserve(tb + 16);
            MPG_HIP(rocprim::select((void *)d->tmp.p, tb, d->f_keys.p, d->f_flag.p, d->f_keys2.p, d->scount.p, (size_t)ng_, st));
            MPG_HIP(rocprim::select((void *)d->tmp.p, tb, d->f_vals.p, d->f_flag.p, d->f_vals2.p, d->scount.p, (size_t)ng_, st));
            unsigned long long m = 0;
            MPG_HIP(hipMemcpyAsync(&m, d->scount.p, sizeof(m), hipMemcpyDeviceToHost, st));
            sync(d);
            if(m > 0) {
                // the smallest owner label per local label: sort by the local label, reduce by key with min
                MPG_HIP(rocprim::radix_sort_pairs(nullptr, tb2, d->f_keys2.p, d->f_keys.p, d->f_vals2.p, d->f_vals.p, (size_t)m, 0, 64, st));
                d->tmp.reserve(tb2 + 16);
                MPG_HIP(rocprim::radix_sort_pairs((void *)d->tmp.p, tb2, d->f_keys2.p, d->f_keys.p, d->f_vals2.p, d->f_vals.p, (size_t)m, 0, 64, st));
                size_t tb3 = 0;
                MPG_HIP(rocprim::reduce_by_key(nullptr, tb3, d->f_keys.p, d->f_vals.p, (unsigned)m, d->f_ukeys.p, d->f_uvals.p, d->scount.p + 1,
                                               rocprim::minimum<unsigned long long>(), rocprim::equal_to<unsigned long long>(), st));
                d->tmp.reserve(tb3 + 16);
                MPG_HIP(rocprim::reduce_by_key((void *)d->tmp.p, tb3, d->f_keys.p, d->f_vals.p, (unsigned)m, d->f_ukeys.p, d->f_uvals.p, d->scount.p + 1,
                                               rocprim::minimum<unsigned long long>(), rocprim::equal_to<unsigned long long>(), st));
                unsigned long long mu = 0;
                MPG_HIP(hipMemcpyAsync(&mu, d->scount.p + 1, sizeof(mu), hipMemcpyDeviceToHost, st));
                sync(d);
                hipLaunchKernelGGL(k_apply_label_map, dim3(nblk(nl)), dim3(256), 0, st, nl, F.label.p, d->f_ukeys.p, d->f_uvals.p, (int64_t)mu);
                changed = 1;
            }
        }
        allreduce_i64(d, &changed, 1, 0);
        if(changed == 0)
            break;
    }
    d->stats[6] = rounds + 1;
    // ---- the parts of the groups among the OWN particles: all parts of a label that a ghost carries too, and the large ones
    int64_t nglab = 0;
    if(ng_ > 0) {
        size_t tb = 0;
        MPG_HIP(rocprim::radix_sort_keys(nullptr, tb, F.label.p + n_own, d->f_keys.p, (size_t)ng_, 0, 64, st));
        d->tmp.reserve(tb + 16);
        MPG_HIP(rocprim::radix_sort_keys((void *)d->tmp.p, tb, F.label.p + n_own, d->f_keys.p, (size_t)ng_, 0, 64, st));
        d->f_glab.reserve((size_t)ng_ + 1);
        size_t tb2 = 0;
        MPG_HIP(rocprim::unique(nullptr, tb2, d->f_keys.p, d->f_glab.p, d->scount.p, (size_t)ng_, rocprim::equal_to<unsigned long long>(), st));
        d->tmp.reserve(tb2 + 16);
        MPG_HIP(rocprim::unique((void *)d->tmp.p, tb2, d->f_keys.p, d->f_glab.p, d->scount.p, (size_t)ng_, rocprim::equal_to<unsigned long long>(), st));
        unsigned long long c = 0;
        MPG_HIP(hipMemcpyAsync(&c, d->scount.p, sizeof(c), hipMemcpyDeviceToHost, st));
        sync(d);
        nglab = (int64_t)c;
    }
    in.vel = d_vel;
    const int64_t nparts = F.catalogue(in, n_own, d->f_glab.p, nglab, false, st);
    // ---- the parts to the rank that keeps the group (MinID % NTask), which adds them up
    std::vector<unsigned long long> h_minid((size_t)nparts + 1);
    std::vector<unsigned> h_len((size_t)nparts + 1);
    std::vector<int> h_lt((size_t)nparts * 6 + 6);
    std::vector<float> h_first((size_t)nparts * 3 + 3);
    std::vector<double> h_acc((size_t)nparts * 27 + 27);
    if(nparts > 0) {
        MPG_HIP(hipMemcpyAsync(h_minid.data(), F.g_minid.p, (size_t)nparts * 8, hipMemcpyDeviceToHost, st));
        MPG_HIP(hipMemcpyAsync(h_len.data(), F.g_len.p, (size_t)nparts * 4, hipMemcpyDeviceToHost, st));
        MPG_HIP(hipMemcpyAsync(h_lt.data(), F.g_lentype.p, (size_t)nparts * 6 * 4, hipMemcpyDeviceToHost, st));
        MPG_HIP(hipMemcpyAsync(h_first.data(), F.g_first.p, (size_t)nparts * 3 * 4, hipMemcpyDeviceToHost, st));
        MPG_HIP(hipMemcpyAsync(h_acc.data(), F.g_acc.p, (size_t)nparts * 27 * 8, hipMemcpyDeviceToHost, st));
        sync(d);
    }
    std::vector<std::vector<GroupRec>> to((size_t)nt);
    for(int64_t g = 0; g < nparts; g++) {
        GroupRec r;
        memset(&r, 0, sizeof(r));
        r.MinID = h_minid[g];
        r.Length = h_len[g];
        r.src = d->me;
        for(int t = 0; t < 6; t++)
            r.LenType[t] = h_lt[6 * g + t];
        for(int k = 0; k < 3; k++)
            r.FirstPos[k] = h_first[3 * g + k];
        memcpy(r.acc, h_acc.data() + 27 * g, 27 * sizeof(double));
        to[(size_t)(r.MinID % (unsigned long long)nt)].push_back(r);
    }
    std::vector<GroupRec> got;
    exchange_host_records(d, to, got);
    // parts of one group side by side, the lowest source rank first (its FirstPos becomes the group's: fof_reduce_base_group keeps the
    // first in the same way); the sums do not depend on the order beyond rounding
    std::stable_sort(got.begin(), got.end(), [](const GroupRec &a, const GroupRec &b) { return a.MinID != b.MinID ? a.MinID < b.MinID : a.src < b.src; });
    d->f_groups.clear();
    for(size_t k = 0; k < got.size();) {
        GroupRec g = got[k];
        size_t j = k + 1;
        for(; j < got.size() && got[j].MinID == g.MinID; j++)
            add_group_part(g, got[j], d->box);
        k = j;
        if(g.Length >= par->FOFHaloMinLength) { // fof.c:800-808
            finish_group(g, d->box);
            d->f_groups.push_back(g);
        }
    }
    // ---- global numbers: by (Length descending, MinID ascending), fof_assign_grnr (fof.c:1106-1155)
    struct LM {
        long long Length;
        unsigned long long MinID;
    };
    std::vector<LM> mine(d->f_groups.size());
    for(size_t k = 0; k < mine.size(); k++)
        mine[k] = LM{d->f_groups[k].Length, d->f_groups[k].MinID};
    std::vector<std::vector<LM>> toall((size_t)nt, mine);
    std::vector<LM> all;
    exchange_host_records(d, toall, all);
    std::vector<size_t> ord(all.size());
    for(size_t k = 0; k < ord.size(); k++)
        ord[k] = k;
    std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) {
        return all[a].Length != all[b].Length ? all[a].Length > all[b].Length : all[a].MinID < all[b].MinID;
    });
    std::vector<std::pair<unsigned long long, long long>> tab(all.size()); // (MinID, GrNr), then by MinID
    for(size_t k = 0; k < ord.size(); k++)
        tab[k] = {all[ord[k]].MinID, (long long)k + 1}; // group numbers start at 1 (fof.c:1150, as fof.hip's k_fof_grnr)
    std::sort(tab.begin(), tab.end());
    d->f_total = (int64_t)all.size();
    d->f_group_grnr.assign(d->f_groups.size(), -1);
    for(size_t k = 0; k < d->f_groups.size(); k++) {
        auto it = std::lower_bound(tab.begin(), tab.end(), std::make_pair(d->f_groups[k].MinID, (long long)-1));
        d->f_group_grnr[k] = it->second;
    }
    // ---- P[].GrNr of the own particles
    if(d_grnr && n_own > 0) {
        const size_t nall = tab.size();
        std::vector<unsigned long long> hk(nall + 1);
        std::vector<long long> hv(nall + 1);
        for(size_t k = 0; k < nall; k++) {
            hk[k] = tab[k].first;
            hv[k] = tab[k].second;
        }
        d->f_minid.reserve(nall + 1);
        d->f_grnr_tab.reserve(nall + 1);
        if(nall) {
            MPG_HIP(hipMemcpyAsync(d->f_minid.p, hk.data(), nall * 8, hipMemcpyHostToDevice, st));
            MPG_HIP(hipMemcpyAsync(d->f_grnr_tab.p, hv.data(), nall * 8, hipMemcpyHostToDevice, st));
        }
        hipLaunchKernelGGL(k_lookup_grnr, dim3(nblk(n_own)), dim3(256), 0, st, n_own, F.label.p, d->f_minid.p, d->f_grnr_tab.p, (int64_t)nall,
                           (long long *)d_grnr);
        sync(d);
    }
    if(ngroups_total)
        *ngroups_total = d->f_total;
    if(ngroups_here)
        *ngroups_here = (int64_t)d->f_groups.size();
    API_END
}

/* the groups this rank keeps after the last mpg_dist_dev_fof_fof (HOST arrays of *ngroups_here entries; NULL = not wanted) */
int mpg_dist_fof_groups(mpg_dist *d, const mpg_fof_groups *out)
{
    API_BEGIN
    MPG_CHECK(d && out, "null argument");
    for(size_t g = 0; g < d->f_groups.size(); g++) {
        const GroupRec &r = d->f_groups[g];
        if(out->MinID)
            out->MinID[g] = r.MinID;
        if(out->Length)
            out->Length[g] = (int)r.Length;
        if(out->GrNr)
            out->GrNr[g] = (int)d->f_group_grnr[g];
        if(out->Mass)
            out->Mass[g] = r.acc[0];
        for(int t = 0; t < 6; t++) {
            if(out->LenType)
                out->LenType[6 * g + t] = r.LenType[t];
            if(out->MassType)
                out->MassType[6 * g + t] = r.acc[1 + t];
        }
        for(int k = 0; k < 3; k++) {
            if(out->CM)
                out->CM[3 * g + k] = r.acc[7 + k];
            if(out->Vel)
                out->Vel[3 * g + k] = r.acc[10 + k];
            if(out->Jmom)
                out->Jmom[3 * g + k] = r.acc[13 + k];
            if(out->FirstPos)
                out->FirstPos[3 * g + k] = r.FirstPos[k];
        }
        if(out->Imom)
            for(int c = 0; c < 9; c++)
                out->Imom[9 * g + c] = r.acc[16 + c];
    }
    API_END
}

} // extern "C"

/* ---- the SPH loops as drop-in calls: the rank's table and the SPH fields in HOST arrays (what shim/sph-hip.c gathers from the
 * slots), after mpg_dist_force_tree_full on the same table ------------------------------------------------------------------ */
namespace {

struct SphField {
    int w;       // doubles per particle
    bool in, out_density, out_hydro;
};
// the double-valued members of mpg_sph_arrays in declaration order, without the two time-bin byte arrays
const SphField SPH_FIELDS[17] = {{1, true, true, false},   // hsml
                                 {1, false, true, false},  // dthsml
                                 {3, true, false, false},  // vel
                                 {3, true, false, false},  // gacc
                                 {3, true, false, false},  // gpm
                                 {3, true, false, false},  // hydroacc_in
                                 {1, true, false, false},  // entropy
                                 {1, true, false, false},  // dtentropy_in
                                 {1, false, true, false},  // density
                                 {1, false, true, false},  // egywtdensity
                                 {1, false, true, false},  // dhsmlegyfac
                                 {1, false, true, false},  // divvel
                                 {1, false, true, false},  // curlvel
                                 {3, false, true, false},  // gradrho
                                 {3, false, false, true},  // hydroacc_out
                                 {1, false, false, true},  // dtentropy_out
                                 {1, false, false, true}}; // maxsignalvel

// pointers of the 17 double fields of a mpg_sph_arrays, in the order of SPH_FIELDS
void sph_field_ptrs(const mpg_sph_arrays *A, const double *p[17])
{
    p[0] = A->hsml;
    p[1] = A->dthsml;
    p[2] = A->vel;
    p[3] = A->gacc;
    p[4] = A->gpm;
    p[5] = A->hydroacc_in;
    p[6] = A->entropy;
    p[7] = A->dtentropy_in;
    p[8] = A->density;
    p[9] = A->egywtdensity;
    p[10] = A->dhsmlegyfac;
    p[11] = A->divvel;
    p[12] = A->curlvel;
    p[13] = A->gradrho;
    p[14] = A->hydroacc_out;
    p[15] = A->dtentropy_out;
    p[16] = A->maxsignalvel;
}

// device copy of the host arrays: inputs uploaded, outputs allocated; returns the device-side struct
mpg_sph_arrays stage_sph(mpg_dist *d, const mpg_sph_arrays *A, int64_t n, bool upload, bool upload_density_out, bool upload_hydro_out)
{
    hipStream_t st = d->eng->stream;
    const double *hp[17];
    sph_field_ptrs(A, hp);
    double *dp[17];
    for(int k = 0; k < 17; k++) {
        dp[k] = nullptr;
        if(!hp[k])
            continue;
        d->o_sph[k].reserve((size_t)SPH_FIELDS[k].w * n + 3);
        dp[k] = d->o_sph[k].p;
        const bool up = (upload && SPH_FIELDS[k].in) || (upload_density_out && SPH_FIELDS[k].out_density) ||
                        (upload_hydro_out && SPH_FIELDS[k].out_hydro);
        if(up && n > 0)
            MPG_HIP(hipMemcpyAsync(dp[k], hp[k], (size_t)SPH_FIELDS[k].w * n * sizeof(double), hipMemcpyHostToDevice, st));
    }
    const uint8_t *hb[2] = {A->tb_hydro, A->tb_grav};
    uint8_t *db[2] = {nullptr, nullptr};
    for(int k = 0; k < 2; k++)
        if(hb[k]) {
            d->o_u8[1 + k].reserve((size_t)n + 1);
            db[k] = d->o_u8[1 + k].p;
            if(upload && n > 0)
                MPG_HIP(hipMemcpyAsync(db[k], hb[k], (size_t)n, hipMemcpyHostToDevice, st));
        }
    sync(d);
    mpg_sph_arrays D;
    memset(&D, 0, sizeof(D));
    D.hsml = dp[0];
    D.dthsml = dp[1];
    D.vel = dp[2];
    D.gacc = dp[3];
    D.gpm = dp[4];
    D.hydroacc_in = dp[5];
    D.tb_hydro = db[0];
    D.tb_grav = db[1];
    D.entropy = dp[6];
    D.dtentropy_in = dp[7];
    D.density = dp[8];
    D.egywtdensity = dp[9];
    D.dhsmlegyfac = dp[10];
    D.divvel = dp[11];
    D.curlvel = dp[12];
    D.gradrho = dp[13];
    D.hydroacc_out = dp[14];
    D.dtentropy_out = dp[15];
    D.maxsignalvel = dp[16];
    return D;
}

void download_sph(mpg_dist *d, const mpg_sph_arrays *A, int64_t n, bool hydro)
{
    hipStream_t st = d->eng->stream;
    const double *hp[17];
    sph_field_ptrs(A, hp);
    for(int k = 0; k < 17; k++) {
        const bool want = hydro ? SPH_FIELDS[k].out_hydro : SPH_FIELDS[k].out_density;
        if(want && hp[k] && n > 0)
            MPG_HIP(hipMemcpyAsync((double *)hp[k], d->o_sph[k].p, (size_t)SPH_FIELDS[k].w * n * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    sync(d);
}

// P[].Type (garbage has been refused by mpg_dist_force_tree_full) onto the device
const uint8_t *stage_types(mpg_dist *d, const mpg_particle_view *P)
{
    const int64_t n = P->n;
    std::vector<uint8_t> ty((size_t)n + 1);
    const mpg_particle_view V = *P;
    const char *b = (const char *)P->base;
    uint8_t *t = ty.data();
    parallel_for(n, [=](int64_t lo, int64_t hi) {
        for(int64_t i = lo; i < hi; i++)
        {
            t[i] = V.off_type >= 0 ? (uint8_t)(*(const uint8_t *)(b + i * V.stride + V.off_type) & 7) : (uint8_t)1;
            // garbage and swallowed black holes are no targets and no neighbours (density.c:521-530, forcetree.c:357-365): type 7
            if(V.off_flags >= 0 && (*(const uint8_t *)(b + i * V.stride + V.off_flags) & 3))
                t[i] = 7;
        }
    });
    d->o_u8[0].reserve((size_t)n + 1);
    if(n > 0)
        MPG_HIP(hipMemcpy(d->o_u8[0].p, ty.data(), (size_t)n, hipMemcpyHostToDevice));
    return d->o_u8[0].p;
}

} // namespace

extern "C" {

// ActiveParticle of a host call onto the device (null: all)
static const int *stage_active(mpg_dist *d, const int *ActiveParticle, int64_t n)
{
    if(!ActiveParticle)
        return nullptr;
    d->o_act.reserve((size_t)n + 1);
    if(n > 0)
        MPG_HIP(hipMemcpy(d->o_act.p, ActiveParticle, (size_t)n * sizeof(int), hipMemcpyHostToDevice));
    return d->o_act.p;
}

int mpg_dist_set_sph_options(mpg_dist *d, int BlackHoleOn)
{
    API_BEGIN
    MPG_CHECK(d, "null argument");
    d->blackholes = BlackHoleOn != 0;
    API_END
}

double mpg_dist_last_max_hsml(mpg_dist *d) { return d ? d->last_hmax : 0; }

int mpg_dist_domain_set_maxpart(mpg_dist *d, int64_t MaxPart)
{
    API_BEGIN
    MPG_CHECK(d && MaxPart >= 0, "null argument");
    d->dom_max_part = MaxPart;
    API_END
}

/* measure_power_spectrum + powerspectrum_sum over the ranks (gravpm.c:110-118, powerspectrum.c:55-91): every rank binned the k_y rows of
 * its slab during mpg_dist_(dev_)gravpm_force; the raw sums (Power, k, Norm, mode counts) are all-reduced as the reference's
 * MPI_Allreduce does, then normalised.  Collective; every rank gets the spectrum. */
int mpg_dist_gravpm_get_powerspectrum(mpg_dist *d, double BoxSize_in_MPC, double *kk, double *Power, int64_t *Nmodes, int *nonzero)
{
    API_BEGIN
    MPG_CHECK(d && kk && Power && Nmodes && nonzero, "null argument");
    mpg_engine *e = d->eng;
    MPG_CHECK(e->pm.nmesh > 0 && e->pm.ps_valid, "power spectrum: no PM step has been run with the measurement on");
    MPG_HIP(hipSetDevice(e->device));
    const size_t nb = (size_t)e->pm.nmesh;
    std::vector<double> acc(2 * nb + 1);
    std::vector<int64_t> modes(nb);
    MPG_HIP(hipMemcpyAsync(acc.data(), e->pm.ps_acc.p, (2 * nb + 1) * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    MPG_HIP(hipMemcpyAsync(modes.data(), e->pm.ps_modes.p, nb * sizeof(int64_t), hipMemcpyDeviceToHost, e->stream));
    sync(d);
    allreduce_host_f64(d, acc.data(), (int64_t)(2 * nb + 1), 0);
    allreduce_i64(d, modes.data(), (int64_t)nb, 0);
    MPG_CHECK(mpg_powerspectrum_sum((int)nb, acc.data(), modes.data(), BoxSize_in_MPC, kk, Power, Nmodes, nonzero) == 0, mpg_last_error());
    API_END
}

int mpg_dist_density(mpg_dist *d, const mpg_particle_view *P, const mpg_sph_arrays *A, const mpg_sph_times *T, const int *ActiveParticle,
                     int64_t NumActiveParticle, int update_hsml, int DoEgyDensity)
{
    API_BEGIN
    MPG_CHECK(d && P && A && T, "null argument");
    MPG_CHECK(d->o_n == P->n && d->n_own_tree == P->n, "mpg_dist_density: call mpg_dist_force_tree_full on this table first");
    MPG_HIP(hipSetDevice(d->eng->device));
    const uint8_t *ty = stage_types(d, P);
    // (a sub-step also uploads the density-loop results the inactive particles hold)
    const mpg_sph_arrays D = stage_sph(d, A, P->n, true, ActiveParticle != nullptr, false);
    const int *act = stage_active(d, ActiveParticle, NumActiveParticle);
    MPG_CHECK(mpg_dist_dev_density_active(d, P->n, ty, &D, T, act, NumActiveParticle, update_hsml, DoEgyDensity) == 0, mpg_last_error());
    download_sph(d, A, P->n, false);
    API_END
}

int mpg_dist_hydro_force(mpg_dist *d, const mpg_particle_view *P, const mpg_sph_arrays *A, const mpg_sph_times *T, const int *ActiveParticle,
                         int64_t NumActiveParticle)
{
    API_BEGIN
    MPG_CHECK(d && P && A && T, "null argument");
    MPG_CHECK(d->sph_n_own == P->n, "mpg_dist_hydro_force: call mpg_dist_density on this table first");
    MPG_HIP(hipSetDevice(d->eng->device));
    // (the inputs are the library's from the density call; a sub-step uploads the hydro results the inactive particles hold)
    const mpg_sph_arrays D = stage_sph(d, A, P->n, false, false, ActiveParticle != nullptr);
    const int *act = stage_active(d, ActiveParticle, NumActiveParticle);
    MPG_CHECK(mpg_dist_dev_hydro_force_active(d, P->n, &D, T, act, NumActiveParticle) == 0, mpg_last_error());
    download_sph(d, A, P->n, true);
    API_END
}

} // extern "C"

/* ---- hierarchical gravity on several ranks: the tree of the ACTIVE particles only (force_tree_active_moments, forcetree.c:129-148;
 * hierarchical_gravity_accelerations of timestep.c) -------------------------------------------------------------------------------
 * The active sets of the short time bins are sparse: the tree of such a set has cells with <= 8 particles far above any domain level,
 * which the all-reduced top of mpg_dist_dev_force_tree_build excludes.  They are also small.  So every rank receives the whole active
 * set (28 bytes per particle, one all-gather), builds ITS tree itself - the tree one GPU would build - and walks its own members. */
namespace {
struct InRange {
    int lo, hi;
    __host__ __device__ bool operator()(const unsigned &ci) const { return (int)ci >= lo && (int)ci < hi; }
};
} // namespace

extern "C" int mpg_dist_dev_grav_short_tree_active_tree(mpg_dist *d, int64_t n_act, const double *d_pos, const float *d_mass, const double *d_oldacc,
                                                        double *d_accel, double *d_potential, double rho0)
{
    API_BEGIN
    MPG_CHECK(d && (n_act == 0 || (d_pos && d_mass && d_accel)), "null argument");
    MPG_CHECK(d->box > 0, "mpg_dist: mpg_dist_set_domain first (the box size)");
    mpg_engine *e = d->eng;
    MPG_HIP(hipSetDevice(e->device));
    hipStream_t st = e->stream;
    // this rank's active particles as one host block: [n][3] positions, then [n] masses
    std::vector<char> blk(sizeof(int64_t) + (size_t)n_act * 28);
    memcpy(blk.data(), &n_act, sizeof(int64_t));
    if(n_act > 0) {
        MPG_HIP(hipMemcpyAsync(blk.data() + 8, d_pos, (size_t)n_act * 24, hipMemcpyDeviceToHost, st));
        MPG_HIP(hipMemcpyAsync(blk.data() + 8 + (size_t)n_act * 24, d_mass, (size_t)n_act * 4, hipMemcpyDeviceToHost, st));
    }
    sync(d);
    std::vector<std::vector<char>> all;
    allgather_host(d, blk.data(), (int64_t)blk.size(), all);
    int64_t ntot = 0, off = 0;
    std::vector<int64_t> cnt(d->nt);
    for(int r = 0; r < d->nt; r++) {
        memcpy(&cnt[r], all[r].data(), sizeof(int64_t));
        if(r < d->me)
            off += cnt[r];
        ntot += cnt[r];
    }
    MPG_CHECK(ntot < (1ll << 27), "mpg_dist_dev_grav_short_tree_active_tree: the active set is not small (use the tree of all particles)");
    d->n_own_tree = -1; // the local tree of mpg_dist_dev_force_tree_build is gone after this call
    d->sph_n_own = -1;
    if(ntot == 0)
        return 0;
    d->lpos.reserve(3 * (size_t)ntot + 3);
    d->lmass.reserve((size_t)ntot + 1);
    int64_t at = 0;
    for(int r = 0; r < d->nt; r++) {
        if(cnt[r] > 0) {
            MPG_HIP(hipMemcpyAsync(d->lpos.p + 3 * at, all[r].data() + 8, (size_t)cnt[r] * 24, hipMemcpyHostToDevice, st));
            MPG_HIP(hipMemcpyAsync(d->lmass.p + at, all[r].data() + 8 + (size_t)cnt[r] * 24, (size_t)cnt[r] * 4, hipMemcpyHostToDevice, st));
        }
        at += cnt[r];
    }
    sync(d); // (`all` is read by the copies)
    d->grav_tree_valid = false;
    MPG_CHECK(mpg_dev_bind_particles(e, ntot, d->lpos.p, d->lmass.p, nullptr, d->box) == 0, mpg_last_error());
    MPG_CHECK(mpg_dev_force_tree_build(e, 63) == 0, mpg_last_error());
    e->full_particle_tree = false; // (force_tree_active_moments, forcetree.c:129-148: P[].Potential and FullTreeGravAccel are not this walk's)
    if(n_act == 0)
        return 0;
    // own members in tree order; inputs / outputs over the whole set, the own range copied in and out
    d->targets.reserve((size_t)ntot + 1);
    d->scount.reserve(4);
    size_t tb = 0;
    const InRange mine{(int)off, (int)(off + n_act)};
    MPG_HIP(rocprim::select(nullptr, tb, e->tree.idx_b.p, (int *)d->targets.p, d->scount.p, (size_t)e->tree.npart, mine, st));
    d->tmp.reserve(tb + 16);
    MPG_HIP(rocprim::select((void *)d->tmp.p, tb, e->tree.idx_b.p, (int *)d->targets.p, d->scount.p, (size_t)e->tree.npart, mine, st));
    d->o_prev.reserve((size_t)ntot + 1);
    d->o_acc.reserve(3 * (size_t)ntot + 3);
    d->o_pot.reserve((size_t)ntot + 1);
    MPG_HIP(hipMemsetAsync(d->o_prev.p, 0, (size_t)ntot * sizeof(double), st));
    if(d_oldacc)
        MPG_HIP(hipMemcpyAsync(d->o_prev.p + off, d_oldacc, (size_t)n_act * sizeof(double), hipMemcpyDeviceToDevice, st));
    (void)d_potential; // (left alone: the tree does not hold every particle, gravshort.h:57-67)
    MPG_CHECK(mpg_dev_grav_short_tree(e, d->o_prev.p, nullptr, nullptr, d->targets.p, n_act, d->o_acc.p, nullptr, rho0) == 0, mpg_last_error());
    MPG_HIP(hipMemcpyAsync(d_accel, d->o_acc.p + 3 * off, (size_t)n_act * 24, hipMemcpyDeviceToDevice, st));
    sync(d);
    d->o_n = -1; // (the staging columns of the host drop-in calls were reused)
    API_END
}

/* the drop-in form: the active particles are gathered from the rank's table, AccelStore[i] of the active particles is assigned
 * (grav_short_reduce / _postprocess with a tree that is not the full particle tree: P[] itself is left alone) */
extern "C" int mpg_dist_grav_short_tree_active_tree(mpg_dist *d, const mpg_particle_view *P, const int *ActiveParticle, int64_t NumActiveParticle,
                                                    double (*AccelStore)[3], double rho0)
{
    API_BEGIN
    MPG_CHECK(d && P && AccelStore, "null argument (the walk on an active-only tree returns its result in AccelStore)");
    MPG_CHECK(P->off_pos >= 0 && P->off_mass >= 0 && P->off_accel >= 0 && P->off_gravpm >= 0, "particle view needs Pos, Mass, FullTreeGravAccel, GravPM");
    const int64_t nlist = ActiveParticle ? NumActiveParticle : P->n;
    MPG_CHECK(nlist >= 0 && nlist <= P->n, "bad NumActiveParticle");
    MPG_HIP(hipSetDevice(d->eng->device));
    hipStream_t st = d->eng->stream;
    // the live members of the list (garbage and swallowed particles are skipped in place: treewalk.c:234, forcetree.c:806)
    std::vector<int64_t> live;
    live.reserve((size_t)nlist);
    for(int64_t k = 0; k < nlist; k++) {
        const int64_t i = ActiveParticle ? ActiveParticle[k] : k;
        MPG_CHECK(i >= 0 && i < P->n, "ActiveParticle index out of range");
        if(P->off_flags >= 0 && (*((const uint8_t *)P->base + i * P->stride + P->off_flags) & 3))
            continue;
        live.push_back(i);
    }
    const int64_t n = (int64_t)live.size();
    const int64_t *lv = live.data();
    std::vector<double> hp(3 * (size_t)n + 3), ho((size_t)n + 1);
    std::vector<float> hm((size_t)n + 1);
    const mpg_particle_view V = *P;
    const char *b = (const char *)P->base;
    const double G = d->eng->pm.G;
    double *pp = hp.data(), *po = ho.data();
    float *pm = hm.data();
    parallel_for(n, [=](int64_t lo, int64_t hi) {
        for(int64_t k = lo; k < hi; k++) {
            const int64_t i = lv[k];
            const char *rec = b + i * V.stride;
            const double *x = (const double *)(rec + V.off_pos), *a = (const double *)(rec + V.off_accel), *g = (const double *)(rec + V.off_gravpm);
            double s2 = 0;
            for(int j = 0; j < 3; j++) {
                pp[3 * k + j] = x[j];
                s2 += (a[j] + g[j]) * (a[j] + g[j]);
            }
            pm[k] = *(const float *)(rec + V.off_mass);
            po[k] = sqrt(s2) / G; // grav_get_abs_accel, gravshort.h:70-80
        }
    });
    DevBuf<double> dp, dold, dacc;
    DevBuf<float> dm;
    dp.reserve(3 * (size_t)n + 3);
    dold.reserve((size_t)n + 1);
    dacc.reserve(3 * (size_t)n + 3);
    dm.reserve((size_t)n + 1);
    if(n > 0) {
        MPG_HIP(hipMemcpyAsync(dp.p, pp, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
        MPG_HIP(hipMemcpyAsync(dold.p, po, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
        MPG_HIP(hipMemcpyAsync(dm.p, pm, (size_t)n * sizeof(float), hipMemcpyHostToDevice, st));
    }
    sync(d);
    MPG_CHECK(mpg_dist_dev_grav_short_tree_active_tree(d, n, dp.p, dm.p, dold.p, dacc.p, nullptr, rho0) == 0, mpg_last_error());
    if(n > 0)
        MPG_HIP(hipMemcpyAsync(pp, dacc.p, 3 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
    sync(d);
    parallel_for(n, [=](int64_t lo, int64_t hi) {
        for(int64_t k = lo; k < hi; k++) {
            const int64_t i = lv[k];
            for(int j = 0; j < 3; j++)
                AccelStore[i][j] = pp[3 * k + j];
        }
    });
    API_END
}
