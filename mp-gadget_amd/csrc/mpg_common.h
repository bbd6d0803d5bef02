// mpg_common.h -- shared declarations of the gfx950 engine (internal; the public surface is include/mpgadget_hip.h)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

namespace mpg {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

[[noreturn]] inline void fail(const char *file, int line, const std::string &msg)
{
    char buf[64];
    snprintf(buf, sizeof(buf), " [%s:%d]", file, line);
    throw Error(msg + buf);
}

#define MPG_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if(_e != hipSuccess)                                                                            \
            ::mpg::fail(__FILE__, __LINE__, std::string("HIP error: ") + hipGetErrorString(_e) + " in " #expr); \
    } while(0)

#define MPG_CHECK(cond, msg)                            \
    do {                                                \
        if(!(cond))                                     \
            ::mpg::fail(__FILE__, __LINE__, (msg));     \
    } while(0)

// Growable device buffer (hipMalloc is slow: buffers are kept and only ever grown).
template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    void reserve(size_t n)
    {
        if(n <= cap)
            return;
        release();
        size_t want = n + n / 16 + 64;
        MPG_HIP(hipMalloc((void **)&p, want * sizeof(T)));
        cap = want;
        // debugging aid: MPG_POISON=1 fills every fresh allocation with 0xFF bytes (NaN as double, -1 as int), so that a kernel
        // reading memory nobody wrote shows up at once instead of depending on what the allocation held before
        static const bool poison = getenv("MPG_POISON") != nullptr;
        if(poison) { // (the fill runs on the null stream: finish it before a kernel on another stream writes the buffer)
            MPG_HIP(hipMemset(p, 0xff, want * sizeof(T)));
            MPG_HIP(hipDeviceSynchronize());
        }
    }
    void release()
    {
        if(p)
            (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
};

constexpr int MAXLEVEL = 21;      // octree levels below the root carried in a 63-bit key
constexpr int NMAXCHILD = 8;      // particles per leaf, forcetree.h:13
constexpr int NTAB = 512;         // rows of the short-range window table, gravity.c:16

// 32-byte source record: a particle (x,y,z,mass) or a node's moments (cofm,mass).
struct alignas(32) Src4 {
    double x, y, z, m;
};
// Node geometry (geometric centre and side length), 32 bytes.
struct alignas(32) NodeGeo {
    double cx, cy, cz, len;
};
// Node links, 16 bytes.  Nodes are stored in depth-first pre-order: an internal node's first child is node+1.
struct alignas(16) NodeLink {
    int sibling; // next node when this one is skipped or used (-1: end of walk); NODE.sibling of forcetree.h:39
    int pstart;  // first particle (tree order) below this node
    int pcount;  // >0: leaf with that many particles (NodeChild.noccupied); 0: internal node
    int level;   // depth (root = 0)
};

// Node links of the level-ordered ("children contiguous") copy of the tree used by the cooperative walk.
struct alignas(16) NodeLinkB {
    // internal node: level-order index of the first child; the children are firstchild .. firstchild+nchild-1.
    // LEAF: which opened leaves among its SIBLINGS may be listed as one run of <= 8 particles by the neighbour search (ngb_walk.h,
    // walk_stepk<MERGE>; sibling leaves are contiguous in tree order): bits 0-3 the particle count of this child's pair (children 2j,
    // 2j+1 of the parent), bits 4-7 of its quad, bits 8-11 of all children - each 0 unless every existing child of the set is a leaf,
    // there are at least two, and they hold <= 8 particles together
    int firstchild;
    int nchild;     // 1..8 occupied octants (0 for a leaf)
    int pstart;     // first particle (tree order) below this node
    int pcount;     // >0: leaf with that many particles; 0: internal node
};

struct TreeView {
    int64_t npart = 0;       // particles in the tree (tree order [0,npart))
    int64_t nnodes = 0;
    const Src4 *src = nullptr;     // [npart + nnodes]: particles then node moments
    const NodeGeo *geo = nullptr;  // [nnodes]
    const NodeLink *link = nullptr;// [nnodes]
    const double *hmax = nullptr;  // [nnodes] or null
    const int *order = nullptr;    // [npart] tree order -> caller index
    // level-ordered copy (root = 0, the children of a node are contiguous): geometry, moments, links, hmax
    const NodeGeo *geoB = nullptr;
    const Src4 *momB = nullptr;
    const NodeLinkB *linkB = nullptr;
    const double *hmaxB = nullptr;
    // level-ordered SEARCH geometry of the SPH loops (TreeBuilder::calc_search_boxes / calc_search_hsmax), or null: per node the cube
    // around the PARTICLES it holds instead of its cell, and the largest smoothing length among them
    const NodeGeo *geoS = nullptr;
    const double *hsmaxS = nullptr;
    // search links of the SPH loops (TreeBuilder::calc_search_links), or null: linkB with the small internal nodes as leaves
    const NodeLinkB *linkS = nullptr;
    // the leaves' particles in blocks of 8 records by level-order node number, zero-mass filled (TreeBuilder::ensure_leaf_pad), or null
    const Src4 *srcL = nullptr;
    double box = 0;
};

struct GravParams {
    double box, invbox;
    double rcut, rcut2;
    double h, hinv, h3inv; // FORCE_SOFTENING and powers
    double h2;             // h * h (a kernel argument, i.e. a scalar register: computed in the kernel it occupied a vector register pair)
    double inv_cell_dx;    // 1 / (cellsize * table dx)
    double errtol;         // ErrTolForceAcc
    double bhangle2;       // opening angle squared in effect for this walk
    double G;
    double cbrtrho0;
    int use_bh;            // TreeUseBH != 0
    int full_tree;
};

} // namespace mpg
