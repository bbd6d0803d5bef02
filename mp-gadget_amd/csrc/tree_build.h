// tree_build.h -- device oct-tree (see tree_build.hip)
#pragma once
#include "mpg_common.h"
#include "../../include/mpgadget_hip.h"

namespace mpg {

// HIP-event stopwatch on the engine stream; fills mpg_phase_times fields.
struct EventTimer {
    mpg_phase_times t{};
    bool enabled = false;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    void init()
    {
        if(!e0) {
            MPG_HIP(hipEventCreate(&e0));
            MPG_HIP(hipEventCreate(&e1));
        }
    }
    void start(hipStream_t st)
    {
        if(!enabled)
            return;
        init();
        MPG_HIP(hipEventRecord(e0, st));
    }
    // store elapsed since the previous start/lap into *dst, then restart
    void lap(hipStream_t st, float *dst)
    {
        if(!enabled)
            return;
        MPG_HIP(hipEventRecord(e1, st));
        MPG_HIP(hipEventSynchronize(e1));
        float ms = 0;
        MPG_HIP(hipEventElapsedTime(&ms, e0, e1));
        *dst = ms;
        MPG_HIP(hipEventRecord(e0, st));
    }
    ~EventTimer()
    {
        if(e0)
            (void)hipEventDestroy(e0);
        if(e1)
            (void)hipEventDestroy(e1);
    }
};

struct TreeBuilder {
    // inputs of the last build
    int64_t ncaller = 0; // particles offered
    int64_t npart = 0;   // particles in the tree (mask)
    int64_t nnodes = 0;
    int maxlevel = 0;
    int minleaflevel = 0; // depth of the shallowest leaf (largest leaf side = 1.001 Box / 2^minleaflevel)
    int force_internal_above = 0; // domain-decomposed runs: cells above this level are never leaves (their local particle sets are incomplete)
    double box = 0;
    bool has_moments = false, has_hmax = false;

    DevBuf<uint64_t> keys_a, keys_b;
    DevBuf<uint32_t> idx_a, idx_b; // idx_b: tree order -> caller index
    DevBuf<uint32_t> hi_a, hi_b, pos_a, pos_b; // short sort: top key bits and positions (tree_build.hip)
    DevBuf<uint8_t> leaflevel;
    DevBuf<uint32_t> cnt, base, node_head; // (node_head: the leaf head each node starts at, round 6)
    DevBuf<int64_t> flags;
    DevBuf<int> wave_ext; // per-wave leaf-level extrema of k_leaflevel
    DevBuf<char> tmp;
    DevBuf<Src4> src;
    DevBuf<NodeGeo> geo;
    DevBuf<NodeLink> link;
    DevBuf<double> hmax;
    // level-ordered copy for the cooperative walk (children of a node contiguous)
    DevBuf<uint32_t> lvl_a, lvl_b, nid_a, nid_b, bfs_of_dfs;
    DevBuf<NodeGeo> geoB;
    DevBuf<Src4> momB;
    DevBuf<NodeLinkB> linkB;
    DevBuf<double> hmaxB;
    bool has_bfs = false;
    // search geometry of the SPH loops (level order): tight cubes around each node's particles, largest Hsml per node
    DevBuf<double> aabb, hsmax;
    DevBuf<NodeGeo> geoS;
    DevBuf<double> hsmaxS;
    bool has_boxes = false, has_hsmax = false;
    // the leaves' particles as blocks of 8 source records indexed by the leaf's level-order node number, short leaves filled up with
    // zero-mass records (the evaluation kernel of the two-kernel gravity walk, grav_walk_split.hip): [(nnodes + 1) * 8], the last block
    // all zero-mass
    DevBuf<Src4> srcL;
    bool has_leaf_pad = false;
    // search links of the SPH loops (level order): linkB with every internal node of <= slink_cap particles turned into a leaf
    DevBuf<NodeLinkB> linkS;
    bool has_slinks = false;
    int slink_cap = 0;

    // force_tree_build (forcetree.c:196-270) without moments
    void build(int64_t n, const double *d_pos, const float *d_mass, const uint8_t *d_type, int mask, double box, hipStream_t st,
               EventTimer *tm, const uint8_t *d_include = nullptr);
    // force_tree_calc_moments (forcetree.c:170-183).  d_hsml_gasbh_treeorder: per tree-order particle, Hsml of
    // gas/BH particles that are not hydro-active, negative otherwise; NULL = no hmax.
    void calc_moments(const double *d_hsml_gasbh_treeorder, hipStream_t st, EventTimer *tm);
    // hmax only (tree built without moments, force_tree_rebuild_mask + update_tree_hmax_father + calc_moments)
    void calc_hmax(const double *d_hsml_gasbh_treeorder, hipStream_t st);
    // build / refresh the level-ordered copy from the depth-first arrays (after moments and/or hmax are known)
    void make_level_order(hipStream_t st);
    // domain-decomposed runs: own-particle sums of the level-(La-1) cells / moments of the nodes above level La from global sums
    void top_partial(int La, int64_t n_own, double *d_out, hipStream_t st);
    // d_flag_later: a zeroed device word the kernel raises instead of the check + synchronisation here (the caller reads it later)
    void top_set(int La, const double *d_sums, hipStream_t st, int *d_flag_later = nullptr);
    void ensure_level_order(hipStream_t st);
    // srcL (above) for the current tree; needs the level-ordered copy.  Positions and masses only: valid until the next build.
    void ensure_leaf_pad(hipStream_t st);
    // The neighbour searches of the SPH loops need, per node, ANY region that contains the node's particles: the reference tests the
    // node's cell (cull_node, treewalk.c:1015-1042); the cube around the particles themselves is contained in it and lets a search drop
    // leaves (a cell split at its 9th particle leaves children of one or two: points and short segments inside cells a mean spacing wide)
    // and whole branches that the cell test opens in vain.  The NEIGHBOUR set is unchanged - a particle within the search radius keeps every
    // node above it alive under either test -, the candidates tested are fewer.  calc_search_boxes: cubes of the current tree (positions
    // only; once per build).  calc_search_hsmax: the largest Hsml below each node, the symmetric search's radius (hydro), where the
    // reference has `hmax`, the reach beyond the CELL's faces (forcetree.c:963).
    void calc_search_boxes(hipStream_t st);
    void calc_search_hsmax(const double *d_hsml_treeorder, hipStream_t st);
    // The reference's tree splits a cell at its 9th particle, which leaves children of one or two particles each: a neighbour search that
    // descends to them tests 8 children to list a few short runs.  The particles below any node are contiguous in tree order, so a search
    // may stop at a node of <= cap particles and list its whole range in runs of 8 (walk_stepk, ngb_walk.h): fewer search steps and fuller
    // test lanes for more candidates; the neighbour set is unchanged (every particle below a node that is kept is tested).  Topology only:
    // valid until the next build.
    void calc_search_links(int cap, hipStream_t st);
    TreeView view() const;
};

} // namespace mpg
