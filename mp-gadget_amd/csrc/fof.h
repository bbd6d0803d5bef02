// fof.h -- friends-of-friends group finder (see fof.hip)
#pragma once
#include "mpg_common.h"
#include "tree_build.h"

namespace mpg {

struct FofInput {
    int64_t n = 0;                       // particles (caller order)
    const double *pos = nullptr;         // [n][3]
    const double *vel = nullptr;         // [n][3] or null
    const float *mass = nullptr;         // [n]
    const uint8_t *type = nullptr;       // [n] or null (all type 1)
    const uint8_t *flags = nullptr;      // bit 0 IsGarbage, bit 1 Swallowed; or null
    const unsigned long long *id = nullptr; // [n] P[].ID
    const double *hsml = nullptr;        // [n] or null: the search-radius hint of gas / stars / black holes (fof.c:1285-1289)
    double box = 0, LL = 0;              // FOFHaloComovingLinkingLength
    int minlen = 32;                     // FOFHaloMinLength
    int secondary_mask = 1 + 16 + 32;    // FOFSecondaryLinkTypes
};

// device arrays to receive the group table (any may be null); groups are in MinID order like fof.Group
struct FofTable {
    unsigned long long *MinID;
    int *Length, *GrNr, *LenType; // LenType[g][6]
    double *Mass, *MassType;      // MassType[g][6]
    double *CM, *Vel, *Jmom;      // [g][3]
    double *Imom;                 // [g][9]
    float *FirstPos;              // [g][3]
};

struct FofEngine {
    int64_t ngroups = 0;
    DevBuf<int> parent, root_of, val, list, sidx, ord_a, ord_b, g_grnr, g_lentype;
    DevBuf<unsigned long long> minid, label, slabel, run_label, g_minid, cnt;
    DevBuf<unsigned> run_count, run_start, g_len, g_start, lenkey_a, lenkey_b, err;
    DevBuf<uint8_t> keep;
    DevBuf<float> g_first;
    DevBuf<double> g_acc;
    DevBuf<long long> p_grnr;
    DevBuf<char> tmp;
    // the tree must hold the particles of the primary link types (force_tree_rebuild_mask); returns the number of groups
    int64_t run(TreeBuilder &tree, const FofInput &in, hipStream_t st);
    // the two halves of run(), used separately when groups span ranks (dist.hip): labels of all in.n particles; then the catalogue
    // of the first n of them (also_keep: sorted labels whose runs are reported whatever their length; finish = false leaves raw sums)
    void compute_labels(TreeBuilder &tree, const FofInput &in, hipStream_t st);
    int64_t catalogue(const FofInput &in, int64_t n, const unsigned long long *also_keep, int64_t nalso, bool finish, hipStream_t st);
    static constexpr int NQ_ = 27; // doubles per group in g_acc (fof.hip NQ)
    void export_groups(const FofTable &out, hipStream_t st);
};

} // namespace mpg
