// peano.h -- Peano-Hilbert keys and the (type, key) particle order on the device (see peano.hip)
#pragma once
#include "mpg_common.h"

namespace mpg {
struct PeanoScratch {
    DevBuf<uint64_t> keys_b;
    DevBuf<int> idx_a, idx_b;
    DevBuf<uint8_t> tk_a, tk_b;
    DevBuf<char> tmp;
    DevBuf<unsigned long long> cnt;
};
void launch_peano_keys(int64_t n, const double *pos, double box, uint64_t *keys, hipStream_t st);
// perm[k] = index of the k-th particle in (TypeKey, Key) order, TypeKey = Type or 255 for garbage (slotsmanager.c:404-452);
// returns the number of live (non-garbage) particles
int64_t order_by_type_and_key(int64_t n, const uint8_t *type, const uint8_t *flags, const uint64_t *keys, int *perm, PeanoScratch &ws,
                              hipStream_t st);
} // namespace mpg
