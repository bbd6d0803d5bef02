"""MT19937 as GSL exposes it (`gsl_rng_mt19937`) -- TEST INFRASTRUCTURE (oracle/README.md): used only to regenerate the
"random" particle sets of the reference's own tests (libgadget/tests/test_gravity.c:283-305, test_density.c:206-235,
test_forcetree.c:358-384), which draw from `gsl_rng_mt19937` seeded with `gsl_rng_set(r, 0)`.

GSL is not in /root/reference (system package); what is restated here is the published algorithm
(Matsumoto & Nishimura 1998, 2002 initialisation `init_genrand`) with GSL's two documented conventions:
  * `gsl_rng_set(r, 0)` selects the generator's default seed 4357;
  * `gsl_rng_uniform(r)` of this generator is one 32-bit draw divided by 2^32 (NOT numpy's 53-bit double from two draws).
tests/test_oracle_kat.py pins it to the algorithm's published outputs and to numpy's independent implementation of the
same 32-bit stream."""
import numpy as np

_N, _M = 624, 397
_UPPER, _LOWER = np.uint32(0x80000000), np.uint32(0x7fffffff)
_A = np.uint32(0x9908b0df)


class GslMT19937:
    def __init__(self, seed=0):
        self.set(seed)

    def set(self, seed):
        s = int(seed) & 0xffffffff
        if s == 0:
            s = 4357                     # gsl: "the seed 0 is replaced by the default seed"
        mt = np.empty(_N, np.uint64)
        mt[0] = s
        for i in range(1, _N):           # init_genrand (2002): mt[i] = 1812433253 * (mt[i-1] ^ (mt[i-1] >> 30)) + i
            p = int(mt[i - 1])
            mt[i] = (1812433253 * (p ^ (p >> 30)) + i) & 0xffffffff
        self.mt = mt.astype(np.uint32)
        self.pos = _N

    def _twist(self):
        mt = self.mt
        # the recurrence reads mt[k+1] (old) and mt[k+M] (new once k+M >= N): three dependency-free segments
        def seg(lo, hi, src_lo):
            y = (mt[lo:hi] & _UPPER) | (mt[lo + 1:hi + 1] & _LOWER)
            mt[lo:hi] = mt[src_lo:src_lo + (hi - lo)] ^ (y >> np.uint32(1)) ^ np.where(y & np.uint32(1), _A, np.uint32(0))
        seg(0, _N - _M, _M)                              # k in [0, 227): uses old mt[k+397]
        seg(_N - _M, 2 * (_N - _M), 0)                   # k in [227, 454): uses new mt[k-227]
        seg(2 * (_N - _M), _N - 1, _N - _M)              # k in [454, 623): uses new mt[k-227]
        y = (mt[_N - 1] & _UPPER) | (mt[0] & _LOWER)
        mt[_N - 1] = mt[_M - 1] ^ (y >> np.uint32(1)) ^ (_A if (y & np.uint32(1)) else np.uint32(0))
        self.pos = 0

    def get(self, count):
        """the next `count` 32-bit outputs (uint32 array)"""
        out = np.empty(count, np.uint32)
        o = 0
        while o < count:
            if self.pos >= _N:
                self._twist()
            k = min(count - o, _N - self.pos)
            y = self.mt[self.pos:self.pos + k].copy()
            y ^= y >> np.uint32(11)
            y ^= (y << np.uint32(7)) & np.uint32(0x9d2c5680)
            y ^= (y << np.uint32(15)) & np.uint32(0xefc60000)
            y ^= y >> np.uint32(18)
            out[o:o + k] = y
            o += k
            self.pos += k
        return out

    def uniform(self, count):
        """gsl_rng_uniform: [0, 1), one 32-bit draw / 2^32 each"""
        return self.get(count).astype(np.float64) / 4294967296.0


def three_population_set(rng, numpart, box):
    """The particle set of do_random_test (test_gravity.c:283-305 = test_density.c:206-235 = test_forcetree.c:358-384): draws in
    particle order, x y z per particle; a quarter uniform, half around Box/2 (width Box/8), a quarter around 0.1 Box (Box/32)."""
    u = rng.uniform(3 * numpart).reshape(numpart, 3)
    pos = np.empty((numpart, 3))
    a, b = numpart // 4, 3 * numpart // 4
    pos[:a] = box * u[:a]
    pos[a:b] = box / 2 + box / 8 * np.exp((u[a:b] - 0.5) ** 2)
    pos[b:] = box * 0.1 + box / 32 * np.exp((u[b:] - 0.5) ** 2)
    return pos
