"""Deterministic synthetic particle sets.

These are the build's own generators (the reference's IC code MP-GenIC needs PFFT+GSL and is out of
scope); identical bytes are fed to the HIP engine and to the CPU oracle.

  s_grid   perturbed grid, xorshift64 seed 1234567 (the construction of SURVEY App. C.5)
  s_clust  three-population clustered set (construction of libgadget/tests/test_gravity.c:283-305)
  s_zel    Zel'dovich-displaced grid, seed 181170 (construction of libgenic/zeldovich.c:208-262:
           grid + psi, psi_k = i k / k^2 delta_k) with a power-law spectrum
"""
import numpy as np

_MASK = (1 << 64) - 1


def xorshift64_uniform(count, seed=1234567, skip=0):
    """u = (s >> 11) * 2^-53 with s ^= s<<13; s ^= s>>7; s ^= s<<17 (64-bit), vectorised by jumping:
    the stream is strictly sequential, so generate it with a small blocked python loop over numpy uint64.
    skip: number of draws of the stream to jump over first (GF(2) jump, O(64^2 log skip))."""
    if skip:
        seed = _apply_gf2(_xorshift_jump_matrix(int(skip)), int(seed))
    out = np.empty(count, dtype=np.float64)
    s = np.uint64(seed)
    # sequential recurrence; ~1e7 draws/s is not reachable in pure python, so use the linear (GF(2)) structure:
    # xorshift is linear over GF(2) -> state_k = M^k state_0.  We step K independent lanes that are
    # offset by `stride` positions each, computed with one slow pass of `stride` steps.
    K = 4096 if count >= 1 << 16 else 1
    if K == 1:
        v = int(seed)
        for i in range(count):
            v ^= (v << 13) & _MASK
            v ^= v >> 7
            v ^= (v << 17) & _MASK
            out[i] = (v >> 11) * 2.0 ** -53
        return out
    stride = (count + K - 1) // K
    # lane starting states: state after j*stride steps, obtained by one sequential pass that records them
    starts = np.empty(K, dtype=np.uint64)
    M = _xorshift_jump_matrix(stride)
    v = int(seed)
    for j in range(K):
        starts[j] = v
        v = _apply_gf2(M, v)
    s = starts.copy()
    buf = np.empty((K, stride), dtype=np.float64)
    with np.errstate(over="ignore"):
        for t in range(stride):
            s ^= s << np.uint64(13)
            s ^= s >> np.uint64(7)
            s ^= s << np.uint64(17)
            buf[:, t] = (s >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    return buf.reshape(-1)[:count]


def _xorshift_step_int(v):
    v ^= (v << 13) & _MASK
    v ^= v >> 7
    v ^= (v << 17) & _MASK
    return v


def _xorshift_jump_matrix(nsteps):
    """64x64 GF(2) matrix (as 64 column ints) of `nsteps` xorshift64 steps, by repeated squaring."""
    cols = [_xorshift_step_int(1 << b) for b in range(64)]   # one step

    def mul(A, B):   # (A o B): apply B then A
        return [_apply_gf2(A, B[b]) for b in range(64)]
    result = [1 << b for b in range(64)]
    base = cols
    n = nsteps
    while n:
        if n & 1:
            result = mul(base, result)
        base = mul(base, base)
        n >>= 1
    return result


def _apply_gf2(M, v):
    r = 0
    b = 0
    while v:
        if v & 1:
            r ^= M[b]
        v >>= 1
        b += 1
    return r


def s_grid_planes(n, ix0, ix1, box=None, seed=1234567, amp=0.3):
    """The particles of s_grid(n) with grid index ix in [ix0, ix1) (a contiguous range of the particle numbering), without
    generating the others: lets every rank of a multi-GPU run build only its own slab of a large set."""
    if box is None:
        box = 64000.0 * n / 64
    sp = box / n
    i0, i1 = ix0 * n * n, ix1 * n * n
    u = xorshift64_uniform(3 * (i1 - i0), seed, skip=3 * i0).reshape(i1 - i0, 3)
    idx = np.arange(i0, i1, dtype=np.int64)
    ijk = np.stack([idx // (n * n), (idx // n) % n, idx % n], axis=1).astype(np.float64)
    pos = np.fmod((ijk + 0.5 + amp * (u - 0.5)) * sp + box, box)
    return pos, np.ones(i1 - i0, dtype=np.float32), box


def s_grid(n, box=None, seed=1234567, amp=0.3):
    """Perturbed grid: Pos_d = fmod((i_d + 0.5 + amp*(u-0.5))*sp + Box, Box), particle i=(ix*n+iy)*n+iz,
    draws in order x,y,z per particle.  Box defaults to 64000*n/64 (kpc/h, examples/dm-small scaling)."""
    if box is None:
        box = 64000.0 * n / 64
    sp = box / n
    N = n ** 3
    u = xorshift64_uniform(3 * N, seed).reshape(N, 3)
    idx = np.arange(N, dtype=np.int64)
    ijk = np.stack([idx // (n * n), (idx // n) % n, idx % n], axis=1).astype(np.float64)
    pos = np.fmod((ijk + 0.5 + amp * (u - 0.5)) * sp + box, box)
    mass = np.ones(N, dtype=np.float32)
    return pos, mass, box


def s_clust(n, box=8.0, seed=0):
    """Three populations: 1/4 uniform, 1/2 around box/2 width box/8*exp((u-.5)^2), 1/4 around 0.1 box width box/32."""
    N = n ** 3
    rng = np.random.RandomState(seed)    # MT19937; genrand_res53 doubles (NOT gsl_rng_uniform's 32-bit draws)
    pos = np.empty((N, 3))
    a, b = N // 4, 3 * N // 4
    pos[:a] = box * rng.random_sample((a, 3))
    pos[a:b] = box / 2 + box / 8 * np.exp((rng.random_sample((b - a, 3)) - 0.5) ** 2)
    pos[b:] = box * 0.1 + box / 32 * np.exp((rng.random_sample((N - b, 3)) - 0.5) ** 2)
    return pos, np.ones(N, dtype=np.float32), box


ZEL_TORCH_MIN = 320  # s_zel: grids from this size on are displaced on the GPU when one is there (tests and golden sets are smaller)


def _zel_positions_torch(n, wn, box, rms_disp, index):
    """s_zel's displacement field and positions with torch.fft on the current GPU; None without torch / a GPU"""
    try:
        import torch
        if not torch.cuda.is_available():
            return None
    except ImportError:
        return None
    dev = torch.device("cuda", torch.cuda.current_device())
    f8 = torch.float64
    sp = box / n
    dk = torch.fft.rfftn(torch.from_numpy(wn).to(dev))
    k1 = torch.fft.fftfreq(n, 1.0 / n, dtype=f8, device=dev)
    kz = torch.arange(n // 2 + 1, dtype=f8, device=dev)
    K = (k1[:, None, None], k1[None, :, None], kz[None, None, :])
    k2 = K[0] * K[0] + K[1] * K[1] + K[2] * K[2]
    k2[0, 0, 0] = 1.0
    dk = dk * k2 ** (index / 4.0)
    dk[0, 0, 0] = 0
    pos = torch.empty(n ** 3, 3, dtype=f8, device=dev)
    for a in range(3):
        pos[:, a] = torch.fft.irfftn(1j * K[a] / k2 * dk, s=(n, n, n), dim=(0, 1, 2)).reshape(-1)
    del dk, k2
    pos *= rms_disp / torch.sqrt(torch.mean(pos ** 2))
    idx = torch.arange(n ** 3, dtype=torch.int64, device=dev)
    for a, q in enumerate((idx // (n * n), (idx // n) % n, idx % n)):
        pos[:, a] = torch.remainder((q.to(f8) + 0.5 + pos[:, a]) * sp, box)
    del idx
    pos[pos <= 0] += box   # positions live in (0, Box] (drift.c:76-79)
    out = pos.cpu().numpy()
    del pos
    torch.cuda.empty_cache()
    return out


def s_zel(n, box=None, seed=181170, rms_disp=0.5, index=-2.0):
    """Zel'dovich displaced grid: x = q + psi(q), psi_k = i k/k^2 delta_k, delta_k Gaussian with P(k) ~ k^index,
    normalised so the per-axis rms displacement is `rms_disp` grid spacings."""
    if box is None:
        box = 64000.0 * n / 64
    sp = box / n
    rng = np.random.RandomState(seed)
    wn = rng.standard_normal((n, n, n))
    if n >= ZEL_TORCH_MIN:
        # the sets of the multi-GPU lines (320^3 ... 512^3, generated by EVERY rank) and of --size 512: the same construction with the
        # transforms on this process's GPU (numpy's single-threaded FFTs take minutes at 512^3; 8 ranks of a node do them side by side).
        # Same white noise, same formulas; the positions differ from the numpy form by the rounding of another FFT.
        pos = _zel_positions_torch(n, wn, box, rms_disp, index)
        if pos is not None:
            return pos, np.ones(n ** 3, dtype=np.float32), box
    dk = np.fft.rfftn(wn)
    k1 = np.fft.fftfreq(n, 1.0 / n)
    kz = np.arange(n // 2 + 1, dtype=np.float64)
    KX, KY, KZ = k1[:, None, None], k1[None, :, None], kz[None, None, :]
    k2 = KX * KX + KY * KY + KZ * KZ
    k2[0, 0, 0] = 1.0
    dk = dk * k2 ** (index / 4.0)
    dk[0, 0, 0] = 0
    psi = []
    for K in (KX, KY, KZ):
        psi.append(np.fft.irfftn(1j * K / k2 * dk, s=(n, n, n), axes=(0, 1, 2)))
    psi = np.stack(psi, axis=-1).reshape(-1, 3)
    psi *= rms_disp / np.sqrt(np.mean(psi ** 2))
    idx = np.arange(n ** 3, dtype=np.int64)
    q = np.stack([idx // (n * n), (idx // n) % n, idx % n], axis=1).astype(np.float64) + 0.5
    pos = np.mod((q + psi) * sp, box)
    pos[pos <= 0] += box   # positions live in (0, Box] (drift.c:76-79)
    return pos, np.ones(n ** 3, dtype=np.float32), box


def hydro_pair(n, box=None):
    """2 x n^3 particles for the gas configurations (BASELINE configs[2] and [4]): gas and dark matter offset from a
    Zel'dovich-displaced grid as MP-GenIC offsets them (genic/main.c:61-63).  Returns (pos, mass float32, type uint8, box): the gas
    (type 0) first, then the dark matter (type 1)."""
    posd, _, box = s_zel(n) if box is None else s_zel(n, box=box)
    sp = box / n
    ob, om = 0.045, 0.3
    posg = np.mod(posd - 0.5 * (om - ob) / om * sp, box)
    posd = np.mod(posd + 0.5 * ob / om * sp, box)
    posg[posg <= 0] += box
    posd[posd <= 0] += box
    pos = np.concatenate([posg, posd])
    mass = np.concatenate([np.full(n ** 3, ob / om, np.float32), np.full(n ** 3, 1 - ob / om, np.float32)])
    typ = np.concatenate([np.zeros(n ** 3, np.uint8), np.ones(n ** 3, np.uint8)])
    return pos, mass, typ, box
