"""Philox4x32-10 on the CPU (numpy, vectorised) -- test infrastructure.

The reference draws from the stdlib Mersenne Twister (``sampler.py:24,40,78``);
a counter-based generator has no counterpart there.  The device sampler keys one
Philox block per draw slot, and this file is the word-for-word CPU statement of
that stream:

    counter = (walk_lo, walk_hi, hop, slot)      key = (seed_lo, seed_hi)
    r       = philox4x32_10(counter, key)[0]
    index   = (r * degree) >> 32

``walk`` is the global walk id (2 * seed_edge_id + side), so the draws do not
depend on batch size or on how seed edges are split over GPUs.
"""
import numpy as np

PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = 0x9E3779B9
PHILOX_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Ten rounds of Philox-4x32 over arrays of counters.

    All arguments broadcast against each other; values are taken modulo 2**32.
    Returns four uint32 arrays.
    """
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & _MASK for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = PHILOX_M0 * c0          # < 2**64: both factors < 2**32
        p1 = PHILOX_M1 * c2
        hi0, lo0 = p0 >> _S32, p0 & _MASK
        hi1, lo1 = p1 >> _S32, p1 & _MASK
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n1 = lo1
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        n3 = lo0
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + PHILOX_W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def draw_u32(walk, hop, slot, rng_seed):
    """The 32-bit word the device sampler uses for (walk, hop, slot)."""
    walk = np.asarray(walk, dtype=np.uint64)
    rng_seed = int(rng_seed) & 0xFFFFFFFFFFFFFFFF
    return philox4x32_10(walk & _MASK, walk >> _S32, hop, slot,
                         rng_seed & 0xFFFFFFFF, rng_seed >> 32)[0]


def index_from_u32(r, degree):
    """``(r * degree) >> 32`` in exact integer arithmetic (degree < 2**31)."""
    return ((np.asarray(r, dtype=np.uint64) * np.asarray(degree, dtype=np.uint64)) >> _S32).astype(np.int64)
