// The reference's other walker: neighbors taken cyclically, without replacement, from a per-node
// cursor, with a fresh permutation of the neighbor list whenever the cursor runs off its end
// (backup/sampler_backup.py:78-96; the cursors are Graph.sampling_index_dict, bipartite_graph.py:
// 28,44-46).
//
// A visit of node v that wants k samples, with x = v's neighbor list of length L and c = cursor[v]:
//     sample t <  L - c           : x[c + t]                                   (no wrap yet)
//     sample t >= L - c, q = t - (L - c), wrap number wn = q / L + 1, i = q % L :
//                                   x[ s_1( s_2( ... s_wn(i) ... )) ]
// because every wrap permutes the list the previous wrap left (`neighbors = permutation(neighbors)`
// rebinds the local list, sampler_backup.py:87; the stored list is never permuted) and restarts at
// position 0.  m = number of wraps = 0 if L - c >= k, else ceil((k - (L - c)) / L); the cursor
// becomes c + k without a wrap and k - (L - c) - (m - 1) L after one.
//
// The cursor of a node depends on every earlier visit of it -- roots in call order, each walk breadth
// first -- so visits are sequential by the reference's own definition; one wavefront walks them in
// order and only the k samples of a visit are spread over its lanes.  This is the variant for
// reproducing that sampler, not a fast path: the with-replacement walker of sampler.hip is.
//
// Permutation t of a run (t counts wraps in visit order):
//   replay : perm_tape holds the index permutations back to back, s_t(i) = tape[offset_t + i];
//   Philox : s_t is a four-round Feistel network over the next even power of two >= L, keyed by
//            Philox4x32-10(counter = (t_lo, t_hi, round, 0x43594331), key = rng_seed), cycle-walked
//            into [0, L).
#include "gs_common.h"

namespace {

struct CyclicParams {
    int n_hops;
    int fan[GS_MAX_HOPS];
    int size[GS_MAX_HOPS];
    int32_t *out[GS_MAX_HOPS];
};

__device__ __forceinline__ uint32_t feistel_perm(uint64_t t, uint32_t L, uint32_t i, uint32_t key0, uint32_t key1) {
    int bits = 2;
    while ((1ull << bits) < (uint64_t)L) bits += 2;             // even number of bits, domain >= L
    const int half = bits >> 1;
    const uint32_t mask = (1u << half) - 1u;
    uint32_t rk[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        rk[r] = gs::philox4x32_10_x((uint32_t)t, (uint32_t)(t >> 32), (uint32_t)r, 0x43594331u, key0, key1);
    uint32_t x = i;
    do {
        uint32_t l = x >> half, r = x & mask;
#pragma unroll
        for (int rd = 0; rd < 4; ++rd) {
            uint32_t f = (r ^ rk[rd]) * 0x9E3779B1u;
            f ^= f >> 15;
            f *= 0x85EBCA77u;
            f ^= f >> 13;
            const uint32_t nl = r, nr = l ^ (f & mask);
            l = nl; r = nr;
        }
        x = (l << half) | r;
    } while (x >= L);
    return x;
}

template <class T>
__device__ __forceinline__ T ld(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(64)
void cyclic_walk_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices, int64_t n_nodes,
                        const int32_t *__restrict__ seeds, int64_t n_seeds, CyclicParams P,
                        int64_t *cursor, uint32_t key0, uint32_t key1,
                        int64_t *state,                         // [0] permutations drawn so far, [1] tape words used so far
                        const int32_t *__restrict__ tape, int64_t tape_len, int32_t *err) {
    const int lane = threadIdx.x;
    int64_t n_perm = ld(state), tape_pos = ld(state + 1);
    for (int64_t w = 0; w < n_seeds; ++w) {
        int32_t root = seeds[w];
        if ((uint64_t)(int64_t)root >= (uint64_t)n_nodes) {
            if (lane == 0) atomicMax(&err[1], 1);
            root = -1;
        }
        for (int h = 0; h < P.n_hops; ++h) {
            const int k = P.fan[h];
            const int parents = h ? P.size[h - 1] : 1;
            int32_t *dst = P.out[h] + w * (int64_t)P.size[h];
            for (int j = 0; j < parents; ++j) {
                // this wave's own earlier stores, read back past the vector cache
                const int32_t v = h ? ld(P.out[h - 1] + w * (int64_t)P.size[h - 1] + j) : root;
                int64_t a = 0, L = 0, c = 0;
                if (v >= 0) {
                    a = indptr[v];
                    L = indptr[v + 1] - a;
                    c = ld(cursor + v);
                }
                if (v < 0 || L == 0) {                            // below a dead slot, or no neighbors (the reference raises)
                    if (v >= 0 && lane == 0) atomicMax(&err[0], v + 1);
                    for (int t = lane; t < k; t += 64) dst[(int64_t)j * k + t] = -1;
                    continue;
                }
                const int64_t first = L - c;                      // samples before the first wrap
                int64_t m = 0, after = 0;
                if (first < k) { after = k - first; m = (after + L - 1) / L; }
                if (tape && m > 0 && tape_pos + m * L > tape_len) {
                    if (lane == 0) atomicMax(&err[1], 2);         // tape too short: stop drawing from it
                    for (int t = lane; t < k; t += 64) dst[(int64_t)j * k + t] = -1;
                    continue;
                }
                for (int t = lane; t < k; t += 64) {
                    int64_t pos;
                    if (t < first) pos = c + t;
                    else {
                        const int64_t q = t - first;
                        const int64_t wn = q / L;                 // 0-based wrap this sample falls in
                        uint32_t i = (uint32_t)(q - wn * L);
                        for (int64_t s = wn; s >= 0; --s)         // innermost permutation first
                            i = tape ? (uint32_t)tape[tape_pos + s * L + i]
                                     : feistel_perm((uint64_t)(n_perm + s), (uint32_t)L, i, key0, key1);
                        pos = i < (uint64_t)L ? (int64_t)i : 0;   // (a tape entry outside [0, L) is clamped to the row)
                    }
                    dst[(int64_t)j * k + t] = indices[a + pos];
                }
                if (lane == 0)
                    __hip_atomic_store(cursor + v, m ? after - (m - 1) * L : c + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                n_perm += m;
                tape_pos += m * L;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");    // the hop's ids are the next hop's parents
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    if (lane == 0) { state[0] = n_perm; state[1] = tape_pos; }
}

}  // namespace

extern "C" int gs_sample_cyclic(const int64_t *indptr, const int32_t *indices, int64_t n_nodes,
                                const int32_t *seeds, int64_t n_seeds,
                                const int32_t *fanouts, int n_hops,
                                int64_t *cursor, uint64_t rng_seed, int64_t *state,
                                const int32_t *perm_tape, int64_t perm_tape_len,
                                int32_t *const *out_hops, int32_t *err_flag, void *stream) {
    if (n_seeds == 0) return GS_OK;
    if (!indptr || !indices || !seeds || !fanouts || !out_hops || !err_flag || !cursor || !state)
        return gs::fail(GS_EINVAL, "gs_sample_cyclic: null pointer argument");
    if (n_hops < 1 || n_hops > GS_MAX_HOPS)
        return gs::fail(GS_EINVAL, "gs_sample_cyclic: n_hops=%d outside [1,%d]", n_hops, GS_MAX_HOPS);
    if (n_seeds < 0 || n_nodes <= 0 || n_nodes > 0x7FFFFFFELL)
        return gs::fail(GS_EINVAL, "gs_sample_cyclic: n_seeds=%lld n_nodes=%lld out of range", (long long)n_seeds, (long long)n_nodes);
    if (perm_tape_len < 0 || (perm_tape_len > 0 && !perm_tape))
        return gs::fail(GS_EINVAL, "gs_sample_cyclic: perm_tape_len=%lld without a tape", (long long)perm_tape_len);
    CyclicParams P{};
    P.n_hops = n_hops;
    int64_t prod = 1;
    for (int h = 0; h < n_hops; ++h) {
        if (fanouts[h] < 1) return gs::fail(GS_EINVAL, "gs_sample_cyclic: fanouts[%d]=%d < 1", h, fanouts[h]);
        if (!out_hops[h]) return gs::fail(GS_EINVAL, "gs_sample_cyclic: out_hops[%d] is null", h);
        prod *= fanouts[h];
        if (prod > (1 << 24)) return gs::fail(GS_ELIMIT, "gs_sample_cyclic: hop %d has more than 2^24 slots per walk", h);
        P.fan[h] = fanouts[h];
        P.size[h] = (int)prod;
        P.out[h] = out_hops[h];
    }
    hipLaunchKernelGGL(cyclic_walk_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, indptr, indices, n_nodes, seeds, n_seeds,
                       P, cursor, (uint32_t)rng_seed, (uint32_t)(rng_seed >> 32), state, perm_tape, perm_tape_len, err_flag);
    return gs::check_launch("gs_sample_cyclic");
}
