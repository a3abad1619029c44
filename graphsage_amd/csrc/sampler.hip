// Kernel 1: CSR multi-hop uniform neighbor sampler, one wavefront per walk.
//
// A walk's hop h has size[h] = fanouts[0]*...*fanouts[h] slots; slot t of hop h
// samples a neighbor of slot t / fanouts[h] of hop h-1 (the root for h = 0),
// which is the order a FIFO breadth-first walk visits them.  The 64 lanes of the
// wave take slots t = lane, lane+64, ...; the ids of a finished hop stay in the
// wave's own LDS region so the next hop can look its parents up without going
// back to memory.  Everything a slot needs (CSR row bounds, one random word, one
// neighbor id) is independent of the other slots, so the kernel is a chain of
// n_hops dependent random reads per lane and lives on memory-level parallelism:
// 4 waves per block, up to 8 blocks per CU.
#include "gs_common.h"

namespace {

struct SampleParams {
    int n_hops;
    int fan[GS_MAX_HOPS];
    int size[GS_MAX_HOPS];      // slots per hop
    int off[GS_MAX_HOPS];       // offset of the hop inside one walk's tape
    int tape_len;               // words of tape per walk
    int lds_per_buf;            // ints per LDS parent buffer (two per wave); 0: parents are re-read
                                // from the previous hop's output array instead (very wide hops)
    int32_t *out[GS_MAX_HOPS];
};

constexpr int kWavesPerBlock = 4;
constexpr int kChunks = 4;

__global__ __launch_bounds__(kWavesPerBlock * 64)
void sample_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                   int64_t n_nodes, const int32_t *__restrict__ seeds, int64_t n_seeds,
                   SampleParams P, uint32_t key0, uint32_t key1, uint64_t walk_base,
                   const int64_t *__restrict__ walk_ids, const uint32_t *__restrict__ replay,
                   int32_t *__restrict__ err) {
    extern __shared__ int32_t lds[];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * kWavesPerBlock + wave;
    if (w >= n_seeds) return;                       // wave-uniform

    int32_t *buf0 = lds + (size_t)wave * 2 * P.lds_per_buf;
    int32_t *buf1 = buf0 + P.lds_per_buf;

    int32_t root = seeds[w];
    if ((uint64_t)(int64_t)root >= (uint64_t)n_nodes) {
        if (lane == 0) atomicMax(&err[1], 1);
        root = -1;
    }
    const uint64_t walk = walk_ids ? (uint64_t)walk_ids[w] : walk_base + (uint64_t)w;
    const uint32_t wlo = (uint32_t)walk, whi = (uint32_t)(walk >> 32);
    const uint32_t *tape = replay ? replay + (size_t)w * P.tape_len : nullptr;

    for (int h = 0; h < P.n_hops; ++h) {
        const int k = P.fan[h];
        const int total = P.size[h];
        const int32_t *par = (h & 1) ? buf0 : buf1;         // ids of hop h-1
        int32_t *cur = (h & 1) ? buf1 : buf0;               // ids of hop h
        const bool use_lds = P.lds_per_buf > 0;
        const bool keep = use_lds && h + 1 < P.n_hops;
        int32_t *dst = P.out[h] + (size_t)w * total;
        const int32_t *prev = h ? P.out[h - 1] + (size_t)w * P.size[h - 1] : nullptr;
        // kChunks * 64 slots per pass: all row-bound loads of a pass are issued before the first
        // neighbor load, so a pass costs two dependent memory round trips however many slots it has
        for (int base = 0; base < total; base += kChunks * 64) {
            int32_t p[kChunks];
            int64_t a[kChunks], b[kChunks];
#pragma unroll
            for (int c = 0; c < kChunks; ++c) {
                const int t = base + c * 64 + lane;
                p[c] = -1;
                if (t < total)
                    p[c] = (h == 0) ? root
                           : use_lds ? par[t / k]
                                     // this wave's own stores of the previous hop, read back past the L1
                                     : __hip_atomic_load(prev + t / k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int32_t pc = p[c] >= 0 ? p[c] : 0;    // node 0 stands in for dead slots
                a[c] = indptr[pc];
                b[c] = indptr[pc + 1];
            }
            int32_t v[kChunks];
#pragma unroll
            for (int c = 0; c < kChunks; ++c) {
                const int t = base + c * 64 + lane;
                const uint32_t deg = (uint32_t)(b[c] - a[c]);
                uint32_t r = 0;
                if (tape) { if (t < total) r = tape[P.off[h] + t]; }
                else r = gs::philox4x32_10_x(wlo, whi, (uint32_t)h, (uint32_t)t, key0, key1);
                const bool live = p[c] >= 0 && deg != 0;
                if (p[c] >= 0 && deg == 0) atomicMax(&err[0], p[c] + 1);
                const int64_t pos = live ? a[c] + (int64_t)__umulhi(r, deg) : 0;
                // plain load: a draw costs one 64-byte sector whatever the cache policy (nontemporal
                // loads were measured 14 % slower, sc1 / sc0 sc1 the same)
                v[c] = indices[pos];
                if (!live) v[c] = -1;
            }
#pragma unroll
            for (int c = 0; c < kChunks; ++c) {
                const int t = base + c * 64 + lane;
                if (t < total) {
                    dst[t] = v[c];
                    if (keep) cur[t] = v[c];
                }
            }
        }
        // make this hop's writes visible to the other lanes of the wave (LDS, or memory when the
        // parents are re-read from the output array)
        if (use_lds) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
}

}  // namespace

extern "C" int gs_sample(const int64_t *indptr, const int32_t *indices, int64_t n_nodes,
                         const int32_t *seeds, int64_t n_seeds,
                         const int32_t *fanouts, int n_hops,
                         uint64_t rng_seed, uint64_t walk_id_base, const int64_t *walk_ids,
                         const uint32_t *replay,
                         int32_t *const *out_hops, int32_t *err_flag, void *stream) {
    if (n_seeds == 0) return GS_OK;                 // empty batch: pointers may be null
    if (!indptr || !indices || !seeds || !fanouts || !out_hops || !err_flag)
        return gs::fail(GS_EINVAL, "gs_sample: null pointer argument");
    if (n_hops < 1 || n_hops > GS_MAX_HOPS)
        return gs::fail(GS_EINVAL, "gs_sample: n_hops=%d outside [1,%d]", n_hops, GS_MAX_HOPS);
    if (n_seeds < 0 || n_nodes <= 0 || n_nodes > 0x7FFFFFFELL)
        return gs::fail(GS_EINVAL, "gs_sample: n_seeds=%lld n_nodes=%lld out of range",
                        (long long)n_seeds, (long long)n_nodes);
    SampleParams P{};
    P.n_hops = n_hops;
    int64_t prod = 1, off = 0;
    for (int h = 0; h < n_hops; ++h) {
        if (fanouts[h] < 1) return gs::fail(GS_EINVAL, "gs_sample: fanouts[%d]=%d < 1", h, fanouts[h]);
        if (!out_hops[h]) return gs::fail(GS_EINVAL, "gs_sample: out_hops[%d] is null", h);
        prod *= fanouts[h];
        if (prod > (1 << 24)) return gs::fail(GS_ELIMIT, "gs_sample: hop %d has more than 2^24 slots per walk", h);
        P.fan[h] = fanouts[h];
        P.size[h] = (int)prod;
        P.off[h] = (int)off;
        P.out[h] = out_hops[h];
        off += prod;
    }
    P.tape_len = (int)off;
    P.lds_per_buf = n_hops >= 2 ? P.size[n_hops - 2] : 1;
    size_t lds_bytes = (size_t)kWavesPerBlock * 2 * P.lds_per_buf * sizeof(int32_t);
    if (lds_bytes > 64 * 1024) {                    // hops too wide for LDS: parents come from the output arrays
        P.lds_per_buf = 0;
        lds_bytes = 0;
    }
    const int64_t blocks = (n_seeds + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks > 0x7FFFFFFFLL) return gs::fail(GS_ELIMIT, "gs_sample: too many seeds for one launch");

    hipLaunchKernelGGL(sample_kernel, dim3((unsigned)blocks), dim3(kWavesPerBlock * 64), lds_bytes,
                       (hipStream_t)stream, indptr, indices, n_nodes, seeds, n_seeds, P,
                       (uint32_t)rng_seed, (uint32_t)(rng_seed >> 32), walk_id_base, walk_ids, replay,
                       err_flag);
    return gs::check_launch("gs_sample");
}
