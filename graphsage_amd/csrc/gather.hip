// Kernel 2: feature-row gather and fused gather-mean / gather-max (forward and backward).
//
// A feature row of F floats is covered by a "row group" of G lanes, each lane
// holding VEC consecutive floats (VEC = 4 when F and the base pointers allow
// 16-byte accesses, else 2 or 1); G is the power of two >= F/VEC, capped at the
// wave width.  With F = 64 a wave therefore moves four 256-byte rows per load
// instruction, every lane 16 bytes, contiguous inside a row.
//
// gather-mean: one row group owns one parent and walks its n_child children,
// UNROLL row loads in flight per lane before the adds, so the sum is formed in
// child order inside registers and no cross-lane reduction is needed.  The
// child ids of the block's tile of parents are staged in LDS with one coalesced
// pass and read back as broadcasts.
//
// gather-max is the same walk with max in place of add (the reference's other
// aggregation, utils.py:110-111); its backward hands the parent's gradient to
// the children that attain the maximum, split evenly between ties -- the rule of
// TensorFlow's reduce_max gradient, which matters after a relu (many exact zeros).
#include "gs_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kUnroll = 8;
constexpr int kMaxTileIds = 8192;          // ints of LDS for a block's child-id tile

template <int VEC> struct Vec;
template <> struct Vec<4> { using type = float4; };
template <> struct Vec<2> { using type = float2; };
template <> struct Vec<1> { using type = float; };

template <int VEC> __device__ __forceinline__ typename Vec<VEC>::type vzero();
template <> __device__ __forceinline__ float4 vzero<4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <> __device__ __forceinline__ float2 vzero<2>() { return make_float2(0.f, 0.f); }
template <> __device__ __forceinline__ float vzero<1>() { return 0.f; }

__device__ __forceinline__ void vadd(float4 &a, const float4 &b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
__device__ __forceinline__ void vadd(float2 &a, const float2 &b) { a.x += b.x; a.y += b.y; }
__device__ __forceinline__ void vadd(float &a, const float &b) { a += b; }
__device__ __forceinline__ void vmax(float4 &a, const float4 &b) { a.x = fmaxf(a.x, b.x); a.y = fmaxf(a.y, b.y); a.z = fmaxf(a.z, b.z); a.w = fmaxf(a.w, b.w); }
__device__ __forceinline__ void vmax(float2 &a, const float2 &b) { a.x = fmaxf(a.x, b.x); a.y = fmaxf(a.y, b.y); }
__device__ __forceinline__ void vmax(float &a, const float &b) { a = fmaxf(a, b); }
template <int VEC> __device__ __forceinline__ typename Vec<VEC>::type vfill(float f);
template <> __device__ __forceinline__ float4 vfill<4>(float f) { return make_float4(f, f, f, f); }
template <> __device__ __forceinline__ float2 vfill<2>(float f) { return make_float2(f, f); }
template <> __device__ __forceinline__ float vfill<1>(float f) { return f; }
__device__ __forceinline__ float4 vdiv(const float4 &a, float d) { return make_float4(a.x / d, a.y / d, a.z / d, a.w / d); }
__device__ __forceinline__ float2 vdiv(const float2 &a, float d) { return make_float2(a.x / d, a.y / d); }
__device__ __forceinline__ float vdiv(const float &a, float d) { return a / d; }

template <int VEC>
__global__ __launch_bounds__(kThreads)
void gather_rows_kernel(const float *__restrict__ feat, int64_t n_rows, int F, int64_t id_offset,
                        const int32_t *__restrict__ ids, int64_t n, float *__restrict__ out, int log2g) {
    using V = typename Vec<VEC>::type;
    const int G = 1 << log2g;
    const int fv = F / VEC;
    const int groups = kThreads >> log2g;
    const int lig = threadIdx.x & (G - 1);
    const int64_t stride = (int64_t)gridDim.x * groups;
    for (int64_t i = (int64_t)blockIdx.x * groups + (threadIdx.x >> log2g); i < n; i += stride) {
        const int64_t row = (int64_t)ids[i] - id_offset;
        const bool ok = (uint64_t)row < (uint64_t)n_rows;
        const V *src = reinterpret_cast<const V *>(feat + (ok ? row : 0) * F);
        V *dst = reinterpret_cast<V *>(out + i * F);
        for (int c = lig; c < fv; c += G) dst[c] = ok ? src[c] : vzero<VEC>();
    }
}

template <int VEC, bool MAXOP>
__global__ __launch_bounds__(kThreads)
void gather_mean_kernel(const float *__restrict__ feat, int64_t n_rows, int F, int64_t id_offset,
                        const int32_t *__restrict__ child_ids, int64_t n_parents, int n_child,
                        float *__restrict__ out, int log2g, int tile_parents, int stage_ids) {
    using V = typename Vec<VEC>::type;
    extern __shared__ int32_t s_ids[];
    const int G = 1 << log2g;
    const int fv = F / VEC;
    const int groups = kThreads >> log2g;
    const int g = threadIdx.x >> log2g;
    const int lig = threadIdx.x & (G - 1);
    const float denom = (float)n_child;

    for (int64_t p0 = (int64_t)blockIdx.x * tile_parents; p0 < n_parents; p0 += (int64_t)gridDim.x * tile_parents) {
        const int np = (int)min((int64_t)tile_parents, n_parents - p0);
        if (stage_ids) {
            __syncthreads();                       // previous tile fully consumed
            const int32_t *src = child_ids + p0 * n_child;
            for (int i = threadIdx.x; i < np * n_child; i += kThreads) s_ids[i] = src[i];
            __syncthreads();
        }
        for (int pl = g; pl < np; pl += groups) {
            const int64_t p = p0 + pl;
            const int32_t *my_ids = stage_ids ? s_ids + pl * n_child
                                              : (child_ids ? child_ids + p * n_child : nullptr);
            for (int c = lig; c < fv; c += G) {
                // identity of the reduction for the slots past n_child; a row outside the table counts as zeros
                const V ident = MAXOP ? vfill<VEC>(-INFINITY) : vzero<VEC>();
                V acc = ident;
                for (int j0 = 0; j0 < n_child; j0 += kUnroll) {
                    V v[kUnroll];
#pragma unroll
                    for (int u = 0; u < kUnroll; ++u) {
                        const int j = j0 + u;
                        v[u] = ident;
                        if (j < n_child) {
                            const int64_t row = my_ids ? (int64_t)my_ids[j] - id_offset : p * n_child + j;
                            v[u] = vzero<VEC>();
                            if ((uint64_t)row < (uint64_t)n_rows)
                                v[u] = reinterpret_cast<const V *>(feat + row * F)[c];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kUnroll; ++u) {
                        if (MAXOP) vmax(acc, v[u]);
                        else vadd(acc, v[u]);
                    }
                }
                reinterpret_cast<V *>(out + p * F)[c] = MAXOP ? acc : vdiv(acc, denom);
            }
        }
    }
}

// The same mean over dense groups (the children of parent p are rows p * NC .. p * NC + NC - 1, NC known at compile
// time, 16-byte columns): a thread owns one 16-byte column of one parent, all NC loads are issued before the first add
// and nothing is conditional, so the launch streams the table (the layers above the first of a wide model read their
// neighbors this way).  The sum is formed in child order from zero and divided by NC as above: the same bits.
template <int NC>
__global__ __launch_bounds__(kThreads)
void dense_mean_kernel(const float4 *__restrict__ feat, int fv, int64_t total, float4 *__restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (idx >= total) return;
    const int64_t p = idx / fv;
    const int c = (int)(idx - p * fv);
    const float4 *src = feat + p * NC * fv + c;
    float4 v[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) v[j] = src[(int64_t)j * fv];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < NC; ++j) vadd(acc, v[j]);
    out[idx] = vdiv(acc, (float)NC);
}

// Backward: one row group per parent broadcasts g/n to its children.
template <int VEC>
__global__ __launch_bounds__(kThreads)
void gather_mean_bwd_kernel(const float *__restrict__ g_mean, const int32_t *__restrict__ child_ids,
                            int64_t n_parents, int n_child, int F, int64_t id_offset, int64_t n_rows,
                            float *__restrict__ g_feat, int log2g) {
    using V = typename Vec<VEC>::type;
    const int G = 1 << log2g;
    const int fv = F / VEC;
    const int groups = kThreads >> log2g;
    const int lig = threadIdx.x & (G - 1);
    const float denom = (float)n_child;
    const int64_t stride = (int64_t)gridDim.x * groups;
    for (int64_t p = (int64_t)blockIdx.x * groups + (threadIdx.x >> log2g); p < n_parents; p += stride) {
        for (int c = lig; c < fv; c += G) {
            const V g = vdiv(reinterpret_cast<const V *>(g_mean + p * F)[c], denom);
            for (int j = 0; j < n_child; ++j) {
                if (child_ids) {
                    const int64_t row = (int64_t)child_ids[p * n_child + j] - id_offset;
                    if ((uint64_t)row >= (uint64_t)n_rows) continue;
                    float *dst = g_feat + row * F + c * VEC;
                    const float *gs_ = reinterpret_cast<const float *>(&g);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) atomicAdd(dst + e, gs_[e]);
                } else {
                    reinterpret_cast<V *>(g_feat + (p * n_child + j) * F)[c] = g;
                }
            }
        }
    }
}

// Backward of the dense group max: child j of parent p gets g[p] where x[j] equals the
// maximum, divided by the number of children that do.
__global__ __launch_bounds__(kThreads)
void group_max_bwd_kernel(const float *__restrict__ g_max, const float *__restrict__ x,
                          const float *__restrict__ out_max, int64_t n_parents, int n_child, int F,
                          float *__restrict__ g_x) {
    const int64_t total = n_parents * F;
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += stride) {
        const int64_t p = i / F;
        const int f = (int)(i - p * F);
        const float m = out_max[i];
        const float *xs = x + p * n_child * F + f;
        int ties = 0;
        for (int j = 0; j < n_child; ++j) ties += xs[(int64_t)j * F] == m;
        const float share = ties ? g_max[i] / (float)ties : 0.f;
        float *gx = g_x + p * n_child * F + f;
        for (int j = 0; j < n_child; ++j) gx[(int64_t)j * F] = xs[(int64_t)j * F] == m ? share : 0.f;
    }
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool aligned8(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }

inline int pick_vec(int F, const void *a, const void *b) {
    if (F % 4 == 0 && aligned16(a) && aligned16(b)) return 4;
    if (F % 2 == 0 && aligned8(a) && aligned8(b)) return 2;
    return 1;
}

inline int group_log2(int F, int vec) {
    int l = gs::ilog2_ceil(F / vec);
    return l > 6 ? 6 : l;
}

inline unsigned grid_for(int64_t items, int per_block, int64_t cap = 1 << 20) {
    int64_t b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

}  // namespace

extern "C" int gs_gather_rows(const float *feat, int64_t n_rows, int F, int64_t id_offset,
                              const int32_t *ids, int64_t n, float *out, void *stream) {
    if (n == 0) return GS_OK;
    if (!feat || !ids || !out) return gs::fail(GS_EINVAL, "gs_gather_rows: null pointer argument");
    if (F < 1 || n < 0 || n_rows < 0) return gs::fail(GS_EINVAL, "gs_gather_rows: F=%d n=%lld n_rows=%lld", F, (long long)n, (long long)n_rows);
    const int vec = pick_vec(F, feat, out);
    const int lg = group_log2(F, vec);
    const int groups = kThreads >> lg;
    const unsigned grid = grid_for(n, groups);
    hipStream_t s = (hipStream_t)stream;
    if (vec == 4) hipLaunchKernelGGL(gather_rows_kernel<4>, dim3(grid), dim3(kThreads), 0, s, feat, n_rows, F, id_offset, ids, n, out, lg);
    else if (vec == 2) hipLaunchKernelGGL(gather_rows_kernel<2>, dim3(grid), dim3(kThreads), 0, s, feat, n_rows, F, id_offset, ids, n, out, lg);
    else hipLaunchKernelGGL(gather_rows_kernel<1>, dim3(grid), dim3(kThreads), 0, s, feat, n_rows, F, id_offset, ids, n, out, lg);
    return gs::check_launch("gs_gather_rows");
}

namespace {

template <bool MAXOP>
int gather_reduce_fwd(const char *what, const float *feat, int64_t n_rows, int F, int64_t id_offset,
                      const int32_t *child_ids, int64_t n_parents, int n_child, float *out_mean, void *stream) {
    if (n_parents == 0) return GS_OK;
    if (!feat || !out_mean) return gs::fail(GS_EINVAL, "%s: null pointer argument", what);
    if (F < 1 || n_child < 1 || n_parents < 0 || n_rows < 0)
        return gs::fail(GS_EINVAL, "%s: F=%d n_child=%d n_parents=%lld n_rows=%lld", what, F, n_child, (long long)n_parents, (long long)n_rows);
    if (!child_ids && n_parents * (int64_t)n_child > n_rows)
        return gs::fail(GS_EINVAL, "%s: dense groups need n_parents*n_child <= n_rows", what);
    const int vec = pick_vec(F, feat, out_mean);
    hipStream_t s = (hipStream_t)stream;
    if (!MAXOP && !child_ids && vec == 4 && (n_child == 5 || n_child == 10 || n_child == 25) &&
        n_parents * (int64_t)(F / 4) <= ((int64_t)1 << 31) * kThreads) {
        const int fv = F / 4;
        const int64_t total = n_parents * fv;
        const unsigned grid = (unsigned)((total + kThreads - 1) / kThreads);
        const float4 *src = reinterpret_cast<const float4 *>(feat);
        float4 *dst = reinterpret_cast<float4 *>(out_mean);
        if (n_child == 5) gs::launch_timed(dense_mean_kernel<5>, dim3(grid), dim3(kThreads), 0, s, src, fv, total, dst);
        else if (n_child == 10) gs::launch_timed(dense_mean_kernel<10>, dim3(grid), dim3(kThreads), 0, s, src, fv, total, dst);
        else gs::launch_timed(dense_mean_kernel<25>, dim3(grid), dim3(kThreads), 0, s, src, fv, total, dst);
        return gs::check_launch(what);
    }
    const int lg = group_log2(F, vec);
    const int groups = kThreads >> lg;
    int tile = groups * 4;
    int stage = child_ids != nullptr;
    if (stage && (int64_t)tile * n_child > kMaxTileIds) tile = groups;
    if (stage && (int64_t)tile * n_child > kMaxTileIds) stage = 0;
    const size_t lds = stage ? (size_t)tile * n_child * sizeof(int32_t) : 0;
    const unsigned grid = grid_for(n_parents, tile);
    if (vec == 4) gs::launch_timed(gather_mean_kernel<4, MAXOP>, dim3(grid), dim3(kThreads), lds, s, feat, n_rows, F, id_offset, child_ids, n_parents, n_child, out_mean, lg, tile, stage);
    else if (vec == 2) gs::launch_timed(gather_mean_kernel<2, MAXOP>, dim3(grid), dim3(kThreads), lds, s, feat, n_rows, F, id_offset, child_ids, n_parents, n_child, out_mean, lg, tile, stage);
    else gs::launch_timed(gather_mean_kernel<1, MAXOP>, dim3(grid), dim3(kThreads), lds, s, feat, n_rows, F, id_offset, child_ids, n_parents, n_child, out_mean, lg, tile, stage);
    return gs::check_launch(what);
}

}  // namespace

extern "C" int gs_gather_mean_fwd(const float *feat, int64_t n_rows, int F, int64_t id_offset,
                                  const int32_t *child_ids, int64_t n_parents, int n_child,
                                  float *out_mean, void *stream) {
    return gather_reduce_fwd<false>("gs_gather_mean_fwd", feat, n_rows, F, id_offset, child_ids, n_parents, n_child,
                                    out_mean, stream);
}

extern "C" int gs_gather_max_fwd(const float *feat, int64_t n_rows, int F, int64_t id_offset,
                                 const int32_t *child_ids, int64_t n_parents, int n_child,
                                 float *out_max, void *stream) {
    return gather_reduce_fwd<true>("gs_gather_max_fwd", feat, n_rows, F, id_offset, child_ids, n_parents, n_child,
                                   out_max, stream);
}

extern "C" int gs_group_max_bwd(const float *g_max, const float *x, const float *out_max, int64_t n_parents,
                                int n_child, int F, float *g_x, void *stream) {
    if (n_parents == 0) return GS_OK;
    if (!g_max || !x || !out_max || !g_x) return gs::fail(GS_EINVAL, "gs_group_max_bwd: null pointer argument");
    if (F < 1 || n_child < 1 || n_parents < 0)
        return gs::fail(GS_EINVAL, "gs_group_max_bwd: F=%d n_child=%d n_parents=%lld", F, n_child, (long long)n_parents);
    const unsigned grid = grid_for(n_parents * (int64_t)F, kThreads, 256 * 64);
    hipLaunchKernelGGL(group_max_bwd_kernel, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, g_max, x, out_max,
                       n_parents, n_child, F, g_x);
    return gs::check_launch("gs_group_max_bwd");
}

extern "C" int gs_gather_mean_bwd(const float *g_mean, const int32_t *child_ids, int64_t n_parents,
                                  int n_child, int F, int64_t id_offset, int64_t n_rows,
                                  float *g_feat, void *stream) {
    if (n_parents == 0) return GS_OK;
    if (!g_mean || !g_feat) return gs::fail(GS_EINVAL, "gs_gather_mean_bwd: null pointer argument");
    if (F < 1 || n_child < 1 || n_parents < 0 || n_rows < 0)
        return gs::fail(GS_EINVAL, "gs_gather_mean_bwd: F=%d n_child=%d n_parents=%lld n_rows=%lld", F, n_child, (long long)n_parents, (long long)n_rows);
    if (!child_ids && n_parents * (int64_t)n_child > n_rows)
        return gs::fail(GS_EINVAL, "gs_gather_mean_bwd: dense groups need n_parents*n_child <= n_rows");
    const int vec = pick_vec(F, g_mean, g_feat);
    const int lg = group_log2(F, vec);
    const int groups = kThreads >> lg;
    const unsigned grid = grid_for(n_parents, groups);
    hipStream_t s = (hipStream_t)stream;
    if (vec == 4) hipLaunchKernelGGL(gather_mean_bwd_kernel<4>, dim3(grid), dim3(kThreads), 0, s, g_mean, child_ids, n_parents, n_child, F, id_offset, n_rows, g_feat, lg);
    else if (vec == 2) hipLaunchKernelGGL(gather_mean_bwd_kernel<2>, dim3(grid), dim3(kThreads), 0, s, g_mean, child_ids, n_parents, n_child, F, id_offset, n_rows, g_feat, lg);
    else hipLaunchKernelGGL(gather_mean_bwd_kernel<1>, dim3(grid), dim3(kThreads), 0, s, g_mean, child_ids, n_parents, n_child, F, id_offset, n_rows, g_feat, lg);
    return gs::check_launch("gs_gather_mean_bwd");
}
