// Pieces shared by the one-kernel backward passes (sage2_bwd_all.hip: two hops; sage3_bwd.hip: three hops): a
// lane's 16 bytes of a row, sums over a wave, the fence between a wave's LDS writes and its reads, and the
// reduction that adds the blocks' slab rows and can apply the optimizer step.
#pragma once

#include "gs_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int V> struct Row;
template <> struct Row<2> {
    float v[2];
    __device__ __forceinline__ void load(const float *p) { const float2 t = *reinterpret_cast<const float2 *>(p); v[0] = t.x; v[1] = t.y; }
};
template <> struct Row<4> {
    float v[4];
    __device__ __forceinline__ void load(const float *p) { const float4 t = *reinterpret_cast<const float4 *>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
};

// Sum over the 64 lanes, the same value in every lane: within rows of 16 lanes by data-parallel moves (no
// LDS crossbar round trips), then the four rows' sums through scalar registers.
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0xB1>(v);                    // quad_perm [1, 0, 3, 2]
    v = dpp_add<0x4E>(v);                    // quad_perm [2, 3, 0, 1]
    v = dpp_add<0x141>(v);                   // row_half_mirror
    v = dpp_add<0x140>(v);                   // row_mirror
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}

// Orders a wave's LDS writes before its own later LDS reads (other lanes' elements); the reads of global
// memory in flight are not waited for.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Adds the slab rows in block order: element e is summed by sixteen threads that take rows b = j, j + 16, ...
// each in increasing order; their partial sums are then added in index order.  A table of segments says
// where each stretch of the slab row goes.
struct Segment {
    int begin, end;
    float *dst;
    float scale;
    int update;                              // the optimizer step follows for these elements
};
// Adam on the elements just summed (the arithmetic of gs_adam_step, element by element)
struct AdamArgs {
    const float *g_base, *lr_dev;
    float *p, *m, *v;
    float lr, b1, b2, eps, g_scale;
};
constexpr int kSegments = 24;
struct ReduceArgs {
    Segment seg[kSegments];
    int n_seg, n_rows, row_floats;
    const float *slab;
    AdamArgs adam;
};

__global__ __launch_bounds__(512)
void sage_bwd_reduce_kernel(const ReduceArgs a) {
    __shared__ float part[512];
    const int e = blockIdx.x * 32 + (threadIdx.x & 31), j = threadIdx.x >> 5;
    float s = 0.f;
    if (e < a.row_floats) {
        const float *p = a.slab + e;
        int b = j;
        for (; b + 112 < a.n_rows; b += 128) {                    // eight reads in flight
            float x[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = p[(size_t)(b + 16 * i) * a.row_floats];
#pragma unroll
            for (int i = 0; i < 8; ++i) s += x[i];
        }
        for (; b < a.n_rows; b += 16) s += p[(size_t)b * a.row_floats];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if (j == 0 && e < a.row_floats) {
        s = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) s += part[g * 32 + threadIdx.x];
        for (int i = 0; i < a.n_seg; ++i)
            if (e >= a.seg[i].begin && e < a.seg[i].end) {
                if (a.seg[i].dst) {
                    float *dst = a.seg[i].dst + (e - a.seg[i].begin);
                    const float g = s * a.seg[i].scale;
                    *dst = g;
                    if (a.seg[i].update) {
                        const long long at = dst - a.adam.g_base;
                        const float lr_t = a.adam.lr_dev ? a.adam.lr_dev[0] : a.adam.lr;
                        gs::adam_element(a.adam.p[at], a.adam.m[at], a.adam.v[at], g * a.adam.g_scale, lr_t, a.adam.b1, a.adam.b2, a.adam.eps);
                    }
                }
                break;
            }
    }
}


// Launches the reduction over n_rows slab rows of row_floats floats (ra.seg / ra.n_seg / ra.adam filled by the caller).
inline int launch_bwd_reduce(ReduceArgs &ra, const float *slab, int n_rows, int row_floats, hipStream_t s, const char *who) {
    ra.n_rows = n_rows; ra.row_floats = row_floats; ra.slab = slab;
    hipLaunchKernelGGL(sage_bwd_reduce_kernel, dim3((unsigned)((row_floats + 31) / 32)), dim3(512), 0, s, ra);
    return gs::check_launch(who);
}

}  // namespace
