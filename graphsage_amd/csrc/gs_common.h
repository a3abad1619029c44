// Shared host/device helpers of libgsage.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "gsage.h"

namespace gs {

// Thread-local error text returned by gs_last_error().
char *error_buffer();
int fail(int code, const char *fmt, ...);

// Checks the launch that was just made on this thread.
inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GS_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return GS_OK;
}

// A pair of events armed by gs_timer_arm(): the next timed launch of this thread is bracketed by
// them exactly (start and stop are stamped by the kernel's own dispatch packet, with no marker
// packets in between), then the pair is disarmed.
struct LaunchTimer {
    hipEvent_t start, stop;
};
LaunchTimer *take_armed_timer();            // NULL when none is armed

// Launch of a kernel that a benchmark may want to time: plain launch unless a timer is armed.
template <class K, class... Args>
inline void launch_timed(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t stream, Args... args) {
    if (LaunchTimer *t = take_armed_timer())
        hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, stream, t->start, t->stop, 0, args...);
    else
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...);
}

// Raises a kernel's dynamic-LDS limit.  The attribute belongs to the (kernel, device) pair, so what has been
// done is remembered per pair: a host thread that moves to another GPU raises the limit there as well.
inline hipError_t raise_lds_limit(const void *kernel, int bytes) {
    struct Done { const void *kernel; int device; };
    constexpr int kSlots = 64;
    static thread_local Done done[kSlots];
    static thread_local int n_done = 0;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    for (int i = 0; i < n_done; ++i)
        if (done[i].kernel == kernel && done[i].device == dev) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && n_done < kSlots) done[n_done++] = Done{kernel, dev};     // (a full table only costs the call)
    return e;
}

constexpr int kWave = 64;

__host__ __device__ inline int ilog2_ceil(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

// Philox4x32-10 (Salmon et al. 2011), first output word only.
__device__ __forceinline__ uint32_t philox4x32_10_x(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                    uint32_t k0, uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    constexpr uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += W0;
        k1 += W1;
    }
    return c0;
}

// Philox4x32-10, all four output words.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    constexpr uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += W0;
        k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// One element of Adam (the TF2 form: p -= lr_t * m / (sqrt(v) + eps)).  Written once and without contraction
// into fused multiply-adds, so that every kernel that applies it gives the same bits.
__device__ __forceinline__ void adam_element(float &p, float &m, float &v, float g, float lr_t, float b1, float b2, float eps) {
#pragma clang fp contract(off)
    const float mi = b1 * m + (1.f - b1) * g;
    const float vi = b2 * v + (1.f - b2) * g * g;
    m = mi;
    v = vi;
    p -= lr_t * mi / (sqrtf(vi) + eps);
}

// Dropout of the aggregator inputs (models.py:152,165), addressed like the sampling draws: the four columns
// 4*c4 .. 4*c4+3 of row `row` of input `site` of walk `walk` are kept where the four words of
// Philox(counter = (walk lo, walk hi, kDropSite | site, row * 64 + c4), key = dropout seed) are >= threshold
// (= rate * 2^32, rounded down); kept values are scaled by 1 / (1 - rate).  Sites of a two-hop walk: 0 the
// hop-2 rows (neighbors of the hop-1 heads, row = j * n1 + k), 1 the hop-1 rows as heads, 2 the hop-1 rows as the
// root's neighbors, 3 the root's row, 4 the layer-1 inputs (row 0 the root's activation, row 1 + j hop-1 head j's).
constexpr uint32_t kDropSite = 0x40000000u;
struct DropKey {
    uint32_t wlo, whi, k0, k1, thr;
    __device__ __forceinline__ float4 apply(float4 v, uint32_t site, uint32_t row, uint32_t c4, float scale) const {
        uint32_t w[4];
        philox4x32_10(wlo, whi, kDropSite | site, row * 64u + c4, k0, k1, w);
        return make_float4(w[0] >= thr ? v.x * scale : 0.f, w[1] >= thr ? v.y * scale : 0.f,
                           w[2] >= thr ? v.z * scale : 0.f, w[3] >= thr ? v.w * scale : 0.f);
    }
};

}  // namespace gs
