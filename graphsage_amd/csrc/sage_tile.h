// Pieces the one-kernel forward passes share: the wave-local LDS fence, the feature table as the gather
// loops see it, the 16-row MFMA tile product on weights kept in LDS in fragment order, and the list of
// (Fu, Fi, H1/2, H2/2) shapes those kernels are instantiated for.
#pragma once
#include "gs_common.h"

namespace gs {
namespace tile {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr size_t kLdsMax = 160 * 1024;

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

// Saved activations are written once and read once by the backward pass: streaming stores keep
// them from displacing feature rows in the caches.
__device__ __forceinline__ void stream_store(float *dst, const float4 &v) {
    __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4 *>(dst));
}

// A feature table as the gather loops see it.  BUF: a 128-bit buffer descriptor (tables below
// 4 GiB): one 32-bit offset register per load and the hardware range check returns zeros for an
// out-of-range offset, so an invalid row costs one select on the offset and nothing after the
// load.  !BUF: plain 64-bit global loads with the zeroing done by a select on the data.
template <bool BUF>
struct RowTable {
    __amdgpu_buffer_rsrc_t rs;
    const char *base;
    long long rows;

    __device__ __forceinline__ void init(const float *p, long long n_rows, int row_bytes) {
        base = reinterpret_cast<const char *>(p);
        rows = n_rows;
        if (BUF) rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, (int)(unsigned)(n_rows * row_bytes), 0x00020000);
    }
    // 16 bytes at byte column `col` of row `row` (row = node id - type offset, computed in 32
    // bits: ids and row counts are below 2^31, so a negative or too large row fails the unsigned
    // compare); zeros when `want` is false or the row is outside the table
    __device__ __forceinline__ float4 load(bool want, unsigned row, unsigned col, int row_bytes) const {
        const bool ok = want && row < (unsigned)rows;
        if (BUF) {
            const unsigned off = ok ? row * (unsigned)row_bytes + col : 0xFFFFFFF0u;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
            return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
        const float4 t = *reinterpret_cast<const float4 *>(base + (size_t)(ok ? row : 0u) * row_bytes + col);
        return ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // What the deepest hop keeps in LDS per sampled node.  BUF: the row's byte offset in the
    // table, or kNoRow for an id outside it, so that the gather loop spends one OR and one select
    // per load (the range check was paid once, here); !BUF: the row number.
    static constexpr unsigned kNoRow = 0xFFFFF000u;       // + any column offset stays out of range, without wrapping
    __device__ __forceinline__ int32_t entry(int32_t id, int id_offset, int row_bytes) const {
        const unsigned row = (unsigned)(id - id_offset);
        if (BUF) return (int32_t)(row < (unsigned)rows ? row * (unsigned)row_bytes : kNoRow);
        return (int32_t)row;
    }
    __device__ __forceinline__ float4 load_entry(bool want, int32_t e, unsigned col, int row_bytes) const {
        if (BUF) {
            const unsigned off = want ? ((unsigned)e | col) : kNoRow;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
            return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
        return load(want, (unsigned)e, col, row_bytes);
    }
};

// D[16 x 16*NT] += A[16 x K] . W[K x 16*NT]; A in LDS (row stride lda), W in fragment order.
// Operands are fetched in batches of 64 k (four 16-byte A reads, 16*NT weight words) ahead of
// the MFMAs that use them, so the LDS latency is paid once per batch, not once per MFMA pair.
template <int K, int NT>
__device__ __forceinline__ void mma_tile(const float *A, int lda, const float *Wl, int wl, int r, int q,
                                         f32x4 (&acc)[NT]) {
    constexpr int KB = K < 64 ? K : 64;                  // k per batch
#pragma unroll
    for (int kb = 0; kb < K; kb += KB) {
        float4 av[KB / 16];
        float wv[KB / 4][NT];
#pragma unroll
        for (int k4 = 0; k4 < KB / 16; ++k4)
            av[k4] = *reinterpret_cast<const float4 *>(A + r * lda + q * (K / 4) + kb / 4 + 4 * k4);
#pragma unroll
        for (int ks = 0; ks < KB / 4; ++ks)
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) wv[ks][ct] = Wl[((kb / 4 + ks) * NT + ct) * 64 + wl];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k4 = 0; k4 < KB / 16; ++k4) {
            const float ae[4] = {av[k4].x, av[k4].y, av[k4].z, av[k4].w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ct = 0; ct < NT; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[e], wv[4 * k4 + e][ct], acc[ct], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// stages W[K x HH] (row-major, global) into LDS fragment order [(ks*NT + ct)*64 + 16 q + r]
__device__ __forceinline__ void stage_weight(float *dst, const float *W, int K, int HH) {
    const int NT = HH / 16, KS = K / 4;
    for (int idx = threadIdx.x; idx < KS * NT * 64; idx += blockDim.x) {
        const int rr = idx & 15, qq = (idx >> 4) & 3, ct = (idx >> 6) % NT, ks = (idx >> 6) / NT;
        dst[idx] = W[(size_t)(qq * KS + ks) * HH + ct * 16 + rr];
    }
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace tile
}  // namespace gs

// (Fu, Fi, H1/2, H2/2) combinations the kernel is instantiated for
#define GS_SAGE2_CASES(X) \
    X(64, 64, 16, 16) X(64, 64, 32, 32) X(64, 64, 32, 16) X(64, 64, 16, 32) \
    X(32, 32, 16, 16) X(32, 32, 32, 32) X(128, 128, 16, 16) X(128, 128, 32, 32) \
    X(64, 32, 16, 16) X(32, 64, 16, 16)
