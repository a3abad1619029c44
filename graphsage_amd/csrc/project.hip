// Kernel 3: per-node-type self/neighbor projection on the fp32 matrix cores
// (forward and backward), plus the l2 normalisation of the final embeddings.
//
//   out[m, 0:H/2]  = x_self[m, :] @ W_self        out[m, H/2:H] = x_mean[m, :] @ W_neigh
//   out = act(out + bias)
//
// rows_gemm_kernel is a row-streaming GEMM  out[M, N] = A[M, K] @ B[K, N]  for
// two independent "sides" in one launch (blockIdx.y picks the side and a
// 64-column block inside it, blockIdx.x a 64-row block).  Forward: the sides are
// the self and the neighbor half.  Backward w.r.t. the inputs: A is the masked
// output gradient of one half and B the transposed weight, selected by strides.
// A block is 4 waves, wave w owning rows 16w..16w+15 of the tile and up to four
// 16x16 accumulators (v_mfma_f32_16x16x4_f32: A[row = lane&15][k = lane>>4],
// B[k = lane>>4][col = lane&15], D[row = 4*(lane>>4) + i][col = lane&15]).
// K is walked in steps of 32 through LDS; out-of-range rows/columns/k are
// zero-filled while staging, so every shape is handled by the same code.
//
// wgrad_kernel forms the weight and bias gradients  dW[K, N] = A[M, K]^T @ G[M, N]
// with the sum over M split into chunks: each block writes the partial sum of
// its chunk to a slab, wgrad_reduce_kernel adds the slabs in chunk order, so
// the result is bitwise reproducible (no float atomics).  The bias gradient is
// the extra row K of the same product (A extended by a column of ones).
//
// LDS images are padded so fragment reads are conflict-free for ds_read_b32
// (bank = dword address mod 32 inside each 32-lane half): tiles read with the
// row on the lane use a row stride of 34 dwords (rows 0..15 land on even banks,
// k and k+1 fill the odd ones), tiles read with the column on the lane a row
// stride of 80 (k and k+1 are 16 banks apart).
#include "gs_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int BM = 64, BN = 64, BK = 32;
constexpr int A_LD = 34, W_LD = 80;
constexpr int kThreads = 256;

struct GemmSide {
    const float *A;      // [M, K], row stride lda
    const float *mask;   // same indexing as A; element used only where mask > 0 (NULL: no mask)
    const float *B;      // B(k, n) = B[k*ldbk + n*ldbn]
    const float *bias;   // [N] or NULL
    float *out;          // [M, N], row stride ldo
    int lda, K, N, ldbk, ldbn, ldo;
    int vec;             // 16-byte loads of A rows are legal
    int ncb;             // 64-column blocks of this side
    // optional indirection: row m of A is row (row_ids[m] - id_off) of the table A points to
    // (n_table rows); an id outside the table contributes a row of zeros
    const int32_t *row_ids;
    long long id_off, n_table;
};

// table row of logical row m (or -1: a row of zeros)
__device__ __forceinline__ int64_t gemm_src_row(const int32_t *row_ids, long long id_off, long long n_table, int64_t m) {
    if (!row_ids) return m;
    const int64_t r = (int64_t)row_ids[m] - id_off;
    return (uint64_t)r < (uint64_t)n_table ? r : -1;
}

__global__ __launch_bounds__(kThreads)
void rows_gemm_kernel(GemmSide s0, GemmSide s1, int64_t M, int act) {
    __shared__ float As[BM * A_LD];
    __shared__ float Ws[BK * W_LD];
    // pick this block's side field by field (a runtime-selected struct copy would live in scratch)
    const bool second = (int)blockIdx.y >= s0.ncb;
#define GS_SEL(f) (second ? s1.f : s0.f)
    const int col0 = ((int)blockIdx.y - (second ? s0.ncb : 0)) * BN;
    const float *__restrict__ A = GS_SEL(A);
    const float *__restrict__ Mk = GS_SEL(mask);
    const float *__restrict__ B = GS_SEL(B);
    const float *__restrict__ bias = GS_SEL(bias);
    float *__restrict__ outp = GS_SEL(out);
    const int K = GS_SEL(K), N = GS_SEL(N), lda = GS_SEL(lda);
    const int ldbk = GS_SEL(ldbk), ldbn = GS_SEL(ldbn), ldo = GS_SEL(ldo), vec = GS_SEL(vec);
    const int32_t *__restrict__ row_ids = GS_SEL(row_ids);
    const long long id_off = GS_SEL(id_off), n_table = GS_SEL(n_table);
#undef GS_SEL
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, kk = lane >> 4;
    const int ncols = min(BN, N - col0);
    const int nt = (ncols + 15) >> 4;

    f32x4 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < K; k0 += BK) {
        // stage the 64 x 32 slice of A: 8 lanes cover one row's 128 bytes
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int row = pass * 32 + (tid >> 3);
            const int kq = (tid & 7) * 4;
            const int64_t grow = row0 + row;
            const int k = k0 + kq;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            const int64_t srow = grow < M ? gemm_src_row(row_ids, id_off, n_table, grow) : -1;
            if (srow >= 0) {
                const float *src = A + srow * lda + k;
                if (vec && k + 3 < K) {
                    const float4 t = *reinterpret_cast<const float4 *>(src);
                    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                    if (Mk) {
                        const float4 m = *reinterpret_cast<const float4 *>(Mk + srow * lda + k);
                        if (!(m.x > 0.f)) v[0] = 0.f;
                        if (!(m.y > 0.f)) v[1] = 0.f;
                        if (!(m.z > 0.f)) v[2] = 0.f;
                        if (!(m.w > 0.f)) v[3] = 0.f;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (k + i < K) {
                            v[i] = src[i];
                            if (Mk && !(Mk[srow * lda + k + i] > 0.f)) v[i] = 0.f;
                        }
                    }
                }
            }
            float *dst = As + row * A_LD + kq;
            dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
        }
        // stage the 32 x 64 slice of B
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int wk = pass * 16 + (tid >> 4);
            const int c4 = (tid & 15) * 4;
            const int k = k0 + wk;
            float *dst = Ws + wk * W_LD + c4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = col0 + c4 + i;
                dst[i] = (k < K && col < N) ? B[(int64_t)k * ldbk + (int64_t)col * ldbn] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            const float a = As[(wave * 16 + r) * A_LD + ks * 4 + kk];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                if (ct < nt) {
                    const float b = Ws[(ks * 4 + kk) * W_LD + ct * 16 + r];
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[ct], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        if (ct < nt) {
            const int col = col0 + ct * 16 + r;
            if (col < N) {
                const float bv = bias ? bias[col] : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int64_t grow = row0 + wave * 16 + kk * 4 + i;
                    if (grow < M) {
                        float v = acc[ct][i] + bv;
                        if (act == GS_ACT_RELU) v = fmaxf(v, 0.f);
                        outp[grow * ldo + col] = v;
                    }
                }
            }
        }
    }
}


// ---- large-shape forward GEMM ----------------------------------------------
//
// For wide layers (K a multiple of 32, N a multiple of 128 per side) the product is
// bound by the fp32 matrix pipe, not by HBM, and the 64x64 tile above spends most of
// its time staging.  This variant: 128x128 block tile, 4 waves as 2x2, each wave a
// 64x64 tile = 4x4 accumulators; K in steps of 32 through two LDS stages, the
// next slice's global loads in flight while the current one is multiplied.
// LDS holds fragments in lane order: the four k values a lane needs for four
// consecutive MFMA steps are adjacent, so one ds_read_b128 per 16x16 tile feeds
// four MFMAs and a wave's read is one contiguous 1 KB.  Within a 16-wide half of
// the k slice, lane quarter q multiplies k = 4q + step at step 0..3 (any split of
// k over (quarter, step) sums the same products).
constexpr int GB_M = 128, GB_N = 128, GB_K = 32;
constexpr int GB_KQ = GB_K / 4;                         // groups of four k per slice
constexpr int GB_NP = GB_KQ / 2;                        // 16-byte pieces of A (and of B) a thread stages per slice
constexpr int GB_NR = 16 / GB_KQ;                       // rows of A one 16-lane group covers while staging
constexpr int GB_TILE = GB_M * GB_K;                    // floats of one A (or B) stage
constexpr int GB_EP_LD = GB_N + 4;                      // epilogue row stride: the four lane quarters land 16 banks apart
constexpr size_t GB_LDS_LOOP = size_t(4) * GB_TILE * sizeof(float);   // A and B, two stages each
constexpr size_t GB_LDS_EPI = size_t(64) * GB_EP_LD * sizeof(float);
constexpr size_t GB_LDS = GB_LDS_LOOP > GB_LDS_EPI ? GB_LDS_LOOP : GB_LDS_EPI;

__global__ __launch_bounds__(kThreads, 2)
void rows_gemm_big_kernel(GemmSide s0, GemmSide s1, int64_t M, int act) {
    extern __shared__ __align__(16) float gb_smem[];
    const int ncb0 = s0.N / GB_N;
    const bool second = (int)blockIdx.y >= ncb0;
#define GS_SEL(f) (second ? s1.f : s0.f)
    const int col0 = ((int)blockIdx.y - (second ? ncb0 : 0)) * GB_N;
    const float *__restrict__ A = GS_SEL(A);
    const float *__restrict__ Am = GS_SEL(mask);           // element of A counts only where the mask is > 0 (or NULL)
    const float *__restrict__ B = GS_SEL(B);
    const float *__restrict__ bias = GS_SEL(bias);
    float *__restrict__ outp = GS_SEL(out);
    const int K = GS_SEL(K), lda = GS_SEL(lda), ldbk = GS_SEL(ldbk), ldbn = GS_SEL(ldbn), ldo = GS_SEL(ldo);
    const int32_t *__restrict__ row_ids = GS_SEL(row_ids);
    const long long id_off = GS_SEL(id_off), n_table = GS_SEL(n_table);
#undef GS_SEL
    const int64_t row0 = (int64_t)blockIdx.x * GB_M;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wave_m = wave & 1, wave_n = wave >> 1;

    // staging roles.  A: GB_KQ lanes cover one row's slice (16 bytes each), 256/GB_KQ rows per pass.
    const int a_kq = tid % GB_KQ;
    const float *a_src[GB_NP];
    const float *a_msk[GB_NP];                          // the same elements of the mask
    bool a_zero[GB_NP];
    int a_dst[GB_NP];
#pragma unroll
    for (int p = 0; p < GB_NP; ++p) {
        const int row = p * (kThreads / GB_KQ) + tid / GB_KQ;
        int64_t grow = row0 + row;
        if (grow > M - 1) grow = M - 1;                 // rows past the end repeat the last one; never stored
        const int64_t srow = gemm_src_row(row_ids, id_off, n_table, grow);
        a_zero[p] = srow < 0;                           // an id outside the table: the row counts as zeros
        a_src[p] = A + (srow < 0 ? 0 : srow) * lda + a_kq * 4;
        a_msk[p] = Am ? Am + (srow < 0 ? 0 : srow) * lda + a_kq * 4 : nullptr;
        // within a fragment block the 16 slots of a lane quarter are rotated by GB_NR * (k quad): the
        // 16-lane groups of this write (GB_NR rows x GB_KQ k quads) and of the fragment read (sixteen
        // rows of one quad) then both touch sixteen different 16-byte bank groups
        a_dst[p] = ((a_kq >> 2) * 8 + (row >> 4)) * 256 +
                   ((a_kq & 3) * 16 + (((row & 15) + GB_NR * a_kq) & 15)) * 4;
    }
    // B: a thread gathers four consecutive k of one column (4 scalar loads, coalesced over columns)
    const float *b_src[GB_NP];
    int b_dst[GB_NP];
#pragma unroll
    for (int j = 0; j < GB_NP; ++j) {
        const int idx = j * 256 + tid;
        const int col = idx & 127, kquad = idx >> 7;
        b_src[j] = B + (int64_t)(kquad * 4) * ldbk + (int64_t)(col0 + col) * ldbn;
        b_dst[j] = ((kquad >> 2) * 8 + (col >> 4)) * 256 + ((kquad & 3) * 16 + (col & 15)) * 4;
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this lane's (rotated) slot in an A fragment block, per half of the k slice
    const int a_slot[2] = {(lane >> 4) * 16 + (((lane & 15) + GB_NR * (lane >> 4)) & 15),
                           (lane >> 4) * 16 + (((lane & 15) + GB_NR * ((lane >> 4) + 4)) & 15)};

    // the slice in flight lives in named registers (an indexed array here ends up in scratch)
    float4 ar0, ar1, ar2 = make_float4(0.f, 0.f, 0.f, 0.f), ar3 = ar2, br0, br1, br2 = ar2, br3 = ar2;
    float4 am0 = make_float4(1.f, 1.f, 1.f, 1.f), am1 = am0, am2 = am0, am3 = am0;   // mask values of the slice in flight
#define GB_LOAD_B(dst_, j_, k0_)                                                                   \
    {                                                                                              \
        const float *src = b_src[j_] + (int64_t)(k0_) * ldbk;                                      \
        dst_ = make_float4(src[0], src[ldbk], src[2 * (int64_t)ldbk], src[3 * (int64_t)ldbk]);     \
    }
#define GB_LOAD_SLICE(k0_)                                                                         \
    {                                                                                              \
        const int k0 = (k0_);                                                                      \
        ar0 = *reinterpret_cast<const float4 *>(a_src[0] + k0);                                    \
        ar1 = *reinterpret_cast<const float4 *>(a_src[1] + k0);                                    \
        if constexpr (GB_NP == 4) {                                                                \
            ar2 = *reinterpret_cast<const float4 *>(a_src[GB_NP - 2] + k0);                        \
            ar3 = *reinterpret_cast<const float4 *>(a_src[GB_NP - 1] + k0);                        \
        }                                                                                          \
        if (Am) {                                            /* applied when the slice is stored */ \
            am0 = *reinterpret_cast<const float4 *>(a_msk[0] + k0);                    \
            am1 = *reinterpret_cast<const float4 *>(a_msk[1] + k0);                    \
            if constexpr (GB_NP == 4) {                                                            \
                am2 = *reinterpret_cast<const float4 *>(a_msk[GB_NP - 2] + k0); \
                am3 = *reinterpret_cast<const float4 *>(a_msk[GB_NP - 1] + k0); \
            }                                                                                      \
        }                                                                                          \
        if (row_ids) {                                                                             \
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);                                     \
            if (a_zero[0]) ar0 = z4;                                                               \
            if (a_zero[1]) ar1 = z4;                                                               \
            if constexpr (GB_NP == 4) { if (a_zero[GB_NP - 2]) ar2 = z4; if (a_zero[GB_NP - 1]) ar3 = z4; } \
        }                                                                                          \
        GB_LOAD_B(br0, 0, k0) GB_LOAD_B(br1, 1, k0)                                                \
        if constexpr (GB_NP == 4) { GB_LOAD_B(br2, GB_NP - 2, k0) GB_LOAD_B(br3, GB_NP - 1, k0) }  \
    }
#define GB_MASK(a_, m_)                                                                            \
    if (!(m_.x > 0.f)) a_.x = 0.f;                                                                 \
    if (!(m_.y > 0.f)) a_.y = 0.f;                                                                 \
    if (!(m_.z > 0.f)) a_.z = 0.f;                                                                 \
    if (!(m_.w > 0.f)) a_.w = 0.f;
#define GB_STORE_SLICE(stage_)                                                                     \
    {                                                                                              \
        if (Am) {                                                                                  \
            GB_MASK(ar0, am0) GB_MASK(ar1, am1)                                                    \
            if constexpr (GB_NP == 4) { GB_MASK(ar2, am2) GB_MASK(ar3, am3) }                      \
        }                                                                                          \
        float *As_w = gb_smem + (stage_) * 2 * GB_TILE, *Bs_w = As_w + GB_TILE;                    \
        *reinterpret_cast<float4 *>(As_w + a_dst[0]) = ar0;                                        \
        *reinterpret_cast<float4 *>(As_w + a_dst[1]) = ar1;                                        \
        *reinterpret_cast<float4 *>(Bs_w + b_dst[0]) = br0;                                        \
        *reinterpret_cast<float4 *>(Bs_w + b_dst[1]) = br1;                                        \
        if constexpr (GB_NP == 4) {                                                                \
            *reinterpret_cast<float4 *>(As_w + a_dst[GB_NP - 2]) = ar2;                            \
            *reinterpret_cast<float4 *>(As_w + a_dst[GB_NP - 1]) = ar3;                            \
            *reinterpret_cast<float4 *>(Bs_w + b_dst[GB_NP - 2]) = br2;                            \
            *reinterpret_cast<float4 *>(Bs_w + b_dst[GB_NP - 1]) = br3;                            \
        }                                                                                          \
    }

    // One ds_read_b128 per 16x16 fragment feeds four MFMAs; the next slice's global loads are
    // issued before the multiply and stored to the other LDS stage after it.  (A variant that
    // also kept a second set of fragment registers to read LDS a phase ahead ran no faster alone
    // and, at 226 instead of 167 registers, left less room for a gather kernel beside it.)
    f32x4 aX0, aX1, aX2, aX3, bX0, bX1, bX2, bX3;
#define GB_READ(S_, stage_, half_)                                                                  \
    {                                                                                              \
        const float *As_r = gb_smem + (stage_) * 2 * GB_TILE + ((half_) * 8 + wave_m * 4) * 256 + a_slot[half_] * 4; \
        const float *Bs_r = gb_smem + (stage_) * 2 * GB_TILE + GB_TILE + ((half_) * 8 + wave_n * 4) * 256 + lane * 4; \
        a##S_##0 = *reinterpret_cast<const f32x4 *>(As_r);                                         \
        b##S_##0 = *reinterpret_cast<const f32x4 *>(Bs_r);                                         \
        a##S_##1 = *reinterpret_cast<const f32x4 *>(As_r + 256);                                   \
        b##S_##1 = *reinterpret_cast<const f32x4 *>(Bs_r + 256);                                   \
        a##S_##2 = *reinterpret_cast<const f32x4 *>(As_r + 512);                                   \
        b##S_##2 = *reinterpret_cast<const f32x4 *>(Bs_r + 512);                                   \
        a##S_##3 = *reinterpret_cast<const f32x4 *>(As_r + 768);                                   \
        b##S_##3 = *reinterpret_cast<const f32x4 *>(Bs_r + 768);                                   \
    }
#define GB_MFMA_ROW(i_, av_, S_, st_)                                                              \
    acc[i_][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[st_], b##S_##0[st_], acc[i_][0], 0, 0, 0); \
    acc[i_][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[st_], b##S_##1[st_], acc[i_][1], 0, 0, 0); \
    acc[i_][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[st_], b##S_##2[st_], acc[i_][2], 0, 0, 0); \
    acc[i_][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[st_], b##S_##3[st_], acc[i_][3], 0, 0, 0);
#define GB_MFMA(S_)                                                                                \
    _Pragma("unroll") for (int st = 0; st < 4; ++st) {                                             \
        GB_MFMA_ROW(0, a##S_##0, S_, st) GB_MFMA_ROW(1, a##S_##1, S_, st)                          \
        GB_MFMA_ROW(2, a##S_##2, S_, st) GB_MFMA_ROW(3, a##S_##3, S_, st)                          \
    }

    static_assert(GB_K == 32, "the main loop walks two 16-wide halves per slice");
    GB_LOAD_SLICE(0);
    GB_STORE_SLICE(0);
    __syncthreads();
    const int n_slices = K / GB_K;
    for (int kt = 0; kt < n_slices; ++kt) {
        const int stage = kt & 1;
        // next slice in flight during the multiply; the last iteration re-reads its own slice (unused)
        GB_LOAD_SLICE(min(kt + 1, n_slices - 1) * GB_K);
        __builtin_amdgcn_sched_barrier(0);  // keep the loads ahead of the multiply (the scheduler would sink them)
        GB_READ(X, stage, 0);
        GB_MFMA(X);
        GB_READ(X, stage, 1);
        GB_MFMA(X);
        __builtin_amdgcn_sched_barrier(0);
        GB_STORE_SLICE(stage ^ 1);          // that stage was last read before the previous barrier
        __syncthreads();
    }
#undef GB_READ
#undef GB_MFMA_ROW
#undef GB_MFMA
#undef GB_LOAD_SLICE
#undef GB_LOAD_B
#undef GB_STORE_SLICE
#undef GB_MASK

    // Epilogue through LDS, 64 rows at a time: accumulators (+ bias, activation) are laid out as
    // rows of the output tile, then every thread writes 16-byte pieces of whole rows (a row of the
    // tile is 512 contiguous bytes of the output) instead of sixty-four 4-byte stores per lane.
    constexpr int EP_LD = GB_EP_LD;
    const int r = lane & 15, q = lane >> 4;
    float bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = bias ? bias[col0 + wave_n * 64 + j * 16 + r] : 0.f;
#pragma unroll 1
    for (int hm = 0; hm < 2; ++hm) {
        if (wave_m == hm) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[i][j][e] + bv[j];
                        if (act == GS_ACT_RELU) v = fmaxf(v, 0.f);
                        gb_smem[(i * 16 + q * 4 + e) * EP_LD + wave_n * 64 + j * 16 + r] = v;
                    }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * kThreads + tid;          // 64 rows x 32 float4
            const int row = idx >> 5, c4 = (idx & 31) * 4;
            const int64_t grow = row0 + hm * 64 + row;
            if (grow < M)
                *reinterpret_cast<float4 *>(outp + grow * ldo + col0 + c4) =
                    *reinterpret_cast<const float4 *>(gb_smem + row * EP_LD + c4);
        }
        __syncthreads();
    }
}

inline bool big_gemm_applies(const GemmSide &s) {
    return s.vec && s.K % GB_K == 0 && s.N % GB_N == 0 && !(s.mask && s.row_ids) &&
           (reinterpret_cast<uintptr_t>(s.A) & 15) == 0 && s.lda % 4 == 0 &&
           (!s.mask || (reinterpret_cast<uintptr_t>(s.mask) & 15) == 0) &&
           (reinterpret_cast<uintptr_t>(s.out) & 15) == 0 && s.ldo % 4 == 0;
}

// ---- weight / bias gradient ------------------------------------------------

struct WgradSide {
    const float *X;      // [M, K] row stride K (the forward input of this half)
    const float *G;      // output gradient of this half, row stride ldg
    const float *mask;   // forward output of this half (relu mask) or NULL, row stride ldg
    float *slab;         // [n_chunks][K + 1][N] partial sums
    int K, N, ldg;
    int nkb, ncb;        // 64-row blocks over K+1, 64-column blocks over N
};

constexpr int WG_ROWS = 64;            // rows of M staged per step

__global__ __launch_bounds__(kThreads)
void wgrad_kernel(WgradSide s0, WgradSide s1, int64_t M, int64_t rows_per_chunk, int vec0, int vec1) {
    __shared__ float Xs[WG_ROWS * W_LD];
    __shared__ float Gs[WG_ROWS * W_LD];
    const int tiles0 = s0.nkb * s0.ncb;
    const bool second = (int)blockIdx.y >= tiles0;
#define GS_SEL(f) (second ? s1.f : s0.f)
    const float *__restrict__ X = GS_SEL(X);
    const float *__restrict__ G = GS_SEL(G);
    const float *__restrict__ Mk = GS_SEL(mask);
    float *__restrict__ slab_base = GS_SEL(slab);
    const int K = GS_SEL(K), N = GS_SEL(N), ldg = GS_SEL(ldg), ncb = GS_SEL(ncb);
#undef GS_SEL
    const bool vec = second ? vec1 : vec0;         // 16-byte loads of X, G and mask rows are legal
    const int tile = (int)blockIdx.y - (second ? tiles0 : 0);
    const int kb0 = (tile / ncb) * 64;            // first row of dW (k index) of this block
    const int col0 = (tile % ncb) * 64;
    const int64_t chunk = blockIdx.x;
    const int64_t m_begin = chunk * rows_per_chunk;
    const int64_t m_end = min(M, m_begin + rows_per_chunk);
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, q = lane >> 4;
    const int nt = (min(64, N - col0) + 15) >> 4;
    const bool bias_block = kb0 == 0;             // this block also forms the column sums (bias gradient)

    f32x4 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    for (int64_t m0 = m_begin; m0 < m_end; m0 += WG_ROWS) {
        // 64 rows x 64 columns of X and of the masked gradient; 16 lanes cover one row
#pragma unroll
        for (int pass = 0; pass < WG_ROWS / 16; ++pass) {
            const int row = pass * 16 + (tid >> 4);
            const int c4 = (tid & 15) * 4;
            const int64_t m = m0 + row;
            const bool live = m < m_end;
            const int k = kb0 + c4, c = col0 + c4;
            float xv[4] = {0.f, 0.f, 0.f, 0.f}, gv[4] = {0.f, 0.f, 0.f, 0.f};
            if (live) {
                if (vec && k + 3 < K) {
                    const float4 t = *reinterpret_cast<const float4 *>(X + m * K + k);
                    xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (k + i < K) xv[i] = X[m * K + k + i];
                }
                if (vec && c + 3 < N) {
                    const float4 t = *reinterpret_cast<const float4 *>(G + m * ldg + c);
                    gv[0] = t.x; gv[1] = t.y; gv[2] = t.z; gv[3] = t.w;
                    if (Mk) {
                        const float4 mk = *reinterpret_cast<const float4 *>(Mk + m * ldg + c);
                        if (!(mk.x > 0.f)) gv[0] = 0.f;
                        if (!(mk.y > 0.f)) gv[1] = 0.f;
                        if (!(mk.z > 0.f)) gv[2] = 0.f;
                        if (!(mk.w > 0.f)) gv[3] = 0.f;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (c + i < N) {
                            gv[i] = G[m * ldg + c + i];
                            if (Mk && !(Mk[m * ldg + c + i] > 0.f)) gv[i] = 0.f;
                        }
                    }
                }
            }
            *reinterpret_cast<float4 *>(Xs + row * W_LD + c4) = make_float4(xv[0], xv[1], xv[2], xv[3]);
            *reinterpret_cast<float4 *>(Gs + row * W_LD + c4) = make_float4(gv[0], gv[1], gv[2], gv[3]);
        }
        __syncthreads();
#pragma unroll
        for (int ms = 0; ms < WG_ROWS / 4; ++ms) {
            const float a = Xs[(ms * 4 + q) * W_LD + wave * 16 + r];        // A[k][m] = X[m][k]
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                if (ct < nt) {
                    const float b = Gs[(ms * 4 + q) * W_LD + ct * 16 + r];
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[ct], 0, 0, 0);
                }
            }
        }
        if (bias_block && wave == 0) {             // lane = column: sum the staged rows in row order
#pragma unroll 8
            for (int row = 0; row < WG_ROWS; ++row) bsum += Gs[row * W_LD + lane];
        }
        __syncthreads();
    }

    float *slab = slab_base + chunk * (int64_t)(K + 1) * N;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        if (ct < nt) {
            const int col = col0 + ct * 16 + r;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = kb0 + wave * 16 + q * 4 + i;
                if (k < K && col < N) slab[(int64_t)k * N + col] = acc[ct][i];
            }
        }
    }
    if (bias_block && wave == 0 && col0 + lane < N) slab[(int64_t)K * N + col0 + lane] = bsum;
}

// Weight gradient of wide layers (K and N multiples of 128): a block forms a 128 x 128 tile of
// dW = X^T G over its chunk of rows on the four-wave 64x64 MFMA scheme of rows_gemm_big_kernel.
// The row index m is the reduction index here, so BOTH operands sit in LDS the way that kernel
// keeps its weight tile: for each column, four consecutive m adjacent, one ds_read_b128 per 16x16
// fragment feeding four MFMAs.  A thread fetches a 4 row x 4 column patch of X and of G (four
// 16-byte loads each) and writes it out transposed, so nothing is loaded twice and no load is
// narrower than 16 bytes.  Same slab layout and reduction as wgrad_kernel (bitwise reproducible).
constexpr int WB_ROWS = 32;                               // rows of M per slice
constexpr int WB_TILE = WB_ROWS * 128;                    // floats of one operand slice
constexpr size_t WB_LDS = size_t(4) * WB_TILE * sizeof(float);   // X and G, two stages each

__global__ __launch_bounds__(kThreads, 2)
void wgrad_big_kernel(WgradSide s0, WgradSide s1, int64_t M, int64_t rows_per_chunk) {
    extern __shared__ __align__(16) float wb_smem[];
    const int tiles0 = (s0.K / 128) * (s0.N / 128);
    // Blocks are numbered so that the tiles of one chunk of rows -- which read the same rows of G and of the
    // mask -- start together and on one XCD (consecutive blocks go to the eight XCDs in turn): groups of eight
    // chunks, within a group tile by tile.  The second reader of a row then finds it in that XCD's L2.
    const int n_tiles_all = tiles0 + (s1.K / 128) * (s1.N / 128);
    const int64_t n_chunks = (M + rows_per_chunk - 1) / rows_per_chunk;
    const int64_t group = (int64_t)blockIdx.x / (8 * n_tiles_all);
    const int in_group = (int)((int64_t)blockIdx.x % (8 * n_tiles_all));
    const int tile_all = in_group >> 3;
    const int64_t chunk = group * 8 + (in_group & 7);
    if (chunk >= n_chunks) return;                               // (the last group is padded to eight chunks)
    const bool second = tile_all >= tiles0;
#define GS_SEL(f) (second ? s1.f : s0.f)
    const float *__restrict__ X = GS_SEL(X);
    const float *__restrict__ G = GS_SEL(G);
    const float *__restrict__ Mk = GS_SEL(mask);
    float *__restrict__ slab_base = GS_SEL(slab);
    const int K = GS_SEL(K), N = GS_SEL(N), ldg = GS_SEL(ldg);
#undef GS_SEL
    const int ncb = N / 128;
    const int tile = tile_all - (second ? tiles0 : 0);
    const int kb0 = (tile / ncb) * 128, col0 = (tile % ncb) * 128;
    const int64_t m_begin = chunk * rows_per_chunk;
    const int64_t m_end = min(M, m_begin + rows_per_chunk);      // > m_begin: the grid has no empty chunks
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wave_m = wave & 1, wave_n = wave >> 1;
    const bool bias_block = kb0 == 0;                            // this block also forms the column sums

    // loader role: rows 4 mq .. 4 mq + 3 of the slice, columns 4 cq .. 4 cq + 3 of the tile.  Sixteen
    // consecutive lanes take 4 column quads x 4 row quads; with the 16-byte slots of a row quad rotated
    // by the row quad's index their transposed writes (and the fragment reads, which undo the rotation)
    // fall on sixteen different bank groups.
    const int mq = (wave >> 1) * 4 + ((lane >> 2) & 3);
    const int cq = (wave & 1) * 16 + ((lane >> 4) & 3) * 4 + (lane & 3);
    const int blk0 = ((mq >> 2) * 8 + (cq >> 2)) * 256 + (mq & 3) * 64;
    int dstf[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) dstf[f] = blk0 + (((cq & 3) * 4 + f + (mq & 3)) & 15) * 4;
    const float *__restrict__ xp = X + kb0 + cq * 4;
    const float *__restrict__ gp = G + col0 + cq * 4;
    const float *__restrict__ kp = Mk ? Mk + col0 + cq * 4 : nullptr;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);                 // column sums of this thread's patch rows
    float4 x0, x1, x2, x3, g0, g1, g2, g3;

    // A slice is only loaded here; the relu mask and the zeroing of rows past the end are applied when it
    // is written to LDS, after the multiply of the slice before it: touching the values any earlier would
    // put the wait for the loads in front of that multiply.
    float4 k0 = make_float4(1.f, 1.f, 1.f, 1.f), k1 = k0, k2 = k0, k3 = k0;
    int64_t m_loaded = m_begin;
#define WB_LOAD_ROW(xr_, gr_, kr_, e_)                                                              \
    {                                                                                               \
        const int64_t m = m0 + mq * 4 + (e_);                                                       \
        const int64_t mc = m < m_end ? m : m_end - 1;           /* rows past the end: loaded, zeroed later */ \
        xr_ = *reinterpret_cast<const float4 *>(xp + mc * K);                                       \
        gr_ = *reinterpret_cast<const float4 *>(gp + mc * ldg);                                     \
        if (kp) kr_ = *reinterpret_cast<const float4 *>(kp + mc * ldg);                             \
    }
#define WB_LOAD(m0_)                                                                                \
    {                                                                                               \
        const int64_t m0 = (m0_);                                                                   \
        m_loaded = m0;                                                                              \
        WB_LOAD_ROW(x0, g0, k0, 0) WB_LOAD_ROW(x1, g1, k1, 1) WB_LOAD_ROW(x2, g2, k2, 2) WB_LOAD_ROW(x3, g3, k3, 3) \
    }
#define WB_FIX_ROW(xr_, gr_, kr_, e_)                                                               \
    {                                                                                               \
        const bool live = m_loaded + mq * 4 + (e_) < m_end;                                         \
        if (!(live && kr_.x > 0.f)) gr_.x = 0.f;                                                    \
        if (!(live && kr_.y > 0.f)) gr_.y = 0.f;                                                    \
        if (!(live && kr_.z > 0.f)) gr_.z = 0.f;                                                    \
        if (!(live && kr_.w > 0.f)) gr_.w = 0.f;                                                    \
        if (!live) xr_ = make_float4(0.f, 0.f, 0.f, 0.f);                                           \
    }
#define WB_STORE(stage_)                                                                            \
    {                                                                                               \
        WB_FIX_ROW(x0, g0, k0, 0) WB_FIX_ROW(x1, g1, k1, 1) WB_FIX_ROW(x2, g2, k2, 2) WB_FIX_ROW(x3, g3, k3, 3) \
        float *Xs = wb_smem + (stage_) * 2 * WB_TILE, *Gs = Xs + WB_TILE;                           \
        *reinterpret_cast<float4 *>(Xs + dstf[0]) = make_float4(x0.x, x1.x, x2.x, x3.x);            \
        *reinterpret_cast<float4 *>(Xs + dstf[1]) = make_float4(x0.y, x1.y, x2.y, x3.y);            \
        *reinterpret_cast<float4 *>(Xs + dstf[2]) = make_float4(x0.z, x1.z, x2.z, x3.z);            \
        *reinterpret_cast<float4 *>(Xs + dstf[3]) = make_float4(x0.w, x1.w, x2.w, x3.w);            \
        *reinterpret_cast<float4 *>(Gs + dstf[0]) = make_float4(g0.x, g1.x, g2.x, g3.x);            \
        *reinterpret_cast<float4 *>(Gs + dstf[1]) = make_float4(g0.y, g1.y, g2.y, g3.y);            \
        *reinterpret_cast<float4 *>(Gs + dstf[2]) = make_float4(g0.z, g1.z, g2.z, g3.z);            \
        *reinterpret_cast<float4 *>(Gs + dstf[3]) = make_float4(g0.w, g1.w, g2.w, g3.w);            \
        bs.x += ((g0.x + g1.x) + g2.x) + g3.x;                                                      \
        bs.y += ((g0.y + g1.y) + g2.y) + g3.y;                                                      \
        bs.z += ((g0.z + g1.z) + g2.z) + g3.z;                                                      \
        bs.w += ((g0.w + g1.w) + g2.w) + g3.w;                                                      \
    }
    f32x4 aX0, aX1, aX2, aX3, bX0, bX1, bX2, bX3;
    const int rd_slot = (lane >> 4) * 16 + (((lane & 15) + (lane >> 4)) & 15);   // this lane's (rotated) slot in a fragment block
#define WB_READ(stage_, half_)                                                                      \
    {                                                                                               \
        const float *As_r = wb_smem + (stage_) * 2 * WB_TILE + ((half_) * 8 + wave_m * 4) * 256 + rd_slot * 4; \
        const float *Bs_r = wb_smem + (stage_) * 2 * WB_TILE + WB_TILE + ((half_) * 8 + wave_n * 4) * 256 + rd_slot * 4; \
        aX0 = *reinterpret_cast<const f32x4 *>(As_r);                                               \
        bX0 = *reinterpret_cast<const f32x4 *>(Bs_r);                                               \
        aX1 = *reinterpret_cast<const f32x4 *>(As_r + 256);                                         \
        bX1 = *reinterpret_cast<const f32x4 *>(Bs_r + 256);                                         \
        aX2 = *reinterpret_cast<const f32x4 *>(As_r + 512);                                         \
        bX2 = *reinterpret_cast<const f32x4 *>(Bs_r + 512);                                         \
        aX3 = *reinterpret_cast<const f32x4 *>(As_r + 768);                                         \
        bX3 = *reinterpret_cast<const f32x4 *>(Bs_r + 768);                                         \
    }
#define WB_MFMA_ROW(i_, av_, st_)                                                                   \
    acc[i_][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[st_], bX0[st_], acc[i_][0], 0, 0, 0);     \
    acc[i_][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[st_], bX1[st_], acc[i_][1], 0, 0, 0);     \
    acc[i_][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[st_], bX2[st_], acc[i_][2], 0, 0, 0);     \
    acc[i_][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[st_], bX3[st_], acc[i_][3], 0, 0, 0);
#define WB_MFMA()                                                                                   \
    _Pragma("unroll") for (int st = 0; st < 4; ++st) {                                              \
        WB_MFMA_ROW(0, aX0, st) WB_MFMA_ROW(1, aX1, st) WB_MFMA_ROW(2, aX2, st) WB_MFMA_ROW(3, aX3, st) \
    }

    const int n_slices = (int)((m_end - m_begin + WB_ROWS - 1) / WB_ROWS);
    WB_LOAD(m_begin);
    WB_STORE(0);
    __syncthreads();
    for (int kt = 0; kt < n_slices; ++kt) {
        const int stage = kt & 1;
        const bool more = kt + 1 < n_slices;
        if (more) WB_LOAD(m_begin + (int64_t)(kt + 1) * WB_ROWS);   // in flight during the multiply
        __builtin_amdgcn_sched_barrier(0);
        WB_READ(stage, 0);
        WB_MFMA();
        WB_READ(stage, 1);
        WB_MFMA();
        __builtin_amdgcn_sched_barrier(0);
        if (more) WB_STORE(stage ^ 1);             // that stage was last read before the previous barrier
        __syncthreads();
    }
#undef WB_LOAD_ROW
#undef WB_FIX_ROW
#undef WB_LOAD
#undef WB_STORE
#undef WB_READ
#undef WB_MFMA_ROW
#undef WB_MFMA

    float *slab = slab_base + chunk * (int64_t)(K + 1) * N;
    const int r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                slab[(int64_t)(kb0 + wave_m * 64 + i * 16 + q * 4 + e) * N + col0 + wave_n * 64 + j * 16 + r] = acc[i][j][e];
    if (bias_block) {                              // eight threads hold partial sums of a column: added in row order
        float *part = wb_smem;                     // [8][128]  (the loop ended with a barrier)
        *reinterpret_cast<float4 *>(part + mq * 128 + cq * 4) = bs;
        __syncthreads();
        if (tid < 128) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) t += part[g * 128 + tid];
            slab[(int64_t)K * N + col0 + tid] = t;
        }
    }
}

inline bool wgrad_big_applies(int Fs, int Fn, int Hh, int H, const void *xs, const void *xm, const void *g, const void *mask) {
    auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return Fs % 128 == 0 && Fn % 128 == 0 && Hh % 128 == 0 && H % 4 == 0 && al16(xs) && al16(xm) && al16(g) && (!mask || al16(mask));
}

// Adds the per-chunk slabs of both halves (of one or two problem instances) in one launch (blockIdx.y = side).  Element e is summed
// by sixteen threads that take chunks c = j, j+16, ... each in increasing order; their partial
// sums are then added in index order: the result does not depend on scheduling.
struct ReduceSide {
    const float *slab;
    float *gW, *gbias;
    int K;
};

struct ReduceSides {
    ReduceSide s[4];                           // blockIdx.y picks one (two halves of one or two problem instances)
};

__global__ __launch_bounds__(kThreads)
void wgrad_reduce_kernel(const ReduceSides sides, int n_chunks, int N) {
    __shared__ float part[kThreads];
    const ReduceSide &sd = sides.s[blockIdx.y];
    const float *__restrict__ slab = sd.slab;
    float *__restrict__ gW = sd.gW;
    float *__restrict__ gbias = sd.gbias;
    const int K = sd.K;
    const int total = (K + 1) * N;
    const int e = blockIdx.x * 16 + (threadIdx.x & 15);
    const int j = threadIdx.x >> 4;
    float s = 0.f;
    if (e < total)
        for (int c = j; c < n_chunks; c += 16) s += slab[(int64_t)c * total + e];
    part[threadIdx.x] = s;
    __syncthreads();
    if (j == 0 && e < total) {
        s = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) s += part[g * 16 + threadIdx.x];
        if (e < K * N) { if (gW) gW[e] = s; }
        else if (gbias) gbias[e - K * N] = s;
    }
}

// The same with four consecutive elements per thread and four interleaved chunk groups per block (N a multiple of
// four, slabs 16-byte aligned): a block reads 1 KB of every chunk's row, eight loads in flight per thread.  Chunk c
// goes to group c mod 4, the groups are added in order: reproducible like the form above (which keeps the shapes
// this one does not take).
__global__ __launch_bounds__(kThreads)
void wgrad_reduce4_kernel(const ReduceSides sides, int n_chunks, int N) {
    __shared__ f32x4 part[kThreads];
    const ReduceSide &sd = sides.s[blockIdx.y];
    const float *__restrict__ slab = sd.slab;
    const int K = sd.K;
    const int total = (K + 1) * N;
    const int e = 4 * (blockIdx.x * 64 + (threadIdx.x & 63));
    const int j = threadIdx.x >> 6;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (e < total) {
        const float *src = slab + e;
        int c = j;
        for (; c + 28 < n_chunks; c += 32) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4 *>(src + (int64_t)(c + 4 * u) * total);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; c < n_chunks; c += 4) acc += *reinterpret_cast<const f32x4 *>(src + (int64_t)c * total);
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (j == 0 && e < total) {
        const int l = threadIdx.x;
        const f32x4 t = ((part[l] + part[64 + l]) + part[128 + l]) + part[192 + l];
        if (e < K * N) { if (sd.gW) *reinterpret_cast<f32x4 *>(sd.gW + e) = t; }      // (K * N is a multiple of four: no straddling)
        else if (sd.gbias) {
#pragma unroll
            for (int i = 0; i < 4; ++i) sd.gbias[e - K * N + i] = t[i];
        }
    }
}

inline void launch_wgrad_reduce(const ReduceSides &rs, int n_sides, int tot, int n_chunks, int N, hipStream_t s) {
    bool vec = (N & 3) == 0;
    for (int i = 0; i < n_sides; ++i)
        vec = vec && (reinterpret_cast<uintptr_t>(rs.s[i].slab) & 15) == 0 &&
              (!rs.s[i].gW || (reinterpret_cast<uintptr_t>(rs.s[i].gW) & 15) == 0);
    if (vec) hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3((unsigned)((tot + 255) / 256), n_sides), dim3(kThreads), 0, s, rs, n_chunks, N);
    else hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((tot + 15) / 16), n_sides), dim3(kThreads), 0, s, rs, n_chunks, N);
}


// ---- weight / bias gradient for narrow outputs, operands straight from global memory --------
//
// dW[k, n] = sum_m X[m, k] * G[m, n] with K in {32, 64, 128} per half and a half-width of 16
// or 32 is bound by reading X and G once, not by the matrix pipe.  One MFMA step covers four
// rows of M: lane (r, q) holds X[m0 + q][V*r .. V*r + V-1] (V = K/16 consecutive floats, one
// vector load; a wave reads whole rows, contiguous) and G[m0 + q][r].  Accumulator t then holds
// the weight rows k = V*i + t: V accumulators cover all K.  Both halves (self, neighbor) run in
// the same wave, so the gradient row is read once.  A block's four waves take consecutive row
// ranges and add their accumulators through LDS in wave order; wgrad_reduce_kernel adds the
// blocks' slabs in block order -- bitwise reproducible like the tiled kernel.
template <int V> struct XVec;
template <> struct XVec<2> { float v[2]; __device__ __forceinline__ void load(const float *p) { const float2 t = *reinterpret_cast<const float2 *>(p); v[0] = t.x; v[1] = t.y; } };
template <> struct XVec<4> { float v[4]; __device__ __forceinline__ void load(const float *p) { const float4 t = *reinterpret_cast<const float4 *>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; } };
template <> struct XVec<8> { float v[8]; __device__ __forceinline__ void load(const float *p) {
    const float4 t = *reinterpret_cast<const float4 *>(p), u = *reinterpret_cast<const float4 *>(p + 4);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; v[4] = u.x; v[5] = u.y; v[6] = u.z; v[7] = u.w; } };

constexpr int WD_UNROLL = 4;            // MFMA steps (of four rows) whose loads are in flight together

// One problem instance: inputs, output gradient, relu mask (or NULL), the two slab arrays.
struct WgradDirectSide {
    const float *Xs, *Xn, *G, *Mk;
    float *slab_s, *slab_n;
    // ids != NULL: row m of the own-row operand is row ids[m] - off of the table Xs (rows outside
    // [0, rows) count as zeros), so the forward pass need not keep a gathered copy of those rows
    const int32_t *ids;
    long long off, rows;
};

// blockIdx.y picks the instance: the two node types of a layer have the same shapes, and one launch
// over both costs no fork and join of streams.
template <int KS, int KN, int NT>
__global__ __launch_bounds__(kThreads)
void wgrad_direct_kernel(const WgradDirectSide p0, const WgradDirectSide p1, int64_t M, int64_t rows_per_block) {
    constexpr int VS = KS / 16, VN = KN / 16, HH = NT * 16, H = 2 * HH;
    const bool second = blockIdx.y != 0;
    const float *__restrict__ Xs = second ? p1.Xs : p0.Xs;
    const float *__restrict__ Xn = second ? p1.Xn : p0.Xn;
    const float *__restrict__ G = second ? p1.G : p0.G;
    const float *__restrict__ Mk = second ? p1.Mk : p0.Mk;
    float *__restrict__ slab_s = second ? p1.slab_s : p0.slab_s;
    float *__restrict__ slab_n = second ? p1.slab_n : p0.slab_n;
    const int32_t *__restrict__ ids = second ? p1.ids : p0.ids;
    const long long id_off = second ? p1.off : p0.off, id_rows = second ? p1.rows : p0.rows;
    constexpr int kSlabS = (KS + 1) * HH, kSlabN = (KN + 1) * HH;
    __shared__ float red[kSlabS + kSlabN];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, q = lane >> 4;
    const int64_t b_begin = (int64_t)blockIdx.x * rows_per_block;
    const int64_t b_end = min(M, b_begin + rows_per_block);
    // consecutive quarter ranges (multiples of 4 rows) for the four waves
    const int64_t per_wave = ((b_end - b_begin + 15) / 16) * 4;
    const int64_t w_begin = min(b_end, b_begin + wave * per_wave), w_end = min(b_end, w_begin + per_wave);

    f32x4 acc_s[VS][NT], acc_n[VN][NT];
    float bias_s[NT], bias_n[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        bias_s[nt] = bias_n[nt] = 0.f;
#pragma unroll
        for (int t = 0; t < VS; ++t) acc_s[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < VN; ++t) acc_n[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // ids of the rows of an iteration are fetched one iteration ahead (a dependent read otherwise)
    int32_t idc[WD_UNROLL];
#pragma unroll
    for (int u = 0; u < WD_UNROLL; ++u) {
        idc[u] = 0;
        if (ids && w_begin < w_end) idc[u] = ids[min(w_begin + 4 * u + q, w_end - 1)];
    }
    for (int64_t m0 = w_begin; m0 < w_end; m0 += 4 * WD_UNROLL) {
        XVec<VS> xs[WD_UNROLL];
        XVec<VN> xn[WD_UNROLL];
        float gs_[WD_UNROLL][NT], gn_[WD_UNROLL][NT];
        int32_t idn[WD_UNROLL];
#pragma unroll
        for (int u = 0; u < WD_UNROLL; ++u) {
            idn[u] = 0;
            if (ids) idn[u] = ids[min(m0 + 4 * WD_UNROLL + 4 * u + q, w_end - 1)];
        }
#pragma unroll
        for (int u = 0; u < WD_UNROLL; ++u) {
            const int64_t m = m0 + 4 * u + q;
            const bool live = m < w_end;
            const int64_t mc = live ? m : w_end - 1;            // a row past the range is read in place, weight 0
            if (ids) {
                const long long row = (long long)idc[u] - id_off;
                const bool ok = (unsigned long long)row < (unsigned long long)id_rows;
                xs[u].load(Xs + (ok ? row : 0) * KS + VS * r);
                if (!ok) {
#pragma unroll
                    for (int t = 0; t < VS; ++t) xs[u].v[t] = 0.f;
                }
            } else {
                xs[u].load(Xs + mc * KS + VS * r);
            }
            xn[u].load(Xn + mc * KN + VN * r);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float a = G[mc * H + nt * 16 + r], b = G[mc * H + HH + nt * 16 + r];
                if (Mk) {
                    if (!(Mk[mc * H + nt * 16 + r] > 0.f)) a = 0.f;
                    if (!(Mk[mc * H + HH + nt * 16 + r] > 0.f)) b = 0.f;
                }
                gs_[u][nt] = live ? a : 0.f;
                gn_[u][nt] = live ? b : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < WD_UNROLL; ++u) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int t = 0; t < VS; ++t)
                    acc_s[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xs[u].v[t], gs_[u][nt], acc_s[t][nt], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < VN; ++t)
                    acc_n[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xn[u].v[t], gn_[u][nt], acc_n[t][nt], 0, 0, 0);
                bias_s[nt] += gs_[u][nt];
                bias_n[nt] += gn_[u][nt];
            }
        }
#pragma unroll
        for (int u = 0; u < WD_UNROLL; ++u) idc[u] = idn[u];
    }
    // column sums: the four row classes (q) of a column, added in q order
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        bias_s[nt] += __shfl_xor(bias_s[nt], 16, 64);
        bias_s[nt] += __shfl_xor(bias_s[nt], 32, 64);
        bias_n[nt] += __shfl_xor(bias_n[nt], 16, 64);
        bias_n[nt] += __shfl_xor(bias_n[nt], 32, 64);
    }
    // the block's sum in LDS, in slab layout ([k][n], then the bias row), waves added in order
#pragma unroll 1
    for (int w = 0; w < kThreads / 64; ++w) {
        if (wave == w) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int t = 0; t < VS; ++t) {
                        float *dst = red + (VS * (4 * q + e) + t) * HH + nt * 16 + r;
                        *dst = w ? *dst + acc_s[t][nt][e] : acc_s[t][nt][e];
                    }
#pragma unroll
                    for (int t = 0; t < VN; ++t) {
                        float *dst = red + kSlabS + (VN * (4 * q + e) + t) * HH + nt * 16 + r;
                        *dst = w ? *dst + acc_n[t][nt][e] : acc_n[t][nt][e];
                    }
                }
                if (q == 0) {
                    float *ds = red + KS * HH + nt * 16 + r, *dn = red + kSlabS + KN * HH + nt * 16 + r;
                    *ds = w ? *ds + bias_s[nt] : bias_s[nt];
                    *dn = w ? *dn + bias_n[nt] : bias_n[nt];
                }
            }
        }
        __syncthreads();
    }
    float *out_s = slab_s + (int64_t)blockIdx.x * kSlabS, *out_n = slab_n + (int64_t)blockIdx.x * kSlabN;
    for (int i = tid; i < kSlabS; i += kThreads) out_s[i] = red[i];
    for (int i = tid; i < kSlabN; i += kThreads) out_n[i] = red[kSlabS + i];
}

#define GS_WGRAD_DIRECT_CASES(X) \
    X(32, 32, 1) X(32, 64, 1) X(64, 32, 1) X(64, 64, 1) X(64, 128, 1) X(128, 64, 1) X(128, 128, 1) X(32, 128, 1) X(128, 32, 1) \
    X(32, 32, 2) X(32, 64, 2) X(64, 32, 2) X(64, 64, 2) X(64, 128, 2) X(128, 64, 2) X(128, 128, 2) X(32, 128, 2) X(128, 32, 2)

// rows per block for the direct kernel: at most 256 blocks (one per CU: fewer slabs for the
// reduction that follows), at least 128 rows each (so the slab count never exceeds the chunk
// count the workspace was sized for)
inline int64_t wgrad_direct_rows(int64_t M) {
    int64_t rpb = (M + 255) / 256;
    rpb = ((rpb + 15) / 16) * 16;
    return rpb < 128 ? 128 : rpb;
}

inline bool wgrad_direct_covers(int Fs, int Fn, int Hh) {
#define GS_CASE(KS_, KN_, NT_) if (Fs == KS_ && Fn == KN_ && Hh == NT_ * 16) return true;
    GS_WGRAD_DIRECT_CASES(GS_CASE)
#undef GS_CASE
    return false;
}

// n_sides (1 or 2) problem instances of one shape in one launch
inline bool launch_wgrad_direct(int Fs, int Fn, int Hh, const WgradDirectSide &p0, const WgradDirectSide &p1, int n_sides,
                                int64_t M, int64_t rpb, int64_t nblocks, hipStream_t s) {
#define GS_CASE(KS_, KN_, NT_)                                                                              \
    if (Fs == KS_ && Fn == KN_ && Hh == NT_ * 16) {                                                         \
        hipLaunchKernelGGL((wgrad_direct_kernel<KS_, KN_, NT_>), dim3((unsigned)nblocks, (unsigned)n_sides), \
                           dim3(kThreads), 0, s, p0, p1, M, rpb);                                           \
        return true;                                                                                        \
    }
    GS_WGRAD_DIRECT_CASES(GS_CASE)
#undef GS_CASE
    return false;
}

// ---- l2 normalisation ---------------------------------------------------------

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One wave per row: sum of squares by lane-strided loads + xor-shuffle tree.
__global__ __launch_bounds__(kThreads)
void l2norm_fwd_kernel(const float *__restrict__ x, int64_t M, int H, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * (kThreads / 64);
    for (int64_t m = (int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6); m < M; m += stride) {
        float ss = 0.f;
        for (int c = lane; c < H; c += 64) { const float v = x[m * H + c]; ss += v * v; }
        ss = wave_sum(ss);
        const float inv = rsqrtf(fmaxf(ss, 1e-12f));
        for (int c = lane; c < H; c += 64) out[m * H + c] = x[m * H + c] * inv;
    }
}

// y = x * s, s = rsqrt(max(ss, eps)):  g_x = s * g - [ss > eps] * s^3 * <g, x> * x
__global__ __launch_bounds__(kThreads)
void l2norm_bwd_kernel(const float *__restrict__ x, const float *__restrict__ g, int64_t M, int H,
                       float *__restrict__ gx) {
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * (kThreads / 64);
    for (int64_t m = (int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6); m < M; m += stride) {
        float ss = 0.f, dot = 0.f;
        for (int c = lane; c < H; c += 64) {
            const float v = x[m * H + c];
            ss += v * v;
            dot += v * g[m * H + c];
        }
        ss = wave_sum(ss);
        dot = wave_sum(dot);
        const float s = rsqrtf(fmaxf(ss, 1e-12f));
        const float k = ss > 1e-12f ? s * s * s * dot : 0.f;
        for (int c = lane; c < H; c += 64) gx[m * H + c] = s * g[m * H + c] - k * x[m * H + c];
    }
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline int check_project_shape(const char *who, int Fs, int Fn, int64_t M, int H, int act) {
    if (H < 2 || (H & 1)) return gs::fail(GS_EINVAL, "%s: H=%d must be a positive multiple of two", who, H);
    if (Fs < 1 || Fn < 1 || M < 0) return gs::fail(GS_EINVAL, "%s: Fs=%d Fn=%d M=%lld", who, Fs, Fn, (long long)M);
    if (act != GS_ACT_LINEAR && act != GS_ACT_RELU) return gs::fail(GS_EINVAL, "%s: unknown activation %d", who, act);
    return GS_OK;
}

// chunks of the M dimension for the weight-gradient partial sums
inline void wgrad_chunks(int64_t M, int64_t *rows_per_chunk, int64_t *n_chunks) {
    int64_t rpc = 128;                   // small chunks: enough blocks to fill the chip at M ~ 1e5
    // Large M: 512 chunks (rows per chunk a multiple of the kernels' 32-row slices), so that chunks x tiles is a
    // whole number of rounds on 256 compute units with two blocks each -- with power-of-two chunk sizes the
    // last round of a launch ran a third full (2208 blocks on 512 places at M = 1.13e6).
    if ((M + rpc - 1) / rpc > 512) rpc = (((M + 511) / 512 + 31) / 32) * 32;
    *rows_per_chunk = rpc;
    *n_chunks = M > 0 ? (M + rpc - 1) / rpc : 0;
}

}  // namespace

namespace {

int project_fwd(const float *x_self, const int32_t *self_ids, int64_t off_self, int64_t n_rows_self, int Fs,
                const float *x_mean, int Fn, int64_t M, const float *W_self, const float *W_neigh,
                const float *bias, int H, int act, float *out, void *stream) {
    if (int rc = check_project_shape("gs_sage_project_fwd", Fs, Fn, M, H, act)) return rc;
    if (M == 0) return GS_OK;
    if (!x_self || !x_mean || !W_self || !W_neigh || !out)
        return gs::fail(GS_EINVAL, "gs_sage_project_fwd: null pointer argument");
    const int Hh = H / 2;
    GemmSide s0{}, s1{};
    s0.A = x_self; s0.B = W_self; s0.bias = bias; s0.out = out;
    s0.lda = Fs; s0.K = Fs; s0.N = Hh; s0.ldbk = Hh; s0.ldbn = 1; s0.ldo = H;
    s0.vec = Fs % 4 == 0 && aligned16(x_self); s0.ncb = (Hh + BN - 1) / BN;
    s0.row_ids = self_ids; s0.id_off = off_self; s0.n_table = n_rows_self;
    s1.A = x_mean; s1.B = W_neigh; s1.bias = bias ? bias + Hh : nullptr; s1.out = out + Hh;
    s1.lda = Fn; s1.K = Fn; s1.N = Hh; s1.ldbk = Hh; s1.ldbn = 1; s1.ldo = H;
    s1.vec = Fn % 4 == 0 && aligned16(x_mean); s1.ncb = s0.ncb;
    const int64_t bx = (M + BM - 1) / BM;
    const int by = s0.ncb + s1.ncb;
    if (bx > 0x7FFFFFFFLL || by > 65535) return gs::fail(GS_ELIMIT, "gs_sage_project_fwd: grid %lld x %d too large", (long long)bx, by);
    if (big_gemm_applies(s0) && big_gemm_applies(s1) && M >= 4 * GB_M) {
        const hipError_t attr = gs::raise_lds_limit(reinterpret_cast<const void *>(rows_gemm_big_kernel), (int)GB_LDS);
        if (attr != hipSuccess) return gs::fail(GS_ELAUNCH, "gs_sage_project_fwd: LDS attribute: %s", hipGetErrorString(attr));
        const int64_t gx = (M + GB_M - 1) / GB_M;
        const int gy = 2 * (Hh / GB_N);
        hipLaunchKernelGGL(rows_gemm_big_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(kThreads), GB_LDS,
                           (hipStream_t)stream, s0, s1, M, act);
        return gs::check_launch("gs_sage_project_fwd");
    }
    hipLaunchKernelGGL(rows_gemm_kernel, dim3((unsigned)bx, (unsigned)by), dim3(kThreads), 0, (hipStream_t)stream,
                       s0, s1, M, act);
    return gs::check_launch("gs_sage_project_fwd");
}

}  // namespace

extern "C" int gs_sage_project_fwd(const float *x_self, int Fs, const float *x_mean, int Fn, int64_t M,
                                   const float *W_self, const float *W_neigh, const float *bias,
                                   int H, int act, float *out, void *stream) {
    return project_fwd(x_self, nullptr, 0, 0, Fs, x_mean, Fn, M, W_self, W_neigh, bias, H, act, out, stream);
}

extern "C" int gs_sage_project_gather_fwd(const float *feat_self, int64_t n_rows_self, int Fs, int64_t off_self,
                                          const int32_t *self_ids, const float *x_mean, int Fn, int64_t M,
                                          const float *W_self, const float *W_neigh, const float *bias,
                                          int H, int act, float *out, void *stream) {
    if (M > 0 && !self_ids) return gs::fail(GS_EINVAL, "gs_sage_project_gather_fwd: null pointer argument");
    if (n_rows_self < 0) return gs::fail(GS_EINVAL, "gs_sage_project_gather_fwd: n_rows_self=%lld", (long long)n_rows_self);
    return project_fwd(feat_self, self_ids, off_self, n_rows_self, Fs, x_mean, Fn, M, W_self, W_neigh, bias, H, act,
                       out, stream);
}

extern "C" size_t gs_sage_project_bwd_workspace(int Fs, int Fn, int H, int64_t M) {
    if (Fs < 1 || Fn < 1 || H < 2 || M < 0) return 0;
    int64_t rpc, nc;
    wgrad_chunks(M, &rpc, &nc);
    return (size_t)nc * (size_t)(Fs + 1 + Fn + 1) * (size_t)(H / 2) * sizeof(float);
}

extern "C" int gs_sage_project_bwd(const float *x_self, int Fs, const float *x_mean, int Fn, int64_t M,
                                   const float *W_self, const float *W_neigh, int H, int act,
                                   const float *out, const float *g_out,
                                   float *g_W_self, float *g_W_neigh, float *g_bias,
                                   float *g_x_self, float *g_x_mean,
                                   void *workspace, size_t workspace_bytes, void *stream) {
    if (int rc = check_project_shape("gs_sage_project_bwd", Fs, Fn, M, H, act)) return rc;
    if (M > 0 && (!x_self || !x_mean || !g_out || ((g_x_self || g_x_mean) && (!W_self || !W_neigh))))
        return gs::fail(GS_EINVAL, "gs_sage_project_bwd: null pointer argument");
    if (M > 0 && act == GS_ACT_RELU && !out) return gs::fail(GS_EINVAL, "gs_sage_project_bwd: relu needs the forward output");
    const int Hh = H / 2;
    hipStream_t s = (hipStream_t)stream;
    const float *mask = act == GS_ACT_RELU ? out : nullptr;
    const bool want_w = g_W_self || g_W_neigh || g_bias;

    if (want_w) {
        int64_t rpc, nc;
        wgrad_chunks(M, &rpc, &nc);
        const size_t need = gs_sage_project_bwd_workspace(Fs, Fn, H, M);
        if (M > 0 && (!workspace || workspace_bytes < need))
            return gs::fail(GS_EINVAL, "gs_sage_project_bwd: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
        float *slab0 = static_cast<float *>(workspace);
        float *slab1 = slab0 + (size_t)nc * (Fs + 1) * Hh;
        bool direct = false;
        if (M > 0 && wgrad_big_applies(Fs, Fn, Hh, H, x_self, x_mean, g_out, mask)) {
            const hipError_t attr = gs::raise_lds_limit(reinterpret_cast<const void *>(wgrad_big_kernel), (int)WB_LDS);
            if (attr != hipSuccess) return gs::fail(GS_ELAUNCH, "gs_sage_project_bwd: %s", hipGetErrorString(attr));
            WgradSide a{}, b{};
            a.X = x_self; a.G = g_out; a.mask = mask; a.slab = slab0; a.K = Fs; a.N = Hh; a.ldg = H;
            b.X = x_mean; b.G = g_out + Hh; b.mask = mask ? mask + Hh : nullptr; b.slab = slab1; b.K = Fn; b.N = Hh; b.ldg = H;
            const int by = (Fs / 128 + Fn / 128) * (Hh / 128);
            hipLaunchKernelGGL(wgrad_big_kernel, dim3((unsigned)(((nc + 7) / 8) * 8 * by)), dim3(kThreads), WB_LDS, s, a, b, M, rpc);
            if (int rc = gs::check_launch("gs_sage_project_bwd (wgrad)")) return rc;
            direct = true;                         // slabs written in the layout the reduction below expects
        } else if (M > 0 && aligned16(x_self) && aligned16(x_mean) && aligned16(g_out) && (!mask || aligned16(mask))) {
            const int64_t rpb = wgrad_direct_rows(M);
            const int64_t nblocks = (M + rpb - 1) / rpb;          // <= nc: the slabs fit the workspace
            slab1 = slab0 + (size_t)nblocks * (Fs + 1) * Hh;
            const WgradDirectSide side{x_self, x_mean, g_out, mask, slab0, slab1, nullptr, 0, 0};
            direct = launch_wgrad_direct(Fs, Fn, Hh, side, side, 1, M, rpb, nblocks, s);
            if (direct) {
                if (int rc = gs::check_launch("gs_sage_project_bwd (wgrad)")) return rc;
                nc = nblocks;
            } else {
                slab1 = slab0 + (size_t)nc * (Fs + 1) * Hh;
            }
        }
        if (M > 0 && !direct) {
            WgradSide a{}, b{};
            a.X = x_self; a.G = g_out; a.mask = mask; a.slab = slab0; a.K = Fs; a.N = Hh; a.ldg = H;
            a.nkb = (Fs + 63) / 64; a.ncb = (Hh + 63) / 64;
            b.X = x_mean; b.G = g_out + Hh; b.mask = mask ? mask + Hh : nullptr; b.slab = slab1; b.K = Fn; b.N = Hh; b.ldg = H;
            b.nkb = (Fn + 63) / 64; b.ncb = a.ncb;
            const int by = a.nkb * a.ncb + b.nkb * b.ncb;
            if (by > 65535) return gs::fail(GS_ELIMIT, "gs_sage_project_bwd: %d weight tiles exceed the grid limit", by);
            const bool g_ok = H % 4 == 0 && Hh % 4 == 0 && aligned16(g_out) && (!mask || aligned16(mask));
            const int v0 = g_ok && Fs % 4 == 0 && aligned16(x_self), v1 = g_ok && Fn % 4 == 0 && aligned16(x_mean);
            hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)nc, (unsigned)by), dim3(kThreads), 0, s, a, b, M, rpc, v0, v1);
            if (int rc = gs::check_launch("gs_sage_project_bwd (wgrad)")) return rc;
        }
        // bias gradient: row Fs of the self slab holds columns 0..Hh-1, row Fn of the neighbor slab Hh..H-1
        ReduceSides rs{};
        rs.s[0] = ReduceSide{slab0, g_W_self, g_bias, Fs};
        rs.s[1] = ReduceSide{slab1, g_W_neigh, g_bias ? g_bias + Hh : nullptr, Fn};
        const int tot = ((Fs > Fn ? Fs : Fn) + 1) * Hh;
        launch_wgrad_reduce(rs, 2, tot, (int)nc, Hh, s);
        if (int rc = gs::check_launch("gs_sage_project_bwd (reduce)")) return rc;
    }

    if ((g_x_self || g_x_mean) && M > 0) {
        GemmSide s0{}, s1{};
        s0.A = g_out; s0.mask = mask; s0.B = W_self; s0.out = g_x_self;
        s0.lda = H; s0.K = Hh; s0.N = g_x_self ? Fs : 0; s0.ldbk = 1; s0.ldbn = Hh; s0.ldo = Fs;
        s0.vec = H % 4 == 0 && Hh % 4 == 0 && aligned16(g_out) && (!mask || aligned16(mask));
        s0.ncb = (s0.N + BN - 1) / BN;
        s1.A = g_out + Hh; s1.mask = mask ? mask + Hh : nullptr; s1.B = W_neigh; s1.out = g_x_mean;
        s1.lda = H; s1.K = Hh; s1.N = g_x_mean ? Fn : 0; s1.ldbk = 1; s1.ldbn = Hh; s1.ldo = Fn;
        s1.vec = s0.vec; s1.ncb = (s1.N + BN - 1) / BN;
        const int64_t bx = (M + BM - 1) / BM;
        const int by = s0.ncb + s1.ncb;
        if (bx > 0x7FFFFFFFLL || by > 65535) return gs::fail(GS_ELIMIT, "gs_sage_project_bwd: grid %lld x %d too large", (long long)bx, by);
        // wide layers: both input gradients on the 128x128-tile kernel (a side that is not wanted has N = 0)
        const bool big0 = s0.N == 0 || big_gemm_applies(s0), big1 = s1.N == 0 || big_gemm_applies(s1);
        if (big0 && big1 && M >= 4 * GB_M) {
            const hipError_t attr = gs::raise_lds_limit(reinterpret_cast<const void *>(rows_gemm_big_kernel), (int)GB_LDS);
            if (attr != hipSuccess) return gs::fail(GS_ELAUNCH, "gs_sage_project_bwd: %s", hipGetErrorString(attr));
            const int64_t gx = (M + GB_M - 1) / GB_M;
            const int gy = s0.N / GB_N + s1.N / GB_N;
            hipLaunchKernelGGL(rows_gemm_big_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(kThreads), GB_LDS, s, s0, s1, M,
                               GS_ACT_LINEAR);
        } else {
            hipLaunchKernelGGL(rows_gemm_kernel, dim3((unsigned)bx, (unsigned)by), dim3(kThreads), 0, s, s0, s1, M, GS_ACT_LINEAR);
        }
        if (int rc = gs::check_launch("gs_sage_project_bwd (dx)")) return rc;
    }
    return GS_OK;
}

extern "C" int gs_sage_project_wgrad_by_id_supported(int Fs, int Fn, int H) {
    return (H % 2 == 0) && wgrad_direct_covers(Fs, Fn, H / 2) && !(Fs % 128 == 0 && Fn % 128 == 0 && (H / 2) % 128 == 0);
}

// Weight and bias gradients of two projections of one shape (the two node types of a layer) with one
// weight-gradient launch and one reduction launch; falls back to two gs_sage_project_bwd calls when
// the shape has no direct kernel.
extern "C" int gs_sage_project_wgrad_pair(const gs_wgrad_problem *a, const gs_wgrad_problem *b,
                                          int Fs, int Fn, int64_t M, int H, int act, void *stream) {
    if (int rc = check_project_shape("gs_sage_project_wgrad_pair", Fs, Fn, M, H, act)) return rc;
    if (!a || !b) return gs::fail(GS_EINVAL, "gs_sage_project_wgrad_pair: null problem");
    const gs_wgrad_problem *pr[2] = {a, b};
    const size_t need = gs_sage_project_bwd_workspace(Fs, Fn, H, M);
    bool direct = M > 0 && wgrad_direct_covers(Fs, Fn, H / 2) && !wgrad_big_applies(Fs, Fn, H / 2, H, a->x_self, a->x_mean, a->g_out, nullptr);
    for (int i = 0; i < 2; ++i) {
        const gs_wgrad_problem &p = *pr[i];
        if (M > 0 && (!p.x_self || !p.x_mean || !p.g_out || !p.g_W_self || !p.g_W_neigh))
            return gs::fail(GS_EINVAL, "gs_sage_project_wgrad_pair: null pointer in problem %d", i);
        if (M > 0 && act == GS_ACT_RELU && !p.out)
            return gs::fail(GS_EINVAL, "gs_sage_project_wgrad_pair: relu needs the forward output (problem %d)", i);
        if (M > 0 && (!p.workspace || p.workspace_bytes < need))
            return gs::fail(GS_EINVAL, "gs_sage_project_wgrad_pair: workspace of %zu bytes needed, %zu given (problem %d)",
                            need, p.workspace_bytes, i);
        const float *mask = act == GS_ACT_RELU ? p.out : nullptr;
        direct = direct && aligned16(p.x_self) && aligned16(p.x_mean) && aligned16(p.g_out) && (!mask || aligned16(mask));
        if (p.self_ids && p.self_rows < 1)
            return gs::fail(GS_EINVAL, "gs_sage_project_wgrad_pair: problem %d gives self_ids but self_rows=%lld", i, (long long)p.self_rows);
    }
    if (M == 0) return GS_OK;
    if (!direct) {
        if (a->self_ids || b->self_ids)
            return gs::fail(GS_ELIMIT, "gs_sage_project_wgrad_pair: own rows by id need a shape of the direct kernel "
                                       "(gs_sage_project_wgrad_by_id_supported); Fs=%d Fn=%d H=%d is not one", Fs, Fn, H);
        for (int i = 0; i < 2; ++i) {
            const gs_wgrad_problem &p = *pr[i];
            // (no input gradients are asked for, so the weights are not read)
            if (int rc = gs_sage_project_bwd(p.x_self, Fs, p.x_mean, Fn, M, nullptr, nullptr, H, act, p.out, p.g_out,
                                             p.g_W_self, p.g_W_neigh, p.g_bias, nullptr, nullptr,
                                             p.workspace, p.workspace_bytes, stream))
                return rc;
        }
        return GS_OK;
    }
    const int Hh = H / 2;
    hipStream_t s = (hipStream_t)stream;
    const int64_t rpb = wgrad_direct_rows(M);
    const int64_t nblocks = (M + rpb - 1) / rpb;                  // at most the chunk count the workspaces are sized for
    WgradDirectSide sd[2];
    ReduceSides rs{};
    for (int i = 0; i < 2; ++i) {
        const gs_wgrad_problem &p = *pr[i];
        float *slab0 = static_cast<float *>(p.workspace);
        float *slab1 = slab0 + (size_t)nblocks * (Fs + 1) * Hh;
        sd[i] = WgradDirectSide{p.x_self, p.x_mean, p.g_out, act == GS_ACT_RELU ? p.out : nullptr, slab0, slab1,
                                p.self_ids, (long long)p.self_off, (long long)p.self_rows};
        rs.s[2 * i] = ReduceSide{slab0, p.g_W_self, p.g_bias, Fs};
        rs.s[2 * i + 1] = ReduceSide{slab1, p.g_W_neigh, p.g_bias ? p.g_bias + Hh : nullptr, Fn};
    }
    if (!launch_wgrad_direct(Fs, Fn, Hh, sd[0], sd[1], 2, M, rpb, nblocks, s))
        return gs::fail(GS_ELIMIT, "gs_sage_project_wgrad_pair: no kernel for Fs=%d Fn=%d H=%d", Fs, Fn, H);
    if (int rc = gs::check_launch("gs_sage_project_wgrad_pair (wgrad)")) return rc;
    const int tot = ((Fs > Fn ? Fs : Fn) + 1) * Hh;
    launch_wgrad_reduce(rs, 4, tot, (int)nblocks, Hh, s);
    return gs::check_launch("gs_sage_project_wgrad_pair (reduce)");
}

extern "C" int gs_l2_normalize_fwd(const float *x, int64_t M, int H, float *out, void *stream) {
    if (H < 1 || M < 0) return gs::fail(GS_EINVAL, "gs_l2_normalize_fwd: M=%lld H=%d", (long long)M, H);
    if (M == 0) return GS_OK;
    if (!x || !out) return gs::fail(GS_EINVAL, "gs_l2_normalize_fwd: null pointer argument");
    int64_t blocks = (M + 3) / 4;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, x, M, H, out);
    return gs::check_launch("gs_l2_normalize_fwd");
}

extern "C" int gs_l2_normalize_bwd(const float *x, const float *g_out, int64_t M, int H, float *g_x, void *stream) {
    if (H < 1 || M < 0) return gs::fail(GS_EINVAL, "gs_l2_normalize_bwd: M=%lld H=%d", (long long)M, H);
    if (M == 0) return GS_OK;
    if (!x || !g_out || !g_x) return gs::fail(GS_EINVAL, "gs_l2_normalize_bwd: null pointer argument");
    int64_t blocks = (M + 3) / 4;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, x, g_out, M, H, g_x);
    return gs::check_launch("gs_l2_normalize_bwd");
}
