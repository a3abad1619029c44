// Fused SAGE layer: feature gather + neighbor mean + both projections + bias +
// activation (+ l2 normalisation) in one kernel, for the shapes whose weights
// fit in registers (the reference model's: feature/hidden widths up to 128,
// half-widths 16 or 32).
//
// Work unit: a wave owns a tile of 16 head nodes.
//   1. ids   -- the tile's child ids (16 * n_child, contiguous) and head ids are
//              staged into the wave's LDS region with coalesced loads.
//   2. self  -- the 16 own rows are fetched row-contiguously (FS/4 lanes x 16 B
//              per row) and written to the LDS tile A_self[16][FS+4].
//   3. mean  -- 64/(FN/4) lane groups each take every GN-th head of the tile and
//              sum its n_child rows as they arrive (up to 8 row loads in flight
//              per lane, 16 B per lane, whole rows per group), divide, and write
//              the mean row to A_mean[16][FN+4].  The (heads, n_child, FN)
//              neighbor tensor never exists in memory.
//   4. mfma  -- D[16 x HH] = A[16 x F] . W[F x HH] on v_mfma_f32_16x16x4_f32 for
//              both halves.  The order of k inside the dot product is free, so
//              lane (row r = lane&15, quarter q = lane>>4) feeds the k-steps from
//              its own contiguous quarter of the row, k = q*F/4 + step: A
//              fragments are plain 16-byte LDS reads and the matching W
//              fragments (W[q*F/4 + step][16*ct + r]) are loaded once per wave
//              and stay in registers for all its tiles.
//   5. out   -- bias, relu/linear, optional row l2-normalisation (sum of squares
//              over the 16 lanes that share a row via xor-shuffles), 64-byte
//              contiguous stores per row and half.
// Waves never synchronise with each other: all LDS hand-offs are wave-local.
#include "gs_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kWaves = 4;
constexpr int kMaxSegs = 2 * GS_MAX_HOPS;
constexpr int kUnroll = 8;

struct Seg {
    const int32_t *head_ids;   // node ids of the heads, or NULL: heads are rows self_row0 + p of feat_self
    const int32_t *child_ids;  // node ids of the children [n_heads * n_child], or NULL: rows child_row0 + p*n_child + j
    long long n_heads;
    long long self_row0;
    long long child_row0;
    long long tile0;           // index of the segment's first tile
    long long out_row0;        // first output row of the segment
    int n_child;
};

struct FusedArgs {
    const float *feat_self;  long long n_rows_self;  long long off_self;
    const float *feat_neigh; long long n_rows_neigh; long long off_neigh;
    const float *W_self, *W_neigh, *bias;
    float *out, *save_self, *save_mean;
    Seg seg[kMaxSegs];
    int n_segs;
    long long n_tiles;
    int act, normalize, max_child;
};

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int FS, int FN, int HH>
__global__ __launch_bounds__(kWaves * 64)
void sage_fused_kernel(const FusedArgs a) {
    static_assert(FS % 16 == 0 && FN % 16 == 0 && HH % 16 == 0, "widths must be multiples of 16");
    constexpr int LDA_S = FS + 4, LDA_N = FN + 4;
    constexpr int NT = HH / 16;
    constexpr int KS_S = FS / 4, KS_N = FN / 4;          // k-steps per operand
    constexpr int FSV = FS / 4, FNV = FN / 4;            // lanes per row (float4 each)
    constexpr int RP_S = 64 / FSV;                       // self rows per load instruction
    constexpr int GN = 64 / FNV;                         // lane groups in the mean phase
    constexpr int H = 2 * HH;

    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int per_wave = 16 * LDA_S + 16 * LDA_N + 16 * a.max_child + 16;
    float *As = lds + (size_t)wave * per_wave;
    float *An = As + 16 * LDA_S;
    int32_t *s_child = reinterpret_cast<int32_t *>(An + 16 * LDA_N);
    int32_t *s_head = s_child + 16 * a.max_child;

    const int r = lane & 15, q = lane >> 4;

    // weight fragments, resident for the whole kernel
    float bs[KS_S][NT], bn[KS_N][NT];
#pragma unroll
    for (int ks = 0; ks < KS_S; ++ks)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) bs[ks][ct] = a.W_self[(size_t)(q * KS_S + ks) * HH + ct * 16 + r];
#pragma unroll
    for (int ks = 0; ks < KS_N; ++ks)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) bn[ks][ct] = a.W_neigh[(size_t)(q * KS_N + ks) * HH + ct * 16 + r];
    float bias_v[2][NT];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) bias_v[hf][ct] = a.bias ? a.bias[hf * HH + ct * 16 + r] : 0.f;

    for (long long tile = (long long)blockIdx.x * kWaves + wave; tile < a.n_tiles;
         tile += (long long)gridDim.x * kWaves) {
        int si = 0;
#pragma unroll
        for (int s = 1; s < kMaxSegs; ++s)
            if (s < a.n_segs && tile >= a.seg[s].tile0) si = s;
        const Seg &sg = a.seg[si];
        const int32_t *head_ids = sg.head_ids;
        const int32_t *child_ids = sg.child_ids;
        const long long n_heads = sg.n_heads, self_row0 = sg.self_row0, child_row0 = sg.child_row0;
        const long long tile0 = sg.tile0, out_row0 = sg.out_row0;
        const int n = sg.n_child;
        const long long p0 = (tile - tile0) * 16;
        const int np = (int)min((long long)16, n_heads - p0);

        // 1. ids
        wave_lds_fence();                                   // previous tile's LDS reads are done
        if (child_ids) {
            const int32_t *src = child_ids + p0 * n;
            for (int i = lane; i < np * n; i += 64) s_child[i] = src[i];
        }
        if (lane < 16) {
            long long row = -1;
            if (lane < np) row = head_ids ? (long long)head_ids[p0 + lane] - a.off_self : self_row0 + p0 + lane;
            if ((unsigned long long)row >= (unsigned long long)a.n_rows_self) row = -1;
            s_head[lane] = (int32_t)row;
        }
        wave_lds_fence();

        // 2. own rows
#pragma unroll
        for (int pass = 0; pass < 16 / RP_S; ++pass) {
            const int rr = pass * RP_S + lane / FSV, c = lane % FSV;
            const int32_t row = s_head[rr];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row >= 0) v = reinterpret_cast<const float4 *>(a.feat_self + (size_t)row * FS)[c];
            *reinterpret_cast<float4 *>(As + rr * LDA_S + 4 * c) = v;
            if (a.save_self && rr < np)
                reinterpret_cast<float4 *>(a.save_self + (size_t)(out_row0 + p0 + rr) * FS)[c] = v;
        }

        // 3. neighbor means
        {
            const int g = lane / FNV, c = lane % FNV;
            const float denom = (float)n;
            for (int pp = g; pp < 16; pp += GN) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pp < np) {
                    for (int j0 = 0; j0 < n; j0 += kUnroll) {
                        float4 v[kUnroll];
#pragma unroll
                        for (int u = 0; u < kUnroll; ++u) {
                            const int j = j0 + u;
                            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (j < n) {
                                const long long row = child_ids ? (long long)s_child[pp * n + j] - a.off_neigh
                                                                : child_row0 + (p0 + pp) * n + j;
                                if ((unsigned long long)row < (unsigned long long)a.n_rows_neigh)
                                    v[u] = reinterpret_cast<const float4 *>(a.feat_neigh + (size_t)row * FN)[c];
                            }
                        }
#pragma unroll
                        for (int u = 0; u < kUnroll; ++u) {
                            acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
                        }
                    }
                    acc.x /= denom; acc.y /= denom; acc.z /= denom; acc.w /= denom;
                    if (a.save_mean)
                        reinterpret_cast<float4 *>(a.save_mean + (size_t)(out_row0 + p0 + pp) * FN)[c] = acc;
                }
                *reinterpret_cast<float4 *>(An + pp * LDA_N + 4 * c) = acc;
            }
        }
        wave_lds_fence();

        // 4. projections
        f32x4 acc_s[NT], acc_n[NT];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) { acc_s[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_n[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int k4 = 0; k4 < KS_S / 4; ++k4) {
            const float4 av = *reinterpret_cast<const float4 *>(As + r * LDA_S + q * KS_S + 4 * k4);
            const float ae[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ct = 0; ct < NT; ++ct)
                    acc_s[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[e], bs[4 * k4 + e][ct], acc_s[ct], 0, 0, 0);
        }
#pragma unroll
        for (int k4 = 0; k4 < KS_N / 4; ++k4) {
            const float4 av = *reinterpret_cast<const float4 *>(An + r * LDA_N + q * KS_N + 4 * k4);
            const float ae[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ct = 0; ct < NT; ++ct)
                    acc_n[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[e], bn[4 * k4 + e][ct], acc_n[ct], 0, 0, 0);
        }

        // 5. epilogue: D[row = 4q + i][col = 16 ct + r]
        float o[2][NT][4];
        float ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float vs = acc_s[ct][i] + bias_v[0][ct];
                float vn = acc_n[ct][i] + bias_v[1][ct];
                if (a.act == GS_ACT_RELU) { vs = fmaxf(vs, 0.f); vn = fmaxf(vn, 0.f); }
                o[0][ct][i] = vs; o[1][ct][i] = vn;
                ss[i] += vs * vs + vn * vn;
            }
        if (a.normalize) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float s = ss[i];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) s += __shfl_xor(s, m, 64);
                const float inv = rsqrtf(fmaxf(s, 1e-12f));
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) { o[0][ct][i] *= inv; o[1][ct][i] *= inv; }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * q + i;
            if (row < np) {
                float *dst = a.out + (size_t)(out_row0 + p0 + row) * H;
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) {
                    dst[ct * 16 + r] = o[0][ct][i];
                    dst[HH + ct * 16 + r] = o[1][ct][i];
                }
            }
        }
    }
}

template <int FS, int FN, int HH>
int launch(const FusedArgs &a, hipStream_t stream) {
    const size_t per_wave = (size_t)(16 * (FS + 4) + 16 * (FN + 4) + 16 * a.max_child + 16) * sizeof(float);
    const size_t lds = per_wave * kWaves;
    if (lds > 64 * 1024) return gs::fail(GS_ELIMIT, "gs_sage_fused_fwd: %zu B of LDS per block needed (limit 65536)", lds);
    long long blocks = (a.n_tiles + kWaves - 1) / kWaves;
    const long long cap = 256 * 8;                      // a few blocks per CU; waves loop over tiles
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((sage_fused_kernel<FS, FN, HH>), dim3((unsigned)blocks), dim3(kWaves * 64), lds, stream, a);
    return gs::check_launch("gs_sage_fused_fwd");
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int gs_sage_fused_supported(int Fs, int Fn, int H, int max_child) {
    const int hh = H / 2;
    const bool f_ok = (Fs == 32 || Fs == 64 || Fs == 128) && (Fn == 32 || Fn == 64 || Fn == 128);
    const bool h_ok = (H % 2 == 0) && (hh == 16 || hh == 32);
    if (!f_ok || !h_ok || max_child < 1) return 0;
    if ((Fs + Fn) / 4 * (hh / 16) > 96) return 0;       // weight fragments must stay in registers
    const size_t lds = (size_t)(16 * (Fs + 4) + 16 * (Fn + 4) + 16 * max_child + 16) * 4 * kWaves;
    return lds <= 64 * 1024;
}

extern "C" int gs_sage_fused_fwd(const float *feat_self, int64_t n_rows_self, int Fs, int64_t off_self,
                                 const float *feat_neigh, int64_t n_rows_neigh, int Fn, int64_t off_neigh,
                                 const gs_fused_segment *segs, int n_segs,
                                 const float *W_self, const float *W_neigh, const float *bias,
                                 int H, int act, int normalize,
                                 float *out, float *save_self, float *save_mean, void *stream) {
    if (n_segs < 0 || n_segs > kMaxSegs) return gs::fail(GS_EINVAL, "gs_sage_fused_fwd: n_segs=%d outside [0,%d]", n_segs, kMaxSegs);
    if (act != GS_ACT_LINEAR && act != GS_ACT_RELU) return gs::fail(GS_EINVAL, "gs_sage_fused_fwd: unknown activation %d", act);
    FusedArgs a{};
    long long tiles = 0, rows = 0;
    int max_child = 1;
    for (int s = 0; s < n_segs; ++s) {
        if (!segs) return gs::fail(GS_EINVAL, "gs_sage_fused_fwd: segs is null");
        const gs_fused_segment &g = segs[s];
        if (g.n_heads < 0 || g.n_child < 1) return gs::fail(GS_EINVAL, "gs_sage_fused_fwd: segment %d has n_heads=%lld n_child=%d", s, (long long)g.n_heads, g.n_child);
        if (!g.head_ids && (g.self_row0 < 0 || g.self_row0 + g.n_heads > n_rows_self))
            return gs::fail(GS_EINVAL, "gs_sage_fused_fwd: segment %d dense head rows exceed feat_self", s);
        if (!g.child_ids && (g.child_row0 < 0 || g.child_row0 + g.n_heads * g.n_child > n_rows_neigh))
            return gs::fail(GS_EINVAL, "gs_sage_fused_fwd: segment %d dense child rows exceed feat_neigh", s);
        Seg &d = a.seg[s];
        d.head_ids = g.head_ids; d.child_ids = g.child_ids; d.n_heads = g.n_heads;
        d.self_row0 = g.self_row0; d.child_row0 = g.child_row0; d.n_child = g.n_child;
        d.tile0 = tiles; d.out_row0 = rows;
        tiles += (g.n_heads + 15) / 16;
        rows += g.n_heads;
        if (g.n_child > max_child) max_child = g.n_child;
    }
    if (tiles == 0) return GS_OK;
    if (!feat_self || !feat_neigh || !W_self || !W_neigh || !out)
        return gs::fail(GS_EINVAL, "gs_sage_fused_fwd: null pointer argument");
    if (!aligned16(feat_self) || !aligned16(feat_neigh) || (save_self && !aligned16(save_self)) || (save_mean && !aligned16(save_mean)))
        return gs::fail(GS_EINVAL, "gs_sage_fused_fwd: feature tables and save buffers must be 16-byte aligned");
    if (!gs_sage_fused_supported(Fs, Fn, H, max_child))
        return gs::fail(GS_ELIMIT, "gs_sage_fused_fwd: shape Fs=%d Fn=%d H=%d n_child<=%d is outside the fused kernel's range; "
                                   "use gs_gather_rows + gs_gather_mean_fwd + gs_sage_project_fwd", Fs, Fn, H, max_child);
    a.feat_self = feat_self; a.n_rows_self = n_rows_self; a.off_self = off_self;
    a.feat_neigh = feat_neigh; a.n_rows_neigh = n_rows_neigh; a.off_neigh = off_neigh;
    a.W_self = W_self; a.W_neigh = W_neigh; a.bias = bias;
    a.out = out; a.save_self = save_self; a.save_mean = save_mean;
    a.n_segs = n_segs; a.n_tiles = tiles; a.act = act; a.normalize = normalize; a.max_child = max_child;
    hipStream_t s = (hipStream_t)stream;
    const int hh = H / 2;
#define GS_CASE(fs, fn, h2) if (Fs == fs && Fn == fn && hh == h2) return launch<fs, fn, h2>(a, s);
    GS_CASE(64, 64, 16) GS_CASE(64, 64, 32)
    GS_CASE(32, 32, 16) GS_CASE(32, 32, 32)
    GS_CASE(64, 32, 16) GS_CASE(32, 64, 16) GS_CASE(64, 32, 32) GS_CASE(32, 64, 32)
    GS_CASE(128, 128, 16)
    GS_CASE(128, 64, 16) GS_CASE(64, 128, 16) GS_CASE(128, 32, 16) GS_CASE(32, 128, 16)
    GS_CASE(128, 64, 32) GS_CASE(64, 128, 32) GS_CASE(128, 32, 32) GS_CASE(32, 128, 32)
#undef GS_CASE
    return gs::fail(GS_ELIMIT, "gs_sage_fused_fwd: no instantiation for Fs=%d Fn=%d H=%d", Fs, Fn, H);
}
