// The seed nodes of a three-hop model.
// What is left above the subtree kernel: layers 0, 1 and 2 at the seeds, whose neighbors are the hop-1
// nodes -- their feature rows (layer 0), their layer-0 outputs (layer 1) and their layer-1 outputs (layer 2).
// One launch for both node types; a wavefront takes kSeedsPerWave seeds through all three layers (rows
// 0 .. kSeedsPerWave-1 of the 16-row MFMA tile: few seeds per wave because the launch is short and lives on
// having many waves' row loads in flight, not on the matrix pipe).
#include "gs_common.h"
#include "sage_tile.h"

namespace {

using namespace gs::tile;

constexpr int kSeedsPerWave = 4;
constexpr int kSeedWaves = 8;

struct Roots3Args {
    const float *feat[2]; long long n_rows[2];
    const int32_t *seeds, *hop1; long long n_pairs, n_nodes; int n0;
    const float *h0, *h1;                              // [2*n_pairs*n0, H1] / [.., H2]: rows as hop1
    const float *ws[3][2], *wn[3][2], *b[3][2];        // by layer and seed type
    int act, normalize;
    float *emb;                                        // [2*n_pairs, H3]
    // training (all NULL otherwise): by seed type, one row per seed -- the neighbor means the three layers read, the
    // outputs of layers 0 and 1, layer 2's output before the normalisation
    float *sv_x_mean0[2], *sv_h0[2], *sv_x_mean1[2], *sv_h1[2], *sv_x_mean2[2], *sv_z[2];
};

// Rows 0 .. R-1 (4*V floats each) of an LDS tile to rows s0 .. of a saved table (those below n_pairs).
template <int V>
__device__ __forceinline__ void save_rows(float *table, long long s0, long long n_pairs, const float *src, int ld, int lane) {
    for (int i = lane; i < kSeedsPerWave * V; i += 64) {
        const int g = i / V, c = i % V;
        if (s0 + g < n_pairs)
            *reinterpret_cast<float4 *>(table + (size_t)(s0 + g) * (4 * V) + 4 * c) = *reinterpret_cast<const float4 *>(src + g * ld + 4 * c);
    }
}

// Sums of the n rows (4*V floats each) of R heads, divided by n, into rows 0 .. R-1 of A.  A load instruction
// covers 64 / V rows; with at least R of them a head's rows are split over GPS lane groups, whose partial sums
// are added at the end.  load(g, j, c, ok): 16 bytes at column chunk c of row j of head g (zeros unless ok).
template <int V, int R, typename Load>
__device__ __forceinline__ void head_means(int lane, int n, float *A, int lda, Load load) {
    constexpr int GN = 64 / V;
    constexpr int GPS = GN >= R ? GN / R : 1;
    constexpr int SPP = GN >= R ? R : GN;
    constexpr int kU = 13;
    const int gi = lane / V, c = lane % V, u = gi % GPS;
    const float d = (float)n;
#pragma unroll
    for (int p = 0; p < R / SPP; ++p) {
        const int g = p * SPP + gi / GPS;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int jb = 0; jb < n; jb += kU * GPS) {
            float4 v[kU];
#pragma unroll
            for (int t = 0; t < kU; ++t) {
                const int j = jb + u + t * GPS;
                v[t] = load(g, min(j, n - 1), c, j < n);
            }
#pragma unroll
            for (int t = 0; t < kU; ++t) { s.x += v[t].x; s.y += v[t].y; s.z += v[t].z; s.w += v[t].w; }
        }
#pragma unroll
        for (int m = V; m < V * GPS; m <<= 1) {
            s.x += __shfl_xor(s.x, m, 64); s.y += __shfl_xor(s.y, m, 64);
            s.z += __shfl_xor(s.z, m, 64); s.w += __shfl_xor(s.w, m, 64);
        }
        if (u == 0) *reinterpret_cast<float4 *>(A + g * lda + 4 * c) = make_float4(s.x / d, s.y / d, s.z / d, s.w / d);
    }
}

template <int FT, int FO, int HH1, int HH2, int HH3, int LDA, int LDS_H>
__device__ __forceinline__ void seeds_body(const Roots3Args &a, long long tile, int side, int lane,
                                           float *A, float *Hs, int32_t *ids,
                                           const float *w0s, const float *w0n, const float *w1s, const float *w1n,
                                           const float *w2s, const float *w2n,
                                           const float *b0, const float *b1, const float *b2) {
    constexpr int R = kSeedsPerWave;
    constexpr int NT1 = HH1 / 16, NT2 = HH2 / 16, NT3 = HH3 / 16, H1 = 2 * HH1, H2 = 2 * HH2, H3 = 2 * HH3;
    constexpr int FTV = FT / 4, FOV = FO / 4, H1V = H1 / 4, H2V = H2 / 4;
    constexpr int RT_N = (R * FTV + 63) / 64;
    const int r = lane & 15, q = lane >> 4;
    const int n0 = a.n0;
    RowTable<false> tabT, tabO;
    tabT.init(a.feat[side], a.n_rows[side], FT * 4);
    tabO.init(a.feat[1 - side], a.n_rows[1 - side], FO * 4);
    const int offT = side ? (int)a.n_rows[0] : 0, offO = side ? 0 : (int)a.n_rows[0];
    int wl = lane;
    asm volatile("" : "+v"(wl));
    const long long s0 = tile * R;                     // first seed of the tile, within its type
    const long long row0 = (long long)side * a.n_pairs + s0;       // its row in the seed-ordered arrays
    const bool save = a.sv_z[0] != nullptr;

    wave_lds_fence();                                  // the previous tile's readers are done
    for (int i = lane; i < R * n0; i += 64) {
        const int g = i / n0;
        ids[i] = s0 + g < a.n_pairs ? a.hop1[row0 * n0 + i] : -1;
    }
    float4 own[RT_N];
#pragma unroll
    for (int k = 0; k < RT_N; ++k) {
        const int idx = k * 64 + lane, g = min(idx / FTV, R - 1), c = idx % FTV;
        const bool ok = idx < R * FTV && s0 + g < a.n_pairs;
        const int32_t sd = ok ? a.seeds[row0 + g] : -1;
        own[k] = tabT.load(ok && (unsigned long long)(long long)sd < (unsigned long long)a.n_nodes,
                           (unsigned)(sd - offT), (unsigned)c * 16u, FT * 4);
    }
    wave_lds_fence();

    // ---- layer 0: neighbors = the hop-1 nodes' feature rows ----
    head_means<FOV, R>(lane, n0, A, LDA, [&](int g, int j, int c, bool ok) {
        const int32_t id = ids[g * n0 + j];
        return tabO.load(ok && id >= 0, (unsigned)(id - offO), (unsigned)c * 16u, FO * 4);
    });
    wave_lds_fence();
    if (save) save_rows<FOV>(a.sv_x_mean0[side], s0, a.n_pairs, A, LDA, lane);
    {
        f32x4 acc_s[NT1], acc_n[NT1];
#pragma unroll
        for (int ct = 0; ct < NT1; ++ct) { acc_s[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_n[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        mma_tile<FO, NT1>(A, LDA, w0n, wl, r, q, acc_n);
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < RT_N; ++k) {
            const int idx = k * 64 + lane;
            if (idx < R * FTV) *reinterpret_cast<float4 *>(A + (idx / FTV) * LDA + 4 * (idx % FTV)) = own[k];
        }
        wave_lds_fence();
        mma_tile<FT, NT1>(A, LDA, w0s, wl, r, q, acc_s);
        if (q == 0) {                                  // rows 0 .. 3 of D: registers 0 .. 3 of lanes 0 .. 15
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int ct = 0; ct < NT1; ++ct) {
                    Hs[i * LDS_H + ct * 16 + r] = fmaxf(acc_s[ct][i] + b0[ct * 16 + r], 0.f);
                    Hs[i * LDS_H + HH1 + ct * 16 + r] = fmaxf(acc_n[ct][i] + b0[HH1 + ct * 16 + r], 0.f);
                }
        }
        wave_lds_fence();
        if (save) save_rows<H1V>(a.sv_h0[side], s0, a.n_pairs, Hs, LDS_H, lane);
    }

    // ---- layer 1: neighbors = the hop-1 nodes' layer-0 outputs (consecutive rows) ----
    head_means<H1V, R>(lane, n0, A, LDA, [&](int g, int j, int c, bool ok) {
        const bool want = ok && s0 + g < a.n_pairs;
        const float4 t = *reinterpret_cast<const float4 *>(a.h0 + ((size_t)(row0 + (want ? g : 0)) * n0 + j) * H1 + 4 * c);
        return want ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    });
    wave_lds_fence();
    if (save) save_rows<H1V>(a.sv_x_mean1[side], s0, a.n_pairs, A, LDA, lane);
    {
        f32x4 acc_s[NT2], acc_n[NT2];
#pragma unroll
        for (int ct = 0; ct < NT2; ++ct) { acc_s[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_n[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        mma_tile<H1, NT2>(A, LDA, w1n, wl, r, q, acc_n);
        wave_lds_fence();
        if (lane < R * H1V)
            *reinterpret_cast<float4 *>(A + (lane / H1V) * LDA + 4 * (lane % H1V)) =
                *reinterpret_cast<const float4 *>(Hs + (lane / H1V) * LDS_H + 4 * (lane % H1V));
        wave_lds_fence();
        mma_tile<H1, NT2>(A, LDA, w1s, wl, r, q, acc_s);
        if (q == 0) {
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int ct = 0; ct < NT2; ++ct) {
                    Hs[i * LDS_H + ct * 16 + r] = fmaxf(acc_s[ct][i] + b1[ct * 16 + r], 0.f);
                    Hs[i * LDS_H + HH2 + ct * 16 + r] = fmaxf(acc_n[ct][i] + b1[HH2 + ct * 16 + r], 0.f);
                }
        }
        wave_lds_fence();
        if (save) save_rows<H2V>(a.sv_h1[side], s0, a.n_pairs, Hs, LDS_H, lane);
    }

    // ---- layer 2: neighbors = the hop-1 nodes' layer-1 outputs ----
    head_means<H2V, R>(lane, n0, A, LDA, [&](int g, int j, int c, bool ok) {
        const bool want = ok && s0 + g < a.n_pairs;
        const float4 t = *reinterpret_cast<const float4 *>(a.h1 + ((size_t)(row0 + (want ? g : 0)) * n0 + j) * H2 + 4 * c);
        return want ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    });
    wave_lds_fence();
    if (save) save_rows<H2V>(a.sv_x_mean2[side], s0, a.n_pairs, A, LDA, lane);
    {
        f32x4 acc_s[NT3], acc_n[NT3];
#pragma unroll
        for (int ct = 0; ct < NT3; ++ct) { acc_s[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_n[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        mma_tile<H2, NT3>(A, LDA, w2n, wl, r, q, acc_n);
        wave_lds_fence();
        if (lane < R * H2V)
            *reinterpret_cast<float4 *>(A + (lane / H2V) * LDA + 4 * (lane % H2V)) =
                *reinterpret_cast<const float4 *>(Hs + (lane / H2V) * LDS_H + 4 * (lane % H2V));
        wave_lds_fence();
        mma_tile<H2, NT3>(A, LDA, w2s, wl, r, q, acc_s);
#pragma unroll
        for (int i = 0; i < R; ++i) {
            float o[2][NT3];
            float ss = 0.f;
#pragma unroll
            for (int ct = 0; ct < NT3; ++ct) {
                o[0][ct] = acc_s[ct][i] + b2[ct * 16 + r];
                o[1][ct] = acc_n[ct][i] + b2[HH3 + ct * 16 + r];
                if (a.act == GS_ACT_RELU) { o[0][ct] = fmaxf(o[0][ct], 0.f); o[1][ct] = fmaxf(o[1][ct], 0.f); }
                ss += o[0][ct] * o[0][ct] + o[1][ct] * o[1][ct];
            }
            if (save && q == 0 && s0 + i < a.n_pairs) {
                float *dz = a.sv_z[side] + (size_t)(s0 + i) * H3;
#pragma unroll
                for (int ct = 0; ct < NT3; ++ct) { dz[ct * 16 + r] = o[0][ct]; dz[HH3 + ct * 16 + r] = o[1][ct]; }
            }
            if (a.normalize) {
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) ss += __shfl_xor(ss, m, 64);
                const float inv = rsqrtf(fmaxf(ss, 1e-12f));
#pragma unroll
                for (int ct = 0; ct < NT3; ++ct) { o[0][ct] *= inv; o[1][ct] *= inv; }
            }
            if (q == 0 && s0 + i < a.n_pairs) {
                float *dst = a.emb + (size_t)(row0 + i) * H3;
#pragma unroll
                for (int ct = 0; ct < NT3; ++ct) { dst[ct * 16 + r] = o[0][ct]; dst[HH3 + ct * 16 + r] = o[1][ct]; }
            }
        }
    }
}

template <int FU, int FI, int HH1, int HH2, int HH3>
struct Roots3Shape {
    static constexpr int H1 = 2 * HH1, H2 = 2 * HH2, H3 = 2 * HH3;
    static constexpr int FMAX = FU > FI ? FU : FI, HMAX = H1 > H2 ? H1 : H2;
    static constexpr int LDA = (FMAX > HMAX ? FMAX : HMAX) + 4;
    static constexpr int W_FLOATS = (FU + FI) * HH1 + 2 * H1 * HH2 + 2 * H2 * HH3 + H1 + H2 + H3;
    static size_t lds_bytes(int n0) {
        const int ids = (kSeedsPerWave * n0 + 3) & ~3;
        return (size_t)(W_FLOATS + kSeedWaves * (16 * LDA + kSeedsPerWave * HMAX + ids)) * sizeof(float);
    }
};

template <int FU, int FI, int HH1, int HH2, int HH3>
__global__ __launch_bounds__(kSeedWaves * 64)
void sage3_seeds_kernel(const Roots3Args a) {
    using S = Roots3Shape<FU, FI, HH1, HH2, HH3>;
    constexpr int H1 = S::H1, H2 = S::H2, H3 = S::H3, LDA = S::LDA, HMAX = S::HMAX;
    extern __shared__ float lds[];
    const int side = blockIdx.x & 1;                   // even blocks: the user seeds
    const int FT = side ? FI : FU, FO = side ? FU : FI;
    float *w0s = lds;
    float *w0n = w0s + FT * HH1;
    float *w1s = w0n + FO * HH1;
    float *w1n = w1s + H1 * HH2;
    float *w2s = w1n + H1 * HH2;
    float *w2n = w2s + H2 * HH3;
    float *bias = w2n + H2 * HH3;                      // b0[H1] b1[H2] b2[H3]
    float *wave0 = bias + H1 + H2 + H3;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ids_n = (kSeedsPerWave * a.n0 + 3) & ~3;
    const int per_wave = 16 * LDA + kSeedsPerWave * HMAX + ids_n;
    float *A = wave0 + (size_t)wave * per_wave;
    float *Hs = A + 16 * LDA;
    int32_t *ids = reinterpret_cast<int32_t *>(Hs + kSeedsPerWave * HMAX);
    stage_weight(w0s, a.ws[0][side], FT, HH1);
    stage_weight(w0n, a.wn[0][side], FO, HH1);
    stage_weight(w1s, a.ws[1][side], H1, HH2);
    stage_weight(w1n, a.wn[1][side], H1, HH2);
    stage_weight(w2s, a.ws[2][side], H2, HH3);
    stage_weight(w2n, a.wn[2][side], H2, HH3);
    for (int i = threadIdx.x; i < H1 + H2 + H3; i += blockDim.x) {
        const float *src = i < H1 ? a.b[0][side] : i < H1 + H2 ? a.b[1][side] : a.b[2][side];
        const int k = i < H1 ? i : i < H1 + H2 ? i - H1 : i - H1 - H2;
        bias[i] = src ? src[k] : 0.f;
    }
    __syncthreads();
    const long long tiles = (a.n_pairs + kSeedsPerWave - 1) / kSeedsPerWave;
    const long long my_blocks = (gridDim.x + 1 - side) >> 1;
    for (long long tile = (long long)(blockIdx.x >> 1) * kSeedWaves + wave; tile < tiles; tile += my_blocks * kSeedWaves) {
        if (side == 0)
            seeds_body<FU, FI, HH1, HH2, HH3, LDA, HMAX>(a, tile, 0, lane, A, Hs, ids, w0s, w0n, w1s, w1n, w2s, w2n,
                                                         bias, bias + H1, bias + H1 + H2);
        else
            seeds_body<FI, FU, HH1, HH2, HH3, LDA, HMAX>(a, tile, 1, lane, A, Hs, ids, w0s, w0n, w1s, w1n, w2s, w2n,
                                                         bias, bias + H1, bias + H1 + H2);
    }
}

template <int FU, int FI, int HH1, int HH2, int HH3>
int launch_seeds(const Roots3Args &a, hipStream_t stream) {
    using S = Roots3Shape<FU, FI, HH1, HH2, HH3>;
    const size_t lds = S::lds_bytes(a.n0);
    if (lds > kLdsMax) return gs::fail(GS_ELIMIT, "gs_sage3_seeds_forward: %zu B of LDS per block needed (limit %zu)", lds, kLdsMax);
    auto kern = sage3_seeds_kernel<FU, FI, HH1, HH2, HH3>;
    if (lds > 64 * 1024) {
        hipError_t e = gs::raise_lds_limit((const void *)kern, (int)kLdsMax);
        if (e != hipSuccess) return gs::fail(GS_ELAUNCH, "gs_sage3_seeds_forward: cannot raise the LDS limit: %s", hipGetErrorString(e));
    }
    const long long tiles = (a.n_pairs + kSeedsPerWave - 1) / kSeedsPerWave;
    long long blocks = 2 * ((tiles + kSeedWaves - 1) / kSeedWaves);
    if (blocks > 1024) blocks = 1024;
    gs::launch_timed(kern, dim3((unsigned)blocks), dim3(kSeedWaves * 64), lds, stream, a);
    return gs::check_launch("gs_sage3_seeds_forward");
}

}  // namespace

extern "C" int gs_sage3_seeds_supported(int Fu, int Fi, int H1, int H2, int H3, int n0) {
    if (n0 < 1 || n0 > 4096 || (H3 != 32 && H3 != 64)) return 0;
    bool found = false;
#define GS_X(fu, fi, a1, a2) found = found || (Fu == fu && Fi == fi && H1 == 2 * a1 && H2 == 2 * a2);
    GS_SAGE2_CASES(GS_X)
#undef GS_X
    if (!found) return 0;
    const int fmax = Fu > Fi ? Fu : Fi, hmax = H1 > H2 ? H1 : H2, lda = (fmax > hmax ? fmax : hmax) + 4;
    const size_t lds = (size_t)((Fu + Fi) * (H1 / 2) + H1 * H2 + H2 * H3 + H1 + H2 + H3 +
                                kSeedWaves * (16 * lda + kSeedsPerWave * hmax + ((kSeedsPerWave * n0 + 3) & ~3))) * sizeof(float);
    return lds <= kLdsMax;
}

static int seeds_forward_impl(const float *feat_user, int64_t n_users, int Fu,
                              const float *feat_item, int64_t n_items, int Fi,
                              const int32_t *seeds, int64_t n_pairs, const int32_t *hop1_ids, int n0,
                              const float *h0_hop1, const float *h1_hop1,
                              const gs_sage_weights *l0_user, const gs_sage_weights *l0_item,
                              const gs_sage_weights *l1_user, const gs_sage_weights *l1_item,
                              const gs_sage_weights *l2_user, const gs_sage_weights *l2_item,
                              int H1, int H2, int H3, int act, int normalize, float *emb,
                              const gs_sage3_seeds_saved *saved, void *stream) {
    if (n_pairs < 0) return gs::fail(GS_EINVAL, "gs_sage3_seeds_forward: n_pairs=%lld", (long long)n_pairs);
    if (n_pairs == 0) return GS_OK;
    const gs_sage_weights *ws[6] = {l0_user, l0_item, l1_user, l1_item, l2_user, l2_item};
    if (!feat_user || !feat_item || !seeds || !hop1_ids || !h0_hop1 || !h1_hop1 || !emb)
        return gs::fail(GS_EINVAL, "gs_sage3_seeds_forward: null pointer argument");
    for (int i = 0; i < 6; ++i)
        if (!ws[i] || !ws[i]->w_self || !ws[i]->w_neigh)
            return gs::fail(GS_EINVAL, "gs_sage3_seeds_forward: weight set %d is null or has a null matrix", i);
    if (n_users < 1 || n_items < 1 || n_users + n_items > 0x7FFFFFFELL)
        return gs::fail(GS_EINVAL, "gs_sage3_seeds_forward: n_users=%lld n_items=%lld out of range", (long long)n_users, (long long)n_items);
    if (act != GS_ACT_LINEAR && act != GS_ACT_RELU) return gs::fail(GS_EINVAL, "gs_sage3_seeds_forward: unknown activation %d", act);
    if (!aligned16(feat_user) || !aligned16(feat_item) || !aligned16(h0_hop1) || !aligned16(h1_hop1))
        return gs::fail(GS_EINVAL, "gs_sage3_seeds_forward: feature tables and activation rows must be 16-byte aligned");
    if (!gs_sage3_seeds_supported(Fu, Fi, H1, H2, H3, n0))
        return gs::fail(GS_ELIMIT, "gs_sage3_seeds_forward: shape Fu=%d Fi=%d H1=%d H2=%d H3=%d fan-out %d is outside this kernel's "
                                   "range; use gs_sage_fused_fwd per layer", Fu, Fi, H1, H2, H3, n0);
    Roots3Args a{};
    a.feat[0] = feat_user; a.feat[1] = feat_item; a.n_rows[0] = n_users; a.n_rows[1] = n_items;
    a.seeds = seeds; a.hop1 = hop1_ids; a.n_pairs = n_pairs; a.n_nodes = n_users + n_items; a.n0 = n0;
    a.h0 = h0_hop1; a.h1 = h1_hop1;
    for (int l = 0; l < 3; ++l)
        for (int t = 0; t < 2; ++t) {
            a.ws[l][t] = ws[2 * l + t]->w_self; a.wn[l][t] = ws[2 * l + t]->w_neigh; a.b[l][t] = ws[2 * l + t]->bias;
        }
    a.act = act; a.normalize = normalize; a.emb = emb;
    if (saved) {
        for (int t = 0; t < 2; ++t) {
            float *const tabs[6] = {saved->x_mean0[t], saved->h0[t], saved->x_mean1[t], saved->h1[t], saved->x_mean2[t], saved->z[t]};
            for (float *tab : tabs)
                if (!tab || !aligned16(tab))
                    return gs::fail(GS_EINVAL, "gs_sage3_seeds_forward_saving: a saved table is null or not 16-byte aligned (type %d)", t);
            a.sv_x_mean0[t] = saved->x_mean0[t]; a.sv_h0[t] = saved->h0[t]; a.sv_x_mean1[t] = saved->x_mean1[t];
            a.sv_h1[t] = saved->h1[t]; a.sv_x_mean2[t] = saved->x_mean2[t]; a.sv_z[t] = saved->z[t];
        }
    }
    hipStream_t s = (hipStream_t)stream;
    const int hh1 = H1 / 2, hh2 = H2 / 2;
#define GS_X(fu, fi, a1, a2) if (Fu == fu && Fi == fi && hh1 == a1 && hh2 == a2) \
        return H3 == 32 ? launch_seeds<fu, fi, a1, a2, 16>(a, s) : launch_seeds<fu, fi, a1, a2, 32>(a, s);
    GS_SAGE2_CASES(GS_X)
#undef GS_X
    return gs::fail(GS_ELIMIT, "gs_sage3_seeds_forward: no instantiation for Fu=%d Fi=%d H1=%d H2=%d", Fu, Fi, H1, H2);
}

extern "C" int gs_sage3_seeds_forward(const float *feat_user, int64_t n_users, int Fu,
                                      const float *feat_item, int64_t n_items, int Fi,
                                      const int32_t *seeds, int64_t n_pairs, const int32_t *hop1_ids, int n0,
                                      const float *h0_hop1, const float *h1_hop1,
                                      const gs_sage_weights *l0_user, const gs_sage_weights *l0_item,
                                      const gs_sage_weights *l1_user, const gs_sage_weights *l1_item,
                                      const gs_sage_weights *l2_user, const gs_sage_weights *l2_item,
                                      int H1, int H2, int H3, int act, int normalize, float *emb, void *stream) {
    return seeds_forward_impl(feat_user, n_users, Fu, feat_item, n_items, Fi, seeds, n_pairs, hop1_ids, n0, h0_hop1, h1_hop1,
                              l0_user, l0_item, l1_user, l1_item, l2_user, l2_item, H1, H2, H3, act, normalize, emb, nullptr, stream);
}

extern "C" int gs_sage3_seeds_forward_saving(const float *feat_user, int64_t n_users, int Fu,
                                             const float *feat_item, int64_t n_items, int Fi,
                                             const int32_t *seeds, int64_t n_pairs, const int32_t *hop1_ids, int n0,
                                             const float *h0_hop1, const float *h1_hop1,
                                             const gs_sage_weights *l0_user, const gs_sage_weights *l0_item,
                                             const gs_sage_weights *l1_user, const gs_sage_weights *l1_item,
                                             const gs_sage_weights *l2_user, const gs_sage_weights *l2_item,
                                             int H1, int H2, int H3, int act, int normalize, float *emb,
                                             const gs_sage3_seeds_saved *saved, void *stream) {
    if (!saved) return gs::fail(GS_EINVAL, "gs_sage3_seeds_forward_saving: saved is null");
    return seeds_forward_impl(feat_user, n_users, Fu, feat_item, n_items, Fi, seeds, n_pairs, hop1_ids, n0, h0_hop1, h1_hop1,
                              l0_user, l0_item, l1_user, l1_item, l2_user, l2_item, H1, H2, H3, act, normalize, emb, saved, stream);
}
