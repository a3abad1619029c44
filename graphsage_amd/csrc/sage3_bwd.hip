// The whole backward pass of the three-hop model (32 units per layer) on the layered forward's saved tables, in
// two kernels plus one reduction -- the pattern of sage2_bwd_all.hip one level further down:
//
//   seeds kernel     per seed s (one wavefront): link head + loss or the given embedding gradients, l2 normalisation
//                    and layer 2 backward (models.py:32, utils.py:103-119), layer 1 backward at the seed, the seed's
//                    row of layer 0; weight gradients of layers 2, 1 and 0 from these rows; and the two rows every
//                    hop-1 node below the seed needs:  a(s) = g_z2[H/2:] . W2_neigh^T / n0  (the gradient at its layer-1
//                    output) and b(s) = g_pre1(s)[H/2:] . W1_neigh^T / n0 (the seed's share of the gradient at its layer-0 output)
//   subtree kernel   per hop-1 node v (one wavefront): g_pre1(v) = a(s) [h1(v) > 0], layer 1 backward at v,
//                    g_pre0(v) = (b(s) + g_pre1(v)[:H/2] . W1_self^T) [h0(v) > 0], and for its n1 children ONE row
//                    c = g_pre1(v)[H/2:] . W1_neigh^T / n1 under n1 relu masks -- so the layer-0 weight gradients over
//                    the hop-2 nodes (most of the rows of a step) read only the saved input rows, never a gradient table
//
// No gradient of an activation passes through memory except a(s), b(s) (two 128-byte rows per seed) and the roots'
// rows between a kernel's two passes.  Tables in the layered engine's row order, per node type t (o = the other type):
//   layer-0 rows  [0, B) seeds of t | [B, B + B n0) hop-1 nodes of t: row B + p n0 + j below seed p of o |
//                 [B (1 + n0), ...) hop-2 nodes of t: row B (1 + n0) + (p n0 + j) n1 + k below seed p of t
//   layer-1 rows  [0, B) seeds | [B, B + B n0) hop-1 nodes
// Sums leave both kernels as slab rows of one layout; the reduction adds them in row order (bitwise reproducible)
// and can apply Adam (sage_bwd_tile.h).
#include "gs_common.h"
#include "sage_bwd_tile.h"

namespace {

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int kH = 32, kHH = 16;             // units per layer, half of it

struct Bwd3Args {
    const float *z[2], *emb[2], *g_emb[2];
    const float *targets, *head_kernel, *head_bias;
    const float *w2_self[2], *w2_neigh[2], *w1_self[2], *w1_neigh[2];
    // by node type t: the seeds' rows, the rows of the hop-1 nodes of type t (below seeds of the other type), the rows of
    // the hop-2 nodes of type t (below seeds of t)
    const float *h1_seed[2], *x_mean2[2], *h0_seed[2], *x_mean1_seed[2], *x_mean0_seed[2];
    const float *h1_hop1[2], *h0_hop1[2], *x_mean1_hop1[2], *x_mean0_hop1[2];
    const float *h0_hop2[2], *x_mean0_hop2[2];
    const uint32_t *mask_hop2[2];           // instead of h0_hop2 when not NULL: bit c of word i says h0_hop2[i][c] > 0
    const int32_t *ids_seed[2], *ids_hop1[2], *ids_hop2[2];
    const float *feat[2];
    long long feat_off[2], feat_rows[2];
    long long n_pairs;
    int n0, n1, normalize, relu;
    float *pred;
    float *slab;                            // seeds kernel's rows first, then the subtree kernel's
    float *a_row[2], *b_row[2];             // [B, 32] per seed type: what the seeds kernel hands the subtrees below a seed
    float *root_seed[2];                    // [B, 32]      layer-0 gradient rows of the seeds, between the two passes
    float *root_sub[2];                     // [B n0, 32]   of the hop-1 nodes below seeds of type t
};

// Slab row, in floats.  What the waves keep in registers: per node type T the layer-0 block [W0_self F x 16 |
// W0_neigh F x 16 | bias 32], the biases of layers 1 and 2 [32] per type, the head [kernel 64 | bias | squared
// error]; then what they keep in LDS: per type [W1_self 32 x 16 | W1_neigh 32 x 16], the same for layer 2.
template <int F> struct Slab3 {
    static constexpr int kL0 = 2 * F * 16 + 32;
    static constexpr int kW = 2 * kH * kHH;                      // one layer's weights of one type
    static constexpr int l0(int T) { return T * kL0; }
    static constexpr int b1(int T) { return 2 * kL0 + T * kH; }
    static constexpr int b2(int T) { return 2 * kL0 + 2 * kH + T * kH; }
    static constexpr int head() { return 2 * kL0 + 4 * kH; }
    static constexpr int regs() { return head() + 2 * kH + 2; }
    static constexpr int w(int layer, int T) { return regs() + ((layer - 1) * 2 + T) * kW; }      // layer 1 or 2
    static constexpr int size() { return regs() + 4 * kW; }
};

template <int F> struct Sums3 {
    static constexpr int V = F / 16;
    using S = Slab3<F>;
    f32x4 w0s[2][V], w0n[2][V];            // lane (r, q): rows k = V * (4 q + e) + t of column r
    float b0s[2], b0n[2], b1[2], b2[2];
    float hk_u, hk_i, hb, hl;

    __device__ __forceinline__ void clear_layer0(int T) {
#pragma unroll
        for (int t = 0; t < V; ++t) { w0s[T][t] = f32x4{0.f, 0.f, 0.f, 0.f}; w0n[T][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        b0s[T] = b0n[T] = 0.f;
    }
    __device__ __forceinline__ void clear_rest() {
        b1[0] = b1[1] = b2[0] = b2[1] = 0.f;
        hk_u = hk_i = hb = hl = 0.f;
    }
    // column sums of the layer-0 gradient: the four row classes (q) of a column, added in q order
    __device__ __forceinline__ void fold_bias0() {
#pragma unroll
        for (int T = 0; T < 2; ++T) {
            b0s[T] += __shfl_xor(b0s[T], 16, 64);
            b0s[T] += __shfl_xor(b0s[T], 32, 64);
            b0n[T] += __shfl_xor(b0n[T], 16, 64);
            b0n[T] += __shfl_xor(b0n[T], 32, 64);
        }
    }
    template <bool STORE> __device__ __forceinline__ void move(float *region, int lane) {
        const int r = lane & 15, q = lane >> 4;
        auto go = [&](float &v, int at) { if (STORE) region[at] = v; else v += region[at]; };
#pragma unroll
        for (int T = 0; T < 2; ++T) {
#pragma unroll
            for (int t = 0; t < V; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = w0s[T][t][e], b = w0n[T][t][e];
                    go(a, S::l0(T) + (V * (4 * q + e) + t) * 16 + r);
                    go(b, S::l0(T) + F * 16 + (V * (4 * q + e) + t) * 16 + r);
                    w0s[T][t][e] = a; w0n[T][t][e] = b;
                }
            if (q == 0) {
                go(b0s[T], S::l0(T) + 2 * F * 16 + r);
                go(b0n[T], S::l0(T) + 2 * F * 16 + 16 + r);
            }
            if (lane < kH) { go(b1[T], S::b1(T) + lane); go(b2[T], S::b2(T) + lane); }
        }
        if (lane < kH) { go(hk_u, S::head() + lane); go(hk_i, S::head() + kH + lane); }
        if (lane == 0) { go(hb, S::head() + 2 * kH); go(hl, S::head() + 2 * kH + 1); }
    }
};

// [n][k] copy of a [k][n] weight matrix (32 x 16) into LDS, by the whole block
__device__ __forceinline__ void stage_t(float *dst, const float *W) {
    for (int e = threadIdx.x; e < kH * kHH; e += kThreads) {
        const int kk = e / kHH, n = e - kk * kHH;
        dst[n * kH + kk] = W[e];
    }
}

// g (this wave's line, 32 floats: [self half | neighbor half]) through one layer's weights of the wave's type:
//   ds[k] = sum_n g[n] Ws[k][n],  dn[k] = sum_n g[16 + n] Wn[k][n]   (every lane its column k = lane & 31),
// and the layer's weight sums of this wave (LDS, [n / 4][lane][n % 4]) take x1 g^T, x1 the lane's input element
// (own row for lanes 0..31, neighbor mean for 32..63).
__device__ __forceinline__ void layer_back(const float *g_s, const float *ws, const float *wn, float *wsum, float x1, int lane,
                                           float &ds, float &dn) {
    const int k = lane & 31, half = lane >> 5;
    ds = dn = 0.f;
    float *w = wsum + 4 * lane;
#pragma unroll
    for (int n4 = 0; n4 < kHH / 4; ++n4) {
        f32x4 acc = *reinterpret_cast<const f32x4 *>(w + n4 * 256);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = 4 * n4 + e;
            const float g0 = g_s[n], g1 = g_s[kHH + n];
            ds = fmaf(g0, ws[n * kH + k], ds);
            dn = fmaf(g1, wn[n * kH + k], dn);
            acc[e] = fmaf(x1, half ? g1 : g0, acc[e]);
        }
        *reinterpret_cast<f32x4 *>(w + n4 * 256) = acc;
    }
}

// The sum over a block's waves and the slab row: register parts through four LDS regions in a fixed order, the LDS
// parts (per wave: layer 1 then layer 2 of the wave's type; even waves type 0, odd waves type 1) added in wave order.
template <int F>
__device__ __forceinline__ void write_slab(Sums3<F> &sums, float *regions, const float *wsum_all, float *out, int wave, int lane) {
    using S = Slab3<F>;
    sums.fold_bias0();
    __syncthreads();
    if (wave >= kWaves / 2) sums.template move<true>(regions + (size_t)(wave - kWaves / 2) * S::regs(), lane);
    __syncthreads();
    if (wave < kWaves / 2) {
        sums.template move<false>(regions + (size_t)wave * S::regs(), lane);
        sums.template move<true>(regions + (size_t)wave * S::regs(), lane);       // (the same places: no other wave reads them before the barrier)
    }
    __syncthreads();
    for (int i = threadIdx.x; i < S::regs(); i += kThreads)
        out[i] = (regions[i] + regions[S::regs() + i]) + (regions[2 * S::regs() + i] + regions[3 * S::regs() + i]);
    for (int i = threadIdx.x; i < 4 * S::kW; i += kThreads) {
        const int lt = i / S::kW, e = i - lt * S::kW;            // lt = (layer - 1) * 2 + T;  e = (half * 32 + k) * 16 + n
        const int layer = lt >> 1, T = lt & 1;
        const int ln = e / kHH, n = e - ln * kHH;
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < kWaves / 2; ++wv)
            v += wsum_all[(size_t)(2 * wv + T) * (2 * S::kW) + layer * S::kW + (n >> 2) * 256 + 4 * ln + (n & 3)];
        out[S::regs() + i] = v;
    }
}

template <int F> constexpr size_t bwd3_lds() {
    // staged weights (8 matrices), the waves' lines, the waves' LDS sums (two layers), four regions
    return (size_t)(8 * kH * kHH + kWaves * 128 + kWaves * 2 * Slab3<F>::kW + (kWaves / 2) * Slab3<F>::regs()) * sizeof(float);
}

// ---- seeds ------------------------------------------------------------------------------------------------
template <int F, bool HEAD, int TT>
__device__ __forceinline__ void seeds_of_type(const Bwd3Args &a, Sums3<F> &sums, const float *wt, float *line, float *wsum,
                                              int lane, long long p_first, long long p_step) {
    constexpr int V = F / 16;
    const int r = lane & 15, q = lane >> 4, k = lane & 31, half = lane >> 5;
    const long long B = a.n_pairs;
    const float *z = a.z[TT], *emb_u = a.emb[0], *emb_i = a.emb[1], *g_emb = a.g_emb[TT];
    const float *h1 = a.h1_seed[TT], *h0 = a.h0_seed[TT], *m2 = a.x_mean2[TT], *m1 = a.x_mean1_seed[TT], *xn0 = a.x_mean0_seed[TT], *feat = a.feat[TT];
    const int32_t *ids = a.ids_seed[TT];
    const long long off = a.feat_off[TT], rows = a.feat_rows[TT];
    float *a_row = a.a_row[TT], *b_row = a.b_row[TT], *root_g = a.root_seed[TT];
    // staged weights: [type][W2s, W2n, W1s, W1n][n][k]
    const float *w2s = wt + (TT * 4 + 0) * kH * kHH, *w2n = w2s + kH * kHH, *w1s = w2n + kH * kHH, *w1n = w1s + kH * kHH;
    float *g_s = line;                                           // this wave's line: a gradient row [32]
    float *wsum1 = wsum, *wsum2 = wsum + Slab3<F>::kW;
    float hk_u = 0.f, hk_i = 0.f, hb = 0.f;
    if (HEAD) {
        if (lane < kH) { hk_u = a.head_kernel[lane]; hk_i = a.head_kernel[kH + lane]; }
        hb = a.head_bias[0];
    }
    sums.clear_rest();
    sums.clear_layer0(1 - TT);                                   // (not touched by this wave: zeros in its slab share)

    struct Seed { float zc, gc, eu, ei, yt, h1v, m2v, h0v, m1v; };
    auto load_seed = [&](Seed &s, long long p) {
        s.zc = z[p * kH + k];
        s.gc = s.eu = s.ei = s.yt = 0.f;
        if (HEAD) { s.eu = emb_u[p * kH + k]; s.ei = emb_i[p * kH + k]; s.yt = a.targets[p]; }
        else s.gc = g_emb[p * kH + k];
        s.h1v = h1[p * kH + k]; s.m2v = m2[p * kH + k];
        s.h0v = h0[p * kH + k]; s.m1v = m1[p * kH + k];
    };
    Seed nxt;
    if (p_first < B) load_seed(nxt, p_first);
    __syncthreads();                                              // weights staged, LDS sums zeroed (every wave passes here once)
    const float inv_b = 1.f / (float)B, inv_n0 = 1.f / (float)a.n0;
    for (long long p = p_first; p < B; p += p_step) {
        const Seed s = nxt;
        load_seed(nxt, p + p_step < B ? p + p_step : p);
        const float zc = lane < kH ? s.zc : 0.f;
        float gc = lane < kH ? s.gc : 0.f;
        if (HEAD) {
            const float eu = lane < kH ? s.eu : 0.f, ei = lane < kH ? s.ei : 0.f;
            const float zh = wave_sum(eu * hk_u + ei * hk_i) + hb;
            const float pr = 1.f / (1.f + __expf(-zh));
            const float d = pr - s.yt;
            const float gzh = 2.f * d * inv_b * pr * (1.f - pr);
            gc = gzh * (TT ? hk_i : hk_u);
            if (TT) sums.hk_i += gzh * ei;
            else {
                sums.hk_u += gzh * eu;
                sums.hb += gzh;
                sums.hl += d * d;
                if (lane == 0) a.pred[p] = pr;
            }
        }
        float gz = gc;
        if (a.normalize) {
            const float ss = wave_sum(zc * zc), dot = wave_sum(zc * gc);
            const float sc = rsqrtf(fmaxf(ss, 1e-12f));
            const float c3 = ss > 1e-12f ? sc * sc * sc * dot : 0.f;
            gz = sc * gc - c3 * zc;
        }
        const float gz2 = (a.relu && !(zc > 0.f)) ? 0.f : gz;     // through layer 2's own activation
        // layer 2 at the seed
        if (lane < kH) g_s[lane] = gz2;
        wave_lds_fence();
        float ds, dn;
        layer_back(g_s, w2s, w2n, wsum2, half ? s.m2v : s.h1v, lane, ds, dn);
        sums.b2[TT] += lane < kH ? gz2 : 0.f;
        if (lane < kH) a_row[p * kH + lane] = dn * inv_n0;        // for the hop-1 nodes below this seed
        const float gp1 = s.h1v > 0.f ? ds : 0.f;                 // through layer 1's relu at the seed
        wave_lds_fence();
        // layer 1 at the seed
        if (lane < kH) g_s[lane] = gp1;
        wave_lds_fence();
        layer_back(g_s, w1s, w1n, wsum1, half ? s.m1v : s.h0v, lane, ds, dn);
        sums.b1[TT] += lane < kH ? gp1 : 0.f;
        if (lane < kH) {
            b_row[p * kH + lane] = dn * inv_n0;
            root_g[p * kH + lane] = s.h0v > 0.f ? ds : 0.f;       // through layer 0's relu: the seed's layer-0 gradient row
        }
        wave_lds_fence();                                         // the line is rewritten by the next seed
    }

    // Second pass: the seeds' own rows (layer 0), four seeds of this wave per step, with the rows put aside above
    // (written by this wave itself: a fence of workgroup scope makes them visible to its own reads).
    sums.clear_layer0(TT);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    for (long long p0 = p_first; p0 < B; p0 += 4 * p_step) {
        const long long pq = p0 + q * p_step;
        const bool live = pq < B;
        const long long p = live ? pq : p0;
        long long own = (long long)ids[p] - off;
        const bool ok = (unsigned long long)own < (unsigned long long)rows;
        if (!ok) own = 0;
        Row<V> xs, xn;
        xs.load(feat + own * F + V * r);
        xn.load(xn0 + p * F + V * r);
        const float gs_ = root_g[p * kH + r], gn_ = root_g[p * kH + 16 + r];
        const float bs = live ? gs_ : 0.f, bn = live ? gn_ : 0.f;
#pragma unroll
        for (int tt = 0; tt < V; ++tt) {
            sums.w0s[TT][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ok ? xs.v[tt] : 0.f, bs, sums.w0s[TT][tt], 0, 0, 0);
            sums.w0n[TT][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xn.v[tt], bn, sums.w0n[TT][tt], 0, 0, 0);
        }
        sums.b0s[TT] += bs;
        sums.b0n[TT] += bn;
    }
}

template <int F, bool HEAD>
__global__ __launch_bounds__(kThreads)
void sage3_bwd_seeds_kernel(const Bwd3Args a) {
    extern __shared__ float smem[];
    using S = Slab3<F>;
    float *wt = smem;                                             // [type][W2s, W2n, W1s, W1n][n][k]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *line = wt + 8 * kH * kHH + wave * 128;
    float *wsum_all = wt + 8 * kH * kHH + kWaves * 128, *wsum = wsum_all + wave * (2 * S::kW);
    float *regions = wsum_all + kWaves * (2 * S::kW);
    Sums3<F> sums;
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        stage_t(wt + (T * 4 + 0) * kH * kHH, a.w2_self[T]);
        stage_t(wt + (T * 4 + 1) * kH * kHH, a.w2_neigh[T]);
        stage_t(wt + (T * 4 + 2) * kH * kHH, a.w1_self[T]);
        stage_t(wt + (T * 4 + 3) * kH * kHH, a.w1_neigh[T]);
    }
    for (int i = threadIdx.x; i < kWaves * 2 * S::kW; i += kThreads) wsum_all[i] = 0.f;
    // even waves of the grid take the user seeds, odd waves the item seeds (the barrier is in seeds_of_type)
    const long long gw = (long long)blockIdx.x * kWaves + wave, p_step = (long long)gridDim.x * (kWaves / 2);
    if (gw & 1) seeds_of_type<F, HEAD, 1>(a, sums, wt, line, wsum, lane, gw >> 1, p_step);
    else seeds_of_type<F, HEAD, 0>(a, sums, wt, line, wsum, lane, gw >> 1, p_step);
    write_slab<F>(sums, regions, wsum_all, a.slab + (size_t)blockIdx.x * S::size(), wave, lane);
}

// ---- subtrees ---------------------------------------------------------------------------------------------
// TT: the type of the seed the subtrees hang below (= the type of their hop-2 children); their roots are of type 1 - TT.
template <int F, int U, bool MASK, int TT>
__device__ __forceinline__ void subtrees_of_type(const Bwd3Args &a, Sums3<F> &sums, const float *wt, float *line, float *wsum,
                                                 int lane, long long i_first, long long i_step) {
    constexpr int V = F / 16, O = 1 - TT;
    const int r = lane & 15, q = lane >> 4, k = lane & 31, half = lane >> 5;
    const long long B = a.n_pairs;
    const int n0 = a.n0, n1 = a.n1, NS = (n1 + 3) >> 2;
    const long long n_sub = B * n0;                              // subtrees of this side
    const float *a_row = a.a_row[TT], *b_row = a.b_row[TT];
    const float *h1 = a.h1_hop1[O], *h0r = a.h0_hop1[O], *m1 = a.x_mean1_hop1[O];
    const float *h0c = a.h0_hop2[TT], *xn_c = a.x_mean0_hop2[TT], *feat_c = a.feat[TT];
    const uint32_t *mk_c = a.mask_hop2[TT];                      // (MASK: instead of h0c)
    const float *xn_r = a.x_mean0_hop1[O], *feat_r = a.feat[O];
    const int32_t *ids_c = a.ids_hop2[TT], *ids_r = a.ids_hop1[O];
    const long long off_c = a.feat_off[TT], rows_c = a.feat_rows[TT], off_r = a.feat_off[O], rows_r = a.feat_rows[O];
    float *root_g = a.root_sub[TT];
    const float *w1s = wt + (O * 2 + 0) * kH * kHH, *w1n = w1s + kH * kHH;       // layer 1 at a root: the root type's weights
    float *g_s = line, *v_s = line + 64;
    sums.clear_rest();
    sums.clear_layer0(TT);

    struct Sub { float av, bv, h1v, h0v, m1v; int32_t idv; uint32_t mwv; };
    auto load_sub = [&](Sub &s, long long i) __attribute__((always_inline)) {
        const long long p = i / n0;
        s.av = a_row[p * kH + k];
        s.bv = b_row[p * kH + k];
        s.h1v = h1[i * kH + k];
        s.h0v = h0r[i * kH + k];
        s.m1v = m1[i * kH + k];
        s.idv = ids_c[i * n1 + min(lane, n1 - 1)];               // lane j: child j (n1 <= 32)
        s.mwv = MASK ? mk_c[i * n1 + min(lane, n1 - 1)] : 0u;
    };
    // The rows of 4 * U children (slots g0 * 4 .. ; lane quarter q takes slot 4 * g + q): own row by id, saved
    // neighbor mean, and the relu masks of columns r and 16 + r (as 1 / 0, or the layer-0 outputs themselves).
    struct Rows { Row<V> xs[U], xn[U]; float ml[U], mh[U]; uint32_t okbits; };
    auto issue = [&](Rows &R, const Sub &s, long long i, int g0) __attribute__((always_inline)) {
        R.okbits = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int g = g0 + u;
            const int j = min(4 * g + q, n1 - 1);                 // (a slot past n1 reads child n1-1 in place, weight 0)
            const unsigned row = (unsigned)(i * n1) + j;          // (rows * F < 2^31: checked by the host)
            long long own = (long long)__shfl(s.idv, j, 64) - off_c;
            const bool ok = (unsigned long long)own < (unsigned long long)rows_c;
            if (!ok) own = 0;
            R.okbits |= (ok ? 1u : 0u) << u;
            R.xs[u].load(feat_c + own * F + V * r);
            R.xn[u].load(xn_c + (row * F + V * r));
            if constexpr (MASK) {                                 // the mask word of the row, from the lane that read it
                const uint32_t mw = (uint32_t)__shfl((int)s.mwv, j, 64);
                R.ml[u] = (mw >> r) & 1u ? 1.f : 0.f;
                R.mh[u] = (mw >> (16 + r)) & 1u ? 1.f : 0.f;
            } else {
                R.ml[u] = h0c[row * kH + r];
                R.mh[u] = h0c[row * kH + 16 + r];
            }
        }
    };
    const float inv_n1 = 1.f / (float)n1;
    // Layer 1 at the root of subtree i; leaves the children's layer-0 gradient row (before their masks) in v_lo / v_hi
    // (columns r and 16 + r) and the root's own layer-0 gradient row in root_g.
    auto root_layer1 = [&](const Sub &s, long long i, float &v_lo, float &v_hi) __attribute__((always_inline)) {
        const float gp1 = s.h1v > 0.f ? s.av : 0.f;               // through layer 1's relu at the root
        if (lane < kH) g_s[lane] = gp1;
        wave_lds_fence();
        float ds, dn;
        layer_back(g_s, w1s, w1n, wsum, half ? s.m1v : s.h0v, lane, ds, dn);
        sums.b1[O] += lane < kH ? gp1 : 0.f;
        if (lane < kH) {
            v_s[lane] = dn * inv_n1;
            root_g[i * kH + lane] = s.h0v > 0.f ? s.bv + ds : 0.f;       // the root's own, through its relu
        }
        wave_lds_fence();
        v_lo = v_s[r]; v_hi = v_s[16 + r];
        wave_lds_fence();                                         // the lines are rewritten by the next subtree
    };
    auto process = [&](const Rows &R, int g0, float v_lo, float v_hi) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int g = g0 + u;
            if (g >= NS) break;
            const bool ok = (R.okbits >> u) & 1u;
            const bool live = 4 * g + q < n1;
            const float bs = (live && R.ml[u] > 0.f) ? v_lo : 0.f;
            const float bn = (live && R.mh[u] > 0.f) ? v_hi : 0.f;
#pragma unroll
            for (int tt = 0; tt < V; ++tt) {
                sums.w0s[TT][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ok ? R.xs[u].v[tt] : 0.f, bs, sums.w0s[TT][tt], 0, 0, 0);
                sums.w0n[TT][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(R.xn[u].v[tt], bn, sums.w0n[TT][tt], 0, 0, 0);
            }
            sums.b0s[TT] += bs;
            sums.b0n[TT] += bn;
        }
    };
    constexpr bool PIPE = U <= 4;                                 // two sets of rows fit the registers
    if constexpr (PIPE) {
        // The first 4 * U children's rows of the NEXT subtree are requested before this subtree's are used, so a
        // wavefront always has a subtree's rows in flight; what a request needs (ids, mask words) is read two ahead.
        const bool any = i_first < n_sub;
        auto at = [&](long long i) { return i < n_sub ? i : (any ? i_first : 0); };      // (past the end: a valid subtree again, unused)
        Sub cur, nxt;
        load_sub(cur, at(i_first));
        load_sub(nxt, at(i_first + i_step));
        __syncthreads();                                          // weights staged, LDS sums zeroed (every wave passes here once)
        Rows R;
        issue(R, cur, at(i_first), 0);
        for (long long i = i_first; i < n_sub; i += i_step) {
            Sub nxt2;
            load_sub(nxt2, at(i + 2 * i_step));
            Rows Rn;
            issue(Rn, nxt, at(i + i_step), 0);
            float v_lo, v_hi;
            root_layer1(cur, i, v_lo, v_hi);
            process(R, 0, v_lo, v_hi);
            for (int g0 = U; g0 < NS; g0 += U) {                  // (more than 4 * U children)
                Rows Rx;
                issue(Rx, cur, i, g0);
                process(Rx, g0, v_lo, v_hi);
            }
            cur = nxt; nxt = nxt2; R = Rn;
        }
    } else {
        Sub nxt;
        if (i_first < n_sub) load_sub(nxt, i_first);
        __syncthreads();                                          // weights staged, LDS sums zeroed (every wave passes here once)
        for (long long i = i_first; i < n_sub; i += i_step) {
            const Sub s = nxt;
            load_sub(nxt, i + i_step < n_sub ? i + i_step : i);
            float v_lo = 0.f, v_hi = 0.f;
            for (int g0 = 0; g0 < NS; g0 += U) {
                Rows R;
                issue(R, s, i, g0);
                if (g0 == 0) root_layer1(s, i, v_lo, v_hi);       // while the first rows arrive
                process(R, g0, v_lo, v_hi);
            }
        }
    }

    // Second pass: the roots' own rows (type O), four subtrees of this wave per step.
    sums.clear_layer0(O);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    for (long long i0 = i_first; i0 < n_sub; i0 += 4 * i_step) {
        const long long iq = i0 + q * i_step;
        const bool live = iq < n_sub;
        const long long i = live ? iq : i0;
        long long own = (long long)ids_r[i] - off_r;
        const bool ok = (unsigned long long)own < (unsigned long long)rows_r;
        if (!ok) own = 0;
        Row<V> xs, xn;
        xs.load(feat_r + own * F + V * r);
        xn.load(xn_r + i * F + V * r);
        const float gs_ = root_g[i * kH + r], gn_ = root_g[i * kH + 16 + r];
        const float bs = live ? gs_ : 0.f, bn = live ? gn_ : 0.f;
#pragma unroll
        for (int tt = 0; tt < V; ++tt) {
            sums.w0s[O][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ok ? xs.v[tt] : 0.f, bs, sums.w0s[O][tt], 0, 0, 0);
            sums.w0n[O][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xn.v[tt], bn, sums.w0n[O][tt], 0, 0, 0);
        }
        sums.b0s[O] += bs;
        sums.b0n[O] += bn;
    }
}

template <int F, int U, bool MASK>
__global__ __launch_bounds__(kThreads)
void sage3_bwd_subtrees_kernel(const Bwd3Args a, int slab_row0) {
    extern __shared__ float smem[];
    using S = Slab3<F>;
    float *wt = smem;                                             // [type][W1s, W1n][n][k] (the first four of eight places)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *line = wt + 8 * kH * kHH + wave * 128;
    float *wsum_all = wt + 8 * kH * kHH + kWaves * 128;
    float *regions = wsum_all + kWaves * (2 * S::kW);
    Sums3<F> sums;
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        stage_t(wt + (T * 2 + 0) * kH * kHH, a.w1_self[T]);
        stage_t(wt + (T * 2 + 1) * kH * kHH, a.w1_neigh[T]);
    }
    for (int i = threadIdx.x; i < kWaves * 2 * S::kW; i += kThreads) wsum_all[i] = 0.f;
    // Even waves take the subtrees below user seeds (roots: items), odd waves those below item seeds.  A wave's LDS
    // sums are filed under the type whose layer-1 weights they belong to: the ROOT's type, the other one than the
    // wave's parity -- write_slab reads wave w's sums as type w & 1, so the waves swap places there (see below).
    const long long gw = (long long)blockIdx.x * kWaves + wave, i_step = (long long)gridDim.x * (kWaves / 2);
    float *wsum = wsum_all + (wave ^ 1) * (2 * S::kW);           // (layer-1 part of the neighbor wave's place)
    if (gw & 1) subtrees_of_type<F, U, MASK, 1>(a, sums, wt, line, wsum, lane, gw >> 1, i_step);
    else subtrees_of_type<F, U, MASK, 0>(a, sums, wt, line, wsum, lane, gw >> 1, i_step);
    write_slab<F>(sums, regions, wsum_all, a.slab + (size_t)(slab_row0 + blockIdx.x) * S::size(), wave, lane);
}

inline int64_t seed_blocks(int64_t n_pairs) {
    int64_t b = (2 * n_pairs + kWaves - 1) / kWaves;
    return b > 256 ? 256 : b;
}
inline int64_t subtree_blocks(int64_t n_pairs, int n0) {
    int64_t b = (2 * n_pairs * n0 + kWaves - 1) / kWaves;
    return b > 256 ? 256 : b;
}
inline int slab3_floats(int F) { return F == 32 ? Slab3<32>::size() : Slab3<64>::size(); }

}  // namespace

extern "C" int gs_sage3_backward_all_supported(int Fu, int Fi, int H1, int H2, int H3, int n0, int n1) {
    return Fu == Fi && (Fu == 32 || Fu == 64) && H1 == kH && H2 == kH && H3 == kH && n0 >= 1 && n1 >= 1 && n1 <= 32 ? 1 : 0;
}

extern "C" size_t gs_sage3_backward_all_workspace(int64_t n_pairs, int n0, int F) {
    if (n_pairs < 1 || n0 < 1 || (F != 32 && F != 64)) return 0;
    const size_t slabs = (size_t)(seed_blocks(n_pairs) + subtree_blocks(n_pairs, n0)) * slab3_floats(F);
    // a, b and the seeds' rows: 3 x [2][B][32]; the hop-1 roots' rows: [2][B n0][32]
    return (slabs + (size_t)6 * n_pairs * kH + (size_t)2 * n_pairs * n0 * kH) * sizeof(float);
}

extern "C" int gs_sage3_backward_all(const gs_sage3_bwd *p, int64_t n_pairs, int n0, int n1, int F, int normalize, int act,
                                     void *workspace, size_t workspace_bytes, const gs_adam_update *adam, void *stream) {
    const char *who = "gs_sage3_backward_all";
    if (!p) return gs::fail(GS_EINVAL, "%s: null argument block", who);
    if (n_pairs < 1 || n0 < 1 || n1 < 1) return gs::fail(GS_EINVAL, "%s: n_pairs=%lld n0=%d n1=%d", who, (long long)n_pairs, n0, n1);
    if (F != 32 && F != 64) return gs::fail(GS_ELIMIT, "%s: F=%d (32 or 64 for both node types)", who, F);
    if (n1 > 32) return gs::fail(GS_ELIMIT, "%s: n1=%d (at most 32)", who, n1);
    if (act != GS_ACT_LINEAR && act != GS_ACT_RELU) return gs::fail(GS_EINVAL, "%s: unknown activation %d", who, act);
    if ((double)n_pairs * (1.0 + n0 + (double)n0 * n1) * F >= 2147483648.0)
        return gs::fail(GS_ELIMIT, "%s: n_pairs=%lld too large (the saved tables are indexed with 32 bits)", who, (long long)n_pairs);
    const bool head = p->targets != nullptr;
    Bwd3Args a{};
    const size_t hop1_rows = (size_t)n_pairs * n0;
    const gs_sage3_bwd_hop1 *h1p = p->hop1;
    const gs_sage3_bwd_hop2 *h2p = p->hop2;
    for (int t = 0; t < 2; ++t) {
        if (!p->z[t] || !p->w2_self[t] || !p->w2_neigh[t] || !p->w1_self[t] || !p->w1_neigh[t] || !p->h0[t] || !p->h1[t] ||
            !p->x_mean0[t] || !p->x_mean1[t] || !p->x_mean2[t] || !p->feat[t] || !p->row_ids[t])
            return gs::fail(GS_EINVAL, "%s: null table or weight pointer (type %d)", who, t);
        if (head ? !p->emb[t] : !p->g_emb[t]) return gs::fail(GS_EINVAL, "%s: %s is null (type %d)", who, head ? "emb" : "g_emb", t);
        if (p->feat_rows[t] < 1) return gs::fail(GS_EINVAL, "%s: feat_rows[%d]=%lld", who, t, (long long)p->feat_rows[t]);
        if (h1p && (!h1p->h0[t] || !h1p->h1[t] || !h1p->x_mean0[t] || !h1p->x_mean1[t] || !h1p->ids[t]))
            return gs::fail(GS_EINVAL, "%s: hop1 has a null table (type %d)", who, t);
        if (h2p && (!h2p->x_mean0[t] || !h2p->ids[t] || (!h2p->h0[t] && !h2p->mask[t])))
            return gs::fail(GS_EINVAL, "%s: hop2 needs x_mean0, ids and h0 or mask (type %d)", who, t);
        if (h2p && !h2p->mask[0] != !h2p->mask[1])
            return gs::fail(GS_EINVAL, "%s: hop2 mask words are given for both node types or for none", who);
        a.z[t] = p->z[t]; a.emb[t] = p->emb[t]; a.g_emb[t] = p->g_emb[t];
        a.w2_self[t] = p->w2_self[t]; a.w2_neigh[t] = p->w2_neigh[t]; a.w1_self[t] = p->w1_self[t]; a.w1_neigh[t] = p->w1_neigh[t];
        a.x_mean2[t] = p->x_mean2[t];
        a.h1_seed[t] = p->h1[t]; a.x_mean1_seed[t] = p->x_mean1[t]; a.h0_seed[t] = p->h0[t]; a.x_mean0_seed[t] = p->x_mean0[t];
        a.ids_seed[t] = p->row_ids[t];
        a.h1_hop1[t] = h1p ? h1p->h1[t] : p->h1[t] + (size_t)n_pairs * kH;
        a.x_mean1_hop1[t] = h1p ? h1p->x_mean1[t] : p->x_mean1[t] + (size_t)n_pairs * kH;
        a.h0_hop1[t] = h1p ? h1p->h0[t] : p->h0[t] + (size_t)n_pairs * kH;
        a.x_mean0_hop1[t] = h1p ? h1p->x_mean0[t] : p->x_mean0[t] + (size_t)n_pairs * F;
        a.ids_hop1[t] = h1p ? h1p->ids[t] : p->row_ids[t] + n_pairs;
        // (without hop1 / hop2 the rows follow the seeds' in the whole tables)
        const size_t above = h1p ? 0 : (size_t)n_pairs + hop1_rows;
        a.x_mean0_hop2[t] = h2p ? h2p->x_mean0[t] : (h1p ? nullptr : p->x_mean0[t] + above * F);
        a.ids_hop2[t] = h2p ? h2p->ids[t] : (h1p ? nullptr : p->row_ids[t] + above);
        a.mask_hop2[t] = h2p ? h2p->mask[t] : nullptr;
        a.h0_hop2[t] = h2p ? h2p->h0[t] : (h1p ? nullptr : p->h0[t] + above * kH);
        if (!a.x_mean0_hop2[t]) return gs::fail(GS_EINVAL, "%s: hop1 is given without hop2", who);
        if ((reinterpret_cast<uintptr_t>(p->feat[t]) | reinterpret_cast<uintptr_t>(a.x_mean0_seed[t]) |
             reinterpret_cast<uintptr_t>(a.x_mean0_hop1[t]) | reinterpret_cast<uintptr_t>(a.x_mean0_hop2[t])) & 15)
            return gs::fail(GS_EINVAL, "%s: feat / x_mean0 must be 16-byte aligned", who);
        a.feat[t] = p->feat[t]; a.feat_off[t] = p->feat_off[t]; a.feat_rows[t] = p->feat_rows[t];
    }
    if (head && (!p->head_kernel || !p->head_bias || !p->pred || !p->g_head_kernel || !p->g_head_bias || !p->loss))
        return gs::fail(GS_EINVAL, "%s: the head needs head_kernel, head_bias, pred, g_head_kernel, g_head_bias and loss", who);
    const size_t need = gs_sage3_backward_all_workspace(n_pairs, n0, F);
    if (!workspace || workspace_bytes < need)
        return gs::fail(GS_EINVAL, "%s: workspace of %zu bytes needed, %zu given", who, need, workspace ? workspace_bytes : (size_t)0);
    a.targets = p->targets; a.head_kernel = p->head_kernel; a.head_bias = p->head_bias; a.pred = p->pred;
    a.n_pairs = n_pairs; a.n0 = n0; a.n1 = n1; a.normalize = normalize != 0; a.relu = act == GS_ACT_RELU;
    const int64_t bt = seed_blocks(n_pairs), bs = subtree_blocks(n_pairs, n0);
    const int row_floats = slab3_floats(F);
    a.slab = static_cast<float *>(workspace);
    float *w = a.slab + (size_t)(bt + bs) * row_floats;
    for (int t = 0; t < 2; ++t) { a.a_row[t] = w; w += n_pairs * kH; }
    for (int t = 0; t < 2; ++t) { a.b_row[t] = w; w += n_pairs * kH; }
    for (int t = 0; t < 2; ++t) { a.root_seed[t] = w; w += n_pairs * kH; }
    for (int t = 0; t < 2; ++t) { a.root_sub[t] = w; w += n_pairs * n0 * kH; }

    // where the stretches of a slab row go
    ReduceArgs ra{};
    const int L0 = 2 * F * 16 + 32, W = 2 * kH * kHH;
    int n = 0;
    bool outside = false;
    auto seg = [&](int begin, int len, float *dst, float scale, bool param = true) {
        const bool upd = adam && dst && param;
        if (upd && (dst < adam->grads || dst + len > adam->grads + adam->n)) outside = true;
        if (n < kSegments) ra.seg[n] = Segment{begin, begin + len, dst, scale, upd ? 1 : 0};
        ++n;
    };
    for (int t = 0; t < 2; ++t) {
        seg(t * L0, F * 16, p->g_w0_self[t], 1.f);
        seg(t * L0 + F * 16, F * 16, p->g_w0_neigh[t], 1.f);
        seg(t * L0 + 2 * F * 16, 32, p->g_b0[t], 1.f);
    }
    for (int t = 0; t < 2; ++t) seg(2 * L0 + t * kH, kH, p->g_b1[t], 1.f);
    for (int t = 0; t < 2; ++t) seg(2 * L0 + 2 * kH + t * kH, kH, p->g_b2[t], 1.f);
    const int hd = 2 * L0 + 4 * kH;
    seg(hd, 2 * kH, head ? p->g_head_kernel : nullptr, 1.f);
    seg(hd + 2 * kH, 1, head ? p->g_head_bias : nullptr, 1.f);
    seg(hd + 2 * kH + 1, 1, head ? p->loss : nullptr, 1.f / (float)n_pairs, false);
    const int regs = hd + 2 * kH + 2;
    for (int t = 0; t < 2; ++t) {
        seg(regs + t * W, kH * kHH, p->g_w1_self[t], 1.f);
        seg(regs + t * W + kH * kHH, kH * kHH, p->g_w1_neigh[t], 1.f);
    }
    for (int t = 0; t < 2; ++t) {
        seg(regs + (2 + t) * W, kH * kHH, p->g_w2_self[t], 1.f);
        seg(regs + (2 + t) * W + kH * kHH, kH * kHH, p->g_w2_neigh[t], 1.f);
    }
    if (n > kSegments) return gs::fail(GS_ELIMIT, "%s: %d segments", who, n);
    if (adam) {
        if (!adam->grads || !adam->params || !adam->m || !adam->v || adam->n < 1)
            return gs::fail(GS_EINVAL, "%s: adam needs grads, params, m, v and n > 0", who);
        if (outside) return gs::fail(GS_EINVAL, "%s: with adam every gradient pointer must lie inside adam->grads[0 .. n)", who);
        ra.adam = AdamArgs{adam->grads, adam->lr_t_device, adam->params, adam->m, adam->v,
                           adam->lr_t, adam->beta_1, adam->beta_2, adam->epsilon, adam->grad_scale};
    }
    ra.n_seg = n;

    hipStream_t s = (hipStream_t)stream;
    // rows of 4 * U children per request: all of a subtree's where that fits two sets of rows in the registers
    const int u_rows = n1 <= 12 ? 3 : n1 <= 16 ? 4 : 8;
    const bool mask = a.mask_hop2[0] != nullptr;
#define GS_B3K(f, u, m)                                                                                             \
    if (F == f && u_rows == u && mask == m) {                                                                       \
        k2 = reinterpret_cast<const void *>(sage3_bwd_subtrees_kernel<f, u, m>);                                    \
        launch2 = [](const Bwd3Args &b, long long nb, size_t lds, hipStream_t st, int row0) {                       \
            hipLaunchKernelGGL((sage3_bwd_subtrees_kernel<f, u, m>), dim3((unsigned)nb), dim3(kThreads), lds, st, b, row0); \
        };                                                                                                          \
    }
#define GS_B3(f)                                                                                                    \
    if (F == f) {                                                                                                   \
        const size_t lds = bwd3_lds<f>();                                                                           \
        const void *k1 = head ? reinterpret_cast<const void *>(sage3_bwd_seeds_kernel<f, true>)                     \
                              : reinterpret_cast<const void *>(sage3_bwd_seeds_kernel<f, false>);                   \
        const void *k2 = nullptr;                                                                                   \
        void (*launch2)(const Bwd3Args &, long long, size_t, hipStream_t, int) = nullptr;                           \
        GS_B3K(f, 3, true) GS_B3K(f, 3, false) GS_B3K(f, 4, true) GS_B3K(f, 4, false) GS_B3K(f, 8, true) GS_B3K(f, 8, false) \
        hipError_t attr = gs::raise_lds_limit(k1, (int)lds);                                                        \
        if (attr == hipSuccess) attr = gs::raise_lds_limit(k2, (int)lds);                                           \
        if (attr != hipSuccess) return gs::fail(GS_ELAUNCH, "%s: %s", who, hipGetErrorString(attr));                \
        if (head) hipLaunchKernelGGL((sage3_bwd_seeds_kernel<f, true>), dim3((unsigned)bt), dim3(kThreads), lds, s, a);   \
        else hipLaunchKernelGGL((sage3_bwd_seeds_kernel<f, false>), dim3((unsigned)bt), dim3(kThreads), lds, s, a); \
        if (int rc = gs::check_launch("gs_sage3_backward_all (seeds)")) return rc;                                  \
        launch2(a, bs, lds, s, (int)bt);                                                                            \
        if (int rc = gs::check_launch("gs_sage3_backward_all (subtrees)")) return rc;                               \
    }
    GS_B3(32) GS_B3(64)
#undef GS_B3
#undef GS_B3K
    return launch_bwd_reduce(ra, a.slab, (int)(bt + bs), row_floats, s, "gs_sage3_backward_all (reduce)");
}
