// The whole backward pass of the two-hop model on the one-kernel forward's saved tables, in one launch
// plus one reduction: the link head and its loss, the l2 normalisation and layer 1 backward, the
// broadcast of the neighbor-mean gradient to the hop-1 heads, and EVERY parameter gradient (head, layer 1,
// layer 0) as per-block partial sums.
//
//   g_z, g_h1 as in sage2_bwd.hip (models.py:32; utils.py:103-119), but neither is written to memory:
//   dW1_self[t]  += h1[t][p]^T      g_z[t][p][:H2/2]       dW1_neigh[t] += x_mean1[t][p]^T g_z[t][p][H2/2:]
//   dW0_self[t]  += x_self[t][p]^T  g_h1[t][p][:H1/2]      dW0_neigh[t] += x_mean[t][p]^T  g_h1[t][p][H1/2:]      (the root)
//   dW0_self[o]  += x_self[o][m]^T  g_h1[o][m][:H1/2]      dW0_neigh[o] += x_mean[o][m]^T  g_h1[o][m][H1/2:]      (m = B + p*n0 + j)
//
// The gradient rows of a walk's n0 hop-1 heads are ONE vector (the mean's gradient) under n0 relu masks, so
// the layer-0 weight gradients -- the only large item: (1 + n0) rows of 2 F floats per walk -- need nothing
// from memory but those rows and one mask word each.  One wavefront per walk, as in sage2_bwd.hip; the rows
// of a walk are taken four at a time (one 16x16x4 MFMA step: lane (r, q) holds 16 bytes of row q); all of a
// walk's rows (up to 4 U) are requested at once, the walk's head and layer-1 arithmetic runs while they
// arrive, and what that needs was read one walk ahead.  A block's eight waves add their sums in a
// fixed tree through LDS and leave one slab row; sage_bwd_reduce_kernel adds the rows in block order, so
// the result does not depend on scheduling.
//
// Builds for timing experiments (tools/build_variant.sh, tools/time_bwd_all.py; results are wrong on purpose):
// -DGS_BA_NO_ROWS every row read from one cached line, -DGS_BA_NO_WALK without the per-walk arithmetic,
// -DGS_BA_NO_MFMA the products on the vector ALU's first element only, -DGS_BA_NO_TAIL without the sum over
// waves, -DGS_BA_NO_LOOP the prologue alone.
#include "gs_common.h"
#include "sage_bwd_tile.h"

namespace {

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int kH1 = 32;                    // layer-0 width this kernel is written for

struct AllArgs {
    const float *z[2], *emb[2], *g_emb[2], *w_self[2], *w_neigh[2];
    const float *targets, *head_kernel, *head_bias;
    const float *h1[2], *x_mean1[2];
    const float *x_self[2], *x_mean0[2];
    const int32_t *ids[2];                  // never NULL: without ids, any readable words (the values are not used)
    const uint32_t *maskw[2];               // never NULL: the same
    long long id_off[2], id_rows[2];
    long long n_pairs;
    int n0, normalize, relu, by_id, masked;
    float in_scale;
    float *pred, *slab;
    float *root_g[2];                       // [n_pairs, 32] per node type: the roots' layer-0 gradient rows, between the two passes
};

// Slab row of one block, in floats.  First what the waves keep in registers: per node type T the layer-0 block
// [W0_self F x 16 | W0_neigh F x 16 | bias 32], the layer-1 biases [H2] per type, the head [kernel 2 H2 | bias |
// squared error]; then what they keep in LDS: per type the layer-1 weights [W1_self 32 x HH2 | W1_neigh 32 x HH2].
template <int F, int HH2> struct Slab {
    static constexpr int H2 = 2 * HH2;
    static constexpr int kL0 = 2 * F * 16 + 32;
    static constexpr int kW1 = 2 * kH1 * HH2;
    static constexpr int l0(int T) { return T * kL0; }
    static constexpr int b1(int T) { return 2 * kL0 + T * H2; }
    static constexpr int head() { return 2 * kL0 + 2 * H2; }
    static constexpr int regs() { return head() + 2 * H2 + 2; }
    static constexpr int w1(int T) { return regs() + T * kW1; }
    static constexpr int size() { return regs() + 2 * kW1; }
};

// Sums a wave keeps in registers, and their places in the slab row.
template <int F, int HH2> struct Sums {
    static constexpr int V = F / 16, H2 = 2 * HH2;
    using S = Slab<F, HH2>;
    f32x4 w0s[2][V], w0n[2][V];            // lane (r, q): rows k = V * (4 q + e) + t of column r
    float b0s[2], b0n[2];                  // lane (r, q): column r (self half), 16 + r (neighbor half); q classes added at the end
    float b1[2];                           // lane c < H2
    float hk_u, hk_i, hb, hl;              // head: lane c < H2 of each kernel half; bias and squared error (every lane the same)

    __device__ __forceinline__ void clear_layer0(int T) {
#pragma unroll
        for (int t = 0; t < V; ++t) { w0s[T][t] = f32x4{0.f, 0.f, 0.f, 0.f}; w0n[T][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        b0s[T] = b0n[T] = 0.f;
    }
    __device__ __forceinline__ void clear_rest() {
        b1[0] = b1[1] = 0.f;
        hk_u = hk_i = hb = hl = 0.f;
    }

    // STORE: region[place] = sum;  else: sum += region[place]
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
            if (lane < H2) go(b1[T], S::b1(T) + lane);
        }
        if (lane < H2) { go(hk_u, S::head() + lane); go(hk_i, S::head() + H2 + lane); }
        if (lane == 0) { go(hb, S::head() + 2 * H2); go(hl, S::head() + 2 * H2 + 1); }
    }
};

// The walks of one wave: all rooted at nodes of type TT (so that every table and every sum is picked when the
// kernel is compiled), pairs p_first, p_first + p_step, ...
template <int F, int HH2, int U, bool HEAD, int TT>
__device__ __forceinline__ void walks_of_type(const AllArgs &a, Sums<F, HH2> &sums, const float *wt, float *line, float *w1_s,
                                              int lane, long long p_first, long long p_step, float hk, float hk_u, float hk_i,
                                              float hb) {
    constexpr int V = F / 16, H2 = 2 * HH2, wsz = HH2 * kH1, O = 1 - TT;
    float *gz_s = line, *v_s = line + 64;                         // this wave's lines: g_z [64], the heads' gradient row [32]
    const int r = lane & 15, q = lane >> 4, k = lane & 31, half = lane >> 5;
    const long long B = a.n_pairs;
    const int n0 = a.n0, NS = (n0 + 3) >> 2;                      // steps of four hop-1 heads
    const bool by_id = a.by_id != 0, masked = a.masked != 0;
    const float *z = a.z[TT], *emb_u = a.emb[0], *emb_i = a.emb[1], *g_emb = a.g_emb[TT], *h1 = a.h1[TT], *xm1 = a.x_mean1[TT];
    const float *xs_h = a.x_self[O], *xn_h = a.x_mean0[O], *xs_r = a.x_self[TT], *xn_r = a.x_mean0[TT];
    const int32_t *ids_h = a.ids[O], *ids_r = a.ids[TT];
    const uint32_t *mw_h = a.maskw[O];
    const long long off_h = a.id_off[O], rows_h = a.id_rows[O], off_r = a.id_off[TT], rows_r = a.id_rows[TT];
    (void)xn_r;
    const float *ws = wt + (2 * TT) * wsz, *wn = ws + wsz;
    float *w1 = w1_s + TT * (HH2 * 64) + 4 * lane;              // [n / 4][lane][n % 4]

    // What a walk needs besides its rows, read one walk ahead: every lane reads every item (lanes past a
    // row's width repeat a column and drop the value when they use it), so that the count of reads in
    // flight does not depend on the path taken.
    struct Walk { float zc, gc, eu, ei, yt, h1r, xm1; int32_t idv; uint32_t mwv; };
    auto load_walk = [&](Walk &s, long long p) {
        const int c = lane & (H2 - 1);
        s.zc = z[p * H2 + c];
        s.gc = s.eu = s.ei = s.yt = 0.f;
        if (HEAD) { s.eu = emb_u[p * H2 + c]; s.ei = emb_i[p * H2 + c]; s.yt = a.targets[p]; }
        else s.gc = g_emb[p * H2 + c];
        s.h1r = h1[p * kH1 + k];
        s.xm1 = xm1[p * kH1 + k];
        const int j = min(lane, n0 - 1);                          // lane j: hop-1 head j of the walk (n0 <= 64)
        s.idv = ids_h[B + p * n0 + j];
        s.mwv = mw_h[p * n0 + j];
    };

    // First pass: the hop-1 heads' rows (type O) of every walk; the roots' gradient rows are put aside.
    sums.clear_rest();
    sums.clear_layer0(O);
    float *root_g = a.root_g[TT];
    Walk nxt;
    if (p_first < B) load_walk(nxt, p_first);
    __syncthreads();                                              // the weights are staged, the layer-1 sums zeroed (every wave passes here once)
    const float inv_b = 1.f / (float)B, inv_n0 = 1.f / (float)n0;
    for (long long p = p_first; p < B; p += p_step) {            // (wave-uniform)
        const Walk s = nxt;
        load_walk(nxt, p + p_step < B ? p + p_step : p);          // (past the end: this walk again, never used)
        float v_lo = 0.f, v_hi = 0.f;                             // the heads' layer-0 gradient row, columns r and 16 + r

        for (int g0 = 0; g0 < NS; g0 += U) {                      // U steps of four rows: all of a walk's when n0 <= 4 U
            Row<V> xs[U], xn[U];
            uint32_t okbits = 0;                                  // bit u: the own row of step u exists
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int g = g0 + u;
                const int j = min(4 * g + q, n0 - 1);             // (a slot past n0 reads row n0-1 in place, weight 0)
                const unsigned row = (unsigned)(B + p * n0) + j;  // (rows * F < 2^31: checked by the host)
                long long own = row;
                bool ok = true;
                if (by_id) {
                    own = (long long)__shfl(s.idv, j, 64) - off_h;
                    ok = (unsigned long long)own < (unsigned long long)rows_h;
                    if (!ok) own = 0;
                }
                okbits |= (ok ? 1u : 0u) << u;
#ifdef GS_BA_NO_ROWS
                xs[u].load(xn_h + V * r + (own & 1)); xn[u].load(xn_r + V * r + (row & 1));
#else
                xs[u].load(xs_h + own * F + V * r);
                xn[u].load(xn_h + (row * F + V * r));
#endif
            }

#ifdef GS_BA_NO_WALK
            if (g0 == 0 && s.zc == 12345.f) {
#else
            if (g0 == 0) {
#endif
                // ---- the walk's head, layer-1 backward and layer-1 weight gradients, while its rows arrive ----
                const float zc = lane < H2 ? s.zc : 0.f;
                float gc = lane < H2 ? s.gc : 0.f;
                if (HEAD) {
                    const float eu = lane < H2 ? s.eu : 0.f, ei = lane < H2 ? s.ei : 0.f;
                    const float zh = wave_sum(eu * hk_u + ei * hk_i) + hb;
                    const float pr = 1.f / (1.f + __expf(-zh));
                    const float d = pr - s.yt;
                    const float gzh = 2.f * d * inv_b * pr * (1.f - pr);
                    gc = gzh * hk;
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
                const float gzm = (a.relu && !(zc > 0.f)) ? 0.f : gz;      // through the layer's own activation
                gz_s[lane] = gzm;
                wave_lds_fence();
                const float x1 = half ? s.xm1 : s.h1r;            // this lane's layer-1 input element: row k of its half's weight
                float ds = 0.f, dn = 0.f;
                // (this wave's layer-1 weight sums: read, added to and written back 16 bytes per lane; nobody else touches them)
#pragma unroll
                for (int n4 = 0; n4 < HH2 / 4; ++n4) {
                    f32x4 acc = *reinterpret_cast<const f32x4 *>(w1 + n4 * 256);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int n = 4 * n4 + e;
                        const float g0v = gz_s[n], g1v = gz_s[HH2 + n];
                        ds = fmaf(g0v, ws[n * kH1 + k], ds);
                        dn = fmaf(g1v, wn[n * kH1 + k], dn);
                        acc[e] = fmaf(x1, half ? g1v : g0v, acc[e]);
                    }
                    *reinterpret_cast<f32x4 *>(w1 + n4 * 256) = acc;
                }
                sums.b1[TT] += gzm;                               // (lanes >= H2 hold zeros)
                ds *= a.in_scale;
                dn *= a.in_scale;
                if (masked && !(s.h1r > 0.f)) ds = 0.f;           // through layer 0's activation (and its dropout)
                if (lane < kH1) { v_s[lane] = dn * inv_n0; root_g[p * kH1 + lane] = ds; }
                wave_lds_fence();
                v_lo = v_s[r]; v_hi = v_s[16 + r];
                wave_lds_fence();                                 // the lines are rewritten by the next walk
            }

            // ---- four rows per step into the layer-0 sums of the heads' node type ----
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int g = g0 + u;
                if (g >= NS) break;                               // (uniform; a step past the last head only read a row in place)
                const bool ok = (okbits >> u) & 1u;
                const int j = min(4 * g + q, n0 - 1);
                const uint32_t mw = masked ? (uint32_t)__shfl((int)s.mwv, j, 64) : 0xFFFFFFFFu;
                const bool live = 4 * g + q < n0;
                const float bs = (live && ((mw >> r) & 1u)) ? v_lo : 0.f;
                const float bn = (live && ((mw >> (16 + r)) & 1u)) ? v_hi : 0.f;
#ifdef GS_BA_NO_MFMA
#pragma unroll
                for (int tt = 0; tt < V; ++tt) { sums.w0s[O][tt][0] += (ok ? xs[u].v[tt] : 0.f) * bs; sums.w0n[O][tt][0] += xn[u].v[tt] * bn; }
#else
#pragma unroll
                for (int tt = 0; tt < V; ++tt) {
                    sums.w0s[O][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ok ? xs[u].v[tt] : 0.f, bs, sums.w0s[O][tt], 0, 0, 0);
                    sums.w0n[O][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xn[u].v[tt], bn, sums.w0n[O][tt], 0, 0, 0);
                }
#endif
                sums.b0s[O] += bs;
                sums.b0n[O] += bn;
            }
        }
    }

    // Second pass: the roots' own rows (type TT), four walks of this wave per step, with the gradient rows the
    // first pass put aside.  They were written by this wave itself: a fence of workgroup scope (the wave's stores
    // have completed; no cache is written back or dropped) makes them visible to its own reads.
    sums.clear_layer0(TT);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    for (long long p0 = p_first; p0 < B; p0 += 4 * p_step) {
        const long long pq = p0 + q * p_step;
        const bool live = pq < B;
        const long long p = live ? pq : p0;
        long long own = p;
        bool ok = true;
        if (by_id) {
            own = (long long)ids_r[p] - off_r;
            ok = (unsigned long long)own < (unsigned long long)rows_r;
            if (!ok) own = 0;
        }
        Row<V> xs, xn;
        xs.load(xs_r + own * F + V * r);
        xn.load(xn_r + p * F + V * r);
        const float gs_ = root_g[p * kH1 + r], gn_ = root_g[p * kH1 + 16 + r];
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

template <int F, int HH2, int U, bool HEAD>
__global__ __launch_bounds__(kThreads)
void sage2_bwd_all_kernel(const AllArgs a) {
    extern __shared__ float smem[];
    constexpr int H2 = 2 * HH2, wsz = HH2 * kH1;
    using S = Slab<F, HH2>;
    float *wt = smem;                                             // [type][self, neigh][n][k]
    // (the wave's number as a scalar: everything derived from it -- walks, rows' bases -- is then scalar too)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *line = smem + 4 * wsz + wave * 128;
    // layer-1 weight sums of this wave: [type][n / 4][lane (half, k)][n % 4], added to in place
    float *w1_all = smem + 4 * wsz + kWaves * 128, *w1_s = w1_all + wave * (2 * HH2 * 64);
    float *regions = w1_all + kWaves * (2 * HH2 * 64);            // four slab rows' register parts for the sum over waves

    Sums<F, HH2> sums;
    float hk_u = 0.f, hk_i = 0.f, hb = 0.f;                       // head: kernel (this lane's column of both halves), bias
    if (HEAD) {
        if (lane < H2) { hk_u = a.head_kernel[lane]; hk_i = a.head_kernel[H2 + lane]; }
        hb = a.head_bias[0];
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float *W = (m & 1) ? a.w_neigh[m >> 1] : a.w_self[m >> 1];
        for (int e = threadIdx.x; e < wsz; e += kThreads) {       // e = k * HH2 + n: the weights are read in memory order
            const int kk = e / HH2, n = e - kk * HH2;
            wt[m * wsz + n * kH1 + kk] = W[e];
        }
    }
    for (int i = threadIdx.x; i < kWaves * 2 * HH2 * 64; i += kThreads) w1_all[i] = 0.f;
    // (the barrier that publishes the staged weights is in walks_of_type, behind the first walk's reads)
#ifdef GS_BA_NO_LOOP
    __syncthreads();
    if (a.n0 > 0) { a.slab[(size_t)blockIdx.x * S::size() + lane] = wt[lane]; return; }
#endif

    // even waves of the grid take the user-rooted walks, odd waves the item-rooted ones
    const long long gw = (long long)blockIdx.x * kWaves + wave, p_step = (long long)gridDim.x * (kWaves / 2);
    if (gw & 1) walks_of_type<F, HH2, U, HEAD, 1>(a, sums, wt, line, w1_s, lane, gw >> 1, p_step, hk_i, hk_u, hk_i, hb);
    else walks_of_type<F, HH2, U, HEAD, 0>(a, sums, wt, line, w1_s, lane, gw >> 1, p_step, hk_u, hk_u, hk_i, hb);

    // column sums of the layer-0 gradient: the four row classes (q) of a column, added in q order
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        sums.b0s[T] += __shfl_xor(sums.b0s[T], 16, 64);
        sums.b0s[T] += __shfl_xor(sums.b0s[T], 32, 64);
        sums.b0n[T] += __shfl_xor(sums.b0n[T], 16, 64);
        sums.b0n[T] += __shfl_xor(sums.b0n[T], 32, 64);
    }
    // the block's sum, in a fixed order: waves 4..7 into 0..3 through LDS, those four added as the row is written
#ifdef GS_BA_NO_TAIL
    if (wave == 0) { sums.template move<true>(regions, lane); a.slab[(size_t)blockIdx.x * S::size() + lane] = regions[lane]; }
    return;
#endif
    __syncthreads();
    if (wave >= kWaves / 2) sums.template move<true>(regions + (size_t)(wave - kWaves / 2) * S::regs(), lane);
    __syncthreads();
    if (wave < kWaves / 2) {
        sums.template move<false>(regions + (size_t)wave * S::regs(), lane);
        sums.template move<true>(regions + (size_t)wave * S::regs(), lane);       // (the same places: no other wave reads them before the barrier)
    }
    __syncthreads();
    float *out = a.slab + (size_t)blockIdx.x * S::size();
    for (int i = threadIdx.x; i < S::regs(); i += kThreads)
        out[i] = (regions[i] + regions[S::regs() + i]) + (regions[2 * S::regs() + i] + regions[3 * S::regs() + i]);
    // the layer-1 weight sums: the eight waves' copies added in wave order
    for (int i = threadIdx.x; i < 2 * S::kW1; i += kThreads) {
        const int T = i / S::kW1, e = i - T * S::kW1;             // e = (half * 32 + k) * HH2 + n
        const int ln = e / HH2, n = e - ln * HH2;
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < kWaves; ++wv) v += w1_all[wv * (2 * HH2 * 64) + T * (HH2 * 64) + (n >> 2) * 256 + 4 * ln + (n & 3)];
        out[S::regs() + i] = v;
    }
}

inline int64_t all_blocks(int64_t n_pairs) {
    int64_t blocks = (2 * n_pairs + kWaves - 1) / kWaves;       // (half of a block's waves per root type)
    return blocks > 256 ? 256 : blocks;
}

template <int F, int HH2> constexpr size_t all_lds() {
    return (size_t)(4 * HH2 * kH1 + kWaves * 128 + kWaves * 2 * HH2 * 64 + (kWaves / 2) * Slab<F, HH2>::regs()) * sizeof(float);
}

inline bool all_covers(int Fu, int Fi, int H1, int H2) {
    return Fu == Fi && (Fu == 32 || Fu == 64) && H1 == kH1 && H2 == 32;     // (wider: the sums no longer fit a wave's registers)
}

inline int slab_floats(int F, int H2) { return F == 32 ? Slab<32, 16>::size() : Slab<64, 16>::size(); }

}  // namespace

extern "C" int gs_sage2_backward_all_supported(int Fu, int Fi, int H1, int H2) { return all_covers(Fu, Fi, H1, H2) ? 1 : 0; }

extern "C" size_t gs_sage2_backward_all_workspace(int64_t n_pairs, int F, int H1, int H2) {
    if (n_pairs < 1 || !all_covers(F, F, H1, H2)) return 0;
    // the blocks' slab rows, then the roots' gradient rows of both node types
    return ((size_t)all_blocks(n_pairs) * (size_t)slab_floats(F, H2) + (size_t)2 * n_pairs * kH1) * sizeof(float);
}

extern "C" int gs_sage2_backward_all(const gs_sage2_bwd_all *p, int64_t n_pairs, int n0, int F, int H1, int H2,
                                     int normalize, int act, void *workspace, size_t workspace_bytes,
                                     const gs_adam_update *adam, void *stream) {
    const char *who = "gs_sage2_backward_all";
    if (!p) return gs::fail(GS_EINVAL, "%s: null argument block", who);
    if (n_pairs < 1 || n0 < 1) return gs::fail(GS_EINVAL, "%s: n_pairs=%lld n0=%d", who, (long long)n_pairs, n0);
    if (n0 > 64) return gs::fail(GS_ELIMIT, "%s: n0=%d (at most 64: a lane per hop-1 head)", who, n0);
    if (!all_covers(F, F, H1, H2))
        return gs::fail(GS_ELIMIT, "%s: F=%d H1=%d H2=%d (F 32 or 64 for both node types, H1 = H2 = 32: "
                                   "gs_sage2_backward_all_supported)", who, F, H1, H2);
    if (act != GS_ACT_LINEAR && act != GS_ACT_RELU) return gs::fail(GS_EINVAL, "%s: unknown activation %d", who, act);
    if (n_pairs * (1 + n0) * F >= ((int64_t)1 << 31))
        return gs::fail(GS_ELIMIT, "%s: n_pairs=%lld too large (the saved tables are indexed with 32 bits)", who, (long long)n_pairs);
    const bool head = p->targets != nullptr;
    AllArgs a{};
    for (int t = 0; t < 2; ++t) {
        if (!p->z[t] || !p->w1_self[t] || !p->w1_neigh[t] || !p->h1[t] || !p->x_mean1[t] || !p->x_self[t] || !p->x_mean[t])
            return gs::fail(GS_EINVAL, "%s: null table or weight pointer (type %d)", who, t);
        if (head ? !p->emb[t] : !p->g_emb[t])
            return gs::fail(GS_EINVAL, "%s: %s is null (type %d)", who, head ? "emb" : "g_emb", t);
        if ((p->self_ids[t] != nullptr) != (p->self_ids[0] != nullptr) || (p->h1_mask[t] != nullptr) != (p->h1_mask[0] != nullptr))
            return gs::fail(GS_EINVAL, "%s: self_ids / h1_mask are given for both node types or for none", who);
        if (p->self_ids[t] && p->self_rows[t] < 1)
            return gs::fail(GS_EINVAL, "%s: self_ids given but self_rows[%d]=%lld", who, t, (long long)p->self_rows[t]);
        if ((reinterpret_cast<uintptr_t>(p->x_self[t]) | reinterpret_cast<uintptr_t>(p->x_mean[t])) & 15)
            return gs::fail(GS_EINVAL, "%s: x_self / x_mean must be 16-byte aligned", who);
        a.z[t] = p->z[t]; a.emb[t] = p->emb[t]; a.g_emb[t] = p->g_emb[t];
        a.w_self[t] = p->w1_self[t]; a.w_neigh[t] = p->w1_neigh[t];
        a.h1[t] = p->h1[t]; a.x_mean1[t] = p->x_mean1[t];
        a.x_self[t] = p->x_self[t]; a.x_mean0[t] = p->x_mean[t];
        // (without ids / mask words the kernel still reads a word per row, from a table that has one)
        a.ids[t] = p->self_ids[t] ? p->self_ids[t] : reinterpret_cast<const int32_t *>(p->x_mean[t]);
        a.maskw[t] = p->h1_mask[t] ? p->h1_mask[t] : reinterpret_cast<const uint32_t *>(p->x_mean[t]);
        a.id_off[t] = p->self_off[t]; a.id_rows[t] = p->self_rows[t];
    }
    if (head && (!p->head_kernel || !p->head_bias || !p->pred || !p->g_head_kernel || !p->g_head_bias || !p->loss))
        return gs::fail(GS_EINVAL, "%s: the head needs head_kernel, head_bias, pred, g_head_kernel, g_head_bias and loss", who);
    const size_t need = gs_sage2_backward_all_workspace(n_pairs, F, H1, H2);
    if (!workspace || workspace_bytes < need)
        return gs::fail(GS_EINVAL, "%s: workspace of %zu bytes needed, %zu given", who, need, workspace ? workspace_bytes : (size_t)0);
    a.targets = p->targets; a.head_kernel = p->head_kernel; a.head_bias = p->head_bias; a.pred = p->pred;
    a.n_pairs = n_pairs; a.n0 = n0; a.normalize = normalize != 0; a.relu = act == GS_ACT_RELU;
    a.in_scale = p->input_scale != 0.f ? p->input_scale : 1.f;
    a.by_id = p->self_ids[0] != nullptr; a.masked = p->h1_mask[0] != nullptr;
    a.slab = static_cast<float *>(workspace);
    a.root_g[0] = a.slab + (size_t)all_blocks(n_pairs) * slab_floats(F, H2);
    a.root_g[1] = a.root_g[0] + (size_t)n_pairs * kH1;
    const int64_t blocks = all_blocks(n_pairs);
    // where the stretches of a slab row go
    ReduceArgs ra{};
    const int HH2 = H2 / 2, L0 = 2 * F * 16 + 32, W1 = 2 * kH1 * HH2;
    int n = 0;
    bool outside = false;                    // a gradient that does not lie in the optimizer's flat buffer
    auto seg = [&](int begin, int len, float *dst, float scale, bool param = true) {
        const bool upd = adam && dst && param;
        if (upd && (dst < adam->grads || dst + len > adam->grads + adam->n)) outside = true;
        ra.seg[n++] = Segment{begin, begin + len, dst, scale, upd ? 1 : 0};
    };
    for (int t = 0; t < 2; ++t) {
        seg(t * L0, F * 16, p->g_w0_self[t], 1.f);
        seg(t * L0 + F * 16, F * 16, p->g_w0_neigh[t], 1.f);
        seg(t * L0 + 2 * F * 16, 32, p->g_b0[t], 1.f);
    }
    for (int t = 0; t < 2; ++t) seg(2 * L0 + t * H2, H2, p->g_b1[t], 1.f);
    const int hd = 2 * L0 + 2 * H2;
    seg(hd, 2 * H2, head ? p->g_head_kernel : nullptr, 1.f);
    seg(hd + 2 * H2, 1, head ? p->g_head_bias : nullptr, 1.f);
    seg(hd + 2 * H2 + 1, 1, head ? p->loss : nullptr, 1.f / (float)n_pairs, false);
    for (int t = 0; t < 2; ++t) {
        const int at = hd + 2 * H2 + 2 + t * W1;
        seg(at, kH1 * HH2, p->g_w1_self[t], 1.f);
        seg(at + kH1 * HH2, kH1 * HH2, p->g_w1_neigh[t], 1.f);
    }
    if (adam) {
        if (!adam->grads || !adam->params || !adam->m || !adam->v || adam->n < 1)
            return gs::fail(GS_EINVAL, "%s: adam needs grads, params, m, v and n > 0", who);
        if (outside) return gs::fail(GS_EINVAL, "%s: with adam every gradient pointer must lie inside adam->grads[0 .. n)", who);
        ra.adam = AdamArgs{adam->grads, adam->lr_t_device, adam->params, adam->m, adam->v,
                           adam->lr_t, adam->beta_1, adam->beta_2, adam->epsilon, adam->grad_scale};
    }
    hipStream_t s = (hipStream_t)stream;
    bool launched = false;
#define GS_ALL(f, hh2, u)                                                                                           \
    if (!launched && F == f && H2 == 2 * hh2 && (u == 8 || n0 <= 4 * u)) {                                                  \
        const void *fn = head ? reinterpret_cast<const void *>(sage2_bwd_all_kernel<f, hh2, u, true>)               \
                              : reinterpret_cast<const void *>(sage2_bwd_all_kernel<f, hh2, u, false>);             \
        const size_t lds = all_lds<f, hh2>();                                                                       \
        const hipError_t attr = gs::raise_lds_limit(fn, (int)lds);                                                  \
        if (attr != hipSuccess) return gs::fail(GS_ELAUNCH, "%s: %s", who, hipGetErrorString(attr));                \
        if (head) hipLaunchKernelGGL((sage2_bwd_all_kernel<f, hh2, u, true>), dim3((unsigned)blocks),               \
                                     dim3(kThreads), lds, s, a);                                                    \
        else hipLaunchKernelGGL((sage2_bwd_all_kernel<f, hh2, u, false>), dim3((unsigned)blocks), dim3(kThreads),   \
                                lds, s, a);                                                                         \
        launched = true;                                                                                            \
    }
    // (a walk's rows in one round of four steps for fan-outs up to 16, of eight up to 32, in two rounds of eight beyond)
    GS_ALL(32, 16, 4) GS_ALL(32, 16, 8) GS_ALL(64, 16, 4) GS_ALL(64, 16, 8)
#undef GS_ALL
    if (!launched) return gs::fail(GS_ELIMIT, "%s: no kernel for F=%d H2=%d", who, F, H2);
    if (int rc = gs::check_launch(who)) return rc;

    ra.n_seg = n;
    return launch_bwd_reduce(ra, a.slab, (int)blocks, slab_floats(F, H2), s, "gs_sage2_backward_all (reduce)");
}
