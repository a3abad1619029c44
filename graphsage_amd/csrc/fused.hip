// Fused SAGE layer: feature gather + neighbor mean + both projections + bias +
// activation (+ l2 normalisation) in one kernel, for the shapes of the reference
// model (feature/hidden widths 32..128, half-widths 16 or 32).
//
// Work unit: a wave owns a tile of 16 head nodes.
//   1. ids   -- the tile's child ids (16 * n_child, contiguous) and head ids are
//              staged into the wave's LDS region with coalesced loads.
//   2. self  -- the loads of the 16 own rows are issued (row-contiguous, FS/4
//              lanes x 16 B per row) and left in flight.
//   3. mean  -- 64/(FN/4) lane groups each take every GN-th head of the tile and
//              sum its n_child rows as they arrive (all of a head's row loads in
//              flight at once when n_child <= UNROLL, else balanced rounds; 16 B
//              per lane, whole rows per group), divide, and write the mean row
//              to the LDS tile A[16][F+4].  The (heads, n_child, FN) neighbor
//              tensor never exists in memory.
//   4. mfma  -- D[16 x HH] = A[16 x F] . W[F x HH] on v_mfma_f32_16x16x4_f32,
//              first for the means, then the own rows go through the same LDS
//              tile for the self half.  The order of k inside the dot product
//              is free, so lane (row r = lane&15, quarter q = lane>>4) feeds the
//              k-steps from its own contiguous quarter of the row, k = q*F/4 +
//              step: A fragments are plain 16-byte LDS reads, and the matching W
//              fragments W[q*F/4 + step][16*ct + r] sit in LDS in exactly the
//              order the lanes read them (staged once per block).
//   5. out   -- bias, relu/linear, optional row l2-normalisation (sum of squares
//              over the 16 lanes that share a row via xor-shuffles), 64-byte
//              contiguous stores per row and half.
// After the one-time weight staging, waves never synchronise with each other:
// all LDS hand-offs are wave-local.
#include "fused_args.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kWaves = 4;
using gs::FusedArgs;
using gs::Seg;
using gs::kMaxSegs;
constexpr int kUnroll = 10;
constexpr size_t kLdsLimit = 64 * 1024;

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int FS, int FN, int HH>
struct FusedShape {
    static constexpr int FMAX = FS > FN ? FS : FN;
    static constexpr int LDA = FMAX + 4;
    static constexpr int NT = HH / 16;
    static constexpr int KS_S = FS / 4, KS_N = FN / 4;
    static constexpr int W_FLOATS = (KS_S + KS_N) * NT * 64;
    static size_t lds_bytes(int max_child) {
        return (size_t)(W_FLOATS + kWaves * (16 * LDA + 16 * max_child + 16)) * sizeof(float);
    }
};

constexpr int kDrawChunks = 2;                               // x 64 draws of a tile kept in flight in registers

// Draws of up to kDrawChunks * 64 children of one tile, in three steps so that the two dependent reads
// of a draw (row bounds, then the neighbor) can be spread over other work.
struct DrawRegs {
    int32_t hid[kDrawChunks];
    int64_t ra[kDrawChunks], rb[kDrawChunks];
    int32_t v[kDrawChunks];
};

__device__ __forceinline__ int find_seg(const FusedArgs &a, long long tile) {
    int si = 0;
#pragma unroll
    for (int s = 1; s < kMaxSegs; ++s)
        if (s < a.n_segs && tile >= a.seg[s].tile0) si = s;
    return si;
}

// step 1: the heads' ids and their adjacency row bounds (child i of the tile belongs to head i / n)
__device__ __forceinline__ void draw_bounds(const gs::Draws &d, const Seg &sg, long long p0, int cnt, int n, int base,
                                            int lane, DrawRegs &q) {
#pragma unroll
    for (int c = 0; c < kDrawChunks; ++c) {
        const int i = base + c * 64 + lane;
        q.hid[c] = -1;
        if (i < cnt) q.hid[c] = sg.head_ids[p0 + (unsigned)i / (unsigned)n];
        if ((unsigned long long)(long long)q.hid[c] >= (unsigned long long)d.n_nodes) q.hid[c] = -1;
        const int32_t pc = q.hid[c] >= 0 ? q.hid[c] : 0;     // node 0 stands in for dead slots
        q.ra[c] = d.indptr[pc];
        q.rb[c] = d.indptr[pc + 1];
    }
}

// step 2: the random word of each draw (Philox counter (walk, hop, slot) or tape word) and the neighbor read
__device__ __forceinline__ void draw_ids(const gs::Draws &d, const Seg &sg, long long p0, int cnt, int n, int base,
                                         int lane, DrawRegs &q) {
    const unsigned hpw = (unsigned)sg.heads_per_walk;
    const long long w0 = sg.walk0 + p0 / (long long)hpw;    // wave-uniform
    const unsigned j0 = (unsigned)(p0 % (long long)hpw);
#pragma unroll
    for (int c = 0; c < kDrawChunks; ++c) {
        const int i = base + c * 64 + lane;
        const unsigned pp = (unsigned)i / (unsigned)n, cc = (unsigned)i - pp * (unsigned)n;
        const unsigned jj = j0 + pp;
        const unsigned dw = jj / hpw;
        const long long w = w0 + dw;
        const unsigned t = (jj - dw * hpw) * (unsigned)n + cc;               // slot inside the walk's hop
        uint32_t r = 0;
        if (i < cnt) {
            if (d.replay) r = d.replay[(size_t)w * d.tape_len + sg.tape_off + t];
            else {
                const unsigned long long walk = d.walk_ids ? (unsigned long long)d.walk_ids[w]
                                                           : d.walk_base + (unsigned long long)w;
                r = gs::philox4x32_10_x((uint32_t)walk, (uint32_t)(walk >> 32), (uint32_t)sg.hop, t, d.key0, d.key1);
            }
        }
        const uint32_t deg = (uint32_t)(q.rb[c] - q.ra[c]);
        const bool live = q.hid[c] >= 0 && deg != 0;
        if (q.hid[c] >= 0 && deg == 0) atomicMax(&d.err[0], q.hid[c] + 1);
        q.v[c] = d.indices[live ? q.ra[c] + (int64_t)__umulhi(r, deg) : 0];
        if (!live) q.v[c] = -1;
    }
}

// step 3: the ids into the wave's LDS tile (and into memory when the caller wants them)
__device__ __forceinline__ void draw_store(const Seg &sg, long long p0, int cnt, int n, int base, int lane,
                                           const DrawRegs &q, int32_t *s_child) {
#pragma unroll
    for (int c = 0; c < kDrawChunks; ++c) {
        const int i = base + c * 64 + lane;
        if (i < cnt) {
            s_child[i] = q.v[c];
            if (sg.out_child_ids) sg.out_child_ids[p0 * n + i] = q.v[c];
        }
    }
}

// SAMPLED: segments with hop >= 0 draw their children's ids in the kernel (the deepest hop of a walk,
// most of its samples) exactly as gs_sample would: same CSR reads, same Philox counter or tape word.
template <int FS, int FN, int HH, bool DENSE, bool SAMPLED = false>
__global__ __launch_bounds__(kWaves * 64, 4)
void sage_fused_kernel(const FusedArgs a) {
    static_assert(FS % 16 == 0 && FN % 16 == 0 && HH % 16 == 0, "widths must be multiples of 16");
    using S = FusedShape<FS, FN, HH>;
    constexpr int LDA = S::LDA, NT = S::NT, KS_S = S::KS_S, KS_N = S::KS_N;
    constexpr int FSV = FS / 4, FNV = FN / 4;            // lanes per row (float4 each)
    constexpr int RP_S = 64 / FSV;                       // own rows per load instruction
    constexpr int NPASS = 16 / RP_S;
    constexpr int GN = 64 / FNV;                         // lane groups in the mean phase
    constexpr int H = 2 * HH;

    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *Wl_s = lds;                                   // [KS_S][NT][64] fragment order
    float *Wl_n = Wl_s + KS_S * NT * 64;
    const int per_wave = 16 * LDA + 16 * a.max_child + 16;
    float *A = Wl_n + KS_N * NT * 64 + (size_t)wave * per_wave;
    int32_t *s_child = reinterpret_cast<int32_t *>(A + 16 * LDA);
    int32_t *s_head = s_child + 16 * a.max_child;

    const int r = lane & 15, q = lane >> 4;

    // weight fragments in the order lanes consume them: [(ks*NT + ct)*64 + 16 q + r]
    for (int idx = threadIdx.x; idx < KS_S * NT * 64; idx += kWaves * 64) {
        const int rr = idx & 15, qq = (idx >> 4) & 3, ct = (idx >> 6) % NT, ks = (idx >> 6) / NT;
        Wl_s[idx] = a.W_self[(size_t)(qq * KS_S + ks) * HH + ct * 16 + rr];
    }
    for (int idx = threadIdx.x; idx < KS_N * NT * 64; idx += kWaves * 64) {
        const int rr = idx & 15, qq = (idx >> 4) & 3, ct = (idx >> 6) % NT, ks = (idx >> 6) / NT;
        Wl_n[idx] = a.W_neigh[(size_t)(qq * KS_N + ks) * HH + ct * 16 + rr];
    }
    float bias_v[2][NT];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) bias_v[hf][ct] = a.bias ? a.bias[hf * HH + ct * 16 + r] : 0.f;
    __syncthreads();

    const long long stride = (long long)gridDim.x * kWaves;
    DrawRegs pre{};                                          // draws of this wave's next tile, in flight
    bool have_ids = false;                                   // ... and whether `pre` holds this tile's ids
    for (long long tile = (long long)blockIdx.x * kWaves + wave; tile < a.n_tiles; tile += stride) {
        const int si = find_seg(a, tile);
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
        if (SAMPLED && sg.hop >= 0) {
            // Children of this tile's heads are drawn here, exactly as gs_sample draws them.  Usually
            // they were requested while the previous tile's rows were being gathered (`pre`);
            // otherwise (first tile of the wave, very wide fan-outs) chunk by chunk on the spot.
            if (have_ids) {
                draw_store(sg, p0, np * n, n, 0, lane, pre, s_child);
            } else {
                for (int base = 0; base < np * n; base += 64 * kDrawChunks) {
                    DrawRegs now{};
                    draw_bounds(a.draws, sg, p0, np * n, n, base, lane, now);
                    draw_ids(a.draws, sg, p0, np * n, n, base, lane, now);
                    draw_store(sg, p0, np * n, n, base, lane, now, s_child);
                }
            }
        } else if (!DENSE) {
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

        // 1b. the next tile's draws: its heads' row bounds are requested now and arrive under the
        // first round of row gathers below; the neighbor reads follow after that round.
        bool ahead = false;
        long long p0n = 0;
        int cntn = 0, nn = 1, sin = 0;
        if (SAMPLED && tile + stride < a.n_tiles) {
            sin = find_seg(a, tile + stride);
            const Seg &sn = a.seg[sin];
            p0n = (tile + stride - sn.tile0) * 16;
            nn = sn.n_child;
            cntn = (int)min((long long)16, sn.n_heads - p0n) * nn;
            ahead = sn.hop >= 0 && cntn <= 64 * kDrawChunks;
            if (ahead) draw_bounds(a.draws, sn, p0n, cntn, nn, 0, lane, pre);
        }

        // 2. own rows: issue the loads, consume them after the mean phase.  Invalid rows read
        // row 0 and are zeroed by a select, so the load stream has no branches.
        float4 own[NPASS];
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int rr = pass * RP_S + lane / FSV, c = lane % FSV;
            const int32_t row = s_head[rr];
            const float4 t = reinterpret_cast<const float4 *>(a.feat_self + (size_t)(row >= 0 ? row : 0) * FS)[c];
            own[pass] = row >= 0 ? t : make_float4(0.f, 0.f, 0.f, 0.f);
        }

        // 3. neighbor means -> A.  Lane group g owns heads g, g + GN, ...; their children form one
        // list of HPG * n rows that is fetched kUnroll rows at a time whatever the fan-out (two heads
        // per round at n = 5), and summed in child order; a head's mean leaves when its last child
        // has been added.
        {
            constexpr int HPG = 16 / GN;
            const int g = lane / FNV, c = lane % FNV;
            const float denom = (float)n;
            const int R = HPG * n;
            const int rounds = (R + kUnroll - 1) / kUnroll;
            const int per = (R + rounds - 1) / rounds;
            const float4 *table = reinterpret_cast<const float4 *>(a.feat_neigh) + c;
            int kl = 0, jl = 0;                              // head / child the load stream is at
            int ka = 0, ja = 0;                              // ... and the sum
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int f0 = 0; f0 < R; f0 += per) {
                float4 v[kUnroll];
                unsigned okmask = 0;
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    const bool in = (u < per) && (f0 + u < R);
                    const int pp = g + kl * GN;
                    const bool head_ok = in && pp < np;
                    long long rw = -1;
                    if (DENSE) rw = child_row0 + (p0 + pp) * n + jl;
                    else if (head_ok) rw = (long long)s_child[pp * n + jl] - a.off_neigh;
                    const bool ok = head_ok && ((unsigned long long)rw < (unsigned long long)a.n_rows_neigh);
                    okmask |= (unsigned)ok << u;
                    v[u] = table[(ok ? rw : 0) * FNV];
                    if (in && ++jl == n) { jl = 0; ++kl; }
                }
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    const bool in = (u < per) && (f0 + u < R);
                    const bool ok = (okmask >> u) & 1;
                    acc.x += ok ? v[u].x : 0.f; acc.y += ok ? v[u].y : 0.f;
                    acc.z += ok ? v[u].z : 0.f; acc.w += ok ? v[u].w : 0.f;
                    if (in && ++ja == n) {
                        const int pp = g + ka * GN;
                        acc.x /= denom; acc.y /= denom; acc.z /= denom; acc.w /= denom;
                        if (a.save_mean && pp < np)
                            reinterpret_cast<float4 *>(a.save_mean + (size_t)(out_row0 + p0 + pp) * FN)[c] = acc;
                        *reinterpret_cast<float4 *>(A + pp * LDA + 4 * c) = acc;
                        acc = make_float4(0.f, 0.f, 0.f, 0.f);
                        ja = 0; ++ka;
                    }
                }
                if (SAMPLED && ahead && f0 == 0)             // the next tile's row bounds have arrived
                    draw_ids(a.draws, a.seg[sin], p0n, cntn, nn, 0, lane, pre);
            }
        }
        have_ids = ahead;
        wave_lds_fence();

        // 4a. neighbor half.  The fragment offset goes through an empty asm so the weight
        // reads stay inside the tile loop instead of being hoisted into 2*F/4 registers.
        int wl = lane;
        asm volatile("" : "+v"(wl));
        f32x4 acc_s[NT], acc_n[NT];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) { acc_s[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_n[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int k4 = 0; k4 < KS_N / 4; ++k4) {
            const float4 av = *reinterpret_cast<const float4 *>(A + r * LDA + q * KS_N + 4 * k4);
            const float ae[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ct = 0; ct < NT; ++ct)
                    acc_n[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[e], Wl_n[((4 * k4 + e) * NT + ct) * 64 + wl],
                                                                     acc_n[ct], 0, 0, 0);
        }
        wave_lds_fence();                                   // A is free again

        // 4b. own rows -> A -> self half
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int rr = pass * RP_S + lane / FSV, c = lane % FSV;
            *reinterpret_cast<float4 *>(A + rr * LDA + 4 * c) = own[pass];
            if (a.save_self && rr < np)
                reinterpret_cast<float4 *>(a.save_self + (size_t)(out_row0 + p0 + rr) * FS)[c] = own[pass];
        }
        wave_lds_fence();
#pragma unroll
        for (int k4 = 0; k4 < KS_S / 4; ++k4) {
            const float4 av = *reinterpret_cast<const float4 *>(A + r * LDA + q * KS_S + 4 * k4);
            const float ae[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ct = 0; ct < NT; ++ct)
                    acc_s[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[e], Wl_s[((4 * k4 + e) * NT + ct) * 64 + wl],
                                                                     acc_s[ct], 0, 0, 0);
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
    const size_t lds = FusedShape<FS, FN, HH>::lds_bytes(a.max_child);
    if (lds > kLdsLimit) return gs::fail(GS_ELIMIT, "gs_sage_fused_fwd: %zu B of LDS per block needed (limit %zu)", lds, kLdsLimit);
    long long blocks = (a.n_tiles + kWaves - 1) / kWaves;
    const long long cap = 256 * 8;                      // a few blocks per CU; waves loop over tiles
    if (blocks > cap) blocks = cap;
    bool sampled = false;
    for (int s = 0; s < a.n_segs; ++s) sampled |= a.seg[s].hop >= 0;
    if (sampled)
        gs::launch_timed(sage_fused_kernel<FS, FN, HH, false, true>, dim3((unsigned)blocks), dim3(kWaves * 64), lds, stream, a);
    else if (a.seg[0].child_ids)
        gs::launch_timed(sage_fused_kernel<FS, FN, HH, false>, dim3((unsigned)blocks), dim3(kWaves * 64), lds, stream, a);
    else
        gs::launch_timed(sage_fused_kernel<FS, FN, HH, true>, dim3((unsigned)blocks), dim3(kWaves * 64), lds, stream, a);
    return gs::check_launch("gs_sage_fused_fwd");
}

inline size_t fused_lds_bytes(int Fs, int Fn, int hh, int max_child) {
    const int fmax = Fs > Fn ? Fs : Fn;
    return (size_t)((Fs / 4 + Fn / 4) * (hh / 16) * 64 + kWaves * (16 * (fmax + 4) + 16 * max_child + 16)) * sizeof(float);
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// shapes of the one-wave-per-16-heads kernel of this file
static bool narrow_supported(int Fs, int Fn, int H, int max_child) {
    const int hh = H / 2;
    const bool f_ok = (Fs == 32 || Fs == 64 || Fs == 128) && (Fn == 32 || Fn == 64 || Fn == 128);
    const bool h_ok = (H % 2 == 0) && (hh == 16 || hh == 32);
    if (!f_ok || !h_ok || max_child < 1) return 0;
    return fused_lds_bytes(Fs, Fn, hh, max_child) <= kLdsLimit;
}

extern "C" int gs_sage_fused_supported(int Fs, int Fn, int H, int max_child) {
    return narrow_supported(Fs, Fn, H, max_child) || gs::sage_wide_supported(Fs, Fn, H, max_child);
}

static int fused_fwd(const float *feat_self, int64_t n_rows_self, int Fs, int64_t off_self,
                     const float *feat_neigh, int64_t n_rows_neigh, int Fn, int64_t off_neigh,
                     const gs_fused_segment *segs, int n_segs,
                     const float *W_self, const float *W_neigh, const float *bias,
                     int H, int act, int normalize,
                     float *out, float *save_self, float *save_mean,
                     const gs_fused_sampling *sampling, const gs_fused_draw *draws, void *workspace, size_t workspace_bytes,
                     void *stream);

extern "C" size_t gs_sage_fused_workspace_bytes(int Fs, int Fn, int H, int max_child) {
    if (narrow_supported(Fs, Fn, H, max_child) || !gs::sage_wide_supported(Fs, Fn, H, max_child)) return 0;
    return gs::sage_wide_workspace_bytes(Fs, Fn, H);
}

extern "C" int gs_sage_fused_fwd_ws(const float *feat_self, int64_t n_rows_self, int Fs, int64_t off_self,
                                    const float *feat_neigh, int64_t n_rows_neigh, int Fn, int64_t off_neigh,
                                    const gs_fused_segment *segs, int n_segs,
                                    const float *W_self, const float *W_neigh, const float *bias,
                                    int H, int act, int normalize,
                                    float *out, float *save_self, float *save_mean,
                                    void *workspace, size_t workspace_bytes, void *stream) {
    return fused_fwd(feat_self, n_rows_self, Fs, off_self, feat_neigh, n_rows_neigh, Fn, off_neigh, segs, n_segs,
                     W_self, W_neigh, bias, H, act, normalize, out, save_self, save_mean, nullptr, nullptr,
                     workspace, workspace_bytes, stream);
}

extern "C" int gs_sage_fused_fwd(const float *feat_self, int64_t n_rows_self, int Fs, int64_t off_self,
                                 const float *feat_neigh, int64_t n_rows_neigh, int Fn, int64_t off_neigh,
                                 const gs_fused_segment *segs, int n_segs,
                                 const float *W_self, const float *W_neigh, const float *bias,
                                 int H, int act, int normalize,
                                 float *out, float *save_self, float *save_mean, void *stream) {
    return fused_fwd(feat_self, n_rows_self, Fs, off_self, feat_neigh, n_rows_neigh, Fn, off_neigh, segs, n_segs,
                     W_self, W_neigh, bias, H, act, normalize, out, save_self, save_mean, nullptr, nullptr, nullptr, 0, stream);
}

extern "C" int gs_sage_fused_sampled_fwd(const float *feat_self, int64_t n_rows_self, int Fs, int64_t off_self,
                                         const float *feat_neigh, int64_t n_rows_neigh, int Fn, int64_t off_neigh,
                                         const gs_fused_segment *segs, int n_segs,
                                         const float *W_self, const float *W_neigh, const float *bias,
                                         int H, int act, int normalize,
                                         float *out, float *save_self, float *save_mean,
                                         const gs_fused_sampling *sampling, const gs_fused_draw *draws, void *stream) {
    if (!sampling || !draws) return gs::fail(GS_EINVAL, "gs_sage_fused_sampled_fwd: sampling and draws must be given");
    return fused_fwd(feat_self, n_rows_self, Fs, off_self, feat_neigh, n_rows_neigh, Fn, off_neigh, segs, n_segs,
                     W_self, W_neigh, bias, H, act, normalize, out, save_self, save_mean, sampling, draws, nullptr, 0, stream);
}

static int fused_fwd(const float *feat_self, int64_t n_rows_self, int Fs, int64_t off_self,
                     const float *feat_neigh, int64_t n_rows_neigh, int Fn, int64_t off_neigh,
                     const gs_fused_segment *segs, int n_segs,
                     const float *W_self, const float *W_neigh, const float *bias,
                     int H, int act, int normalize,
                     float *out, float *save_self, float *save_mean,
                     const gs_fused_sampling *sampling, const gs_fused_draw *draws, void *workspace, size_t workspace_bytes,
                     void *stream) {
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
        const bool drawn = draws && draws[s].hop >= 0;
        if (!drawn && !g.child_ids && (g.child_row0 < 0 || g.child_row0 + g.n_heads * g.n_child > n_rows_neigh))
            return gs::fail(GS_EINVAL, "gs_sage_fused_fwd: segment %d dense child rows exceed feat_neigh", s);
        if (!draws && s > 0 && (g.child_ids == nullptr) != (segs[0].child_ids == nullptr))
            return gs::fail(GS_EINVAL, "gs_sage_fused_fwd: segments must all give child ids or all be dense");
        if (draws && !drawn && !g.child_ids)
            return gs::fail(GS_EINVAL, "gs_sage_fused_sampled_fwd: segment %d neither draws its children nor gives child ids", s);
        Seg &d = a.seg[s];
        d.head_ids = g.head_ids; d.child_ids = g.child_ids; d.n_heads = g.n_heads;
        d.self_row0 = g.self_row0; d.child_row0 = g.child_row0; d.n_child = g.n_child;
        d.tile0 = tiles; d.out_row0 = rows;
        d.hop = -1; d.heads_per_walk = 1; d.tape_off = 0; d.walk0 = 0; d.out_child_ids = nullptr;
        if (drawn) {
            const gs_fused_draw &w = draws[s];
            if (!g.head_ids) return gs::fail(GS_EINVAL, "gs_sage_fused_sampled_fwd: segment %d draws children but gives no head ids", s);
            if (w.hop >= GS_MAX_HOPS || w.heads_per_walk < 1 || w.walk0 < 0 || w.tape_off < 0)
                return gs::fail(GS_EINVAL, "gs_sage_fused_sampled_fwd: segment %d: hop=%d heads_per_walk=%d walk0=%lld tape_off=%d",
                                s, w.hop, w.heads_per_walk, (long long)w.walk0, w.tape_off);
            if ((long long)w.heads_per_walk * g.n_child > (1 << 24))
                return gs::fail(GS_ELIMIT, "gs_sage_fused_sampled_fwd: segment %d has more than 2^24 slots per walk", s);
            if (sampling->replay && (long long)w.tape_off + (long long)w.heads_per_walk * g.n_child > sampling->tape_len)
                return gs::fail(GS_EINVAL, "gs_sage_fused_sampled_fwd: segment %d reads past a walk's tape", s);
            d.hop = w.hop; d.heads_per_walk = w.heads_per_walk; d.tape_off = w.tape_off; d.walk0 = w.walk0;
            d.out_child_ids = w.out_child_ids;
        }
        tiles += (g.n_heads + 15) / 16;
        rows += g.n_heads;
        if (g.n_child > max_child) max_child = g.n_child;
    }
    if (tiles == 0) return GS_OK;
    if (!feat_self || !feat_neigh || !W_self || !W_neigh || !out)
        return gs::fail(GS_EINVAL, "gs_sage_fused_fwd: null pointer argument");
    if (n_rows_self < 1 || n_rows_neigh < 1)
        return gs::fail(GS_EINVAL, "gs_sage_fused_fwd: feature tables must have at least one row");
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
    bool any_drawn = false;
    for (int i = 0; i < n_segs; ++i) any_drawn |= a.seg[i].hop >= 0;
    if (any_drawn) {
        if (!sampling->indptr || !sampling->indices || !sampling->err_flag)
            return gs::fail(GS_EINVAL, "gs_sage_fused_sampled_fwd: null pointer in the sampling descriptor");
        if (sampling->n_nodes <= 0 || sampling->n_nodes > 0x7FFFFFFELL)
            return gs::fail(GS_EINVAL, "gs_sage_fused_sampled_fwd: n_nodes=%lld out of range", (long long)sampling->n_nodes);
        if (sampling->replay && sampling->tape_len < 1)
            return gs::fail(GS_EINVAL, "gs_sage_fused_sampled_fwd: a replay tape needs its length per walk");
        a.draws.indptr = sampling->indptr; a.draws.indices = sampling->indices; a.draws.n_nodes = sampling->n_nodes;
        a.draws.key0 = (uint32_t)sampling->rng_seed; a.draws.key1 = (uint32_t)(sampling->rng_seed >> 32);
        a.draws.walk_base = sampling->walk_id_base; a.draws.walk_ids = sampling->walk_ids;
        a.draws.replay = sampling->replay; a.draws.tape_len = sampling->tape_len; a.draws.err = sampling->err_flag;
    }
    hipStream_t s = (hipStream_t)stream;
    if (!narrow_supported(Fs, Fn, H, max_child)) {                                            // wide layers: wide.hip
        if (any_drawn) return gs::fail(GS_ELIMIT, "gs_sage_fused_sampled_fwd: the wide-layer kernel reads child ids; "
                                                  "run gs_sample for every hop and use gs_sage_fused_fwd");
        if (workspace) {
            const size_t need = gs::sage_wide_workspace_bytes(Fs, Fn, H);
            if (workspace_bytes < need)
                return gs::fail(GS_EINVAL, "gs_sage_fused_fwd_ws: workspace of %zu bytes, %zu needed (gs_sage_fused_workspace_bytes)",
                                workspace_bytes, need);
            a.split_ws = workspace;
        }
        return gs::sage_wide_fwd(a, Fs, Fn, H, s);
    }
    const int hh = H / 2;
#define GS_CASE(fs, fn, h2) if (Fs == fs && Fn == fn && hh == h2) return launch<fs, fn, h2>(a, s);
    GS_CASE(64, 64, 16) GS_CASE(64, 64, 32)
    GS_CASE(32, 32, 16) GS_CASE(32, 32, 32)
    GS_CASE(64, 32, 16) GS_CASE(32, 64, 16) GS_CASE(64, 32, 32) GS_CASE(32, 64, 32)
    GS_CASE(128, 128, 16) GS_CASE(128, 128, 32)
    GS_CASE(128, 64, 16) GS_CASE(64, 128, 16) GS_CASE(128, 32, 16) GS_CASE(32, 128, 16)
    GS_CASE(128, 64, 32) GS_CASE(64, 128, 32) GS_CASE(128, 32, 32) GS_CASE(32, 128, 32)
#undef GS_CASE
    return gs::fail(GS_ELIMIT, "gs_sage_fused_fwd: no instantiation for Fs=%d Fn=%d H=%d", Fs, Fn, H);
}
