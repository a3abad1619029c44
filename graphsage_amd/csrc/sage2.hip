// Two-hop forward pass in ONE kernel: sampling, both aggregator layers and the
// l2 normalisation, one wavefront per walk (= one root of a seed edge).
//
// The reference's generator and model are written for exactly two hops
// (generator.py:239-244,339-345); for that case every quantity a root's
// embedding depends on is private to its walk:
//
//   root r (type T) -- n0 sampled neighbors h_j (type O) -- n1 sampled
//   neighbors c_jk (type T) each
//   layer 0:  z_j = relu(W0_O . [x_O(h_j) ; mean_k x_T(c_jk)] + b0_O)      j < n0
//             z_r = relu(W0_T . [x_T(r)   ; mean_j x_O(h_j)]  + b0_T)
//   layer 1:  e_r = l2norm(W1_T . [z_r ; mean_j z_j] + b1_T)
//
// so a wave can sample its walk into LDS, stream the 1 + n0 + n0*n1 feature rows
// exactly once (the rows of the h_j feed both z_j and the root's neighbor mean),
// keep the layer-0 activations in LDS and write nothing but the final embedding
// row.  Neither the sampled ids nor any intermediate tensor touch memory unless
// the caller asks for the ids.
//
// Per wave: hop-1 heads are processed in tiles of 16 (own rows issued first and
// left in flight, the children's rows summed as they arrive by 64/(F/4) lane
// groups, both halves of the projection on v_mfma_f32_16x16x4_f32 through one
// LDS A-tile), then the root's tile, then layer 1 on the LDS-resident
// activations.  All weight matrices of the model sit in LDS in MFMA B-fragment
// order (staged once per block); a lane's k-steps walk its own contiguous
// quarter of the A row (k = q*K/4 + step), which is legal because the order of
// the terms of a dot product is free.  Waves never synchronise with each other
// after the weight staging: every LDS hand-off is wave-local.
#include "gs_common.h"
#include "sage_tile.h"

namespace {

using namespace gs::tile;


// Waves per block (one block per CU).  With 32-wide activations everything a wave keeps in LDS lets
// sixteen waves fit: four per SIMD, 128 registers each, one head's children in flight per lane
// group.  64-wide activations double the weights and the activation rows: the forms that keep the rows
// (saving, subtrees) fit eight waves (two per SIMD, 256 registers each) and keep two heads' children in flight
// per lane group instead: the same bytes in flight per CU.
// (The plain walk kernel at 64-wide activations keeps no row of the hop-1 heads' activations -- layer 1 needs only
// their mean, which is summed as the tiles finish: sum_rows() -- so twelve waves fit where the rows allowed eight.)
constexpr bool sum_rows(int hh1, bool save, bool sub) { return hh1 >= 32 && !save && !sub; }
// (128-wide features: the weights take half of LDS, eight waves are all that fit)
constexpr int max_waves(int hh1, int fmax) { return hh1 >= 32 ? (fmax > 64 ? 8 : 12) : 16; }
constexpr int heads_in_flight(int hh1) { return hh1 >= 32 ? 2 : 1; }
// Subtree mode: twelve waves (three per SIMD, 170 registers each).  At sixteen the group loop spills 39 vector
// registers, and the spill traffic costs more than the four extra waves hide (588 against 528 us per launch on
// three hops at 64-wide features, fan-outs [25, 10, 5]).
constexpr int sub_waves(int hh1) { return hh1 >= 32 ? 8 : 12; }
// The saving form (training) likewise: 31 spilled vector registers at sixteen waves, none at twelve (training step
// 194 -> 189 us on the default workload); the plain walk kernel has none at sixteen and is faster there.
constexpr int block_waves(int hh1, bool save, bool sub, int fmax, bool drop = false) {
    if (sub) return sub_waves(hh1);
    if (save) return (hh1 < 32 && fmax <= 64 && !drop) ? max_waves(hh1, fmax) : sub_waves(hh1);
    return max_waves(hh1, fmax);
}
constexpr int kUnroll = 10;              // row loads in flight per lane in the mean phase
constexpr int kChunks = 4;               // 64-slot chunks sampled per pass
constexpr int kGroup = 8;                // subtree mode: roots a wavefront takes together

struct Sage2Args {
    const int64_t *indptr; const int32_t *indices; long long n_nodes;
    const float *feat[2]; long long n_rows[2];         // [0] users, [1] items; item id offset = n_rows[0]
    const int32_t *seeds; long long n_pairs;
    int n0, n1;
    unsigned key0, key1; unsigned long long walk_base;
    const int64_t *walk_ids; const uint32_t *replay;
    const float *w0_self[2], *w0_neigh[2], *b0[2];     // layer 0, by head type
    const float *w1_self[2], *w1_neigh[2], *b1[2];     // layer 1, by head type
    int normalize;
    float *emb; int32_t *out_hop1, *out_hop2; int32_t *err;
    int ids_per_wave, h1_rows;
    // training: what the backward pass reads, by node type (NULL pointers when SAVE is off).
    // Tables of n_pairs * (1 + n0) rows: row p is the root of pair p of that type, row
    // n_pairs + p*n0 + j hop-1 head j of the walk rooted at the other type's root of pair p.
    float *sv_x_self[2], *sv_x_mean[2], *sv_h1[2];
    float *sv_x_mean1[2], *sv_z[2];                   // n_pairs rows: layer-1 neighbor mean, pre-norm output
    int32_t *sv_self_ids[2];                          // same rows as sv_x_self: the node id instead of its feature row
    uint32_t *sv_h1_mask[2];                          // n_pairs * n0 * H1/32 words: [h1 > 0] of the hop-1 heads, instead of their rows
    // training with dropout (DROP instantiations): threshold rate * 2^32, 1 / (1 - rate), key
    uint32_t drop_thr, drop_k0, drop_k1;
    float drop_scale;
    // Subtree mode (sub_n > 0): root i is hop-1 node i % sub_n of seed walk i / sub_n of a deeper model, whose
    // hops 1 and 2 this kernel draws (sampler hop = local hop + 1, slots shifted by the subtree's position);
    // the item-typed roots come first (they hang below the user-rooted seed walks).
    int sub_n;
    long long tape_len, tape_off0, tape_off1;         // replay tape: words per seed walk, offsets of local hops 0 / 1
    float *out_h0;                                    // nullable: [roots, H1] layer-0 output of every root
    // nullable: the 2*n_pairs/sub_n seed nodes.  Then the roots themselves are drawn here (sampler hop 0 of the
    // seed walks, slot = the root's position below its seed) and written to hop1_out instead of read from seeds.
    const int32_t *seeds0;
    int32_t *hop1_out;
    // subtree mode, training (SAVE): sv_x_mean / sv_h1_mask hold, by node type, the neighbor means and the relu-mask
    // words of the heads (row = root index within its type * n0 + head); and by the roots' type, one row per root:
    float *sv_root_mean0[2], *sv_root_mean1[2];       // the layer-0 / layer-1 neighbor means of the roots
};

template <int FU, int FI, int HH1, int HH2>
struct Sage2Shape {
    static constexpr int FMAX = FU > FI ? FU : FI;
    static constexpr int H1 = 2 * HH1, H2 = 2 * HH2;
    static constexpr int KMAX = FMAX > H1 ? FMAX : H1;
    static constexpr int LDA = KMAX + 4;
    static constexpr int LDH = H1;
    static constexpr int NT1 = HH1 / 16, NT2 = HH2 / 16;
    // fragment floats: layer 0 per type (self: own width, neigh: other width), layer 1 per type (2 x H1)
    static constexpr int W0_FLOATS = 2 * (FU + FI) * HH1;
    static constexpr int W1_FLOATS = 2 * H1 * HH2;          // a block serves one side: one layer-1 weight set
    static constexpr int B_FLOATS = 2 * H1 + H2;
    // roots: 0 for whole walks, kGroup in subtree mode; summed: the heads' activation rows are not kept
    static size_t lds_bytes(int n0, int n1, int waves, int roots = 0, bool summed = false) {
        const int ids = ((roots ? roots : 1) * (n0 + n0 * n1) + 3) & ~3;
        const int h1_rows = roots ? roots : summed ? 1 : 1 + n0;   // subtree mode keeps the roots' rows only
        return (size_t)(W0_FLOATS + W1_FLOATS + B_FLOATS + waves * (16 * LDA + h1_rows * LDH + ids)) * sizeof(float);
    }
};

// One walk.  FT / FO: feature widths of the root's type and of the other type.
template <int FT, int FO, int HH1, int HH2, int LDA, int LDH, bool BUF, bool SAVE, bool DROP>
__device__ __forceinline__ void walk_body(const Sage2Args &a, long long w, int side, int lane,
                                          int32_t root, int64_t ra, int64_t rb,
                                          int32_t next_root, int64_t &next_ra, int64_t &next_rb,
                                          float *A, float *H1t, int32_t *s_ids,
                                          const float *W0s_T, const float *W0n_T, const float *W0s_O, const float *W0n_O,
                                          const float *W1s_T, const float *W1n_T,
                                          const float *b0_T, const float *b0_O, const float *b1_T) {
    constexpr int NT1 = HH1 / 16, NT2 = HH2 / 16, H1 = 2 * HH1, H2 = 2 * HH2;
    constexpr int FTV = FT / 4, FOV = FO / 4, H1V = H1 / 4;
    const int r = lane & 15, q = lane >> 4;
    const int n0 = a.n0, n1 = a.n1;
    RowTable<BUF> tabT, tabO;
    tabT.init(a.feat[side], a.n_rows[side], FT * 4);
    tabO.init(a.feat[1 - side], a.n_rows[1 - side], FO * 4);
    const int offT = side ? (int)a.n_rows[0] : 0, offO = side ? 0 : (int)a.n_rows[0];
    int32_t *hop1 = s_ids, *hop2 = s_ids + n0;
    int wl = lane;
    asm volatile("" : "+v"(wl));            // keeps the weight-fragment reads inside the walk loop
    // The lane number as the saving stores see it: a value of this walk, so that the per-lane part of their addresses
    // is computed where a store is made.  From `lane` itself the compiler computes every table's per-lane address once
    // per kernel and keeps the fourteen of them (two registers each) alive through the gathers.
    int sl = lane;
    if constexpr (SAVE) asm volatile("" : "+v"(sl));

    // ---- sampling ---------------------------------------------------------------
    // the caller maps out-of-range roots to -1
    if (root < 0 && lane == 0) atomicMax(&a.err[1], 1);
    const unsigned long long walk = a.walk_ids ? (unsigned long long)a.walk_ids[w] : a.walk_base + (unsigned long long)w;
    const unsigned wlo = (unsigned)walk, whi = (unsigned)(walk >> 32);
    const gs::DropKey dk{wlo, whi, a.drop_k0, a.drop_k1, a.drop_thr};       // (DROP only)
    const float dscale = a.drop_scale;
    const uint32_t *tape = a.replay ? a.replay + (size_t)w * (n0 + n0 * n1) : nullptr;
    const uint32_t *tape0 = tape;
    const uint32_t *tape1 = tape ? tape + n0 : nullptr;
    wave_lds_fence();                        // the previous walk's readers of s_ids / A / H1t are done
    int32_t my_hop1 = -1;                    // lane t: hop-1 head t (SAVE: stored later, see below)
    {
        const uint32_t deg = (uint32_t)(rb - ra);
        if (root >= 0 && deg == 0 && lane == 0) atomicMax(&a.err[0], root + 1);
        const bool live = root >= 0 && deg != 0;
        for (int t = lane; t < n0; t += 64) {                // (n0 <= 64: one trip)
            const uint32_t rr = tape ? tape0[t] : gs::philox4x32_10_x(wlo, whi, 0u, (uint32_t)t, a.key0, a.key1);
            int32_t v = a.indices[live ? ra + (int64_t)__umulhi(rr, deg) : 0];
            if (!live) v = -1;
            hop1[t] = v;
            my_hop1 = v;
            if (a.out_hop1) a.out_hop1[(size_t)w * n0 + t] = v;
        }
    }
    wave_lds_fence();
    {
        const int total = n0 * n1;
        for (int base = 0; base < total; base += kChunks * 64) {
            int32_t p[kChunks];
            int64_t ca[kChunks], cb[kChunks];
#pragma unroll
            for (int c = 0; c < kChunks; ++c) {
                const int t = base + c * 64 + lane;
                p[c] = t < total ? hop1[t / n1] : -1;
                const int32_t pc = p[c] >= 0 ? p[c] : 0;
                ca[c] = a.indptr[pc];
                cb[c] = a.indptr[pc + 1];
            }
            int32_t v[kChunks];
#pragma unroll
            for (int c = 0; c < kChunks; ++c) {
                const int t = base + c * 64 + lane;
                const uint32_t deg = (uint32_t)(cb[c] - ca[c]);
                uint32_t rr = 0;
                if (tape) { if (t < total) rr = tape1[t]; }
                else rr = gs::philox4x32_10_x(wlo, whi, 1u, (uint32_t)t, a.key0, a.key1);
                const bool live = p[c] >= 0 && deg != 0;
                if (p[c] >= 0 && deg == 0) atomicMax(&a.err[0], p[c] + 1);
                v[c] = a.indices[live ? ca[c] + (int64_t)__umulhi(rr, deg) : 0];
                if (!live) v[c] = -1;
            }
#pragma unroll
            for (int c = 0; c < kChunks; ++c) {
                const int t = base + c * 64 + lane;
                if (t < total) {
                    hop2[t] = tabT.entry(v[c], offT, FT * 4);
                    if (a.out_hop2) a.out_hop2[(size_t)w * total + t] = v[c];
                }
            }
        }
    }
    wave_lds_fence();
    {   // row bounds of the NEXT walk's root: in flight while this walk gathers
        const int32_t nr = next_root >= 0 ? next_root : 0;
        next_ra = a.indptr[nr];
        next_rb = a.indptr[nr + 1];
    }

    // ---- layer 0, hop-1 heads (type O; children type T) ---------------------------------
    const long long pair = w - (side ? a.n_pairs : 0);
    const long long sv_h1row = a.n_pairs + pair * n0;       // first row of this walk's hop-1 heads in the type-O tables
    constexpr int RP_O = 64 / FOV, NPASS_O = 16 / RP_O;     // own rows per load instruction / passes per tile
    constexpr int GN_T = 64 / FTV;                           // lane groups summing children rows
    float4 rootsum = make_float4(0.f, 0.f, 0.f, 0.f);        // partial sum of the hop-1 rows, for the root's mean
    constexpr bool SUMROWS = sum_rows(HH1, SAVE, false);
    float hsum_s[NT1], hsum_n[NT1];                          // SUMROWS: this lane's part (rows 4q .. 4q+3 of every tile, column
#pragma unroll                                               // ct*16 + r of either half) of the sum of the heads' activations
    for (int ct = 0; ct < NT1; ++ct) { hsum_s[ct] = 0.f; hsum_n[ct] = 0.f; }
    const int rounds = (n1 + kUnroll - 1) / kUnroll;
    const int per = (n1 + rounds - 1) / rounds;
    for (int j0 = 0; j0 < n0; j0 += 16) {
        const int np = min(16, n0 - j0);
        float4 own[NPASS_O];
#pragma unroll
        for (int pass = 0; pass < NPASS_O; ++pass) {
            const int rr = pass * RP_O + lane / FOV, c = lane % FOV;
            const int32_t id = hop1[j0 + min(rr, np - 1)];
            own[pass] = tabO.load(rr < np, (unsigned)(id - offO), (unsigned)c * 16u, FO * 4);
        }
        {
            // kHeads heads per lane group and iteration: their row loads are all issued before
            // the first add, which is what keeps enough bytes in flight per wave
            const int g = lane / FTV, c = lane % FTV;
            const unsigned col = (unsigned)c * 16u;
            const float denom = (float)n1;
            constexpr int kHeads = heads_in_flight(HH1);
            for (int pp0 = g; pp0 < 16; pp0 += kHeads * GN_T) {
                float4 acc[kHeads];
#pragma unroll
                for (int hd = 0; hd < kHeads; ++hd) acc[hd] = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int c0 = 0; c0 < n1; c0 += per) {
                    float4 v[kHeads][kUnroll];
#pragma unroll
                    for (int hd = 0; hd < kHeads; ++hd) {
                        const int pp = pp0 + hd * GN_T;
                        const int32_t *ids = hop2 + (j0 + min(pp, np - 1)) * n1;
                        int32_t cid[kUnroll];                 // all id reads first: one LDS wait per batch
#pragma unroll
                        for (int u = 0; u < kUnroll; ++u) cid[u] = ids[min(c0 + u, n1 - 1)];
#pragma unroll
                        for (int u = 0; u < kUnroll; ++u) {
                            const bool want = (pp < np) && (u < per) && (c0 + u < n1);
                            v[hd][u] = tabT.load_entry(want, cid[u], col, FT * 4);
                        }
                    }
#pragma unroll
                    for (int hd = 0; hd < kHeads; ++hd)
#pragma unroll
                        for (int u = 0; u < kUnroll; ++u) {
                            if (DROP) {
                                const int pp = pp0 + hd * GN_T;
                                if (pp < np && u < per && c0 + u < n1)
                                    v[hd][u] = dk.apply(v[hd][u], 0u, (uint32_t)((j0 + pp) * n1 + c0 + u), (uint32_t)c, dscale);
                            }
                            acc[hd].x += v[hd][u].x; acc[hd].y += v[hd][u].y;
                            acc[hd].z += v[hd][u].z; acc[hd].w += v[hd][u].w;
                        }
                }
#pragma unroll
                for (int hd = 0; hd < kHeads; ++hd) {
                    const int pp = pp0 + hd * GN_T;
                    if (pp < 16) {
                        const float4 mean = make_float4(acc[hd].x / denom, acc[hd].y / denom, acc[hd].z / denom, acc[hd].w / denom);
                        *reinterpret_cast<float4 *>(A + pp * LDA + 4 * c) = mean;
                    }
                }
            }
        }
        wave_lds_fence();
        if (SAVE) {                          // the tile of neighbor means, out of LDS: np consecutive table rows
            float *dst = a.sv_x_mean[1 - side] + (size_t)(sv_h1row + j0) * FT;
            for (int i = sl; i < np * FTV; i += 64)
                stream_store(dst + 4 * i, *reinterpret_cast<const float4 *>(A + (i / FTV) * LDA + 4 * (i % FTV)));
            // The heads' node ids (their rows in the other type's tables) go out here and not where they are drawn: the
            // wait for a load counts the stores issued before it too, and the draws of hop 2 wait right behind those of
            // hop 1, while nothing waits for memory until the matrix products below are through.
            if (j0 == 0 && a.sv_self_ids[1 - side] && sl < n0)
                a.sv_self_ids[1 - side][sv_h1row + sl] = my_hop1;
        }
        f32x4 acc_s[NT1], acc_n[NT1];
#pragma unroll
        for (int ct = 0; ct < NT1; ++ct) { acc_s[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_n[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        mma_tile<FT, NT1>(A, LDA, W0n_O, wl, r, q, acc_n);
        wave_lds_fence();
#pragma unroll
        for (int pass = 0; pass < NPASS_O; ++pass) {
            const int rr = pass * RP_O + lane / FOV, c = lane % FOV;
            float4 as_head = own[pass], as_neigh = own[pass];
            if (DROP) {                      // the same rows enter twice, each time behind its own Dropout (models.py:152,165)
                as_head = dk.apply(own[pass], 1u, (uint32_t)(j0 + rr), (uint32_t)c, dscale);
                as_neigh = dk.apply(own[pass], 2u, (uint32_t)(j0 + rr), (uint32_t)c, dscale);
            }
            *reinterpret_cast<float4 *>(A + rr * LDA + 4 * c) = as_head;
            if (SAVE && a.sv_x_self[1 - side] && rr < np)
                stream_store(a.sv_x_self[1 - side] + (size_t)(sv_h1row + j0 + rr) * FO + 4 * c, as_head);
            rootsum.x += as_neigh.x; rootsum.y += as_neigh.y; rootsum.z += as_neigh.z; rootsum.w += as_neigh.w;
        }
        wave_lds_fence();
        mma_tile<FO, NT1>(A, LDA, W0s_O, wl, r, q, acc_s);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * q + i;
            if (row < np) {
                float *dst = H1t + (1 + j0 + row) * LDH;
#pragma unroll
                for (int ct = 0; ct < NT1; ++ct) {
                    const float vs = fmaxf(acc_s[ct][i] + b0_O[ct * 16 + r], 0.f);
                    const float vn = fmaxf(acc_n[ct][i] + b0_O[HH1 + ct * 16 + r], 0.f);
                    if (SUMROWS) {           // layer 1 reads only the mean of these rows: summed here, never stored
                        hsum_s[ct] += vs;
                        hsum_n[ct] += vn;
                    } else {
                        dst[ct * 16 + r] = vs;
                        dst[HH1 + ct * 16 + r] = vn;
                    }
                }
            }
        }
        wave_lds_fence();                    // A is rewritten by the next tile
    }

    // ---- layer 0, the root (type T; its neighbors are the hop-1 heads, type O) -------------
    {
        // column chunk c of the sum of the hop-1 rows: add the 64/FOV row groups
#pragma unroll
        for (int m = FOV; m < 64; m <<= 1) {
            rootsum.x += __shfl_xor(rootsum.x, m, 64); rootsum.y += __shfl_xor(rootsum.y, m, 64);
            rootsum.z += __shfl_xor(rootsum.z, m, 64); rootsum.w += __shfl_xor(rootsum.w, m, 64);
        }
        const float d0 = (float)n0;
        float4 own_root = make_float4(0.f, 0.f, 0.f, 0.f);
        {
            own_root = tabT.load(root >= 0 && lane < FTV, (unsigned)(root - offT), (unsigned)(lane % FTV) * 16u, FT * 4);
            if (DROP && lane < FTV) own_root = dk.apply(own_root, 3u, 0u, (uint32_t)lane, dscale);
        }
        if (lane < FOV) {
            const float4 mean = make_float4(rootsum.x / d0, rootsum.y / d0, rootsum.z / d0, rootsum.w / d0);
            *reinterpret_cast<float4 *>(A + 4 * lane) = mean;
            if (SAVE) stream_store(a.sv_x_mean[side] + (size_t)pair * FO + 4 * sl, mean);
        }
        if (SAVE && a.sv_x_self[side] && sl < FTV) stream_store(a.sv_x_self[side] + (size_t)pair * FT + 4 * sl, own_root);
        if (SAVE && a.sv_self_ids[side] && sl == 0) a.sv_self_ids[side][pair] = root;
        wave_lds_fence();
        f32x4 acc_s[NT1], acc_n[NT1];
#pragma unroll
        for (int ct = 0; ct < NT1; ++ct) { acc_s[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_n[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        mma_tile<FO, NT1>(A, LDA, W0n_T, wl, r, q, acc_n);
        wave_lds_fence();
        if (lane < FTV) *reinterpret_cast<float4 *>(A + 4 * lane) = own_root;
        wave_lds_fence();
        mma_tile<FT, NT1>(A, LDA, W0s_T, wl, r, q, acc_s);
        if (q == 0) {                        // row 0 of D lives in register 0 of lanes 0..15
#pragma unroll
            for (int ct = 0; ct < NT1; ++ct) {
                H1t[ct * 16 + r] = fmaxf(acc_s[ct][0] + b0_T[ct * 16 + r], 0.f);
                H1t[HH1 + ct * 16 + r] = fmaxf(acc_n[ct][0] + b0_T[HH1 + ct * 16 + r], 0.f);
            }
        }
        wave_lds_fence();
    }

    // ---- layer 1 on the LDS-resident activations (root only, type T) ---------------------
    if (DROP) {
        // Dropout of the layer-1 inputs, in place: row 0 (the root's activation, its own input) and rows 1 .. n0
        // (the hop-1 heads', its neighbors).  What is saved below are these values: relu keeps them >= 0, so
        // "> 0" afterwards says "active and kept", which is all the backward pass asks of them.
        for (int i = lane; i < (1 + n0) * H1V; i += 64) {
            float4 *pv = reinterpret_cast<float4 *>(H1t + (i / H1V) * LDH + 4 * (i % H1V));
            *pv = dk.apply(*pv, 4u, (uint32_t)(i / H1V), (uint32_t)(i % H1V), dscale);
        }
        wave_lds_fence();
    }
    {
        constexpr int G1 = 64 / H1V;
        const int g = lane / H1V, c = lane % H1V;
        const float d0 = (float)n0;
        if (SUMROWS) {
            // the four lane quarters hold the sums of different rows of the same columns
#pragma unroll
            for (int ct = 0; ct < NT1; ++ct) {
                hsum_s[ct] += __shfl_xor(hsum_s[ct], 16, 64); hsum_s[ct] += __shfl_xor(hsum_s[ct], 32, 64);
                hsum_n[ct] += __shfl_xor(hsum_n[ct], 16, 64); hsum_n[ct] += __shfl_xor(hsum_n[ct], 32, 64);
                if (q == 0) {
                    A[ct * 16 + r] = hsum_s[ct] / d0;
                    A[HH1 + ct * 16 + r] = hsum_n[ct] / d0;
                }
            }
        } else {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int j = g; j < n0; j += G1) {
                const float4 t = *reinterpret_cast<const float4 *>(H1t + (1 + j) * LDH + 4 * c);
                s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
            }
#pragma unroll
            for (int m = H1V; m < 64; m <<= 1) {
                s.x += __shfl_xor(s.x, m, 64); s.y += __shfl_xor(s.y, m, 64);
                s.z += __shfl_xor(s.z, m, 64); s.w += __shfl_xor(s.w, m, 64);
            }
            if (lane < H1V) {
                const float4 mean = make_float4(s.x / d0, s.y / d0, s.z / d0, s.w / d0);
                *reinterpret_cast<float4 *>(A + 4 * lane) = mean;
                if (SAVE) stream_store(a.sv_x_mean1[side] + (size_t)pair * H1 + 4 * sl, mean);
            }
        }
        if (SAVE) {
            // layer-0 activations: the root's row, then the hop-1 heads' rows as one contiguous block (LDH == H1)
            if (sl < H1V)
                stream_store(a.sv_h1[side] + (size_t)pair * H1 + 4 * sl, *reinterpret_cast<const float4 *>(H1t + 4 * sl));
            if (a.sv_h1_mask[1 - side]) {
                // the hop-1 heads' activations are read back only for their relu mask: one bit per element,
                // 64 / H1 rows per ballot (lane = row * H1 + column: the ballot's words are the rows' words).  Lane t
                // keeps the two words of ballot t, and all go out in two store instructions (one lane storing word
                // after word costs a store and its acknowledgement per word, which the next walk's first loads wait for).
                constexpr int RPB = 64 / H1;
                uint32_t *mw = a.sv_h1_mask[1 - side] + (size_t)pair * n0 * (H1 / 32);
                const int n_ballots = (n0 + RPB - 1) / RPB;  // (<= 64: n0 <= 64)
                uint32_t w_lo = 0, w_hi = 0;
                for (int t = 0; t < n_ballots; ++t) {
                    const int j = t * RPB + lane / H1;
                    const bool on = j < n0 && H1t[(1 + min(j, n0 - 1)) * LDH + lane % H1] > 0.f;
                    const unsigned long long bits = __ballot(on);
                    if (lane == t) { w_lo = (uint32_t)bits; w_hi = (uint32_t)(bits >> 32); }
                }
                if (sl < n_ballots) {
                    mw[2 * sl] = w_lo;
                    if (H1 == 64 || 2 * sl + 1 < n0) mw[2 * sl + 1] = w_hi;
                }
            } else {
                float *dst = a.sv_h1[1 - side] + (size_t)sv_h1row * H1;
                for (int i = sl; i < n0 * H1V; i += 64)
                    stream_store(dst + 4 * i, *reinterpret_cast<const float4 *>(H1t + LDH + 4 * i));
            }
        }
        wave_lds_fence();
        f32x4 acc_s[NT2], acc_n[NT2];
#pragma unroll
        for (int ct = 0; ct < NT2; ++ct) { acc_s[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_n[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        mma_tile<H1, NT2>(A, LDA, W1n_T, wl, r, q, acc_n);
        wave_lds_fence();
        if (lane < H1V) *reinterpret_cast<float4 *>(A + 4 * lane) = *reinterpret_cast<const float4 *>(H1t + 4 * lane);
        wave_lds_fence();
        mma_tile<H1, NT2>(A, LDA, W1s_T, wl, r, q, acc_s);
        float o[2][NT2];
        float ss = 0.f;
#pragma unroll
        for (int ct = 0; ct < NT2; ++ct) {
            o[0][ct] = acc_s[ct][0] + b1_T[ct * 16 + r];
            o[1][ct] = acc_n[ct][0] + b1_T[HH2 + ct * 16 + r];
            ss += o[0][ct] * o[0][ct] + o[1][ct] * o[1][ct];
        }
        if (SAVE && q == 0) {
            float *dz = a.sv_z[side] + (size_t)pair * H2 + (sl & 15);
#pragma unroll
            for (int ct = 0; ct < NT2; ++ct) { dz[ct * 16] = o[0][ct]; dz[HH2 + ct * 16] = o[1][ct]; }
        }
        if (a.normalize) {
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) ss += __shfl_xor(ss, m, 64);
            const float inv = rsqrtf(fmaxf(ss, 1e-12f));
#pragma unroll
            for (int ct = 0; ct < NT2; ++ct) { o[0][ct] *= inv; o[1][ct] *= inv; }
        }
        if (q == 0) {
            float *dst = a.emb + (size_t)w * H2;
#pragma unroll
            for (int ct = 0; ct < NT2; ++ct) {
                dst[ct * 16 + r] = o[0][ct];
                dst[HH2 + ct * 16 + r] = o[1][ct];
            }
        }
    }
}

// Neighbor means of a tile of up to 16 heads into A: KH heads per lane group and iteration, up to KU children
// each, their row loads all issued before the first add (what keeps enough bytes in flight per wave).
template <int FT, int LDA, bool BUF, int KH, int KU>
__device__ __forceinline__ void tile_means(const RowTable<BUF> &tabT, const int32_t *hop2, int j0, int np, int n1,
                                           int lane, float *A) {
    constexpr int FTV = FT / 4, GN_T = 64 / FTV;
    static_assert(KH * GN_T <= 16, "more heads per iteration than a tile has");
    const int g = lane / FTV, c = lane % FTV;
    const unsigned col = (unsigned)c * 16u;
    const float denom = (float)n1;
    const int rounds = (n1 + KU - 1) / KU;
    const int per = (n1 + rounds - 1) / rounds;
    for (int pp0 = g; pp0 < 16; pp0 += KH * GN_T) {
        float4 acc[KH];
#pragma unroll
        for (int hd = 0; hd < KH; ++hd) acc[hd] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int c0 = 0; c0 < n1; c0 += per) {
            float4 v[KH][KU];
#pragma unroll
            for (int hd = 0; hd < KH; ++hd) {
                const int pp = pp0 + hd * GN_T;
                const int32_t *ids = hop2 + (j0 + min(pp, np - 1)) * n1;
                int32_t cid[KU];                          // all id reads first: one LDS wait per batch
#pragma unroll
                for (int u = 0; u < KU; ++u) cid[u] = ids[min(c0 + u, n1 - 1)];
#pragma unroll
                for (int u = 0; u < KU; ++u) {
                    const bool want = (pp < np) && (u < per) && (c0 + u < n1);
                    v[hd][u] = tabT.load_entry(want, cid[u], col, FT * 4);
                }
            }
#pragma unroll
            for (int hd = 0; hd < KH; ++hd)
#pragma unroll
                for (int u = 0; u < KU; ++u) {
                    acc[hd].x += v[hd][u].x; acc[hd].y += v[hd][u].y;
                    acc[hd].z += v[hd][u].z; acc[hd].w += v[hd][u].w;
                }
        }
#pragma unroll
        for (int hd = 0; hd < KH; ++hd) {
            const int pp = pp0 + hd * GN_T;
            if (pp < 16)
                *reinterpret_cast<float4 *>(A + pp * LDA + 4 * c) =
                    make_float4(acc[hd].x / denom, acc[hd].y / denom, acc[hd].z / denom, acc[hd].w / denom);
        }
    }
}

// Subtree mode, G roots per wavefront.  A subtree of fan-outs [10, 5] is 61 rows behind the same chain of
// dependent reads as a whole walk; taking G subtrees through that chain together -- their hop-1 heads as ONE
// list of G * n0 heads, the G roots as G rows of one tile -- amortises it the way a walk of fan-outs [25, 10]
// does.  Roots are consecutive entries of the hop-1 array (they may belong to different seed walks: walk id
// and slot offsets are per root).  The arithmetic of walk_body one hop down (sampler hops 1 and 2, slots shifted by
// the root's position below its seed, relu after layer 1, no normalisation); the neighbor sum of a root is taken
// over its heads in list order.
template <int FT, int FO, int HH1, int HH2, int LDA, int LDH, bool BUF, int G, bool SAVE>
__device__ __forceinline__ void group_body(const Sage2Args &a, long long w0, int n_valid, int side, int lane,
                                           int32_t rootv, int64_t rav, int64_t rbv, unsigned long long walkv,
                                           float *A, float *H1t, int32_t *s_ids,
                                           const float *W0s_T, const float *W0n_T, const float *W0s_O, const float *W0n_O,
                                           const float *W1s_T, const float *W1n_T,
                                           const float *b0_T, const float *b0_O, const float *b1_T) {
    constexpr int NT1 = HH1 / 16, NT2 = HH2 / 16, H1 = 2 * HH1, H2 = 2 * HH2;
    constexpr int FTV = FT / 4, FOV = FO / 4, H1V = H1 / 4;
    static_assert(!SAVE || H1 == 32, "the saved relu masks are one word per row");
    constexpr int RS_N = (G * FOV + 63) / 64, RT_N = (G * FTV + 63) / 64;
    constexpr int L1_N = (G * H1V + 63) / 64;
    static_assert(G >= 1 && G <= 16, "the roots are rows 0 .. G-1 of one 16-row tile");
    const int r = lane & 15, q = lane >> 4;
    const int n0 = a.n0, n1 = a.n1, NH = G * n0;
    RowTable<BUF> tabT, tabO;
    tabT.init(a.feat[side], a.n_rows[side], FT * 4);
    tabO.init(a.feat[1 - side], a.n_rows[1 - side], FO * 4);
    const int offT = side ? (int)a.n_rows[0] : 0, offO = side ? 0 : (int)a.n_rows[0];
    int32_t *hop1 = s_ids, *hop2 = s_ids + NH;
    int wl = lane;
    asm volatile("" : "+v"(wl));            // keeps the weight-fragment reads inside the group loop
    // The lane number as the stores to memory see it.  In the saving form a value of this group, so that no table's
    // per-lane address is computed once and kept through the gathers (fifteen spilled registers otherwise); the plain
    // form has few stores and is 1.5 % faster without.
    int sl = lane;
    if constexpr (SAVE) asm volatile("" : "+v"(sl));
    // saving: the group's first root within its node type, and the heads that belong to roots that exist
    const long long sv_root0 = w0 - (side ? 0 : a.n_pairs);
    const int nh_valid = n_valid * n0;

    // lane g < G carries root g: id (-1 / -2: none), CSR row bounds, walk id; position below its seed walk
    const long long my_w = w0 + min(lane, n_valid - 1);          // (lanes past the valid roots repeat the last one)
    const long long my_seed = my_w / a.sub_n;
    const unsigned my_j = (unsigned)(my_w - my_seed * a.sub_n);
    if (lane < n_valid && rootv < -1) atomicMax(&a.err[1], 1);
    if (lane < n_valid && rootv >= 0 && rbv == rav) atomicMax(&a.err[0], rootv + 1);
    // the roots' own rows: requested now, used after the heads' tiles
    float4 own_root[RT_N];
#pragma unroll
    for (int k = 0; k < RT_N; ++k) {
        const int idx = k * 64 + lane, g = min(idx / FTV, G - 1), c = idx % FTV;
        const int32_t rg = __shfl(rootv, g, 64);
        own_root[k] = tabT.load(idx < G * FTV && g < n_valid && rg >= 0, (unsigned)(rg - offT), (unsigned)c * 16u, FT * 4);
    }

    // ---- sampling ---------------------------------------------------------------
    wave_lds_fence();                        // the previous group's readers of s_ids / A / H1t are done
    for (int t0 = 0; t0 < NH; t0 += 64) {    // (whole-wave trips: the shuffles below read lanes 0 .. G-1)
        const int t = min(t0 + sl, NH - 1);
        const int g = t / n0, tl = t - g * n0;
        const int32_t rg = __shfl(rootv, g, 64);
        const int64_t ra = __shfl(rav, g, 64), rb = __shfl(rbv, g, 64);
        const unsigned long long wk = __shfl(walkv, g, 64);
        const long long sw = __shfl(my_seed, g, 64);
        const unsigned slot = __shfl(my_j, g, 64) * (unsigned)n0 + (unsigned)tl;
        const uint32_t deg = (uint32_t)(rb - ra);
        const bool live = g < n_valid && rg >= 0 && deg != 0;
        const uint32_t rr = a.replay ? a.replay[(size_t)sw * a.tape_len + a.tape_off0 + slot]
                                     : gs::philox4x32_10_x((unsigned)wk, (unsigned)(wk >> 32), 1u, slot, a.key0, a.key1);
        int32_t v = a.indices[live ? ra + (int64_t)__umulhi(rr, deg) : 0];
        if (!live) v = -1;
        if (t0 + sl < NH) {
            hop1[t] = v;
            if (a.out_hop1 && g < n_valid) a.out_hop1[(size_t)(w0 + g) * n0 + tl] = v;
        }
    }
    wave_lds_fence();
    {
        const int total = NH * n1, per_root = n0 * n1;
        for (int base = 0; base < total; base += kChunks * 64) {
            int32_t p[kChunks];
            int64_t ca[kChunks], cb[kChunks];
#pragma unroll
            for (int c = 0; c < kChunks; ++c) {
                const int t = base + c * 64 + lane;
                p[c] = t < total ? hop1[t / n1] : -1;
                const int32_t pc = p[c] >= 0 ? p[c] : 0;
                ca[c] = a.indptr[pc];
                cb[c] = a.indptr[pc + 1];
            }
            int32_t v[kChunks];
#pragma unroll
            for (int c = 0; c < kChunks; ++c) {
                const int t = min(base + c * 64 + lane, total - 1);
                const int g = t / per_root, local = t - g * per_root;
                const unsigned long long wk = __shfl(walkv, g, 64);
                const long long sw = __shfl(my_seed, g, 64);
                const unsigned slot = __shfl(my_j, g, 64) * (unsigned)per_root + (unsigned)local;
                const uint32_t deg = (uint32_t)(cb[c] - ca[c]);
                const uint32_t rr = a.replay ? a.replay[(size_t)sw * a.tape_len + a.tape_off1 + slot]
                                             : gs::philox4x32_10_x((unsigned)wk, (unsigned)(wk >> 32), 2u, slot, a.key0, a.key1);
                const bool live = p[c] >= 0 && deg != 0;
                if (p[c] >= 0 && deg == 0) atomicMax(&a.err[0], p[c] + 1);
                v[c] = a.indices[live ? ca[c] + (int64_t)__umulhi(rr, deg) : 0];
                if (!live) v[c] = -1;
            }
#pragma unroll
            for (int c = 0; c < kChunks; ++c) {
                const int t = base + c * 64 + sl;
                if (t < total) {
                    const int g = t / per_root, local = t - g * per_root;
                    hop2[t] = tabT.entry(v[c], offT, FT * 4);
                    if (a.out_hop2 && g < n_valid) a.out_hop2[(size_t)(w0 + g) * per_root + local] = v[c];
                }
            }
        }
    }
    wave_lds_fence();

    // ---- layer 0, the heads (type O; children type T) ------------------------------------
    float4 rs[RS_N];                         // lane (g, c): column chunk c of the sum of root g's heads' own rows
    float4 hs[L1_N];                         // lane (g, c): the same of root g's heads' layer-0 outputs
#pragma unroll
    for (int k = 0; k < RS_N; ++k) rs[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < L1_N; ++k) hs[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int RP_O = 64 / FOV, NPASS_O = 16 / RP_O;     // own rows per load instruction / passes per tile
    constexpr int GN_T = 64 / FTV;                           // lane groups summing children rows
    for (int j0 = 0; j0 < NH; j0 += 16) {
        const int np = min(16, NH - j0);
        float4 own[NPASS_O];
#pragma unroll
        for (int pass = 0; pass < NPASS_O; ++pass) {
            const int rr = pass * RP_O + lane / FOV, c = lane % FOV;
            const int32_t id = hop1[j0 + min(rr, np - 1)];
            own[pass] = tabO.load(rr < np, (unsigned)(id - offO), (unsigned)c * 16u, FO * 4);
        }
        // short child lists (at most kUnroll / 2 per head): twice the heads per iteration instead, the same
        // number of row loads in flight per lane
        constexpr int KH = heads_in_flight(HH1);
        constexpr bool kShort = 2 * KH * GN_T <= 16;      // (a tile has the heads for it)
        if (kShort && 2 * n1 <= kUnroll) tile_means<FT, LDA, BUF, kShort ? 2 * KH : KH, kUnroll / 2>(tabT, hop2, j0, np, n1, lane, A);
        else tile_means<FT, LDA, BUF, KH, kUnroll>(tabT, hop2, j0, np, n1, lane, A);
        wave_lds_fence();
        const int nv = min(np, nh_valid - j0);                   // rows of this tile the saved tables have (may be < 1)
        if (SAVE) {                          // the tile of neighbor means, out of LDS: nv consecutive table rows
            float *dst = a.sv_x_mean[1 - side] + (size_t)(sv_root0 * n0 + j0) * FT;
            for (int i = sl; i < nv * FTV; i += 64)
                stream_store(dst + 4 * i, *reinterpret_cast<const float4 *>(A + (i / FTV) * LDA + 4 * (i % FTV)));
        }
        f32x4 acc_s[NT1], acc_n[NT1];
#pragma unroll
        for (int ct = 0; ct < NT1; ++ct) { acc_s[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_n[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        mma_tile<FT, NT1>(A, LDA, W0n_O, wl, r, q, acc_n);
        wave_lds_fence();
#pragma unroll
        for (int pass = 0; pass < NPASS_O; ++pass) {
            const int rr = pass * RP_O + lane / FOV, c = lane % FOV;
            *reinterpret_cast<float4 *>(A + rr * LDA + 4 * c) = own[pass];
        }
        wave_lds_fence();
        mma_tile<FO, NT1>(A, LDA, W0s_O, wl, r, q, acc_s);
        // the tile's own rows (still in A) summed per root: they are the roots' layer-0 neighbors
#pragma unroll
        for (int k = 0; k < RS_N; ++k) {
            const int idx = k * 64 + lane, g = idx / FOV, c = idx % FOV;
            if (idx < G * FOV) {
                const int lo = max(j0, g * n0) - j0, hi = min(j0 + np, (g + 1) * n0) - j0;
                for (int rr = lo; rr < hi; ++rr) {
                    const float4 t = *reinterpret_cast<const float4 *>(A + rr * LDA + 4 * c);
                    rs[k].x += t.x; rs[k].y += t.y; rs[k].z += t.z; rs[k].w += t.w;
                }
            }
        }
        wave_lds_fence();
        // the tile's activations go through A (the operands are used up) into the per-root sums: layer 1 needs
        // only the mean of a root's heads, so no head's row outlives its tile
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float *dst = A + (4 * q + i) * LDA;
#pragma unroll
            for (int ct = 0; ct < NT1; ++ct) {
                dst[ct * 16 + r] = fmaxf(acc_s[ct][i] + b0_O[ct * 16 + r], 0.f);
                dst[HH1 + ct * 16 + r] = fmaxf(acc_n[ct][i] + b0_O[HH1 + ct * 16 + r], 0.f);
            }
        }
        wave_lds_fence();
        if constexpr (SAVE) {
            // the heads' rows are read back only for their relu masks: one word per row (bit c: column c > 0), two rows
            // per ballot; lane t keeps the words of rows 2t and 2t + 1
            uint32_t w_lo = 0, w_hi = 0;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const unsigned long long bits = __ballot(A[(2 * t + (lane >> 5)) * LDA + (lane & 31)] > 0.f);
                if (lane == t) { w_lo = (uint32_t)bits; w_hi = (uint32_t)(bits >> 32); }
            }
            uint32_t *mw = a.sv_h1_mask[1 - side] + (size_t)(sv_root0 * n0 + j0) + 2 * sl;      // (8-byte aligned: kGroup and 16 are even)
            if (sl < 8) {
                if (2 * sl + 1 < nv) *reinterpret_cast<uint2 *>(mw) = make_uint2(w_lo, w_hi);
                else if (2 * sl < nv) *mw = w_lo;
            }
        }
#pragma unroll
        for (int k = 0; k < L1_N; ++k) {
            const int idx = k * 64 + lane, g = idx / H1V, c = idx % H1V;
            if (idx < G * H1V) {
                const int lo = max(j0, g * n0) - j0, hi = min(j0 + np, (g + 1) * n0) - j0;
                for (int rr = lo; rr < hi; ++rr) {
                    const float4 t = *reinterpret_cast<const float4 *>(A + rr * LDA + 4 * c);
                    hs[k].x += t.x; hs[k].y += t.y; hs[k].z += t.z; hs[k].w += t.w;
                }
            }
        }
        wave_lds_fence();                    // A is rewritten by the next tile
    }


    // ---- layer 0, the roots (type T): rows 0 .. G-1 of one tile --------------------------------
    {
        const float d0 = (float)n0;
#pragma unroll
        for (int k = 0; k < RS_N; ++k) {
            const int idx = k * 64 + sl;
            if (idx < G * FOV) {
                const float4 mean = make_float4(rs[k].x / d0, rs[k].y / d0, rs[k].z / d0, rs[k].w / d0);
                *reinterpret_cast<float4 *>(A + (idx / FOV) * LDA + 4 * (idx % FOV)) = mean;
                if (SAVE && idx / FOV < n_valid)
                    stream_store(a.sv_root_mean0[side] + (size_t)(sv_root0 + idx / FOV) * FO + 4 * (idx % FOV), mean);
            }
        }
        wave_lds_fence();
        f32x4 acc_s[NT1], acc_n[NT1];
#pragma unroll
        for (int ct = 0; ct < NT1; ++ct) { acc_s[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_n[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        mma_tile<FO, NT1>(A, LDA, W0n_T, wl, r, q, acc_n);
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < RT_N; ++k) {
            const int idx = k * 64 + lane;
            if (idx < G * FTV) *reinterpret_cast<float4 *>(A + (idx / FTV) * LDA + 4 * (idx % FTV)) = own_root[k];
        }
        wave_lds_fence();
        mma_tile<FT, NT1>(A, LDA, W0s_T, wl, r, q, acc_s);
#pragma unroll
        for (int i = 0; i < 4; ++i) {        // row 4q + i of D lives in register i of lane quarter q
            const int row = 4 * q + i;
            if (row < G) {
#pragma unroll
                for (int ct = 0; ct < NT1; ++ct) {
                    H1t[row * LDH + ct * 16 + r] = fmaxf(acc_s[ct][i] + b0_T[ct * 16 + r], 0.f);
                    H1t[row * LDH + HH1 + ct * 16 + r] = fmaxf(acc_n[ct][i] + b0_T[HH1 + ct * 16 + r], 0.f);
                }
            }
        }
        wave_lds_fence();
    }

    // ---- layer 1 at the roots, on the LDS-resident activations ----------------------------------
    {
        const float d0 = (float)n0;
#pragma unroll
        for (int k = 0; k < L1_N; ++k) {     // lane (g, c): column chunk c of the mean of root g's heads' activations
            const int idx = k * 64 + sl;
            if (idx < G * H1V) {
                const int g = idx / H1V, c = idx % H1V;
                const float4 mean = make_float4(hs[k].x / d0, hs[k].y / d0, hs[k].z / d0, hs[k].w / d0);
                *reinterpret_cast<float4 *>(A + g * LDA + 4 * c) = mean;
                if (SAVE && g < n_valid) stream_store(a.sv_root_mean1[side] + (size_t)(sv_root0 + g) * H1 + 4 * c, mean);
                if (a.out_h0 && g < n_valid)     // the root's layer-0 output (the layer above reads it as a neighbor's)
                    *reinterpret_cast<float4 *>(a.out_h0 + (size_t)(w0 + g) * H1 + 4 * c) = *reinterpret_cast<const float4 *>(H1t + g * LDH + 4 * c);
            }
        }
        wave_lds_fence();
        f32x4 acc_s[NT2], acc_n[NT2];
#pragma unroll
        for (int ct = 0; ct < NT2; ++ct) { acc_s[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_n[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        mma_tile<H1, NT2>(A, LDA, W1n_T, wl, r, q, acc_n);
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < L1_N; ++k) {
            const int idx = k * 64 + lane;
            if (idx < G * H1V)
                *reinterpret_cast<float4 *>(A + (idx / H1V) * LDA + 4 * (idx % H1V)) =
                    *reinterpret_cast<const float4 *>(H1t + (idx / H1V) * LDH + 4 * (idx % H1V));
        }
        wave_lds_fence();
        mma_tile<H1, NT2>(A, LDA, W1s_T, wl, r, q, acc_s);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * q + i;
            if (row < n_valid) {
                float *dst = a.emb + (size_t)(w0 + row) * H2 + (sl & 15);
#pragma unroll
                for (int ct = 0; ct < NT2; ++ct) {           // an inner layer: through relu, no normalisation
                    dst[ct * 16] = fmaxf(acc_s[ct][i] + b1_T[ct * 16 + r], 0.f);
                    dst[HH2 + ct * 16] = fmaxf(acc_n[ct][i] + b1_T[HH2 + ct * 16 + r], 0.f);
                }
            }
        }
    }
}

template <int FU, int FI, int HH1, int HH2, bool BUF, bool SAVE, bool SUB, bool DROP = false>
__global__ __launch_bounds__(block_waves(HH1, SAVE, SUB, FU > FI ? FU : FI, DROP) * 64)
void sage2_kernel(const Sage2Args a) {
    using S = Sage2Shape<FU, FI, HH1, HH2>;
    constexpr int H1 = S::H1, H2 = S::H2, LDA = S::LDA, LDH = S::LDH;
    extern __shared__ float lds[];
    // Even blocks take the walks rooted at user heads, odd blocks those rooted at item heads, so
    // a block needs one layer-1 weight set.  Weights in fragment order: layer 0 [user self, user
    // neigh, item self, item neigh], layer 1 [self, neigh] of the block's root type.
    const int side = blockIdx.x & 1;
    float *w0s_u = lds;
    float *w0n_u = w0s_u + FU * HH1;
    float *w0s_i = w0n_u + FI * HH1;
    float *w0n_i = w0s_i + FI * HH1;
    float *w1s = w0n_i + FU * HH1;
    float *w1n = w1s + H1 * HH2;
    float *bias = w1n + H1 * HH2;            // b0_u[H1] b0_i[H1] b1[H2]
    float *wave0 = bias + 2 * H1 + H2;
    // the wave index as a scalar: everything derived from it (walk number, root, LDS bases, output
    // rows) then lives in scalar registers
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int per_wave = 16 * LDA + a.h1_rows * LDH + a.ids_per_wave;
    float *A = wave0 + (size_t)wave * per_wave;
    float *H1t = A + 16 * LDA;
    int32_t *s_ids = reinterpret_cast<int32_t *>(H1t + a.h1_rows * LDH);
    const int n_waves = blockDim.x >> 6;
    const long long my_blocks = (gridDim.x + 1 - side) >> 1;       // blocks serving this side
    const long long stride = my_blocks * n_waves;
    // whole walks: the user roots first; subtrees: the item-typed roots first
    const long long base = (SUB ? !side : side) ? a.n_pairs : 0;
    auto load_root = [&](long long p) -> int32_t {                 // -1 for ids outside the graph
        const int32_t v = a.seeds[base + p];
        if ((unsigned long long)(long long)v < (unsigned long long)a.n_nodes) return v;
        return SUB && v != -1 ? -2 : -1;
    };
    long long p = (long long)(blockIdx.x >> 1) * n_waves + wave;
    const bool has_work = p < a.n_pairs;
    // the root of a walk and its CSR row bounds are fetched one walk ahead; for the first walk that
    // is here, so the two dependent reads are in flight while the block stages the weights
    int32_t root = (!SUB && has_work) ? load_root(p) : -1;
    int64_t ra = a.indptr[root >= 0 ? root : 0], rb = a.indptr[(root >= 0 ? root : 0) + 1];

    stage_weight(w0s_u, a.w0_self[0], FU, HH1);
    stage_weight(w0n_u, a.w0_neigh[0], FI, HH1);
    stage_weight(w0s_i, a.w0_self[1], FI, HH1);
    stage_weight(w0n_i, a.w0_neigh[1], FU, HH1);
    stage_weight(w1s, a.w1_self[side], H1, HH2);
    stage_weight(w1n, a.w1_neigh[side], H1, HH2);
    for (int i = threadIdx.x; i < 2 * H1 + H2; i += blockDim.x) {
        float v = 0.f;
        if (i < H1) { if (a.b0[0]) v = a.b0[0][i]; }
        else if (i < 2 * H1) { if (a.b0[1]) v = a.b0[1][i - H1]; }
        else { if (a.b1[side]) v = a.b1[side][i - 2 * H1]; }
        bias[i] = v;
    }
    __syncthreads();
    if (!has_work) return;

    if constexpr (SUB) {
        // groups of kGroup consecutive roots; lane g < kGroup fetches root g's id, row bounds and walk id, one
        // group ahead
        const long long n_groups = (a.n_pairs + kGroup - 1) / kGroup;
        auto fetch = [&](long long grp, int32_t &rv, int64_t &rav, int64_t &rbv, unsigned long long &wkv) {
            const long long pr = grp * kGroup + min(lane, kGroup - 1);
            const bool mine = grp < n_groups && pr < a.n_pairs;
            const long long gi = base + min(pr, a.n_pairs - 1);
            const long long seed = gi / a.sub_n;
            wkv = a.walk_ids ? (unsigned long long)a.walk_ids[seed] : a.walk_base + (unsigned long long)seed;
            if (a.seeds0) {
                // the root is draw number gi % sub_n of its seed walk's first hop, as the sampler makes it
                const unsigned slot = (unsigned)(gi - seed * a.sub_n);
                int32_t sd = mine ? a.seeds0[seed] : -1;
                if (mine && (unsigned long long)(long long)sd >= (unsigned long long)a.n_nodes) {
                    if (lane < kGroup) atomicMax(&a.err[1], 1);
                    sd = -1;
                }
                const int32_t sc = sd >= 0 ? sd : 0;
                const int64_t sa = a.indptr[sc], sb = a.indptr[sc + 1];
                const uint32_t deg = (uint32_t)(sb - sa);
                const uint32_t rr = a.replay ? a.replay[(size_t)seed * a.tape_len + slot]
                                             : gs::philox4x32_10_x((unsigned)wkv, (unsigned)(wkv >> 32), 0u, slot, a.key0, a.key1);
                const bool live = sd >= 0 && deg != 0;
                if (sd >= 0 && deg == 0 && lane < kGroup) atomicMax(&a.err[0], sd + 1);
                rv = a.indices[live ? sa + (int64_t)__umulhi(rr, deg) : 0];
                if (!live) rv = -1;
                if (mine && lane < kGroup) a.hop1_out[gi] = rv;
            } else {
                rv = mine ? load_root(pr) : -1;
            }
            const int32_t rc = rv >= 0 ? rv : 0;
            rav = a.indptr[rc];
            rbv = a.indptr[rc + 1];
        };
        long long grp = (long long)(blockIdx.x >> 1) * n_waves + wave;
        int32_t rv; int64_t rav, rbv; unsigned long long wkv;
        fetch(grp, rv, rav, rbv, wkv);
        for (; grp < n_groups; grp += stride) {
            int32_t nrv; int64_t nrav, nrbv; unsigned long long nwkv;
            fetch(grp + stride, nrv, nrav, nrbv, nwkv);
            const int n_valid = (int)min((long long)kGroup, a.n_pairs - grp * kGroup);
            if (side == 0)
                group_body<FU, FI, HH1, HH2, LDA, LDH, BUF, kGroup, SAVE>(a, base + grp * kGroup, n_valid, 0, lane, rv, rav, rbv, wkv, A, H1t, s_ids,
                                                                    w0s_u, w0n_u, w0s_i, w0n_i, w1s, w1n, bias, bias + H1, bias + 2 * H1);
            else
                group_body<FI, FU, HH1, HH2, LDA, LDH, BUF, kGroup, SAVE>(a, base + grp * kGroup, n_valid, 1, lane, rv, rav, rbv, wkv, A, H1t, s_ids,
                                                                    w0s_i, w0n_i, w0s_u, w0n_u, w1s, w1n, bias + H1, bias, bias + 2 * H1);
            rv = nrv; rav = nrav; rbv = nrbv; wkv = nwkv;
        }
        return;
    }

    for (; p < a.n_pairs; p += stride) {
        const long long pn = p + stride;
        const int32_t next_root = pn < a.n_pairs ? load_root(pn) : -1;
        int64_t nra = 0, nrb = 0;
        if (side == 0)
            walk_body<FU, FI, HH1, HH2, LDA, LDH, BUF, SAVE, DROP>(a, base + p, 0, lane, root, ra, rb, next_root, nra, nrb, A, H1t, s_ids,
                                                  w0s_u, w0n_u, w0s_i, w0n_i, w1s, w1n, bias, bias + H1, bias + 2 * H1);
        else
            walk_body<FI, FU, HH1, HH2, LDA, LDH, BUF, SAVE, DROP>(a, base + p, 1, lane, root, ra, rb, next_root, nra, nrb, A, H1t, s_ids,
                                                  w0s_i, w0n_i, w0s_u, w0n_u, w1s, w1n, bias + H1, bias, bias + 2 * H1);
        root = next_root; ra = nra; rb = nrb;
    }
}

template <int FU, int FI, int HH1, int HH2, bool BUF, bool SAVE, bool SUB = false, bool DROP = false>
int launch_one(const Sage2Args &a, hipStream_t stream) {
    using S = Sage2Shape<FU, FI, HH1, HH2>;
    constexpr int roots = SUB ? kGroup : 0;
    constexpr bool summed = sum_rows(HH1, SAVE, SUB);
    int waves = block_waves(HH1, SAVE, SUB, S::FMAX, DROP);                 // as many waves per CU as the LDS budget admits
    while (waves > 4 && S::lds_bytes(a.n0, a.n1, waves, roots, summed) > kLdsMax) waves -= 2;
    const size_t lds = S::lds_bytes(a.n0, a.n1, waves, roots, summed);
    if (lds > kLdsMax) return gs::fail(GS_ELIMIT, "gs_sage2_forward: %zu B of LDS per block needed (limit %zu)", lds, kLdsMax);
    auto kern = sage2_kernel<FU, FI, HH1, HH2, BUF, SAVE, SUB, DROP>;
    if (lds > 64 * 1024) {
        hipError_t e = gs::raise_lds_limit((const void *)kern, (int)kLdsMax);
        if (e != hipSuccess) return gs::fail(GS_ELAUNCH, "gs_sage2_forward: cannot raise the LDS limit: %s", hipGetErrorString(e));
    }
    long long blocks = 2 * ((a.n_pairs + waves - 1) / waves);   // even blocks: user roots, odd: item roots
    if (blocks > 256) blocks = 256;                      // one block per CU, waves loop over walks
    Sage2Args b = a;
    if (summed) b.h1_rows = 1;
    gs::launch_timed(kern, dim3((unsigned)blocks), dim3(waves * 64), lds, stream, b);
    return gs::check_launch("gs_sage2_forward");
}

template <int FU, int FI, int HH1, int HH2>
int launch(const Sage2Args &a, hipStream_t stream) {
    // buffer addressing needs both tables inside a 4 GiB window
    const long long lim = 0xFFFFF000LL;                  // RowTable::kNoRow must lie past the end of both tables
    const bool buf = a.n_rows[0] * FU * 4 < lim && a.n_rows[1] * FI * 4 < lim;
    if constexpr (HH1 == 16) {
        if (a.sub_n && a.sv_root_mean0[0])
            return buf ? launch_one<FU, FI, HH1, HH2, true, true, true>(a, stream) : launch_one<FU, FI, HH1, HH2, false, true, true>(a, stream);
    }
    if (a.sub_n) return buf ? launch_one<FU, FI, HH1, HH2, true, false, true>(a, stream) : launch_one<FU, FI, HH1, HH2, false, false, true>(a, stream);
    if (a.sv_z[0] && a.drop_thr)
        return buf ? launch_one<FU, FI, HH1, HH2, true, true, false, true>(a, stream) : launch_one<FU, FI, HH1, HH2, false, true, false, true>(a, stream);
    if (a.sv_z[0]) return buf ? launch_one<FU, FI, HH1, HH2, true, true>(a, stream) : launch_one<FU, FI, HH1, HH2, false, true>(a, stream);
    return buf ? launch_one<FU, FI, HH1, HH2, true, false>(a, stream) : launch_one<FU, FI, HH1, HH2, false, false>(a, stream);
}

// LDS of the smallest block this kernel runs with (4 waves)
inline size_t sage2_lds(int Fu, int Fi, int H1, int H2, int n0, int n1, int roots = 0) {
    const int fmax = Fu > Fi ? Fu : Fi, kmax = fmax > H1 ? fmax : H1;
    const int ids = ((roots ? roots : 1) * (n0 + n0 * n1) + 3) & ~3;
    return (size_t)((Fu + Fi) * H1 + H1 * H2 + 2 * H1 + H2 +
                    4 * (16 * (kmax + 4) + (roots ? roots : 1 + n0) * H1 + ids)) * sizeof(float);
}


}  // namespace

extern "C" int gs_sage2_supported(int Fu, int Fi, int H1, int H2, int n0, int n1) {
    if (n0 < 1 || n1 < 1 || n0 > 64 || (long long)n0 * n1 > 4096 || (H1 & 1) || (H2 & 1)) return 0;
    bool found = false;
#define GS_X(fu, fi, a1, a2) found = found || (Fu == fu && Fi == fi && H1 == 2 * a1 && H2 == 2 * a2);
    GS_SAGE2_CASES(GS_X)
#undef GS_X
    return found && sage2_lds(Fu, Fi, H1, H2, n0, n1) <= kLdsMax;
}

extern "C" int gs_sage2_subtrees_supported(int Fu, int Fi, int H1, int H2, int n1, int n2) {
    return gs_sage2_supported(Fu, Fi, H1, H2, n1, n2) && sage2_lds(Fu, Fi, H1, H2, n1, n2, kGroup) <= kLdsMax;
}

namespace {

int sage2_forward_impl(const int64_t *indptr, const int32_t *indices, int64_t n_nodes,
                       const float *feat_user, int64_t n_users, int Fu,
                       const float *feat_item, int64_t n_items, int Fi,
                       const int32_t *seeds, int64_t n_pairs, int n0, int n1,
                       uint64_t rng_seed, uint64_t walk_id_base, const int64_t *walk_ids,
                       const uint32_t *replay,
                       const gs_sage_weights *l0_user, const gs_sage_weights *l0_item,
                       const gs_sage_weights *l1_user, const gs_sage_weights *l1_item,
                       int H1, int H2, int normalize,
                       float *emb, int32_t *out_hop1, int32_t *out_hop2,
                       const gs_sage2_saved *saved, const gs_sage2_subtrees *sub, int32_t *err_flag, void *stream) {

    if (n_pairs < 0) return gs::fail(GS_EINVAL, "gs_sage2_forward: n_pairs=%lld", (long long)n_pairs);
    if (n_pairs == 0) return GS_OK;
    if (!indptr || !indices || !feat_user || !feat_item || (!seeds && !(sub && sub->seeds)) || !emb || !err_flag ||
        !l0_user || !l0_item || !l1_user || !l1_item)
        return gs::fail(GS_EINVAL, "gs_sage2_forward: null pointer argument");
    const gs_sage_weights *ws[4] = {l0_user, l0_item, l1_user, l1_item};
    for (int i = 0; i < 4; ++i)
        if (!ws[i]->w_self || !ws[i]->w_neigh) return gs::fail(GS_EINVAL, "gs_sage2_forward: weight set %d has a null matrix", i);
    if (n_users < 1 || n_items < 1 || n_nodes != n_users + n_items)
        return gs::fail(GS_EINVAL, "gs_sage2_forward: n_nodes=%lld must equal n_users + n_items = %lld + %lld",
                        (long long)n_nodes, (long long)n_users, (long long)n_items);
    if (!aligned16(feat_user) || !aligned16(feat_item))
        return gs::fail(GS_EINVAL, "gs_sage2_forward: feature tables must be 16-byte aligned");
    if (!gs_sage2_supported(Fu, Fi, H1, H2, n0, n1))
        return gs::fail(GS_ELIMIT, "gs_sage2_forward: shape Fu=%d Fi=%d H1=%d H2=%d fan-outs [%d,%d] is outside this kernel's "
                                   "range; use gs_sample + gs_sage_fused_fwd", Fu, Fi, H1, H2, n0, n1);
    Sage2Args a{};
    a.indptr = indptr; a.indices = indices; a.n_nodes = n_nodes;
    a.feat[0] = feat_user; a.feat[1] = feat_item; a.n_rows[0] = n_users; a.n_rows[1] = n_items;
    a.seeds = seeds; a.n_pairs = n_pairs; a.n0 = n0; a.n1 = n1;
    a.tape_len = n0 + (long long)n0 * n1; a.tape_off0 = 0; a.tape_off1 = n0;
    if (sub) {
        if (saved) return gs::fail(GS_EINVAL, "gs_sage2_forward_subtrees: no saved-activation form");
        if (sub->roots_per_walk < 1 || n_pairs > ((int64_t)1 << 40) / sub->roots_per_walk)
            return gs::fail(GS_EINVAL, "gs_sage2_forward_subtrees: roots_per_walk=%d", sub->roots_per_walk);
        if (replay && (sub->tape_len < 1 || sub->tape_off_hop2 < 0 || sub->tape_off_hop3 < 0 ||
                       sub->tape_off_hop2 + (int64_t)sub->roots_per_walk * n0 > sub->tape_len ||
                       sub->tape_off_hop3 + (int64_t)sub->roots_per_walk * n0 * n1 > sub->tape_len))
            return gs::fail(GS_EINVAL, "gs_sage2_forward_subtrees: the hops do not fit a tape of %lld words per walk", (long long)sub->tape_len);
        if (sub->out_h0 && !aligned16(sub->out_h0)) return gs::fail(GS_EINVAL, "gs_sage2_forward_subtrees: out_h0 must be 16-byte aligned");
        a.sub_n = sub->roots_per_walk;
        a.n_pairs = n_pairs * sub->roots_per_walk;           // roots per node type
        a.tape_len = sub->tape_len; a.tape_off0 = sub->tape_off_hop2; a.tape_off1 = sub->tape_off_hop3;
        a.out_h0 = sub->out_h0;
        a.seeds0 = sub->seeds; a.hop1_out = sub->out_hop1;
        const bool any = sub->sv_x_mean0_hop2[0] || sub->sv_x_mean0_hop2[1] || sub->sv_mask_hop2[0] || sub->sv_mask_hop2[1] ||
                         sub->sv_x_mean0_hop1[0] || sub->sv_x_mean0_hop1[1] || sub->sv_x_mean1_hop1[0] || sub->sv_x_mean1_hop1[1];
        if (any) {
            if (H1 != 32) return gs::fail(GS_ELIMIT, "gs_sage2_forward_subtrees: the saving form has H1 = 32 (got %d)", H1);
            for (int t = 0; t < 2; ++t) {
                if (!sub->sv_x_mean0_hop2[t] || !sub->sv_mask_hop2[t] || !sub->sv_x_mean0_hop1[t] || !sub->sv_x_mean1_hop1[t])
                    return gs::fail(GS_EINVAL, "gs_sage2_forward_subtrees: the sv_ tables are given all or none");
                if (!aligned16(sub->sv_x_mean0_hop2[t]) || !aligned16(sub->sv_x_mean0_hop1[t]) || !aligned16(sub->sv_x_mean1_hop1[t]) ||
                    (reinterpret_cast<uintptr_t>(sub->sv_mask_hop2[t]) & 7))
                    return gs::fail(GS_EINVAL, "gs_sage2_forward_subtrees: the sv_ tables must be 16-byte (masks: 8-byte) aligned");
                a.sv_x_mean[t] = sub->sv_x_mean0_hop2[t]; a.sv_h1_mask[t] = sub->sv_mask_hop2[t];
                a.sv_root_mean0[t] = sub->sv_x_mean0_hop1[t]; a.sv_root_mean1[t] = sub->sv_x_mean1_hop1[t];
            }
        }
        if (sub->seeds && !sub->out_hop1)
            return gs::fail(GS_EINVAL, "gs_sage2_forward_subtrees: seeds given without out_hop1");
        if (sub->seeds && replay && sub->roots_per_walk > sub->tape_len)
            return gs::fail(GS_EINVAL, "gs_sage2_forward_subtrees: the hops do not fit a tape of %lld words per walk", (long long)sub->tape_len);
        if (!gs_sage2_subtrees_supported(Fu, Fi, H1, H2, n0, n1))
            return gs::fail(GS_ELIMIT, "gs_sage2_forward_subtrees: fan-outs [%d,%d] are too large for %d subtrees per wavefront", n0, n1, kGroup);
    }
    a.key0 = (unsigned)rng_seed; a.key1 = (unsigned)(rng_seed >> 32); a.walk_base = walk_id_base;
    a.walk_ids = walk_ids; a.replay = replay;
    a.w0_self[0] = l0_user->w_self; a.w0_neigh[0] = l0_user->w_neigh; a.b0[0] = l0_user->bias;
    a.w0_self[1] = l0_item->w_self; a.w0_neigh[1] = l0_item->w_neigh; a.b0[1] = l0_item->bias;
    a.w1_self[0] = l1_user->w_self; a.w1_neigh[0] = l1_user->w_neigh; a.b1[0] = l1_user->bias;
    a.w1_self[1] = l1_item->w_self; a.w1_neigh[1] = l1_item->w_neigh; a.b1[1] = l1_item->bias;
    a.normalize = sub ? 0 : normalize; a.emb = emb; a.out_hop1 = out_hop1; a.out_hop2 = out_hop2; a.err = err_flag;
    a.ids_per_wave = ((sub ? kGroup : 1) * (n0 + n0 * n1) + 3) & ~3;
    a.h1_rows = sub ? kGroup : 1 + n0;
    if (saved) {
        for (int t = 0; t < 2; ++t) {
            if ((!saved->x_self[t] && !saved->self_ids[t]) || !saved->x_mean[t] || !saved->h1[t] || !saved->x_mean1[t] || !saved->z[t])
                return gs::fail(GS_EINVAL, "gs_sage2_forward: saved-activation set has a null pointer");
            a.sv_self_ids[t] = saved->self_ids[t];
            if (!aligned16(saved->x_self[t]) || !aligned16(saved->x_mean[t]) || !aligned16(saved->h1[t]) ||
                !aligned16(saved->x_mean1[t]))
                return gs::fail(GS_EINVAL, "gs_sage2_forward: saved-activation buffers must be 16-byte aligned");
            a.sv_x_self[t] = saved->x_self[t]; a.sv_x_mean[t] = saved->x_mean[t]; a.sv_h1[t] = saved->h1[t];
            a.sv_x_mean1[t] = saved->x_mean1[t]; a.sv_z[t] = saved->z[t];
            a.sv_h1_mask[t] = saved->h1_mask[t];
        }
        if (!saved->h1_mask[0] != !saved->h1_mask[1])
            return gs::fail(GS_EINVAL, "gs_sage2_forward: h1_mask must be given for both node types or for none");
        if (!(saved->dropout_rate >= 0.f) || saved->dropout_rate >= 1.f)
            return gs::fail(GS_EINVAL, "gs_sage2_forward: dropout_rate=%g outside [0, 1)", (double)saved->dropout_rate);
        if (saved->dropout_rate > 0.f) {
            // the backward pass reads "active and kept" from what is saved: mask words for the hop-1 heads, and
            // the own rows as they entered the layer (not their ids)
            if (!saved->h1_mask[0] || !saved->x_self[0] || !saved->x_self[1])
                return gs::fail(GS_EINVAL, "gs_sage2_forward: dropout needs saved->h1_mask and saved->x_self");
            const double thr = (double)saved->dropout_rate * 4294967296.0;
            a.drop_thr = (uint32_t)(thr < 1.0 ? 1.0 : thr);
            a.drop_scale = 1.0f / (1.0f - saved->dropout_rate);
            a.drop_k0 = (uint32_t)saved->dropout_seed; a.drop_k1 = (uint32_t)(saved->dropout_seed >> 32);
            for (int t = 0; t < 2; ++t) a.sv_self_ids[t] = nullptr;
        }
    }
    hipStream_t s = (hipStream_t)stream;
    const int hh1 = H1 / 2, hh2 = H2 / 2;
#define GS_X(fu, fi, a1, a2) if (Fu == fu && Fi == fi && hh1 == a1 && hh2 == a2) return launch<fu, fi, a1, a2>(a, s);
    GS_SAGE2_CASES(GS_X)
#undef GS_X
    return gs::fail(GS_ELIMIT, "gs_sage2_forward: no instantiation for Fu=%d Fi=%d H1=%d H2=%d", Fu, Fi, H1, H2);
}

}  // namespace

extern "C" int gs_sage2_forward(const int64_t *indptr, const int32_t *indices, int64_t n_nodes,
                                const float *feat_user, int64_t n_users, int Fu,
                                const float *feat_item, int64_t n_items, int Fi,
                                const int32_t *seeds, int64_t n_pairs, int n0, int n1,
                                uint64_t rng_seed, uint64_t walk_id_base, const int64_t *walk_ids,
                                const uint32_t *replay,
                                const gs_sage_weights *l0_user, const gs_sage_weights *l0_item,
                                const gs_sage_weights *l1_user, const gs_sage_weights *l1_item,
                                int H1, int H2, int normalize,
                                float *emb, int32_t *out_hop1, int32_t *out_hop2,
                                const gs_sage2_saved *saved, int32_t *err_flag, void *stream) {
    return sage2_forward_impl(indptr, indices, n_nodes, feat_user, n_users, Fu, feat_item, n_items, Fi, seeds, n_pairs, n0, n1,
                              rng_seed, walk_id_base, walk_ids, replay, l0_user, l0_item, l1_user, l1_item, H1, H2, normalize,
                              emb, out_hop1, out_hop2, saved, nullptr, err_flag, stream);
}

extern "C" int gs_sage2_forward_subtrees(const int64_t *indptr, const int32_t *indices, int64_t n_nodes,
                                         const float *feat_user, int64_t n_users, int Fu,
                                         const float *feat_item, int64_t n_items, int Fi,
                                         const int32_t *hop1_ids, int64_t n_pairs, int n1, int n2,
                                         uint64_t rng_seed, uint64_t walk_id_base, const int64_t *walk_ids,
                                         const uint32_t *replay,
                                         const gs_sage_weights *l0_user, const gs_sage_weights *l0_item,
                                         const gs_sage_weights *l1_user, const gs_sage_weights *l1_item,
                                         int H1, int H2,
                                         float *h1_out, int32_t *out_hop2, int32_t *out_hop3,
                                         const gs_sage2_subtrees *sub, int32_t *err_flag, void *stream) {
    if (!sub) return gs::fail(GS_EINVAL, "gs_sage2_forward_subtrees: null pointer argument");
    return sage2_forward_impl(indptr, indices, n_nodes, feat_user, n_users, Fu, feat_item, n_items, Fi, hop1_ids, n_pairs, n1, n2,
                              rng_seed, walk_id_base, walk_ids, replay, l0_user, l0_item, l1_user, l1_item, H1, H2, 0,
                              h1_out, out_hop2, out_hop3, nullptr, sub, err_flag, stream);
}
