// Fused SAGE layer for wide inputs (feature / hidden widths that are multiples of
// 32, half output width a multiple of 128): the projection is bound by the matrix
// pipe and the neighbor mean by memory, so one kernel runs both at once, on
// different waves of the same block.
//
// Work unit: a block owns tiles of 128 head nodes; per tile and 128-column tile of
// each output half it accumulates a 128x128 fp32 product over k slices of 32:
// first the own rows against W_self, then the neighbor means against W_neigh.
//
//   producer waves   eight waves.  Per k slice the 128 x 32 operand slice: for the neighbor
//                    half the n_child rows of every head are summed as they arrive,
//                    16 bytes per lane, eight lanes per row, and scaled by 1/n_child;
//                    the own-row half is the same code with one "child", the head
//                    itself.  The slice goes into one of two LDS stages in MFMA
//                    fragment order.  A thread keeps the loads of two slices in
//                    flight: while slice s is multiplied, slice s+1 moves from
//                    registers to LDS and the loads of slice s+3 are issued behind
//                    those of s+2.  These loads are issued and waited for by hand
//                    (see load16_async): the compiler's own bookkeeping turns every
//                    wait into a wait for all loads in flight.
//   consumer waves   four waves, each a 64x64 quarter of the tile: one
//                    ds_read_b128 per 16x16 fragment feeds four
//                    v_mfma_f32_16x16x4_f32.  They also fetch the 32 x 128 weight
//                    slice of the next k slice before the multiply (16-byte pieces of
//                    the row-major weight, which is the order the lanes read them in:
//                    MFMA j of a lane covers column 4r + j) and put it into the other
//                    LDS stage after it.  Bias, activation and 16-byte row stores go
//                    through a wave-private LDS patch (storing straight from the
//                    accumulators, which hold four consecutive columns per row, was
//                    measured 8 % slower: the next tile's MFMAs then wait for the
//                    stores to have read their registers).
//
// All waves meet at one barrier per k slice.  Row ids of a tile (node id - table
// offset, or "counts as zeros") are staged in LDS one tile ahead.  Neither the
// gathered rows nor the means go through memory; when the caller asks for copies
// (save_self / save_mean, for a backward pass) the producers also write each
// finished piece there.
#include "fused_args.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using gs::FusedArgs;
using gs::Seg;
using gs::kMaxSegs;

constexpr int WM = 128, WN = 128, WK = 32;
constexpr int W_TILE = WM * WK;                 // floats of one operand slice (A or B): 16 KB
constexpr int EP_LD = 68;                       // row stride of the epilogue patch (lane quarters 16 banks apart)
constexpr int kConsumerThreads = 256;           // four waves, 2 x 2 quarters of the tile
constexpr int kMaxChild = 32;

// Hand-off through a wave-private LDS patch: a wave's LDS instructions execute in order, so only the
// compiler has to keep them in order (a workgroup-scope fence here would also wait for the wave's
// global stores to drain).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Block barrier for the LDS hand-offs of this kernel: LDS writes done, then s_barrier.  (__syncthreads
// also fences global memory, which would wait for every gather load in flight at each barrier.)
__device__ __forceinline__ void block_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Position in a block's sequence of k slices: tile -> column tile -> half -> slice.
struct Cursor {
    int tile;           // global tile index (tiles of 128 heads: 32 bits hold any count the host accepts)
    int local;          // index among this block's tiles (parity picks the id buffer)
    int ct, side, ks;   // column tile, 0 own rows / 1 neighbor means, k slice
    int n_child;        // children per head in the tile's segment; looked up when the tile changes, so that
    float inv;          // a slice costs no scalar loads (their round trips were on every slice's critical path)
    long long row0;     // output row of the tile's first head
    int np;             // heads in the tile (128, fewer at the end of a segment)
};

struct Shape {
    int Fs, Fn, HH;
    int ns, nn, nct;    // k slices per half, column tiles per half
    int stride;         // tiles between two tiles of one block
};

// (uniform values are kept to 32 bits where they are compared: the scalar unit has no ordered 64-bit
// compare, and the vector compare the compiler falls back to needs temporaries in vector registers)
__device__ __forceinline__ int find_seg(const FusedArgs &a, int tile) {
    int si = 0;
#pragma unroll
    for (int s = 1; s < kMaxSegs; ++s)
        if (s < a.n_segs && tile >= (int)a.seg[s].tile0) si = s;
    return __builtin_amdgcn_readfirstlane(si);       // a scalar: a.seg[si] is then read with scalar loads
}

__device__ __forceinline__ void enter_tile(Cursor &c, const FusedArgs &a) {
    const Seg &sg = a.seg[find_seg(a, c.tile)];
    const long long p0 = (long long)(c.tile - (int)sg.tile0) * WM;
    c.n_child = sg.n_child;
    c.inv = 1.0f / (float)c.n_child;
    c.row0 = sg.out_row0 + p0;
    c.np = (int)min((long long)WM, sg.n_heads - p0);
}

__device__ __forceinline__ void advance(Cursor &c, const Shape &sh, const FusedArgs &a) {
    if (++c.ks < (c.side ? sh.nn : sh.ns)) return;
    c.ks = 0;
    if (++c.side < 2) return;
    c.side = 0;
    if (++c.ct < sh.nct) return;
    c.ct = 0;
    c.tile += sh.stride;
    ++c.local;
    enter_tile(c, a);
}

// ---- producer -----------------------------------------------------------------
// LDS keeps one 32-bit key per row to fetch: its byte offset in the table (tables below 4 GB) or
// its row index (BIG), kNoKey for "counts as zeros".
constexpr uint32_t kNoKey = 0xFFFFFFFFu;

template <int NPT, int CH, bool BIG>
struct Producer {
    static constexpr int PP = 1024 / NPT;            // 16-byte pieces of A (and of B) per thread and slice
    static constexpr int RSTEP = NPT / 8;            // rows between two pieces of a thread
    static constexpr int IDT = NPT;
    static constexpr int IDC = WM * kMaxChild / IDT; // child ids a thread stages per tile, at most

    const FusedArgs &a;
    const Shape &sh;
    float *st;
    uint32_t *hrow, *crow;
    int ptid, itid, a_kq, a_r0;
    int a_dst[PP];                                   // where this thread's pieces go in an LDS stage (floats)

    // a k slice in flight (two per thread)
    struct Slice {
        f32x4 v[PP][CH], acc[PP];
        uint32_t mask;                               // bit p*CH+u: load (p, u) stands in for a zero row
        int cnt;                                     // children in the chunk in flight
        bool bad;                                    // some lane of this wave has such a load
        float inv;
        uint32_t save_lo[PP], save_hi[PP];           // address (two halves) where a copy of the finished piece goes, or 0
    };
    // ids in flight
    int32_t cv[IDC], hv;

    __device__ __forceinline__ Producer(const FusedArgs &a_, const Shape &sh_, float *st_, uint32_t *hrow_, uint32_t *crow_, int ptid_, int itid_)
        : a(a_), sh(sh_), st(st_), hrow(hrow_), crow(crow_), ptid(ptid_), itid(itid_), a_kq(ptid_ & 7), a_r0(ptid_ >> 3) {
#pragma unroll
        for (int p = 0; p < PP; ++p) {
            const int row = p * RSTEP + a_r0;
            // fragment block (k half, 16-row block), slot (k quad, row) with the row rotated by 2 * (k quad):
            // this write and the fragment read both touch sixteen different 16-byte bank groups
            a_dst[p] = ((a_kq >> 2) * 8 + (row >> 4)) * 256 + ((a_kq & 3) * 16 + (((row & 15) + 2 * a_kq) & 15)) * 4;
        }
    }

    // 16 bytes at base + offset, issued without the compiler knowing that the result arrives later:
    // the compiler places its own waits by a conservative count of what may be in flight, which, with
    // two slices per thread and branches around, comes out as "wait for everything".  These loads are
    // waited for by hand (wait_last_chunk), knowing that every begin() ends with exactly PP*CH of them.
    __device__ __forceinline__ static f32x4 load16_async(const char *base, uint32_t off) {
        f32x4 r;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(off), "s"(base) : "memory");
        return r;
    }
    __device__ __forceinline__ static f32x4 load16_async(const char *addr) {
        f32x4 r;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(addr) : "memory");
        return r;
    }

    // cnt (<= CH) children of every piece, starting at child c0: always CH loads per piece, none of
    // them conditional; a zero row is fetched from row 0 and dropped when the chunk is added, slots
    // past cnt repeat the last child (a cache hit).  `base` is the table shifted to this slice's first
    // column (uniform), `fb` the bytes of a table row.  ASYNC: see load16_async.
    template <bool ASYNC>
    __device__ __forceinline__ void load_chunk(Slice &s, const char *__restrict__ base, uint32_t fb, const uint32_t *rows, int n, int c0, int cnt) {
        uint32_t mask = 0u;                          // bit p*CH+u: load (p, u) stands in for a zero row
        uint32_t key[PP][CH];                        // all keys first: the LDS reads overlap instead of alternating with loads
#pragma unroll
        for (int p = 0; p < PP; ++p)
#pragma unroll
            for (int u = 0; u < CH; ++u) key[p][u] = rows[(p * RSTEP + a_r0) * n + c0 + min(u, cnt - 1)];
#pragma unroll
        for (int p = 0; p < PP; ++p)
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const bool none = key[p][u] == kNoKey;
                mask |= none ? 1u << (p * CH + u) : 0u;
                const uint32_t k = none ? 0u : key[p][u];
#ifdef WIDE_DIAG_NO_GATHER                              // timing aid (-DWIDE_DIAG_NO_GATHER): everything but the row loads
                s.v[p][u] = f32x4{1.f, 1.f, 1.f, (float)k};
                continue;
#endif
                if (ASYNC) {
                    if (BIG) s.v[p][u] = load16_async(base + (unsigned long long)k * fb + (uint32_t)(a_kq * 16));
                    else s.v[p][u] = load16_async(base, k + (uint32_t)(a_kq * 16));
                } else {
                    if (BIG) s.v[p][u] = *reinterpret_cast<const f32x4 *>(base + (unsigned long long)k * fb + (uint32_t)(a_kq * 16));
                    else s.v[p][u] = *reinterpret_cast<const f32x4 *>(base + (k + (uint32_t)(a_kq * 16)));
                }
            }
        s.mask = mask;
        s.bad = __any(mask != 0u);
        s.cnt = cnt;
    }

    // the PP*CH loads this thread issued BEFORE its latest PP*CH have arrived
    __device__ __forceinline__ void wait_last_chunk(Slice &s) {
        static_assert(PP * CH <= 32, "one mask bit per load; vmcnt is a 6-bit count");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PP * CH) : "memory");
#pragma unroll
        for (int p = 0; p < PP; ++p)
#pragma unroll
            for (int u = 0; u < CH; ++u) asm volatile("" : "+v"(s.v[p][u]));      // uses stay behind the wait
    }

    __device__ __forceinline__ void accumulate(Slice &s) {
        if (s.bad) {                                               // uniform, rare: drop the stand-ins
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < PP; ++p)
#pragma unroll
                for (int u = 0; u < CH; ++u) s.v[p][u] = (s.mask >> (p * CH + u)) & 1u ? z4 : s.v[p][u];
        }
        if (s.cnt == 1) {                                          // own rows
#pragma unroll
            for (int p = 0; p < PP; ++p) s.acc[p] += s.v[p][0];
            return;
        }
        if (s.cnt == CH) {                                         // the common case without per-child selects
#pragma unroll
            for (int u = 0; u < CH; ++u)
#pragma unroll
                for (int p = 0; p < PP; ++p) s.acc[p] += s.v[p][u];
            return;
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (u < s.cnt) {                                       // uniform
#pragma unroll
                for (int p = 0; p < PP; ++p) s.acc[p] += s.v[p][u];
            }
        }
    }

    // loads of one slice: all chunks of children but the last are summed here, the last stays in flight
    __device__ __forceinline__ void begin(Slice &s, const Cursor &c) {
        const int n = c.side ? c.n_child : 1;
        const int F = c.side ? sh.Fn : sh.Fs;
        const unsigned long long b64 = reinterpret_cast<unsigned long long>(c.side ? a.feat_neigh : a.feat_self) + (size_t)c.ks * (WK * 4);
        const char *__restrict__ base = reinterpret_cast<const char *>(                      // uniform: in scalar registers
            ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b64 >> 32)) << 32) |
            (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b64));   // (the builtin returns a signed int)
        const uint32_t *rows = c.side ? crow + (c.local & 1) * WM * a.max_child : hrow + (c.local & 1) * WM;
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < PP; ++p) s.acc[p] = z4;
        const uint32_t fb = (uint32_t)F * 4u;
        const int nch = (n + CH - 1) / CH;
#pragma unroll 1
        for (int ch = 0; ch + 1 < nch; ++ch) {
            load_chunk<false>(s, base, fb, rows, n, ch * CH, CH);
            accumulate(s);
        }
        load_chunk<true>(s, base, fb, rows, n, (nch - 1) * CH, n - (nch - 1) * CH);
        s.inv = c.side ? c.inv : 1.0f;
        // copies for a backward pass: the own rows / the neighbor means, written once (first column tile)
        float *sv = c.side ? a.save_mean : a.save_self;
#pragma unroll
        for (int p = 0; p < PP; ++p) {
            const int row = p * RSTEP + a_r0;
            const unsigned long long addr = (sv && c.ct == 0 && row < c.np)
                ? reinterpret_cast<unsigned long long>(sv + (c.row0 + row) * F + c.ks * WK + a_kq * 4) : 0ull;
            s.save_lo[p] = (uint32_t)addr;
            s.save_hi[p] = (uint32_t)(addr >> 32);
        }
    }

    // the slice in flight: last chunk added, mean taken, both operand slices written to LDS stage `buf`
    __device__ __forceinline__ void finish(Slice &s, int buf) {
        float *As = st + buf * 2 * W_TILE;
        wait_last_chunk(s);
        accumulate(s);
#pragma unroll
        for (int p = 0; p < PP; ++p) {
            s.acc[p] *= s.inv;
            *reinterpret_cast<f32x4 *>(As + a_dst[p]) = s.acc[p];
            const unsigned long long addr = ((unsigned long long)s.save_hi[p] << 32) | s.save_lo[p];
            if (addr) *reinterpret_cast<f32x4 *>(addr) = s.acc[p];
        }
    }

    // ids of a tile: loads first, LDS writes later (the loads of the slice in flight are older and done by then)
    __device__ __forceinline__ void ids_issue(int tile) {
        const Seg &sg = a.seg[find_seg(a, tile)];
        const long long p0 = (long long)(tile - (int)sg.tile0) * WM;
        const int np = (int)min((long long)WM, sg.n_heads - p0);
        const int n = sg.n_child;
        const int32_t *__restrict__ cids = sg.child_ids;
        const int32_t *__restrict__ hids = sg.head_ids;
        const long long last = (long long)np * n - 1;             // np >= 1: a tile holds at least one head
        if (cids) {                                                // unconditional loads (clamped), sorted out at commit
#pragma unroll
            for (int i = 0; i < IDC; ++i) cv[i] = cids[p0 * n + min((long long)(i * IDT + itid), last)];
        } else {
#pragma unroll
            for (int i = 0; i < IDC; ++i) cv[i] = (int32_t)(sg.child_row0 + p0 * n + i * IDT + itid);
        }
        const int hp = min(itid, np - 1);
        hv = hids ? hids[p0 + hp] : (int32_t)(sg.self_row0 + p0 + hp);
    }

    __device__ __forceinline__ void ids_commit(int tile, int parity) {
        const Seg &sg = a.seg[find_seg(a, tile)];
        const long long p0 = (long long)(tile - (int)sg.tile0) * WM;
        const int np = (int)min((long long)WM, sg.n_heads - p0);
        const int n = sg.n_child;
        uint32_t *cdst = crow + parity * WM * a.max_child;
        const uint32_t fbn = BIG ? 1u : (uint32_t)sh.Fn * 4u, fbs = BIG ? 1u : (uint32_t)sh.Fs * 4u;
        // (every loaded id is read here whether it is used or not: a load whose result is never read
        // would stay "in flight" in the compiler's books and cost a wait wherever its register is reused)
#pragma unroll
        for (int i = 0; i < IDC; ++i) {
            const int idx = i * IDT + itid;
            const long long rr = (long long)cv[i] - (sg.child_ids ? a.off_neigh : 0);
            uint32_t key = kNoKey;                                 // also the slots of heads past the end of the segment
            if (idx < np * n && (unsigned long long)rr < (unsigned long long)a.n_rows_neigh) key = (uint32_t)rr * fbn;
            if (idx < WM * n) cdst[idx] = key;
        }
        {
            const long long rr = (long long)hv - (sg.head_ids ? a.off_self : 0);
            uint32_t key = kNoKey;
            if (itid < np && (unsigned long long)rr < (unsigned long long)a.n_rows_self) key = (uint32_t)rr * fbs;
            if (itid < WM) hrow[parity * WM + itid] = key;
        }
    }
};

// ---- kernel ---------------------------------------------------------------------
template <int NPT, int CH, bool BIG>
__global__ __launch_bounds__(kConsumerThreads + NPT)
void sage_wide_kernel(const FusedArgs a, const int Fs, const int Fn, const int HH) {
    extern __shared__ __align__(16) float w_smem[];
    float *st = w_smem;                                            // two stages of [A slice | B slice]
    float *ep = st + 4 * W_TILE;                                   // four epilogue patches [16][EP_LD]
    uint32_t *hrow = reinterpret_cast<uint32_t *>(ep + 4 * 16 * EP_LD); // [2][WM] keys of the own rows of a tile
    uint32_t *crow = hrow + 2 * WM;                                 // [2][WM * max_child] child rows of a tile

    const int tid = threadIdx.x;
    Shape sh;
    sh.Fs = Fs; sh.Fn = Fn; sh.HH = HH;
    sh.ns = Fs / WK; sh.nn = Fn / WK; sh.nct = HH / WN;
    sh.stride = (int)gridDim.x;
    const int n_tiles = (int)a.n_tiles;
    const int my_tiles = (n_tiles - (int)blockIdx.x + sh.stride - 1) / sh.stride;
    const int T = my_tiles * sh.nct * (sh.ns + sh.nn);              // k slices (= barriers) of this block
    const int H = 2 * HH;

    if (tid >= kConsumerThreads) {
        Producer<NPT, CH, BIG> pr(a, sh, st, hrow, crow, tid - kConsumerThreads, tid - kConsumerThreads);
        typename Producer<NPT, CH, BIG>::Slice s0, s1;             // slices of even / odd index
        Cursor cl{(int)blockIdx.x, 0, 0, 0, 0, 1, 1.f, 0, 0};      // next slice to load
        enter_tile(cl, a);
        Cursor cc = cl;                                            // slice the consumers multiply
        int gl = 0;                                                // index of cl; past the last slice it stays there
        // (the surplus loads and LDS writes at the end are harmless and keep the loop free of branches
        // around loads, which is what lets a wait for one slice leave the next one in flight)
#define WD_NEXT() { if (gl + 1 < T) { advance(cl, sh, a); ++gl; } }
        pr.ids_issue(cl.tile);
        pr.ids_commit(cl.tile, 0);
        block_sync();
        pr.begin(s0, cl); WD_NEXT()
        pr.begin(s1, cl); WD_NEXT()
        pr.finish(s0, 0);
        pr.begin(s0, cl); WD_NEXT()
        block_sync();
        // While the consumers multiply slice g: slice g+1 goes from registers to the other LDS stage
        // (its loads were issued two barriers ago), then the freed registers take the loads of slice g+3.
#define WD_TICK(S_, buf_)                                                                           \
        {                                                                                           \
            const int next_tile = cc.tile + sh.stride;                                              \
            const bool stage_ids = cc.ct == 0 && cc.side == 0 && cc.ks == 0 && next_tile < n_tiles; \
            pr.finish(S_, buf_);                                                                    \
            pr.begin(S_, cl); WD_NEXT()                                                             \
            if (stage_ids) { pr.ids_issue(next_tile); pr.ids_commit(next_tile, (cc.local + 1) & 1); } \
            advance(cc, sh, a);                                                                     \
            block_sync();                                                                           \
        }
        for (int g = 0; g < T; g += 2) {
            WD_TICK(s1, 1)
            if (g + 1 < T) WD_TICK(s0, 0)
        }
#undef WD_TICK
#undef WD_NEXT
        return;
    }

    // ---- consumers ----
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // the wave index as a scalar
    const int wave_m = wave & 1, wave_n = wave >> 1;
    const int r = lane & 15, q = lane >> 4;
    const int a_slot0 = q * 16 + ((r + 2 * q) & 15), a_slot1 = q * 16 + ((r + 2 * (q + 4)) & 15);
    float *E = ep + wave * 16 * EP_LD;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // bias, activation, store of a finished 128x128 product (this wave's 64x64 quarter), 16 rows at a time
    auto epilogue = [&](const Cursor &d) {
        const Seg &sg = a.seg[find_seg(a, d.tile)];
        const long long p0 = (long long)(d.tile - (int)sg.tile0) * WM;
        const int np = (int)min((long long)WM, sg.n_heads - p0);
        const int colbase = d.side * HH + d.ct * WN + wave_n * 64;
        float bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = a.bias ? a.bias[colbase + 4 * r + j] : 0.f;   // MFMA j of a lane: column 4r + j
        float *__restrict__ outp = a.out + (sg.out_row0 + p0) * H + colbase;
        const uint32_t lane_off = (uint32_t)(q * H + r * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float val = acc[i][j][e] + bv[j];
                    if (a.act == GS_ACT_RELU) val = fmaxf(val, 0.f);
                    E[(q * 4 + e) * EP_LD + 4 * r + j] = val;
                    acc[i][j][e] = 0.f;
                }
            wave_lds_fence();
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                // four rows per store instruction: a uniform base per group of four, one lane offset for all
                const int row = it * 4 + q, trow0 = wave_m * 64 + i * 16 + it * 4;
                float *__restrict__ rows4 = outp + (long long)trow0 * H;
                if (trow0 + q < np)
                    *reinterpret_cast<float4 *>(rows4 + lane_off) = *reinterpret_cast<const float4 *>(E + row * EP_LD + r * 4);
            }
            wave_lds_fence();
        }
    };

    f32x4 aX0, aX1, aX2, aX3, bX0, bX1, bX2, bX3;
#define WD_READ(S_, base_, half_, slot_)                                                            \
    {                                                                                               \
        const float *As_r = (base_) + ((half_) * 8 + wave_m * 4) * 256 + (slot_) * 4;               \
        const float *Bs_r = (base_) + W_TILE + ((half_) * 8 + wave_n) * 256 + lane * 4;             \
        a##S_##0 = *reinterpret_cast<const f32x4 *>(As_r);                                          \
        b##S_##0 = *reinterpret_cast<const f32x4 *>(Bs_r);                                          \
        a##S_##1 = *reinterpret_cast<const f32x4 *>(As_r + 256);                                    \
        b##S_##1 = *reinterpret_cast<const f32x4 *>(Bs_r + 512);                                    \
        a##S_##2 = *reinterpret_cast<const f32x4 *>(As_r + 512);                                    \
        b##S_##2 = *reinterpret_cast<const f32x4 *>(Bs_r + 1024);                                   \
        a##S_##3 = *reinterpret_cast<const f32x4 *>(As_r + 768);                                    \
        b##S_##3 = *reinterpret_cast<const f32x4 *>(Bs_r + 1536);                                   \
    }
    // a##S##i: row tile i, element st = k step; b##S##st: k step st, element j = column tile (columns 4r + j)
#define WD_MFMA_ROW(i_, av_, bv_, st_)                                                              \
    acc[i_][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[st_], bv_[0], acc[i_][0], 0, 0, 0);       \
    acc[i_][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[st_], bv_[1], acc[i_][1], 0, 0, 0);       \
    acc[i_][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[st_], bv_[2], acc[i_][2], 0, 0, 0);       \
    acc[i_][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[st_], bv_[3], acc[i_][3], 0, 0, 0);
#define WD_MFMA_STEP(S_, st_)                                                                       \
    WD_MFMA_ROW(0, a##S_##0, b##S_##st_, st_) WD_MFMA_ROW(1, a##S_##1, b##S_##st_, st_)             \
    WD_MFMA_ROW(2, a##S_##2, b##S_##st_, st_) WD_MFMA_ROW(3, a##S_##3, b##S_##st_, st_)
#ifdef WIDE_DIAG_NO_MFMA                                    // timing aid (-DWIDE_DIAG_NO_MFMA): everything but the matrix instructions
#define WD_MFMA(S_) asm volatile("" :: "v"(a##S_##0), "v"(a##S_##1), "v"(a##S_##2), "v"(a##S_##3), "v"(b##S_##0), "v"(b##S_##1), "v"(b##S_##2), "v"(b##S_##3));
#else
#define WD_MFMA(S_) WD_MFMA_STEP(S_, 0) WD_MFMA_STEP(S_, 1) WD_MFMA_STEP(S_, 2) WD_MFMA_STEP(S_, 3)
#endif

    // weight slices, for the slice after the one being multiplied: a thread fetches four 16-byte pieces
    // W[k][4c .. 4c+3] of the row-major weight (coalesced: 32 threads cover a 128-column row) and writes
    // them to the other LDS stage when the multiply is done.  MFMA j of a 64-column group covers the
    // columns n = 4r + j, so lane (r, q) needs exactly such a piece -- k = 16 half + 4q + st, c = 16 g + r --
    // for its four MFMAs of k step st: the pieces go to LDS as they are, in the order the lanes read them.
    // (piece j of this thread: k row (j*256 + tid) >> 5, column piece (j*256 + tid) & 31; the offsets are
    // recomputed where used rather than kept in registers)
#define WD_B_ROW(j_) (8 * (j_) + (tid >> 5))
#define WD_B_C4 (tid & 31)
#define WD_B_DST(j_) (W_TILE + ((((WD_B_ROW(j_) >> 4) * 4 + (WD_B_ROW(j_) & 3)) * 2 + (WD_B_C4 >> 4)) * 64 + ((WD_B_ROW(j_) >> 2) & 3) * 16 + (WD_B_C4 & 15)) * 4)
#define WD_B_OFF(j_) ((uint32_t)(WD_B_ROW(j_) * HH + WD_B_C4 * 4))
    float4 bw0, bw1, bw2, bw3;
#define WD_LOAD_B(cur_)                                                                             \
    {                                                                                               \
        const float *__restrict__ W0 = ((cur_).side ? a.W_neigh : a.W_self) + (long long)((cur_).ks * WK) * HH + (cur_).ct * WN; \
        bw0 = *reinterpret_cast<const float4 *>(W0 + WD_B_OFF(0));                                  \
        bw1 = *reinterpret_cast<const float4 *>(W0 + WD_B_OFF(1));                                  \
        bw2 = *reinterpret_cast<const float4 *>(W0 + WD_B_OFF(2));                                  \
        bw3 = *reinterpret_cast<const float4 *>(W0 + WD_B_OFF(3));                                  \
    }
#define WD_STORE_B(stage_)                                                                          \
    {                                                                                               \
        float *Bw = st + (stage_) * 2 * W_TILE;                                                     \
        *reinterpret_cast<float4 *>(Bw + WD_B_DST(0)) = bw0;                                           \
        *reinterpret_cast<float4 *>(Bw + WD_B_DST(1)) = bw1;                                           \
        *reinterpret_cast<float4 *>(Bw + WD_B_DST(2)) = bw2;                                           \
        *reinterpret_cast<float4 *>(Bw + WD_B_DST(3)) = bw3;                                           \
    }

    Cursor c{(int)blockIdx.x, 0, 0, 0, 0, 1, 1.f, 0, 0}, done = c;
    bool pending = false;
    block_sync();
    WD_LOAD_B(c);
    WD_STORE_B(0);
    Cursor cn = c;                                                  // the slice after c
    advance(cn, sh, a);
    block_sync();
    for (int g = 0; g < T; ++g) {
        if (pending) epilogue(done);
        const float *S = st + (int)(g & 1) * 2 * W_TILE;
        if (g + 1 < T) WD_LOAD_B(cn);
        __builtin_amdgcn_sched_barrier(0);
        WD_READ(X, S, 0, a_slot0);
        WD_MFMA(X);
        WD_READ(X, S, 1, a_slot1);
        WD_MFMA(X);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < T) WD_STORE_B((int)((g + 1) & 1));
        advance(cn, sh, a);
        pending = c.ks + 1 == (c.side ? sh.nn : sh.ns);
        done = c;
        advance(c, sh, a);
        block_sync();
    }
    if (pending) epilogue(done);
#undef WD_READ
#undef WD_LOAD_B
#undef WD_B_ROW
#undef WD_B_C4
#undef WD_B_DST
#undef WD_B_OFF
#undef WD_STORE_B
#undef WD_MFMA_ROW
#undef WD_MFMA_STEP
#undef WD_MFMA
}

// rows of x scaled to unit l2 norm, in place (one wave per row)
__global__ __launch_bounds__(256)
void l2norm_inplace_kernel(float *x, int64_t M, int H) {
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += stride) {
        float ss = 0.f;
        for (int c = lane; c < H; c += 64) { const float v = x[m * H + c]; ss += v * v; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float inv = rsqrtf(fmaxf(ss, 1e-12f));
        for (int c = lane; c < H; c += 64) x[m * H + c] *= inv;
    }
}

constexpr int kProducerThreads = 512, kChunk = 5;

inline size_t wide_lds_bytes(int max_child) {
    return (size_t)(4 * W_TILE + 4 * 16 * EP_LD) * sizeof(float) + (size_t)(2 * WM + 2 * WM * max_child) * sizeof(int32_t);
}

}  // namespace

bool gs::sage_wide_supported(int Fs, int Fn, int H, int max_child) {
    const int hh = H / 2;
    return H > 0 && H % 2 == 0 && hh % WN == 0 && Fs >= 2 * WK && Fn >= 2 * WK && Fs % WK == 0 && Fn % WK == 0 &&
           max_child >= 1 && max_child <= kMaxChild;
}

int gs::sage_wide_fwd(FusedArgs a, int Fs, int Fn, int H, hipStream_t stream) {
    long long tiles = 0, rows = 0;
    for (int s = 0; s < a.n_segs; ++s) {
        a.seg[s].tile0 = tiles;
        tiles += (a.seg[s].n_heads + WM - 1) / WM;
        rows += a.seg[s].n_heads;
    }
    a.n_tiles = tiles;
    if (tiles == 0) return GS_OK;
    if (tiles >= (1ll << 24)) return gs::fail(GS_ELIMIT, "gs_sage_fused_fwd: more than 2^31 head nodes in one call");
    if ((reinterpret_cast<uintptr_t>(a.W_self) | reinterpret_cast<uintptr_t>(a.W_neigh)) & 15)
        return gs::fail(GS_EINVAL, "gs_sage_fused_fwd: wide layers read the weights 16 bytes at a time; W_self / W_neigh must be 16-byte aligned");
    // tables below 4 GB are addressed with 32-bit byte offsets
    const bool big = (unsigned long long)a.n_rows_self * Fs * 4ull >= (1ull << 32) - 1024 ||
                     (unsigned long long)a.n_rows_neigh * Fn * 4ull >= (1ull << 32) - 1024;
    static const hipError_t attr = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sage_wide_kernel<kProducerThreads, kChunk, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)wide_lds_bytes(kMaxChild));
        if (e != hipSuccess) return e;
        return hipFuncSetAttribute(reinterpret_cast<const void *>(sage_wide_kernel<kProducerThreads, kChunk, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)wide_lds_bytes(kMaxChild));
    }();
    if (attr != hipSuccess) return gs::fail(GS_ELAUNCH, "gs_sage_fused_fwd: %s", hipGetErrorString(attr));
    static const int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        return n;
    }();
    const long long blocks = tiles < n_cu ? tiles : n_cu;          // one block per CU (LDS), tiles dealt round-robin
    const dim3 grid((unsigned)blocks), block(kConsumerThreads + kProducerThreads);
    if (big)
        gs::launch_timed(sage_wide_kernel<kProducerThreads, kChunk, true>, grid, block, wide_lds_bytes(a.max_child), stream, a, Fs, Fn, H / 2);
    else
        gs::launch_timed(sage_wide_kernel<kProducerThreads, kChunk, false>, grid, block, wide_lds_bytes(a.max_child), stream, a, Fs, Fn, H / 2);
    int rc = gs::check_launch("gs_sage_fused_fwd");
    if (rc != GS_OK || !a.normalize) return rc;
    long long nb = (rows + 3) / 4;
    if (nb > 65536) nb = 65536;
    hipLaunchKernelGGL(l2norm_inplace_kernel, dim3((unsigned)nb), dim3(256), 0, stream, a.out, (int64_t)rows, H);
    return gs::check_launch("gs_sage_fused_fwd");
}
