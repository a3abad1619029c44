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
//                    itself.  The slice goes into one of kStages LDS stages in MFMA
//                    fragment order.  A thread keeps the loads of two slices in
//                    flight.  These loads are issued and waited for by hand
//                    (see load16_async): the compiler's own bookkeeping turns every
//                    wait into a wait for all loads in flight.  Every wave stages the
//                    row ids of its own 16 heads of a tile in a private LDS patch, one
//                    tile ahead, so that producers never wait for one another.
//   consumer waves   four waves, each a 64x64 quarter of the tile: one
//                    ds_read_b128 per 16x16 fragment feeds four
//                    v_mfma_f32_16x16x4_f32.  The weight fragments come straight from
//                    global memory (the weights stay in L2) in the order the lanes use
//                    them -- MFMA j of a lane covers column 4r + j, so a lane reads 16
//                    bytes of a weight row -- and each register is reloaded for the next
//                    slice as soon as its k step has been issued, which puts seven k steps
//                    between a load and its use.  Bias, activation and 16-byte row
//                    stores go through a wave-private LDS patch (storing straight from
//                    the accumulators, which hold four consecutive columns per row, was
//                    measured 8 % slower: the next tile's MFMAs then wait for the
//                    stores to have read their registers).
//
// The two sides meet only through the stages: a producer wave adds one to full[stage] after its
// part of a slice is written, a consumer wave adds one to empty[stage] after its reads of the stage
// have been issued (a wave's LDS operations execute in order, so the counter is behind them), and
// each side waits for the count that says the other is done with the stage -- eight producer parts
// before a slice is multiplied, four consumer reads before it is overwritten.  There is no block
// barrier in the loop: producers run ahead by up to kStages slices in LDS plus two in registers,
// which is what absorbs the spread of the row loads' latency.  Every wait is a bounded spin (a
// block whose count never arrives writes NaN to its first output element instead of hanging).
//
// Two forms of the product.  Without a workspace the operands stay fp32 and the consumers run
// v_mfma_f32_16x16x4_f32.  That instruction executes on the SIMD's vector ALU: while a wave streams it, no other
// wave of that SIMD issues a vector instruction (tools/micro/mfma_valu.cpp: time of both = sum of the times), so
// the producers' adds and the matrix product take turns and the kernel runs at about half of either's rate.
// With a workspace (SPLIT) every fp32 operand is cut into three bf16 pieces whose sum is the operand exactly
// (8 + 8 + 8 significand bits, by truncation), the nine piece products -- each exact in fp32 -- are accumulated in
// fp32 by v_mfma_f32_16x16x32_bf16, which runs on the matrix cores beside the vector ALU at sixteen times the
// rate: 9/16 of the matrix time, overlapped with the producers.  The result is the fp32 product with the same
// (or smaller) rounding error; only the order of the additions differs.  The weights are cut once per call by
// a small kernel into the workspace (transposed, so that a lane's eight k values are 16 consecutive bytes), the
// gathered operand by the producers as they write a stage.
//
// Neither the gathered rows nor the means go through memory; when the caller asks for copies
// (save_self / save_mean, for a backward pass) the producers also write each
// finished piece there.
#include "fused_args.h"

#include <algorithm>
#include <vector>

namespace {

// Diagnostic build (-DGS_WIDE_STAMP, tools/build_variant.sh): shader-clock stamps around the waits and phases of
// one producer and one consumer wave of two blocks, printed when the block ends.  No stamp executes otherwise.
#ifdef GS_WIDE_STAMP
#define STAMP(v_) const long long v_ = clock64();
#define STAMP_ADD(acc_, v_) acc_ += clock64() - v_;
#define STAMP_VARS(...) long long __VA_ARGS__;
#else
#define STAMP(v_)
#define STAMP_ADD(acc_, v_)
#define STAMP_VARS(...)
#endif

using f32x4 = __attribute__((ext_vector_type(4))) float;
// Pointers built from integers are generic to the compiler, and a generic ("flat") access may be an
// LDS access: while one is pending every wait for a load becomes a wait for all of them.  These
// say "global memory" outright.
using gf32x4 = __attribute__((address_space(1))) f32x4;
__device__ __forceinline__ const gf32x4 *global_cptr(const void *p) { return (const gf32x4 *)(unsigned long long)p; }
__device__ __forceinline__ gf32x4 *global_ptr(unsigned long long p) { return (gf32x4 *)p; }
using gs::FusedArgs;
using gs::Seg;
using gs::kMaxSegs;
using gs::TilePlan;

constexpr int WM = 128, WN = 128, WK = 32;
constexpr int W_TILE = WM * WK;                 // floats of one operand slice (A or B): 16 KB
constexpr int W_PLANE_B = WM * WK * 2;          // SPLIT: bytes of one bf16 plane of a slice (128 rows x 64 B): 8 KB
constexpr int W_TILE_S = 3 * W_PLANE_B / 4;     // SPLIT: floats of a stage (three planes): 24 KB
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
constexpr int EP_LD = 68;                       // row stride of the epilogue patch (lane quarters 16 banks apart)
constexpr int kConsumerThreads = 256;           // four waves, 2 x 2 quarters of the tile
constexpr int kConsumerWaves = 4, kProducerWaves = 8;
constexpr int kMaxChild = 32;
constexpr int kMaxStages = 8;                   // LDS stages of the operand slice, at most; a launch takes what LDS holds
constexpr int kWaveRows = WM / kProducerWaves;  // heads of a tile whose rows one producer wave fetches: 16
constexpr int kSpinLimit = 1 << 22;             // polls before a wait gives up (each tens of cycles or more)

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

// Keeps a branch a branch (see Producer::accumulate).
__device__ __forceinline__ void kBranch() { asm volatile(""); }

// Counters in LDS that hand a stage from one side to the other.  flag_add comes after the wave's
// earlier LDS operations in program order and LDS executes a wave's operations in order, so no wait
// is needed before it (a workgroup-scope release would also wait for the wave's global loads).
__device__ __forceinline__ void flag_add(uint32_t *f) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t flag_peek(const uint32_t *f) {
    return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// true once *f has reached `need` (counters only grow; the difference is compared, so wrapping is harmless)
__device__ __forceinline__ bool flag_wait(const uint32_t *f, uint32_t need, uint32_t seen) {
    int spins = 0;
    while ((int32_t)((uint32_t)__builtin_amdgcn_readfirstlane((int)seen) - need) < 0) {
        if (++spins > kSpinLimit) return false;
        __builtin_amdgcn_s_sleep(1);
        seen = flag_peek(f);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return true;
}

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
    // (a pointer that says "LDS" in its type: read through a generic pointer these words would be flat loads, which
    // count as reads of global memory and would make every wait behind them a wait for all row loads in flight)
    const __attribute__((address_space(3))) int *plan;
                        // LDS [kMaxSegs][2]: of segment s this block takes the tiles first, first + step, ... below `end` (numbered
                        // inside the segment); [2 * kMaxSegs]: step, the number of blocks of its class
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

// A block's tiles: segment after segment, every step-th tile of the part of the segment its class of blocks
// takes (the host sizes the parts so that the blocks' gather work is level: a tile of a segment with 25 children
// per head costs four of one with 5).  Blocks of a class work on neighboring tiles at any time, as round-robin
// dealing would have them: runs of consecutive tiles per block put all blocks on the same memory channels at once.
__device__ __forceinline__ int plan_first(const Shape &sh, const FusedArgs &a, int from_seg) {
    int si = kMaxSegs;                               // the first segment from from_seg on of which the block has a tile left
#pragma unroll
    for (int s = kMaxSegs - 1; s >= 0; --s)
        if (s >= from_seg && s < a.n_segs && sh.plan[2 * s] < sh.plan[2 * s + 1]) si = s;
    si = __builtin_amdgcn_readfirstlane(si);         // (a scalar, as in find_seg: a.seg[si] is then read with scalar loads)
    if (si == kMaxSegs) return (int)a.n_tiles;       // no tile left
    return (int)a.seg[si].tile0 + __builtin_amdgcn_readfirstlane(sh.plan[2 * si]);
}
__device__ __forceinline__ int plan_next(int tile, const Shape &sh, const FusedArgs &a) {
    const int s = find_seg(a, tile);
    const int next = tile + __builtin_amdgcn_readfirstlane(sh.plan[2 * kMaxSegs]);
    return next < (int)a.seg[s].tile0 + __builtin_amdgcn_readfirstlane(sh.plan[2 * s + 1]) ? next : plan_first(sh, a, s + 1);
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
    c.tile = plan_next(c.tile, sh, a);
    ++c.local;
    enter_tile(c, a);
}

// ---- producer -----------------------------------------------------------------
// LDS keeps one 32-bit key per row to fetch: its byte offset in the table (tables below 4 GB) or
// its row index (BIG), kNoKey for "counts as zeros".
constexpr uint32_t kNoKey = 0xFFFFFFFFu;

// x = hi + mid + lo exactly, each piece a bf16 (the top half of an fp32 word): truncation keeps every
// difference exact, 24 significand bits = 3 x 8.  (Results below the bf16 normal range lose their lowest piece.)
__device__ __forceinline__ void split3(float x, uint32_t &hi, uint32_t &mid, uint32_t &lo) {
    hi = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(hi);
    mid = __float_as_uint(r1) & 0xFFFF0000u;
    lo = __float_as_uint(r1 - __uint_as_float(mid));
}
// the top halves of two fp32 words as one register: lo half from a, hi half from b
__device__ __forceinline__ uint32_t pack_hi16(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

template <int NPT, int CH, bool BIG, bool SPLIT>
struct Producer {
    static_assert(NPT == 64 * kProducerWaves, "eight producer waves: a wave fetches rows 8w .. 8w+7 and 64+8w .. 64+8w+7 of a tile");
    static constexpr int PP = 1024 / NPT;            // 16-byte pieces of A per thread and slice: 2
    static constexpr int RSTEP = NPT / 8;            // rows between two pieces of a thread: 64
    static constexpr int IDR = 8 * kMaxChild / 64;   // loads of a lane for the child ids of one run of eight heads, at most
    static constexpr int IDC = PP * IDR;             // child ids a lane stages per tile, at most
    static constexpr int kIdLoads = IDC + 1;         // loads of ids_issue(): the child ids and one head id

    const FusedArgs &a;
    const Shape &sh;
    float *st;
    uint32_t *hrow, *crow;                           // this wave's patches: [2][16] own-row keys, [2][16 * max_child] child keys
    int ptid, lane, wv, a_kq, a_r0;
    int a_dst[PP];                                   // where this thread's pieces go in an LDS stage (floats)

    // a k slice in flight (two per thread)
    struct Slice {
        f32x4 v[PP][CH], acc[PP];
        uint32_t mask;                               // bit p*CH+u: load (p, u) stands in for a zero row
        int cnt;                                     // children in the chunk in flight (its loads: PP * cnt)
        bool bad;                                    // some lane of this wave has such a load
        bool fresh;                                  // no chunk has been summed into acc yet
        float inv;
        unsigned long long save_base;                // uniform: where copies of the finished pieces go (0: nowhere), and
        uint32_t save_off[PP];                       // this thread's byte offsets from it (kNoKey: not this piece)
    };
    // ids in flight
    int32_t cv[IDC], hv;
    bool clean[2];                                   // per id patch: every key of this wave's heads names a table row

    __device__ __forceinline__ Producer(const FusedArgs &a_, const Shape &sh_, float *st_, uint32_t *hrow_, uint32_t *crow_, int ptid_)
        : a(a_), sh(sh_), st(st_), ptid(ptid_), lane(ptid_ & 63), wv(__builtin_amdgcn_readfirstlane(ptid_ >> 6)),
          a_kq(ptid_ & 7), a_r0(ptid_ >> 3) {
        hrow = hrow_ + wv * 2 * kWaveRows;
        crow = crow_ + wv * 2 * kWaveRows * a.max_child;
#pragma unroll
        for (int p = 0; p < PP; ++p) {
            const int row = p * RSTEP + a_r0;
            // fragment block (k half, 16-row block), slot (k quad, row) with the row rotated by 2 * (k quad):
            // this write and the fragment read both touch sixteen different 16-byte bank groups
            a_dst[p] = ((a_kq >> 2) * 8 + (row >> 4)) * 256 + ((a_kq & 3) * 16 + (((row & 15) + 2 * a_kq) & 15)) * 4;
            // SPLIT: a plane is [128 rows][32 k] bf16, 64 bytes per row; a lane of the product reads the 16 bytes
            // of k = 8q .. 8q+7, so the four 16-byte pieces of a row are permuted by bits 2-3 of the row (sixteen
            // consecutive rows then cover all banks); this thread's four k are half of such a piece.  In bytes.
            if (SPLIT) a_dst[p] = row * 64 + (((a_kq >> 1) ^ ((row >> 2) & 3)) * 16) + (a_kq & 1) * 8;
        }
    }

    // 16 bytes at base + offset, issued without the compiler knowing that the result arrives later:
    // the compiler places its own waits by a conservative count of what may be in flight, which, with
    // two slices per thread and branches around, comes out as "wait for everything".  These loads are
    // waited for by hand (wait_last_chunk), knowing that a begin() leaves exactly PP * cnt of them in
    // flight.
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

    // Children c0 .. c0 + cnt - 1 (cnt <= CH) of every piece.  ONE: the chunk is a single child (the
    // own rows): PP loads.  Otherwise always CH loads per piece, none of them conditional: slots past
    // cnt repeat the last child (a cache hit) and are left out of the sum.  `base` is the table shifted
    // to this slice's first column (uniform), `fb` the bytes of a table row.  With CLEAN (every key of
    // the tile names a row) nothing is checked; otherwise a zero row is fetched from row 0 and dropped
    // when the chunk is added.  ASYNC: see load16_async.
    template <bool ONE, bool ASYNC, bool CLEAN>
    __device__ __forceinline__ void load_chunk_n(Slice &s, const char *__restrict__ base, uint32_t fb, const uint32_t *mine, int n, int cnt) {
        constexpr int NU = ONE ? 1 : CH;
        uint32_t mask = 0u;                          // bit p*CH+u: load (p, u) stands in for a zero row
        uint32_t key[PP][NU];                        // all keys first: the LDS reads overlap instead of alternating with loads
#pragma unroll
        for (int p = 0; p < PP; ++p)
#pragma unroll
            for (int u = 0; u < NU; ++u) key[p][u] = mine[p * 8 * n + min(u, cnt - 1)];
#pragma unroll
        for (int p = 0; p < PP; ++p)
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                uint32_t k = key[p][u];
                if (!CLEAN) {
                    const bool none = k == kNoKey;
                    mask |= none ? 1u << (p * CH + u) : 0u;
                    k = none ? 0u : k;
                }
                if (BIG) s.v[p][u] = load16_async(base + (unsigned long long)k * fb + (uint32_t)(a_kq * 16));
                else s.v[p][u] = load16_async(base, k + (uint32_t)(a_kq * 16));
            }
        if (!ASYNC) {                                // a chunk that is summed right away: loads return in order, so
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // "all of mine" is "all": the older ones are done before them
#pragma unroll
            for (int p = 0; p < PP; ++p)
#pragma unroll
                for (int u = 0; u < NU; ++u) asm volatile("" : "+v"(s.v[p][u]));
        }
        // (the registers this shape leaves alone count as written too, at no cost: a register written on
        // one path only stays live across the other, and the allocator then runs out)
#pragma unroll
        for (int p = 0; p < PP; ++p)
#pragma unroll
            for (int u = NU; u < CH; ++u) asm volatile("" : "=v"(s.v[p][u]));
        s.mask = mask;
        s.bad = CLEAN ? false : (bool)__any(mask != 0u);
        s.cnt = cnt;
    }
    template <bool ASYNC>
    __device__ __forceinline__ void load_chunk(Slice &s, const char *__restrict__ base, uint32_t fb, const uint32_t *rows, int n, int c0, int cnt, bool clean) {
        const uint32_t *mine = rows + (lane >> 3) * n + c0;
        if (ASYNC && cnt == 1) {                                   // uniform
            if (clean) load_chunk_n<true, ASYNC, true>(s, base, fb, mine, n, 1);
            else load_chunk_n<true, ASYNC, false>(s, base, fb, mine, n, 1);
        } else {
            if (clean) load_chunk_n<false, ASYNC, true>(s, base, fb, mine, n, cnt);
            else load_chunk_n<false, ASYNC, false>(s, base, fb, mine, n, cnt);
        }
    }

    // the loads of this slice's last chunk have arrived: everything but what this thread issued after
    // them -- the other slice's chunk of `newer` children per piece (PP loads if that is one child,
    // PP * CH otherwise) and, with `ids`, the kIdLoads id loads of ids_issue()
    __device__ __forceinline__ void wait_last_chunk(Slice &s, int newer, bool ids) {
        static_assert(PP * CH <= 32, "one mask bit per load");
        static_assert(PP * CH + kIdLoads < 64, "vmcnt is a 6-bit count");
        // Invariant behind the counts: between the loads this wait is for and the wait itself, this thread has
        // issued exactly one begin() of the other slice -- whose last chunk is PP loads (a single child: the own
        // rows) or PP * CH loads (children; slots past the count repeat the last child, so the number does not
        // depend on it) -- and, when `ids`, one ids_issue() of kIdLoads loads, in that order or the reverse.
        // The checking build (-DGS_WIDE_SAFE_WAITS, libgsage_safewait.so) waits for everything instead; a test
        // compares the two builds bit for bit, so a miscount shows up as a difference.
#ifdef GS_WIDE_SAFE_WAITS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        (void)newer; (void)ids;
#else
        if (!ids) {                                                // uniform
            if (newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PP) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PP * CH) : "memory");
        } else {
            if (newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PP + kIdLoads) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PP * CH + kIdLoads) : "memory");
        }
#endif
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
        // (kBranch in an arm: these conditions are uniform and the arms cheap, which the compiler answers
        // by computing both arms and selecting per lane -- more vector instructions than either arm)
        if (s.fresh) {                                             // uniform: the sum starts with child 0
            kBranch();
#pragma unroll
            for (int p = 0; p < PP; ++p) s.acc[p] = s.v[p][0];
        } else {
            kBranch();
#pragma unroll
            for (int p = 0; p < PP; ++p) s.acc[p] += s.v[p][0];
        }
        s.fresh = false;
        if (s.cnt == 1) return;                                    // own rows
        if (s.cnt == CH) {                                         // the common case without per-child branches
            kBranch();
#pragma unroll
            for (int u = 1; u < CH; ++u)
#pragma unroll
                for (int p = 0; p < PP; ++p) s.acc[p] += s.v[p][u];
            return;
        }
#pragma unroll
        for (int u = 1; u < CH; ++u) {
            if (u < s.cnt) {                                       // uniform
                kBranch();
#pragma unroll
                for (int p = 0; p < PP; ++p) s.acc[p] += s.v[p][u];
            }
        }
    }

    // loads of one slice: all chunks of children but the last are summed here, the last stays in flight.
    // `clean`: see load_chunk.
    __device__ __forceinline__ void begin(Slice &s, const Cursor &c, bool clean) {
        const int n = c.side ? c.n_child : 1;
        const int F = c.side ? sh.Fn : sh.Fs;
        const unsigned long long b64 = reinterpret_cast<unsigned long long>(c.side ? a.feat_neigh : a.feat_self) + (size_t)c.ks * (WK * 4);
        const char *__restrict__ base = reinterpret_cast<const char *>(                      // uniform: in scalar registers
            ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b64 >> 32)) << 32) |
            (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b64));   // (the builtin returns a signed int)
        const uint32_t *rows = c.side ? crow + (c.local & 1) * kWaveRows * a.max_child : hrow + (c.local & 1) * kWaveRows;
        const uint32_t fb = (uint32_t)F * 4u;
        const int nch = (n + CH - 1) / CH;
        s.fresh = true;
#pragma unroll 1
        for (int ch = 0; ch + 1 < nch; ++ch) {
            load_chunk<false>(s, base, fb, rows, n, ch * CH, CH, clean);
            accumulate(s);
        }
        load_chunk<true>(s, base, fb, rows, n, (nch - 1) * CH, n - (nch - 1) * CH, clean);
        s.inv = c.side ? c.inv : 1.0f;
        // copies for a backward pass: the own rows / the neighbor means, written once (first column tile)
        float *sv = c.side ? a.save_mean : a.save_self;
        if (sv && c.ct == 0) {                                     // uniform
            s.save_base = reinterpret_cast<unsigned long long>(sv + c.row0 * F + c.ks * WK);
#pragma unroll
            for (int p = 0; p < PP; ++p) {
                const int row = p * RSTEP + a_r0;
                s.save_off[p] = row < c.np ? ((uint32_t)row * (uint32_t)F + (uint32_t)a_kq * 4u) * 4u : kNoKey;
            }
        } else {
            s.save_base = 0ull;
        }
    }

    // the slice in flight: last chunk added, mean taken, this thread's pieces written to LDS stage `buf`.
    // `newer`, `ids`: what this thread issued after this slice's loads (see wait_last_chunk).
    __device__ __forceinline__ void finish(Slice &s, int buf, int newer, bool ids) {
        float *As = st + buf * (SPLIT ? W_TILE_S : W_TILE);
        wait_last_chunk(s, newer, ids);
        accumulate(s);
        if (s.inv != 1.0f) {                                       // uniform
            kBranch();
#pragma unroll
            for (int p = 0; p < PP; ++p) s.acc[p] *= s.inv;
        }
        if (SPLIT) {
            char *Ab = reinterpret_cast<char *>(As);
#pragma unroll
            for (int p = 0; p < PP; ++p) {
                uint32_t hi[4], mid[4], lo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) split3(s.acc[p][e], hi[e], mid[e], lo[e]);
                *reinterpret_cast<uint2 *>(Ab + a_dst[p]) = make_uint2(pack_hi16(hi[0], hi[1]), pack_hi16(hi[2], hi[3]));
                *reinterpret_cast<uint2 *>(Ab + W_PLANE_B + a_dst[p]) = make_uint2(pack_hi16(mid[0], mid[1]), pack_hi16(mid[2], mid[3]));
                *reinterpret_cast<uint2 *>(Ab + 2 * W_PLANE_B + a_dst[p]) = make_uint2(pack_hi16(lo[0], lo[1]), pack_hi16(lo[2], lo[3]));
            }
        } else {
#pragma unroll
            for (int p = 0; p < PP; ++p) *reinterpret_cast<f32x4 *>(As + a_dst[p]) = s.acc[p];
        }
        if (s.save_base) {                                         // uniform
#pragma unroll
            for (int p = 0; p < PP; ++p)
                if (s.save_off[p] != kNoKey) *global_ptr(s.save_base + s.save_off[p]) = s.acc[p];
        }
    }

    // Row ids of this wave's 16 heads of a tile (two runs of eight consecutive heads, 8 n_child
    // consecutive child ids each).  Patch layout: child key of (run, head j, child c) at (run * 8 + j) * n + c.
    // The loads -- always kIdLoads of them, from a stand-in address where the ids are row numbers that
    // need no load -- are hidden from the compiler like the row loads, so that they can stay in flight
    // for a whole slice: ids_commit() waits for them by count.
    __device__ __forceinline__ static int32_t load4_async(const int32_t *addr) {
        int32_t r;
        asm volatile("global_load_dword %0, %1, off" : "=v"(r) : "v"(addr) : "memory");
        return r;
    }
    __device__ __forceinline__ void ids_issue(int tile) {
        const Seg &sg = a.seg[find_seg(a, tile)];
        const long long p0 = (long long)(tile - (int)sg.tile0) * WM;
        const int np = (int)min((long long)WM, sg.n_heads - p0);
        const int n = sg.n_child;
        const int32_t *standin = reinterpret_cast<const int32_t *>(a.W_self);
        const int32_t *cids = sg.child_ids ? sg.child_ids + p0 * n : standin;
        const int32_t *hids = sg.head_ids ? sg.head_ids + p0 : standin;
        const long long last = sg.child_ids ? (long long)np * n - 1 : 0;   // np >= 1: a tile holds at least one head
#pragma unroll
        for (int run = 0; run < PP; ++run)
#pragma unroll
            for (int i = 0; i < IDR; ++i)                          // unconditional loads (clamped), sorted out at commit
                cv[run * IDR + i] = load4_async(cids + min((long long)(run * RSTEP + 8 * wv) * n + i * 64 + lane, last));
        const int hp = sg.head_ids ? min((lane >> 3 & 1) * RSTEP + 8 * wv + (lane & 7), np - 1) : 0;
        hv = load4_async(hids + hp);
    }

    // `newer`: children per piece of the one chunk of row loads issued after ids_issue() (0: none)
    __device__ __forceinline__ void ids_commit(int tile, int parity, int newer) {
        const Seg &sg = a.seg[find_seg(a, tile)];
        const long long p0 = (long long)(tile - (int)sg.tile0) * WM;
        const int np = (int)min((long long)WM, sg.n_heads - p0);
        const int n = sg.n_child;
        uint32_t *cdst = crow + parity * kWaveRows * a.max_child;
        const uint32_t fbn = BIG ? 1u : (uint32_t)sh.Fn * 4u, fbs = BIG ? 1u : (uint32_t)sh.Fs * 4u;
#ifdef GS_WIDE_SAFE_WAITS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        (void)newer;
#else
        if (newer == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // uniform
        else if (newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PP) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PP * CH) : "memory");
#endif
#pragma unroll
        for (int i = 0; i < IDC; ++i) asm volatile("" : "+v"(cv[i]));                       // uses stay behind the wait
        asm volatile("" : "+v"(hv));
        if (!sg.child_ids) {                                       // rows of an activation buffer: numbered, not loaded
#pragma unroll
            for (int run = 0; run < PP; ++run)
#pragma unroll
                for (int i = 0; i < IDR; ++i)
                    cv[run * IDR + i] = (int32_t)(sg.child_row0 + (p0 + run * RSTEP + 8 * wv) * n + i * 64 + lane);
        }
        if (!sg.head_ids) hv = (int32_t)(sg.self_row0 + p0 + min((lane >> 3 & 1) * RSTEP + 8 * wv + (lane & 7), np - 1));
        bool none = false;
#pragma unroll
        for (int run = 0; run < PP; ++run) {
            const int l0 = (run * RSTEP + 8 * wv) * n;
#pragma unroll
            for (int i = 0; i < IDR; ++i) {
                const int idx = i * 64 + lane;
                const long long rr = (long long)cv[run * IDR + i] - (sg.child_ids ? a.off_neigh : 0);
                uint32_t key = kNoKey;                             // also the slots of heads past the end of the segment
                if (l0 + idx < np * n && (unsigned long long)rr < (unsigned long long)a.n_rows_neigh) key = (uint32_t)rr * fbn;
                if (idx < 8 * n) {
                    cdst[run * 8 * n + idx] = key;
                    none |= key == kNoKey;
                }
            }
        }
        {
            const int row = (lane >> 3 & 1) * RSTEP + 8 * wv + (lane & 7);
            const long long rr = (long long)hv - (sg.head_ids ? a.off_self : 0);
            uint32_t key = kNoKey;
            if (row < np && (unsigned long long)rr < (unsigned long long)a.n_rows_self) key = (uint32_t)rr * fbs;
            if (lane < kWaveRows) {
                hrow[parity * kWaveRows + lane] = key;
                none |= key == kNoKey;
            }
        }
        const bool ok = !__any(none);
        if (parity) clean[1] = ok; else clean[0] = ok;
        wave_lds_fence();                                          // read back by this wave only
    }
};

// ---- kernel ---------------------------------------------------------------------
template <int NPT, int CH, bool BIG, bool SPLIT>
__global__ __launch_bounds__(kConsumerThreads + NPT)
void sage_wide_kernel(const FusedArgs a, const TilePlan tp, const int Fs, const int Fn, const int HH, const int kStages) {
    extern __shared__ __align__(16) float w_smem[];
    float *st = w_smem;                                            // kStages operand slices
    // after them: the four epilogue patches [16][EP_LD] (fp32 form) or the two buffers of the weight pieces (SPLIT)
    float *ep = st + kStages * (SPLIT ? W_TILE_S : W_TILE);
    uint32_t *full = reinterpret_cast<uint32_t *>(ep + (SPLIT ? 2 * W_TILE_S : 4 * 16 * EP_LD));  // [kMaxStages] producer parts written, ever
    uint32_t *empty = full + kMaxStages;                           // [kMaxStages] consumer reads issued, ever
    uint32_t *bfull = empty + kMaxStages;                          // SPLIT [2]: consumer waves whose part of a weight buffer has landed, ever
    uint32_t *bempty = bfull + 2;                                  // SPLIT [2]: consumer waves that have read a weight buffer, ever
    int *plan = reinterpret_cast<int *>(bempty + 2);               // [kMaxSegs][2] + 4: this block's tiles in every segment (Shape::plan)
    uint32_t *hrow = bempty + 2 + 2 * kMaxSegs + 4;                // [8 waves][2][16] keys of the own rows of a tile
    uint32_t *crow = hrow + kProducerWaves * 2 * kWaveRows;        // [8 waves][2][16 * max_child] child rows of a tile

    const int tid = threadIdx.x;
    Shape sh;
    sh.Fs = Fs; sh.Fn = Fn; sh.HH = HH;
    sh.ns = Fs / WK; sh.nn = Fn / WK; sh.nct = HH / WN;
    sh.plan = (const __attribute__((address_space(3))) int *)plan;
    const int n_tiles = (int)a.n_tiles;
    const int H = 2 * HH;
    bool stalled = false;                                          // some wait of this wave gave up

    if (tid < 2 * kMaxStages + 4) full[tid] = 0u;
    if (tid == 0) {                                                // this block's share, from its class (scalar reads of the arguments)
        int ci = 0;
        for (int i = 1; i < tp.n_cls; ++i)
            if ((int)blockIdx.x >= tp.cls[i].b0) ci = i;
        ci = __builtin_amdgcn_readfirstlane(ci);
        const int b0 = tp.cls[ci].b0;
        const int width = (ci + 1 < tp.n_cls ? tp.cls[ci + 1].b0 : (int)gridDim.x) - b0;
#pragma unroll
        for (int s = 0; s < kMaxSegs; ++s) {
            const int cnt = s < a.n_segs ? tp.cls[ci].count[s] : 0;
            const int first = s < a.n_segs ? tp.cls[ci].start[s] + ((int)blockIdx.x - b0) : 0;
            plan[2 * s] = first;
            plan[2 * s + 1] = first + cnt * width;                 // (past the last one)
        }
        plan[2 * kMaxSegs] = width;
    }
    block_sync();                                                  // the only block barrier
    int my_tiles = 0;
#pragma unroll
    for (int s = 0; s < kMaxSegs; ++s)
        my_tiles += (__builtin_amdgcn_readfirstlane(plan[2 * s + 1]) - __builtin_amdgcn_readfirstlane(plan[2 * s])) /
                    __builtin_amdgcn_readfirstlane(plan[2 * kMaxSegs]);
    const int T = my_tiles * sh.nct * (sh.ns + sh.nn);              // k slices of this block
    const int first_tile = plan_first(sh, a, 0);
    if (my_tiles == 0) return;

    if (tid >= kConsumerThreads) {
        Producer<NPT, CH, BIG, SPLIT> pr(a, sh, st, hrow, crow, tid - kConsumerThreads);
        typename Producer<NPT, CH, BIG, SPLIT>::Slice s0, s1;      // slices of even / odd index
        Cursor cl{first_tile, 0, 0, 0, 0, 1, 1.f, 0, 0};      // next slice to load
        enter_tile(cl, a);
        Cursor cc = cl;                                            // slice that goes to LDS next
        int gl = 0;                                                // index of cl; past the last slice it stays there
        // (the surplus loads at the end are harmless and keep the loop free of branches around loads,
        // which is what lets a wait for one slice leave the next one in flight)
#define WD_NEXT() { if (gl + 1 < T) { advance(cl, sh, a); ++gl; } }
        pr.clean[0] = pr.clean[1] = false;
        pr.ids_issue(cl.tile);
        pr.ids_commit(cl.tile, 0, 0);
#define WD_BEGIN(S_) pr.begin(S_, cl, (cl.local & 1) ? pr.clean[1] : pr.clean[0]); WD_NEXT()
        WD_BEGIN(s0)
        WD_BEGIN(s1)
        int stage = 0;
        STAMP_VARS(t_empty[2] = {0, 0}, t_fin[2] = {0, 0}, t_ids = 0, t_beg[2] = {0, 0}, t_begin = clock64())
        uint32_t use = 0;                                          // times `stage` has been filled before
        // Slice g: wait until the consumers have read what the stage held, move the slice from registers
        // to the stage (its loads were issued two slices ago), then the freed registers take the loads of
        // slice g+2.  At the first slice of a tile the ids of the block's next tile are requested, ahead
        // of the row loads of that slice; one slice later they have arrived and go to the id patch.
        bool ids_wait = false;                                     // id loads in flight, issued in the previous slice
        int next_tile = n_tiles;                                   // the tile after cc's, looked up at a tile's first slice
#define WD_PRODUCE(S_, O_)                                                                          \
            STAMP(tf0) pr.finish(S_, stage, O_.cnt, ids_wait);                                      \
            flag_add(full + stage);                                                                 \
            STAMP_ADD(t_fin[cc.side], tf0) STAMP(ti0)                                               \
            if (ids_wait) { pr.ids_commit(next_tile, (cc.local + 1) & 1, O_.cnt); ids_wait = false; } \
            if (stage_ids) { pr.ids_issue(next_tile); ids_wait = true; }                            \
            STAMP_ADD(t_ids, ti0) STAMP_VARS(bside = cl.side) STAMP(tb0)                            \
            WD_BEGIN(S_)                                                                            \
            STAMP_ADD(t_beg[bside], tb0)
#define WD_TICK(S_, O_)                                                                             \
        {                                                                                           \
            const bool tile_begins = cc.ct == 0 && cc.side == 0 && cc.ks == 0;                      \
            if (tile_begins) next_tile = plan_next(cc.tile, sh, a);                                 \
            const bool stage_ids = tile_begins && next_tile < n_tiles;                              \
            STAMP(t0s)                                                                              \
            if (use && !flag_wait(empty + stage, use * kConsumerWaves, flag_peek(empty + stage))) stalled = true; \
            STAMP_ADD(t_empty[cc.side], t0s)                                                        \
            WD_PRODUCE(S_, O_)                                                                      \
            advance(cc, sh, a);                                                                     \
            if (++stage == kStages) { stage = 0; ++use; }                                           \
        }
        for (int g = 0; g < T; g += 2) {
            WD_TICK(s0, s1)
            if (g + 1 < T) WD_TICK(s1, s0)
        }
#undef WD_TICK
#undef WD_PRODUCE
#undef WD_BEGIN
#undef WD_NEXT
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the surplus loads
        if (stalled && (tid & 63) == 0) a.out[0] = __builtin_nanf("");
#ifdef GS_WIDE_STAMP
        if ((tid & 63) == 0 && (blockIdx.x == 0 || blockIdx.x == 100) && (tid >> 6) == 9)
            printf("blk %d producer: total %lld  wait-empty %lld/%lld  finish %lld/%lld  ids %lld  begin %lld/%lld  (own rows/neighbors) slices %d\n",
                   (int)blockIdx.x, clock64() - t_begin, t_empty[0], t_empty[1], t_fin[0], t_fin[1], t_ids, t_beg[0], t_beg[1], T);
#endif
        return;
    }

    // ---- consumers ----
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // the wave index as a scalar
    const int wave_m = wave & 1, wave_n = wave >> 1;
    const int r = lane & 15, q = lane >> 4;
    const int a_slot0 = q * 16 + ((r + 2 * q) & 15), a_slot1 = q * 16 + ((r + 2 * (q + 4)) & 15);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // bias, activation, store of a finished 128x128 product (this wave's 64x64 quarter), 16 rows at a time
    auto epilogue = [&](const Cursor &d) {
        // (lane-dependent offsets are derived here from a value the compiler cannot see through: hoisted
        // out of the slice loop they would occupy two dozen registers that the loop has no room for)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int r = lane_e & 15, q = lane_e >> 4;
        float *E = ep + wave * 16 * EP_LD;
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

    if constexpr (SPLIT) {
        // ---- consumers, bf16 pieces on v_mfma_f32_16x16x32_bf16 ----
        // Lane (r, q) of the A operand holds row r, k = 8q .. 8q+7 of a 16 x 32 tile: one 16-byte LDS read per
        // (row tile, plane), all twelve of a slice resident.  The B operand is column r, the same k: 16 bytes of a
        // transposed weight plane, column 4r + j for column tile j (so that a lane's four accumulators of a row
        // tile are four consecutive columns).  A weight slice -- 128 columns x 32 k x 3 planes = 24 KB, the same
        // for every tile -- comes from the workspace (L2) through LDS: loads queue behind the gathers in this
        // compute unit's memory pipeline and take microseconds, far longer than registers could cover, so each
        // consumer wave copies a quarter of the NEXT slice's weights into the other of two LDS buffers with
        // loads that write LDS directly (no registers), a whole slice ahead.  The copy's image is the read
        // order: chunk (column half, column tile, plane) = 64 lanes x 16 bytes, so a lane reads back exactly the
        // bytes its namesake lane fetched, free of bank conflicts.  Two counters per buffer order the four
        // waves: parts landed (before anyone reads), readers done (before anyone overwrites).
        const int rowsw = (r >> 2) & 3;
        const int a_lane = (wave_m * 64 + r) * 64 + ((q ^ rowsw) * 16);               // bytes inside a plane
        const uint16_t *wsp = reinterpret_cast<const uint16_t *>(a.split_ws);
        const unsigned long long ws_side[2] = {reinterpret_cast<unsigned long long>(wsp),
                                               reinterpret_cast<unsigned long long>(wsp + (size_t)3 * HH * Fs)};   // planes of W_self, of W_neigh
        const char *bring = reinterpret_cast<const char *>(ep);                      // two buffers of kBSlice bytes
        constexpr int kBSlice = 3 * W_PLANE_B;                                       // 24 KB
        const uint32_t bring_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)reinterpret_cast<uintptr_t>(ep));
        // this wave's six chunks of a slice: column half wave >> 1, column tiles 2 (wave & 1) and + 1, planes 0 .. 2
        const int cp_wn = wave >> 1, cp_j0 = 2 * (wave & 1);
        auto issue_b = [&](const Cursor &cur, int buf) {
            const int F_ = cur.side ? Fn : Fs;
            // [plane][column][k]: 16 bytes at (column, 8q)
            const unsigned long long src0 = ws_side[cur.side] +
                2ull * ((unsigned long long)(cur.ct * WN + cp_wn * 64 + 4 * r + cp_j0) * F_ + cur.ks * WK + 8 * q);
            const unsigned long long plane = 2ull * HH * F_, col = 2ull * F_;
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const unsigned long long src = src0 + (t / 3) * col + (t % 3) * plane;
                const uint32_t dst = bring_lds + buf * kBSlice + ((cp_wn * 4 + cp_j0 + t / 3) * 3 + t % 3) * 1024;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
            }
        };
        bf16x8 aP[3][4], bA[3], bB[3];
#define WS_READ_B(dst_, buf_, j_)                                                                    \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                             \
            dst_[pl] = *reinterpret_cast<const bf16x8 *>(bring + (buf_) * kBSlice + ((wave_n * 4 + (j_)) * 3 + pl) * 1024 + lane * 16);
        // nine piece products per (row tile, column tile); the smallest first
#define WS_MFMA(j_, b_)                                                                              \
        _Pragma("unroll") for (int pa = 2; pa >= 0; --pa)                                            \
            _Pragma("unroll") for (int pb = 2; pb >= 0; --pb)                                        \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                        \
                    acc[i][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aP[pa][i], b_[pb], acc[i][j_], 0, 0, 0);

        // bias, activation and store of this wave's 64 x 64 quarter straight from the accumulators: register e of
        // accumulator (i, j) is row 16i + 4q + e, column 4r + j, so the four accumulators of a row tile give a
        // lane four consecutive columns: one 16-byte store per (row tile, e), 256 bytes of each of four rows
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        auto load_bias = [&](const Cursor &d) {
            if (a.bias) {
                const int colbase = d.side * HH + d.ct * WN + wave_n * 64;
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[j] = a.bias[colbase + 4 * r + j];
            }
        };
        auto store_quarter = [&](const Cursor &d) {
            const Seg &sg = a.seg[find_seg(a, d.tile)];
            const long long p0 = (long long)(d.tile - (int)sg.tile0) * WM;
            const int np = (int)min((long long)WM, sg.n_heads - p0);
            const int colbase = d.side * HH + d.ct * WN + wave_n * 64;
            float *__restrict__ outp = a.out + (sg.out_row0 + p0 + wave_m * 64) * H + colbase;
            const uint32_t lane_off = (uint32_t)(4 * q * H + r * 4);
            const int row_q = wave_m * 64 + 4 * q;
            const bool relu = a.act == GS_ACT_RELU;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f32x4 v = {acc[i][0][e] + bv[0], acc[i][1][e] + bv[1], acc[i][2][e] + bv[2], acc[i][3][e] + bv[3]};
                    if (relu) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                    }
                    if (row_q + i * 16 + e < np)
                        *reinterpret_cast<f32x4 *>(outp + (long long)(i * 16 + e) * H + lane_off) = v;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };

        Cursor c{first_tile, 0, 0, 0, 0, 1, 1.f, 0, 0};
        Cursor cn = c;                                              // the slice after c (the last one stays put)
        issue_b(c, 0);
        if (T > 1) advance(cn, sh, a);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        flag_add(bfull);
        if (!flag_wait(full, kProducerWaves, flag_peek(full))) stalled = true;
        int stage = 0;
        uint32_t use = 1;
        STAMP_VARS(t_full[2] = {0, 0}, t_epi = 0, t_b = 0, t_begin = clock64())
        for (int g = 0; g < T; ++g) {
            const char *S = reinterpret_cast<const char *>(st + stage * W_TILE_S) + a_lane;
            const int buf = g & 1, nbuf = buf ^ 1;
            int nstage = stage + 1;
            uint32_t nuse = use;
            if (nstage == kStages) { nstage = 0; ++nuse; }
            // the next slice's weights into the other buffer, once its readers (slice g - 1) are done
            STAMP(t0b)
            if (g + 1 < T) {
                const uint32_t reads = (uint32_t)((g + 1) >> 1);                    // earlier slices that used it
                if (reads && !flag_wait(bempty + nbuf, reads * kConsumerWaves, flag_peek(bempty + nbuf))) stalled = true;
                issue_b(cn, nbuf);
            }
            if (c.ks == 0) load_bias(c);
            // this slice's weights: all four parts landed
            if (!flag_wait(bfull + buf, (uint32_t)((g >> 1) + 1) * kConsumerWaves, flag_peek(bfull + buf))) stalled = true;
            STAMP_ADD(t_b, t0b)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int i = 0; i < 4; ++i) aP[pl][i] = *reinterpret_cast<const bf16x8 *>(S + pl * W_PLANE_B + i * 1024);
            flag_add(empty + stage);                               // all reads of the stage are issued
            const uint32_t seen = flag_peek(full + nstage);
            WS_READ_B(bA, buf, 0)
            __builtin_amdgcn_sched_barrier(0);
            WS_READ_B(bB, buf, 1)
            WS_MFMA(0, bA)
            __builtin_amdgcn_sched_barrier(0);
            WS_READ_B(bA, buf, 2)
            WS_MFMA(1, bB)
            __builtin_amdgcn_sched_barrier(0);
            WS_READ_B(bB, buf, 3)
            flag_add(bempty + buf);                                // all reads of the weight buffer are issued
            WS_MFMA(2, bA)
            __builtin_amdgcn_sched_barrier(0);
            WS_MFMA(3, bB)
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < T) {                                       // this wave's part of the next slice's weights has landed
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                flag_add(bfull + nbuf);
            }
            if (c.ks + 1 == (c.side ? sh.nn : sh.ns)) {
                STAMP(t0e)
                store_quarter(c);
                STAMP_ADD(t_epi, t0e)
            }
            STAMP(t0f)
            if (g + 1 < T && !flag_wait(full + nstage, nuse * kProducerWaves, seen)) stalled = true;
            STAMP_ADD(t_full[cn.side], t0f)
            advance(c, sh, a);
            if (g + 2 < T) advance(cn, sh, a);
            stage = nstage;
            use = nuse;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (stalled && lane == 0) a.out[0] = __builtin_nanf("");
#ifdef GS_WIDE_STAMP
        if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 100) && wave == 1)
            printf("blk %d consumer (pieces): total %lld  wait-full %lld/%lld  weights %lld  epilogue %lld  slices %d\n", (int)blockIdx.x,
                   clock64() - t_begin, t_full[0], t_full[1], t_b, t_epi, T);
#endif
#undef WS_READ_B
#undef WS_MFMA
        return;
    }

    // a##S##i: row tile i, element st = k step (k = 16 half + 4q + st); b##S##st: k step st, element j =
    // column tile (columns 4r + j)
    f32x4 aX0, aX1, aX2, aX3, aY0, aY1, aY2, aY3;
    f32x4 bX0, bX1, bX2, bX3, bY0, bY1, bY2, bY3;
#define WD_READ_A(S_, base_, half_, slot_)                                                          \
    {                                                                                               \
        const float *As_r = (base_) + ((half_) * 8 + wave_m * 4) * 256 + (slot_) * 4;               \
        a##S_##0 = *reinterpret_cast<const f32x4 *>(As_r);                                          \
        a##S_##1 = *reinterpret_cast<const f32x4 *>(As_r + 256);                                    \
        a##S_##2 = *reinterpret_cast<const f32x4 *>(As_r + 512);                                    \
        a##S_##3 = *reinterpret_cast<const f32x4 *>(As_r + 768);                                    \
    }
#define WD_MFMA_ROW(i_, av_, bv_, st_)                                                              \
    acc[i_][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[st_], bv_[0], acc[i_][0], 0, 0, 0);       \
    acc[i_][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[st_], bv_[1], acc[i_][1], 0, 0, 0);       \
    acc[i_][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[st_], bv_[2], acc[i_][2], 0, 0, 0);       \
    acc[i_][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[st_], bv_[3], acc[i_][3], 0, 0, 0);
#define WD_MFMA_STEP(S_, st_)                                                                       \
    WD_MFMA_ROW(0, a##S_##0, b##S_##st_, st_) WD_MFMA_ROW(1, a##S_##1, b##S_##st_, st_)             \
    WD_MFMA_ROW(2, a##S_##2, b##S_##st_, st_) WD_MFMA_ROW(3, a##S_##3, b##S_##st_, st_)
    // Weight fragment of k step st of a half slice: W[ks * 32 + 16 half + 4q + st][column tile + 64 wave_n + 4r .. + 3],
    // 16 lanes on 256 consecutive bytes of each of four weight rows.
    const float *Wb;                                                // this lane's piece of k step 0 of the half being loaded
#define WD_B_BASE(cur_, half_)                                                                      \
    Wb = ((cur_).side ? a.W_neigh : a.W_self) + (long long)((cur_).ks * WK + (half_) * 16 + 4 * q) * HH + (cur_).ct * WN + wave_n * 64 + 4 * r;
#define WD_LOAD_B(S_, st_) b##S_##st_ = *reinterpret_cast<const f32x4 *>(Wb + (st_) * HH);
    // one k step of a half, then the reload of its weight register for the next slice
#define WD_STEP(S_, st_) { WD_MFMA_STEP(S_, st_) __builtin_amdgcn_sched_barrier(0); WD_LOAD_B(S_, st_) __builtin_amdgcn_sched_barrier(0); }

    Cursor c{first_tile, 0, 0, 0, 0, 1, 1.f, 0, 0};
    Cursor cn = c;                                                  // the slice after c (the last one stays put)
    WD_B_BASE(c, 0) WD_LOAD_B(X, 0) WD_LOAD_B(X, 1) WD_LOAD_B(X, 2) WD_LOAD_B(X, 3)
    WD_B_BASE(c, 1) WD_LOAD_B(Y, 0) WD_LOAD_B(Y, 1) WD_LOAD_B(Y, 2) WD_LOAD_B(Y, 3)
    if (T > 1) advance(cn, sh, a);
    if (!flag_wait(full, kProducerWaves, flag_peek(full))) stalled = true;
    WD_READ_A(X, st, 0, a_slot0);
    int stage = 0;
    STAMP_VARS(t_full[2] = {0, 0}, t_epi = 0, t_begin = clock64())
    uint32_t use = 1;                                               // times the stage of slice g has been filled, this one included
    for (int g = 0; g < T; ++g) {
        const float *S = st + stage * W_TILE;
        int nstage = stage + 1;
        uint32_t nuse = use;
        if (nstage == kStages) { nstage = 0; ++nuse; }
        WD_READ_A(Y, S, 1, a_slot1);
        flag_add(empty + stage);                                   // all reads of the stage are issued
        const uint32_t seen = flag_peek(full + nstage);            // (answered while the first half is multiplied)
        __builtin_amdgcn_sched_barrier(0);
        WD_B_BASE(cn, 0)
        WD_STEP(X, 0) WD_STEP(X, 1) WD_STEP(X, 2) WD_STEP(X, 3)
        if (g + 1 < T) {
            STAMP(t0f)
            if (!flag_wait(full + nstage, nuse * kProducerWaves, seen)) stalled = true;
            STAMP_ADD(t_full[cn.side], t0f)
            WD_READ_A(X, st + nstage * W_TILE, 0, a_slot0);
        }
        __builtin_amdgcn_sched_barrier(0);
        WD_B_BASE(cn, 1)
        WD_STEP(Y, 0) WD_STEP(Y, 1) WD_STEP(Y, 2) WD_STEP(Y, 3)
        if (c.ks + 1 == (c.side ? sh.nn : sh.ns)) {
            STAMP(t0e)
            epilogue(c);
            STAMP_ADD(t_epi, t0e)
        }
        advance(c, sh, a);
        if (g + 2 < T) advance(cn, sh, a);
        stage = nstage;
        use = nuse;
    }
    if (stalled && lane == 0) a.out[0] = __builtin_nanf("");
#ifdef GS_WIDE_STAMP
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 100) && wave == 1)
        printf("blk %d consumer: total %lld  wait-full %lld/%lld  epilogue %lld  slices %d\n", (int)blockIdx.x, clock64() - t_begin,
               t_full[0], t_full[1], t_epi, T);
#endif
#undef WD_READ_A
#undef WD_MFMA_ROW
#undef WD_MFMA_STEP
#undef WD_B_BASE
#undef WD_LOAD_B
#undef WD_STEP
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

inline size_t wide_lds_bytes(int max_child, bool split, int stages) {
    return (size_t)(stages * (split ? W_TILE_S : W_TILE) + (split ? 2 * W_TILE_S : 4 * 16 * EP_LD)) * sizeof(float) +
           (size_t)(2 * kMaxStages + 4 + 2 * kMaxSegs + 4 + 2 * WM + 2 * WM * max_child) * sizeof(int32_t);
}
constexpr size_t kWideLdsMax = 160 * 1024;
// fp32 form: four stages.  SPLIT: as many as fit beside the two weight buffers and the id patches (the stages are
// how far the producers run ahead of the product)
inline int wide_stages(int max_child, bool split) {
    if (!split) return 4;
    int n = 2;
    while (n < kMaxStages && wide_lds_bytes(max_child, true, n + 1) <= kWideLdsMax) ++n;
    return n;
}

// W [F][HH] fp32 -> three bf16 planes [plane][HH][F] (transposed: a lane of the product reads eight consecutive k
// of one column), plane 0 the leading piece.  One thread per element; the reads are coalesced, the 2-byte writes
// are not (a few hundred KB once per call).
__global__ __launch_bounds__(256)
void wide_split_weights_kernel(const float *__restrict__ W, int F, int HH, uint16_t *__restrict__ planes) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)F * HH) return;
    const int k = (int)(idx / HH), n = (int)(idx % HH);
    uint32_t hi, mid, lo;
    split3(W[idx], hi, mid, lo);
    const size_t at = (size_t)n * F + k, plane = (size_t)HH * F;
    planes[at] = (uint16_t)(hi >> 16);
    planes[plane + at] = (uint16_t)(mid >> 16);
    planes[2 * plane + at] = (uint16_t)(lo >> 16);
}

}  // namespace

bool gs::sage_wide_supported(int Fs, int Fn, int H, int max_child) {
    const int hh = H / 2;
    return H > 0 && H % 2 == 0 && hh % WN == 0 && Fs >= 2 * WK && Fn >= 2 * WK && Fs % WK == 0 && Fn % WK == 0 &&
           max_child >= 1 && max_child <= kMaxChild;
}

size_t gs::sage_wide_workspace_bytes(int Fs, int Fn, int H) {
    return (size_t)3 * (H / 2) * ((size_t)Fs + Fn) * sizeof(uint16_t);
}

namespace {
using namespace gs;

// The blocks' shares of the tiles.  A tile's time grows with the children it gathers per head -- measured on 256-wide
// rows: 490 / 680 / 1179 us for 2048 tiles of 5 / 10 / 25 children, i.e. proportional to 9 + children -- so tiles
// dealt round-robin leave the blocks that drew the few expensive tiles (the seeds' segment: 25 children) behind
// the others at the end of the launch.  Here every segment, the most expensive
// first, is handed out in rounds to the blocks that have the least work so far; blocks with equal shares form a
// class (at most one new class per segment), and a class's part of a segment is dealt round-robin inside the class.
constexpr double kTileFixedCost = 9.0;      // a tile's time in units of one child per head (see above)

int plan_tiles(const FusedArgs &a, int blocks, TilePlan &tp) {
    struct Class { int b0, b1; double load; long long cnt[kMaxSegs]; };
    std::vector<Class> cls(1);
    cls[0] = Class{0, blocks, 0.0, {}};
    int order[kMaxSegs];
    for (int s = 0; s < a.n_segs; ++s) order[s] = s;
    std::stable_sort(order, order + a.n_segs, [&](int x, int y) { return a.seg[x].n_child > a.seg[y].n_child; });
    for (int oi = 0; oi < a.n_segs; ++oi) {
        const int s = order[oi];
        const double cost = kTileFixedCost + a.seg[s].n_child;
        long long left = (a.seg[s].n_heads + WM - 1) / WM;
        while (left > 0) {
            size_t m = 0;
            double lo2 = 1e300, hi = -1e300;
            for (size_t i = 0; i < cls.size(); ++i) {
                if (cls[i].load < cls[m].load) m = i;
                if (cls[i].load > hi) hi = cls[i].load;
            }
            for (size_t i = 0; i < cls.size(); ++i)
                if (i != m && cls[i].load < lo2) lo2 = cls[i].load;
            if (hi - cls[m].load < cost && left >= blocks) {        // level: whole rounds for everybody
                const long long q = left / blocks;
                for (auto &c : cls) { c.cnt[s] += q; c.load += q * cost; }
                left -= q * blocks;
                continue;
            }
            const long long width = cls[m].b1 - cls[m].b0;
            long long k = 1;                                        // rounds until the class catches up with the next one
            if (lo2 < 1e299 && lo2 > cls[m].load) k = std::max(1ll, (long long)((lo2 - cls[m].load) / cost));
            k = std::min(k, left / width);
            if (k >= 1) {
                cls[m].cnt[s] += k;
                cls[m].load += k * cost;
                left -= k * width;
            } else {                                                // fewer tiles than blocks in the class: its first blocks take one more
                Class rest = cls[m];
                rest.b0 = cls[m].b0 + (int)left;
                cls[m].b1 = rest.b0;
                cls[m].cnt[s] += 1;
                cls[m].load += cost;
                cls.insert(cls.begin() + m + 1, rest);
                left = 0;
            }
        }
    }
    if ((int)cls.size() > kMaxTileClasses) return gs::fail(GS_ELIMIT, "gs_sage_fused_fwd: %zu classes of blocks", cls.size());
    tp.n_cls = (int)cls.size();
    long long run[kMaxSegs] = {};
    for (int i = 0; i < tp.n_cls; ++i) {
        tp.cls[i].b0 = cls[i].b0;
        for (int s = 0; s < kMaxSegs; ++s) {
            tp.cls[i].start[s] = (int)run[s];
            tp.cls[i].count[s] = s < a.n_segs ? (int)cls[i].cnt[s] : 0;
            if (s < a.n_segs) run[s] += (long long)(cls[i].b1 - cls[i].b0) * cls[i].cnt[s];
        }
    }
    return GS_OK;
}

}  // namespace

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
    const bool split = a.split_ws != nullptr;
    const void *kern = split ? (big ? reinterpret_cast<const void *>(sage_wide_kernel<kProducerThreads, kChunk, true, true>)
                                    : reinterpret_cast<const void *>(sage_wide_kernel<kProducerThreads, kChunk, false, true>))
                             : (big ? reinterpret_cast<const void *>(sage_wide_kernel<kProducerThreads, kChunk, true, false>)
                                    : reinterpret_cast<const void *>(sage_wide_kernel<kProducerThreads, kChunk, false, false>));
    const hipError_t attr = gs::raise_lds_limit(kern, (int)kWideLdsMax);
    if (attr != hipSuccess) return gs::fail(GS_ELAUNCH, "gs_sage_fused_fwd: %s", hipGetErrorString(attr));
    static const int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        return n;
    }();
    const long long blocks = tiles < n_cu ? tiles : n_cu;          // one block per CU (LDS)
    TilePlan tp{};
    if (int rc = plan_tiles(a, (int)blocks, tp)) return rc;
    const dim3 grid((unsigned)blocks), block(kConsumerThreads + kProducerThreads);
    const int stages = wide_stages(a.max_child, split);
    const size_t lds = wide_lds_bytes(a.max_child, split, stages);
    if (split) {
        if (reinterpret_cast<uintptr_t>(a.split_ws) & 15)
            return gs::fail(GS_EINVAL, "gs_sage_fused_fwd_ws: the workspace must be 16-byte aligned");
        uint16_t *ws = static_cast<uint16_t *>(a.split_ws);
        const int hh = H / 2;
        hipLaunchKernelGGL(wide_split_weights_kernel, dim3((unsigned)(((long long)Fs * hh + 255) / 256)), dim3(256), 0, stream, a.W_self, Fs, hh, ws);
        hipLaunchKernelGGL(wide_split_weights_kernel, dim3((unsigned)(((long long)Fn * hh + 255) / 256)), dim3(256), 0, stream, a.W_neigh, Fn, hh,
                           ws + (size_t)3 * hh * Fs);
        if (big)
            gs::launch_timed(sage_wide_kernel<kProducerThreads, kChunk, true, true>, grid, block, lds, stream, a, tp, Fs, Fn, hh, stages);
        else
            gs::launch_timed(sage_wide_kernel<kProducerThreads, kChunk, false, true>, grid, block, lds, stream, a, tp, Fs, Fn, hh, stages);
    } else if (big)
        gs::launch_timed(sage_wide_kernel<kProducerThreads, kChunk, true, false>, grid, block, lds, stream, a, tp, Fs, Fn, H / 2, stages);
    else
        gs::launch_timed(sage_wide_kernel<kProducerThreads, kChunk, false, false>, grid, block, lds, stream, a, tp, Fs, Fn, H / 2, stages);
    int rc = gs::check_launch("gs_sage_fused_fwd");
    if (rc != GS_OK || !a.normalize) return rc;
    long long nb = (rows + 3) / 4;
    if (nb > 65536) nb = 65536;
    hipLaunchKernelGGL(l2norm_inplace_kernel, dim3((unsigned)nb), dim3(256), 0, stream, a.out, (int64_t)rows, H);
    return gs::check_launch("gs_sage_fused_fwd");
}
