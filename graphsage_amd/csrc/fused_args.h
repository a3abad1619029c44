// Arguments shared by the fused aggregator-layer kernels (fused.hip: narrow layers,
// one wave per 16 heads; wide.hip: wide layers, one block per 128 heads).
#pragma once

#include "gs_common.h"

namespace gs {

constexpr int kMaxSegs = 2 * GS_MAX_HOPS;

struct Seg {
    const int32_t *head_ids;   // node ids of the heads, or NULL: heads are rows self_row0 + p of feat_self
    const int32_t *child_ids;  // node ids of the children [n_heads * n_child], or NULL: rows child_row0 + p*n_child + j
    long long n_heads;
    long long self_row0;
    long long child_row0;
    long long tile0;           // index of the segment's first tile
    long long out_row0;        // first output row of the segment
    int n_child;
    // children drawn inside the kernel (hop >= 0) instead of read from child_ids:
    int hop;                   // the draws are gs_sample's hop `hop` (-1: not drawn)
    int heads_per_walk;        // heads of this segment per walk (= size of the walk's hop `hop`-1, 1 for the roots)
    int tape_off;              // offset of hop `hop` inside one walk's replay tape
    long long walk0;           // index of the walk head 0 belongs to (row of walk_ids / of the tape)
    int32_t *out_child_ids;    // nullable: receives the drawn ids [n_heads * n_child]
};

// What the in-kernel draws read: the adjacency and the keys of gs_sample.
struct Draws {
    const int64_t *indptr;
    const int32_t *indices;
    long long n_nodes;
    uint32_t key0, key1;
    unsigned long long walk_base;
    const int64_t *walk_ids;
    const uint32_t *replay;
    long long tape_len;
    int32_t *err;
};

// Wide layers: which tiles a block takes.  Blocks b0 .. (the next class's b0) - 1 form a class of w blocks; the class
// takes, of segment s, the w * count[s] tiles from start[s] on (tiles numbered inside the segment), dealt round-robin:
// block b takes start[s] + (b - b0) + i * w, i < count[s].
constexpr int kMaxTileClasses = 12;
struct TileClass {
    int b0;
    int start[kMaxSegs], count[kMaxSegs];
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
    Draws draws;               // used when some segment has hop >= 0
    void *split_ws;            // wide layers: workspace of the bf16-piece form of the product (NULL: the fp32 form)
};

struct TilePlan {              // wide layers: the blocks' shares (a kernel argument of its own, set by sage_wide_fwd)
    TileClass cls[kMaxTileClasses];
    int n_cls;
};

// Wide layers (wide.hip).  `a.seg[].tile0` / `a.n_tiles` are recomputed for 128-head tiles.
bool sage_wide_supported(int Fs, int Fn, int H, int max_child);
int sage_wide_fwd(FusedArgs a, int Fs, int Fn, int H, hipStream_t stream);
size_t sage_wide_workspace_bytes(int Fs, int Fn, int H);

}  // namespace gs
