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

// Wide layers (wide.hip).  `a.seg[].tile0` / `a.n_tiles` are recomputed for 128-head tiles.
bool sage_wide_supported(int Fs, int Fn, int H, int max_child);
int sage_wide_fwd(FusedArgs a, int Fs, int Fn, int H, hipStream_t stream);

}  // namespace gs
