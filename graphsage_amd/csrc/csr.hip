// Adjacency build on the device: edge columns -> CSR with every node's
// neighbors in edge-column order (duplicates kept).
//
// The reference makes one np.where scan of the whole edge array per node
// (bipartite_graph.py:53-60), from an edge array that already holds both
// directions (train_movielens.py:130-132).  Here the edge array is taken in
// chunks; a chunk's sources are radix-sorted (hipCUB), which turns every node's
// entries in the chunk into one run.  Pass 1 adds the run lengths into the
// degrees (a popular item would serialise an atomic histogram: one address, tens
// of millions of increments), a scan turns degrees into row starts, pass 2 sorts
// (source, destination) pairs stably and appends each run to its node's row
// through a per-node cursor.  No atomics anywhere, so the result is the same on
// every run and on every GPU.
#include "gs_common.h"

#include <hipcub/hipcub.hpp>

namespace gs {

constexpr int64_t kSortChunk = int64_t(1) << 30;   // entries per radix-sort call
constexpr int kCsrBlock = 256;

struct CsrPlan {
    int64_t chunk;        // entries per sort call
    size_t sort_bytes;    // hipCUB temporary storage for one call
    size_t off_cursor, off_lstart, off_keys, off_vals, off_sort, total;
};

static inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

static int make_plan(int64_t n_half, int64_t n_nodes, int64_t sort_chunk, CsrPlan *p) {
    const int64_t cap = sort_chunk > 0 ? sort_chunk : kSortChunk;
    p->chunk = n_half < cap ? (n_half > 0 ? n_half : 1) : cap;
    size_t sort_bytes = 0, scan_bytes = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (const int32_t *)nullptr, (int32_t *)nullptr,
                                                      (const int32_t *)nullptr, (int32_t *)nullptr, p->chunk, 0, 32,
                                                      (hipStream_t)0);
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_csr_build: radix sort size query: %s", hipGetErrorString(e));
    size_t keys_bytes = 0;
    e = hipcub::DeviceRadixSort::SortKeys(nullptr, keys_bytes, (const int32_t *)nullptr, (int32_t *)nullptr, p->chunk, 0,
                                          32, (hipStream_t)0);
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_csr_build: radix sort size query: %s", hipGetErrorString(e));
    if (keys_bytes > sort_bytes) sort_bytes = keys_bytes;
    e = hipcub::DeviceScan::InclusiveSum(nullptr, scan_bytes, (int64_t *)nullptr, (int64_t *)nullptr, int(n_nodes),
                                         (hipStream_t)0);
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_csr_build: scan size query: %s", hipGetErrorString(e));
    p->sort_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;   // one region serves both
    size_t off = 0;
    p->off_cursor = off; off += align256(size_t(n_nodes) * 8);
    p->off_lstart = off; off += align256(size_t(n_nodes) * 4);
    p->off_keys = off;   off += align256(size_t(p->chunk) * 4);
    p->off_vals = off;   off += align256(size_t(p->chunk) * 4);
    p->off_sort = off;   off += align256(p->sort_bytes);
    p->total = off;
    return GS_OK;
}

// Flags ids outside [0, n_nodes).
__global__ __launch_bounds__(kCsrBlock) void csr_check_kernel(const int32_t *__restrict__ a,
                                                              const int32_t *__restrict__ b, int64_t n_pairs,
                                                              int64_t n_nodes, int32_t *__restrict__ err_flag) {
    const int64_t stride = int64_t(gridDim.x) * kCsrBlock;
    bool bad = false;
    for (int64_t i = int64_t(blockIdx.x) * kCsrBlock + threadIdx.x; i < n_pairs; i += stride) {
        const int32_t s = a[i], d = b[i];
        bad |= !(s >= 0 && s < n_nodes && d >= 0 && d < n_nodes);
    }
    if (bad) atomicMax(err_flag, 1);
}

// Pass 1, after csr_run_starts_kernel on a chunk of sorted sources: the last entry of every
// run adds the run length to its node's degree (one run per node per chunk, chunks in stream order).
__global__ __launch_bounds__(kCsrBlock) void csr_add_runs_kernel(const int32_t *__restrict__ keys, int64_t n,
                                                                 const int32_t *__restrict__ lstart,
                                                                 int64_t *__restrict__ deg,
                                                                 const int32_t *__restrict__ err_flag) {
    const int64_t j = int64_t(blockIdx.x) * kCsrBlock + threadIdx.x;
    if (j >= n || *err_flag) return;
    const int32_t k = keys[j];
    if (j == n - 1 || keys[j + 1] != k) deg[k] += (j - lstart[k]) + 1;
}

// keys: one sorted chunk.  The first entry of every run records where the run starts.
__global__ __launch_bounds__(kCsrBlock) void csr_run_starts_kernel(const int32_t *__restrict__ keys, int64_t n,
                                                                   int32_t *__restrict__ lstart,
                                                                   const int32_t *__restrict__ err_flag) {
    const int64_t j = int64_t(blockIdx.x) * kCsrBlock + threadIdx.x;
    if (j >= n || *err_flag) return;                    // bad ids were seen: nothing is placed
    const int32_t k = keys[j];
    if (j == 0 || keys[j - 1] != k) lstart[k] = int32_t(j);
}

// Appends the chunk's runs to the rows: entry j of node k lands at cursor[k] + (j - run start).
__global__ __launch_bounds__(kCsrBlock) void csr_place_kernel(const int32_t *__restrict__ keys,
                                                              const int32_t *__restrict__ vals, int64_t n,
                                                              const int32_t *__restrict__ lstart,
                                                              const int64_t *__restrict__ cursor,
                                                              int32_t *__restrict__ indices,
                                                              const int32_t *__restrict__ err_flag) {
    const int64_t j = int64_t(blockIdx.x) * kCsrBlock + threadIdx.x;
    if (j >= n || *err_flag) return;
    const int32_t k = keys[j];
    indices[cursor[k] + (j - lstart[k])] = vals[j];
}

// After placing: the last entry of every run moves its node's cursor past the run.
__global__ __launch_bounds__(kCsrBlock) void csr_advance_kernel(const int32_t *__restrict__ keys, int64_t n,
                                                                const int32_t *__restrict__ lstart,
                                                                int64_t *__restrict__ cursor,
                                                                const int32_t *__restrict__ err_flag) {
    const int64_t j = int64_t(blockIdx.x) * kCsrBlock + threadIdx.x;
    if (j >= n || *err_flag) return;
    const int32_t k = keys[j];
    if (j == n - 1 || keys[j + 1] != k) cursor[k] += (j - lstart[k]) + 1;
}

}  // namespace gs

extern "C" int gs_csr_build_workspace(int64_t n_pairs, int64_t n_nodes, int64_t sort_chunk, size_t *bytes) {
    using namespace gs;
    if (!bytes) return fail(GS_EINVAL, "gs_csr_build_workspace: null pointer argument");
    if (n_pairs < 0 || n_nodes < 0 || n_nodes > INT32_MAX)
        return fail(GS_EINVAL, "gs_csr_build_workspace: bad sizes (pairs %lld, nodes %lld)", (long long)n_pairs,
                    (long long)n_nodes);
    CsrPlan p;
    if (int rc = make_plan(n_pairs, n_nodes, sort_chunk, &p)) return rc;
    *bytes = p.total;
    return GS_OK;
}

extern "C" int gs_csr_build(const int32_t *a, const int32_t *b, int64_t n_pairs, int symmetric, int64_t n_nodes,
                            int64_t *indptr, int32_t *indices, int32_t *err_flag, int64_t sort_chunk, void *workspace,
                            size_t ws_bytes, void *stream_) {
    using namespace gs;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_pairs < 0 || n_nodes < 0 || n_nodes > INT32_MAX)
        return fail(GS_EINVAL, "gs_csr_build: bad sizes (pairs %lld, nodes %lld)", (long long)n_pairs,
                    (long long)n_nodes);
    if (!indptr || !err_flag) return fail(GS_EINVAL, "gs_csr_build: null pointer argument");
    hipError_t e = hipMemsetAsync(indptr, 0, size_t(n_nodes + 1) * 8, stream);
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_csr_build: memset: %s", hipGetErrorString(e));
    if (n_pairs == 0 || n_nodes == 0) return GS_OK;
    if (!a || !b || !indices || !workspace) return fail(GS_EINVAL, "gs_csr_build: null pointer argument");
    CsrPlan p;
    if (int rc = make_plan(n_pairs, n_nodes, sort_chunk, &p)) return rc;
    if (ws_bytes < p.total)
        return fail(GS_EINVAL, "gs_csr_build: workspace of %zu bytes, %zu needed", ws_bytes, p.total);

    char *ws = static_cast<char *>(workspace);
    int64_t *cursor = reinterpret_cast<int64_t *>(ws + p.off_cursor);
    int32_t *lstart = reinterpret_cast<int32_t *>(ws + p.off_lstart);
    int32_t *keys = reinterpret_cast<int32_t *>(ws + p.off_keys);
    int32_t *vals = reinterpret_cast<int32_t *>(ws + p.off_vals);
    void *sort_ws = ws + p.off_sort;

    int end_bit = 1;
    while (end_bit < 31 && (int64_t(1) << end_bit) < n_nodes) ++end_bit;
    const int n_halves = symmetric ? 2 : 1;

    {
        const int64_t want = (n_pairs + kCsrBlock - 1) / kCsrBlock;
        const int grid = int(want < 256 * 32 ? want : 256 * 32);
        csr_check_kernel<<<grid, kCsrBlock, 0, stream>>>(a, b, n_pairs, n_nodes, err_flag);
        if (int rc = check_launch("gs_csr_build: check")) return rc;
    }
    // pass 1: degrees into indptr[1..] from the runs of each sorted chunk of sources
    for (int half = 0; half < n_halves; ++half) {
        const int32_t *src = half ? b : a;
        for (int64_t c0 = 0; c0 < n_pairs; c0 += p.chunk) {
            const int64_t n = n_pairs - c0 < p.chunk ? n_pairs - c0 : p.chunk;
            size_t sb = p.sort_bytes;
            e = hipcub::DeviceRadixSort::SortKeys(sort_ws, sb, src + c0, keys, n, 0, end_bit, stream);
            if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_csr_build: radix sort: %s", hipGetErrorString(e));
            const int grid = int((n + kCsrBlock - 1) / kCsrBlock);
            csr_run_starts_kernel<<<grid, kCsrBlock, 0, stream>>>(keys, n, lstart, err_flag);
            csr_add_runs_kernel<<<grid, kCsrBlock, 0, stream>>>(keys, n, lstart, indptr + 1, err_flag);
            if (int rc = check_launch("gs_csr_build: count")) return rc;
        }
    }
    // an inclusive scan in place turns degrees into row ends = next row starts
    {
        size_t scan_bytes = p.sort_bytes;
        e = hipcub::DeviceScan::InclusiveSum(sort_ws, scan_bytes, indptr + 1, indptr + 1, int(n_nodes), stream);
        if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_csr_build: scan: %s", hipGetErrorString(e));
        e = hipMemcpyAsync(cursor, indptr, size_t(n_nodes) * 8, hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_csr_build: cursor copy: %s", hipGetErrorString(e));
    }

    // pass 2, entries in column order: the pairs as given, then (for symmetric input) the flipped pairs
    for (int half = 0; half < n_halves; ++half) {
        const int32_t *src = half ? b : a, *dst = half ? a : b;
        for (int64_t c0 = 0; c0 < n_pairs; c0 += p.chunk) {
            const int64_t n = n_pairs - c0 < p.chunk ? n_pairs - c0 : p.chunk;
            size_t sb = p.sort_bytes;
            e = hipcub::DeviceRadixSort::SortPairs(sort_ws, sb, src + c0, keys, dst + c0, vals, n, 0, end_bit, stream);
            if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_csr_build: radix sort: %s", hipGetErrorString(e));
            const int grid = int((n + kCsrBlock - 1) / kCsrBlock);
            csr_run_starts_kernel<<<grid, kCsrBlock, 0, stream>>>(keys, n, lstart, err_flag);
            csr_place_kernel<<<grid, kCsrBlock, 0, stream>>>(keys, vals, n, lstart, cursor, indices, err_flag);
            csr_advance_kernel<<<grid, kCsrBlock, 0, stream>>>(keys, n, lstart, cursor, err_flag);
            if (int rc = check_launch("gs_csr_build: place")) return rc;
        }
    }
    return GS_OK;
}
