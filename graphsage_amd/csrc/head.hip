// Link head, loss and optimizer update of the training step.
//
//   pred[b] = sigmoid(<concat(user_emb[b], item_emb[b]), kernel> + bias)
//   loss    = mean_b (pred[b] - y[b])^2
// and, in the same pass, the gradients of the loss with respect to both
// embeddings, the kernel and the bias.  One wave walks rows; lane c owns columns
// c, c+64, ... of the concatenated embedding, the dot product is an xor-shuffle
// tree.  Kernel/bias/loss partial sums leave each block as one slab row and are
// added in block order by a second launch: bitwise reproducible, no atomics.
//
// gs_adam_step is the Keras/TF2 Adam update on flat parameter storage.
#include "gs_common.h"

#include <atomic>

namespace {

constexpr int kThreads = 256;
constexpr int kMaxCols = 8;                 // columns per lane: 2E <= 512

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <bool TRAIN>
__global__ __launch_bounds__(kThreads)
void link_head_kernel(const float *__restrict__ eu, const float *__restrict__ ei, int64_t B, int E,
                      const float *__restrict__ kernel, const float *__restrict__ bias,
                      const float *__restrict__ y, float *__restrict__ pred,
                      float *__restrict__ g_eu, float *__restrict__ g_ei, float *__restrict__ slab) {
    __shared__ float red[kThreads / 64][kMaxCols * 64 + 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = 2 * E;
    float kv[kMaxCols], gk[kMaxCols];
#pragma unroll
    for (int i = 0; i < kMaxCols; ++i) {
        const int c = lane + 64 * i;
        kv[i] = c < C ? kernel[c] : 0.f;
        gk[i] = 0.f;
    }
    const float b0 = bias[0];
    float gb = 0.f, loss = 0.f;
    const float inv_b = 1.f / (float)B;
    const int64_t stride = (int64_t)gridDim.x * (kThreads / 64);
    for (int64_t row = (int64_t)blockIdx.x * (kThreads / 64) + wave; row < B; row += stride) {
        float x[kMaxCols];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < kMaxCols; ++i) {
            const int c = lane + 64 * i;
            x[i] = 0.f;
            if (c < C) x[i] = c < E ? eu[row * E + c] : ei[row * E + (c - E)];
            dot += x[i] * kv[i];
        }
        const float z = wave_sum(dot) + b0;
        const float p = 1.f / (1.f + __expf(-z));
        if (lane == 0) pred[row] = p;
        if (TRAIN) {
            const float d = p - y[row];
            const float gz = 2.f * d * inv_b * p * (1.f - p);
            loss += d * d;                   // identical in every lane; lane 0's copy is used
            gb += gz;
#pragma unroll
            for (int i = 0; i < kMaxCols; ++i) {
                const int c = lane + 64 * i;
                if (c < C) {
                    gk[i] += gz * x[i];
                    const float gx = gz * kv[i];
                    if (c < E) g_eu[row * E + c] = gx; else g_ei[row * E + (c - E)] = gx;
                }
            }
        }
    }
    if (TRAIN) {
#pragma unroll
        for (int i = 0; i < kMaxCols; ++i) red[wave][lane + 64 * i] = gk[i];
        if (lane == 0) { red[wave][kMaxCols * 64] = gb; red[wave][kMaxCols * 64 + 1] = loss; }
        __syncthreads();
        float *out = slab + (size_t)blockIdx.x * (C + 2);
        for (int c = threadIdx.x; c < C + 2; c += kThreads) {
            const int src = c < C ? c : kMaxCols * 64 + (c - C);
            out[c] = ((red[0][src] + red[1][src]) + red[2][src]) + red[3][src];
        }
    }
}

// 16 threads per output element take blocks b = j, j+16, ... in increasing order; their partial
// sums are added in index order.
__global__ __launch_bounds__(kThreads)
void link_head_reduce_kernel(const float *__restrict__ slab, int n_blocks, int C, float inv_b,
                             float *__restrict__ g_kernel, float *__restrict__ g_bias, float *__restrict__ loss) {
    __shared__ float part[kThreads];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    const int j = threadIdx.x >> 4;
    float s = 0.f;
    if (c < C + 2)
        for (int b = j; b < n_blocks; b += 16) s += slab[(size_t)b * (C + 2) + c];
    part[threadIdx.x] = s;
    __syncthreads();
    if (j == 0 && c < C + 2) {
        s = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) s += part[g * 16 + threadIdx.x];
        if (c < C) g_kernel[c] = s;
        else if (c == C) g_bias[0] = s;
        else loss[0] = s * inv_b;
    }
}

__global__ __launch_bounds__(kThreads)
void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                 int64_t n, float lr_arg, const float *__restrict__ lr_dev, float b1, float b2, float eps,
                 float grad_scale) {
    const float lr_t = lr_dev ? lr_dev[0] : lr_arg;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        gs::adam_element(p[i], m[i], v[i], g[i] * grad_scale, lr_t, b1, b2, eps);
    }
}

// One step's inputs into the buffers a captured graph reads, and the step's learning rate: four
// small device copies in one launch.
__global__ __launch_bounds__(kThreads)
void stage_batch_kernel(const int32_t *__restrict__ seeds, int32_t *__restrict__ seeds_dst, int64_t n_seeds,
                        const int64_t *__restrict__ walks, int64_t *__restrict__ walks_dst, int64_t n_walks,
                        const float *__restrict__ targets, float *__restrict__ targets_dst, int64_t n_targets,
                        float lr_t, float *__restrict__ lr_dst) {
    const int64_t i0 = (int64_t)blockIdx.x * kThreads + threadIdx.x, stride = (int64_t)gridDim.x * kThreads;
    for (int64_t i = i0; i < n_seeds; i += stride) seeds_dst[i] = seeds[i];
    for (int64_t i = i0; i < n_walks; i += stride) walks_dst[i] = walks[i];
    for (int64_t i = i0; i < n_targets; i += stride) targets_dst[i] = targets[i];
    if (i0 == 0 && lr_dst) lr_dst[0] = lr_t;
}

inline int head_blocks(int64_t B) {
    int64_t b = (B + 15) / 16;               // four waves per block, a few rows per wave
    if (b < 1) b = 1;
    if (b > 256) b = 256;
    return (int)b;
}

}  // namespace

extern "C" size_t gs_link_head_workspace(int64_t B, int E) {
    if (B < 0 || E < 1) return 0;
    return (size_t)head_blocks(B) * (size_t)(2 * E + 2) * sizeof(float);
}

extern "C" int gs_link_head_fwd(const float *emb_user, const float *emb_item, int64_t B, int E,
                                const float *kernel, const float *bias, float *pred, void *stream) {
    if (B < 0 || E < 1) return gs::fail(GS_EINVAL, "gs_link_head_fwd: B=%lld E=%d", (long long)B, E);
    if (2 * E > kMaxCols * 64) return gs::fail(GS_ELIMIT, "gs_link_head_fwd: embedding width %d exceeds %d", E, kMaxCols * 32);
    if (B == 0) return GS_OK;
    if (!emb_user || !emb_item || !kernel || !bias || !pred) return gs::fail(GS_EINVAL, "gs_link_head_fwd: null pointer argument");
    hipLaunchKernelGGL(link_head_kernel<false>, dim3(head_blocks(B)), dim3(kThreads), 0, (hipStream_t)stream,
                       emb_user, emb_item, B, E, kernel, bias, nullptr, pred, nullptr, nullptr, nullptr);
    return gs::check_launch("gs_link_head_fwd");
}

extern "C" int gs_link_head_mse_fwd_bwd(const float *emb_user, const float *emb_item, int64_t B, int E,
                                        const float *kernel, const float *bias, const float *targets,
                                        float *pred, float *loss, float *g_emb_user, float *g_emb_item,
                                        float *g_kernel, float *g_bias,
                                        void *workspace, size_t workspace_bytes, void *stream) {
    if (B < 1 || E < 1) return gs::fail(GS_EINVAL, "gs_link_head_mse_fwd_bwd: B=%lld E=%d", (long long)B, E);
    if (2 * E > kMaxCols * 64) return gs::fail(GS_ELIMIT, "gs_link_head_mse_fwd_bwd: embedding width %d exceeds %d", E, kMaxCols * 32);
    if (!emb_user || !emb_item || !kernel || !bias || !targets || !pred || !loss || !g_emb_user || !g_emb_item ||
        !g_kernel || !g_bias || !workspace)
        return gs::fail(GS_EINVAL, "gs_link_head_mse_fwd_bwd: null pointer argument");
    const size_t need = gs_link_head_workspace(B, E);
    if (workspace_bytes < need) return gs::fail(GS_EINVAL, "gs_link_head_mse_fwd_bwd: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
    const int blocks = head_blocks(B);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(link_head_kernel<true>, dim3(blocks), dim3(kThreads), 0, s, emb_user, emb_item, B, E, kernel, bias,
                       targets, pred, g_emb_user, g_emb_item, static_cast<float *>(workspace));
    if (int rc = gs::check_launch("gs_link_head_mse_fwd_bwd")) return rc;
    hipLaunchKernelGGL(link_head_reduce_kernel, dim3((unsigned)((2 * E + 2 + 15) / 16)), dim3(kThreads), 0, s,
                       static_cast<const float *>(workspace), blocks,
                       2 * E, 1.f / (float)B, g_kernel, g_bias, loss);
    return gs::check_launch("gs_link_head_mse_fwd_bwd (reduce)");
}

extern "C" int gs_link_head_reduce(const void *slab, int n_rows, int E, int64_t B, float *g_kernel, float *g_bias,
                                   float *loss, void *stream) {
    if (n_rows < 1 || E < 1 || B < 1) return gs::fail(GS_EINVAL, "gs_link_head_reduce: n_rows=%d E=%d B=%lld", n_rows, E, (long long)B);
    if (!slab || !g_kernel || !g_bias || !loss) return gs::fail(GS_EINVAL, "gs_link_head_reduce: null pointer argument");
    hipLaunchKernelGGL(link_head_reduce_kernel, dim3((unsigned)((2 * E + 2 + 15) / 16)), dim3(kThreads), 0, (hipStream_t)stream,
                       static_cast<const float *>(slab), n_rows, 2 * E, 1.f / (float)B, g_kernel, g_bias, loss);
    return gs::check_launch("gs_link_head_reduce");
}

// ---- the same step for callers that cannot allocate ---------------------------------------
// The slab of per-block partial sums lives in device memory that belongs to the library's code
// object: every GPU the library runs on holds its own copy at its own address until the process
// ends.  Largest case: 256 blocks x (2 * 256 + 2) floats = 526 336 bytes.
constexpr int kStaticBlocks = 256;                       // head_blocks() never exceeds it
constexpr int kStaticMaxE = 256;                         // = kMaxCols * 32, the widest embedding of the kernels above
constexpr size_t kStaticFloats = (size_t)kStaticBlocks * (2 * kStaticMaxE + 2);

static __device__ __align__(256) float scratch[kStaticFloats];

namespace {

constexpr int kMaxDevices = 64;
std::atomic<float *> g_scratch_ptr[kMaxDevices];         // device address of `scratch`, per GPU

// The array's name is only a handle on the host; its address on the current GPU comes from the
// runtime, once per device (no allocation, nothing enqueued).
int static_scratch(float **out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return gs::fail(GS_ELAUNCH, "gs_link_head_mse_fwd_bwd_static: hipGetDevice: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= kMaxDevices) return gs::fail(GS_ELIMIT, "gs_link_head_mse_fwd_bwd_static: device %d beyond %d", dev, kMaxDevices);
    float *p = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!p) {
        e = hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return gs::fail(GS_ELAUNCH, "gs_link_head_mse_fwd_bwd_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
        g_scratch_ptr[dev].store(p, std::memory_order_release);      // every thread finds the same address
    }
    *out = p;
    return GS_OK;
}

}  // namespace

extern "C" int gs_link_head_mse_fwd_bwd_static(const float *emb_user, const float *emb_item, int64_t B, int E,
                                               const float *kernel, const float *bias, const float *targets,
                                               float *pred, float *loss, float *g_emb_user, float *g_emb_item,
                                               float *g_kernel, float *g_bias, void *stream) {
    if (B < 1 || E < 1) return gs::fail(GS_EINVAL, "gs_link_head_mse_fwd_bwd_static: B=%lld E=%d", (long long)B, E);
    if (E > kStaticMaxE)
        return gs::fail(GS_ELIMIT, "gs_link_head_mse_fwd_bwd_static: embedding width %d exceeds the %d the static scratch "
                                   "(%zu bytes) is sized for", E, kStaticMaxE, kStaticFloats * sizeof(float));
    const int blocks = head_blocks(B);
    if (blocks > kStaticBlocks || (size_t)blocks * (size_t)(2 * E + 2) > kStaticFloats)
        return gs::fail(GS_ELIMIT, "gs_link_head_mse_fwd_bwd_static: %d blocks of %d floats exceed the static scratch", blocks, 2 * E + 2);
    float *slab = nullptr;
    if (int rc = static_scratch(&slab)) return rc;
    // every block writes its whole slab row before the reduction reads rows [0, blocks): nothing of an
    // earlier call is read, so nothing needs clearing
    return gs_link_head_mse_fwd_bwd(emb_user, emb_item, B, E, kernel, bias, targets, pred, loss, g_emb_user, g_emb_item,
                                    g_kernel, g_bias, slab, (size_t)blocks * (size_t)(2 * E + 2) * sizeof(float), stream);
}

extern "C" int gs_adam_step(float *params, const float *grads, float *m, float *v, int64_t n,
                            float lr_t, const float *lr_t_device, float beta1, float beta2, float eps,
                            float grad_scale, void *stream) {
    if (n < 0) return gs::fail(GS_EINVAL, "gs_adam_step: n=%lld", (long long)n);
    if (n == 0) return GS_OK;
    if (!params || !grads || !m || !v) return gs::fail(GS_EINVAL, "gs_adam_step: null pointer argument");
    int64_t blocks = (n + kThreads - 1) / kThreads;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, params, grads, m, v, n,
                       lr_t, lr_t_device, beta1, beta2, eps, grad_scale);
    return gs::check_launch("gs_adam_step");
}

extern "C" int gs_stage_batch(const int32_t *seeds, int32_t *seeds_dst, int64_t n_seeds,
                              const int64_t *walk_ids, int64_t *walk_ids_dst, int64_t n_walk_ids,
                              const float *targets, float *targets_dst, int64_t n_targets,
                              float lr_t, float *lr_t_dst, void *stream) {
    if (n_seeds < 0 || n_walk_ids < 0 || n_targets < 0)
        return gs::fail(GS_EINVAL, "gs_stage_batch: counts %lld %lld %lld", (long long)n_seeds, (long long)n_walk_ids, (long long)n_targets);
    if ((n_seeds && (!seeds || !seeds_dst)) || (n_walk_ids && (!walk_ids || !walk_ids_dst)) || (n_targets && (!targets || !targets_dst)))
        return gs::fail(GS_EINVAL, "gs_stage_batch: null pointer argument");
    int64_t n = n_seeds > n_walk_ids ? n_seeds : n_walk_ids;
    if (n_targets > n) n = n_targets;
    if (n == 0 && !lr_t_dst) return GS_OK;
    int64_t blocks = (n + kThreads - 1) / kThreads;
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(stage_batch_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream,
                       seeds, seeds_dst, n_seeds, walk_ids, walk_ids_dst, n_walk_ids, targets, targets_dst, n_targets,
                       lr_t, lr_t_dst);
    return gs::check_launch("gs_stage_batch");
}
