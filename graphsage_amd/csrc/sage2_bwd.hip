// Backward of the top of the two-hop model, one wavefront per root: from the gradients of the
// embeddings down to the gradients of the layer-0 activations, in one launch.
//
//   g_z[t][p]                = l2-normalisation backward of g_emb[t][p] at z[t][p]   (models.py:32)
//   g_h1[t][p]               = (g_z[t][p][:H2/2] . W1_self[t]^T)                      (utils.py:109-119, the root's own input)
//   g_h1[o][B + p*n0 + j]    = (g_z[t][p][H2/2:] . W1_neigh[t]^T) / n0,  j < n0       (utils.py:103-109: the mean over its n0
//                                                                                      hop-1 heads, which are of the other type o)
//
// with the row layout gs_sage2_forward uses for its saved tables (row p: root of pair p of that type;
// row B + p*n0 + j: hop-1 head j of the walk rooted at the other type's root of pair p), so the two
// g_h1 tables are written completely: every row exactly once.  This replaces gs_l2_normalize_bwd, the
// input-gradient half of gs_sage_project_bwd and gs_gather_mean_bwd of the layer-by-layer route -- three
// dependent launches per node type of a few microseconds each -- by one.  The layer-1 weight gradients
// are still taken from g_z by gs_sage_project_bwd.
//
// A row is at most 64 wide, so lane c owns column c: the two 32- or 64-long dot products per lane read
// the weights from LDS, staged once per block transposed ([n][k], lanes along k: conflict-free), and
// g_z from a per-wave LDS line (broadcast reads).
#include "gs_common.h"

namespace {

constexpr int kThreads = 1024;             // sixteen waves share one staged copy of the weights
constexpr int kWaves = kThreads / 64;
constexpr int kMaxH = 64;

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct TailArgs {
    const float *z[2], *g_emb[2], *w_self[2], *w_neigh[2];
    float *g_z[2], *g_h1[2];
    long long n_pairs;
    int n0, H1, H2, normalize, relu;
    // HEAD: the link head and its loss in the same launch (g_emb is then not read)
    const float *emb[2], *head_kernel, *head_bias, *targets;
    float *pred, *head_slab;
    // layer-0 relu masks (or NULL): the roots' saved layer-0 outputs, the hop-1 heads' mask words
    const float *h1[2];
    const uint32_t *h1_mask[2];
    float in_scale;                          // factor on the layer-1 input gradients (dropout's 1 / (1 - rate); 1 otherwise)
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// HEAD: the gradient of the embeddings is not an input but comes from the link head evaluated here,
// pred = sigmoid(<concat(user_emb, item_emb), kernel> + bias) with the mean-squared-error loss
// (link_utils.py:167-168,244-247; train_movielens.py:186): both walks of a pair evaluate the pair's
// score (two 64-byte rows), the walk of type t takes g_emb[t] = gz * kernel[t half] from it and adds
// gz * emb[t] to its lane's partial sum of the kernel gradient's t half; the user-side walk also
// owns the pair's prediction, its share of the loss and of the bias gradient.  A block leaves its
// partial sums as one slab row [2E kernel | bias | loss] (waves added in wave order), and
// gs_link_head_reduce adds the rows in block order.
template <int H1, int HH2, bool HEAD>
__global__ __launch_bounds__(kThreads)
void sage2_tail_bwd_kernel(const TailArgs a) {
    extern __shared__ float smem[];
    constexpr int H2 = 2 * HH2, wsz = HH2 * H1;
    float *wt = smem;                                             // [type][self, neigh][n][k]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *gz_s = smem + 4 * wsz + wave * kMaxH;                  // this wave's g_z row
    const long long walks = 2 * a.n_pairs;
    const long long w_first = (long long)blockIdx.x * kWaves + wave, w_step = (long long)gridDim.x * kWaves;

    // the first walk's rows are requested before the weights, so that the two round trips overlap
    float zc = 0.f, gc = 0.f, eu = 0.f, ei = 0.f, yt = 0.f;
    if (w_first < walks && lane < H2) {
        const int t = w_first >= a.n_pairs;
        const long long p = w_first - (t ? a.n_pairs : 0);
        zc = a.z[t][p * H2 + lane];
        if (HEAD) { eu = a.emb[0][p * H2 + lane]; ei = a.emb[1][p * H2 + lane]; yt = a.targets[p]; }
        else gc = a.g_emb[t][p * H2 + lane];
    }
    float hk_u = 0.f, hk_i = 0.f, hb = 0.f;                       // head: kernel (this lane's column of both halves), bias
    float acc_ku = 0.f, acc_ki = 0.f, acc_b = 0.f, acc_l = 0.f;   // head: this wave's partial gradient sums and loss
    if (HEAD) {
        if (lane < H2) { hk_u = a.head_kernel[lane]; hk_i = a.head_kernel[H2 + lane]; }
        hb = a.head_bias[0];
    }
    const float inv_b = 1.f / (float)a.n_pairs;
    for (int i = threadIdx.x; i < 4 * wsz; i += kThreads) {
        const int m = i / wsz, e = i - m * wsz;                   // e = k * HH2 + n: the weights are read in memory order
        const int k = e / HH2, n = e - k * HH2;
        const float *W = (m & 1) ? a.w_neigh[m >> 1] : a.w_self[m >> 1];
        wt[m * wsz + n * H1 + k] = W[e];
    }
    __syncthreads();

    const int k = lane & (H1 - 1);                                // H1 = 32: lanes 32..63 repeat 0..31
    const float inv_n0 = 1.f / (float)a.n0;
    for (long long w = w_first; w < walks; w += w_step) {
        const int t = w >= a.n_pairs, o = 1 - t;
        const long long p = w - (t ? a.n_pairs : 0);
        if (w != w_first) {
            zc = gc = eu = ei = 0.f;
            if (lane < H2) {
                zc = a.z[t][p * H2 + lane];
                if (HEAD) { eu = a.emb[0][p * H2 + lane]; ei = a.emb[1][p * H2 + lane]; yt = a.targets[p]; }
                else gc = a.g_emb[t][p * H2 + lane];
            }
        }
        if (HEAD) {
            const float zh = wave_sum(eu * hk_u + ei * hk_i) + hb;
            const float pr = 1.f / (1.f + __expf(-zh));
            const float d = pr - yt;
            const float gzh = 2.f * d * inv_b * pr * (1.f - pr);
            gc = gzh * (t ? hk_i : hk_u);
            if (t) acc_ki += gzh * ei;
            else {
                acc_ku += gzh * eu;
                acc_b += gzh;
                acc_l += d * d;
                if (lane == 0) a.pred[p] = pr;
            }
        }
        float gz = gc;
        if (a.normalize) {
            const float ss = wave_sum(zc * zc), dot = wave_sum(zc * gc);
            const float s = rsqrtf(fmaxf(ss, 1e-12f));
            const float c3 = ss > 1e-12f ? s * s * s * dot : 0.f;
            gz = s * gc - c3 * zc;
        }
        if (lane < H2) a.g_z[t][p * H2 + lane] = gz;
        gz_s[lane] = (a.relu && !(zc > 0.f)) ? 0.f : gz;          // through the layer's own activation
        wave_lds_fence();
        const float *ws = wt + (2 * t) * wsz, *wn = ws + wsz;
        float ds = 0.f, dn = 0.f;
#pragma unroll
        for (int n = 0; n < HH2; ++n) {
            ds = fmaf(gz_s[n], ws[n * H1 + k], ds);
            dn = fmaf(gz_s[HH2 + n], wn[n * H1 + k], dn);
        }
        ds *= a.in_scale;
        dn *= a.in_scale;
        if (lane < H1) {
            if (a.h1[t] && !(a.h1[t][p * H1 + lane] > 0.f)) ds = 0.f;     // through layer 0's activation (and its dropout)
            a.g_h1[t][p * H1 + lane] = ds;
        }
        // the n0 hop-1 heads of this walk: n0 * H1 consecutive floats repeating the H1 values dn / n0.  Stored
        // 16 bytes per lane: the values go through the wave's LDS line so that a lane holds columns 4*lane .. +3
        // (mod H1), which is what every one of its stores needs (256 is a multiple of H1)
        wave_lds_fence();                                         // the dot products have read gz_s
        if (lane < H1) gz_s[lane] = dn * inv_n0;
        wave_lds_fence();
        const f32x4 v4 = *reinterpret_cast<const f32x4 *>(gz_s + ((4 * lane) & (H1 - 1)));
        f32x4 *dst = reinterpret_cast<f32x4 *>(a.g_h1[o] + (a.n_pairs + p * a.n0) * H1);
        const int span4 = a.n0 * (H1 / 4);
        if (a.h1_mask[o]) {
            // element 4 i of the walk's n0 * H1 block: word (4 i) / 32, bits (4 i) % 32 .. + 3
            const uint32_t *mw = a.h1_mask[o] + (size_t)p * a.n0 * (H1 / 32);
            for (int i = lane; i < span4; i += 64) {
                const uint32_t bits = mw[i >> 3] >> ((i & 7) * 4);
                f32x4 m = v4;
                if (!(bits & 1u)) m[0] = 0.f;
                if (!(bits & 2u)) m[1] = 0.f;
                if (!(bits & 4u)) m[2] = 0.f;
                if (!(bits & 8u)) m[3] = 0.f;
                dst[i] = m;
            }
        } else {
            for (int i = lane; i < span4; i += 64) dst[i] = v4;
        }
        wave_lds_fence();                                         // gz_s is rewritten by the next walk
    }
    if (HEAD) {
        // block slab row: columns [0, H2) kernel gradient user half, [H2, 2 H2) item half, then bias, loss
        __syncthreads();                                          // all waves are done with the staged weights
        float *red = smem;                                        // [kWaves][2 * H2 + 2] over the weight area
        constexpr int C = 2 * H2 + 2;
        if (lane < H2) { red[wave * C + lane] = acc_ku; red[wave * C + H2 + lane] = acc_ki; }
        if (lane == 0) { red[wave * C + 2 * H2] = acc_b; red[wave * C + 2 * H2 + 1] = acc_l; }
        __syncthreads();
        if (threadIdx.x < C) {
            float s = 0.f;
#pragma unroll
            for (int wv = 0; wv < kWaves; ++wv) s += red[wv * C + threadIdx.x];
            a.head_slab[(size_t)blockIdx.x * C + threadIdx.x] = s;
        }
    }
}

inline int64_t tail_blocks(int64_t n_pairs) {
    int64_t blocks = (2 * n_pairs + kWaves - 1) / kWaves;
    return blocks > 256 ? 256 : blocks;
}

int launch_tail(const TailArgs &a, bool head, hipStream_t s) {
    const int64_t blocks = tail_blocks(a.n_pairs);
    const int H1 = a.H1, H2 = a.H2;
    const size_t lds = (size_t)(4 * (H2 / 2) * H1 + kWaves * kMaxH) * sizeof(float);        // at most 36 KB
#define GS_TAIL(h1, hh2) \
    if (H1 == h1 && H2 == 2 * hh2) { \
        if (head) hipLaunchKernelGGL((sage2_tail_bwd_kernel<h1, hh2, true>), dim3((unsigned)blocks), dim3(kThreads), lds, s, a); \
        else hipLaunchKernelGGL((sage2_tail_bwd_kernel<h1, hh2, false>), dim3((unsigned)blocks), dim3(kThreads), lds, s, a); \
    }
    GS_TAIL(32, 16) GS_TAIL(32, 32) GS_TAIL(64, 16) GS_TAIL(64, 32)
#undef GS_TAIL
    return gs::check_launch(head ? "gs_sage2_backward_tail_head" : "gs_sage2_backward_tail");
}

int set_l0_mask(TailArgs &a, const gs_sage2_l0_mask *m, const char *who) {
    a.in_scale = 1.f;
    if (!m) return GS_OK;
    if (m->input_scale != 0.f) a.in_scale = m->input_scale;
    for (int t = 0; t < 2; ++t) {
        if (!m->h1[t] || !m->h1_mask[t]) return gs::fail(GS_EINVAL, "%s: l0_mask has a null pointer", who);
        a.h1[t] = m->h1[t];
        a.h1_mask[t] = m->h1_mask[t];
    }
    return GS_OK;
}

}  // namespace

extern "C" int gs_sage2_backward_tail(const float *z_user, const float *z_item,
                                      const float *g_emb_user, const float *g_emb_item,
                                      int64_t n_pairs, int n0, int H1, int H2, int normalize, int act,
                                      const float *w1_self_user, const float *w1_neigh_user,
                                      const float *w1_self_item, const float *w1_neigh_item,
                                      float *g_z_user, float *g_z_item, float *g_h1_user, float *g_h1_item,
                                      const gs_sage2_l0_mask *l0_mask, void *stream) {
    if (n_pairs < 0 || n0 < 1) return gs::fail(GS_EINVAL, "gs_sage2_backward_tail: n_pairs=%lld n0=%d", (long long)n_pairs, n0);
    if ((H1 != 32 && H1 != 64) || (H2 != 32 && H2 != 64))
        return gs::fail(GS_ELIMIT, "gs_sage2_backward_tail: H1=%d H2=%d (each 32 or 64: the widths of gs_sage2_forward)", H1, H2);
    if (act != GS_ACT_LINEAR && act != GS_ACT_RELU) return gs::fail(GS_EINVAL, "gs_sage2_backward_tail: unknown activation %d", act);
    if (n_pairs == 0) return GS_OK;
    if (!z_user || !z_item || !g_emb_user || !g_emb_item || !w1_self_user || !w1_neigh_user || !w1_self_item ||
        !w1_neigh_item || !g_z_user || !g_z_item || !g_h1_user || !g_h1_item)
        return gs::fail(GS_EINVAL, "gs_sage2_backward_tail: null pointer argument");
    if ((reinterpret_cast<uintptr_t>(g_h1_user) | reinterpret_cast<uintptr_t>(g_h1_item)) & 15)
        return gs::fail(GS_EINVAL, "gs_sage2_backward_tail: g_h1 tables must be 16-byte aligned");
    if (n_pairs > ((int64_t)1 << 40) / (1 + n0)) return gs::fail(GS_ELIMIT, "gs_sage2_backward_tail: n_pairs=%lld too large", (long long)n_pairs);
    TailArgs a{};
    a.z[0] = z_user; a.z[1] = z_item; a.g_emb[0] = g_emb_user; a.g_emb[1] = g_emb_item;
    a.w_self[0] = w1_self_user; a.w_neigh[0] = w1_neigh_user; a.w_self[1] = w1_self_item; a.w_neigh[1] = w1_neigh_item;
    a.g_z[0] = g_z_user; a.g_z[1] = g_z_item; a.g_h1[0] = g_h1_user; a.g_h1[1] = g_h1_item;
    a.n_pairs = n_pairs; a.n0 = n0; a.H1 = H1; a.H2 = H2; a.normalize = normalize != 0; a.relu = act == GS_ACT_RELU;
    if (int rc = set_l0_mask(a, l0_mask, "gs_sage2_backward_tail")) return rc;
    return launch_tail(a, false, (hipStream_t)stream);
}

extern "C" size_t gs_sage2_tail_head_workspace(int64_t n_pairs, int H2) {
    if (n_pairs < 0 || H2 < 1) return 0;
    return (size_t)tail_blocks(n_pairs) * (size_t)(2 * H2 + 2) * sizeof(float);
}

extern "C" int gs_sage2_tail_head_rows(int64_t n_pairs) { return (int)tail_blocks(n_pairs); }

extern "C" int gs_sage2_backward_tail_head(const float *z_user, const float *z_item,
                                           const float *emb_user, const float *emb_item, const float *targets,
                                           const float *head_kernel, const float *head_bias,
                                           int64_t n_pairs, int n0, int H1, int H2, int normalize, int act,
                                           const float *w1_self_user, const float *w1_neigh_user,
                                           const float *w1_self_item, const float *w1_neigh_item,
                                           float *g_z_user, float *g_z_item, float *g_h1_user, float *g_h1_item,
                                           float *pred, void *head_slab, size_t head_slab_bytes,
                                           const gs_sage2_l0_mask *l0_mask, void *stream) {
    if (n_pairs < 1 || n0 < 1) return gs::fail(GS_EINVAL, "gs_sage2_backward_tail_head: n_pairs=%lld n0=%d", (long long)n_pairs, n0);
    if ((H1 != 32 && H1 != 64) || (H2 != 32 && H2 != 64))
        return gs::fail(GS_ELIMIT, "gs_sage2_backward_tail_head: H1=%d H2=%d (each 32 or 64: the widths of gs_sage2_forward)", H1, H2);
    if (act != GS_ACT_LINEAR && act != GS_ACT_RELU) return gs::fail(GS_EINVAL, "gs_sage2_backward_tail_head: unknown activation %d", act);
    if (!z_user || !z_item || !emb_user || !emb_item || !targets || !head_kernel || !head_bias || !w1_self_user ||
        !w1_neigh_user || !w1_self_item || !w1_neigh_item || !g_z_user || !g_z_item || !g_h1_user || !g_h1_item ||
        !pred || !head_slab)
        return gs::fail(GS_EINVAL, "gs_sage2_backward_tail_head: null pointer argument");
    if ((reinterpret_cast<uintptr_t>(g_h1_user) | reinterpret_cast<uintptr_t>(g_h1_item)) & 15)
        return gs::fail(GS_EINVAL, "gs_sage2_backward_tail_head: g_h1 tables must be 16-byte aligned");
    if (n_pairs > ((int64_t)1 << 40) / (1 + n0)) return gs::fail(GS_ELIMIT, "gs_sage2_backward_tail_head: n_pairs=%lld too large", (long long)n_pairs);
    const size_t need = gs_sage2_tail_head_workspace(n_pairs, H2);
    if (head_slab_bytes < need)
        return gs::fail(GS_EINVAL, "gs_sage2_backward_tail_head: workspace of %zu bytes needed, %zu given", need, head_slab_bytes);
    TailArgs a{};
    a.z[0] = z_user; a.z[1] = z_item;
    a.w_self[0] = w1_self_user; a.w_neigh[0] = w1_neigh_user; a.w_self[1] = w1_self_item; a.w_neigh[1] = w1_neigh_item;
    a.g_z[0] = g_z_user; a.g_z[1] = g_z_item; a.g_h1[0] = g_h1_user; a.g_h1[1] = g_h1_item;
    a.n_pairs = n_pairs; a.n0 = n0; a.H1 = H1; a.H2 = H2; a.normalize = normalize != 0; a.relu = act == GS_ACT_RELU;
    a.emb[0] = emb_user; a.emb[1] = emb_item; a.head_kernel = head_kernel; a.head_bias = head_bias; a.targets = targets;
    a.pred = pred; a.head_slab = static_cast<float *>(head_slab);
    if (int rc = set_l0_mask(a, l0_mask, "gs_sage2_backward_tail_head")) return rc;
    return launch_tail(a, true, (hipStream_t)stream);
}
