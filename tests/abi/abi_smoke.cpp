// libgsage.so driven from plain C++ with nothing but the HIP runtime: the C ABI of
// include/gsage.h is the whole boundary (device pointers, sizes, a stream).
//
// Builds a small bipartite graph with gs_csr_build, replays a walk tape through
// gs_sample, gathers and averages neighbor rows, projects them, and checks every
// stage against the same arithmetic done on the host (adjacency in edge-column
// order as bipartite_graph.py:56-58, choices() as floor(u * degree) per
// sampler.py:78, mean + two Dense halves + relu per utils.py:103-119).
//
//   hipcc --offload-arch=gfx950 -I include tests/abi/abi_smoke.cpp -L graphsage_amd -lgsage -o tests/abi/abi_smoke
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gsage.h"

#define HIP_OK(x)                                                                    \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } \
    } while (0)
#define GS_CHECK(x)                                                                  \
    do {                                                                             \
        int rc_ = (x);                                                               \
        if (rc_ != GS_OK) { std::printf("%s -> %d: %s\n", #x, rc_, gs_last_error()); return 3; } \
    } while (0)

template <class T> static T *to_device(const std::vector<T> &v) {
    T *p = nullptr;
    if (hipMalloc(&p, (v.size() ? v.size() : 1) * sizeof(T)) != hipSuccess) return nullptr;
    if (!v.empty() && hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return p;
}
template <class T> static std::vector<T> to_host(const T *p, size_t n) {
    std::vector<T> v(n);
    if (n && hipMemcpy(v.data(), p, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) v.clear();
    return v;
}

int main() {
    std::printf("gs_version %d\n", gs_version());
    const int NU = 50, NI = 12, N = NU + NI, F = 24, H = 32, n0 = 5, n1 = 3;
    // undirected pairs (user, item); a small LCG keeps the program self-contained
    uint64_t st = 12345;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(st >> 33); };
    std::vector<int32_t> a, b;
    for (int u = 0; u < NU; ++u) { a.push_back(u); b.push_back(NU + u % NI); }         // everybody has a neighbor
    for (int i = 0; i < 400; ++i) { a.push_back(rnd() % NU); b.push_back(NU + rnd() % NI); }
    const int64_t P = (int64_t)a.size();

    // host adjacency in entry order: pairs a->b, then b->a
    std::vector<std::vector<int32_t>> adj(N);
    for (int64_t i = 0; i < P; ++i) adj[a[i]].push_back(b[i]);
    for (int64_t i = 0; i < P; ++i) adj[b[i]].push_back(a[i]);

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    int32_t *d_a = to_device(a), *d_b = to_device(b);
    int64_t *d_indptr; int32_t *d_indices, *d_err;
    HIP_OK(hipMalloc(&d_indptr, (N + 1) * sizeof(int64_t)));
    HIP_OK(hipMalloc(&d_indices, 2 * P * sizeof(int32_t)));
    HIP_OK(hipMalloc(&d_err, 2 * sizeof(int32_t)));
    HIP_OK(hipMemsetAsync(d_err, 0, 2 * sizeof(int32_t), stream));
    size_t ws_bytes = 0;
    GS_CHECK(gs_csr_build_workspace(P, N, 0, &ws_bytes));
    void *d_ws; HIP_OK(hipMalloc(&d_ws, ws_bytes ? ws_bytes : 1));
    GS_CHECK(gs_csr_build(d_a, d_b, P, 1, N, d_indptr, d_indices, d_err, 0, d_ws, ws_bytes, stream));
    HIP_OK(hipStreamSynchronize(stream));
    auto indptr = to_host(d_indptr, N + 1);
    auto indices = to_host(d_indices, 2 * P);
    int bad = 0;
    for (int v = 0; v < N; ++v) {
        if (indptr[v + 1] - indptr[v] != (int64_t)adj[v].size()) { ++bad; continue; }
        for (size_t j = 0; j < adj[v].size(); ++j) bad += indices[indptr[v] + j] != adj[v][j];
    }
    std::printf("gs_csr_build: %d mismatches over %d nodes, %lld entries\n", bad, N, (long long)(2 * P));
    if (bad) return 1;

    // walks: 8 roots (4 users, 4 items), fan-outs {n0, n1}, draws replayed from a tape
    std::vector<int32_t> seeds = {0, 7, 19, 49, NU, NU + 3, NU + 5, NU + 11};
    const int S = (int)seeds.size(), T = n0 + n0 * n1;
    std::vector<uint32_t> tape(S * T);
    for (auto &w : tape) w = rnd() * 2u + (rnd() & 1u);
    tape[0] = 0u; tape[1] = 0xFFFFFFFFu;                                              // first / last neighbor
    int32_t *d_seeds = to_device(seeds); uint32_t *d_tape = to_device(tape);
    int32_t *d_h1, *d_h2;
    HIP_OK(hipMalloc(&d_h1, S * n0 * sizeof(int32_t)));
    HIP_OK(hipMalloc(&d_h2, S * n0 * n1 * sizeof(int32_t)));
    int32_t fan[2] = {n0, n1};
    int32_t *outs[2] = {d_h1, d_h2};
    GS_CHECK(gs_sample(d_indptr, d_indices, N, d_seeds, S, fan, 2, 0, 0, nullptr, d_tape, outs, d_err, stream));
    HIP_OK(hipStreamSynchronize(stream));
    auto h1 = to_host(d_h1, (size_t)S * n0), h2 = to_host(d_h2, (size_t)S * n0 * n1);
    auto err = to_host(d_err, 2);
    bad = err[0] || err[1];
    auto pick = [&](int v, uint32_t r) { return adj[v][(size_t)(((uint64_t)r * adj[v].size()) >> 32)]; };
    for (int w = 0; w < S; ++w) {
        for (int t = 0; t < n0; ++t) bad += h1[w * n0 + t] != pick(seeds[w], tape[w * T + t]);
        for (int t = 0; t < n0 * n1; ++t) bad += h2[w * n0 * n1 + t] != pick(h1[w * n0 + t / n1], tape[w * T + n0 + t]);
    }
    std::printf("gs_sample (tape replay): %d mismatches over %d walks\n", bad, S);
    if (bad) return 1;

    // aggregate the user-rooted walks' roots: own rows (users), mean over hop-1 rows (items), projection
    std::vector<float> fu(NU * F), fi(NI * F), ws(F * H / 2), wn(F * H / 2), bias(H);
    auto unit = [&]() { return (float)(rnd() % 2001) / 1000.f - 1.f; };
    for (auto &x : fu) x = unit();
    for (auto &x : fi) x = unit();
    for (auto &x : ws) x = 0.2f * unit();
    for (auto &x : wn) x = 0.2f * unit();
    for (auto &x : bias) x = 0.1f * unit();
    float *d_fu = to_device(fu), *d_fi = to_device(fi), *d_wself = to_device(ws), *d_wn = to_device(wn), *d_bias = to_device(bias);
    const int M = 4;                                                                  // the four user roots
    float *d_xs, *d_xm, *d_out;
    HIP_OK(hipMalloc(&d_xs, M * F * sizeof(float)));
    HIP_OK(hipMalloc(&d_xm, M * F * sizeof(float)));
    HIP_OK(hipMalloc(&d_out, M * H * sizeof(float)));
    GS_CHECK(gs_gather_rows(d_fu, NU, F, 0, d_seeds, M, d_xs, stream));
    GS_CHECK(gs_gather_mean_fwd(d_fi, NI, F, NU, d_h1, M, n0, d_xm, stream));
    GS_CHECK(gs_sage_project_fwd(d_xs, F, d_xm, F, M, d_wself, d_wn, d_bias, H, GS_ACT_RELU, d_out, stream));
    HIP_OK(hipStreamSynchronize(stream));
    auto out = to_host(d_out, (size_t)M * H);
    double worst = 0.0;
    for (int m = 0; m < M; ++m) {
        std::vector<double> mean(F, 0.0);
        for (int j = 0; j < n0; ++j)
            for (int f = 0; f < F; ++f) mean[f] += fi[(size_t)(h1[m * n0 + j] - NU) * F + f];
        for (int f = 0; f < F; ++f) mean[f] /= n0;
        for (int c = 0; c < H; ++c) {
            double acc = bias[c];
            for (int f = 0; f < F; ++f)
                acc += c < H / 2 ? (double)fu[(size_t)seeds[m] * F + f] * ws[f * (H / 2) + c]
                                 : mean[f] * wn[f * (H / 2) + (c - H / 2)];
            if (acc < 0) acc = 0;
            worst = std::fmax(worst, std::fabs(acc - out[m * H + c]));
        }
    }
    std::printf("gather + mean + projection: max |device - host| = %.3e\n", worst);
    if (!(worst < 1e-5)) return 1;

    // an argument the library must refuse, with its reason
    int rc = gs_sage_project_fwd(d_xs, F, d_xm, F, M, d_wself, d_wn, d_bias, 31, GS_ACT_RELU, d_out, stream);
    std::printf("odd output width refused: rc=%d (%s)\n", rc, gs_last_error());
    if (rc != GS_EINVAL) return 1;
    std::printf("abi smoke ok\n");
    return 0;
}
