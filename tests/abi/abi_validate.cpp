// Host side of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer, without a GPU.
//
// Every entry point of include/gsage.h is called with arguments it must refuse (null pointers,
// negative or oversized counts, shapes no kernel is built for, workspaces that are too small,
// segment / draw tables that contradict each other) and with empty inputs it must accept.  All of
// these return from the argument checks, the SampleParams / segment-table packing or the size
// arithmetic BEFORE anything is launched, so the program needs no device: it is linked against a
// build of the library whose host code is instrumented (tests/abi/libgsage_asan.so, `make asan`),
// and any out-of-bounds access, misaligned read, signed overflow or bad shift on those paths ends
// it with a sanitizer report.  Pointers marked "device" below are never dereferenced on the host;
// they point into a small host array only so that they are non-null and 16-byte aligned.
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "gsage.h"

static int failures = 0;

#define REFUSED(code, call)                                                                          \
    do {                                                                                             \
        const int rc_ = (call);                                                                      \
        const char *msg_ = gs_last_error();                                                          \
        if (rc_ != (code) || !msg_ || !msg_[0]) {                                                    \
            std::printf("line %d: expected %d with a message, got %d (%s)\n", __LINE__, (code), rc_, msg_ ? msg_ : ""); \
            ++failures;                                                                              \
        }                                                                                            \
    } while (0)

#define ACCEPTED(call)                                                                               \
    do {                                                                                             \
        const int rc_ = (call);                                                                      \
        if (rc_ != GS_OK) { std::printf("line %d: expected GS_OK, got %d (%s)\n", __LINE__, rc_, gs_last_error()); ++failures; } \
    } while (0)

#define EXPECT(cond)                                                                                 \
    do { if (!(cond)) { std::printf("line %d: %s\n", __LINE__, #cond); ++failures; } } while (0)

int main() {
    alignas(256) static char arena[4096];
    float *f = reinterpret_cast<float *>(arena);                 // stands for device float memory
    int32_t *i32 = reinterpret_cast<int32_t *>(arena + 1024);
    int64_t *i64 = reinterpret_cast<int64_t *>(arena + 2048);
    float *odd = reinterpret_cast<float *>(arena + 4);           // not 16-byte aligned

    EXPECT(gs_version() >= 4);
    EXPECT(gs_last_error() != nullptr);

    // ---- timers (no event is created here: that needs a device) ----
    REFUSED(GS_EINVAL, gs_timer_create(nullptr));
    ACCEPTED(gs_timer_arm(nullptr));
    float us = 0.f;
    REFUSED(GS_EINVAL, gs_timer_elapsed_us(nullptr, &us));
    ACCEPTED(gs_timer_destroy(nullptr));

    // ---- adjacency build ----
    size_t bytes = 0;
    REFUSED(GS_EINVAL, gs_csr_build_workspace(10, 10, 0, nullptr));
    REFUSED(GS_EINVAL, gs_csr_build_workspace(-1, 10, 0, &bytes));
    REFUSED(GS_EINVAL, gs_csr_build_workspace(10, (int64_t)1 << 33, 0, &bytes));
    // (the size of the sort / scan scratch is asked of the device library: without a device that query
    // fails with GS_ELAUNCH after the plan arithmetic, which is what runs under the sanitizers here)
    {
        const int64_t cases[4][3] = {{0, 0, 0}, {1000, 77, 0}, {(int64_t)4000000000LL, 110000000, 0}, {1000, 77, 7}};
        for (const auto &c : cases) {
            bytes = 0;
            const int rc = gs_csr_build_workspace(c[0], c[1], c[2], &bytes);
            EXPECT(rc == GS_OK || rc == GS_ELAUNCH);
            if (rc == GS_OK && c[0] > 0) EXPECT(bytes >= (size_t)c[0] * 4);
        }
    }
    REFUSED(GS_EINVAL, gs_csr_build(i32, i32, -5, 0, 10, i64, i32, i32, 0, arena, 64, nullptr));
    REFUSED(GS_EINVAL, gs_csr_build(i32, i32, 5, 0, 10, nullptr, i32, i32, 0, arena, 64, nullptr));
    REFUSED(GS_EINVAL, gs_csr_build(i32, i32, 5, 1, (int64_t)1 << 32, i64, i32, i32, 0, arena, 64, nullptr));

    // ---- samplers: the hop table (sizes, tape offsets) is packed on the host ----
    int32_t fan2[2] = {25, 10}, fan0[2] = {25, 0}, fan_big[3] = {4096, 4096, 2}, fan4[4] = {3, 2, 2, 2};
    int32_t *outs[4] = {i32, i32, i32, i32}, *outs_hole[2] = {i32, nullptr};
    ACCEPTED(gs_sample(nullptr, nullptr, 0, nullptr, 0, nullptr, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr));   // empty batch
    REFUSED(GS_EINVAL, gs_sample(nullptr, i32, 100, i32, 4, fan2, 2, 1, 0, nullptr, nullptr, outs, i32, nullptr));
    REFUSED(GS_EINVAL, gs_sample(i64, i32, 100, i32, 4, fan2, 0, 1, 0, nullptr, nullptr, outs, i32, nullptr));
    REFUSED(GS_EINVAL, gs_sample(i64, i32, 100, i32, 4, fan4, GS_MAX_HOPS + 1, 1, 0, nullptr, nullptr, outs, i32, nullptr));
    REFUSED(GS_EINVAL, gs_sample(i64, i32, 100, i32, 4, fan0, 2, 1, 0, nullptr, nullptr, outs, i32, nullptr));
    REFUSED(GS_EINVAL, gs_sample(i64, i32, 100, i32, 4, fan2, 2, 1, 0, nullptr, nullptr, outs_hole, i32, nullptr));
    REFUSED(GS_ELIMIT, gs_sample(i64, i32, 100, i32, 4, fan_big, 3, 1, 0, nullptr, nullptr, outs, i32, nullptr));
    REFUSED(GS_EINVAL, gs_sample(i64, i32, 0, i32, 4, fan2, 2, 1, 0, nullptr, nullptr, outs, i32, nullptr));
    REFUSED(GS_EINVAL, gs_sample(i64, i32, (int64_t)1 << 31, i32, 4, fan2, 2, 1, 0, nullptr, nullptr, outs, i32, nullptr));
    REFUSED(GS_EINVAL, gs_sample(i64, i32, 100, i32, -4, fan2, 2, 1, 0, nullptr, nullptr, outs, i32, nullptr));

    ACCEPTED(gs_sample_cyclic(nullptr, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, nullptr));
    REFUSED(GS_EINVAL, gs_sample_cyclic(i64, i32, 100, i32, 4, fan2, 2, nullptr, 1, i64, nullptr, 0, outs, i32, nullptr));
    REFUSED(GS_EINVAL, gs_sample_cyclic(i64, i32, 100, i32, 4, fan2, 2, i64, 1, nullptr, nullptr, 0, outs, i32, nullptr));
    REFUSED(GS_EINVAL, gs_sample_cyclic(i64, i32, 100, i32, 4, fan2, 5, i64, 1, i64, nullptr, 0, outs, i32, nullptr));
    REFUSED(GS_EINVAL, gs_sample_cyclic(i64, i32, 100, i32, 4, fan2, 2, i64, 1, i64, nullptr, 12, outs, i32, nullptr));   // length without a tape
    REFUSED(GS_EINVAL, gs_sample_cyclic(i64, i32, 100, i32, 4, fan2, 2, i64, 1, i64, i32, -1, outs, i32, nullptr));
    REFUSED(GS_EINVAL, gs_sample_cyclic(i64, i32, 100, i32, 4, fan0, 2, i64, 1, i64, nullptr, 0, outs, i32, nullptr));
    REFUSED(GS_ELIMIT, gs_sample_cyclic(i64, i32, 100, i32, 4, fan_big, 3, i64, 1, i64, nullptr, 0, outs, i32, nullptr));
    REFUSED(GS_EINVAL, gs_sample_cyclic(i64, i32, 100, i32, 4, fan2, 2, i64, 1, i64, nullptr, 0, outs_hole, i32, nullptr));

    // ---- gathers ----
    ACCEPTED(gs_gather_rows(nullptr, 0, 64, 0, nullptr, 0, nullptr, nullptr));
    REFUSED(GS_EINVAL, gs_gather_rows(nullptr, 10, 64, 0, i32, 4, f, nullptr));
    REFUSED(GS_EINVAL, gs_gather_rows(f, 10, 0, 0, i32, 4, f, nullptr));
    REFUSED(GS_EINVAL, gs_gather_rows(f, 10, 64, 0, i32, -4, f, nullptr));
    REFUSED(GS_EINVAL, gs_gather_mean_fwd(f, 10, 64, 0, i32, 4, 0, f, nullptr));
    REFUSED(GS_EINVAL, gs_gather_mean_fwd(nullptr, 10, 64, 0, i32, 4, 5, f, nullptr));
    REFUSED(GS_EINVAL, gs_gather_mean_fwd(f, 10, 64, 0, nullptr, 4, 5, f, nullptr));      // dense groups larger than the table
    REFUSED(GS_EINVAL, gs_gather_max_fwd(f, 10, -3, 0, i32, 4, 5, f, nullptr));
    REFUSED(GS_EINVAL, gs_gather_mean_bwd(f, i32, -1, 5, 64, 0, 10, f, nullptr));
    REFUSED(GS_EINVAL, gs_gather_mean_bwd(f, nullptr, 4, 5, 64, 0, 10, f, nullptr));      // 20 dense rows into a 10-row table
    REFUSED(GS_EINVAL, gs_group_max_bwd(nullptr, f, f, 4, 5, 64, f, nullptr));
    REFUSED(GS_EINVAL, gs_group_max_bwd(f, f, f, 4, 0, 64, f, nullptr));

    // ---- projection ----
    ACCEPTED(gs_sage_project_fwd(nullptr, 64, nullptr, 64, 0, nullptr, nullptr, nullptr, 32, GS_ACT_RELU, nullptr, nullptr));
    REFUSED(GS_EINVAL, gs_sage_project_fwd(f, 64, f, 64, 10, f, f, f, 33, GS_ACT_RELU, f, nullptr));
    REFUSED(GS_EINVAL, gs_sage_project_fwd(f, 64, f, 64, 10, f, f, f, 32, 7, f, nullptr));
    REFUSED(GS_EINVAL, gs_sage_project_fwd(f, 0, f, 64, 10, f, f, f, 32, GS_ACT_RELU, f, nullptr));
    REFUSED(GS_EINVAL, gs_sage_project_fwd(nullptr, 64, f, 64, 10, f, f, f, 32, GS_ACT_RELU, f, nullptr));
    REFUSED(GS_EINVAL, gs_sage_project_gather_fwd(f, 10, 64, 0, nullptr, f, 64, 10, f, f, f, 32, GS_ACT_RELU, f, nullptr));
    EXPECT(gs_sage_project_bwd_workspace(0, 64, 32, 10) == 0);
    EXPECT(gs_sage_project_bwd_workspace(64, 64, 32, (int64_t)1 << 40) > 0);              // huge M: size arithmetic in 64 bits
    const size_t ws_need = gs_sage_project_bwd_workspace(64, 64, 32, 1000);
    EXPECT(ws_need > 0);
    REFUSED(GS_EINVAL, gs_sage_project_bwd(f, 64, f, 64, 1000, f, f, 32, GS_ACT_RELU, f, f, f, f, f, nullptr, nullptr, arena, ws_need - 4, nullptr));
    REFUSED(GS_EINVAL, gs_sage_project_bwd(f, 64, f, 64, 1000, f, f, 32, GS_ACT_RELU, nullptr, f, f, f, f, nullptr, nullptr, arena, ws_need, nullptr));
    REFUSED(GS_EINVAL, gs_sage_project_bwd(f, 64, f, 64, 1000, nullptr, nullptr, 32, GS_ACT_LINEAR, f, f, nullptr, nullptr, nullptr, f, f, arena, 0, nullptr));
    REFUSED(GS_EINVAL, gs_sage_project_bwd(f, 64, f, 64, -1, f, f, 32, GS_ACT_LINEAR, f, f, f, f, f, nullptr, nullptr, arena, ws_need, nullptr));
    EXPECT(gs_sage_project_wgrad_by_id_supported(64, 64, 32) == 1);
    EXPECT(gs_sage_project_wgrad_by_id_supported(24, 19, 32) == 0);
    EXPECT(gs_sage_project_wgrad_by_id_supported(64, 64, 31) == 0);
    gs_wgrad_problem pa{}, pb{};
    pa.x_self = f; pa.x_mean = f; pa.out = f; pa.g_out = f; pa.g_W_self = f; pa.g_W_neigh = f; pa.workspace = arena;
    pa.workspace_bytes = ws_need;
    pb = pa;
    REFUSED(GS_EINVAL, gs_sage_project_wgrad_pair(nullptr, &pb, 64, 64, 1000, 32, GS_ACT_RELU, nullptr));
    pb.workspace_bytes = 8;
    REFUSED(GS_EINVAL, gs_sage_project_wgrad_pair(&pa, &pb, 64, 64, 1000, 32, GS_ACT_RELU, nullptr));
    pb = pa; pb.out = nullptr;
    REFUSED(GS_EINVAL, gs_sage_project_wgrad_pair(&pa, &pb, 64, 64, 1000, 32, GS_ACT_RELU, nullptr));
    pb = pa; pb.self_ids = i32; pb.self_rows = 0;
    REFUSED(GS_EINVAL, gs_sage_project_wgrad_pair(&pa, &pb, 64, 64, 1000, 32, GS_ACT_RELU, nullptr));
    pb.self_rows = 50;                                                                    // by id on a shape without the direct kernel
    {
        gs_wgrad_problem qa = pa, qb = pb;
        const size_t w2 = gs_sage_project_bwd_workspace(24, 24, 32, 1000);
        qa.workspace_bytes = qb.workspace_bytes = w2 <= sizeof(arena) ? w2 : sizeof(arena);
        if (w2 <= sizeof(arena)) REFUSED(GS_ELIMIT, gs_sage_project_wgrad_pair(&qa, &qb, 24, 24, 1000, 32, GS_ACT_RELU, nullptr));
    }
    ACCEPTED(gs_sage_project_wgrad_pair(&pa, &pa, 64, 64, 0, 32, GS_ACT_RELU, nullptr));
    REFUSED(GS_EINVAL, gs_l2_normalize_fwd(f, 10, 0, f, nullptr));
    REFUSED(GS_EINVAL, gs_l2_normalize_bwd(f, f, -1, 32, f, nullptr));

    // ---- fused layer: the segment table ----
    EXPECT(gs_sage_fused_supported(64, 64, 32, 25) == 1);
    EXPECT(gs_sage_fused_supported(24, 19, 32, 25) == 0);
    EXPECT(gs_sage_fused_supported(256, 256, 256, 5) == 1);
    EXPECT(gs_sage_fused_supported(256, 256, 256, 33) == 0);
    EXPECT(gs_sage_fused_supported(64, 64, 32, 0) == 0);
    gs_fused_segment sg[9];
    std::memset(sg, 0, sizeof(sg));
    for (auto &s : sg) { s.head_ids = i32; s.child_ids = i32; s.n_heads = 100; s.n_child = 5; }
    ACCEPTED(gs_sage_fused_fwd(f, 10, 64, 0, f, 10, 64, 0, sg, 0, f, f, f, 32, GS_ACT_RELU, 0, f, nullptr, nullptr, nullptr));   // no segments
    REFUSED(GS_EINVAL, gs_sage_fused_fwd(f, 10, 64, 0, f, 10, 64, 0, sg, 9, f, f, f, 32, GS_ACT_RELU, 0, f, nullptr, nullptr, nullptr));
    REFUSED(GS_EINVAL, gs_sage_fused_fwd(f, 10, 64, 0, f, 10, 64, 0, sg, -1, f, f, f, 32, GS_ACT_RELU, 0, f, nullptr, nullptr, nullptr));
    REFUSED(GS_EINVAL, gs_sage_fused_fwd(f, 10, 64, 0, f, 10, 64, 0, sg, 2, f, f, f, 32, 9, 0, f, nullptr, nullptr, nullptr));
    REFUSED(GS_EINVAL, gs_sage_fused_fwd(f, 10, 64, 0, f, 10, 64, 0, nullptr, 2, f, f, f, 32, GS_ACT_RELU, 0, f, nullptr, nullptr, nullptr));
    sg[1].n_child = 0;
    REFUSED(GS_EINVAL, gs_sage_fused_fwd(f, 10, 64, 0, f, 10, 64, 0, sg, 2, f, f, f, 32, GS_ACT_RELU, 0, f, nullptr, nullptr, nullptr));
    sg[1].n_child = 5; sg[1].child_ids = nullptr;                 // mixes id-addressed and dense segments
    REFUSED(GS_EINVAL, gs_sage_fused_fwd(f, 10, 64, 0, f, 100000, 64, 0, sg, 2, f, f, f, 32, GS_ACT_RELU, 0, f, nullptr, nullptr, nullptr));
    sg[1].child_ids = i32; sg[1].head_ids = nullptr; sg[1].self_row0 = 5;                 // dense heads past the end of the table
    REFUSED(GS_EINVAL, gs_sage_fused_fwd(f, 10, 64, 0, f, 10, 64, 0, sg, 2, f, f, f, 32, GS_ACT_RELU, 0, f, nullptr, nullptr, nullptr));
    sg[1].head_ids = i32; sg[1].self_row0 = 0;
    REFUSED(GS_EINVAL, gs_sage_fused_fwd(nullptr, 10, 64, 0, f, 10, 64, 0, sg, 2, f, f, f, 32, GS_ACT_RELU, 0, f, nullptr, nullptr, nullptr));
    REFUSED(GS_EINVAL, gs_sage_fused_fwd(odd, 10, 64, 0, f, 10, 64, 0, sg, 2, f, f, f, 32, GS_ACT_RELU, 0, f, nullptr, nullptr, nullptr));
    REFUSED(GS_EINVAL, gs_sage_fused_fwd(f, 0, 64, 0, f, 10, 64, 0, sg, 2, f, f, f, 32, GS_ACT_RELU, 0, f, nullptr, nullptr, nullptr));
    REFUSED(GS_ELIMIT, gs_sage_fused_fwd(f, 10, 24, 0, f, 10, 19, 0, sg, 2, f, f, f, 32, GS_ACT_RELU, 0, f, nullptr, nullptr, nullptr));
    sg[0].n_child = 4000;                                          // more children than a block's LDS holds
    REFUSED(GS_ELIMIT, gs_sage_fused_fwd(f, 10, 64, 0, f, 10, 64, 0, sg, 2, f, f, f, 32, GS_ACT_RELU, 0, f, nullptr, nullptr, nullptr));
    sg[0].n_child = 5;
    // the workspace form: sizes, and a workspace that is too small
    EXPECT(gs_sage_fused_workspace_bytes(64, 64, 32, 25) == 0);                    // narrow layers use none
    EXPECT(gs_sage_fused_workspace_bytes(256, 256, 256, 5) == (size_t)3 * 128 * 512 * 2);
    EXPECT(gs_sage_fused_workspace_bytes(256, 256, 256, 33) == 0);                 // outside every kernel's range
    ACCEPTED(gs_sage_fused_fwd_ws(f, 10, 64, 0, f, 10, 64, 0, sg, 0, f, f, f, 32, GS_ACT_RELU, 0, f, nullptr, nullptr, f, 16, nullptr));
    REFUSED(GS_EINVAL, gs_sage_fused_fwd_ws(f, 10, 256, 0, f, 10, 256, 0, sg, 2, f, f, f, 256, GS_ACT_RELU, 0, f, nullptr, nullptr, f, 1000, nullptr));
    REFUSED(GS_EINVAL, gs_sage_fused_fwd_ws(f, 10, 64, 0, f, 10, 64, 0, sg, 9, f, f, f, 32, GS_ACT_RELU, 0, f, nullptr, nullptr, nullptr, 0, nullptr));

    // ---- fused layer with draws inside: sampling descriptor and draw table ----
    gs_fused_sampling sp{};
    sp.indptr = i64; sp.indices = i32; sp.n_nodes = 100; sp.rng_seed = 1; sp.err_flag = i32; sp.tape_len = 275;
    gs_fused_draw dr[8];
    std::memset(dr, 0, sizeof(dr));
    for (auto &d : dr) d.hop = -1;
    REFUSED(GS_EINVAL, gs_sage_fused_sampled_fwd(f, 10, 64, 0, f, 10, 64, 0, sg, 2, f, f, f, 32, GS_ACT_RELU, 0, f, nullptr, nullptr, nullptr, dr, nullptr));
    REFUSED(GS_EINVAL, gs_sage_fused_sampled_fwd(f, 10, 64, 0, f, 10, 64, 0, sg, 2, f, f, f, 32, GS_ACT_RELU, 0, f, nullptr, nullptr, &sp, nullptr, nullptr));
    dr[1].hop = 1; dr[1].heads_per_walk = 25; dr[1].tape_off = 25; dr[1].walk0 = 0;
    sg[1].child_ids = nullptr; sg[1].n_child = 10;
    sg[1].head_ids = nullptr;                                      // a drawn segment needs its heads' node ids
    REFUSED(GS_EINVAL, gs_sage_fused_sampled_fwd(f, 10, 64, 0, f, 10, 64, 0, sg, 2, f, f, f, 32, GS_ACT_RELU, 0, f, nullptr, nullptr, &sp, dr, nullptr));
    sg[1].head_ids = i32;
    dr[1].hop = GS_MAX_HOPS;
    REFUSED(GS_EINVAL, gs_sage_fused_sampled_fwd(f, 10, 64, 0, f, 10, 64, 0, sg, 2, f, f, f, 32, GS_ACT_RELU, 0, f, nullptr, nullptr, &sp, dr, nullptr));
    dr[1].hop = 1; dr[1].heads_per_walk = 0;
    REFUSED(GS_EINVAL, gs_sage_fused_sampled_fwd(f, 10, 64, 0, f, 10, 64, 0, sg, 2, f, f, f, 32, GS_ACT_RELU, 0, f, nullptr, nullptr, &sp, dr, nullptr));
    dr[1].heads_per_walk = 25; dr[1].tape_off = 100; sp.replay = reinterpret_cast<const uint32_t *>(i32);   // 100 + 250 words > 275 per walk
    REFUSED(GS_EINVAL, gs_sage_fused_sampled_fwd(f, 10, 64, 0, f, 10, 64, 0, sg, 2, f, f, f, 32, GS_ACT_RELU, 0, f, nullptr, nullptr, &sp, dr, nullptr));
    dr[1].tape_off = 25; sp.tape_len = 0;
    REFUSED(GS_EINVAL, gs_sage_fused_sampled_fwd(f, 10, 64, 0, f, 10, 64, 0, sg, 2, f, f, f, 32, GS_ACT_RELU, 0, f, nullptr, nullptr, &sp, dr, nullptr));
    sp.tape_len = 275; sp.replay = nullptr; sp.indptr = nullptr;
    REFUSED(GS_EINVAL, gs_sage_fused_sampled_fwd(f, 10, 64, 0, f, 10, 64, 0, sg, 2, f, f, f, 32, GS_ACT_RELU, 0, f, nullptr, nullptr, &sp, dr, nullptr));
    sp.indptr = i64; sp.n_nodes = 0;
    REFUSED(GS_EINVAL, gs_sage_fused_sampled_fwd(f, 10, 64, 0, f, 10, 64, 0, sg, 2, f, f, f, 32, GS_ACT_RELU, 0, f, nullptr, nullptr, &sp, dr, nullptr));
    sp.n_nodes = 100;
    dr[1].heads_per_walk = 1 << 23;                               // more than 2^24 slots per walk
    REFUSED(GS_ELIMIT, gs_sage_fused_sampled_fwd(f, 10, 64, 0, f, 10, 64, 0, sg, 2, f, f, f, 32, GS_ACT_RELU, 0, f, nullptr, nullptr, &sp, dr, nullptr));
    dr[1].heads_per_walk = 25;
    sg[0].n_child = 5; sg[1].n_child = 5;                         // wide layers read ids: no draws there
    REFUSED(GS_ELIMIT, gs_sage_fused_sampled_fwd(f, 10, 256, 0, f, 10, 256, 0, sg, 2, f, f, f, 256, GS_ACT_RELU, 0, f, nullptr, nullptr, &sp, dr, nullptr));

    // ---- whole two-hop forward ----
    EXPECT(gs_sage2_supported(64, 64, 32, 32, 25, 10) == 1);
    EXPECT(gs_sage2_supported(24, 19, 32, 32, 10, 5) == 0);
    EXPECT(gs_sage2_supported(64, 64, 32, 32, 65, 10) == 0);
    EXPECT(gs_sage2_supported(64, 64, 32, 32, 25, 0) == 0);
    EXPECT(gs_sage2_supported(64, 64, 33, 32, 25, 10) == 0);
    gs_sage_weights w{f, f, f}, w_hole{f, nullptr, f};
    ACCEPTED(gs_sage2_forward(nullptr, nullptr, 0, nullptr, 0, 64, nullptr, 0, 64, nullptr, 0, 25, 10, 1, 0, nullptr, nullptr,
                              nullptr, nullptr, nullptr, nullptr, 32, 32, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
    REFUSED(GS_EINVAL, gs_sage2_forward(i64, i32, 100, f, 60, 64, f, 40, 64, i32, -1, 25, 10, 1, 0, nullptr, nullptr, &w, &w, &w, &w, 32, 32, 1, f, nullptr, nullptr, nullptr, i32, nullptr));
    REFUSED(GS_EINVAL, gs_sage2_forward(i64, i32, 100, f, 60, 64, f, 40, 64, i32, 8, 25, 10, 1, 0, nullptr, nullptr, &w, nullptr, &w, &w, 32, 32, 1, f, nullptr, nullptr, nullptr, i32, nullptr));
    REFUSED(GS_EINVAL, gs_sage2_forward(i64, i32, 100, f, 60, 64, f, 40, 64, i32, 8, 25, 10, 1, 0, nullptr, nullptr, &w, &w_hole, &w, &w, 32, 32, 1, f, nullptr, nullptr, nullptr, i32, nullptr));
    REFUSED(GS_EINVAL, gs_sage2_forward(i64, i32, 101, f, 60, 64, f, 40, 64, i32, 8, 25, 10, 1, 0, nullptr, nullptr, &w, &w, &w, &w, 32, 32, 1, f, nullptr, nullptr, nullptr, i32, nullptr));
    REFUSED(GS_EINVAL, gs_sage2_forward(i64, i32, 100, odd, 60, 64, f, 40, 64, i32, 8, 25, 10, 1, 0, nullptr, nullptr, &w, &w, &w, &w, 32, 32, 1, f, nullptr, nullptr, nullptr, i32, nullptr));
    REFUSED(GS_ELIMIT, gs_sage2_forward(i64, i32, 100, f, 60, 24, f, 40, 19, i32, 8, 25, 10, 1, 0, nullptr, nullptr, &w, &w, &w, &w, 32, 32, 1, f, nullptr, nullptr, nullptr, i32, nullptr));
    gs_sage2_saved sv{};
    sv.x_mean[0] = sv.x_mean[1] = f; sv.h1[0] = sv.h1[1] = f; sv.x_mean1[0] = sv.x_mean1[1] = f; sv.z[0] = sv.z[1] = f;
    REFUSED(GS_EINVAL, gs_sage2_forward(i64, i32, 100, f, 60, 64, f, 40, 64, i32, 8, 25, 10, 1, 0, nullptr, nullptr, &w, &w, &w, &w, 32, 32, 1, f, nullptr, nullptr, &sv, i32, nullptr));   // neither own rows nor ids
    sv.self_ids[0] = sv.self_ids[1] = i32; sv.x_mean[1] = odd;
    REFUSED(GS_EINVAL, gs_sage2_forward(i64, i32, 100, f, 60, 64, f, 40, 64, i32, 8, 25, 10, 1, 0, nullptr, nullptr, &w, &w, &w, &w, 32, 32, 1, f, nullptr, nullptr, &sv, i32, nullptr));
    sv.x_mean[1] = f; sv.h1_mask[0] = reinterpret_cast<uint32_t *>(i32);          // mask words for one node type only
    REFUSED(GS_EINVAL, gs_sage2_forward(i64, i32, 100, f, 60, 64, f, 40, 64, i32, 8, 25, 10, 1, 0, nullptr, nullptr, &w, &w, &w, &w, 32, 32, 1, f, nullptr, nullptr, &sv, i32, nullptr));

    // ---- the same kernel below the first hop of a three-hop model ----
    {
        gs_sage2_subtrees sub{};
        sub.roots_per_walk = 25; sub.tape_len = 25 + 250 + 1250; sub.tape_off_hop2 = 25; sub.tape_off_hop3 = 275;
        ACCEPTED(gs_sage2_forward_subtrees(i64, i32, 100, f, 60, 64, f, 40, 64, i32, 0, 10, 5, 1, 0, nullptr, nullptr, &w, &w, &w, &w, 32, 32, f, nullptr, nullptr, &sub, i32, nullptr));
        REFUSED(GS_EINVAL, gs_sage2_forward_subtrees(i64, i32, 100, f, 60, 64, f, 40, 64, i32, 8, 10, 5, 1, 0, nullptr, nullptr, &w, &w, &w, &w, 32, 32, f, nullptr, nullptr, nullptr, i32, nullptr));
        REFUSED(GS_ELIMIT, gs_sage2_forward_subtrees(i64, i32, 100, f, 60, 24, f, 40, 19, i32, 8, 10, 5, 1, 0, nullptr, nullptr, &w, &w, &w, &w, 32, 32, f, nullptr, nullptr, &sub, i32, nullptr));
        sub.tape_off_hop3 = sub.tape_len - 10;                  // hop 3 runs past the end of a walk's tape
        REFUSED(GS_EINVAL, gs_sage2_forward_subtrees(i64, i32, 100, f, 60, 64, f, 40, 64, i32, 8, 10, 5, 1, 0, nullptr,
                                                     reinterpret_cast<const uint32_t *>(i32), &w, &w, &w, &w, 32, 32, f, nullptr, nullptr, &sub, i32, nullptr));
        sub.tape_off_hop3 = 275; sub.roots_per_walk = 0;
        REFUSED(GS_EINVAL, gs_sage2_forward_subtrees(i64, i32, 100, f, 60, 64, f, 40, 64, i32, 8, 10, 5, 1, 0, nullptr, nullptr, &w, &w, &w, &w, 32, 32, f, nullptr, nullptr, &sub, i32, nullptr));
        sub.roots_per_walk = 25; sub.out_h0 = odd;
        REFUSED(GS_EINVAL, gs_sage2_forward_subtrees(i64, i32, 100, f, 60, 64, f, 40, 64, i32, 8, 10, 5, 1, 0, nullptr, nullptr, &w, &w, &w, &w, 32, 32, f, nullptr, nullptr, &sub, i32, nullptr));
        sub.out_h0 = f;
        // neither hop-1 ids nor seeds to draw them from; seeds without a place for the drawn ids
        REFUSED(GS_EINVAL, gs_sage2_forward_subtrees(i64, i32, 100, f, 60, 64, f, 40, 64, nullptr, 8, 10, 5, 1, 0, nullptr, nullptr, &w, &w, &w, &w, 32, 32, f, nullptr, nullptr, &sub, i32, nullptr));
        sub.seeds = i32;
        REFUSED(GS_EINVAL, gs_sage2_forward_subtrees(i64, i32, 100, f, 60, 64, f, 40, 64, nullptr, 8, 10, 5, 1, 0, nullptr, nullptr, &w, &w, &w, &w, 32, 32, f, nullptr, nullptr, &sub, i32, nullptr));
        // the saving form: all eight tables or none, aligned, 32 units in layer 0
        sub.seeds = nullptr;
        sub.sv_x_mean0_hop2[0] = f;
        REFUSED(GS_EINVAL, gs_sage2_forward_subtrees(i64, i32, 100, f, 60, 64, f, 40, 64, i32, 8, 10, 5, 1, 0, nullptr, nullptr, &w, &w, &w, &w, 32, 32, f, nullptr, nullptr, &sub, i32, nullptr));
        for (int t = 0; t < 2; ++t) {
            sub.sv_x_mean0_hop2[t] = sub.sv_x_mean0_hop1[t] = sub.sv_x_mean1_hop1[t] = f;
            sub.sv_mask_hop2[t] = reinterpret_cast<uint32_t *>(i32);
        }
        REFUSED(GS_ELIMIT, gs_sage2_forward_subtrees(i64, i32, 100, f, 60, 64, f, 40, 64, i32, 8, 10, 5, 1, 0, nullptr, nullptr, &w, &w, &w, &w, 64, 32, f, nullptr, nullptr, &sub, i32, nullptr));
        sub.sv_x_mean1_hop1[1] = odd;
        REFUSED(GS_EINVAL, gs_sage2_forward_subtrees(i64, i32, 100, f, 60, 64, f, 40, 64, i32, 8, 10, 5, 1, 0, nullptr, nullptr, &w, &w, &w, &w, 32, 32, f, nullptr, nullptr, &sub, i32, nullptr));
        sub.sv_x_mean1_hop1[1] = f;
        sub.sv_mask_hop2[0] = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(i32) + 4);       // 8-byte alignment
        REFUSED(GS_EINVAL, gs_sage2_forward_subtrees(i64, i32, 100, f, 60, 64, f, 40, 64, i32, 8, 10, 5, 1, 0, nullptr, nullptr, &w, &w, &w, &w, 32, 32, f, nullptr, nullptr, &sub, i32, nullptr));
    }

    EXPECT(gs_sage2_subtrees_supported(64, 64, 32, 32, 10, 5) == 1);
    EXPECT(gs_sage2_subtrees_supported(64, 64, 32, 32, 60, 60) == 0);      // supported as whole walks, too many ids for eight subtrees
    EXPECT(gs_sage2_supported(64, 64, 32, 32, 60, 60) == 1);
    EXPECT(gs_sage2_subtrees_supported(24, 19, 32, 32, 10, 5) == 0);

    // ---- layers 0, 1, 2 at the seeds of a three-hop model ----
    EXPECT(gs_sage3_seeds_supported(64, 64, 32, 32, 32, 25) == 1);
    EXPECT(gs_sage3_seeds_supported(64, 64, 32, 32, 48, 25) == 0);
    EXPECT(gs_sage3_seeds_supported(24, 19, 32, 32, 32, 25) == 0);
    EXPECT(gs_sage3_seeds_supported(64, 64, 32, 32, 32, 0) == 0);
    ACCEPTED(gs_sage3_seeds_forward(f, 60, 64, f, 40, 64, i32, 0, i32, 25, f, f, &w, &w, &w, &w, &w, &w, 32, 32, 32, GS_ACT_LINEAR, 1, f, nullptr));
    REFUSED(GS_EINVAL, gs_sage3_seeds_forward(f, 60, 64, f, 40, 64, i32, -1, i32, 25, f, f, &w, &w, &w, &w, &w, &w, 32, 32, 32, GS_ACT_LINEAR, 1, f, nullptr));
    REFUSED(GS_EINVAL, gs_sage3_seeds_forward(f, 60, 64, f, 40, 64, nullptr, 8, i32, 25, f, f, &w, &w, &w, &w, &w, &w, 32, 32, 32, GS_ACT_LINEAR, 1, f, nullptr));
    REFUSED(GS_EINVAL, gs_sage3_seeds_forward(f, 60, 64, f, 40, 64, i32, 8, i32, 25, f, f, &w, &w, &w, &w, &w, nullptr, 32, 32, 32, GS_ACT_LINEAR, 1, f, nullptr));
    REFUSED(GS_EINVAL, gs_sage3_seeds_forward(f, 60, 64, f, 40, 64, i32, 8, i32, 25, f, f, &w, &w, &w_hole, &w, &w, &w, 32, 32, 32, GS_ACT_LINEAR, 1, f, nullptr));
    REFUSED(GS_EINVAL, gs_sage3_seeds_forward(f, 60, 64, f, 40, 64, i32, 8, i32, 25, f, odd, &w, &w, &w, &w, &w, &w, 32, 32, 32, GS_ACT_LINEAR, 1, f, nullptr));
    REFUSED(GS_EINVAL, gs_sage3_seeds_forward(f, 60, 64, f, 40, 64, i32, 8, i32, 25, f, f, &w, &w, &w, &w, &w, &w, 32, 32, 32, 7, 1, f, nullptr));
    REFUSED(GS_EINVAL, gs_sage3_seeds_forward(f, 0, 64, f, 40, 64, i32, 8, i32, 25, f, f, &w, &w, &w, &w, &w, &w, 32, 32, 32, GS_ACT_LINEAR, 1, f, nullptr));
    REFUSED(GS_ELIMIT, gs_sage3_seeds_forward(f, 60, 64, f, 40, 64, i32, 8, i32, 25, f, f, &w, &w, &w, &w, &w, &w, 32, 32, 48, GS_ACT_LINEAR, 1, f, nullptr));
    {
        gs_sage3_seeds_saved ss{};
        REFUSED(GS_EINVAL, gs_sage3_seeds_forward_saving(f, 60, 64, f, 40, 64, i32, 8, i32, 25, f, f, &w, &w, &w, &w, &w, &w, 32, 32, 32, GS_ACT_LINEAR, 1, f, nullptr, nullptr));
        REFUSED(GS_EINVAL, gs_sage3_seeds_forward_saving(f, 60, 64, f, 40, 64, i32, 8, i32, 25, f, f, &w, &w, &w, &w, &w, &w, 32, 32, 32, GS_ACT_LINEAR, 1, f, &ss, nullptr));
        for (int t = 0; t < 2; ++t) ss.x_mean0[t] = ss.h0[t] = ss.x_mean1[t] = ss.h1[t] = ss.x_mean2[t] = ss.z[t] = f;
        ACCEPTED(gs_sage3_seeds_forward_saving(f, 60, 64, f, 40, 64, i32, 0, i32, 25, f, f, &w, &w, &w, &w, &w, &w, 32, 32, 32, GS_ACT_LINEAR, 1, f, &ss, nullptr));
        ss.h1[1] = odd;
        REFUSED(GS_EINVAL, gs_sage3_seeds_forward_saving(f, 60, 64, f, 40, 64, i32, 8, i32, 25, f, f, &w, &w, &w, &w, &w, &w, 32, 32, 32, GS_ACT_LINEAR, 1, f, &ss, nullptr));
        ss.h1[1] = f;
        REFUSED(GS_EINVAL, gs_sage3_seeds_forward_saving(f, 60, 64, f, 40, 64, nullptr, 8, i32, 25, f, f, &w, &w, &w, &w, &w, &w, 32, 32, 32, GS_ACT_LINEAR, 1, f, &ss, nullptr));
    }

    // ---- top of the backward pass ----
    ACCEPTED(gs_sage2_backward_tail(f, f, f, f, 0, 25, 32, 32, 1, GS_ACT_LINEAR, f, f, f, f, f, f, f, f, nullptr, nullptr));
    REFUSED(GS_EINVAL, gs_sage2_backward_tail(f, f, f, f, -1, 25, 32, 32, 1, GS_ACT_LINEAR, f, f, f, f, f, f, f, f, nullptr, nullptr));
    REFUSED(GS_EINVAL, gs_sage2_backward_tail(f, f, f, f, 8, 0, 32, 32, 1, GS_ACT_LINEAR, f, f, f, f, f, f, f, f, nullptr, nullptr));
    REFUSED(GS_ELIMIT, gs_sage2_backward_tail(f, f, f, f, 8, 25, 48, 32, 1, GS_ACT_LINEAR, f, f, f, f, f, f, f, f, nullptr, nullptr));
    REFUSED(GS_EINVAL, gs_sage2_backward_tail(f, f, f, f, 8, 25, 32, 32, 1, 5, f, f, f, f, f, f, f, f, nullptr, nullptr));
    REFUSED(GS_EINVAL, gs_sage2_backward_tail(f, nullptr, f, f, 8, 25, 32, 32, 1, GS_ACT_LINEAR, f, f, f, f, f, f, f, f, nullptr, nullptr));
    REFUSED(GS_EINVAL, gs_sage2_backward_tail(f, f, f, f, 8, 25, 32, 32, 1, GS_ACT_LINEAR, f, f, f, f, f, f, odd, f, nullptr, nullptr));
    REFUSED(GS_ELIMIT, gs_sage2_backward_tail(f, f, f, f, (int64_t)1 << 39, 25, 32, 32, 1, GS_ACT_LINEAR, f, f, f, f, f, f, f, f, nullptr, nullptr));
    {
        gs_sage2_l0_mask lm{};                                  // a mask set with a missing pointer
        lm.h1[0] = lm.h1[1] = f; lm.h1_mask[0] = reinterpret_cast<const uint32_t *>(f);
        REFUSED(GS_EINVAL, gs_sage2_backward_tail(f, f, f, f, 8, 25, 32, 32, 1, GS_ACT_LINEAR, f, f, f, f, f, f, f, f, &lm, nullptr));
    }
    EXPECT(gs_sage2_tail_head_workspace(-1, 32) == 0);
    EXPECT(gs_sage2_tail_head_rows(1) == 1 && gs_sage2_tail_head_rows(4096) == 256 && gs_sage2_tail_head_rows((int64_t)1 << 40) == 256);
    const size_t th = gs_sage2_tail_head_workspace(4096, 32);
    EXPECT(th == (size_t)256 * 66 * 4);
    REFUSED(GS_EINVAL, gs_sage2_backward_tail_head(f, f, f, f, f, f, f, 0, 25, 32, 32, 1, GS_ACT_LINEAR, f, f, f, f, f, f, f, f, f, arena, th, nullptr, nullptr));
    REFUSED(GS_ELIMIT, gs_sage2_backward_tail_head(f, f, f, f, f, f, f, 8, 25, 32, 128, 1, GS_ACT_LINEAR, f, f, f, f, f, f, f, f, f, arena, th, nullptr, nullptr));
    REFUSED(GS_EINVAL, gs_sage2_backward_tail_head(f, f, f, f, nullptr, f, f, 8, 25, 32, 32, 1, GS_ACT_LINEAR, f, f, f, f, f, f, f, f, f, arena, th, nullptr, nullptr));
    REFUSED(GS_EINVAL, gs_sage2_backward_tail_head(f, f, f, f, f, f, f, 4096, 25, 32, 32, 1, GS_ACT_LINEAR, f, f, f, f, f, f, f, f, f, arena, th - 4, nullptr, nullptr));

    // ---- the whole two-hop backward pass in one launch ----
    EXPECT(gs_sage2_backward_all_supported(64, 64, 32, 32) == 1 && gs_sage2_backward_all_supported(32, 32, 32, 32) == 1);
    EXPECT(gs_sage2_backward_all_supported(64, 32, 32, 32) == 0 && gs_sage2_backward_all_supported(64, 64, 64, 32) == 0 &&
           gs_sage2_backward_all_supported(64, 64, 32, 64) == 0 && gs_sage2_backward_all_supported(128, 128, 32, 32) == 0);
    EXPECT(gs_sage2_backward_all_workspace(0, 64, 32, 32) == 0 && gs_sage2_backward_all_workspace(4096, 128, 32, 32) == 0);
    const size_t ba = gs_sage2_backward_all_workspace(4096, 64, 32, 32);
    EXPECT(ba == ((size_t)256 * (2 * (2 * 64 * 16 + 32) + 2 * (2 * 32 * 16 + 32) + 66) + (size_t)2 * 4096 * 32) * 4);
    EXPECT(gs_sage2_backward_all_workspace(3, 32, 32, 32) == ((size_t)1 * (2 * (2 * 32 * 16 + 32) + 2 * (2 * 32 * 16 + 32) + 66) + 2 * 3 * 32) * 4);
    {
        gs_sage2_bwd_all p{};
        REFUSED(GS_EINVAL, gs_sage2_backward_all(nullptr, 8, 25, 64, 32, 32, 1, GS_ACT_LINEAR, arena, ba, nullptr, nullptr));
        REFUSED(GS_EINVAL, gs_sage2_backward_all(&p, 0, 25, 64, 32, 32, 1, GS_ACT_LINEAR, arena, ba, nullptr, nullptr));
        REFUSED(GS_EINVAL, gs_sage2_backward_all(&p, 8, 0, 64, 32, 32, 1, GS_ACT_LINEAR, arena, ba, nullptr, nullptr));
        REFUSED(GS_ELIMIT, gs_sage2_backward_all(&p, 8, 25, 128, 32, 32, 1, GS_ACT_LINEAR, arena, ba, nullptr, nullptr));
        REFUSED(GS_ELIMIT, gs_sage2_backward_all(&p, 8, 25, 64, 32, 64, 1, GS_ACT_LINEAR, arena, ba, nullptr, nullptr));
        REFUSED(GS_EINVAL, gs_sage2_backward_all(&p, 8, 25, 64, 32, 32, 1, 77, arena, ba, nullptr, nullptr));
        REFUSED(GS_ELIMIT, gs_sage2_backward_all(&p, 8, 65, 64, 32, 32, 1, GS_ACT_LINEAR, arena, ba, nullptr, nullptr));                  // a lane per hop-1 head
        REFUSED(GS_ELIMIT, gs_sage2_backward_all(&p, (int64_t)1 << 21, 25, 64, 32, 32, 1, GS_ACT_LINEAR, arena, ba, nullptr, nullptr));   // 32-bit row offsets
        REFUSED(GS_EINVAL, gs_sage2_backward_all(&p, 8, 25, 64, 32, 32, 1, GS_ACT_LINEAR, arena, ba, nullptr, nullptr));   // null tables
        for (int t = 0; t < 2; ++t) {
            p.z[t] = p.w1_self[t] = p.w1_neigh[t] = p.h1[t] = p.x_mean1[t] = p.x_self[t] = p.x_mean[t] = f;
            p.g_emb[t] = f;
        }
        REFUSED(GS_EINVAL, gs_sage2_backward_all(&p, 4096, 25, 64, 32, 32, 1, GS_ACT_LINEAR, arena, ba - 4, nullptr, nullptr));
        REFUSED(GS_EINVAL, gs_sage2_backward_all(&p, 8, 25, 64, 32, 32, 1, GS_ACT_LINEAR, nullptr, ba, nullptr, nullptr));
        p.g_emb[1] = nullptr;
        REFUSED(GS_EINVAL, gs_sage2_backward_all(&p, 8, 25, 64, 32, 32, 1, GS_ACT_LINEAR, arena, ba, nullptr, nullptr));
        p.g_emb[1] = f;
        p.targets = f;                                           // the head form: emb, head weights and outputs are needed
        REFUSED(GS_EINVAL, gs_sage2_backward_all(&p, 8, 25, 64, 32, 32, 1, GS_ACT_LINEAR, arena, ba, nullptr, nullptr));
        p.emb[0] = p.emb[1] = f;
        REFUSED(GS_EINVAL, gs_sage2_backward_all(&p, 8, 25, 64, 32, 32, 1, GS_ACT_LINEAR, arena, ba, nullptr, nullptr));
        p.targets = nullptr;
        p.self_ids[0] = i32;                                     // ids for one node type only
        REFUSED(GS_EINVAL, gs_sage2_backward_all(&p, 8, 25, 64, 32, 32, 1, GS_ACT_LINEAR, arena, ba, nullptr, nullptr));
        p.self_ids[1] = i32;                                     // ids without the table's row count
        REFUSED(GS_EINVAL, gs_sage2_backward_all(&p, 8, 25, 64, 32, 32, 1, GS_ACT_LINEAR, arena, ba, nullptr, nullptr));
        p.self_ids[0] = p.self_ids[1] = nullptr;
        p.h1_mask[1] = reinterpret_cast<const uint32_t *>(f);    // mask words for one node type only
        REFUSED(GS_EINVAL, gs_sage2_backward_all(&p, 8, 25, 64, 32, 32, 1, GS_ACT_LINEAR, arena, ba, nullptr, nullptr));
        p.h1_mask[1] = nullptr;
        {
            float *gw = reinterpret_cast<float *>(arena + 3072);
            for (int t = 0; t < 2; ++t) p.g_w0_self[t] = p.g_w0_neigh[t] = p.g_w1_self[t] = p.g_w1_neigh[t] = gw;
            gs_adam_update ad{};
            REFUSED(GS_EINVAL, gs_sage2_backward_all(&p, 8, 25, 64, 32, 32, 1, GS_ACT_LINEAR, arena, ba, &ad, nullptr));   // empty
            ad.grads = f; ad.params = ad.m = ad.v = f; ad.n = 16;                                  // gradients outside grads[0 .. n)
            REFUSED(GS_EINVAL, gs_sage2_backward_all(&p, 8, 25, 64, 32, 32, 1, GS_ACT_LINEAR, arena, ba, &ad, nullptr));
            for (int t = 0; t < 2; ++t) p.g_w0_self[t] = p.g_w0_neigh[t] = p.g_w1_self[t] = p.g_w1_neigh[t] = nullptr;
        }
        p.x_mean[0] = reinterpret_cast<const float *>(reinterpret_cast<const char *>(f) + 4);   // not 16-byte aligned
        REFUSED(GS_EINVAL, gs_sage2_backward_all(&p, 8, 25, 64, 32, 32, 1, GS_ACT_LINEAR, arena, ba, nullptr, nullptr));
    }

    // ---- the whole three-hop backward pass (32 units per layer) ----
    EXPECT(gs_sage3_backward_all_supported(64, 64, 32, 32, 32, 25, 10) == 1 && gs_sage3_backward_all_supported(32, 32, 32, 32, 32, 1, 32) == 1);
    EXPECT(gs_sage3_backward_all_supported(64, 32, 32, 32, 32, 25, 10) == 0 && gs_sage3_backward_all_supported(64, 64, 32, 64, 32, 25, 10) == 0 &&
           gs_sage3_backward_all_supported(128, 128, 32, 32, 32, 25, 10) == 0 && gs_sage3_backward_all_supported(64, 64, 32, 32, 32, 25, 33) == 0);
    EXPECT(gs_sage3_backward_all_workspace(0, 25, 64) == 0 && gs_sage3_backward_all_workspace(8, 25, 128) == 0);
    const size_t b3 = gs_sage3_backward_all_workspace(4096, 25, 64);
    {
        const size_t row = 2 * (2 * 64 * 16 + 32) + 4 * 32 + 66 + 4 * 1024;
        EXPECT(b3 == ((size_t)(256 + 256) * row + (size_t)6 * 4096 * 32 + (size_t)2 * 4096 * 25 * 32) * 4);
        gs_sage3_bwd p{};
        REFUSED(GS_EINVAL, gs_sage3_backward_all(nullptr, 8, 25, 10, 64, 1, GS_ACT_LINEAR, arena, b3, nullptr, nullptr));
        REFUSED(GS_EINVAL, gs_sage3_backward_all(&p, 0, 25, 10, 64, 1, GS_ACT_LINEAR, arena, b3, nullptr, nullptr));
        REFUSED(GS_EINVAL, gs_sage3_backward_all(&p, 8, 25, 0, 64, 1, GS_ACT_LINEAR, arena, b3, nullptr, nullptr));
        REFUSED(GS_ELIMIT, gs_sage3_backward_all(&p, 8, 25, 10, 128, 1, GS_ACT_LINEAR, arena, b3, nullptr, nullptr));
        REFUSED(GS_ELIMIT, gs_sage3_backward_all(&p, 8, 25, 33, 64, 1, GS_ACT_LINEAR, arena, b3, nullptr, nullptr));
        REFUSED(GS_EINVAL, gs_sage3_backward_all(&p, 8, 25, 10, 64, 1, 77, arena, b3, nullptr, nullptr));
        REFUSED(GS_ELIMIT, gs_sage3_backward_all(&p, (int64_t)1 << 18, 25, 10, 64, 1, GS_ACT_LINEAR, arena, b3, nullptr, nullptr));   // 32-bit row offsets
        REFUSED(GS_EINVAL, gs_sage3_backward_all(&p, 8, 25, 10, 64, 1, GS_ACT_LINEAR, arena, b3, nullptr, nullptr));   // null tables
        for (int t = 0; t < 2; ++t) {
            p.z[t] = p.g_emb[t] = p.w2_self[t] = p.w2_neigh[t] = p.w1_self[t] = p.w1_neigh[t] = f;
            p.h0[t] = p.h1[t] = p.x_mean0[t] = p.x_mean1[t] = p.x_mean2[t] = p.feat[t] = f;
            p.row_ids[t] = i32;
        }
        REFUSED(GS_EINVAL, gs_sage3_backward_all(&p, 8, 25, 10, 64, 1, GS_ACT_LINEAR, arena, b3, nullptr, nullptr));   // feat_rows
        p.feat_rows[0] = p.feat_rows[1] = 100;
        REFUSED(GS_EINVAL, gs_sage3_backward_all(&p, 4096, 25, 10, 64, 1, GS_ACT_LINEAR, arena, b3 - 4, nullptr, nullptr));
        REFUSED(GS_EINVAL, gs_sage3_backward_all(&p, 8, 25, 10, 64, 1, GS_ACT_LINEAR, nullptr, b3, nullptr, nullptr));
        p.targets = f;                                           // the head form needs emb, head weights and outputs
        REFUSED(GS_EINVAL, gs_sage3_backward_all(&p, 8, 25, 10, 64, 1, GS_ACT_LINEAR, arena, b3, nullptr, nullptr));
        p.emb[0] = p.emb[1] = f;
        REFUSED(GS_EINVAL, gs_sage3_backward_all(&p, 8, 25, 10, 64, 1, GS_ACT_LINEAR, arena, b3, nullptr, nullptr));
        p.targets = nullptr;
        gs_adam_update ad{};
        REFUSED(GS_EINVAL, gs_sage3_backward_all(&p, 8, 25, 10, 64, 1, GS_ACT_LINEAR, arena, b3, &ad, nullptr));       // empty
        ad.grads = f; ad.params = ad.m = ad.v = f; ad.n = 16;
        p.g_w2_self[0] = reinterpret_cast<float *>(arena + 3072);                                                     // outside grads[0 .. n)
        REFUSED(GS_EINVAL, gs_sage3_backward_all(&p, 8, 25, 10, 64, 1, GS_ACT_LINEAR, arena, b3, &ad, nullptr));
        p.g_w2_self[0] = nullptr;
        p.x_mean0[1] = reinterpret_cast<const float *>(reinterpret_cast<const char *>(f) + 4);                        // not 16-byte aligned
        REFUSED(GS_EINVAL, gs_sage3_backward_all(&p, 8, 25, 10, 64, 1, GS_ACT_LINEAR, arena, b3, nullptr, nullptr));
        p.x_mean0[1] = f;
        // the rows below the seeds apart: hop1 needs hop2, every table of either, hop2 its outputs as rows or as masks
        gs_sage3_bwd_hop1 h1{};
        gs_sage3_bwd_hop2 h2{};
        p.hop1 = &h1;
        REFUSED(GS_EINVAL, gs_sage3_backward_all(&p, 8, 25, 10, 64, 1, GS_ACT_LINEAR, arena, b3, nullptr, nullptr));   // empty hop1
        for (int t = 0; t < 2; ++t) { h1.h0[t] = h1.h1[t] = h1.x_mean0[t] = h1.x_mean1[t] = f; h1.ids[t] = i32; }
        REFUSED(GS_EINVAL, gs_sage3_backward_all(&p, 8, 25, 10, 64, 1, GS_ACT_LINEAR, arena, b3, nullptr, nullptr));   // no hop2
        p.hop2 = &h2;
        for (int t = 0; t < 2; ++t) { h2.x_mean0[t] = f; h2.ids[t] = i32; }
        REFUSED(GS_EINVAL, gs_sage3_backward_all(&p, 8, 25, 10, 64, 1, GS_ACT_LINEAR, arena, b3, nullptr, nullptr));   // neither h0 nor mask
        h2.mask[0] = h2.mask[1] = reinterpret_cast<const uint32_t *>(i32);
        h2.x_mean0[0] = reinterpret_cast<const float *>(reinterpret_cast<const char *>(f) + 4);
        REFUSED(GS_EINVAL, gs_sage3_backward_all(&p, 8, 25, 10, 64, 1, GS_ACT_LINEAR, arena, b3, nullptr, nullptr));   // alignment
    }

    // ---- link head, optimizer, staging ----
    EXPECT(gs_link_head_workspace(-1, 32) == 0 && gs_link_head_workspace(4096, 32) == (size_t)256 * 66 * 4);
    ACCEPTED(gs_link_head_fwd(nullptr, nullptr, 0, 32, nullptr, nullptr, nullptr, nullptr));
    REFUSED(GS_EINVAL, gs_link_head_fwd(f, f, -1, 32, f, f, f, nullptr));
    REFUSED(GS_ELIMIT, gs_link_head_fwd(f, f, 8, 300, f, f, f, nullptr));
    REFUSED(GS_EINVAL, gs_link_head_fwd(f, nullptr, 8, 32, f, f, f, nullptr));
    REFUSED(GS_EINVAL, gs_link_head_mse_fwd_bwd(f, f, 0, 32, f, f, f, f, f, f, f, f, f, arena, 4096, nullptr));
    REFUSED(GS_ELIMIT, gs_link_head_mse_fwd_bwd(f, f, 8, 257, f, f, f, f, f, f, f, f, f, arena, 4096, nullptr));
    REFUSED(GS_EINVAL, gs_link_head_mse_fwd_bwd(f, f, 8, 32, f, f, f, f, f, f, f, f, f, nullptr, 4096, nullptr));
    REFUSED(GS_EINVAL, gs_link_head_mse_fwd_bwd(f, f, 4096, 32, f, f, f, f, f, f, f, f, f, arena, 100, nullptr));
    REFUSED(GS_EINVAL, gs_link_head_mse_fwd_bwd_static(f, f, 0, 32, f, f, f, f, f, f, f, f, f, nullptr));
    REFUSED(GS_ELIMIT, gs_link_head_mse_fwd_bwd_static(f, f, 8, 257, f, f, f, f, f, f, f, f, f, nullptr));
    REFUSED(GS_EINVAL, gs_link_head_reduce(arena, 0, 32, 8, f, f, f, nullptr));
    REFUSED(GS_EINVAL, gs_link_head_reduce(nullptr, 4, 32, 8, f, f, f, nullptr));
    REFUSED(GS_EINVAL, gs_link_head_reduce(arena, 4, 32, 0, f, f, f, nullptr));
    ACCEPTED(gs_adam_step(nullptr, nullptr, nullptr, nullptr, 0, 0.01f, nullptr, 0.9f, 0.999f, 1e-7f, 1.f, nullptr));
    REFUSED(GS_EINVAL, gs_adam_step(f, f, f, f, -3, 0.01f, nullptr, 0.9f, 0.999f, 1e-7f, 1.f, nullptr));
    REFUSED(GS_EINVAL, gs_adam_step(f, nullptr, f, f, 10, 0.01f, nullptr, 0.9f, 0.999f, 1e-7f, 1.f, nullptr));
    ACCEPTED(gs_stage_batch(nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, 0, 0.f, nullptr, nullptr));
    REFUSED(GS_EINVAL, gs_stage_batch(i32, i32, -1, i64, i64, 0, f, f, 0, 0.f, nullptr, nullptr));
    REFUSED(GS_EINVAL, gs_stage_batch(i32, nullptr, 8, i64, i64, 0, f, f, 0, 0.f, nullptr, nullptr));

    // the message of the last refusal stays readable until the next one
    REFUSED(GS_EINVAL, gs_l2_normalize_fwd(f, 10, 0, f, nullptr));
    EXPECT(std::strstr(gs_last_error(), "gs_l2_normalize_fwd") != nullptr);

    if (failures) { std::printf("%d check(s) failed\n", failures); return 1; }
    std::printf("abi_validate ok\n");
    return 0;
}
