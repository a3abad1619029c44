/*
 * libgsage.so -- C ABI of the MI355X (gfx950) bipartite GraphSAGE hot path.
 *
 * The reference project is pure Python and has no FFI of its own; the Python
 * host in graphsage_amd/ keeps the reference's classes and calls the entry
 * points below through ctypes.  Each entry point names the reference lines
 * whose arithmetic it replaces.
 *
 * Conventions
 *   - every pointer marked "device" is device memory owned by the caller; the
 *     library never allocates, frees or keeps device memory;
 *   - work is enqueued on the caller's hipStream_t (`stream`, may be NULL for
 *     the default stream); no entry point synchronises;
 *   - node ids are int32, row counts and CSR offsets int64;
 *   - return value 0 = GS_OK, negative = error, text in gs_last_error()
 *     (thread-local, valid until the next failing call on that thread);
 *   - the only global state is that thread-local error text: entry points are
 *     re-entrant and may be called from several host threads and streams.
 */
#ifndef GSAGE_H_
#define GSAGE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_OK        0
#define GS_EINVAL   (-1)   /* null pointer, negative size, unsupported shape    */
#define GS_ELAUNCH  (-2)   /* the HIP runtime refused the launch                */
#define GS_ELIMIT   (-3)   /* a size exceeds what the kernels were built for   */

#define GS_MAX_HOPS  4

/* activation codes of gs_sage_project_*: models.py:69 uses "relu" for all but
 * the last layer and "linear" for the last. */
#define GS_ACT_LINEAR 0
#define GS_ACT_RELU   1

int gs_version(void);
const char *gs_last_error(void);

/*
 * Exact timing of one launch, for benchmarks (no counterpart in the reference).
 * A timer is a pair of HIP events.  gs_timer_arm(t) makes the calling thread's
 * next launch of one of the path's main kernels (gs_sage2_forward,
 * gs_sage_fused_fwd, gs_gather_mean_fwd / gs_gather_max_fwd) carry the pair in
 * its own dispatch, so start and stop are the kernel's first and last cycle and
 * nothing extra is enqueued; the timer disarms itself.  gs_timer_elapsed_us
 * waits for that launch and returns its duration.  gs_timer_arm(NULL) disarms.
 */
int gs_timer_create(void **timer);
int gs_timer_arm(void *timer);
int gs_timer_elapsed_us(void *timer, float *microseconds);
int gs_timer_destroy(void *timer);

/*
 * Adjacency build: edge columns -> CSR.  Replaces Graph.get_neighbor_dict
 * (bipartite_graph.py:48-60: one np.where scan of the edge array per node) and,
 * with symmetric != 0, the edge doubling of train_movielens.py:130-132
 * (edges = concat(edges, flip(edges))).
 *
 *   a, b      device int32[n_pairs]: entry i is the edge a[i] -> b[i]; with
 *             symmetric != 0 the entries are the n_pairs edges a -> b followed by
 *             the n_pairs edges b -> a, i.e. the reference's doubled edge array
 *   indptr    device int64[n_nodes+1] (written), indices device int32[entries]
 *             (written), entries = n_pairs * (symmetric ? 2 : 1): the neighbors of
 *             v are listed in the order of their entries, duplicates kept -- the
 *             order of neighbor_dict[v] (bipartite_graph.py:56-58)
 *   err_flag  device int32[1], zeroed by the caller: becomes 1 if an id lies
 *             outside [0, n_nodes); indices is then left unwritten
 *   sort_chunk entries per radix-sort pass, 0 = default (2^30)
 *   workspace device scratch of at least gs_csr_build_workspace(...) bytes
 * The result does not depend on timing (no atomics decide positions), so every
 * GPU of a data-parallel job builds the same replica.
 */
int gs_csr_build_workspace(int64_t n_pairs, int64_t n_nodes, int64_t sort_chunk, size_t *bytes);
int gs_csr_build(const int32_t *a, const int32_t *b, int64_t n_pairs, int symmetric, int64_t n_nodes,
                 int64_t *indptr, int32_t *indices, int32_t *err_flag, int64_t sort_chunk,
                 void *workspace, size_t workspace_bytes, void *stream);

/*
 * Multi-hop uniform neighbor sampling with replacement over a CSR.
 * Replaces BreadthFirstWalker.run_breadth_first_walk (sampler.py:47-96) and the
 * dict lookups it makes (bipartite_graph.py:62-68); writing hop h of all walks
 * as one dense [n_seeds, fanouts[0]*...*fanouts[h]] array in breadth-first
 * (= slot) order also replaces the regroup of generator.py:314-333.
 *
 *   indptr   device int64[n_nodes+1], indices device int32[indptr[n_nodes]]:
 *            neighbors of v are indices[indptr[v] .. indptr[v+1]) in the order
 *            the reference's neighbor_dict lists them (bipartite_graph.py:56-58)
 *   seeds    device int32[n_seeds] roots (the generator passes all source heads
 *            followed by all destination heads, generator.py:301-309)
 *   fanouts  HOST int32[n_hops], 1 <= n_hops <= GS_MAX_HOPS
 *   draw for (walk, hop, slot):
 *            replay != NULL: r = replay[w*T + off[hop] + slot], w = index of the
 *                     seed in this call, T = sum of the hop sizes, off[h] = sum
 *                     of the sizes of hops < h  (the order in which the
 *                     reference consumes its random stream, sampler.py:65-90)
 *            replay == NULL: r = Philox4x32-10(counter = (walk_lo, walk_hi, hop,
 *                     slot), key = rng_seed)[0], walk = walk_ids ? walk_ids[w]
 *                     : walk_id_base + w
 *            neighbor index = (r * degree) >> 32
 *   out_hops HOST array of n_hops device pointers, out_hops[h] = int32[n_seeds *
 *            size[h]]
 *   err_flag device int32[2], zeroed by the caller: [0] becomes 1 + the largest
 *            node id that was asked for samples but has no neighbors (the
 *            reference raises for it, sampler.py:81-82); [1] becomes 1 if a seed
 *            lies outside [0, n_nodes).  Samples below such a node are -1.
 */
int gs_sample(const int64_t *indptr, const int32_t *indices, int64_t n_nodes,
              const int32_t *seeds, int64_t n_seeds,
              const int32_t *fanouts, int n_hops,
              uint64_t rng_seed, uint64_t walk_id_base, const int64_t *walk_ids,
              const uint32_t *replay,
              int32_t *const *out_hops, int32_t *err_flag, void *stream);

/*
 * The reference's alternate walker (backup/sampler_backup.py:47-114): instead of
 * drawing with replacement, a visit of node v takes the next fanouts[hop] entries
 * of v's neighbor list starting at a per-node cursor (Graph.sampling_index_dict,
 * bipartite_graph.py:28,44-46), and when the cursor runs off the end of the list
 * the rest is taken from the start of a fresh permutation of it
 * (sampler_backup.py:85-96).  Arguments and output layout as gs_sample, plus
 *   cursor   device int64[n_nodes], read AND written: the cursors carry over from
 *            call to call, as the dict does in the reference
 *   state    device int64[2], read and written: [0] permutations drawn so far (the
 *            Philox counter of the next one), [1] tape words consumed so far
 *   perm_tape / perm_tape_len  nullable device int32: replay mode, the index
 *            permutations the reference's numpy.random.permutation calls returned,
 *            back to back in call order (entry i of a permutation p of a list x
 *            means permutation(x)[i] = x[p[i]]); without it permutation t is a
 *            Feistel network keyed by Philox4x32-10(counter (t, round), rng_seed)
 *   err_flag [0] as gs_sample (1 + id of a visited node without neighbors: the
 *            reference raises ValueError, sampler_backup.py:100); [1] = 1 for a root
 *            outside the graph, 2 when the tape is too short
 * A cursor depends on every earlier visit of its node, in the reference's visit
 * order (roots in call order, each walk breadth first), so the visits are
 * sequential by definition: one wavefront walks them in order.  It reproduces
 * that sampler; gs_sample is the fast one.
 */
int gs_sample_cyclic(const int64_t *indptr, const int32_t *indices, int64_t n_nodes,
                     const int32_t *seeds, int64_t n_seeds,
                     const int32_t *fanouts, int n_hops,
                     int64_t *cursor, uint64_t rng_seed, int64_t *state,
                     const int32_t *perm_tape, int64_t perm_tape_len,
                     int32_t *const *out_hops, int32_t *err_flag, void *stream);

/*
 * out[i, :] = feat[ids[i] - id_offset, :].
 * Replaces Graph.get_node_features_from_node (bipartite_graph.py:70-76; id_offset
 * is its correction_constant for items, 0 for users) as called from
 * GraphSAGEGenerator._get_features (generator.py:270-273).
 * Rows outside [0, n_rows) (e.g. the -1 ids below an isolated node) give zeros.
 */
int gs_gather_rows(const float *feat, int64_t n_rows, int F, int64_t id_offset,
                   const int32_t *ids, int64_t n, float *out, void *stream);

/*
 * out_mean[p, :] = (1/n_child) * sum_j feat[child_ids[p*n_child + j] - id_offset, :]
 * The feature gather above fused with the neighbor mean of Aggregator.call
 * (utils.py:103-109, K.mean over axis 2): the (B, heads, n, F) neighbor tensor
 * the reference materialises is never written.
 * child_ids == NULL means the children of parent p are rows p*n_child .. +n_child
 * of feat (the layer >= 1 case, where the neighbor input is already a dense
 * activation tensor, models.py:151-160).
 */
int gs_gather_mean_fwd(const float *feat, int64_t n_rows, int F, int64_t id_offset,
                       const int32_t *child_ids, int64_t n_parents, int n_child,
                       float *out_mean, void *stream);

/*
 * Backward of gs_gather_mean_fwd with respect to feat:
 *   g_feat[child - id_offset, :] += g_mean[p, :] / n_child      (float atomics)
 * child_ids == NULL: g_feat[p*n_child + j, :] = g_mean[p, :] / n_child (plain
 * stores, every row written exactly once).  No reference counterpart beyond
 * TensorFlow's autodiff of utils.py:109.
 */
int gs_gather_mean_bwd(const float *g_mean, const int32_t *child_ids, int64_t n_parents,
                       int n_child, int F, int64_t id_offset, int64_t n_rows,
                       float *g_feat, void *stream);

/*
 * The reference's other aggregation, agg_option='max' (utils.py:110-111:
 * K.max(x_neigh, axis=2)): per parent and feature, the maximum over the n_child
 * children.  Arguments as gs_gather_mean_fwd (child_ids NULL = dense groups); an
 * id outside the table contributes a row of zeros.
 */
int gs_gather_max_fwd(const float *feat, int64_t n_rows, int F, int64_t id_offset,
                      const int32_t *child_ids, int64_t n_parents, int n_child,
                      float *out_max, void *stream);

/*
 * Gradient of the dense group max: x device float[n_parents*n_child, F] the
 * children, out_max the forward result, g_max its gradient; g_x[child] =
 * g_max[parent] / (number of children attaining the maximum) where the child
 * attains it, else 0 -- TensorFlow's reduce_max rule (ties share evenly).
 * Feature tables are inputs of the model, not variables (models.py:241), so the
 * indexed form needs no gradient.
 */
int gs_group_max_bwd(const float *g_max, const float *x, const float *out_max, int64_t n_parents,
                     int n_child, int F, float *g_x, void *stream);

/*
 * out[m, :] = act(concat(x_self[m, :] @ W_self, x_mean[m, :] @ W_neigh) + bias)
 * The projection part of Aggregator.call (utils.py:109,114-119) with weights
 * shaped as Aggregator.build makes them (utils.py:72-89): W_self [Fs, H/2],
 * W_neigh [Fn, H/2], bias [H] or NULL, all row-major device float.  H even.
 * fp32 MFMA (v_mfma_f32_16x16x4_f32).
 */
int gs_sage_project_fwd(const float *x_self, int Fs, const float *x_mean, int Fn, int64_t M,
                        const float *W_self, const float *W_neigh, const float *bias,
                        int H, int act, float *out, void *stream);

/*
 * The same with the own rows taken from a feature table by node id while the
 * product is formed: x_self[m, :] = feat_self[self_ids[m] - off_self, :] (an id
 * outside the table counts as a row of zeros), so the gathered copy of
 * Graph.get_node_features_from_node (bipartite_graph.py:70-76, generator.py:
 * 270-277) is never written.  self_ids device int32[M].
 */
int gs_sage_project_gather_fwd(const float *feat_self, int64_t n_rows_self, int Fs, int64_t off_self,
                               const int32_t *self_ids, const float *x_mean, int Fn, int64_t M,
                               const float *W_self, const float *W_neigh, const float *bias,
                               int H, int act, float *out, void *stream);

/*
 * Backward of gs_sage_project_fwd.  `out` is the forward result (needed for the
 * relu mask), g_out its gradient.  Any of the g_* outputs may be NULL to skip
 * it.  g_W_self / g_W_neigh / g_bias are overwritten (not accumulated).
 * workspace: device scratch of at least gs_sage_project_bwd_workspace() bytes,
 * used for per-block partial sums that a second kernel adds in a fixed order,
 * so weight gradients are bitwise reproducible.
 */
size_t gs_sage_project_bwd_workspace(int Fs, int Fn, int H, int64_t M);
int gs_sage_project_bwd(const float *x_self, int Fs, const float *x_mean, int Fn, int64_t M,
                        const float *W_self, const float *W_neigh, int H, int act,
                        const float *out, const float *g_out,
                        float *g_W_self, float *g_W_neigh, float *g_bias,
                        float *g_x_self, float *g_x_mean,
                        void *workspace, size_t workspace_bytes, void *stream);

/*
 * The weight and bias gradients of gs_sage_project_bwd for TWO projections of one
 * shape at once -- the two node types of a layer, whose Aggregators have the same
 * widths when both feature tables do (utils.py:72-89) -- with one weight-gradient
 * launch and one reduction launch, so that a training step needs no second
 * stream for them.  Each problem brings its own workspace of
 * gs_sage_project_bwd_workspace(Fs, Fn, H, M) bytes.  out is the forward output
 * (read for the relu mask, may be NULL for a linear layer); g_bias may be NULL.
 * self_ids != NULL: the own rows are not given as a gathered [M, Fs] copy; x_self
 * is then the feature table [self_rows, Fs] itself and row m of the operand is its
 * row self_ids[m] - self_off (an id outside the table counts as a row of zeros,
 * as in gs_sage_project_gather_fwd), so the forward pass of a training step need
 * not write those rows anywhere.  Only for the shapes for which
 * gs_sage_project_wgrad_by_id_supported() != 0 (GS_ELIMIT otherwise).
 */
typedef struct gs_wgrad_problem {
    const float *x_self, *x_mean;   /* [M, Fs], [M, Fn]  */
    const float *out, *g_out;       /* [M, H]            */
    float *g_W_self, *g_W_neigh;    /* [Fs, H/2], [Fn, H/2] */
    float *g_bias;                  /* [H] or NULL       */
    void *workspace;
    size_t workspace_bytes;
    const int32_t *self_ids;        /* [M] or NULL       */
    int64_t self_off, self_rows;
} gs_wgrad_problem;

int gs_sage_project_wgrad_by_id_supported(int Fs, int Fn, int H);
int gs_sage_project_wgrad_pair(const gs_wgrad_problem *a, const gs_wgrad_problem *b,
                               int Fs, int Fn, int64_t M, int H, int act, void *stream);

/*
 * out[m, :] = x[m, :] * rsqrt(max(sum(x[m, :]^2), 1e-12)) -- K.l2_normalize as
 * applied to the final embeddings (models.py:32,186-190) -- and its backward.
 */
int gs_l2_normalize_fwd(const float *x, int64_t M, int H, float *out, void *stream);
int gs_l2_normalize_bwd(const float *x, const float *g_out, int64_t M, int H,
                        float *g_x, void *stream);

/*
 * One whole aggregator layer in a single kernel:
 *   out[p, :] = act(concat(self[p] @ W_self, mean_j(child[p, j]) @ W_neigh) + bias)
 * (optionally followed by the row l2 normalisation), i.e. gs_gather_rows +
 * gs_gather_mean_fwd + gs_sage_project_fwd (+ gs_l2_normalize_fwd) without
 * writing their intermediates: Aggregator.call (utils.py:103-120) applied to
 * rows that are fetched by node id (generator.py:270-277, bipartite_graph.py:
 * 70-76), and models.py:186-190 for the normalisation.
 *
 * The head nodes are given as up to 2*GS_MAX_HOPS segments (HOST array) that all
 * use the same feature tables and weights (the slots of one node type in one
 * layer, models.py:79-90); output rows follow segment order.
 *   head_ids  device node ids of the segment's heads, or NULL: head p is row
 *             self_row0 + p of feat_self (layer >= 1: inputs are activations)
 *   child_ids device node ids [n_heads * n_child] of the sampled neighbors, or
 *             NULL: child j of head p is row child_row0 + p*n_child + j of
 *             feat_neigh
 * save_self / save_mean: optional device [rows, Fs] / [rows, Fn] copies of the
 * gathered own rows and neighbor means (what the backward pass needs).
 * Supported shapes: gs_sage_fused_supported(Fs, Fn, H, max n_child) != 0;
 * otherwise GS_ELIMIT is returned and the three separate entry points apply.
 * Two kernels stand behind this entry point: narrow layers (widths 32/64/128,
 * H/2 of 16 or 32) keep all weights in LDS and give a wavefront 16 heads at a
 * time; wide layers (widths multiples of 32 and at least 64, H/2 a multiple of
 * 128, n_child <= 32) are tiled 128 heads x 128 columns with the gathers on
 * their own wavefronts beside the matrix product.  For wide layers normalize
 * runs as a second small kernel on out.
 */
typedef struct gs_fused_segment {
    const int32_t *head_ids;
    const int32_t *child_ids;
    int64_t n_heads;
    int64_t self_row0;
    int64_t child_row0;
    int32_t n_child;
} gs_fused_segment;

int gs_sage_fused_supported(int Fs, int Fn, int H, int max_child);
int gs_sage_fused_fwd(const float *feat_self, int64_t n_rows_self, int Fs, int64_t off_self,
                      const float *feat_neigh, int64_t n_rows_neigh, int Fn, int64_t off_neigh,
                      const gs_fused_segment *segs, int n_segs,
                      const float *W_self, const float *W_neigh, const float *bias,
                      int H, int act, int normalize,
                      float *out, float *save_self, float *save_mean, void *stream);

/*
 * gs_sage_fused_fwd with a device workspace, for wide layers (utils.py:109-119, the two matrix products).  The
 * fp32 matrix instruction of this part runs on the vector ALU, so in the form without workspace the product and
 * the neighbor sums of one compute unit take turns.  With a workspace every fp32 operand is cut into three bf16
 * pieces that add up to it exactly (8 + 8 + 8 significand bits); the nine piece products are exact in fp32 and
 * are accumulated in fp32 on the matrix cores, beside the vector work.  The result is the fp32 product up to
 * the order of its additions: same tolerance against the float64 restatement as the form without workspace.
 *   gs_sage_fused_workspace_bytes  bytes of workspace this shape uses: 3 * (H/2) * (Fs + Fn) * 2 for wide
 *                                  layers, 0 where the narrow kernel applies (the workspace is then ignored)
 *   workspace                      device memory, 16-byte aligned, at least that many bytes; overwritten by
 *                                  every call (the cut weights), so calls in flight at once need one each.
 *                                  NULL: exactly gs_sage_fused_fwd.
 */
size_t gs_sage_fused_workspace_bytes(int Fs, int Fn, int H, int max_child);
int gs_sage_fused_fwd_ws(const float *feat_self, int64_t n_rows_self, int Fs, int64_t off_self,
                         const float *feat_neigh, int64_t n_rows_neigh, int Fn, int64_t off_neigh,
                         const gs_fused_segment *segs, int n_segs,
                         const float *W_self, const float *W_neigh, const float *bias,
                         int H, int act, int normalize,
                         float *out, float *save_self, float *save_mean,
                         void *workspace, size_t workspace_bytes, void *stream);

/*
 * gs_sage_fused_fwd with the children of some segments DRAWN inside the kernel
 * instead of read from child_ids: the deepest hop of a walk holds most of its
 * samples (1250 of 1525 for fan-outs 25/10/5), and drawing them where their
 * feature rows are gathered saves the sampler's pass over them, the id round trip
 * through memory, and hides the draws' two dependent reads behind the gathers of
 * other wavefronts.  The draws are exactly those of gs_sample (sampler.py:65-90,
 * same CSR reads, same Philox counter (walk, hop, slot) or tape word, same
 * err_flag convention), so gs_sample over hops 0..L-2 followed by this call gives
 * the ids gs_sample over hops 0..L-1 gives.
 *   sampling  the adjacency and keys, as the arguments of the same names of
 *             gs_sample; tape_len = words of replay tape per walk (sum of the hop
 *             sizes), only read when replay != NULL
 *   draws     HOST array of n_segs entries.  hop < 0: the segment reads child_ids
 *             as in gs_sage_fused_fwd.  hop >= 0: head p of the segment is node
 *             head_ids[p] (required), it belongs to walk walk0 + p / heads_per_walk
 *             (an index into walk_ids / an offset from walk_id_base / a row of the
 *             tape) and is slot j = p % heads_per_walk of that walk's hop `hop`-1
 *             (the root for hop 0); its child c is the draw for slot j * n_child + c
 *             of hop `hop`, tape word replay[walk * tape_len + tape_off + slot].
 *             out_child_ids (nullable) receives the drawn ids [n_heads * n_child].
 * Narrow layers only (the wide-layer kernel reads ids): GS_ELIMIT otherwise.
 */
typedef struct gs_fused_sampling {
    const int64_t *indptr;
    const int32_t *indices;
    int64_t n_nodes;
    uint64_t rng_seed;
    uint64_t walk_id_base;
    const int64_t *walk_ids;
    const uint32_t *replay;
    int64_t tape_len;
    int32_t *err_flag;
} gs_fused_sampling;

typedef struct gs_fused_draw {
    int32_t hop;
    int32_t heads_per_walk;
    int32_t tape_off;
    int64_t walk0;
    int32_t *out_child_ids;
} gs_fused_draw;

int gs_sage_fused_sampled_fwd(const float *feat_self, int64_t n_rows_self, int Fs, int64_t off_self,
                              const float *feat_neigh, int64_t n_rows_neigh, int Fn, int64_t off_neigh,
                              const gs_fused_segment *segs, int n_segs,
                              const float *W_self, const float *W_neigh, const float *bias,
                              int H, int act, int normalize,
                              float *out, float *save_self, float *save_mean,
                              const gs_fused_sampling *sampling, const gs_fused_draw *draws, void *stream);

/*
 * The whole two-hop forward pass in one kernel, one wavefront per walk: sampling
 * (as gs_sample), both aggregator layers (as gs_sage_fused_fwd) and the l2
 * normalisation, for the reference's own model depth (its generator and type
 * adjacency list are written for exactly two hops, generator.py:239-244,
 * 339-345; layers per models.py:146-190).  Only the embeddings are written:
 *   emb[w, :]  w < n_pairs: user embedding of pair w; w >= n_pairs: item
 *              embedding of pair w - n_pairs                (device float [2*n_pairs, H2])
 * seeds are the 2*n_pairs roots (user heads, then item heads); users are ids
 * [0, n_users), items [n_users, n_users + n_items) = n_nodes.  Draws, replay
 * tape and err_flag exactly as gs_sample with fanouts {n0, n1}.  out_hop1 /
 * out_hop2 (nullable) receive the sampled ids ([2*n_pairs, n0] and
 * [2*n_pairs, n0*n1]) when the caller wants them.
 * Weight sets are per (layer, head node type), shaped as Aggregator.build makes
 * them (utils.py:72-89); bias may be NULL.
 * gs_sage2_supported() != 0 tells whether this build covers the shape; if not,
 * GS_ELIMIT is returned and gs_sample + gs_sage_fused_fwd (or the unfused
 * entry points) compute the same thing.
 *
 * saved (nullable): for training, the kernel also writes what the backward pass
 * reads, so that the forward half of a step stays one launch.  Index 0 = user,
 * 1 = item.  x_self / x_mean / h1 are tables of n_pairs * (1 + n0) rows per node
 * type: row p belongs to the root of pair p of that type, row n_pairs + p*n0 + j
 * to hop-1 head j of the walk rooted at the OTHER type's root of pair p.
 *   x_self[t]  [rows, F_t]        own feature rows (layer-0 self input)
 *   x_mean[t]  [rows, F_other(t)] neighbor means (layer-0 neighbor input)
 *   h1[t]      [rows, H1]         layer-0 output after relu (layer-1 self input)
 *   x_mean1[t] [n_pairs, H1]      layer-1 neighbor input of the roots
 *   z[t]       [n_pairs, H2]      layer-1 output before the l2 normalisation
 */
typedef struct gs_sage_weights {
    const float *w_self;    /* [F_self,  H/2] */
    const float *w_neigh;   /* [F_neigh, H/2] */
    const float *bias;      /* [H] or NULL    */
} gs_sage_weights;

typedef struct gs_sage2_saved {
    float *x_self[2];       /* may be NULL when self_ids is given */
    float *x_mean[2];
    float *h1[2];
    float *x_mean1[2];
    float *z[2];
    int32_t *self_ids[2];   /* nullable: [rows] the node id of each row of x_self (4 bytes instead of a feature row;
                               gs_sage_project_wgrad_pair reads the rows from the table by these ids) */
    uint32_t *h1_mask[2];   /* nullable, both or none: [n_pairs * n0 * H1/32] words.  The hop-1 heads' rows of h1 (rows
                               n_pairs ..) are read back only for their relu mask (their values enter layer 1 through
                               x_mean1), so with h1_mask they are NOT written; bit c % 32 of word
                               (row - n_pairs) * H1/32 + c / 32 says whether h1[row, c] > 0.  gs_sage2_backward_tail
                               applies these masks (its l0_mask argument) */
    float dropout_rate;     /* 0: none.  In (0, 1): Dropout of every aggregator input (models.py:152,165), addressed
                               like the draws so that it can be restated: the four columns 4*c4 .. 4*c4+3 of row
                               `row` of input `site` of walk w are kept where the four words of Philox4x32-10(counter
                               = (walk id lo, hi, 0x40000000 | site, row * 64 + c4), key = dropout_seed) are >=
                               floor(rate * 2^32), and scaled by 1 / (1 - rate).  Sites: 0 hop-2 rows (row = j * n1 +
                               k), 1 hop-1 rows as heads, 2 hop-1 rows as the root's neighbors, 3 the root's row, 4 the
                               layer-1 inputs (row 0 the root's layer-0 output, row 1 + j hop-1 head j's).  Needs
                               x_self and h1_mask (self_ids is ignored): what is saved are the inputs as they entered
                               the layers, and h1 / h1_mask then say "active and kept" */
    uint64_t dropout_seed;
} gs_sage2_saved;

int gs_sage2_supported(int Fu, int Fi, int H1, int H2, int n0, int n1);
int gs_sage2_forward(const int64_t *indptr, const int32_t *indices, int64_t n_nodes,
                     const float *feat_user, int64_t n_users, int Fu,
                     const float *feat_item, int64_t n_items, int Fi,
                     const int32_t *seeds, int64_t n_pairs, int n0, int n1,
                     uint64_t rng_seed, uint64_t walk_id_base, const int64_t *walk_ids,
                     const uint32_t *replay,
                     const gs_sage_weights *l0_user, const gs_sage_weights *l0_item,
                     const gs_sage_weights *l1_user, const gs_sage_weights *l1_item,
                     int H1, int H2, int normalize,
                     float *emb, int32_t *out_hop1, int32_t *out_hop2,
                     const gs_sage2_saved *saved, int32_t *err_flag, void *stream);

/*
 * The same kernel below the first hop of a THREE-hop model (models.py:127-190 with three layers;
 * sampler.py:65-90 to depth 3): every hop-1 node v of every walk is the root of a two-hop subtree -- v, its
 * n1 sampled neighbors, their n2 sampled neighbors each -- and everything layer 1 of v depends on is
 * private to that subtree.  A wavefront takes eight consecutive subtrees at a time (their hop-2 nodes as one
 * list of heads, so that the chain of dependent reads is paid once per eight), draws hops 2 and 3, gathers
 * each subtree's 1 + n1 + n1*n2 feature rows, and leaves
 *   h1_out[i, :]  = relu(layer 1 at v)     the row layer 2 reads as a neighbor's          [roots, H2]
 *   out_h0[i, :]  = relu(layer 0 at v)     the row layer 1 of the walk's root reads       [roots, H1]
 * so that neither the ids of hops 2 and 3 nor the layer-0 outputs of the hop-2 nodes pass through memory,
 * and the sampler's pass over those hops (most of a walk's draws) is hidden behind the gathers.
 *   hop1_ids   [2*n_pairs * roots_per_walk] the hop-1 nodes as gs_sample writes them (fanouts[0] =
 *              roots_per_walk): those of the user-rooted walks (items) first; root i belongs to seed walk
 *              i / roots_per_walk.  An id of -1 (below an isolated root) gives rows of zeros' images.
 *              May be NULL when sub->seeds is given.
 *   walk_ids   per SEED walk [2*n_pairs] (or NULL: walk_id_base + seed walk), rng_seed, replay: as gs_sample;
 *              the draws are exactly those of gs_sample over three hops (counter hop = 1 and 2, slot =
 *              the node's slot in the walk); the tape holds tape_len words per seed walk, hop 2 from word
 *              tape_off_hop2 on, hop 3 from tape_off_hop3 on
 *   n1, n2     fan-outs of hops 2 and 3;  l0_* / l1_*: the weights of layers 0 and 1
 *   out_hop2 / out_hop3 (nullable) receive the drawn ids in gs_sample's layout
 *   sub->seeds (nullable) [2*n_pairs] the seed nodes, user seeds first.  Then hop 1 is drawn here as well
 *              (counter hop 0, slot = the root's position below its seed, tape words [0, roots_per_walk) of
 *              the seed walk; err_flag as gs_sample for isolated and out-of-range seeds), one group ahead of
 *              the group being gathered, and written to sub->out_hop1 [2*n_pairs * roots_per_walk]: no
 *              sampler launch is left.
 * What remains for the caller are the seed nodes themselves: layer 0 and layer 1 at the seeds (neighbors: the
 * hop-1 nodes' feature rows / out_h0 rows) and layer 2 (neighbors: h1_out rows): gs_sage3_seeds_forward, or
 * gs_sage_fused_fwd per layer.
 * Training (the sv_ tables, all or none; H1 = 32): what gs_sage3_backward_all reads below the first hop is
 * written on the way, indexed by the node type t of the rows (o = the other type), i = the index of a hop-1
 * node among those of its type (the item-typed ones hang below the user seeds):
 *   sv_x_mean0_hop2[t] [n_pairs*roots_per_walk*n1, F_o]  the layer-0 neighbor mean of hop-2 node k below hop-1
 *                      node i (of type o), row i*n1 + k
 *   sv_mask_hop2[t]    [n_pairs*roots_per_walk*n1]       the same rows: bit c says layer-0 output column c > 0
 *                      (the outputs themselves are needed by nothing: layer 1 reads their mean)
 *   sv_x_mean0_hop1[t] [n_pairs*roots_per_walk, F_o], sv_x_mean1_hop1[t] [.., H1]   the layer-0 / layer-1
 *                      neighbor means of hop-1 node i
 * together with out_h0, h1_out (the hop-1 nodes' outputs), out_hop2 (the hop-2 ids) and the hop-1 ids.
 * Shapes: gs_sage2_subtrees_supported(Fu, Fi, H1, H2, n1, n2) != 0 (those of gs_sage2_supported whose eight
 * subtrees' ids fit a wavefront's share of LDS); others get GS_ELIMIT.
 */
typedef struct gs_sage2_subtrees {
    int32_t roots_per_walk;
    int64_t tape_len, tape_off_hop2, tape_off_hop3;
    float *out_h0;
    const int32_t *seeds;
    int32_t *out_hop1;
    float *sv_x_mean0_hop2[2];
    uint32_t *sv_mask_hop2[2];
    float *sv_x_mean0_hop1[2];
    float *sv_x_mean1_hop1[2];
} gs_sage2_subtrees;

int gs_sage2_subtrees_supported(int Fu, int Fi, int H1, int H2, int n1, int n2);
int gs_sage2_forward_subtrees(const int64_t *indptr, const int32_t *indices, int64_t n_nodes,
                              const float *feat_user, int64_t n_users, int Fu,
                              const float *feat_item, int64_t n_items, int Fi,
                              const int32_t *hop1_ids, int64_t n_pairs, int n1, int n2,
                              uint64_t rng_seed, uint64_t walk_id_base, const int64_t *walk_ids,
                              const uint32_t *replay,
                              const gs_sage_weights *l0_user, const gs_sage_weights *l0_item,
                              const gs_sage_weights *l1_user, const gs_sage_weights *l1_item,
                              int H1, int H2,
                              float *h1_out, int32_t *out_hop2, int32_t *out_hop3,
                              const gs_sage2_subtrees *sub, int32_t *err_flag, void *stream);

/*
 * Layers 0, 1 and 2 at the seed nodes of a three-hop model, both node types in one launch, on what
 * gs_sage2_forward_subtrees left (models.py:146-190; l2 normalisation utils.py:109-119 as gs_sage_fused_fwd):
 *   seeds     [2*n_pairs] user seeds, then item seeds;  hop1_ids [2*n_pairs * n0] their sampled neighbors
 *   h0_hop1   [2*n_pairs * n0, H1]  layer-0 outputs of the hop-1 nodes (out_h0 above)
 *   h1_hop1   [2*n_pairs * n0, H2]  layer-1 outputs of the hop-1 nodes (h1_out above)
 *   emb       [2*n_pairs, H3]       act(layer 2), l2-normalised when normalize != 0
 * Layers 0 and 1 go through relu; act (GS_ACT_*) is layer 2's.  Neighbor means are sums over all n0 slots
 * divided by n0 (a missing neighbor, id -1, contributes a zero feature row in layer 0 and its stored rows in
 * layers 1 and 2, as in the layered entry points).  A wavefront takes four seeds through all three layers.
 * Shapes: gs_sage3_seeds_supported() != 0 (the widths of gs_sage2_supported, H3 = 32 or 64).
 */
int gs_sage3_seeds_supported(int Fu, int Fi, int H1, int H2, int H3, int n0);
int gs_sage3_seeds_forward(const float *feat_user, int64_t n_users, int Fu,
                           const float *feat_item, int64_t n_items, int Fi,
                           const int32_t *seeds, int64_t n_pairs, const int32_t *hop1_ids, int n0,
                           const float *h0_hop1, const float *h1_hop1,
                           const gs_sage_weights *l0_user, const gs_sage_weights *l0_item,
                           const gs_sage_weights *l1_user, const gs_sage_weights *l1_item,
                           const gs_sage_weights *l2_user, const gs_sage_weights *l2_item,
                           int H1, int H2, int H3, int act, int normalize, float *emb, void *stream);

/*
 * The same, leaving what the backward pass reads at the seeds (gs_sage3_backward_all): per seed type t, one row per
 * seed -- x_mean0 [n_pairs, F_o], x_mean1 [n_pairs, H1], x_mean2 [n_pairs, H2] the neighbor means the three layers
 * read; h0 [n_pairs, H1], h1 [n_pairs, H2] the outputs of layers 0 and 1; z [n_pairs, H3] layer 2's output after its
 * activation, before the normalisation.  All twelve tables are required, 16-byte aligned.
 */
typedef struct gs_sage3_seeds_saved {
    float *x_mean0[2], *h0[2], *x_mean1[2], *h1[2], *x_mean2[2], *z[2];
} gs_sage3_seeds_saved;

int gs_sage3_seeds_forward_saving(const float *feat_user, int64_t n_users, int Fu,
                                  const float *feat_item, int64_t n_items, int Fi,
                                  const int32_t *seeds, int64_t n_pairs, const int32_t *hop1_ids, int n0,
                                  const float *h0_hop1, const float *h1_hop1,
                                  const gs_sage_weights *l0_user, const gs_sage_weights *l0_item,
                                  const gs_sage_weights *l1_user, const gs_sage_weights *l1_item,
                                  const gs_sage_weights *l2_user, const gs_sage_weights *l2_item,
                                  int H1, int H2, int H3, int act, int normalize, float *emb,
                                  const gs_sage3_seeds_saved *saved, void *stream);

/*
 * The layer-0 relu masks for gs_sage2_backward_tail / gs_sage2_backward_tail_head (their nullable l0_mask argument), in which case g_h1 leaves as the
 * gradient at the layer-0 PRE-activations (g_h1 * [h1 > 0]) and the layer-0 weight gradients take it with
 * act = linear, without reading the forward output again:
 *   h1[t]       [>= n_pairs, H1]  rows of the roots: the saved layer-0 outputs themselves
 *   h1_mask[t]  the mask words gs_sage2_forward wrote for the hop-1 heads (gs_sage2_saved.h1_mask)
 */
typedef struct gs_sage2_l0_mask {
    const float *h1[2];
    const uint32_t *h1_mask[2];
    float input_scale;      /* 0 or 1: none.  Otherwise the factor the forward pass applied to the layer-1 inputs it kept
                               (dropout: 1 / (1 - rate), gs_sage2_saved.dropout_rate); the gradients that pass the
                               masks are multiplied by it */
} gs_sage2_l0_mask;

/*
 * Backward of the top of that model, one launch for both node types: from the
 * gradients of the embeddings to the gradients of the layer-0 activations, in
 * the row layout of the gs_sage2_saved tables; index 0 / "_user" = user,
 * 1 / "_item" = item, o = the other type:
 *   g_z[t][p, :]             = gradient at z[t][p, :] of the l2 normalisation
 *                              (models.py:32; copied through when normalize == 0)
 *   g_h1[t][p, :]            = act'(z) * g_z[t][p, :H2/2] . w1_self[t]^T
 *                              (utils.py:109-119: the root's own layer-1 input)
 *   g_h1[o][n_pairs + p*n0 + j, :] = act'(z) * g_z[t][p, H2/2:] . w1_neigh[t]^T / n0,  j < n0
 *                              (utils.py:103-109: the mean over the root's n0 hop-1 heads)
 * Both g_h1 tables ([n_pairs * (1 + n0), H1]) are written completely.  It is
 * gs_l2_normalize_bwd + the input gradients of gs_sage_project_bwd +
 * gs_gather_mean_bwd of the layer-1 backward in one kernel; the layer-1 weight
 * gradients follow from g_z through gs_sage_project_bwd as usual.
 * w1_* are the layer-1 weights [H1, H2/2]; act is the layer-1 activation (z is
 * its output).  H1 and H2 are 32 or 64 (the widths of gs_sage2_forward), else
 * GS_ELIMIT.
 */
int gs_sage2_backward_tail(const float *z_user, const float *z_item,
                           const float *g_emb_user, const float *g_emb_item,
                           int64_t n_pairs, int n0, int H1, int H2, int normalize, int act,
                           const float *w1_self_user, const float *w1_neigh_user,
                           const float *w1_self_item, const float *w1_neigh_item,
                           float *g_z_user, float *g_z_item, float *g_h1_user, float *g_h1_item,
                           const gs_sage2_l0_mask *l0_mask, void *stream);

/*
 * gs_sage2_backward_tail with the link head and its loss evaluated in the same
 * launch (link_utils.py:167-168,244-247; MSE of train_movielens.py:186): instead
 * of reading g_emb it takes the embeddings emb_user / emb_item [n_pairs, H2] (the
 * output of gs_sage2_forward), the head's kernel [2 H2] and bias [1] and targets
 * [n_pairs], writes pred [n_pairs] and uses the head's own gradient at the
 * embeddings.  The head's parameter gradients and the loss leave the launch as
 * per-block partial sums in head_slab (gs_sage2_tail_head_workspace bytes,
 * gs_sage2_tail_head_rows rows of 2 H2 + 2 floats: kernel gradient, bias
 * gradient, sum of squared errors); gs_link_head_reduce adds them in row order
 * into g_kernel [2 H2], g_bias [1] and loss [1] (= sum / B) -- bitwise
 * reproducible, and off the critical path of the rest of the backward pass.
 */
size_t gs_sage2_tail_head_workspace(int64_t n_pairs, int H2);
int gs_sage2_tail_head_rows(int64_t n_pairs);
int gs_sage2_backward_tail_head(const float *z_user, const float *z_item,
                                const float *emb_user, const float *emb_item, const float *targets,
                                const float *head_kernel, const float *head_bias,
                                int64_t n_pairs, int n0, int H1, int H2, int normalize, int act,
                                const float *w1_self_user, const float *w1_neigh_user,
                                const float *w1_self_item, const float *w1_neigh_item,
                                float *g_z_user, float *g_z_item, float *g_h1_user, float *g_h1_item,
                                float *pred, void *head_slab, size_t head_slab_bytes, const gs_sage2_l0_mask *l0_mask, void *stream);
int gs_link_head_reduce(const void *slab, int n_rows, int E, int64_t B, float *g_kernel, float *g_bias,
                        float *loss, void *stream);

/*
 * The WHOLE backward pass of the two-hop model on the tables gs_sage2_forward saved, in one launch plus one
 * reduction (train_movielens.py:183-189: loss, gradients of every trainable weight): what
 * gs_sage2_backward_tail_head computes (head, loss, l2 normalisation and layer 1 backward, models.py:32,
 * utils.py:103-119), and with it the parameter gradients of the head, of layer 1 and of layer 0 for both node
 * types -- the work of gs_link_head_reduce and of two gs_sage_project_wgrad_pair calls.  Neither g_z nor
 * g_h1 passes through memory: the gradient rows of a walk's n0 hop-1 heads are one vector under n0 relu masks,
 * so the layer-0 weight gradients need only the saved input rows and one mask word per row.
 *   z, emb, h1, x_mean1, x_mean, h1_mask: the gs_sage2_saved tables / the forward's output, per node type
 *   x_self, self_ids, self_off, self_rows: the own rows of layer 0 [n_pairs * (1 + n0), F], or -- with
 *       self_ids [n_pairs * (1 + n0)] -- the feature TABLE of the type (self_rows rows, the first holding node
 *       self_off), read by id (ids outside the table count as rows of zeros), for both types or for none
 *   h1_mask NULL (both): layer 0 has NO activation.  Otherwise layer 0 is relu, the hop-1 heads' masks are the
 *       words, the roots' masks are read off h1 (> 0)
 *   targets != NULL: the link head (head_kernel [2 H2], head_bias [1]) and the mean-squared-error loss are
 *       evaluated here (pred [n_pairs], g_head_kernel [2 H2], g_head_bias [1], loss [1] are written; emb is
 *       read); targets == NULL: the embeddings' gradients are g_emb [n_pairs, H2] and the head fields are ignored
 *   input_scale: 0 or 1 none; else dropout's 1 / (1 - rate) as in gs_sage2_l0_mask
 *   g_w0_self / g_w0_neigh [F, H1/2], g_b0 [H1] (nullable), g_w1_self / g_w1_neigh [H1, H2/2], g_b1 [H2]
 *       (nullable): overwritten
 * act is layer 1's activation, normalize as in gs_sage2_forward.  Partial sums leave the first launch as
 * one row per block in `workspace` (gs_sage2_backward_all_workspace bytes) and are added in row order:
 * bitwise reproducible.  The workspace also holds the roots' layer-0 gradient rows between the kernel's two
 * passes.  Shapes: gs_sage2_backward_all_supported() != 0 (both feature widths F = 32 or 64, H1 = H2 = 32: a
 * wavefront keeps the layer-0 sums of one node type in registers, the layer-1 sums in LDS), n0 <= 64 (a lane per
 * hop-1 head) and n_pairs * (1 + n0) * F < 2^31 (32-bit offsets into the saved tables); else GS_ELIMIT.
 */
/*
 * adam (nullable argument of gs_sage2_backward_all): the optimizer step of gs_adam_step, applied by the
 * reduction to every gradient element it has just written -- one launch less when no collective stands
 * between the gradients and the update (train_movielens.py:183, Adam).  Every g_* pointer of the call must
 * lie inside grads[0 .. n); params, m, v are flat buffers indexed like grads; the rate comes from
 * lr_t_device[0] when that is not NULL (a captured graph), else from lr_t; elements of the flat buffers
 * that no g_* pointer covers are left alone.
 */
typedef struct gs_adam_update {
    const float *grads;
    float *params, *m, *v;
    int64_t n;
    const float *lr_t_device;
    float lr_t, beta_1, beta_2, epsilon, grad_scale;
} gs_adam_update;

typedef struct gs_sage2_bwd_all {
    const float *z[2], *emb[2], *g_emb[2];
    const float *targets, *head_kernel, *head_bias;
    const float *w1_self[2], *w1_neigh[2];
    const float *h1[2], *x_mean1[2];
    const uint32_t *h1_mask[2];
    const float *x_self[2], *x_mean[2];
    const int32_t *self_ids[2];
    int64_t self_off[2], self_rows[2];
    float input_scale;
    float *pred;
    float *g_w0_self[2], *g_w0_neigh[2], *g_b0[2];
    float *g_w1_self[2], *g_w1_neigh[2], *g_b1[2];
    float *g_head_kernel, *g_head_bias, *loss;
} gs_sage2_bwd_all;

/*
 * The same for THREE hops with 32 units per layer (models.py:146-190 with three layers; train_movielens.py:183-189), on
 * the tables the layered forward pass saved: a kernel over the seeds (head and loss or the given embedding gradients,
 * l2 normalisation, layers 2 and 1 at the seed, the seed's layer-0 row), a kernel over the hop-1 nodes (layer 1 at the
 * node, its own layer-0 row, and the layer-0 rows of its n1 children, whose gradient is one row under n1 relu
 * masks), one reduction (optionally with the Adam step, as above).  No gradient of an activation is written
 * except two 128-byte rows per seed.  Tables per node type t, rows in this order (o = the other type):
 *   layer-0 tables (h0 [.., 32], x_mean0 [.., F], row_ids [..]):  [0, B) the seeds of t; [B, B + B n0) the hop-1
 *       nodes of t, row B + p n0 + j below seed p of o; then the hop-2 nodes of t, row B (1 + n0) + (p n0 + j) n1 + k
 *       below seed p of t
 *   layer-1 tables (h1, x_mean1 [.., 32]):  [0, B) seeds, [B, B + B n0) hop-1 nodes;   x_mean2, z, emb [B, 32]: seeds
 *   h0 / h1: the layers' outputs after relu;  x_mean*: the neighbor means the layers read;  row_ids: the node id of
 *   every layer-0 row (own rows are read from feat, feat_rows rows whose first holds node feat_off; ids outside count
 *   as rows of zeros)
 * The rows below the seeds may live apart from the seeds' (what gs_sage2_forward_subtrees and
 * gs_sage3_seeds_forward_saving leave): hop1 != NULL gives the hop-1 rows of every table -- h0, h1, x_mean1 [B n0, 32],
 * x_mean0 [B n0, F], ids [B n0], all required -- and hop2 != NULL the hop-2 rows -- x_mean0 [B n0 n1, F], ids, and
 * h0 [.., 32] or, instead, mask [..] words whose bit c says h0[row, c] > 0; the whole tables then need only the rows
 * above.
 * act is layer 2's activation; weights w1_*, w2_* [32, 16]; gradients g_w0_* [F, 16], g_w1_*, g_w2_* [32, 16], g_b*
 * [32] (nullable).  Shapes: gs_sage3_backward_all_supported() != 0 (F = 32 or 64 for both node types, 32 units in
 * all three layers, n1 <= 32) and n_pairs (1 + n0 + n0 n1) F < 2^31; else GS_ELIMIT.
 */
typedef struct gs_sage3_bwd_hop1 {
    const float *h0[2], *h1[2], *x_mean0[2], *x_mean1[2];
    const int32_t *ids[2];
} gs_sage3_bwd_hop1;

typedef struct gs_sage3_bwd_hop2 {
    const float *h0[2];         /* or NULL with mask */
    const uint32_t *mask[2];
    const float *x_mean0[2];
    const int32_t *ids[2];
} gs_sage3_bwd_hop2;

typedef struct gs_sage3_bwd {
    const float *z[2], *emb[2], *g_emb[2];
    const float *targets, *head_kernel, *head_bias;
    const float *w2_self[2], *w2_neigh[2], *w1_self[2], *w1_neigh[2];
    const float *h0[2], *h1[2], *x_mean0[2], *x_mean1[2], *x_mean2[2];
    const float *feat[2];
    int64_t feat_off[2], feat_rows[2];
    const int32_t *row_ids[2];
    float *pred;
    float *g_w0_self[2], *g_w0_neigh[2], *g_b0[2];
    float *g_w1_self[2], *g_w1_neigh[2], *g_b1[2];
    float *g_w2_self[2], *g_w2_neigh[2], *g_b2[2];
    float *g_head_kernel, *g_head_bias, *loss;
    const gs_sage3_bwd_hop1 *hop1;          /* nullable */
    const gs_sage3_bwd_hop2 *hop2;          /* nullable: required with hop1 */
} gs_sage3_bwd;


int gs_sage3_backward_all_supported(int Fu, int Fi, int H1, int H2, int H3, int n0, int n1);
size_t gs_sage3_backward_all_workspace(int64_t n_pairs, int n0, int F);
int gs_sage3_backward_all(const gs_sage3_bwd *p, int64_t n_pairs, int n0, int n1, int F, int normalize, int act,
                          void *workspace, size_t workspace_bytes, const gs_adam_update *adam, void *stream);

int gs_sage2_backward_all_supported(int Fu, int Fi, int H1, int H2);
size_t gs_sage2_backward_all_workspace(int64_t n_pairs, int F, int H1, int H2);
int gs_sage2_backward_all(const gs_sage2_bwd_all *p, int64_t n_pairs, int n0, int F, int H1, int H2,
                          int normalize, int act, void *workspace, size_t workspace_bytes,
                          const gs_adam_update *adam, void *stream);

/*
 * Link head of the reference's training script: concat(user_emb, item_emb) ->
 * Dense(1, sigmoid) (link_utils.py:167-168,244-247 as used at
 * train_movielens.py:179), kernel device float[2E], bias device float[1].
 *   pred[b] = sigmoid(<concat(u[b], i[b]), kernel> + bias)          device float[B]
 * gs_link_head_mse_fwd_bwd additionally evaluates the mean-squared-error loss
 * of train_movielens.py:186 against targets[B] (loss: device float[1]) and all
 * its gradients in the same pass: g_emb_user / g_emb_item [B, E], g_kernel [2E],
 * g_bias [1] (overwritten).  workspace: device scratch of at least
 * gs_link_head_workspace(B, E) bytes (per-block partial sums, added in block
 * order).  E <= 256.
 */
size_t gs_link_head_workspace(int64_t B, int E);
int gs_link_head_fwd(const float *emb_user, const float *emb_item, int64_t B, int E,
                     const float *kernel, const float *bias, float *pred, void *stream);
int gs_link_head_mse_fwd_bwd(const float *emb_user, const float *emb_item, int64_t B, int E,
                             const float *kernel, const float *bias, const float *targets,
                             float *pred, float *loss, float *g_emb_user, float *g_emb_item,
                             float *g_kernel, float *g_bias,
                             void *workspace, size_t workspace_bytes, void *stream);

/*
 * gs_link_head_mse_fwd_bwd for callers that cannot allocate device memory (code
 * under HIP graph capture, tight serving or training loops, harnesses that
 * forbid hipMalloc).  No counterpart in the reference.  Same arguments minus the
 * workspace: the per-block partial sums go to a static device array inside the
 * library (526 336 bytes, held by every GPU the library runs on until the process
 * ends), whose address on the current device is looked up on the first call on
 * that device.  The call allocates and frees nothing.  Limit: E <= 256 (the
 * width the array is sized for; any B), else GS_ELIMIT.  All calls on one GPU
 * share that GPU's array: calls that overlap on one GPU (two streams, two
 * threads) race; calls ordered on one stream do not.  Results are bitwise those
 * of gs_link_head_mse_fwd_bwd.
 */
int gs_link_head_mse_fwd_bwd_static(const float *emb_user, const float *emb_item, int64_t B, int E,
                                    const float *kernel, const float *bias, const float *targets,
                                    float *pred, float *loss, float *g_emb_user, float *g_emb_item,
                                    float *g_kernel, float *g_bias, void *stream);

/*
 * Adam update on flat storage, the optimizer of train_movielens.py:185 in the
 * TF2/Keras formulation: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 * p -= lr_t * m / (sqrt(v) + eps), with the caller passing the bias-corrected
 * rate lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) -- by value, or, when
 * lr_t_device != NULL, as a device float read at run time (so a captured HIP
 * graph of the step can be replayed while the rate changes).  Every gradient
 * is multiplied by grad_scale as it is read (grads itself is left as it is): a
 * data-parallel job sums its ranks' gradients with one all-reduce and passes
 * 1 / world here instead of dividing in a pass of its own; 1.0f otherwise.
 */
int gs_adam_step(float *params, const float *grads, float *m, float *v, int64_t n,
                 float lr_t, const float *lr_t_device, float beta1, float beta2, float eps,
                 float grad_scale, void *stream);

/*
 * Hands one batch to a step whose launches are replayed from a captured graph:
 * the batch's roots, walk ids and targets (what LinkSequence.__getitem__ slices
 * out of the epoch, generator.py:84-98) are copied into the fixed buffers the
 * graph reads, and the step's Adam rate into lr_t_dst (nullable), in one launch
 * instead of four small copies.  A group with count 0 is skipped.
 */
int gs_stage_batch(const int32_t *seeds, int32_t *seeds_dst, int64_t n_seeds,
                   const int64_t *walk_ids, int64_t *walk_ids_dst, int64_t n_walk_ids,
                   const float *targets, float *targets_dst, int64_t n_targets,
                   float lr_t, float *lr_t_dst, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GSAGE_H_ */
