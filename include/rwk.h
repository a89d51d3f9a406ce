/*
 * librwk -- MI355X (gfx950) kernels for the random-walk-kernel layer of RW_NN.
 *
 * C ABI only: plain pointers and sizes, no torch types.  Every pointer argument is a DEVICE
 * pointer unless its comment says "host".  Every function enqueues work on `stream`
 * (a hipStream_t passed as void*; never the null stream implicitly) and returns immediately;
 * the return value is RWK_OK or a negative RWK_E_* code (bad arguments are rejected on the
 * host before anything is launched) or a positive hipError_t from the launch.
 * The library never allocates or frees device memory and keeps no mutable global state:
 * the caller owns inputs, outputs and workspaces.
 *
 * The reference (Xiang227/GNN) has no native interface; each entry point below replaces the
 * PyTorch operators of the cited lines of rwgnn2/model.py (identical to rwgnn/model.py).
 *
 * Index types inside the kernels are int32: N, nnz, G must be < 2^31.
 * Layouts (all row-major, float32 unless noted):
 *   X      (L, N, D)   X[0] = sigmoid(fc(features)), X[i] = A X[i-1]
 *   Z      (L, H, S, D) Z[0] = features_hidden,      Z[i] = Ah Z[i-1]
 *   Ah     (H, S, S)   symmetric hidden adjacency
 *   K      (G, L*H)    column i*H + h                               (model.py:67)
 *   CSR    crow int32 (N+1), col int32 (nnz), val float32 (nnz)
 *   nodes are visited in "sorted position" order p = 0..N-1: node(p) = perm ? perm[p] : p,
 *   gid[p] = graph id of node(p), non-decreasing in p; seg_ptr[g] = first p of graph g.
 */
#ifndef RWK_H_
#define RWK_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RWK_VERSION 100

#define RWK_OK 0
#define RWK_E_NULL (-1)        /* a required pointer is null */
#define RWK_E_SIZE (-2)        /* a size is negative, zero where it must not be, or too large */
#define RWK_E_UNSUPPORTED (-3) /* hyper-parameters outside what the kernels are built for */
#define RWK_E_ALIGN (-4)       /* a pointer is not aligned as documented */
#define RWK_E_STATIC_SIZE (-5) /* the input needs more of the library's fixed device scratch than it holds */
#define RWK_E_DEVICE (-6)      /* the stream's device is not the current device */

typedef void *rwk_stream_t;

int rwk_version(void);
const char *rwk_error_string(int code);

/* ---------------------------------------------------------------------------------------
 * Graph structure (integer work; results must be bit-exact).
 * ------------------------------------------------------------------------------------- */

/* Inspect a COO edge list (what utils.py:38-44 hands to the model as `adj`).
 * flags (int32[4], device) receives:
 *   [0] = number of positions e>0 where (row,col)[e] <= (row,col)[e-1]  (0 <=> strictly
 *         row-major sorted, hence duplicate-free: the layout generate_batches produces)
 *   [1] = number of entries with row or col outside [0, n)
 * flags must be zeroed by the caller. */
int rwk_coo_inspect(const int64_t *rows, const int64_t *cols, int64_t nnz, int64_t n,
                    int32_t *flags, rwk_stream_t stream);

/* CSR of a strictly row-major sorted COO list; replaces the COO->CSR conversion inside
 * torch.spmm (model.py:55).  crow has n+1 entries. */
int rwk_csr_from_sorted_coo(const int64_t *rows, const int64_t *cols, const float *vals,
                            int64_t nnz, int64_t n, int32_t *crow, int32_t *col, float *val,
                            rwk_stream_t stream);

/* flag (int32[1], zeroed by caller) += number of stored entries (r,c,v) for which (c,r) is
 * missing or holds a different value: 0 <=> A == A^T bit for bit, so A^T x may reuse A's CSR. */
int rwk_csr_count_asymmetric(const int32_t *crow, const int32_t *col, const float *val,
                             int64_t n, int32_t *flag, rwk_stream_t stream);

/* Inspect graph_indicator (model.py:32).  flags (int32[4], zeroed by caller):
 *   [0] = number of positions p>0 with gi[p] < gi[p-1]  (0 <=> already sorted)
 *   [1] = max id, [2] = number of negative ids,
 *   [3] = number of positions with gi[p] > gi[p-1] + 1, plus one if gi[0] != 0: for sorted ids, 0 <=> the ids
 *         cover 0..max without gaps (torch.unique's counts then have no empty graph). */
int rwk_indicator_inspect(const int64_t *gi, int64_t n, int32_t *flags, rwk_stream_t stream);

/* From graph ids in sorted-position order (int64, non-decreasing, values in [0,G)):
 * gid32[p] = (int32) ids[p]; seg_ptr[g] = first p with ids[p] >= g (g = 0..G);
 * counts[g] = seg_ptr[g+1]-seg_ptr[g] as int64 (the `counts` of torch.unique, model.py:32);
 * counts_f32[g] = (float)counts[g], the divisor of model.py:63-64.  empty (int32[1], zeroed) += number of
 * graphs with no node (the reference would raise on such input). */
int rwk_segments_build(const int64_t *ids_sorted, int64_t n, int64_t n_graphs, int32_t *gid32,
                       int32_t *seg_ptr, int64_t *counts, float *counts_f32, int32_t *empty,
                       rwk_stream_t stream);

/* Batch assembly on the device (replaces the scipy LIL block-diagonal assembly of
 * generate_batches, utils.py:47-85).  The data set is one block-diagonal CSR over all its
 * nodes (ds_row_ptr int64 (N_all+1), ds_col int32 data-set node ids, ds_val).  For the n nodes
 * of a batch: src_node[v] = data-set node, col_delta[v] = (first batch node of v's graph) -
 * (first data-set node of v's graph), out_ptr[v] = exclusive prefix sum of the row lengths.
 * Writes the batch's COO list (rows, cols int64, vals float32), row-major sorted, bit-identical
 * to the reference's lil_matrix -> tocoo() result. */
int rwk_batch_gather_coo(const int64_t *ds_row_ptr, const int32_t *ds_col, const float *ds_val,
                         const int64_t *src_node, const int64_t *col_delta, const int64_t *out_ptr,
                         int64_t n, int64_t *rows, int64_t *cols, float *vals, rwk_stream_t stream);

/* A batch and its structure in ONE launch (generate_batches' loop body utils.py:60-83 together with what
 * torch.unique, model.py:32, and torch.spmm's COO->CSR conversion, model.py:55, derive from the batch on every
 * call).  The data set is the block-diagonal CSR of rwk_batch_gather_coo with features ds_feat (N_all, n_feat)
 * and labels ds_y.  table is int64 (4, b+1) on the device, computed by the host from its per-graph sizes:
 *   row 0: first batch node of batch graph j (entry b = n), row 1: first data-set node of that graph,
 *   row 2: first stored entry of the graph in the batch (entry b = nnz), row 3: the graph's id in the data set.
 * Outputs (each may be NULL; the three COO arrays only together): the batch's COO list as
 * rwk_batch_gather_coo writes it, feat (n, n_feat), gi = graph_indicator (n), y (b); the int32 CSR
 * (crow n+1, col, val), col16 as rwk_csr_compact writes it (ask for it only when every graph of the batch has
 * at most 32512 nodes), and gid, seg_ptr (b+1), counts, counts_f32 as rwk_segments_build writes them. */
int rwk_batch_build(const int64_t *ds_row_ptr, const int32_t *ds_col, const float *ds_val, const float *ds_feat,
                    int64_t n_feat, const int64_t *ds_y, const int64_t *table, int64_t b, int64_t n, int64_t nnz,
                    int64_t *coo_rows, int64_t *coo_cols, float *coo_vals, float *feat, int64_t *gi, int64_t *y,
                    int32_t *crow, int32_t *col, float *val, int16_t *col16, int32_t *gid, int32_t *seg_ptr,
                    int64_t *counts, float *counts_f32, rwk_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * (1) CSR SpMM   y = A x (+ y)          replaces torch.spmm, model.py:55, and its backward
 * ------------------------------------------------------------------------------------- */

/* x, y: (n, d) row-major, d <= 64.  accumulate != 0: y += A x.
 * The caller picks the schedule from the degree histogram:
 * row_lanes = 0: tile kernel for sparse rows (mean degree up to a few tens).  A workgroup owns
 * 256/ceil(d/4) consecutive rows; their stored entries are read entry-parallel (coalesced
 * columns and values, all x-row gathers of a pass in flight together) into LDS and then added
 * row by row in stored order; rows with more than 64 entries (hubs) are summed by the whole
 * workgroup in a fixed tree order.  long_rows / n_long / long_threshold are ignored.
 * row_lanes (power of two, 1..64): lanes that share the stored entries of one row in the
 * packed pass (dense rows).
 * When long_rows != NULL and n_long > 0, rows with more than `long_threshold` entries are
 * skipped by the packed pass and must be listed in long_rows (n_long of them, ascending):
 * each is then summed by a whole wavefront.  Summation order is fixed in both passes. */
int rwk_spmm_csr(const int32_t *crow, const int32_t *col, const float *val, int64_t n, int64_t d,
                 const float *x, float *y, int accumulate, int32_t row_lanes,
                 const int32_t *long_rows, int64_t n_long, int32_t long_threshold,
                 rwk_stream_t stream);

/* Compact structure for the tile kernel (same product, same bits; model.py:55 on the block-diagonal,
 * unweighted batches that rwgnn2/utils.py:47-85 builds).  rwk_csr_compact writes col16[e] = col[e] minus the
 * first row of the 256-row block that e's row lies in, as int16 (nnz of them), and adds to flags[0] the number
 * of entries whose distance does not fit 16 bits and to flags[1] the number of stored values other than 1.0f
 * (flags: int32[2], zeroed by the caller).  When flags[0] == 0, rwk_spmm_csr_compact computes y = A x (+ y) with
 * the tile kernel from (crow, col16); val == NULL declares every stored value to be 1.0f (allowed when
 * flags[1] == 0): 2 bytes per stored entry are read instead of 8. */
int rwk_csr_compact(const int32_t *crow, const int32_t *col, const float *val, int64_t n, int16_t *col16,
                    int32_t *flags, rwk_stream_t stream);
int rwk_spmm_csr_compact(const int32_t *crow, const int16_t *col16, const float *val, int64_t n, int64_t d,
                         const float *x, float *y, int accumulate, rwk_stream_t stream);

/* The tile SpMM (on int32 columns `col`, or on 16-bit distances `col16` when that is not NULL; val == NULL:
 * all values 1.0f) and the first level of an independent row reduction (red_in (red_p, red_v), red_p > 32,
 * -> red_out32 (32, red_v), exactly rwk_reduce_rows_to32) in ONE launch.  The backward pass has such a pair:
 * dx[i-1] += A^T dx[i] and the reduction of the dZ partials, both fed by rwk_fused_backward. */
int rwk_spmm_tile_reduce_rows(const int32_t *crow, const int32_t *col, const int16_t *col16, const float *val,
                              int64_t n, int64_t d, const float *x, float *y, int accumulate,
                              const float *red_in, int64_t red_p, int64_t red_v, float *red_out32,
                              rwk_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * (2) hidden graphs   Ah = sym(triu(relu(adj_hidden))),  Z[i] = Ah Z[i-1]
 *     replaces model.py:39-42 and :56 for all steps in one launch.
 * ------------------------------------------------------------------------------------- */

/* adj_hidden (H, S(S-1)/2), features_hidden (H,S,D) -> ah (H,S,S), z (L,H,S,D).
 * rng_state / rng_used: optional uint64[2] {seed, offset}.  When both are non-null the
 * launch copies *rng_state to *rng_used and then increments rng_state[1], entirely on the
 * device, so that a captured graph draws a fresh dropout mask on every replay. */
int rwk_hidden_forward(const float *adj_hidden, const float *features_hidden, int64_t h, int64_t s,
                       int64_t d, int64_t max_step, float *ah, float *z, uint64_t *rng_state,
                       uint64_t *rng_used, rwk_stream_t stream);

/* Backward of the chain.  dz_partials (n_partials, L, H, S, D): per-workgroup partial sums of
 * the direct gradient of each Z[i] written by rwk_fused_backward; they are added in index
 * order (deterministic).  Outputs d_adj_hidden (H, S(S-1)/2), d_features_hidden (H,S,D). */
int rwk_hidden_backward(const float *adj_hidden, const float *ah, const float *z,
                        const float *dz_partials, int64_t n_partials, int64_t h, int64_t s,
                        int64_t d, int64_t max_step, float *d_adj_hidden,
                        float *d_features_hidden, rwk_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * (3)+(4) fused random-walk kernel   replaces model.py:45, :53, :57-64, :67 and backward
 *
 *   K[g, i*H+h] = c_g * sum_{p in g} sum_s m[i,h,s,n] * (Z0[h,s].X0[n]) * (Zi[h,s].Xi[n])
 *
 * The (H*S) x N products never leave registers.  Dropout (model.py:58) is a counter-based
 * Philox4x32-7 mask regenerated identically in backward from rng (uint64[2] {seed,offset},
 * device memory; NULL or drop_p == 0 means no dropout).
 *
 * Work split: "streams" of `stream_len` consecutive sorted positions; stream c covers
 * p in [c*stream_len, (c+1)*stream_len).  Partial sums of stream c for graph g go to slot
 * c + g of `k_partials` ((n_streams + G) x L*H floats); rwk_fused_reduce adds the slots of
 * each graph in increasing c and applies c_g (1/(1-p), and 1/|g| when normalize).
 * ------------------------------------------------------------------------------------- */

/* Workspace sizes in floats (pure functions, host side). */
int64_t rwk_fused_n_streams(int64_t n, int64_t stream_len);
int64_t rwk_fused_k_partials_size(int64_t n, int64_t stream_len, int64_t n_graphs, int64_t h,
                                  int64_t max_step);
/* Number of (L,H,S,D) partial slabs rwk_fused_backward writes (= its workgroup count);
 * 0 when the hyper-parameters are unsupported. */
int64_t rwk_fused_dz_partials_count(int64_t n, int64_t stream_len, int64_t d, int64_t h, int64_t s,
                                    int64_t max_step);

int rwk_fused_forward(const float *x, const float *z, const int32_t *gid, const int32_t *perm,
                      int64_t n, int64_t d, int64_t h, int64_t s, int64_t max_step,
                      int64_t stream_len, float drop_p, const uint64_t *rng, float *k_partials,
                      rwk_stream_t stream);

/* Dropout keep bits handed from the forward to the backward pass (rwgnn2/model.py:64 applies one mask in the
 * forward; autograd replays it in the backward).  Without the hand-over the backward runs the counter-based generator
 * again (same bits, no memory); on the matrix-core path (hidden_dim 8..64) the generator costs as much vector-ALU
 * time in the backward as in the forward, and vector-ALU time is not hidden behind fp32 MFMAs.
 * rwk_fused_keep_words: 64-bit words of the buffer when the hand-over is built for these sizes, else 0 (one 32-bit
 * word per (node, hidden graph) when s * max_step <= 32, i.e. (n * h + 1) / 2, else one 64-bit word, n * h).
 * rwk_fused_forward_bits / rwk_fused_backward_bits: as rwk_fused_forward / rwk_fused_backward with that buffer
 * (keep_out / keep_in; NULL = the plain entry point); RWK_E_UNSUPPORTED when rwk_fused_keep_words is 0 and a buffer
 * is passed.  Same results bit for bit. */
int64_t rwk_fused_keep_words(int64_t n, int64_t d, int64_t h, int64_t s, int64_t max_step);
int rwk_fused_forward_bits(const float *x, const float *z, const int32_t *gid, const int32_t *perm,
                           int64_t n, int64_t d, int64_t h, int64_t s, int64_t max_step,
                           int64_t stream_len, float drop_p, const uint64_t *rng, float *k_partials,
                           uint64_t *keep_out, rwk_stream_t stream);
int rwk_fused_backward_bits(const float *x, const float *z, const int32_t *gid, const int32_t *perm,
                            const float *d_k, const float *counts_f32, int64_t n, int64_t d, int64_t h,
                            int64_t s, int64_t max_step, int64_t stream_len, float drop_p,
                            int normalize, const uint64_t *rng, float *dx, float *dz_partials,
                            float *workspace, const uint64_t *keep_in, rwk_stream_t stream);

int rwk_fused_reduce(const float *k_partials, const int32_t *seg_ptr, const float *counts_f32,
                     int64_t n_graphs, int64_t h, int64_t max_step, int64_t stream_len,
                     float drop_p, int normalize, float *k, rwk_stream_t stream);

/* d_k (G, L*H) -> dx (L, N, D): the DIRECT gradient of each X[i] (the caller then chains
 * dx[i-1] += A^T dx[i] with rwk_spmm_csr), and dz_partials (see rwk_hidden_backward). */
int rwk_fused_backward(const float *x, const float *z, const int32_t *gid, const int32_t *perm,
                       const float *d_k, const float *counts_f32, int64_t n, int64_t d, int64_t h,
                       int64_t s, int64_t max_step, int64_t stream_len, float drop_p,
                       int normalize, const uint64_t *rng, float *dx, float *dz_partials,
                       float *workspace, rwk_stream_t stream);

/* rwk_fused_backward with the end of the head's backward pass inside (training step, hidden_dim <= 4,
 * max_step * h <= 32; rwk_fused_backward_bn_supported says whether): instead of reading d_k, the kernel forms the
 * gradient behind the batch-norm of model.py:68 itself, from what rwk_head_train_forward left: dy (G, L*H), the
 * gradient at the batch-norm's output for d(loss) = 1; head_partials (n_head_partials, head_nv), whose columns
 * [o_gamma, o_gamma + 2 L H) hold the per-workgroup sums for gamma and beta; k (G, L*H), mean, invstd, bn_w; d_loss
 * (one float, device memory).  Same values as rwk_head_train_backward's d_k followed by rwk_fused_backward, bit for
 * bit, with one launch less (the parameter gradients of the head are then summed by
 * rwk_hidden_backward_reduce_adam_head). */
int rwk_fused_backward_bn_supported(int64_t d, int64_t h, int64_t s, int64_t max_step);
int rwk_fused_backward_bn(const float *x, const float *z, const int32_t *gid, const int32_t *perm,
                          const float *counts_f32, int64_t n, int64_t d, int64_t h, int64_t s, int64_t max_step,
                          int64_t stream_len, float drop_p, int normalize, const uint64_t *rng, float *dx,
                          float *dz_partials, const float *k, const float *dy, const float *mean, const float *invstd,
                          const float *bn_w, const float *head_partials, int64_t n_head_partials, int64_t head_nv,
                          int64_t o_gamma, const float *d_loss, int64_t n_graphs, rwk_stream_t stream);

/* Floats of `workspace` rwk_fused_backward needs (0: pass NULL).  Currently always 0: on the
 * matrix-core path the blocks of 16 hidden graphs are added into dx in place, in block order. */
int64_t rwk_fused_backward_workspace(int64_t n, int64_t d, int64_t h, int64_t s, int64_t max_step);

/* Which kernel family serves these hyper-parameters: 1 = register-resident vector-ALU kernels
 * (hidden_dim <= 4), 2 = fp32 matrix-core kernels (hidden_dim 5..64), 0 = generic LDS kernels. */
int rwk_fused_path(int64_t d, int64_t h, int64_t s, int64_t max_step);

/* Recommended stream_len for rwk_fused_forward (backward = 0) or rwk_fused_backward (backward = 1):
 * the number of sorted positions per stream that fills the 256 CUs in whole rounds of resident
 * workgroups.  The two launches may use different values; any stream_len >= 1 is valid, and the
 * partial-buffer sizes above must be queried with the value actually passed. */
int64_t rwk_fused_stream_len(int64_t n, int64_t d, int64_t h, int64_t s, int64_t max_step, int backward);

/* Keep-S variants (rwgnn/rwgnn_pyg.py:132-149: the features of that model are kept per hidden NODE):
 *   K[g, (i*H + h)*S + s] = 1/(1-p) * sum_{p in g} m[i,h,s,n] * (Z0[h,s].X0[n]) * (Zi[h,s].Xi[n])
 * k_partials has rwk_fused_k_partials_size(n, stream_len, G, H*S, L) floats and is reduced with
 * rwk_fused_reduce(..., h = H*S, ..., normalize = 0, ...); d_k is (G, L*H*S) in the same column order; dx, dz_partials
 * and the dropout mask are those of rwk_fused_backward (rwk_fused_dz_partials_count rows).  Built for hidden_dim <= 4,
 * 9..16 hidden graphs, S > 1 and L*S <= 20 (rwk_fused_keep_s_supported); other shapes can treat every
 * (hidden graph, hidden node) pair as a hidden graph of size one and use the entry points above. */
int rwk_fused_keep_s_supported(int64_t d, int64_t h, int64_t s, int64_t max_step);
int rwk_fused_forward_keep_s(const float *x, const float *z, const int32_t *gid, const int32_t *perm, int64_t n,
                             int64_t d, int64_t h, int64_t s, int64_t max_step, int64_t stream_len,
                             float drop_p, const uint64_t *rng, float *k_partials, rwk_stream_t stream);
int rwk_fused_backward_keep_s(const float *x, const float *z, const int32_t *gid, const int32_t *perm,
                              const float *d_k, int64_t n, int64_t d, int64_t h, int64_t s, int64_t max_step,
                              int64_t stream_len, float drop_p, const uint64_t *rng, float *dx,
                              float *dz_partials, rwk_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Classifier head   log_softmax(fc2(dropout(relu(fc1(bn(K))))))     replaces model.py:68-72
 * (BatchNorm1d, Linear, ReLU, Dropout, Linear, log_softmax) and their backward: two launches
 * each way instead of ~35.  K: (g, c) with c = max_step * hidden_graphs; w1: (p, c); w2: (nc, p).
 * Limits: c <= 256, nc <= 32, p * c <= 8192 (rwk_head_supported); callers outside them keep
 * the torch modules.
 * ------------------------------------------------------------------------------------- */
int rwk_head_supported(int64_t g, int64_t c, int64_t p, int64_t nc);
/* workgroups of every head launch = rows of `stats` and of the backward `partials` */
int64_t rwk_head_n_blocks(int64_t g);
/* floats of `stats` (training forward): per workgroup {count, mean[c], M2[c]} */
int64_t rwk_head_stats_size(int64_t g, int64_t c);
/* floats of `grads`, and of one row of `partials`: [dW2 | db2 | dW1 | db1 | dbn_weight | dbn_bias] */
int64_t rwk_head_grad_size(int64_t c, int64_t p, int64_t nc);

/* training != 0: batch statistics (merged over the workgroups in a fixed order), running_mean /
 * running_var / num_batches_tracked (may be NULL) updated as torch.nn.BatchNorm1d does, dropout
 * with keep probability 1 - drop_p from the counter-based generator: {seed, offset} is read
 * from rng_state, copied to rng_used, and the offset in rng_state is advanced by one (so a
 * replayed HIP graph draws a fresh mask).  training == 0: running statistics, no dropout.
 * mean_out, invstd_out (c each) and h1d_out (g, p: hidden activations after ReLU and dropout)
 * are what rwk_head_backward needs; they may be NULL in inference.  out: (g, nc) log-probabilities. */
int rwk_head_forward(const float *k, int64_t g, int64_t c, int64_t p, int64_t nc, const float *bn_w,
                     const float *bn_b, float *running_mean, float *running_var,
                     int64_t *num_batches_tracked, float momentum, float eps, int training, float *stats,
                     const float *w1, const float *b1, const float *w2, const float *b2, float drop_p,
                     uint64_t *rng_state, uint64_t *rng_used, float *mean_out, float *invstd_out,
                     float *h1d_out, float *out, rwk_stream_t stream);

/* d_out: gradient of the log-probabilities.  dy: (g, c) scratch; partials: (rwk_head_n_blocks,
 * rwk_head_grad_size) scratch; grads: rwk_head_grad_size floats out; d_k: (g, c) out or NULL. */
int rwk_head_backward(const float *k, int64_t g, int64_t c, int64_t p, int64_t nc, const float *bn_w,
                      const float *bn_b, const float *mean, const float *invstd, int training,
                      const float *w1, const float *w2, const float *h1d, const float *out,
                      const float *d_out, float drop_p, float *dy, float *partials, float *grads,
                      float *d_k, rwk_stream_t stream);

/* The head together with the mean cross-entropy that rwgnn2/main.py:85 applies to its output (output =
 * model(...); loss = F.cross_entropy(output, y)), without the two cross-entropy launches:
 * rwk_head_forward_loss also writes, per workgroup, the sum of -log p(label) over its rows to loss_partials
 * (rwk_head_n_blocks(g) floats); rwk_head_backward_loss forms d_out = d_loss[0] / g * (softmax(out) -
 * onehot(labels)) in place (d_out is ignored and may be NULL when labels is given) and, when loss is
 * non-null, writes the mean loss (the partials added in a fixed order, divided by g) to loss[0].  The loss
 * value is therefore complete after the backward call; a caller that needs it without a backward pass calls
 * rwk_cross_entropy_forward on out.  labels == NULL (with loss_partials, d_loss, loss NULL): exactly
 * rwk_head_forward / rwk_head_backward. */
int rwk_head_forward_loss(const float *k, int64_t g, int64_t c, int64_t p, int64_t nc, const float *bn_w,
                          const float *bn_b, float *running_mean, float *running_var,
                          int64_t *num_batches_tracked, float momentum, float eps, int training, float *stats,
                          const float *w1, const float *b1, const float *w2, const float *b2, float drop_p,
                          uint64_t *rng_state, uint64_t *rng_used, float *mean_out, float *invstd_out,
                          float *h1d_out, float *out, const int64_t *labels, float *loss_partials,
                          rwk_stream_t stream);
int rwk_head_backward_loss(const float *k, int64_t g, int64_t c, int64_t p, int64_t nc, const float *bn_w,
                           const float *bn_b, const float *mean, const float *invstd, int training,
                           const float *w1, const float *w2, const float *h1d, const float *out,
                           const float *d_out, float drop_p, float *dy, float *partials, float *grads,
                           float *d_k, const int64_t *labels, const float *d_loss, const float *loss_partials,
                           float *loss, rwk_stream_t stream);

/* rwk_fused_reduce (model.py:60-64,67) and the batch statistics of the K it writes (first half of model.py:68) in one
 * launch: stats receives rwk_fused_reduce_stats_blocks(g, C) block partials {count, 3 unused, mean[C], M2[C]},
 * C = h * max_step <= 256, each over rwk_fused_reduce_stats_rows(g, C) consecutive rows (room for them: rwk_head_stats_size(g, C)
 * floats); rng_state / rng_used not NULL: the head's dropout generator state is drawn and advanced, as the
 * statistics launch of rwk_head_forward does.  rwk_head_apply_loss is rwk_head_forward_loss in training mode without
 * its statistics launch, on such partials. */
int64_t rwk_fused_reduce_stats_rows(int64_t g, int64_t c);
int64_t rwk_fused_reduce_stats_blocks(int64_t g, int64_t c);
int rwk_fused_reduce_stats(const float *k_partials, const int32_t *seg_ptr, const float *counts_f32, int64_t n_graphs,
                           int64_t h, int64_t max_step, int64_t stream_len, float drop_p, int normalize, float *k,
                           float *stats, uint64_t *rng_state, uint64_t *rng_used, rwk_stream_t stream);
int rwk_head_apply_loss(const float *k, int64_t g, int64_t c, int64_t p, int64_t nc, const float *bn_w,
                        const float *bn_b, float *running_mean, float *running_var, int64_t *num_batches_tracked,
                        float momentum, float eps, const float *stats, int64_t n_stats, const float *w1,
                        const float *b1, const float *w2, const float *b2, float drop_p, const uint64_t *rng_used,
                        float *mean_out, float *invstd_out, float *h1d_out, float *out, const int64_t *labels,
                        float *loss_partials, rwk_stream_t stream);

/* The head of the training step (model.py:68-72 followed by F.cross_entropy, rwgnn2/main.py:84-85) with the per-tile
 * part of its backward pass done in the forward launch, where the tile's activations are still in LDS: out, per-block
 * loss sums, dy (g, c) and per-block parameter-gradient rows (rwk_head_n_blocks(g) x rwk_head_grad_size) for
 * d(loss) = 1.  n_stats_ready > 0: stats holds that many block partials (rwk_fused_reduce_stats) and rng_used the
 * drawn generator state; 0: the statistics launch runs first, as in rwk_head_forward.  rwk_head_train_backward is
 * the one launch of the backward pass: multiplies by d_loss[0], adds the partial rows in block order (grads, layout
 * of rwk_head_backward), forms d_k and finishes the mean loss (`loss` of the forward call, when not NULL, is set to
 * NaN until then).  Training mode only; rwk_head_train_supported says
 * whether the sizes fit (both passes share one LDS allocation). */
int rwk_head_train_supported(int64_t g, int64_t c, int64_t p, int64_t nc);
int rwk_head_train_forward(const float *k, int64_t g, int64_t c, int64_t p, int64_t nc, const float *bn_w,
                           const float *bn_b, float *running_mean, float *running_var, int64_t *num_batches_tracked,
                           float momentum, float eps, float *stats, int64_t n_stats_ready, const float *w1,
                           const float *b1, const float *w2, const float *b2, float drop_p, uint64_t *rng_state,
                           uint64_t *rng_used, float *mean_out, float *invstd_out, float *out, const int64_t *labels,
                           float *loss_partials, float *dy, float *partials, float *loss, rwk_stream_t stream);
int rwk_head_train_backward(const float *k, int64_t g, int64_t c, int64_t p, int64_t nc, const float *bn_w,
                            const float *mean, const float *invstd, const float *dy, const float *partials,
                            const float *d_loss, const float *loss_partials, float *loss, float *grads, float *d_k,
                            rwk_stream_t stream);

/* Mean cross-entropy of (g, nc) scores against int64 labels in [0, nc): F.cross_entropy with its
 * default arguments, as rwgnn2/main.py:85,92 applies it to the model's output.  loss: device
 * float[1].  Backward: d_scores = d_loss[0] / g * (softmax(scores) - onehot(labels)).
 * One launch each way (torch: six). */
int rwk_cross_entropy_forward(const float *scores, const int64_t *labels, int64_t g, int64_t nc,
                              float *loss, rwk_stream_t stream);
int rwk_cross_entropy_backward(const float *scores, const int64_t *labels, int64_t g, int64_t nc,
                               const float *d_loss, float *d_scores, rwk_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Input projection  X0 = sigmoid(features W^T + b)   replaces model.py:43 for narrow inputs
 * (input_dim <= 16; wider inputs go through a GEMM library and only the sigmoid is applied here)
 * ------------------------------------------------------------------------------------- */

/* features (n, f), weight (d, f), bias (d) -> x0 (n, d). */
int rwk_project_forward(const float *features, const float *weight, const float *bias, int64_t n,
                        int64_t f, int64_t d, float *x0, rwk_stream_t stream);

/* Workgroups rwk_project_backward launches = rows of its `partials` output (host side, pure). */
int64_t rwk_project_backward_blocks(int64_t n, int64_t d);

/* dx0 (n, d): gradient of x0.  partials (blocks, d*f + d): per-workgroup sums of
 * [dW (d, f) | db (d)], to be added row by row with rwk_reduce_rows.  d_features (n, f) is
 * written when non-null.  Requires f <= 16, d <= 64. */
int rwk_project_backward(const float *dx0, const float *x0, const float *features,
                         const float *weight, int64_t n, int64_t f, int64_t d, float *partials,
                         float *d_features, rwk_stream_t stream);

/* The same two kernels without the sigmoid: y = x W^T + b for narrow inputs (f <= 16 input, d <= 64 output
 * channels), the Linear inside RWConv (rwgnn/rwgnn_pyg.py:23,46) at hidden_dim 4, which a GEMM library serves
 * badly (N x 4 times 4 x 4, and an N-long reduction for the weight gradient).  Backward: per-workgroup partial
 * rows [dW (d, f) | db (d)] (rwk_project_backward_blocks(n, d) of them, add with rwk_reduce_rows) and d_x (n, f),
 * which may be NULL. */
int rwk_linear_forward(const float *x, const float *weight, const float *bias, int64_t n, int64_t f, int64_t d,
                       float *y, rwk_stream_t stream);
int rwk_linear_backward(const float *d_y, const float *x, const float *weight, int64_t n, int64_t f, int64_t d,
                        float *partials, float *d_x, rwk_stream_t stream);

/* rwk_hidden_forward and rwk_project_forward in ONE launch (model.py:39-43,56: neither depends on the other),
 * for hidden_dim d = 4 and f <= 16 input features (RWK_E_UNSUPPORTED otherwise: launch the two separately).
 * Arguments and results as the two stand-alone entry points. */
int rwk_project_hidden_forward(const float *features, const float *weight, const float *bias, int64_t n,
                               int64_t f, int64_t d, float *x0, const float *adj_hidden,
                               const float *features_hidden, int64_t h, int64_t s, int64_t max_step,
                               float *ah, float *z, uint64_t *rng_state, uint64_t *rng_used,
                               rwk_stream_t stream);

/* rwk_hidden_backward and rwk_project_backward in ONE launch (their inputs, the reduced dZ partials and
 * dx[0] after the SpMM chain, are both ready at that point of the backward pass); f <= 16, d <= 64. */
int rwk_project_hidden_backward(const float *dx0, const float *x0, const float *features,
                                const float *weight, int64_t n, int64_t f, int64_t d, float *partials,
                                float *d_features, const float *adj_hidden, const float *ah,
                                const float *z, const float *dz_partials, int64_t n_partials, int64_t h,
                                int64_t s, int64_t max_step, float *d_adj_hidden,
                                float *d_features_hidden, rwk_stream_t stream);

/* Launches of the backward pass that do the work of two neighbouring kernels of the chain (hidden_dim 4, input
 * width f <= 16, max_step >= 2), built on the tile SpMM kernel (same gathers, same order of additions per row as
 * rwk_spmm_csr with row_lanes = 0 / rwk_spmm_csr_compact; col16 not NULL selects the compact columns, val NULL
 * means every stored value is 1.0).
 *
 * rwk_spmm_project_backward: autograd of rwgnn2/model.py:43 and of the first propagation step (model.py:55, i = 1)
 * together, d_pre = (dx0 + A^T dx1) * x0 * (1 - x0) with A^T as
 * CSR, partial rows [dW (4, f) | db (4)] (rwk_spmm_project_backward_blocks(n) of them) and d_features (n, f) when
 * not NULL; red_in (red_p > 32 rows, red_v columns) not NULL: rwk_reduce_rows_to32 of it into red_out32 in the same
 * launch.
 *
 * rwk_hidden_backward_reduce: rwk_hidden_backward and, in the same launch, the column sums of red_in (red_p rows,
 * red_v <= 64 columns: the projection's partial rows) into red_out. */
int64_t rwk_spmm_project_backward_blocks(int64_t n);
int rwk_spmm_project_backward(const float *dx0, const float *dx1, const float *x0, const float *features,
                              const float *weight, int64_t n, int64_t f, const int32_t *crow_t, const int32_t *col_t,
                              const int16_t *col16_t, const float *val_t, float *partials, float *d_features,
                              const float *red_in, int64_t red_p, int64_t red_v, float *red_out32,
                              rwk_stream_t stream);
int rwk_hidden_backward_reduce(const float *adj_hidden, const float *ah, const float *z, const float *dz_partials,
                               int64_t n_partials, int64_t h, int64_t s, int64_t d, int64_t max_step,
                               float *d_adj_hidden, float *d_features_hidden, const float *red_in, int64_t red_p,
                               int64_t red_v, float *red_out, rwk_stream_t stream);

/* rwk_hidden_backward_reduce with the optimiser's update of every parameter inside: the last launch of a
 * single-process training step (the end of the autograd backward of model.py:39-43,56 and optimizer.step() of
 * rwgnn2/main.py:87 with torch.optim.Adam's arithmetic, as rwk_adam_step).  Adam is element-wise, so the thread
 * that finishes a gradient element applies it: hidden graph h's workgroup updates its rows of adj_hidden and
 * features_hidden, the column-sum workgroup the red_v values (fc.weight | fc.bias), further workgroups the n_rest
 * remaining tensors, whose gradients rest_srcs[k] (rest_sizes[k] floats, NULL: zeros) an earlier launch left in memory.
 * param / exp_avg / exp_avg_sq / grad_out (may be NULL: the flat gradient is not kept) are flat buffers of n floats
 * holding adj_hidden at off_adj, features_hidden at off_feat, the column sums at off_red - together exactly
 * [0, rest_off) - and the remaining tensors back to back in [rest_off, n).  step_counters:
 * rwk_step_tail_n_counters(h, n, rest_off) int64 at one common step count (each workgroup advances its own).
 * param == NULL: no update (exp_avg, exp_avg_sq, lr, step_counters are not touched and may be NULL); the launch then
 * only completes the flat gradient in grad_out (required) - the data-parallel step, whose all-reduce and
 * rwk_adam_step follow.  corr_slot (may be NULL): four int64 of device memory, zeroed once, that the launches of
 * consecutive steps hand Adam's two bias corrections through: the column-sum workgroup, which ends early, computes the
 * next step's pair (double-precision power loops and a root, 0.6 us in one thread) so that the next launch's
 * workgroups need not; a slot whose step count does not match is ignored.  The slot does not record beta1 / beta2:
 * it belongs to ONE optimiser with fixed betas - zero it when the betas change, never share it. */
int64_t rwk_step_tail_n_counters(int64_t h, int64_t n, int64_t rest_off);
int rwk_hidden_backward_reduce_adam(const float *adj_hidden, const float *ah, const float *z, const float *dz_partials,
                                    int64_t n_partials, int64_t h, int64_t s, int64_t d, int64_t max_step,
                                    float *d_adj_hidden, float *d_features_hidden, const float *red_in, int64_t red_p,
                                    int64_t red_v, float *red_out, float *param, float *exp_avg, float *exp_avg_sq,
                                    float *grad_out, int64_t n, int64_t off_adj, int64_t off_feat, int64_t off_red,
                                    const float *const *rest_srcs, const int64_t *rest_sizes, int64_t n_rest,
                                    int64_t rest_off, const float *lr, float beta1, float beta2, float eps,
                                    float weight_decay, int64_t *step_counters, int64_t *corr_slot, rwk_stream_t stream);

/* The same launch when the head's finishing launch (rwk_head_train_backward) is left out and its batch-norm part
 * runs inside rwk_fused_backward_bn: the n_rest remaining tensors are the head's parameters, and their gradients are
 * summed here from the per-workgroup rows head_partials (n_head_partials <= 256 rows of head_nv floats, for
 * d(loss) = 1) - tensor k takes columns [rest_columns[k], rest_columns[k] + rest_sizes[k]) - in the order of the
 * head's finishing kernel, times d_loss[0]: bit for bit the gradients that kernel writes.  head_grads_out (head_nv
 * floats, may be NULL) receives them in the partial rows' layout; loss_out (may be NULL, then loss_partials too)
 * the mean of the n_head_partials per-workgroup loss sums over n_graphs.  Eight lanes serve one element:
 * step_counters holds rwk_step_tail_n_counters_head(h, n, rest_off) int64.  No counterpart in the reference beyond
 * the lines the two launches it merges already cite (model.py:68-72 backward, main.py:85-87). */
int64_t rwk_step_tail_n_counters_head(int64_t h, int64_t n, int64_t rest_off);
int rwk_hidden_backward_reduce_adam_head(const float *adj_hidden, const float *ah, const float *z,
                                         const float *dz_partials, int64_t n_partials, int64_t h, int64_t s, int64_t d,
                                         int64_t max_step, float *d_adj_hidden, float *d_features_hidden,
                                         const float *red_in, int64_t red_p, int64_t red_v, float *red_out,
                                         float *param, float *exp_avg, float *exp_avg_sq, float *grad_out, int64_t n,
                                         int64_t off_adj, int64_t off_feat, int64_t off_red,
                                         const int64_t *rest_columns, const int64_t *rest_sizes, int64_t n_rest,
                                         int64_t rest_off, const float *lr, float beta1, float beta2, float eps,
                                         float weight_decay, int64_t *step_counters, int64_t *corr_slot,
                                         const float *head_partials,
                                         int64_t n_head_partials, int64_t head_nv, const float *d_loss,
                                         float *head_grads_out, const float *loss_partials, float *loss_out,
                                         int64_t n_graphs, rwk_stream_t stream);

/* out[v] = sum_p in[p, v], rows added in a fixed order (p > 32: 32 row blocks of ceil(p/32) rows,
 * each summed by 4 interleaved slices, then the 32 block sums in index order; v <= 64 columns: one
 * launch, interleaved lanes per column and a halving tree), so the result is bitwise
 * reproducible.  workspace: rwk_reduce_rows_workspace(p, v) floats (0 when p <= 32). */
int64_t rwk_reduce_rows_workspace(int64_t p, int64_t v);
int rwk_reduce_rows(const float *in, int64_t p, int64_t v, float *out, float *workspace,
                    rwk_stream_t stream);
/* First level only: p > 32 rows -> out32 (32, v).  rwk_reduce_rows_level1(p) = rows left
 * after it (p itself when p <= 32, else 32); rwk_hidden_backward adds up to 32 rows itself. */
int64_t rwk_reduce_rows_level1(int64_t p);
int rwk_reduce_rows_to32(const float *in, int64_t p, int64_t v, float *out32, rwk_stream_t stream);
/* ---------------------------------------------------------------------------------------
 * The layer WITHOUT dropout (model.eval(), or dropout 0) through per-graph Gram matrices: rwgnn2/model.py:45-64 with
 * the identity mask factorises as K[g, i*H+h] = sum_s Z0[h,s]^T C[g,i] Zi[h,s], C[g,i] = sum_{n in g} X0[n] (x)
 * Xi[n] - d*d multiply-adds per node and step instead of 2*H*S*d.  The reference's validation and test loops
 * (rwgnn2/main.py:110-117,148-151) run this forward every epoch.  hidden_dim d <= 4, h <= 64, s <= 64, max_step <= 4
 * (rwk_gram_supported).  x (max_step, n, d), z (max_step, h, s, d), seg_ptr (n_graphs + 1) and perm / counts_f32 as
 * rwk_fused_forward takes them.
 *   rwk_gram_forward:  k (n_graphs, max_step*h), column i*h + hh, divided by counts when normalize; c_out (may be
 *                      NULL) keeps C as (n_graphs, max_step, 4, 4) for the backward.
 *   rwk_gram_backward: given d_k and C: dx (max_step, n, d) (every row written; dC[g,i] = sum_h d_k[g, i*h+hh]
 *                      sum_s Z0 (x) Zi) and, in a second launch, dz (max_step, h, s, d) through
 *                      dM[i,h] = sum_g d_k[g, i*h+hh] C[g,i] (graphs added in a fixed strided order).
 * Fixed summation orders throughout: bitwise reproducible.
 * ------------------------------------------------------------------------------------- */
int rwk_gram_supported(int64_t d, int64_t h, int64_t s, int64_t max_step);
int rwk_gram_forward(const float *x, const float *z, const int32_t *seg_ptr, const int32_t *perm,
                     const float *counts_f32, int64_t n, int64_t n_graphs, int64_t d, int64_t h, int64_t s,
                     int64_t max_step, int normalize, float *k, float *c_out, rwk_stream_t stream);
int rwk_gram_backward(const float *x, const float *z, const int32_t *seg_ptr, const int32_t *perm,
                      const float *counts_f32, const float *d_k, const float *c_in, int64_t n, int64_t n_graphs,
                      int64_t d, int64_t h, int64_t s, int64_t max_step, int normalize, float *dx, float *dz,
                      rwk_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * RW_NN.forward in evaluation mode (rwgnn2/model.py:31-72 under model.eval(): identity dropout, batch-norm on its
 * running statistics - what the validation and test loops rwgnn2/main.py:110-117,148-151 call) as ONE launch, one
 * workgroup per graph: projection, propagation on the graph's own CSR rows, hidden-graph chain, Gram matrices, K row
 * and the head, all in LDS.  Requires a block-diagonal matrix (no stored entry joins two graphs), graph ids sorted,
 * hidden_dim d <= 4, at most max_nodes nodes per graph, and rwk_eval_forward_lds(...) != 0 (bytes of LDS the launch
 * takes; 0 = the shape does not fit: use the separate launches).  feat: raw features (n, f), f <= 16, with fc_w (d, f)
 * and fc_b - or, with fc_w NULL, the pre-activations (n, d).  crow / col / val: the int32 CSR (val NULL: all ones).
 * out (n_graphs, nc) log-probabilities, or NULL to stop after k_out (n_graphs, max_step*h) (either may be NULL, not
 * both); the head arguments are read only when out is given.
 * ------------------------------------------------------------------------------------- */
int64_t rwk_eval_forward_lds(int64_t d, int64_t h, int64_t s, int64_t max_step, int64_t f, int64_t max_nodes,
                             int64_t p, int64_t nc);
int rwk_eval_forward(const float *feat, int64_t f, const float *fc_w, const float *fc_b, const float *adj_hidden,
                     const float *feat_hidden, const int32_t *crow, const int32_t *col, const float *val,
                     const int32_t *seg_ptr, const float *counts_f32, int64_t n, int64_t n_graphs, int64_t max_nodes,
                     int64_t d, int64_t h, int64_t s, int64_t max_step, int normalize, const float *bn_w,
                     const float *bn_b, const float *running_mean, const float *running_var, float eps,
                     const float *w1, const float *b1, const float *w2, const float *b2, int64_t p, int64_t nc,
                     float *k_out, float *out, rwk_stream_t stream);

/* The same launch on graphs that were never assembled into a batch (the validation and test loops of
 * rwgnn2/main.py:110-117,148-151 over a data set that does not stay resident as batches): graph g of the launch is
 * entry g of `table`, the (4, b+1) offset table of rwk_batch_build, and its feature rows, 64-bit row pointers and
 * columns are read straight from that call's device-resident data set (unweighted; fc_w, fc_b required).  Nothing
 * is written but k_out / out: the same values as rwk_batch_build followed by rwk_eval_forward. */
int rwk_eval_forward_gather(const int64_t *table, const int64_t *ds_row_ptr, const int32_t *ds_col, const float *ds_feat,
                            int64_t f, const float *fc_w, const float *fc_b, const float *adj_hidden,
                            const float *feat_hidden, int64_t n_graphs, int64_t max_nodes, int64_t d, int64_t h,
                            int64_t s, int64_t max_step, int normalize, const float *bn_w, const float *bn_b,
                            const float *running_mean, const float *running_var, float eps, const float *w1,
                            const float *b1, const float *w2, const float *b2, int64_t p, int64_t nc, float *k_out,
                            float *out, rwk_stream_t stream);

/* The node features of every step and the hidden graphs in one launch (model.py:39-43,55-56 at hidden_dim 4 with
 * at most 16 input features), for block-diagonal batches with sorted graph ids: the workgroup of graph g projects
 * its nodes into LDS and gathers each propagation step from there; xs (max_step, n, 4) receives every X_i; the
 * first h workgroups do what rwk_hidden_forward does (ah, z, generator state).  Replaces
 * rwk_project_hidden_forward followed by max_step - 1 SpMM launches; same bits for rows of at most 64 stored
 * entries.  rwk_x_chain_lds: bytes of LDS the launch takes, 0 when a graph of max_nodes nodes does not fit. */
int64_t rwk_x_chain_lds(int64_t f, int64_t max_nodes, int64_t s);
int rwk_x_chain_hidden_forward(const float *features, const float *weight, const float *bias, int64_t n, int64_t f,
                               const int32_t *crow, const int32_t *col, const float *val, const int32_t *seg_ptr,
                               int64_t n_graphs, int64_t max_nodes, int64_t max_step, float *xs,
                               const float *adj_hidden, const float *features_hidden, int64_t h, int64_t s, float *ah,
                               float *z, uint64_t *rng_state, uint64_t *rng_used, rwk_stream_t stream);

/* The same launch for a batch that is NOT assembled yet (generate_batches' loop body, utils.py:60-83, inside the
 * forward's first launch): the workgroup of batch graph g reads the graph's feature rows, 64-bit row pointers and
 * columns from the device-resident data set of rwk_batch_build (unweighted; `table` is that call's (4, b+1) table) and,
 * besides xs, leaves behind what the launches after it read of the batch: feat (n, f), y (b), crow (n+1), col16
 * (rwk_csr_compact's form: every graph of the batch at most 32512 nodes), gid, seg_ptr (b+1), counts_f32 - the same
 * values rwk_batch_build writes into those arrays.  max_step >= 2 (the first propagation step writes the rows'
 * pointers and columns).  Same xs, ah, z as rwk_batch_build followed by rwk_x_chain_hidden_forward. */
typedef struct rwk_x_chain_gather_t {
    const int64_t *table, *ds_row_ptr;
    const int32_t *ds_col;
    const float *ds_feat;
    const int64_t *ds_y;
    float *feat;
    int64_t *y;
    int32_t *crow;
    int16_t *col16;
    int32_t *gid, *seg_ptr;
    float *counts_f32;
} rwk_x_chain_gather_t;

int rwk_x_chain_hidden_forward_gather(const rwk_x_chain_gather_t *gather, const float *weight, const float *bias,
                                      int64_t n, int64_t f, int64_t n_graphs, int64_t max_nodes, int64_t max_step,
                                      float *xs, const float *adj_hidden, const float *features_hidden, int64_t h,
                                      int64_t s, float *ah, float *z, uint64_t *rng_state, uint64_t *rng_used,
                                      rwk_stream_t stream);

/* The same chain for LARGE graphs of a block-diagonal batch (thousands of nodes, power-law hubs; rwgnn2/model.py:43,55
 * at hidden_dim 4, at most 16 input features, sorted graph ids): one workgroup of up to 1024 threads per graph with
 * the graph in LDS - 16 bytes per node for the current rows, the row pointers, and the columns as 16-bit node numbers
 * within the graph (entries beyond the LDS share are read from memory).  Rows of at most 16 stored entries are
 * summed by their own thread in stored order (same bits as the SpMM launches), rows of at most 128 by 16 lanes, longer
 * ones by a wavefront (lane j adds entries j, j + lanes, ... in order, lane sums in a fixed butterfly).
 * rwk_graph_chain_forward_lds: LDS bytes, 0 = unsupported (graphs of more than 4096 nodes). */
int64_t rwk_graph_chain_forward_lds(int64_t f, int64_t max_nodes, int64_t s);
int rwk_graph_chain_forward(const float *features, const float *weight, const float *bias, int64_t n, int64_t f,
                            const int32_t *crow, const int32_t *col, const float *val, const int32_t *seg_ptr,
                            int64_t n_graphs, int64_t max_nodes, int64_t max_step, float *xs, const float *adj_hidden,
                            const float *features_hidden, int64_t h, int64_t s, float *ah, float *z,
                            uint64_t *rng_state, uint64_t *rng_used, rwk_stream_t stream);

/* Its backward (autograd of rwgnn2/model.py:55 and :43): dX_{i-1} += A^T dX_i for i = max_step-1 .. 1 with the
 * graph's gradient rows in LDS, then on the finished dX_0 the projection's backward, one partial row
 * [dW (4, f) | db (4)] per graph in partials (n_graphs, 4 f + 4) (summed by rwk_hidden_backward_reduce and its
 * variants, or rwk_reduce_rows), d_features (n, f) when not NULL.  dx (max_step, n, 4) holds every step's own share
 * (rwk_fused_backward*); it is only read.  crow_t / col_t / val_t: CSR of A^T (val_t NULL: every value is 1).
 * Replaces max_step - 2 accumulating SpMM launches and rwk_spmm_project_backward.  red_in (may be NULL): the first
 * level of an independent row reduction rides in trailing workgroups, as in rwk_spmm_project_backward. */
int64_t rwk_graph_chain_backward_lds(int64_t f, int64_t max_nodes);
int rwk_graph_chain_backward(const float *dx, const float *x0, const float *features, const float *weight, int64_t n,
                             int64_t f, const int32_t *crow_t, const int32_t *col_t, const float *val_t,
                             const int32_t *seg_ptr, int64_t n_graphs, int64_t max_nodes, int64_t max_step,
                             float *partials, float *d_features, const float *red_in, int64_t red_p, int64_t red_v,
                             float *red_out32, rwk_stream_t stream);

/* rwk_reduce_rows for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same kernels, same order of
 * additions, same bits as rwk_reduce_rows; the 32 x v level-1 sums live in a device array of the library itself
 * (RWK_REDUCE_ROWS_STATIC_MAX_V columns, 2 MiB, held by every GPU the library runs on until the process ends)
 * instead of a caller-owned workspace.  v above that limit (when a workspace is needed at all, i.e. p > 32 and
 * (v > 64 or p > 2^20)) returns RWK_E_STATIC_SIZE.  Calls no hipMalloc / hipFree.  All calls on one GPU share that
 * GPU's array: calls that overlap on one GPU (different streams, or different host threads without stream
 * order between them) race.  The device that owns `stream` must be the current device (RWK_E_DEVICE otherwise: every
 * GPU has its own copy of the array at its own address).  The first call on a device looks that address up
 * (hipGetSymbolAddress); under graph capture, make one call per device before the capture begins. */
#define RWK_REDUCE_ROWS_STATIC_MAX_V 16384
int rwk_reduce_rows_static(const float *in, int64_t p, int64_t v, float *out, rwk_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Adam over one flat parameter buffer (torch.optim.Adam semantics, rwgnn2/main.py:79,87):
 * one launch per step.  lr: device float[1].  step_counters: rwk_adam_n_counters(n) device
 * int64, zero before the first step (the step count lives on the device, so a captured
 * graph replays correctly).
 * ------------------------------------------------------------------------------------- */
int64_t rwk_adam_n_counters(int64_t n);

/* dst = concatenation of n_chunks device arrays: srcs (HOST array of device pointers; a NULL
 * entry stands for zeros) with sizes (HOST array of element counts).  The pointers travel as
 * kernel arguments, 16 chunks per launch: per-parameter gradients -> one flat bucket for the
 * all-reduce and the Adam launch, instead of one accumulation kernel per parameter. */
int rwk_gather_chunks(const float *const *srcs, const int64_t *sizes, int64_t n_chunks, float *dst,
                      rwk_stream_t stream);
int rwk_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                  const float *lr, float beta1, float beta2, float eps, float weight_decay,
                  int64_t *step_counters, rwk_stream_t stream);

/* The same update with the gradient read from n_chunks <= 16 separate tensors (srcs[k]: sizes[k] floats, NULL =
 * zeros; sum of sizes = n) in the order of the flat parameter buffer: rwk_gather_chunks and rwk_adam_step in
 * one launch, for a single process (no all-reduce between the two).  grad_out (n floats) receives the flat
 * gradient when not NULL.  More than 16 chunks: RWK_E_UNSUPPORTED (gather first). */
int rwk_adam_step_chunks(float *param, const float *const *srcs, const int64_t *sizes, int64_t n_chunks,
                         float *grad_out, float *exp_avg, float *exp_avg_sq, int64_t n, const float *lr,
                         float beta1, float beta2, float eps, float weight_decay, int64_t *step_counters,
                         rwk_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * One call per direction for the whole training-mode model (model.py:32-72 and its autograd backward): the same
 * launches as the entry points above, in the same order, with every intermediate in two caller-owned workspaces.
 * An eager training loop (rwgnn2/main.py:82-88) then pays two host calls and three allocations per step instead
 * of a dozen calls and some thirty allocations.  Forward = rwk_x_chain_hidden_forward, rwk_fused_forward_bits,
 * rwk_fused_reduce_stats, rwk_head_apply_loss; backward = rwk_head_backward, rwk_fused_backward_bits, the
 * transposed propagation steps (the first of them with the level-1 sum of the hidden-feature partials in its
 * launch), rwk_spmm_project_backward, rwk_hidden_backward_reduce.  Results are those of the separate calls, bit
 * for bit.
 *
 * Served shape (rwk_model_plan returns RWK_E_UNSUPPORTED otherwise): hidden_dim 4, 2 <= max_step <= 4, at most 15
 * input features, a block-diagonal batch with sorted graph ids whose graphs fit rwk_x_chain_lds, rows of at most
 * 64 stored entries, a transposed matrix the tile SpMM kernel serves, a head rwk_head_supported serves,
 * h * max_step <= 256.
 *
 * rwk_model_step_t (HOST struct) names the batch, the parameters and the hyper-parameters; the library reads it
 * during the call only.  val / val_t may be NULL when every stored value is 1.0 and (for val_t) col16_t is given.
 * rng_state ({seed, offset}, advanced by the forward call) may be NULL when drop_p == 0.
 * rwk_model_plan fills rwk_model_plan_t (HOST struct) for one (batch structure, hyper-parameters) pair: workspace
 * sizes in bytes and the launch constants; keep it and pass it to both calls.
 * forward_ws (plan->forward_bytes, 256-byte aligned) is written by rwk_model_forward and read by
 * rwk_model_backward; backward_ws (plan->backward_bytes) is scratch of the backward call.  out: (n_graphs, nc)
 * log-probabilities.  grads (plan->grad_floats floats) receives
 * [d adj_hidden | d features_hidden | d fc.weight (4, f) | d fc.bias (4) | the head's gradient vector in the
 * layout of rwk_head_backward].
 * ------------------------------------------------------------------------------------- */
typedef struct rwk_model_step_t {
    /* batch */
    int64_t n, f, nnz, n_graphs, max_graph_nodes;
    int32_t max_degree, max_degree_t;                        /* longest row of A and of A^T */
    const float *features;                                   /* (n, f) */
    const int32_t *crow, *col;                               /* A, CSR */
    const float *val;
    const int32_t *crow_t, *col_t;                           /* A^T, CSR (the same arrays when A is symmetric) */
    const int16_t *col16_t;                                  /* compact columns of A^T or NULL */
    const float *val_t;
    int32_t unit_values_t, long_threshold_t;
    const int32_t *seg_ptr, *gid;
    const float *counts_f32;
    /* hyper-parameters */
    int64_t h, s, d, max_step, p, nc;
    int32_t normalize, reserved;
    float drop_p, momentum, eps, reserved_f;
    /* parameters and state */
    const float *fc_w, *fc_b, *adj_hidden, *features_hidden, *bn_w, *bn_b, *w1, *b1, *w2, *b2;
    float *running_mean, *running_var;
    int64_t *num_batches_tracked;
    uint64_t *rng_state;
} rwk_model_step_t;

typedef struct rwk_model_plan_t {
    int64_t forward_bytes, backward_bytes, grad_floats;
    int64_t stream_len_f, stream_len_b, n_partials, dz_rows, stats_blocks, spb_blocks, head_blocks, head_grad_floats;
    int64_t off_f[12];                                       /* byte offsets into forward_ws */
    int64_t off_b[8];                                        /* byte offsets into backward_ws */
    int64_t train_step;                                      /* not 0: rwk_model_train_step serves this shape too */
} rwk_model_plan_t;

int rwk_model_plan(const rwk_model_step_t *m, rwk_model_plan_t *plan);
int rwk_model_forward(const rwk_model_step_t *m, const rwk_model_plan_t *plan, void *forward_ws, float *out,
                      rwk_stream_t stream);
int rwk_model_backward(const rwk_model_step_t *m, const rwk_model_plan_t *plan, const void *forward_ws,
                       void *backward_ws, const float *out, const float *d_out, float *grads, rwk_stream_t stream);

/* The whole training step of rwgnn2/main.py:82-88 for one process in ONE host call: forward, cross-entropy, backward
 * and (when the tail's param is not NULL) the Adam update - the eight launches of the captured step, here for loops
 * that cannot replay a graph because every batch has its own shape.  Launches: rwk_x_chain_hidden_forward,
 * rwk_fused_forward_bits, rwk_fused_reduce_stats, rwk_head_train_forward, rwk_fused_backward_bn, the transposed
 * propagation steps, rwk_spmm_project_backward, rwk_hidden_backward_reduce_adam_head (plan->train_step 1), or, where
 * rwk_fused_backward_bn is not built (max_step > 2; plan->train_step 2), rwk_head_train_backward,
 * rwk_fused_backward_bits and rwk_hidden_backward_reduce_adam in their places; same bits as those calls.
 * Served when plan->train_step is not 0 (the shape of rwk_model_forward with s > 1 and a head
 * rwk_head_train_supported serves).  labels: int64 class index per graph.  d_loss: device float, the gradient of
 * the loss (1.0).  loss (device float) is complete when the step has run.  grads as in rwk_model_backward;
 * tail->grad_out (tail->n floats) receives the flat gradient in the order of the flat parameter buffer
 * [adj_hidden | features_hidden | fc.weight | fc.bias | bn.weight | bn.bias | fc1.weight | fc1.bias | fc2.weight |
 * fc2.bias] (tail->rest_sizes: the sizes of the last six, tail->rest_columns: where each starts in the head's
 * gradient vector).  tail->param NULL (data-parallel step): no update, the flat gradient only. */
typedef struct rwk_train_tail_t {
    float *param, *exp_avg, *exp_avg_sq, *grad_out;
    int64_t n, off_adj, off_feat, off_red, rest_off, n_rest;
    int64_t rest_columns[8], rest_sizes[8];
    const float *lr;
    float beta1, beta2, eps, weight_decay;
    int64_t *step_counters, *corr_slot;
} rwk_train_tail_t;

int rwk_model_train_step(const rwk_model_step_t *m, const rwk_model_plan_t *plan, const rwk_train_tail_t *tail,
                         void *forward_ws, void *backward_ws, const int64_t *labels, const float *d_loss, float *out,
                         float *loss, float *grads, rwk_stream_t stream);

/* A whole pass of training steps over batches that are assembled right before their step, in ONE host call: the
 * loop of rwgnn2/main.py:103-106 (`for i in range(n_train_batches): train(...)`) for a data set too large to keep
 * its batches resident (utils.py:47-85 builds them all up front).  Per batch: rwk_batch_build without the arrays only
 * tensors are made of (COO list, int64 graph ids and sizes; the value array too when the batch is unweighted with
 * compact columns) into slot t % depth; then rwk_model_train_step on `main` with the loss of batch t at losses[t].
 * Where the assembly runs (use_side):
 *   RWK_PASS_SAME_STREAM  on `main`, in front of its step;
 *   RWK_PASS_SIDE_STREAM  on `side`, one batch ahead of the step that reads it (the side stream first waits for
 *                         everything queued on `main`, a slot's next assembly for the step that last read it);
 *   RWK_PASS_IN_FORWARD   inside the step's first launch (rwk_x_chain_hidden_forward_gather: no assembly launch at
 *                         all) for the batches that allow it - an unweighted data set, compact columns - and on
 *                         `main` in front of the step for the others.
 * The same values in the same arrays as those calls made batch by batch: the same bits.
 *   ds:      the device-resident data set (rwk_batch_build's ds_* arguments); unit_values: every stored value 1.0f
 *   tables:  the offset tables of all batches, batch t's (4, b+1) table at tables + batches[t].table_off
 *   batches: host-side shapes; max_degree = longest row of the batch, fits16 = every graph has at most 32512 nodes,
 *            long_threshold = the packed SpMM's long-row cut for this batch (0: none)
 *   model:   hyper-parameters, parameter, running-statistics and generator-state pointers (the batch fields are
 *            filled in per batch from `batches` and `slots`); the data set's adjacency must be symmetric
 * rwk_model_train_pass_plan runs no launch: need->served = the leading batches the pass serves (it stops at the
 * first batch rwk_model_plan refuses or whose plan->train_step differs from the first batch's - the caller steps that
 * one its own way and may call again behind it), and the largest workspaces among them.  rwk_model_train_pass takes
 * exactly that many batches and workspaces of at least those sizes (`room` = the plan's result; RWK_E_SIZE when a
 * batch needs more). */
#define RWK_PASS_SAME_STREAM 0
#define RWK_PASS_SIDE_STREAM 1
#define RWK_PASS_IN_FORWARD 2

typedef struct rwk_pass_dataset_t {
    const int64_t *row_ptr;
    const int32_t *col;
    const float *val, *features;
    int64_t n_feat;
    const int64_t *y;
    int32_t unit_values, reserved;
} rwk_pass_dataset_t;

typedef struct rwk_pass_slot_t {
    float *feats;
    int64_t *y;
    int32_t *crow, *col;
    float *val;
    int16_t *col16;                                          /* may be NULL */
    int32_t *gid, *seg_ptr;
    float *counts_f32;
} rwk_pass_slot_t;

typedef struct rwk_pass_batch_t {
    int64_t b, n, nnz, max_graph_nodes, table_off;
    int32_t max_degree, fits16, long_threshold, reserved;
} rwk_pass_batch_t;

typedef struct rwk_pass_need_t {
    int64_t served, train_step, forward_bytes, backward_bytes, grad_floats, out_floats;
} rwk_pass_need_t;

int rwk_model_train_pass_plan(const rwk_pass_dataset_t *ds, const rwk_pass_batch_t *batches, int64_t n_batches,
                              const rwk_pass_slot_t *slots, int64_t depth, const rwk_model_step_t *model,
                              rwk_pass_need_t *need);
int rwk_model_train_pass(const rwk_pass_dataset_t *ds, const int64_t *tables, const rwk_pass_batch_t *batches,
                         int64_t n_batches, const rwk_pass_slot_t *slots, int64_t depth,
                         const rwk_model_step_t *model, const rwk_train_tail_t *tail, const rwk_pass_need_t *room,
                         void *forward_ws, void *backward_ws, float *grads, float *out, const float *d_loss,
                         float *losses, rwk_stream_t main, rwk_stream_t side, int use_side);

/* Test hook: writes the keep-mask exactly as the fused kernels draw it, one byte per element,
 * shape (L, H, S, N), for the {seed, offset} in rng. */
int rwk_dropout_mask(const uint64_t *rng, int64_t n, int64_t h, int64_t s, int64_t max_step,
                     float drop_p, uint8_t *keep, rwk_stream_t stream);

/* ---- distance-biased transition operator (rwgnn2/rwgnn_dis_bias6.py:109-128 and :222-250) --------------------
 * P = (1 - c) * rownorm(W) + c / n_g per graph, W[i][j] = f(hop distance) off the diagonal, every pair further
 * apart than max_distance at max_distance + 1.  W = w_far * (1 1^T - I) + Y with Y sparse over the
 * max_distance-hop neighbourhoods; these entry points build Y's structure and apply the dense remainder, Y x is
 * rwk_spmm_csr on (row_start, out_col, values).
 *
 * rwk_khop_count / rwk_khop_fill: breadth-first search from every node inside its own graph (gid: graph of each
 * node, seg_ptr: first node of each graph; nodes of a graph contiguous; max_graph_nodes: size of the largest
 * graph, at most 409,600).  count: row_count[i] = nodes within 1..max_distance hops of i, level_count[i *
 * max_distance + d - 1] = nodes at exactly d hops.  fill: with row_start = exclusive prefix sum of row_count
 * (n + 1 entries), writes each row's (node, distance) pairs ordered by distance, then node.  The reference computes
 * the same distances from dense powers of the adjacency matrix (:109-128) or scipy's dijkstra with a limit
 * (:130-154). */
int rwk_khop_count(const int32_t *crow, const int32_t *col, const int32_t *gid, const int32_t *seg_ptr, int64_t n,
                   int64_t max_graph_nodes, int64_t max_distance, int32_t *row_count, int32_t *level_count,
                   rwk_stream_t stream);
int rwk_khop_fill(const int32_t *crow, const int32_t *col, const int32_t *gid, const int32_t *seg_ptr, int64_t n,
                  int64_t max_graph_nodes, int64_t max_distance, const int32_t *row_start, int32_t *out_col,
                  int32_t *out_dist, rwk_stream_t stream);

/* out[g][c] = sum over the rows of graph g of x[r][c] (fixed order); d <= 256. */
int rwk_segment_sum(const float *x, const int32_t *seg_ptr, int64_t n_graphs, int64_t d, float *out,
                    rwk_stream_t stream);

/* out[i][c] = a_i * (w_far * (s1[g][c] - x[i][c]) + y[i][c]) + graph_scale[g] * s2[g][c] with g = gid[i] and
 * a_i = row_scale[i] (1 when row_scale is NULL).  Forward of P x (:244): x, y = Y x, s1 = s2 = per-graph sums of
 * x, row_scale = (1 - c) / rowsum, graph_scale = c / n_g.  Backward (P^T g): x = v = row_scale * g, y = Y v,
 * s1 = sums of v, s2 = sums of g, row_scale NULL. */
int rwk_distance_bias_combine(const float *x, const float *y, const float *s1, const float *s2, const int32_t *gid,
                              const float *row_scale, const float *graph_scale, float w_far, int64_t n, int64_t d,
                              float *out, rwk_stream_t stream);

/* out[i][c] = x[i][c] * scale[i] */
int rwk_row_scale(const float *x, const float *scale, int64_t n, int64_t d, float *out, rwk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RWK_H_ */
