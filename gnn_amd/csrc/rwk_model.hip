// One host call per direction for the training-mode model: the launches of model.py:32-72 and of their autograd
// backward, sequenced here instead of in the caller's language, with every intermediate carved from two
// caller-owned workspaces.  Host code only: the kernels are those behind the entry points called below, in the
// order the separate calls make them, so the results are theirs bit for bit.
#include "rwk_common.h"

namespace {

constexpr int64_t kAlign = 256;

inline int64_t aligned(int64_t bytes) { return (bytes + kAlign - 1) / kAlign * kAlign; }

// forward workspace slots
enum { F_XS, F_Z, F_AH, F_RNG_LAYER, F_KPART, F_K, F_STATS, F_RNG_HEAD, F_MEAN, F_INVSTD, F_H1D, F_COUNT };
// backward workspace slots
enum { B_DX, B_DZ_PART, B_WS, B_DZ32, B_SPB, B_DY, B_HEAD_PART, B_DK, B_COUNT };
static_assert(F_COUNT <= 12 && B_COUNT <= 8, "rwk_model_plan_t holds 12 forward and 8 backward offsets");

// the rule of the tile SpMM schedule at hidden_dim 4: sparse rows or hubs
inline bool tile_spmm_serves(int64_t n, int64_t nnz, int32_t max_degree) {
    const double mean = (double)nnz / (double)(n > 0 ? n : 1);
    if (mean >= 32.0) return false;
    return max_degree > 128 || mean * 256.0 >= 48.0;
}

inline int64_t pairs(int64_t s) { return s * (s - 1) / 2; }

template <class T>
inline T *at(void *ws, int64_t off) { return reinterpret_cast<T *>(static_cast<char *>(ws) + off); }
template <class T>
inline const T *at(const void *ws, int64_t off) { return reinterpret_cast<const T *>(static_cast<const char *>(ws) + off); }

}  // namespace

extern "C" int rwk_model_plan(const rwk_model_step_t *m, rwk_model_plan_t *plan) {
    if (!m || !plan) return RWK_E_NULL;
    if (m->n <= 0 || m->n_graphs < 2 || m->f <= 0 || m->nnz <= 0 || m->max_graph_nodes <= 0) return RWK_E_SIZE;
    const int64_t c = m->h * m->max_step;
    if (m->d != 4 || m->max_step < 2 || m->max_step > 4 || m->f > 15 || m->h < 1 || m->h > 64 || m->s < 1 ||
        c > 256 || m->max_degree > 64 || !(m->drop_p >= 0.f) || !(m->drop_p < 1.f))
        return RWK_E_UNSUPPORTED;
    if (rwk_x_chain_lds(m->f, m->max_graph_nodes, m->s) <= 0) return RWK_E_UNSUPPORTED;
    if (!tile_spmm_serves(m->n, m->nnz, m->max_degree_t)) return RWK_E_UNSUPPORTED;
    if (!rwk_head_supported(m->n_graphs, c, m->p, m->nc)) return RWK_E_UNSUPPORTED;
    if (rwk_fused_keep_words(m->n, m->d, m->h, m->s, m->max_step) != 0) return RWK_E_UNSUPPORTED;

    rwk_model_plan_t pl = {};
    pl.stream_len_f = rwk_fused_stream_len(m->n, m->d, m->h, m->s, m->max_step, 0);
    pl.stream_len_b = rwk_fused_stream_len(m->n, m->d, m->h, m->s, m->max_step, 1);
    pl.n_partials = rwk_fused_dz_partials_count(m->n, pl.stream_len_b, m->d, m->h, m->s, m->max_step);
    if (pl.stream_len_f <= 0 || pl.stream_len_b <= 0 || pl.n_partials <= 0) return RWK_E_UNSUPPORTED;
    pl.dz_rows = pl.n_partials > 32 ? rwk_reduce_rows_level1(pl.n_partials) : pl.n_partials;
    pl.stats_blocks = rwk_fused_reduce_stats_blocks(m->n_graphs, c);
    pl.spb_blocks = rwk_spmm_project_backward_blocks(m->n);
    pl.head_blocks = rwk_head_n_blocks(m->n_graphs);
    pl.head_grad_floats = rwk_head_grad_size(c, m->p, m->nc);
    const int64_t k_slots = rwk_fused_k_partials_size(m->n, pl.stream_len_f, m->n_graphs, m->h, m->max_step);
    const int64_t ws_floats = rwk_fused_backward_workspace(m->n, m->d, m->h, m->s, m->max_step);
    const int64_t zv = m->max_step * m->h * m->s * m->d;                 // elements of Z (and of one dZ partial)
    const int64_t wb = m->d * m->f + m->d;

    const int64_t fsz[F_COUNT] = {
        4 * m->max_step * m->n * m->d,                                   // xs
        4 * zv,                                                          // z
        4 * m->h * m->s * m->s,                                          // ah
        16,                                                              // the layer's drawn generator state
        4 * k_slots,                                                     // K slot partials
        4 * m->n_graphs * c,                                             // K
        4 * rwk_head_stats_size(m->n_graphs, c),                         // batch-statistics partials
        16,                                                              // the head's drawn generator state
        4 * c, 4 * c,                                                    // batch mean, 1 / std
        4 * m->n_graphs * m->p,                                          // fc1 output after ReLU and dropout
    };
    int64_t off = 0;
    for (int i = 0; i < F_COUNT; ++i) { pl.off_f[i] = off; off += aligned(fsz[i]); }
    pl.forward_bytes = off;
    const int64_t bsz[B_COUNT] = {
        4 * m->max_step * m->n * m->d,                                   // dX
        4 * pl.n_partials * zv,                                          // dZ partials
        4 * ws_floats,                                                   // the fused backward's own workspace
        pl.n_partials > 32 ? 4 * pl.dz_rows * zv : 0,                    // dZ partials after the first level
        4 * pl.spb_blocks * wb,                                          // [dW | db] partial rows
        4 * m->n_graphs * c,                                             // dY (batch-norm output gradient)
        4 * pl.head_blocks * pl.head_grad_floats,                        // the head's gradient partial rows
        4 * m->n_graphs * c,                                             // dK
    };
    off = 0;
    for (int i = 0; i < B_COUNT; ++i) { pl.off_b[i] = off; off += aligned(bsz[i]); }
    pl.backward_bytes = off;
    pl.grad_floats = m->h * pairs(m->s) + m->h * m->s * m->d + wb + pl.head_grad_floats;
    // 1: the head's batch-norm backward rides in the fused backward kernel; 2: the head finishes in its own launch
    pl.train_step = !(m->s > 1 && rwk_head_train_supported(m->n_graphs, c, m->p, m->nc)) ? 0
                    : rwk_fused_backward_bn_supported(m->d, m->h, m->s, m->max_step) ? 1 : 2;
    *plan = pl;
    return RWK_OK;
}

static int check_step(const rwk_model_step_t *m, const rwk_model_plan_t *plan) {
    if (!m || !plan) return RWK_E_NULL;
    if (!m->features || !m->crow || !m->col || !m->crow_t || !m->col_t || !m->seg_ptr || !m->gid || !m->counts_f32 ||
        !m->fc_w || !m->fc_b || !m->features_hidden || !m->bn_w || !m->bn_b || !m->w1 || !m->b1 || !m->w2 || !m->b2 ||
        !m->running_mean || !m->running_var)
        return RWK_E_NULL;
    if (m->s > 1 && !m->adj_hidden) return RWK_E_NULL;
    if (m->drop_p > 0.f && !m->rng_state) return RWK_E_NULL;
    if (!m->val_t && !(m->col16_t && m->unit_values_t)) return RWK_E_NULL;
    if (plan->forward_bytes <= 0 || plan->backward_bytes <= 0) return RWK_E_SIZE;
    return RWK_OK;
}

extern "C" int rwk_model_forward(const rwk_model_step_t *m, const rwk_model_plan_t *plan, void *forward_ws, float *out,
                                 rwk_stream_t stream) {
    int rc = check_step(m, plan);
    if (rc) return rc;
    if (!forward_ws || !out) return RWK_E_NULL;
    if (reinterpret_cast<uintptr_t>(forward_ws) % kAlign) return RWK_E_ALIGN;
    const int64_t *o = plan->off_f;
    const bool drop = m->drop_p > 0.f;
    const int64_t c = m->h * m->max_step;
    float *xs = at<float>(forward_ws, o[F_XS]), *z = at<float>(forward_ws, o[F_Z]);
    float *k = at<float>(forward_ws, o[F_K]), *stats = at<float>(forward_ws, o[F_STATS]);
    uint64_t *rng_layer = drop ? at<uint64_t>(forward_ws, o[F_RNG_LAYER]) : nullptr;
    uint64_t *rng_head = drop ? at<uint64_t>(forward_ws, o[F_RNG_HEAD]) : nullptr;
    // model.py:39-43,55-56: X_0 .. X_{L-1}, the hidden graphs, the layer's draw of the generator state
    rc = rwk_x_chain_hidden_forward(m->features, m->fc_w, m->fc_b, m->n, m->f, m->crow, m->col, m->val, m->seg_ptr,
                                    m->n_graphs, m->max_graph_nodes, m->max_step, xs, m->adj_hidden, m->features_hidden,
                                    m->h, m->s, at<float>(forward_ws, o[F_AH]), z, drop ? m->rng_state : nullptr,
                                    rng_layer, stream);
    if (rc) return rc;
    // model.py:45-64: contraction, mask, product, per-stream sums
    rc = rwk_fused_forward_bits(xs, z, m->gid, nullptr, m->n, m->d, m->h, m->s, m->max_step, plan->stream_len_f,
                                m->drop_p, rng_layer, at<float>(forward_ws, o[F_KPART]), nullptr, stream);
    if (rc) return rc;
    // model.py:60-67: K, the head's batch statistics and its draw of the generator state
    rc = rwk_fused_reduce_stats(at<float>(forward_ws, o[F_KPART]), m->seg_ptr, m->counts_f32, m->n_graphs, m->h,
                                m->max_step, plan->stream_len_f, m->drop_p, m->normalize, k, stats,
                                drop ? m->rng_state : nullptr, rng_head, stream);
    if (rc) return rc;
    // model.py:68-72
    return rwk_head_apply_loss(k, m->n_graphs, c, m->p, m->nc, m->bn_w, m->bn_b, m->running_mean, m->running_var,
                               m->num_batches_tracked, m->momentum, m->eps, stats, plan->stats_blocks, m->w1, m->b1,
                               m->w2, m->b2, m->drop_p, rng_head, at<float>(forward_ws, o[F_MEAN]),
                               at<float>(forward_ws, o[F_INVSTD]), at<float>(forward_ws, o[F_H1D]), out, nullptr, nullptr,
                               stream);
}

extern "C" int rwk_model_backward(const rwk_model_step_t *m, const rwk_model_plan_t *plan, const void *forward_ws,
                                  void *backward_ws, const float *out, const float *d_out, float *grads,
                                  rwk_stream_t stream) {
    int rc = check_step(m, plan);
    if (rc) return rc;
    if (!forward_ws || !backward_ws || !out || !d_out || !grads) return RWK_E_NULL;
    if (reinterpret_cast<uintptr_t>(forward_ws) % kAlign || reinterpret_cast<uintptr_t>(backward_ws) % kAlign)
        return RWK_E_ALIGN;
    const int64_t *of = plan->off_f, *ob = plan->off_b;
    const bool drop = m->drop_p > 0.f;
    const int64_t c = m->h * m->max_step, nd = m->n * m->d;
    const int64_t zv = m->max_step * m->h * m->s * m->d, wb_n = m->d * m->f + m->d;
    const float *xs = at<float>(forward_ws, of[F_XS]), *z = at<float>(forward_ws, of[F_Z]);
    const float *ah = at<float>(forward_ws, of[F_AH]), *k = at<float>(forward_ws, of[F_K]);
    const uint64_t *rng_layer = drop ? at<uint64_t>(forward_ws, of[F_RNG_LAYER]) : nullptr;
    float *dx = at<float>(backward_ws, ob[B_DX]), *dz_part = at<float>(backward_ws, ob[B_DZ_PART]);
    float *d_k = at<float>(backward_ws, ob[B_DK]), *spb = at<float>(backward_ws, ob[B_SPB]);
    float *d_adj = m->s > 1 ? grads : nullptr;
    float *d_feat = grads + m->h * pairs(m->s);
    float *wb = d_feat + m->h * m->s * m->d;
    float *head_grads = wb + wb_n;
    // the head (model.py:68-72)
    rc = rwk_head_backward(k, m->n_graphs, c, m->p, m->nc, m->bn_w, m->bn_b, at<float>(forward_ws, of[F_MEAN]),
                           at<float>(forward_ws, of[F_INVSTD]), 1, m->w1, m->w2, at<float>(forward_ws, of[F_H1D]), out,
                           d_out, m->drop_p, at<float>(backward_ws, ob[B_DY]), at<float>(backward_ws, ob[B_HEAD_PART]),
                           head_grads, d_k, stream);
    if (rc) return rc;
    // model.py:45-64
    const int64_t ws_floats = rwk_fused_backward_workspace(m->n, m->d, m->h, m->s, m->max_step);
    rc = rwk_fused_backward_bits(xs, z, m->gid, nullptr, d_k, m->counts_f32, m->n, m->d, m->h, m->s, m->max_step,
                                 plan->stream_len_b, m->drop_p, m->normalize, rng_layer, dx, dz_part,
                                 ws_floats ? at<float>(backward_ws, ob[B_WS]) : nullptr, nullptr, stream);
    if (rc) return rc;
    // first level of the dZ partial sum: independent of the propagation chain, it rides in its first launch
    bool job = plan->n_partials > 32;
    const float *dz = job ? at<float>(backward_ws, ob[B_DZ32]) : dz_part;
    float *dz32 = job ? at<float>(backward_ws, ob[B_DZ32]) : nullptr;
    const bool compact = m->col16_t != nullptr;
    const float *val_tile = (compact && m->unit_values_t) ? nullptr : m->val_t;
    for (int64_t i = m->max_step - 1; i > 1; --i) {                      // dX[i-1] += A^T dX[i] (model.py:55)
        if (job) {
            rc = rwk_spmm_tile_reduce_rows(m->crow_t, m->col_t, m->col16_t, val_tile, m->n, m->d, dx + i * nd,
                                           dx + (i - 1) * nd, 1, dz_part, plan->n_partials, zv, dz32, stream);
            job = false;
        } else if (compact) {
            rc = rwk_spmm_csr_compact(m->crow_t, m->col16_t, m->unit_values_t ? nullptr : m->val_t, m->n, m->d,
                                      dx + i * nd, dx + (i - 1) * nd, 1, stream);
        } else {
            rc = rwk_spmm_csr(m->crow_t, m->col_t, m->val_t, m->n, m->d, dx + i * nd, dx + (i - 1) * nd, 1, 0, nullptr,
                              0, m->long_threshold_t, stream);
        }
        if (rc) return rc;
    }
    // the last transposed step, the projection's backward (model.py:43) and, when still due, the dZ level
    rc = rwk_spmm_project_backward(dx, dx + nd, xs, m->features, m->fc_w, m->n, m->f, m->crow_t, m->col_t, m->col16_t,
                                   val_tile, spb, nullptr, job ? dz_part : nullptr, job ? plan->n_partials : 0,
                                   job ? zv : 0, job ? dz32 : nullptr, stream);
    if (rc) return rc;
    // hidden graphs (model.py:34-37,56) and the [dW | db] sum
    return rwk_hidden_backward_reduce(m->adj_hidden, ah, z, dz, plan->dz_rows, m->h, m->s, m->d, m->max_step, d_adj,
                                      d_feat, spb, plan->spb_blocks, wb_n, wb, stream);
}

// gather: the batch is not assembled yet - the first launch reads the data set and writes the batch's arrays
static int train_step_impl(const rwk_model_step_t *m, const rwk_model_plan_t *plan, const rwk_train_tail_t *tail,
                           void *forward_ws, void *backward_ws, const int64_t *labels, const float *d_loss,
                           float *out, float *loss, float *grads, rwk_stream_t stream,
                           const rwk_x_chain_gather_t *gather) {
    int rc = check_step(m, plan);
    if (rc) return rc;
    if (!tail || !forward_ws || !backward_ws || !labels || !d_loss || !out || !loss || !grads || !tail->grad_out)
        return RWK_E_NULL;
    if (tail->param && (!tail->exp_avg || !tail->exp_avg_sq || !tail->lr || !tail->step_counters)) return RWK_E_NULL;
    if (reinterpret_cast<uintptr_t>(forward_ws) % kAlign || reinterpret_cast<uintptr_t>(backward_ws) % kAlign)
        return RWK_E_ALIGN;
    if (!plan->train_step || tail->n_rest != 6) return RWK_E_UNSUPPORTED;
    const int64_t *of = plan->off_f, *ob = plan->off_b;
    const bool drop = m->drop_p > 0.f;
    const int64_t c = m->h * m->max_step, nd = m->n * m->d;
    const int64_t zv = m->max_step * m->h * m->s * m->d, wb_n = m->d * m->f + m->d;
    if (tail->off_red + wb_n != tail->rest_off) return RWK_E_SIZE;
    float *xs = at<float>(forward_ws, of[F_XS]), *z = at<float>(forward_ws, of[F_Z]);
    float *ah = at<float>(forward_ws, of[F_AH]), *k = at<float>(forward_ws, of[F_K]);
    float *stats = at<float>(forward_ws, of[F_STATS]);
    float *mean = at<float>(forward_ws, of[F_MEAN]), *invstd = at<float>(forward_ws, of[F_INVSTD]);
    uint64_t *rng_layer = drop ? at<uint64_t>(forward_ws, of[F_RNG_LAYER]) : nullptr;
    uint64_t *rng_head = drop ? at<uint64_t>(forward_ws, of[F_RNG_HEAD]) : nullptr;
    float *dx = at<float>(backward_ws, ob[B_DX]), *dz_part = at<float>(backward_ws, ob[B_DZ_PART]);
    float *spb = at<float>(backward_ws, ob[B_SPB]), *dy = at<float>(backward_ws, ob[B_DY]);
    float *head_part = at<float>(backward_ws, ob[B_HEAD_PART]);
    float *loss_part = at<float>(forward_ws, of[F_H1D]);     // (the fc1 output is not kept in this shape: its room)
    float *d_k = at<float>(backward_ws, ob[B_DK]);
    const bool fold = plan->train_step == 1;
    float *d_adj = grads, *d_feat = grads + m->h * pairs(m->s);
    float *wb = d_feat + m->h * m->s * m->d, *head_grads = wb + wb_n;
    // forward (model.py:32-72) with the loss sums and the tile part of the head's backward (main.py:85)
    if (gather)
        rc = rwk_x_chain_hidden_forward_gather(gather, m->fc_w, m->fc_b, m->n, m->f, m->n_graphs, m->max_graph_nodes,
                                               m->max_step, xs, m->adj_hidden, m->features_hidden, m->h, m->s, ah, z,
                                               drop ? m->rng_state : nullptr, rng_layer, stream);
    else
        rc = rwk_x_chain_hidden_forward(m->features, m->fc_w, m->fc_b, m->n, m->f, m->crow, m->col, m->val, m->seg_ptr,
                                        m->n_graphs, m->max_graph_nodes, m->max_step, xs, m->adj_hidden,
                                        m->features_hidden, m->h, m->s, ah, z, drop ? m->rng_state : nullptr, rng_layer,
                                        stream);
    if (rc) return rc;
    rc = rwk_fused_forward_bits(xs, z, m->gid, nullptr, m->n, m->d, m->h, m->s, m->max_step, plan->stream_len_f,
                                m->drop_p, rng_layer, at<float>(forward_ws, of[F_KPART]), nullptr, stream);
    if (rc) return rc;
    rc = rwk_fused_reduce_stats(at<float>(forward_ws, of[F_KPART]), m->seg_ptr, m->counts_f32, m->n_graphs, m->h,
                                m->max_step, plan->stream_len_f, m->drop_p, m->normalize, k, stats,
                                drop ? m->rng_state : nullptr, rng_head, stream);
    if (rc) return rc;
    rc = rwk_head_train_forward(k, m->n_graphs, c, m->p, m->nc, m->bn_w, m->bn_b, m->running_mean, m->running_var,
                                m->num_batches_tracked, m->momentum, m->eps, stats, plan->stats_blocks, m->w1, m->b1,
                                m->w2, m->b2, m->drop_p, nullptr, rng_head, mean, invstd, out, labels, loss_part, dy,
                                head_part, loss, stream);
    if (rc) return rc;
    if (fold) {
        // backward: the head's batch-norm backward rides in the fused kernel's prologue
        const int64_t o_gamma = m->nc * m->p + m->nc + m->p * c + m->p;
        rc = rwk_fused_backward_bn(xs, z, m->gid, nullptr, m->counts_f32, m->n, m->d, m->h, m->s, m->max_step,
                                   plan->stream_len_b, m->drop_p, m->normalize, rng_layer, dx, dz_part, k, dy, mean,
                                   invstd, m->bn_w, head_part, plan->head_blocks, plan->head_grad_floats, o_gamma,
                                   d_loss, m->n_graphs, stream);
    } else {
        // the head finishes in its own launch (dK, its parameter gradients, the loss value), then the layer
        rc = rwk_head_train_backward(k, m->n_graphs, c, m->p, m->nc, m->bn_w, mean, invstd, dy, head_part, d_loss,
                                     loss_part, loss, head_grads, d_k, stream);
        if (rc) return rc;
        const int64_t ws_floats = rwk_fused_backward_workspace(m->n, m->d, m->h, m->s, m->max_step);
        rc = rwk_fused_backward_bits(xs, z, m->gid, nullptr, d_k, m->counts_f32, m->n, m->d, m->h, m->s, m->max_step,
                                     plan->stream_len_b, m->drop_p, m->normalize, rng_layer, dx, dz_part,
                                     ws_floats ? at<float>(backward_ws, ob[B_WS]) : nullptr, nullptr, stream);
    }
    if (rc) return rc;
    bool job = plan->n_partials > 32;
    const float *dz = job ? at<float>(backward_ws, ob[B_DZ32]) : dz_part;
    float *dz32 = job ? at<float>(backward_ws, ob[B_DZ32]) : nullptr;
    const bool compact = m->col16_t != nullptr;
    const float *val_tile = (compact && m->unit_values_t) ? nullptr : m->val_t;
    for (int64_t i = m->max_step - 1; i > 1; --i) {
        if (job) {
            rc = rwk_spmm_tile_reduce_rows(m->crow_t, m->col_t, m->col16_t, val_tile, m->n, m->d, dx + i * nd,
                                           dx + (i - 1) * nd, 1, dz_part, plan->n_partials, zv, dz32, stream);
            job = false;
        } else if (compact) {
            rc = rwk_spmm_csr_compact(m->crow_t, m->col16_t, m->unit_values_t ? nullptr : m->val_t, m->n, m->d,
                                      dx + i * nd, dx + (i - 1) * nd, 1, stream);
        } else {
            rc = rwk_spmm_csr(m->crow_t, m->col_t, m->val_t, m->n, m->d, dx + i * nd, dx + (i - 1) * nd, 1, 0, nullptr,
                              0, m->long_threshold_t, stream);
        }
        if (rc) return rc;
    }
    rc = rwk_spmm_project_backward(dx, dx + nd, xs, m->features, m->fc_w, m->n, m->f, m->crow_t, m->col_t, m->col16_t,
                                   val_tile, spb, nullptr, job ? dz_part : nullptr, job ? plan->n_partials : 0,
                                   job ? zv : 0, job ? dz32 : nullptr, stream);
    if (rc) return rc;
    if (!fold) {
        // hidden graphs, the [dW | db] sum and the update; the head's gradients are in memory already
        const float *srcs[8];
        for (int64_t i = 0; i < tail->n_rest; ++i) srcs[i] = head_grads + tail->rest_columns[i];
        return rwk_hidden_backward_reduce_adam(
            m->adj_hidden, ah, z, dz, plan->dz_rows, m->h, m->s, m->d, m->max_step, d_adj, d_feat, spb,
            plan->spb_blocks, wb_n, wb, tail->param, tail->exp_avg, tail->exp_avg_sq, tail->grad_out, tail->n,
            tail->off_adj, tail->off_feat, tail->off_red, srcs, tail->rest_sizes, tail->n_rest, tail->rest_off, tail->lr,
            tail->beta1, tail->beta2, tail->eps, tail->weight_decay, tail->step_counters, tail->corr_slot, stream);
    }
    // hidden graphs, the [dW | db] sum, the head's parameter-gradient sums, the loss value and the update
    return rwk_hidden_backward_reduce_adam_head(
        m->adj_hidden, ah, z, dz, plan->dz_rows, m->h, m->s, m->d, m->max_step, d_adj, d_feat, spb, plan->spb_blocks,
        wb_n, wb, tail->param, tail->exp_avg, tail->exp_avg_sq, tail->grad_out, tail->n, tail->off_adj, tail->off_feat,
        tail->off_red, tail->rest_columns, tail->rest_sizes, tail->n_rest, tail->rest_off, tail->lr, tail->beta1,
        tail->beta2, tail->eps, tail->weight_decay, tail->step_counters, tail->corr_slot, head_part, plan->head_blocks,
        plan->head_grad_floats, d_loss, head_grads, loss_part, loss, m->n_graphs, stream);
}

extern "C" int rwk_model_train_step(const rwk_model_step_t *m, const rwk_model_plan_t *plan, const rwk_train_tail_t *tail,
                                    void *forward_ws, void *backward_ws, const int64_t *labels, const float *d_loss,
                                    float *out, float *loss, float *grads, rwk_stream_t stream) {
    return train_step_impl(m, plan, tail, forward_ws, backward_ws, labels, d_loss, out, loss, grads, stream, nullptr);
}

// ---------------------------------------------------------------------------------------------------------------
// A whole pass over batches that are assembled right before their step (the loop of main.py:103-106 over batches
// that are never resident twice): per batch the lean rwk_batch_build on the side stream into one of `depth` slots,
// one batch ahead of the step that reads it, and rwk_model_train_step on the main stream - the calls and the
// stream order a caller would make batch by batch, issued from one host call.
// ---------------------------------------------------------------------------------------------------------------
namespace {

void fill_step(rwk_model_step_t *m, const rwk_pass_dataset_t *ds, const rwk_pass_batch_t &bt, const rwk_pass_slot_t &sl) {
    const bool unit = ds->unit_values != 0;
    const bool compact = bt.fits16 && sl.col16 != nullptr;
    m->n = bt.n; m->f = ds->n_feat; m->nnz = bt.nnz; m->n_graphs = bt.b; m->max_graph_nodes = bt.max_graph_nodes;
    m->max_degree = m->max_degree_t = bt.max_degree;
    m->features = sl.feats;
    m->crow = m->crow_t = sl.crow;
    m->col = m->col_t = sl.col;
    m->val = unit ? nullptr : sl.val;
    m->col16_t = compact ? sl.col16 : nullptr;
    m->val_t = (compact && unit) ? nullptr : sl.val;      // a lean assembly leaves that value array unwritten
    m->unit_values_t = unit ? 1 : 0;
    m->long_threshold_t = bt.long_threshold;
    m->seg_ptr = sl.seg_ptr; m->gid = sl.gid; m->counts_f32 = sl.counts_f32;
}

bool pass_serves(const rwk_pass_batch_t &bt) { return bt.b >= 2 && bt.n > 0 && bt.nnz > 0 && bt.max_degree <= 64; }

int build_lean(const rwk_pass_dataset_t *ds, const int64_t *tables, const rwk_pass_batch_t &bt,
               const rwk_pass_slot_t &sl, rwk_stream_t stream) {
    const bool unit = ds->unit_values != 0;
    const bool compact = bt.fits16 && sl.col16 != nullptr;
    return rwk_batch_build(ds->row_ptr, ds->col, ds->val, ds->features, ds->n_feat, ds->y, tables + bt.table_off, bt.b,
                           bt.n, bt.nnz, nullptr, nullptr, nullptr, ds->n_feat ? sl.feats : nullptr, nullptr, sl.y,
                           sl.crow, sl.col, (compact && unit) ? nullptr : sl.val, compact ? sl.col16 : nullptr, sl.gid,
                           sl.seg_ptr, nullptr, sl.counts_f32, stream);
}

}  // namespace

extern "C" int rwk_model_train_pass_plan(const rwk_pass_dataset_t *ds, const rwk_pass_batch_t *batches,
                                         int64_t n_batches, const rwk_pass_slot_t *slots, int64_t depth,
                                         const rwk_model_step_t *model, rwk_pass_need_t *need) {
    if (!ds || !batches || !slots || !model || !need) return RWK_E_NULL;
    if (n_batches < 0 || depth < 1) return RWK_E_SIZE;
    rwk_pass_need_t nd = {};
    for (int64_t t = 0; t < n_batches; ++t) {
        if (!pass_serves(batches[t])) break;
        rwk_model_step_t m = *model;
        fill_step(&m, ds, batches[t], slots[t % depth]);
        rwk_model_plan_t pl;
        if (rwk_model_plan(&m, &pl) != RWK_OK || !pl.train_step) break;
        if (t == 0) nd.train_step = pl.train_step;
        else if (pl.train_step != nd.train_step) break;    // the last launch changes shape: the caller's business
        if (pl.forward_bytes > nd.forward_bytes) nd.forward_bytes = pl.forward_bytes;
        if (pl.backward_bytes > nd.backward_bytes) nd.backward_bytes = pl.backward_bytes;
        if (pl.grad_floats > nd.grad_floats) nd.grad_floats = pl.grad_floats;
        if (m.n_graphs * m.nc > nd.out_floats) nd.out_floats = m.n_graphs * m.nc;
        nd.served = t + 1;
    }
    *need = nd;
    return RWK_OK;
}

extern "C" int rwk_model_train_pass(const rwk_pass_dataset_t *ds, const int64_t *tables, const rwk_pass_batch_t *batches,
                                    int64_t n_batches, const rwk_pass_slot_t *slots, int64_t depth,
                                    const rwk_model_step_t *model, const rwk_train_tail_t *tail,
                                    const rwk_pass_need_t *room, void *forward_ws, void *backward_ws, float *grads,
                                    float *out, const float *d_loss, float *losses, rwk_stream_t main_,
                                    rwk_stream_t side_, int use_side) {
    if (!ds || !tables || !batches || !slots || !model || !tail || !room || !forward_ws || !backward_ws || !grads ||
        !out || !d_loss || !losses)
        return RWK_E_NULL;
    if (n_batches <= 0 || depth < 1 || depth > 8 || use_side < 0 || use_side > 2) return RWK_E_SIZE;
    const bool in_forward = use_side == RWK_PASS_IN_FORWARD;
    if (in_forward) use_side = RWK_PASS_SAME_STREAM;         // (the batches it does not serve: assembled on `main`)
    hipStream_t main = (hipStream_t)main_, side = (hipStream_t)side_;
    hipEvent_t ready[8] = {}, done[8] = {}, start = nullptr;
    bool done_set[8] = {};
    int rc = RWK_OK;
    auto hip = [&](hipError_t e) { if (e != hipSuccess && rc == RWK_OK) rc = (int)e; return e == hipSuccess; };
    if (use_side) {
        // the side stream starts behind everything queued on the main stream (the previous pass read the slots)
        if (!hip(hipEventCreateWithFlags(&start, hipEventDisableTiming))) return rc;
        for (int64_t k = 0; k < depth && rc == RWK_OK; ++k) {
            hip(hipEventCreateWithFlags(&ready[k], hipEventDisableTiming));
            hip(hipEventCreateWithFlags(&done[k], hipEventDisableTiming));
        }
        if (rc == RWK_OK) hip(hipEventRecord(start, main));
        if (rc == RWK_OK) hip(hipStreamWaitEvent(side, start, 0));
    }
    auto prepare = [&](int64_t t) {
        const int64_t k = t % depth;
        if (use_side) {
            if (done_set[k] && !hip(hipStreamWaitEvent(side, done[k], 0))) return;
            const int r = build_lean(ds, tables, batches[t], slots[k], side_);
            if (r) { if (rc == RWK_OK) rc = r; return; }
            hip(hipEventRecord(ready[k], side));
        } else {
            const int r = build_lean(ds, tables, batches[t], slots[k], main_);
            if (r && rc == RWK_OK) rc = r;
        }
    };
    if (rc == RWK_OK && use_side) prepare(0);
    for (int64_t t = 0; t < n_batches && rc == RWK_OK; ++t) {
        const int64_t k = t % depth;
        if (use_side) {
            if (t + 1 < n_batches) prepare(t + 1);
            if (rc != RWK_OK) break;
            if (!hip(hipStreamWaitEvent(main, ready[k], 0))) break;
        }
        // the assembly inside the forward's first launch: an unweighted batch with compact columns
        const rwk_pass_slot_t &sl = slots[k];
        const bool gather = in_forward && ds->unit_values && batches[t].fits16 && sl.col16 != nullptr &&
                            ds->n_feat >= 1;
        if (!use_side && !gather) {
            prepare(t);
            if (rc != RWK_OK) break;
        }
        rwk_model_step_t m = *model;
        fill_step(&m, ds, batches[t], slots[k]);
        rwk_model_plan_t pl;
        int r = pass_serves(batches[t]) ? rwk_model_plan(&m, &pl) : RWK_E_UNSUPPORTED;
        if (r == RWK_OK && (pl.train_step != room->train_step || pl.forward_bytes > room->forward_bytes ||
                            pl.backward_bytes > room->backward_bytes || pl.grad_floats > room->grad_floats ||
                            m.n_graphs * m.nc > room->out_floats))
            r = RWK_E_SIZE;
        if (r == RWK_OK) {
            const rwk_x_chain_gather_t gt{tables + batches[t].table_off, ds->row_ptr, ds->col, ds->features, ds->y,
                                          sl.feats, sl.y, sl.crow, sl.col16, sl.gid, sl.seg_ptr, sl.counts_f32};
            r = train_step_impl(&m, &pl, tail, forward_ws, backward_ws, sl.y, d_loss, out, losses + t, grads, main_,
                                gather ? &gt : nullptr);
        }
        if (r) { rc = r; break; }
        if (use_side && hip(hipEventRecord(done[k], main))) done_set[k] = true;
    }
    if (use_side) {
        // (an event may be destroyed while work that waits on it is queued: its resources go when that work is done)
        for (int64_t k = 0; k < depth; ++k) {
            if (ready[k]) (void)hipEventDestroy(ready[k]);
            if (done[k]) (void)hipEventDestroy(done[k]);
        }
        if (start) (void)hipEventDestroy(start);
    }
    return rc;
}
