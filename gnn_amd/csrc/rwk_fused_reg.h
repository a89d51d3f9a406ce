// Register-resident variant of the fused random-walk kernels (hidden_dim <= 4, common hidden-graph sizes).
#pragma once

#include "rwk_common.h"

namespace rwk_reg {

// The head's batch-norm backward folded into the backward kernel (partials == nullptr: off, d_k is read instead).
// The kernel then forms d(loss)/d(K) of the graphs it walks itself: every workgroup adds the head's per-workgroup
// partial rows of the gamma / beta columns (n_partials rows of nv floats, the 2c sums start at column o_ga) and a
// lane computes its own elements from dy (gradient at the batch-norm's output for d(loss) = 1), K and the
// batch statistics.  c = max_step * h columns, 2c <= 64.
struct BnFold {
    const float *k, *dy, *mean, *invstd, *bn_w, *partials, *d_loss;
    int n_partials, nv, o_ga;
    float inv_g;                   // 1 / number of graphs of the batch
};

struct Args {
    const float *x, *z;            // (L,N,D), (L,H,S,D)
    const int *gid, *perm;
    int n, d, h, s, max_step, stream_len;
    uint32_t t0, t1;               // dropout thresholds (rwk_drop_thresholds)
    bool drop;
    const unsigned long long *rng;
    hipStream_t stream;
    // forward
    float *k_partials;
    // backward
    const float *d_k, *counts_f32;
    float scale;
    int normalize;
    float *dx, *dz_partials;
    // features (and their gradients) per hidden node instead of summed over the hidden nodes
    bool keep_s;
    BnFold bn;
};

// true when (d, h, s, max_step) has a compiled register-resident instance
bool supported(int64_t d, int64_t h, int64_t s, int64_t max_step);
// true when the keep-S variant of (d, h, s, max_step) is compiled
bool keep_s_supported(int64_t d, int64_t h, int64_t s, int64_t max_step);
// true when the backward kernel of (d, h, s, max_step) can take the head's batch-norm backward (Args::bn)
bool bn_fold_supported(int64_t d, int64_t h, int64_t s, int64_t max_step);
// workgroups of the backward launch = rows of dz_partials
int64_t backward_blocks(int64_t n, int64_t stream_len, int64_t h);
// positions per stream that fill the chip in whole rounds of resident workgroups
int64_t stream_len(int64_t n, int64_t h, int64_t s, int64_t max_step, bool bwd);
int forward(const Args &a);
int backward(const Args &a);

}  // namespace rwk_reg
