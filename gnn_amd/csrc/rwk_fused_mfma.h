// Matrix-core (fp32 MFMA) variant of the fused random-walk kernels for hidden_dim > 4, where the
// contraction over D is a real dense product (e.g. 64 hidden graphs x 10 nodes, hidden_dim 64).
#pragma once

#include "rwk_common.h"

namespace rwk_mfma {

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
    float *dx;                     // (L,N,D)
    float *dx_partials;            // (ceil(H/16), L, N, D) workspace, needed when H > 16
    float *dz_partials;            // (backward_blocks, L, H, S, D)
    // dropout keep bits per (node, hidden graph), N x H words of 32 bits (S * L <= 32) or 64 bits: written by the
    // forward, read by the backward
    // (NULL: the forward does not leave them / the backward runs the generator itself)
    unsigned long long *keep;
};

bool supported(int64_t d, int64_t h, int64_t s, int64_t max_step);
// workgroup columns of the backward launch = rows of dz_partials
int64_t backward_blocks(int64_t n, int64_t stream_len);
// 64-bit words of Args::keep when the backward for these sizes can take the forward's keep bits, else 0
int64_t keep_words(int64_t n, int64_t d, int64_t h, int64_t s, int64_t max_step);
// floats of dx_partials the backward needs (0 when H <= 16)
int64_t dx_partials_size(int64_t n, int64_t d, int64_t h, int64_t max_step);
// positions per stream that fill the chip in whole rounds of resident workgroups
int64_t stream_len(int64_t n, int64_t h, bool backward);
int forward(const Args &a);
int backward(const Args &a);

}  // namespace rwk_mfma
