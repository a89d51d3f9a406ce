// Pieces of the head's backward pass that more than one kernel uses: the column sums of the per-workgroup partial
// rows the head's training kernel leaves (fixed order: bitwise the same wherever they are formed) and the
// batch-norm input gradient of one element (rwgnn2/model.py:68, backward of nn.BatchNorm1d in training mode).
#pragma once

#include "rwk_common.h"

// The first 128 block partials of one element, fetched ahead of their use: lane8 holds blocks lane8, lane8 + 8, ...
// sum() adds them in rwk_sum_partials8's order (and goes on with any further blocks as it does).  The eight lanes
// of an element are then added with rwk_group_sum(acc, 8).
struct RwkFinPartials {
    float v[16];
    __device__ inline void load(const float *__restrict__ partials, int n_partials, int nv, int e, int lane8) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int b = lane8 + 8 * u;
            v[u] = partials[(size_t)(b < n_partials ? b : n_partials - 1) * nv + e];
        }
    }
    __device__ inline float sum(const float *__restrict__ partials, int n_partials, int nv, int e, int lane8) const {
        float acc = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += lane8 + 8 * u < n_partials ? v[u] : 0.f;
        for (int b0 = lane8 + 128; b0 < n_partials; b0 += 64) {
            float w[8], m[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + 8 * u;
                w[u] = partials[(size_t)(b < n_partials ? b : n_partials - 1) * nv + e];
                m[u] = b < n_partials ? 1.f : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fmaf(w[u], m[u], acc);
        }
        return acc;
    }
};

// partials[b][e] added over b = lane8, lane8 + 8, ... in that order, eight loads in flight at a time.  A block past
// the last one is loaded at the last block's index and enters the sum times 0.0f: with a select on the loaded value
// the compiler branches around each load and waits for each in turn (eight trips to memory instead of one).
__device__ static inline float rwk_sum_partials8(const float *__restrict__ partials, int n_partials, int nv, int e,
                                                 int lane8) {
    float acc = 0.f;
    for (int b0 = lane8; b0 < n_partials; b0 += 64) {
        float v[8], m[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = b0 + 8 * u;
            v[u] = partials[(size_t)(b < n_partials ? b : n_partials - 1) * nv + e];
            m[u] = b < n_partials ? 1.f : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fmaf(v[u], m[u], acc);
    }
    return acc;
}

// d(loss)/d(k) of one element behind the batch-norm: dy is the gradient at the batch-norm's output (for d(loss) = 1),
// dl the loss gradient, dga / dbe the column sums of dy * xhat and dy (already times dl), sc = gamma * invstd.
// Every rounding is spelled out (no contraction), so the kernels that use it agree bit for bit.
__device__ static inline float rwk_bn_input_grad(float dy, float k, float mu, float is, float sc, float dga, float dbe,
                                                 float inv_g, float dl, bool training) {
    float v = __fmul_rn(dy, dl);
    if (training) {
        const float xh = __fmul_rn(__fsub_rn(k, mu), is);
        v = __fsub_rn(__fsub_rn(v, __fmul_rn(dbe, inv_g)), __fmul_rn(__fmul_rn(xh, dga), inv_g));
    }
    return __fmul_rn(v, sc);
}
