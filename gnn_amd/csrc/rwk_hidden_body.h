// Device bodies of the hidden-graph kernels (one workgroup per hidden graph), shared by the stand-alone kernels
// in rwk_hidden.hip and the merged launches in rwk_merged.hip.
#pragma once

#include "rwk_common.h"

constexpr int kHiddenBlock = 256;

// slices the backward body splits the partial sum over (host side; also fixes its LDS size)
static inline int64_t hidden_backward_slices(int64_t v_total, int64_t n_partials) {
    int64_t n_slices = kHiddenBlock / v_total;
    if (n_slices < 1) n_slices = 1;
    if (n_slices > n_partials) n_slices = n_partials;
    if (n_slices > 16) n_slices = 16;
    return n_slices;
}
static inline size_t hidden_forward_lds(int64_t s, int64_t d) { return (size_t)(s * s + 2 * s * d) * sizeof(float); }
static inline size_t hidden_backward_lds(int64_t s, int64_t d, int64_t max_step, int64_t n_slices) {
    const int64_t v_total = max_step * s * d;
    return (size_t)(2 * s * s + v_total + n_slices * v_total) * sizeof(float);
}

// k-th pair (r, c), r < c, of the strict upper triangle in row-major order.
__device__ inline void triu_pair(int k, int s, int *r_out, int *c_out) {
    int r = 0, rem = k;
    while (rem >= s - 1 - r) {
        rem -= s - 1 - r;
        ++r;
    }
    *r_out = r;
    *c_out = r + 1 + rem;
}

// workgroup `h` of kHiddenBlock threads; lds: s*s + 2*s*d floats
__device__ inline void hidden_forward_body(int h, float *lds, const float *__restrict__ adj_hidden,
                                           const float *__restrict__ feat, int n_hidden, int s, int d, int max_step,
                                           float *__restrict__ ah_out, float *__restrict__ z_out,
                                           unsigned long long *rng_state, unsigned long long *rng_used) {
    constexpr int kBlock = kHiddenBlock;
    float *ah = lds;               // s*s
    float *za = ah + s * s;        // s*d
    float *zb = za + s * d;        // s*d
    const int tid = threadIdx.x;
    const int n_pairs = s * (s - 1) / 2;
    for (int i = tid; i < s * s; i += kBlock) ah[i] = 0.f;
    __syncthreads();
    for (int k = tid; k < n_pairs; k += kBlock) {
        int r, c;
        triu_pair(k, s, &r, &c);
        const float w = fmaxf(adj_hidden[(size_t)h * n_pairs + k], 0.f);
        ah[r * s + c] = w;
        ah[c * s + r] = w;
    }
    for (int i = tid; i < s * d; i += kBlock) {
        const float v = feat[(size_t)h * s * d + i];
        za[i] = v;
        z_out[(size_t)h * s * d + i] = v;
    }
    __syncthreads();
    for (int i = tid; i < s * s; i += kBlock) ah_out[(size_t)h * s * s + i] = ah[i];
    const size_t step_stride = (size_t)n_hidden * s * d;
    for (int step = 1; step < max_step; ++step) {
        for (int i = tid; i < s * d; i += kBlock) {
            const int r = i / d, col = i % d;
            float acc = 0.f;
            for (int c = 0; c < s; ++c) acc = fmaf(ah[r * s + c], za[c * d + col], acc);
            zb[i] = acc;
            z_out[step * step_stride + (size_t)h * s * d + i] = acc;
        }
        __syncthreads();
        float *t = za; za = zb; zb = t;
    }
    if (h == 0 && tid == 0 && rng_state != nullptr && rng_used != nullptr) {
        const unsigned long long seed = rng_state[0], off = rng_state[1];
        rng_used[0] = seed;
        rng_used[1] = off;
        rng_state[1] = off + 1ull;
    }
}

// What else happens to a finished gradient element in the thread that holds it (nothing by default; the training
// step's last launch applies the optimiser's update there, see rwk_step.hip).
struct NoGradTail {
    __device__ inline void features_hidden(long long /*index*/, float /*g*/) const {}
    __device__ inline void adj_hidden(long long /*index*/, float /*g*/) const {}
};

// workgroup `h` of kHiddenBlock threads; lds: 2*s*s + (1 + n_slices) * max_step*s*d floats
// adj_hidden carries no __restrict__: with an optimiser tail attached the same thread writes the updated parameter
// back to that memory (the tail's parameter buffer is the parameter itself).
template <typename TAIL = NoGradTail>
__device__ inline void hidden_backward_body(int h, float *lds, const float *adj_hidden,
                                            const float *__restrict__ ah_in, const float *__restrict__ z,
                                            const float *__restrict__ dz_partials, int n_partials, int n_hidden,
                                            int s, int d, int max_step, float *__restrict__ d_adj_hidden,
                                            float *__restrict__ d_feat, int n_slices, const TAIL &tail = TAIL()) {
    constexpr int kBlock = kHiddenBlock;
    const int sd = s * d;
    const int v_total = max_step * sd;           // gradient values owned by this hidden graph
    float *ah = lds;                             // s*s
    float *dah = ah + s * s;                     // s*s
    float *dz = dah + s * s;                     // max_step * s*d
    float *slice = dz + v_total;                 // n_slices * v_total
    const int tid = threadIdx.x;
    const size_t step_stride = (size_t)n_hidden * sd;
    const size_t part_stride = (size_t)max_step * step_stride;

    // 1. fixed-order sum of the partials: slice q adds partials q, q+n_slices, ... ; slices are
    //    then added in increasing q.
    for (int w = tid; w < n_slices * v_total; w += kBlock) {
        const int q = w / v_total, v = w % v_total;
        const int step = v / sd, e = v % sd;
        const float *src = dz_partials + step * step_stride + (size_t)h * sd + e;
        float acc = 0.f;
        for (int p0 = q; p0 < n_partials; p0 += 8 * n_slices) {     // eight loads in flight, added in order
            float part[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int p = p0 + u * n_slices;
                part[u] = src[(size_t)(p < n_partials ? p : n_partials - 1) * part_stride];
                if (p >= n_partials) part[u] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += part[u];
        }
        slice[w] = acc;
    }
    for (int i = tid; i < s * s; i += kBlock) {
        ah[i] = ah_in[(size_t)h * s * s + i];
        dah[i] = 0.f;
    }
    __syncthreads();
    for (int v = tid; v < v_total; v += kBlock) {
        float acc = 0.f;
        for (int q = 0; q < n_slices; ++q) acc += slice[q * v_total + v];
        dz[v] = acc;
    }
    __syncthreads();

    // 2. walk the chain back
    for (int step = max_step - 1; step >= 1; --step) {
        const float *zprev = z + (step - 1) * step_stride + (size_t)h * sd;
        const float *dzi = dz + step * sd;
        float *dzp = dz + (step - 1) * sd;
        for (int i = tid; i < s * s; i += kBlock) {
            const int r = i / s, c = i % s;
            float acc = 0.f;
            for (int k = 0; k < d; ++k) acc = fmaf(dzi[r * d + k], zprev[c * d + k], acc);
            dah[i] += acc;
        }
        for (int i = tid; i < sd; i += kBlock) {
            const int c = i / d, k = i % d;
            float acc = 0.f;
            for (int r = 0; r < s; ++r) acc = fmaf(ah[r * s + c], dzi[r * d + k], acc);
            dzp[i] += acc;
        }
        __syncthreads();
    }

    // 3. outputs
    for (int i = tid; i < sd; i += kBlock) {
        d_feat[(size_t)h * sd + i] = dz[i];
        tail.features_hidden((long long)h * sd + i, dz[i]);
    }
    // the entries of adj_hidden are finished by the threads at the top of the workgroup, those of features_hidden by the
    // threads at the bottom: with a tail attached (an Adam update per element: three divisions and a square root in a
    // dependent chain) the two run in different wavefronts, side by side, instead of one after the other
    const int n_pairs = s * (s - 1) / 2;
    for (int k = kBlock - 1 - tid; k < n_pairs; k += kBlock) {
        int r, c;
        triu_pair(k, s, &r, &c);
        const float a = adj_hidden[(size_t)h * n_pairs + k];
        const float g = a > 0.f ? dah[r * s + c] + dah[c * s + r] : 0.f;
        d_adj_hidden[(size_t)h * n_pairs + k] = g;
        tail.adj_hidden((long long)h * n_pairs + k, g);
    }
}

