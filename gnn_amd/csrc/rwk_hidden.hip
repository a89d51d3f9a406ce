// Hidden-graph kernels: one workgroup per hidden graph, the whole graph (adjacency S x S and
// the S x D feature slab of every step) stays in LDS for all max_step steps.
//
//   forward : Ah = sym(triu_1(relu(adj_hidden))),  Z[i] = Ah Z[i-1]          (one launch)
//   backward: sums the per-workgroup partial gradients of Z[i] written by the fused backward
//             kernel in a fixed order, then walks the chain back:
//             dAh += dZ[i] Z[i-1]^T,  dZ[i-1] += Ah^T dZ[i];  d_adj_hidden = (dAh+dAh^T)[triu] * 1[adj_hidden > 0]
#include "rwk_common.h"

namespace {

constexpr int kBlock = 256;

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

__global__ void __launch_bounds__(kBlock)
hidden_forward_kernel(const float *__restrict__ adj_hidden, const float *__restrict__ feat, int n_hidden, int s,
                      int d, int max_step, float *__restrict__ ah_out, float *__restrict__ z_out,
                      unsigned long long *rng_state, unsigned long long *rng_used) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *ah = lds;               // s*s
    float *za = ah + s * s;        // s*d
    float *zb = za + s * d;        // s*d
    const int h = blockIdx.x;
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

__global__ void __launch_bounds__(kBlock)
hidden_backward_kernel(const float *__restrict__ adj_hidden, const float *__restrict__ ah_in,
                       const float *__restrict__ z, const float *__restrict__ dz_partials, int n_partials,
                       int n_hidden, int s, int d, int max_step, float *__restrict__ d_adj_hidden,
                       float *__restrict__ d_feat, int n_slices) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int sd = s * d;
    const int v_total = max_step * sd;           // gradient values owned by this hidden graph
    float *ah = lds;                             // s*s
    float *dah = ah + s * s;                     // s*s
    float *dz = dah + s * s;                     // max_step * s*d
    float *slice = dz + v_total;                 // n_slices * v_total
    const int h = blockIdx.x;
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
    for (int i = tid; i < sd; i += kBlock) d_feat[(size_t)h * sd + i] = dz[i];
    const int n_pairs = s * (s - 1) / 2;
    for (int k = tid; k < n_pairs; k += kBlock) {
        int r, c;
        triu_pair(k, s, &r, &c);
        const float a = adj_hidden[(size_t)h * n_pairs + k];
        d_adj_hidden[(size_t)h * n_pairs + k] = a > 0.f ? dah[r * s + c] + dah[c * s + r] : 0.f;
    }
}

constexpr size_t kLdsLimit = 160 * 1024;

}  // namespace

extern "C" {

int rwk_hidden_forward(const float *adj_hidden, const float *features_hidden, int64_t h, int64_t s, int64_t d,
                       int64_t max_step, float *ah, float *z, uint64_t *rng_state, uint64_t *rng_used,
                       rwk_stream_t stream) {
    RWK_REQUIRE_PTR(features_hidden);
    RWK_REQUIRE_PTR(ah);
    RWK_REQUIRE_PTR(z);
    if (h <= 0 || s <= 0 || d <= 0 || max_step <= 0 || h > 65535) return RWK_E_SIZE;
    if (s > 1) RWK_REQUIRE_PTR(adj_hidden);
    const size_t lds = (size_t)(s * s + 2 * s * d) * sizeof(float);
    if (lds > kLdsLimit) return RWK_E_UNSUPPORTED;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)hidden_forward_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(hidden_forward_kernel, dim3((unsigned)h), dim3(kBlock), lds, (hipStream_t)stream, adj_hidden,
                       features_hidden, (int)h, (int)s, (int)d, (int)max_step, ah, z,
                       (unsigned long long *)rng_state, (unsigned long long *)rng_used);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

int rwk_hidden_backward(const float *adj_hidden, const float *ah, const float *z, const float *dz_partials,
                        int64_t n_partials, int64_t h, int64_t s, int64_t d, int64_t max_step,
                        float *d_adj_hidden, float *d_features_hidden, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(ah);
    RWK_REQUIRE_PTR(z);
    RWK_REQUIRE_PTR(dz_partials);
    RWK_REQUIRE_PTR(d_features_hidden);
    if (h <= 0 || s <= 0 || d <= 0 || max_step <= 0 || n_partials <= 0 || h > 65535) return RWK_E_SIZE;
    if (s > 1) {
        RWK_REQUIRE_PTR(adj_hidden);
        RWK_REQUIRE_PTR(d_adj_hidden);
    }
    const int64_t v_total = max_step * s * d;
    int64_t n_slices = kBlock / v_total;
    if (n_slices < 1) n_slices = 1;
    if (n_slices > n_partials) n_slices = n_partials;
    if (n_slices > 16) n_slices = 16;
    const size_t lds = (size_t)(2 * s * s + v_total + n_slices * v_total) * sizeof(float);
    if (lds > kLdsLimit) return RWK_E_UNSUPPORTED;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)hidden_backward_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(hidden_backward_kernel, dim3((unsigned)h), dim3(kBlock), lds, (hipStream_t)stream, adj_hidden,
                       ah, z, dz_partials, (int)n_partials, (int)h, (int)s, (int)d, (int)max_step, d_adj_hidden,
                       d_features_hidden, (int)n_slices);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

}  // extern "C"
