// Hidden-graph kernels: one workgroup per hidden graph, the whole graph (adjacency S x S and
// the S x D feature slab of every step) stays in LDS for all max_step steps.
//
//   forward : Ah = sym(triu_1(relu(adj_hidden))),  Z[i] = Ah Z[i-1]          (one launch)
//   backward: sums the per-workgroup partial gradients of Z[i] written by the fused backward
//             kernel in a fixed order, then walks the chain back:
//             dAh += dZ[i] Z[i-1]^T,  dZ[i-1] += Ah^T dZ[i];  d_adj_hidden = (dAh+dAh^T)[triu] * 1[adj_hidden > 0]
#include "rwk_hidden_body.h"

namespace {

constexpr int kBlock = kHiddenBlock;

__global__ void __launch_bounds__(kBlock)
hidden_forward_kernel(const float *__restrict__ adj_hidden, const float *__restrict__ feat, int n_hidden, int s,
                      int d, int max_step, float *__restrict__ ah_out, float *__restrict__ z_out,
                      unsigned long long *rng_state, unsigned long long *rng_used) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    hidden_forward_body(blockIdx.x, lds, adj_hidden, feat, n_hidden, s, d, max_step, ah_out, z_out, rng_state,
                        rng_used);
}

__global__ void __launch_bounds__(kBlock)
hidden_backward_kernel(const float *__restrict__ adj_hidden, const float *__restrict__ ah_in,
                       const float *__restrict__ z, const float *__restrict__ dz_partials, int n_partials,
                       int n_hidden, int s, int d, int max_step, float *__restrict__ d_adj_hidden,
                       float *__restrict__ d_feat, int n_slices) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    hidden_backward_body(blockIdx.x, lds, adj_hidden, ah_in, z, dz_partials, n_partials, n_hidden, s, d, max_step,
                         d_adj_hidden, d_feat, n_slices);
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
    const size_t lds = hidden_forward_lds(s, d);
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
    const int64_t n_slices = hidden_backward_slices(max_step * s * d, n_partials);
    const size_t lds = hidden_backward_lds(s, d, max_step, n_slices);
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
