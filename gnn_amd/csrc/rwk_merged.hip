// Two independent small kernels in one launch.  The hidden-graph kernels (one workgroup per hidden graph, a few
// microseconds of latency-bound work) do not depend on the input projection that runs next to them in the forward
// pass, nor on its backward; launched one after the other each pays its own launch and drain.  Here the first
// n_hidden workgroups of a launch run the hidden-graph body and the rest the projection body; both bodies are the
// ones the stand-alone kernels use, so the results are the same bits.
#include "rwk_hidden_body.h"
#include "rwk_project_body.h"

namespace {

constexpr int kBlock = 256;
static_assert(kHiddenBlock == kBlock && kProjectBlock == kBlock, "the merged launches use one workgroup size");
constexpr size_t kLdsLimit = 160 * 1024;

template <int FP>
__global__ void __launch_bounds__(kBlock)
project_hidden_forward_kernel(const float *__restrict__ adj_hidden, const float *__restrict__ feat_hidden, int n_hidden,
                              int s, int d, int max_step, float *__restrict__ ah_out, float *__restrict__ z_out,
                              unsigned long long *rng_state, unsigned long long *rng_used,
                              const float *__restrict__ feat, const float *__restrict__ w, const float *__restrict__ b,
                              int n, int f, float *__restrict__ x0) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if ((int)blockIdx.x < n_hidden)
        hidden_forward_body(blockIdx.x, lds, adj_hidden, feat_hidden, n_hidden, s, d, max_step, ah_out, z_out, rng_state,
                            rng_used);
    else {
        // XCD k takes the k-th contiguous eighth of the node blocks, as the tile SpMM and the fused kernels that read
        // these rows do (workgroups are dealt round-robin to the 8 XCDs, each with its own L2)
        const int nb = (int)gridDim.x - n_hidden, per_xcd = nb / 8;
        int blk = (int)blockIdx.x - n_hidden;
        if (blk < per_xcd * 8) blk = ((int)blockIdx.x % 8) * per_xcd + blk / 8;
        project_forward4_body<FP>(blk, feat, w, b, n, f, x0);
    }
}

template <int FP>
__global__ void __launch_bounds__(kBlock)
project_hidden_backward_kernel(const float *__restrict__ adj_hidden, const float *__restrict__ ah_in,
                               const float *__restrict__ z, const float *__restrict__ dz_partials, int n_partials,
                               int n_hidden, int s, int d, int max_step, float *__restrict__ d_adj_hidden,
                               float *__restrict__ d_feat_hidden, int n_slices, const float *__restrict__ dx0,
                               const float *__restrict__ x0, const float *__restrict__ feat,
                               const float *__restrict__ w, long long n, int f, int dpad, int proj_blocks,
                               float *__restrict__ partials, float *__restrict__ d_feat) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if ((int)blockIdx.x < n_hidden)
        hidden_backward_body(blockIdx.x, lds, adj_hidden, ah_in, z, dz_partials, n_partials, n_hidden, s, d, max_step,
                             d_adj_hidden, d_feat_hidden, n_slices);
    else
        project_backward_body<FP>((int)blockIdx.x - n_hidden, proj_blocks, lds, dx0, x0, feat, w, n, f, d, dpad,
                                  partials, d_feat);
}

template <typename K>
int allow_lds(K kern, size_t lds) {
    if (lds > kLdsLimit) return RWK_E_UNSUPPORTED;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    return RWK_OK;
}

}  // namespace

extern "C" {

int rwk_project_hidden_forward(const float *features, const float *weight, const float *bias, int64_t n, int64_t f,
                               int64_t d, float *x0, const float *adj_hidden, const float *features_hidden, int64_t h,
                               int64_t s, int64_t max_step, float *ah, float *z, uint64_t *rng_state,
                               uint64_t *rng_used, rwk_stream_t stream_) {
    RWK_REQUIRE_PTR(features);
    RWK_REQUIRE_PTR(weight);
    RWK_REQUIRE_PTR(bias);
    RWK_REQUIRE_PTR(x0);
    RWK_REQUIRE_PTR(features_hidden);
    RWK_REQUIRE_PTR(ah);
    RWK_REQUIRE_PTR(z);
    if (n <= 0 || f <= 0 || d <= 0 || n >= 0x7FFFFFFFll / 4 || h <= 0 || s <= 0 || max_step <= 0 || h > 65535)
        return RWK_E_SIZE;
    if (s > 1) RWK_REQUIRE_PTR(adj_hidden);
    if (d != 4 || f > 16 || ((uintptr_t)x0 & 15u) != 0) return RWK_E_UNSUPPORTED;
    const size_t lds = hidden_forward_lds(s, d);
    const dim3 grid((unsigned)(h + rwk_ceil_div(n, kBlock)));
    hipStream_t stream = (hipStream_t)stream_;
#define RWK_PHF(FP)                                                                                               \
    do {                                                                                                          \
        const int rc = allow_lds(project_hidden_forward_kernel<FP>, lds);                                         \
        if (rc != RWK_OK) return rc;                                                                              \
        hipLaunchKernelGGL((project_hidden_forward_kernel<FP>), grid, dim3(kBlock), lds, stream, adj_hidden,       \
                           features_hidden, (int)h, (int)s, (int)d, (int)max_step, ah, z,                          \
                           (unsigned long long *)rng_state, (unsigned long long *)rng_used, features, weight, bias, \
                           (int)n, (int)f, x0);                                                                    \
    } while (0)
    if (f == 1) RWK_PHF(1);
    else if (f == 2) RWK_PHF(2);
    else if (f <= 4) RWK_PHF(4);
    else if (f <= 8) RWK_PHF(8);
    else RWK_PHF(16);
#undef RWK_PHF
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

int rwk_project_hidden_backward(const float *dx0, const float *x0, const float *features, const float *weight,
                                int64_t n, int64_t f, int64_t d, float *partials, float *d_features,
                                const float *adj_hidden, const float *ah, const float *z, const float *dz_partials,
                                int64_t n_partials, int64_t h, int64_t s, int64_t max_step, float *d_adj_hidden,
                                float *d_features_hidden, rwk_stream_t stream_) {
    RWK_REQUIRE_PTR(dx0);
    RWK_REQUIRE_PTR(x0);
    RWK_REQUIRE_PTR(features);
    RWK_REQUIRE_PTR(weight);
    RWK_REQUIRE_PTR(partials);
    RWK_REQUIRE_PTR(ah);
    RWK_REQUIRE_PTR(z);
    RWK_REQUIRE_PTR(dz_partials);
    RWK_REQUIRE_PTR(d_features_hidden);
    if (n <= 0 || f <= 0 || d <= 0 || h <= 0 || s <= 0 || max_step <= 0 || n_partials <= 0 || h > 65535)
        return RWK_E_SIZE;
    if (s > 1) {
        RWK_REQUIRE_PTR(adj_hidden);
        RWK_REQUIRE_PTR(d_adj_hidden);
    }
    if (f > 16 || d > 64) return RWK_E_UNSUPPORTED;
    const int dpad = project_pow2_ceil((int)d);
    const int64_t proj_blocks = rwk_project_backward_blocks(n, d);
    const int64_t n_slices = hidden_backward_slices(max_step * s * d, n_partials);
    const size_t lds_h = hidden_backward_lds(s, d, max_step, n_slices);
    const dim3 grid((unsigned)(h + proj_blocks));
    hipStream_t stream = (hipStream_t)stream_;
#define RWK_PHB(FP)                                                                                                \
    do {                                                                                                           \
        const size_t lds_p = (size_t)(FP + 1) * kBlock * sizeof(float);                                            \
        const size_t lds = lds_h > lds_p ? lds_h : lds_p;                                                          \
        const int rc = allow_lds(project_hidden_backward_kernel<FP>, lds);                                         \
        if (rc != RWK_OK) return rc;                                                                               \
        hipLaunchKernelGGL((project_hidden_backward_kernel<FP>), grid, dim3(kBlock), lds, stream, adj_hidden, ah, z, \
                           dz_partials, (int)n_partials, (int)h, (int)s, (int)d, (int)max_step, d_adj_hidden,       \
                           d_features_hidden, (int)n_slices, dx0, x0, features, weight, (long long)n, (int)f, dpad, \
                           (int)proj_blocks, partials, d_features);                                                 \
    } while (0)
    if (f == 1) RWK_PHB(1);
    else if (f == 2) RWK_PHB(2);
    else if (f <= 4) RWK_PHB(4);
    else if (f <= 8) RWK_PHB(8);
    else RWK_PHB(16);
#undef RWK_PHB
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

}  // extern "C"
