// The hidden-graph backward (autograd of rwgnn2/model.py:39-42,56) and the final sum of the projection's
// per-workgroup gradient rows in one launch: both are a few microseconds of latency-bound work at the end of the
// backward pass, neither depends on the other, and launched one after the other each pays its own launch, ramp
// and drain.  (The launches that do the first propagation step inside the projection's kernels are in rwk_spmm.hip,
// next to the tile kernel they are built from.)
#include "rwk_hidden_body.h"
#include "rwk_project_body.h"
#include "rwk_reduce_body.h"

namespace {

constexpr int kBlock = 256;
static_assert(kHiddenBlock == kBlock && kProjectBlock == kBlock, "the merged launches use one workgroup size");
constexpr size_t kLdsLimit = 160 * 1024;
// Column sums of a (p, v) matrix with v <= 64 by one workgroup of 256 threads: 256 / vp lanes per column (vp = v
// rounded up to a power of two), lane l adds rows l, l + lanes, ... in order (four loads in flight), then the lanes of
// a column are added in a fixed halving tree.  lds: 256 floats.
__device__ inline void reduce_rows_small256_body(float *red, const float *__restrict__ in, int p, int v, int vp,
                                                 float *__restrict__ out) {
    const int tid = threadIdx.x;
    const int colv = tid % vp, lane = tid / vp, lanes = kBlock / vp;
    const int cc = colv < v ? colv : v - 1;
    float acc = 0.f;
    for (int r0 = lane; r0 < p; r0 += 4 * lanes) {
        float a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + u * lanes;
            a[u] = in[(size_t)(r < p ? r : p - 1) * v + cc];
            if (r >= p) a[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += a[u];
    }
    red[tid] = acc;
    __syncthreads();
    for (int off = lanes / 2; off >= 1; off >>= 1) {
        if (lane < off) red[tid] += red[tid + off * vp];
        __syncthreads();
    }
    if (lane == 0 && colv < v) out[colv] = red[colv];
}

__global__ void __launch_bounds__(kBlock)
hidden_backward_reduce_kernel(const float *__restrict__ adj_hidden, const float *__restrict__ ah_in,
                              const float *__restrict__ z, const float *__restrict__ dz_partials, int n_partials,
                              int n_hidden, int s, int d, int max_step, float *__restrict__ d_adj_hidden,
                              float *__restrict__ d_feat_hidden, int n_slices, const float *__restrict__ red_in,
                              int red_p, int red_v, int red_vp, float *__restrict__ red_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if ((int)blockIdx.x < n_hidden)
        hidden_backward_body(blockIdx.x, lds, adj_hidden, ah_in, z, dz_partials, n_partials, n_hidden, s, d, max_step,
                             d_adj_hidden, d_feat_hidden, n_slices);
    else
        reduce_rows_small256_body(lds, red_in, red_p, red_v, red_vp, red_out);
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

int rwk_hidden_backward_reduce(const float *adj_hidden, const float *ah, const float *z, const float *dz_partials,
                               int64_t n_partials, int64_t h, int64_t s, int64_t d, int64_t max_step,
                               float *d_adj_hidden, float *d_features_hidden, const float *red_in, int64_t red_p,
                               int64_t red_v, float *red_out, rwk_stream_t stream_) {
    RWK_REQUIRE_PTR(ah);
    RWK_REQUIRE_PTR(z);
    RWK_REQUIRE_PTR(dz_partials);
    RWK_REQUIRE_PTR(d_features_hidden);
    RWK_REQUIRE_PTR(red_in);
    RWK_REQUIRE_PTR(red_out);
    if (h <= 0 || s <= 0 || d <= 0 || max_step <= 0 || n_partials <= 0 || h > 65535) return RWK_E_SIZE;
    if (red_p <= 0 || red_p > (1 << 20) || red_v <= 0) return RWK_E_SIZE;
    if (red_v > 64) return RWK_E_UNSUPPORTED;
    if (s > 1) {
        RWK_REQUIRE_PTR(adj_hidden);
        RWK_REQUIRE_PTR(d_adj_hidden);
    }
    const int64_t n_slices = hidden_backward_slices(max_step * s * d, n_partials);
    size_t lds = hidden_backward_lds(s, d, max_step, n_slices);
    if (lds < kBlock * sizeof(float)) lds = kBlock * sizeof(float);
    const int rc = allow_lds(hidden_backward_reduce_kernel, lds);
    if (rc != RWK_OK) return rc;
    hipLaunchKernelGGL(hidden_backward_reduce_kernel, dim3((unsigned)(h + 1)), dim3(kBlock), lds, (hipStream_t)stream_,
                       adj_hidden, ah, z, dz_partials, (int)n_partials, (int)h, (int)s, (int)d, (int)max_step,
                       d_adj_hidden, d_features_hidden, (int)n_slices, red_in, (int)red_p, (int)red_v,
                       project_pow2_ceil((int)red_v), red_out);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

}  // extern "C"
