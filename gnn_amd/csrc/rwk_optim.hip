// Adam update over one flat parameter buffer: one launch per training step instead of the ~45 small
// launches torch.optim.Adam issues for the 10 parameter tensors of RW_NN (rwgnn2/main.py:79,87).
// The step count lives on the device (one counter per workgroup, so no cross-workgroup ordering is
// needed) and the learning rate is read from device memory, so a captured HIP graph replays correctly
// and a scheduler can change the rate without re-capturing.
#include "rwk_common.h"

namespace {

constexpr int kBlock = 256;

// b^t for an integer step count by repeated squaring: a few dozen double multiplications instead of the
// log/exp evaluation behind pow() (one thread computes it while its workgroup waits)
__device__ inline double pow_int(double b, long long t) {
    double r = 1.0;
    while (t > 0) {
        if (t & 1) r *= b;
        b *= b;
        t >>= 1;
    }
    return r;
}

__global__ void __launch_bounds__(kBlock)
adam_step_kernel(float *__restrict__ param, const float *__restrict__ grad, float *__restrict__ exp_avg,
                 float *__restrict__ exp_avg_sq, long long n, const float *__restrict__ lr_ptr, float beta1,
                 float beta2, float eps, float weight_decay, long long *__restrict__ step_counters) {
    __shared__ float corr[2];
    if (threadIdx.x == 0) {
        const long long t = step_counters[blockIdx.x] + 1;
        step_counters[blockIdx.x] = t;
        corr[0] = (float)(1.0 - pow_int((double)beta1, t));
        corr[1] = (float)sqrt(1.0 - pow_int((double)beta2, t));
    }
    __syncthreads();
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float lr = lr_ptr[0];
    float p = param[i];
    float g = grad[i];
    if (weight_decay != 0.f) g = fmaf(weight_decay, p, g);
    const float m = fmaf(beta1, exp_avg[i], (1.f - beta1) * g);
    const float v = fmaf(beta2, exp_avg_sq[i], (1.f - beta2) * g * g);
    exp_avg[i] = m;
    exp_avg_sq[i] = v;
    const float denom = sqrtf(v) / corr[1] + eps;
    param[i] = p - (lr / corr[0]) * (m / denom);
}

// The same update with the gradient taken straight from up to 16 per-parameter gradient tensors (chunk k covers
// the flat indices [off[k], off[k+1])), so that a single-process step needs no separate gather launch; the flat
// gradient is still written to grad_out when that is not null.
constexpr int kAdamChunks = 16;
struct AdamChunks {
    const float *src[kAdamChunks];
    long long off[kAdamChunks];          // unused entries: LLONG_MAX
};

__global__ void __launch_bounds__(kBlock)
adam_step_chunks_kernel(float *__restrict__ param, AdamChunks a, float *__restrict__ grad_out,
                        float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq, long long n,
                        const float *__restrict__ lr_ptr, float beta1, float beta2, float eps, float weight_decay,
                        long long *__restrict__ step_counters) {
    __shared__ float corr[2];
    if (threadIdx.x == 0) {
        const long long t = step_counters[blockIdx.x] + 1;
        step_counters[blockIdx.x] = t;
        corr[0] = (float)(1.0 - pow_int((double)beta1, t));
        corr[1] = (float)sqrt(1.0 - pow_int((double)beta2, t));
    }
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long ic = i < n ? i : n - 1;
    // compare-select chain instead of indexing the argument table (a dynamic index would move it to scratch)
    const float *src = a.src[0];
    long long base = a.off[0];
#pragma unroll
    for (int k = 1; k < kAdamChunks; ++k) {
        const bool ge = ic >= a.off[k];
        src = ge ? a.src[k] : src;
        base = ge ? a.off[k] : base;
    }
    float g = src ? src[ic - base] : 0.f;
    float p = param[ic];
    const float m0 = exp_avg[ic], v0 = exp_avg_sq[ic];
    const float lr = lr_ptr[0];
    __syncthreads();
    if (i >= n) return;
    if (grad_out != nullptr) grad_out[i] = g;
    if (weight_decay != 0.f) g = fmaf(weight_decay, p, g);
    const float m = fmaf(beta1, m0, (1.f - beta1) * g);
    const float v = fmaf(beta2, v0, (1.f - beta2) * g * g);
    exp_avg[i] = m;
    exp_avg_sq[i] = v;
    const float denom = sqrtf(v) / corr[1] + eps;
    param[i] = p - (lr / corr[0]) * (m / denom);
}

constexpr int kGatherMax = 16;
struct GatherArgs {
    const float *src[kGatherMax];
    long long dst_off[kGatherMax];
    long long size[kGatherMax];
};

// chunk blockIdx.y of the argument table -> its slice of dst (zeros where the chunk has no source)
__global__ void __launch_bounds__(kBlock)
gather_chunks_kernel(GatherArgs a, float *__restrict__ dst) {
    const int ch = blockIdx.y;
    const float *src = a.src[ch];
    float *out = dst + a.dst_off[ch];
    const long long n = a.size[ch];
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
        out[i] = src ? src[i] : 0.f;
}

}  // namespace

extern "C" {

int rwk_gather_chunks(const float *const *srcs, const int64_t *sizes, int64_t n_chunks, float *dst,
                      rwk_stream_t stream_) {
    if (n_chunks < 0) return RWK_E_SIZE;
    if (n_chunks == 0) return RWK_OK;
    RWK_REQUIRE_PTR(srcs);
    RWK_REQUIRE_PTR(sizes);
    RWK_REQUIRE_PTR(dst);
    hipStream_t stream = (hipStream_t)stream_;
    long long off = 0;
    for (int64_t c0 = 0; c0 < n_chunks; c0 += kGatherMax) {
        GatherArgs a;
        const int m = (int)(n_chunks - c0 < kGatherMax ? n_chunks - c0 : kGatherMax);
        long long largest = 1;
        for (int i = 0; i < kGatherMax; ++i) {
            const bool used = i < m;
            if (used && sizes[c0 + i] < 0) return RWK_E_SIZE;
            a.src[i] = used ? srcs[c0 + i] : nullptr;
            a.size[i] = used ? sizes[c0 + i] : 0;
            a.dst_off[i] = off;
            if (used) {
                off += sizes[c0 + i];
                if (sizes[c0 + i] > largest) largest = sizes[c0 + i];
            }
        }
        long long bx = rwk_ceil_div(largest, kBlock);
        if (bx > 1024) bx = 1024;
        hipLaunchKernelGGL(gather_chunks_kernel, dim3((unsigned)bx, (unsigned)m), dim3(kBlock), 0, stream, a, dst);
        RWK_LAUNCH_CHECK();
    }
    return RWK_OK;
}

int rwk_adam_step_chunks(float *param, const float *const *srcs, const int64_t *sizes, int64_t n_chunks,
                         float *grad_out, float *exp_avg, float *exp_avg_sq, int64_t n, const float *lr, float beta1,
                         float beta2, float eps, float weight_decay, int64_t *step_counters, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(param);
    RWK_REQUIRE_PTR(srcs);
    RWK_REQUIRE_PTR(sizes);
    RWK_REQUIRE_PTR(exp_avg);
    RWK_REQUIRE_PTR(exp_avg_sq);
    RWK_REQUIRE_PTR(lr);
    RWK_REQUIRE_PTR(step_counters);
    if (n <= 0 || n >= 0x7FFFFFFFll * kBlock || n_chunks < 1) return RWK_E_SIZE;
    if (n_chunks > kAdamChunks) return RWK_E_UNSUPPORTED;
    AdamChunks a;
    long long off = 0;
    for (int k = 0; k < kAdamChunks; ++k) {
        const bool used = k < n_chunks;
        if (used && sizes[k] < 0) return RWK_E_SIZE;
        a.src[k] = used ? srcs[k] : nullptr;
        a.off[k] = used ? off : 0x7FFFFFFFFFFFFFFFll;
        if (used) off += sizes[k];
    }
    if (off != n) return RWK_E_SIZE;
    hipLaunchKernelGGL(adam_step_chunks_kernel, dim3((unsigned)rwk_ceil_div(n, kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, param, a, grad_out, exp_avg, exp_avg_sq, (long long)n, lr, beta1, beta2,
                       eps, weight_decay, (long long *)step_counters);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

int64_t rwk_adam_n_counters(int64_t n) { return n > 0 ? rwk_ceil_div(n, kBlock) : 0; }

int rwk_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                  const float *lr, float beta1, float beta2, float eps, float weight_decay,
                  int64_t *step_counters, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(param);
    RWK_REQUIRE_PTR(grad);
    RWK_REQUIRE_PTR(exp_avg);
    RWK_REQUIRE_PTR(exp_avg_sq);
    RWK_REQUIRE_PTR(lr);
    RWK_REQUIRE_PTR(step_counters);
    if (n <= 0 || n >= 0x7FFFFFFFll * kBlock) return RWK_E_SIZE;
    hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)rwk_ceil_div(n, kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, (long long)n, lr, beta1, beta2, eps,
                       weight_decay, (long long *)step_counters);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

}  // extern "C"
