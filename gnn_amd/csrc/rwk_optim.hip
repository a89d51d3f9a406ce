// Adam update over one flat parameter buffer: one launch per training step instead of the ~45 small
// launches torch.optim.Adam issues for the 10 parameter tensors of RW_NN (rwgnn2/main.py:79,87).
// The step count lives on the device (one counter per workgroup, so no cross-workgroup ordering is
// needed) and the learning rate is read from device memory, so a captured HIP graph replays correctly
// and a scheduler can change the rate without re-capturing.
#include "rwk_common.h"

namespace {

constexpr int kBlock = 256;

__global__ void __launch_bounds__(kBlock)
adam_step_kernel(float *__restrict__ param, const float *__restrict__ grad, float *__restrict__ exp_avg,
                 float *__restrict__ exp_avg_sq, long long n, const float *__restrict__ lr_ptr, float beta1,
                 float beta2, float eps, float weight_decay, long long *__restrict__ step_counters) {
    __shared__ float corr[2];
    if (threadIdx.x == 0) {
        const long long t = step_counters[blockIdx.x] + 1;
        step_counters[blockIdx.x] = t;
        corr[0] = (float)(1.0 - pow((double)beta1, (double)t));
        corr[1] = (float)sqrt(1.0 - pow((double)beta2, (double)t));
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

}  // namespace

extern "C" {

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
