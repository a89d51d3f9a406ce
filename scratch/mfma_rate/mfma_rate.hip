// Issue-rate micro-benchmark of v_mfma_f32_16x16x4_f32 on gfx950: cycles per MFMA and wavefront for chains of
// different dependency distance, with 1 or 2 wavefronts per SIMD, with and without LDS operand reads in between.
// hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int CHAINS, bool LDS>
__global__ void __launch_bounds__(512) rate_kernel(float *out, unsigned long long *cyc, int iters, float a0, float b0) {
    __shared__ float tile[8 * 1024];
    for (int i = threadIdx.x; i < 8 * 1024; i += blockDim.x) tile[i] = (float)(i & 7) * 1e-3f;
    __syncthreads();
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    const float *my = tile + (threadIdx.x >> 6) * 1024 + (threadIdx.x & 63) * 4;
    float4 nxt = *reinterpret_cast<const float4 *>(my);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        float4 cur = nxt;
        if (LDS) nxt = *reinterpret_cast<const float4 *>(my + ((it & 3) << 8));
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float av = LDS ? (k & 1 ? cur.y : cur.x) + a : a;
            acc[k % CHAINS] = MFMA(av, b, acc[k % CHAINS]);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    f32x4 s = acc[0];
    for (int c = 1; c < CHAINS; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int CHAINS, bool LDS>
void run(const char *name, int threads, float *out, unsigned long long *cyc) {
    const int iters = 4096;
    hipLaunchKernelGGL((rate_kernel<CHAINS, LDS>), dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0f, 1e-4f);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((rate_kernel<CHAINS, LDS>), dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0f, 1e-4f);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-28s %d waves/SIMD: %.1f cycles per MFMA and wavefront (%.1f per SIMD slot)\n", name, threads / 256,
           (double)h[0] / (iters * 8.0), (double)h[0] / (iters * 8.0) / (threads / 256));
}

int main() {
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipMalloc(&cyc, 8 * sizeof(unsigned long long));
    run<1, false>("1 chain", 256, out, cyc);
    run<2, false>("2 chains alternating", 256, out, cyc);
    run<4, false>("4 chains", 256, out, cyc);
    run<1, false>("1 chain", 512, out, cyc);
    run<2, false>("2 chains alternating", 512, out, cyc);
    run<4, false>("4 chains", 512, out, cyc);
    run<2, true>("2 chains + LDS operand", 256, out, cyc);
    run<2, true>("2 chains + LDS operand", 512, out, cyc);
    return 0;
}
