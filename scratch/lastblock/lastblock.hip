// Micro-benchmark: a two-level column reduction as (a) two launches, (b) one launch whose last workgroup to finish
// adds the partial rows (device-scope fence + atomic counter; no workgroup ever waits for another).
// Build: hipcc --offload-arch=gfx950 -O3 scratch/lastblock/lastblock.hip -o scratch/lastblock/lastblock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kCols = 8;

__device__ inline float block_partial(const float *x, int n_per_block, int col) {
    float acc = 0.f;
    const float *p = x + (size_t)blockIdx.x * n_per_block * kCols;
    for (int r = threadIdx.x / kCols; r < n_per_block; r += 256 / kCols) acc += p[r * kCols + col];
    return acc;
}

__device__ inline void reduce_lanes_and_store(float acc, int col, float *dst) {
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < kCols) {
        float s = 0.f;
        for (int q = 0; q < 256 / kCols; ++q) s += red[q * kCols + threadIdx.x];
        dst[threadIdx.x] = s;
    }
}

__global__ void __launch_bounds__(256) partial_kernel(const float *x, int n_per_block, float *partials) {
    const int col = threadIdx.x % kCols;
    reduce_lanes_and_store(block_partial(x, n_per_block, col), col, partials + (size_t)blockIdx.x * kCols);
}

__device__ inline void final_sum(const float *partials, int p, float *out) {
    // 1024 or 256 threads: column = tid % 8, lanes add rows lane, lane + lanes, ... four in flight, then a tree
    __shared__ float red2[1024];
    const int col = threadIdx.x % kCols, lane = threadIdx.x / kCols, lanes = blockDim.x / kCols;
    float acc = 0.f;
    for (int r0 = lane; r0 < p; r0 += 4 * lanes) {
        float a[4];
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + u * lanes;
            a[u] = __hip_atomic_load(partials + (size_t)(r < p ? r : p - 1) * kCols + col, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
            if (r >= p) a[u] = 0.f;
        }
        for (int u = 0; u < 4; ++u) acc += a[u];
    }
    red2[threadIdx.x] = acc;
    __syncthreads();
    for (int off = lanes / 2; off >= 1; off >>= 1) {
        if (lane < off) red2[threadIdx.x] += red2[threadIdx.x + off * kCols];
        __syncthreads();
    }
    if (lane == 0) out[col] = red2[col];
}

__global__ void __launch_bounds__(1024) final_kernel(const float *partials, int p, float *out) {
    final_sum(partials, p, out);
}

__global__ void __launch_bounds__(256) partial_lastblock_kernel(const float *x, int n_per_block, float *partials,
                                                                unsigned int *counter, float *out) {
    const int col = threadIdx.x % kCols;
    reduce_lanes_and_store(block_partial(x, n_per_block, col), col, partials + (size_t)blockIdx.x * kCols);
    __shared__ unsigned int ticket;
    __threadfence();                                   // partial row visible device-wide before the ticket
    __syncthreads();
    if (threadIdx.x == 0) ticket = atomicAdd(counter, 1u);
    __syncthreads();
    if (ticket != gridDim.x - 1) return;               // not the last workgroup: done (nobody waits)
    __threadfence();
    final_sum(partials, (int)gridDim.x, out);
    if (threadIdx.x == 0) *counter = 0u;               // ready for the next launch
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

int main() {
    const int blocks_list[] = {128, 602, 1024};
    for (int blocks : blocks_list) {
        const int n_per_block = 512;
        const size_t n = (size_t)blocks * n_per_block * kCols;
        std::vector<float> h(n);
        for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
        float *x, *partials, *out_a, *out_b;
        unsigned int *counter;
        CK(hipMalloc(&x, n * 4));
        CK(hipMalloc(&partials, (size_t)blocks * kCols * 4));
        CK(hipMalloc(&out_a, kCols * 4));
        CK(hipMalloc(&out_b, kCols * 4));
        CK(hipMalloc(&counter, 4));
        CK(hipMemset(counter, 0, 4));
        CK(hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice));
        hipStream_t s;
        CK(hipStreamCreate(&s));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        const int iters = 2000;
        float ms_a, ms_b;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int it = 0; it < iters; ++it) {
                hipLaunchKernelGGL(partial_kernel, dim3(blocks), dim3(256), 0, s, x, n_per_block, partials);
                hipLaunchKernelGGL(final_kernel, dim3(1), dim3(1024), 0, s, partials, blocks, out_a);
            }
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms_a, e0, e1));
            CK(hipEventRecord(e0, s));
            for (int it = 0; it < iters; ++it)
                hipLaunchKernelGGL(partial_lastblock_kernel, dim3(blocks), dim3(256), 0, s, x, n_per_block, partials,
                                   counter, out_b);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms_b, e0, e1));
        }
        float ha[kCols], hb[kCols];
        CK(hipMemcpy(ha, out_a, sizeof(ha), hipMemcpyDeviceToHost));
        CK(hipMemcpy(hb, out_b, sizeof(hb), hipMemcpyDeviceToHost));
        double diff = 0;
        for (int c = 0; c < kCols; ++c) diff = fmax(diff, fabs((double)ha[c] - hb[c]));
        printf("blocks %4d: two launches %.2f us/iter, one launch with last-workgroup sum %.2f us/iter, max |diff| %.3g (sum[0] %.4f)\n",
               blocks, ms_a / iters * 1e3, ms_b / iters * 1e3, diff, ha[0]);
        CK(hipFree(x)); CK(hipFree(partials)); CK(hipFree(out_a)); CK(hipFree(out_b)); CK(hipFree(counter));
    }
    return 0;
}
