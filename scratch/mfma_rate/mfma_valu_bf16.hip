// Do MFMAs of one wavefront overlap with vector-ALU instructions (a) of the other wavefront on the same SIMD,
// (b) of the same wavefront placed between them?  gfx950.  hipcc --offload-arch=gfx950 -O3 mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// mode 0: every wavefront MFMAs; 1: every wavefront integer multiply-adds; 2: wavefronts 0-3 MFMA, 4-7 integer;
// 3: every wavefront one MFMA then NV integer multiply-adds, alternating
template <int NV>
__global__ void __launch_bounds__(512) k(float *out, unsigned long long *cyc, int iters, int mode, uint32_t seed) {
    const int wave = threadIdx.x >> 6;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    bf16x8 a, b; for (int q = 0; q < 8; ++q) { a[q] = (__bf16)(1.0f + threadIdx.x * 1e-3f + q); b[q] = (__bf16)(1e-2f * q); }
    uint32_t c0 = seed + threadIdx.x, c1 = seed * 3 + 1;
    // modes 4, 5: as 2 with the integer wavefronts at priority 3 / roles swapped (integer on the older wavefronts)
    const bool do_mfma = mode == 0 || ((mode == 2 || mode == 4) && wave < 4) || (mode == 5 && wave >= 4);
    const bool do_valu = mode == 1 || ((mode == 2 || mode == 4) && wave >= 4) || (mode == 5 && wave < 4);
    if (mode == 4 && do_valu) __builtin_amdgcn_s_setprio(3);
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (mode == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (m & 1) acc1 = MFMA(a, b, acc1); else acc0 = MFMA(a, b, acc0);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const uint64_t p = (uint64_t)0xD2511F53u * c0;
                    c0 = (uint32_t)(p >> 32) ^ c1;
                    c1 = (uint32_t)p;
                }
            }
        }
    } else if (mode == 6 || mode == 7) {                     // wavefronts 0-3: MFMA + idle states; 4-7: integer
        if (wave < 4) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    if (m & 1) acc1 = MFMA(a, b, acc1); else acc0 = MFMA(a, b, acc0);
                    asm volatile("s_nop 5");
                    if (mode == 7) asm volatile("s_nop 0");
                }
            }
        } else {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int v = 0; v < 8 * NV; ++v) {
                    const uint64_t p = (uint64_t)0xD2511F53u * c0;
                    c0 = (uint32_t)(p >> 32) ^ c1;
                    c1 = (uint32_t)p;
                }
            }
        }
    } else if (do_mfma) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (m & 1) acc1 = MFMA(a, b, acc1); else acc0 = MFMA(a, b, acc0);
            }
        }
    } else if (do_valu) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int v = 0; v < 8 * NV; ++v) {
                const uint64_t p = (uint64_t)0xD2511F53u * c0;
                c0 = (uint32_t)(p >> 32) ^ c1;
                c1 = (uint32_t)p;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc1[1] + (float)(c0 ^ c1);
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

template <int NV>
void run(const char *name, int mode, float *out, unsigned long long *cyc) {
    const int iters = 2048;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<NV>), dim3(256), dim3(512), 0, 0, out, cyc, iters, mode, 12345u);
        (void)hipDeviceSynchronize();
    }
    unsigned long long h[8];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-60s wave 0: %7.1f  wave 4: %7.1f cycles per group (8 MFMAs and/or %d multiply-xor pairs)\n", name,
           (double)h[0] / iters, (double)h[4] / iters, 8 * NV);
}

int main() {
    float *out;
    unsigned long long *cyc;
    (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    (void)hipMalloc(&cyc, 8 * sizeof(unsigned long long));
    run<4>("all 8 wavefronts MFMA (2 per SIMD)", 0, out, cyc);
    run<4>("all 8 wavefronts integer (2 per SIMD), 32 pairs", 1, out, cyc);
    run<4>("wavefronts 0-3 MFMA, 4-7 integer (32 pairs)", 2, out, cyc);
    run<4>("every wavefront: MFMA, 4 pairs, MFMA, ... (2 per SIMD)", 3, out, cyc);
    run<2>("every wavefront: MFMA, 2 pairs, MFMA, ... (2 per SIMD)", 3, out, cyc);
    run<4>("wavefronts 0-3 MFMA, 4-7 integer at priority 3 (32 pairs)", 4, out, cyc);
    run<4>("wavefronts 0-3 integer, 4-7 MFMA (32 pairs)", 5, out, cyc);
    run<4>("wavefronts 0-3 MFMA + s_nop 5, 4-7 integer (32 pairs)", 6, out, cyc);
    run<4>("wavefronts 0-3 MFMA + s_nop 5 + s_nop 0, 4-7 integer (32 pairs)", 7, out, cyc);
    run<8>("all integer, 64 pairs", 1, out, cyc);
    run<8>("wavefronts 0-3 MFMA, 4-7 integer (64 pairs)", 2, out, cyc);
    return 0;
}
