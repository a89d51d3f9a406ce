// Issue-rate micro-benchmark: cycles per wave-instruction on one SIMD for the instruction classes the
// hidden_dim <= 4 fused kernels are made of, at 1, 2 and 4 wavefronts per SIMD.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int UNROLL = 8;   // groups of 8 independent instructions per loop body

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

enum Op { FMA, PKFMA, XOR, MAD64, MULHI, MULLO, CNDMASK, CMPSDWA, ADDDPP, MUL24, ALIGNBIT, MFMA16, MFMA4, MFMA16_FMA4, MFMA16_FMA6, MFMA16_FMA8, MFMA16_XOR6, NOPS, CND64, CNDIND, CVTUB, SUBCLAMP, BFE, ANDB, ASHR, SUBSDWA, MULF, ADDF, MOV, ADDU, BITOP3, PERM, MFMA16A, MFMA16DEP, MFMA4A, MFMA16_FMA0, MFMA16_FMA2, MFMA16_FMA12, MFMA16_FMA16, SPLIT, CMPCND };

template <int OP>
__global__ void __launch_bounds__(1024) bench(unsigned long long *out, float *sink, int iters) {
    float f[8];
    unsigned u[8];
    unsigned long long q[8];
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8];
    f4 acc[4];
    for (int i = 0; i < 8; ++i) {
        f[i] = threadIdx.x * 0.001f + i;
        u[i] = threadIdx.x * 2654435761u + i;
        q[i] = u[i];
        p[i] = f2{f[i], f[i] + 1.f};
    }
    for (int i = 0; i < 4; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    const float m = 1.0001f, a = 0.5f;
    const unsigned um = 0xD2511F53u + threadIdx.x, um2 = 0xCD9E8D57u ^ threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < UNROLL; ++r) {
            if (OP == FMA) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(m), "v"(a));
                REP8(S)
#undef S
            } else if (OP == PKFMA) {
#define S(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
                REP8(S)
#undef S
            } else if (OP == XOR) {
#define S(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(um));
                REP8(S)
#undef S
            } else if (OP == MAD64) {
#define S(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q[i]) : "v"(u[i]), "v"(um) : "vcc");
                REP8(S)
#undef S
            } else if (OP == MULHI) {
#define S(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(um));
                REP8(S)
#undef S
            } else if (OP == MULLO) {
#define S(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(um));
                REP8(S)
#undef S
            } else if (OP == CNDMASK) {
#define S(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(um) : );
                REP8(S)
#undef S
            } else if (OP == CMPSDWA) {
#define S(i) asm volatile("v_cmp_lt_u32_sdwa vcc, %0, %1 src0_sel:BYTE_1 src1_sel:DWORD" : : "v"(u[i]), "v"(um) : "vcc");
                REP8(S)
#undef S
            } else if (OP == ADDDPP) {
#define S(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(f[i]));
                REP8(S)
#undef S
            } else if (OP == MUL24) {
#define S(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(um));
                REP8(S)
#undef S
            } else if (OP == ALIGNBIT) {
#define S(i) asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(u[i]));
                REP8(S)
#undef S
            } else if (OP == MFMA16) {
#define S(i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[i], f[(i + 1) & 7], acc[i & 3], 0, 0, 0);
                REP8(S)
#undef S
            } else if (OP == MFMA4) {
#define S(i) acc[i & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(f[i], f[(i + 1) & 7], acc[i & 3], 0, 0, 0);
                REP8(S)
#undef S
            } else if (OP == MFMA16_FMA4 || OP == MFMA16_FMA6 || OP == MFMA16_FMA8) {
                constexpr int NF = OP == MFMA16_FMA4 ? 4 : (OP == MFMA16_FMA6 ? 6 : 8);
#define S(i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(m), "v"(a)); \
             for (int z = 0; z < NF; ++z) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[z]) : "v"(m), "v"(a));
                REP8(S)
#undef S
            } else if (OP == MFMA16_XOR6) {
#define S(i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(m), "v"(a)); \
             for (int z = 0; z < 6; ++z) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[z]) : "v"(um));
                REP8(S)
#undef S
            } else if (OP == CND64) {
#define S(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(u[i]) : "v"(um) : "s20", "s21");
                REP8(S)
#undef S
            } else if (OP == CNDIND) {
#define S(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u[i]) : "v"(um), "v"(um2));
                REP8(S)
#undef S
            } else if (OP == CMPCND) {
#define S(i) asm volatile("v_cmp_lt_u32 vcc, %1, %2\n v_cndmask_b32 %0, 0, %3, vcc" : "=v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(um), "v"(um2) : "vcc");
                REP8(S)
#undef S
            } else if (OP == CVTUB) {
#define S(i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(f[i]) : "v"(u[i]));
                REP8(S)
#undef S
            } else if (OP == SUBCLAMP) {
#define S(i) asm volatile("v_sub_f32_e64 %0, %1, %0 clamp" : "+v"(f[i]) : "v"(m));
                REP8(S)
#undef S
            } else if (OP == BFE) {
#define S(i) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(u[i]));
                REP8(S)
#undef S
            } else if (OP == ANDB) {
#define S(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(um));
                REP8(S)
#undef S
            } else if (OP == ASHR) {
#define S(i) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(u[i]));
                REP8(S)
#undef S
            } else if (OP == SUBSDWA) {
#define S(i) asm volatile("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(u[i]) : "v"(um));
                REP8(S)
#undef S
            } else if (OP == MULF) {
#define S(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(m));
                REP8(S)
#undef S
            } else if (OP == ADDF) {
#define S(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(m));
                REP8(S)
#undef S
            } else if (OP == MOV) {
#define S(i) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(um));
                REP8(S)
#undef S
            } else if (OP == ADDU) {
#define S(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(um));
                REP8(S)
#undef S
            } else if (OP == BITOP3) {
#define S(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(u[i]) : "v"(um), "v"(um2));
                REP8(S)
#undef S
            } else if (OP == PERM) {
#define S(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(um), "v"(um2));
                REP8(S)
#undef S
            } else if (OP == MFMA16A) {
#define S(i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(m), "v"(a));
                REP8(S)
#undef S
            } else if (OP == MFMA16DEP) {
#define S(i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(m), "v"(a));
                REP8(S)
#undef S
            } else if (OP == MFMA4A) {
#define S(i) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(m), "v"(a));
                REP8(S)
#undef S
            } else if (OP == MFMA16_FMA0 || OP == MFMA16_FMA2 || OP == MFMA16_FMA12 || OP == MFMA16_FMA16) {
                constexpr int NF = OP == MFMA16_FMA0 ? 0 : (OP == MFMA16_FMA2 ? 2 : (OP == MFMA16_FMA12 ? 12 : 16));
#define S(i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(m), "v"(a)); \
             for (int z = 0; z < NF; ++z) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[z & 7]) : "v"(m), "v"(a));
                REP8(S)
#undef S
            } else if (OP == SPLIT) {
                // even-numbered SIMD partners: waves 0..3 of the workgroup issue matrix instructions only, the rest
                // vector FMAs only (wave-uniform branch)
                if ((threadIdx.x >> 6) < 4) {
#define S(i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(m), "v"(a));
                    REP8(S)
#undef S
                } else {
#define S(i) for (int z = 0; z < 8; ++z) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[z]) : "v"(m), "v"(a));
                    REP8(S)
#undef S
                }
            } else if (OP == NOPS) {
#define S(i) asm volatile("s_nop 0");
                REP8(S)
#undef S
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    unsigned x = 0;
    for (int i = 0; i < 8; ++i) { s += f[i] + p[i].x + p[i].y; x ^= u[i] ^ (unsigned)q[i] ^ (unsigned)(q[i] >> 32); }
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 1.2345f && x == 77u) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
void run(const char *name, int per_group) {
    const int iters = 2000;
    unsigned long long *out;
    float *sink;
    CHECK(hipMalloc(&out, sizeof(unsigned long long) * 256 * 16));
    CHECK(hipMalloc(&sink, 4));
    printf("%-14s", name);
    for (int wps : {1, 2, 4}) {
        const int threads = 256 * wps;
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((bench<OP>), dim3(256), dim3(threads), 0, 0, out, sink, iters);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(256 * threads / 64);
        CHECK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
        if (OP == SPLIT) {
            std::vector<unsigned long long> a_, b_;
            const int wpb = threads / 64;
            for (size_t k = 0; k < h.size(); ++k) ((int)(k % wpb) < 4 ? a_ : b_).push_back(h[k]);
            std::sort(a_.begin(), a_.end());
            std::sort(b_.begin(), b_.end());
            if (!b_.empty())
                printf("  wps%d: mfma-waves %.1f cyc/mfma, fma-waves %.2f cyc/fma", wps, (double)a_[a_.size() / 2] / (iters * UNROLL * 8.0),
                       (double)b_[b_.size() / 2] / (iters * UNROLL * 64.0));
            continue;
        }
        std::sort(h.begin(), h.end());
        const double cyc = (double)h[h.size() / 2];
        const double n_inst = (double)iters * UNROLL * 8 * per_group;
        // cycles per instruction as seen by one wave, and SIMD-level cycles per instruction
        printf("  wps%d: %6.2f/wave %6.2f/simd", wps, cyc / n_inst, cyc / n_inst / wps);
    }
    printf("\n");
    CHECK(hipFree(out));
    CHECK(hipFree(sink));
}

int main() {
    printf("cycles per wave-instruction (median wave; /simd = divided by wavefronts per SIMD)\n");
    run<FMA>("v_fma_f32", 1);
    run<PKFMA>("v_pk_fma_f32", 1);
    run<XOR>("v_xor_b32", 1);
    run<MAD64>("v_mad_u64_u32", 1);
    run<MULHI>("v_mul_hi_u32", 1);
    run<MULLO>("v_mul_lo_u32", 1);
    run<MUL24>("v_mul_u32_u24", 1);
    run<ALIGNBIT>("v_alignbit", 1);
    run<CNDMASK>("v_cndmask", 1);
    run<CMPSDWA>("v_cmp_sdwa", 1);
    run<ADDDPP>("v_add_f32_dpp", 1);
    run<MFMA16>("mfma16x16x4", 1);
    run<MFMA4>("mfma4x4x1", 1);
    run<MFMA16_FMA4>("mfma16+4fma", 5);
    run<MFMA16_FMA6>("mfma16+6fma", 7);
    run<MFMA16_FMA8>("mfma16+8fma", 9);
    run<MFMA16_XOR6>("mfma16+6xor", 7);
    run<NOPS>("s_nop", 1);
    run<CND64>("cndmask_e64", 1);
    run<CNDIND>("cndmask_indep", 1);
    run<CMPCND>("cmp+cndmask", 2);
    run<CVTUB>("cvt_f32_ubyte1", 1);
    run<SUBCLAMP>("sub_f32 clamp", 1);
    run<BFE>("v_bfe_u32", 1);
    run<ANDB>("v_and_b32", 1);
    run<ASHR>("v_ashrrev_i32", 1);
    run<SUBSDWA>("v_sub_u32_sdwa", 1);
    run<MULF>("v_mul_f32", 1);
    run<ADDF>("v_add_f32", 1);
    run<MOV>("v_mov_b32", 1);
    run<ADDU>("v_add_u32", 1);
    run<BITOP3>("v_bitop3_b32", 1);
    run<PERM>("v_perm_b32", 1);
    run<MFMA16A>("mfma16 4acc", 1);
    run<MFMA16DEP>("mfma16 1acc", 1);
    run<MFMA4A>("mfma4x4x1_16b", 1);
    run<MFMA16_FMA0>("mfma16+0fma", 1);
    run<MFMA16_FMA2>("mfma16+2fma", 3);
    run<MFMA16_FMA12>("mfma16+12fma", 13);
    run<MFMA16_FMA16>("mfma16+16fma", 17);
    run<SPLIT>("split roles", 1);
    return 0;
}
