// Device bodies of the narrow-input projection kernels, shared by the stand-alone kernels in rwk_project.hip and
// the merged launches in rwk_merged.hip.
#pragma once

#include "rwk_common.h"

constexpr int kProjectBlock = 256;

static inline int project_pow2_ceil(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

// hidden_dim 4 (the reference's default): one thread per node, one 16-byte store per row
// SIGMOID = false: the same bodies without the activation (a plain Linear with narrow input)
template <int FP, bool SIGMOID = true>
__device__ inline void project_forward4_body(int block, const float *__restrict__ feat, const float *__restrict__ w,
                                             const float *__restrict__ b, int n, int f, float *__restrict__ x0) {
    const int node = block * kProjectBlock + threadIdx.x;
    if (node >= n) return;
    float fr[FP];
#pragma unroll
    for (int k = 0; k < FP; ++k) fr[k] = feat[(size_t)node * f + (k < f ? k : f - 1)];
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float acc = b[c];
#pragma unroll
        for (int k = 0; k < FP; ++k)
            if (k < f) acc = fmaf(fr[k], w[c * f + k], acc);
        o[c] = SIGMOID ? 1.0f / (1.0f + expf(-acc)) : acc;
    }
    reinterpret_cast<float4 *>(x0)[node] = make_float4(o[0], o[1], o[2], o[3]);
}

// Thread (slot, c): c = output channel, slot strides over nodes.  FP = padded input width.
// partials row layout: [c * f + k] for dW (d*f values) then [d*f + c] for db.
// workgroup `block` of `grid`; lds: (FP + 1) * kProjectBlock floats
//
// Wide outputs (dpad >= 16: at most 16 nodes per round of the workgroup) with more than one input feature: a node's
// feature row is shared by the dpad lanes of its slot, and read from memory per lane it costs one address per lane
// and feature - at hidden_dim 64 with 16 features the address unit, not HBM, then bounds the kernel (38 us for
// 59 MB).  The workgroup then walks ONE contiguous range of nodes, copies their feature rows to LDS tile by tile
// with coalesced loads and reads them from there (16-byte broadcast reads); x0 and dx0 stay per-lane loads (one
// contiguous row per slot).
constexpr int kProjectTileNodes = 64;
template <int FP, bool SIGMOID = true>
__device__ inline void project_backward_body(int block, int grid, float *lds, const float *__restrict__ dx0,
                                             const float *__restrict__ x0, const float *__restrict__ feat,
                                             const float *__restrict__ w, long long n, int f, int d, int dpad,
                                             float *__restrict__ partials, float *__restrict__ d_feat) {
    constexpr int kBlock = kProjectBlock;
    const int tid = threadIdx.x;
    const int c = tid % dpad;
    const int slot = tid / dpad;
    const int slots = kBlock / dpad;
    const bool own = c < d;
    float acc_w[FP];
#pragma unroll
    for (int k = 0; k < FP; ++k) acc_w[k] = 0.f;
    float acc_b = 0.f;
    float wr[FP];
#pragma unroll
    for (int k = 0; k < FP; ++k) wr[k] = (own && k < f) ? w[(size_t)c * f + k] : 0.f;
    if (FP >= 4 && dpad >= 16) {
        static_assert(kProjectTileNodes * 16 <= 17 * kProjectBlock, "the feature tile lives in the reduction buffer");
        float *tile = lds;                                   // [kProjectTileNodes][FP]
        const long long per_wg = ((n + grid - 1) / grid + kProjectTileNodes - 1) / kProjectTileNodes * kProjectTileNodes;
        const long long first = (long long)block * per_wg;
        const long long last = first + per_wg < n ? first + per_wg : n;
        const int cc = own ? c : d - 1;
        const float inv_f = 1.0f / (float)f;
        const int per_thread = kProjectTileNodes / slots;    // rounds of a tile (4, 8 or 16)
        for (long long base = first; base < last; base += kProjectTileNodes) {
            const int nodes = (int)(last - base < kProjectTileNodes ? last - base : kProjectTileNodes);
            __syncthreads();                                 // the previous tile has been read
            for (int i = tid; i < nodes * f; i += kBlock) {
                const int r = (int)(((float)i + 0.5f) * inv_f);                 // i / f, exact for i < 2^17
                tile[r * FP + (i - r * f)] = feat[base * f + i];
            }
            __syncthreads();
            for (int u0 = 0; u0 < per_thread; u0 += 4) {     // uniform: the optional group sum sees every lane
                float xv[4], dv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int local = (u0 + u) * slots + slot;
                    const long long nc = base + (local < nodes ? local : nodes - 1);
                    xv[u] = SIGMOID ? x0[nc * d + cc] : 0.f;
                    dv[u] = dx0[nc * d + cc];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int local = (u0 + u) * slots + slot;
                    const bool valid = local < nodes && own;
                    const float dpre = valid ? (SIGMOID ? dv[u] * xv[u] * (1.0f - xv[u]) : dv[u]) : 0.f;
                    acc_b += dpre;
                    float fr[FP];
                    const int lr = local < nodes ? local : nodes - 1;
#pragma unroll
                    for (int k = 0; k < FP; k += 4) {
                        const float4 v = *reinterpret_cast<const float4 *>(tile + lr * FP + k);
                        fr[k] = v.x; fr[k + 1] = v.y; fr[k + 2] = v.z; fr[k + 3] = v.w;
                    }
#pragma unroll
                    for (int k = 0; k < FP; ++k) acc_w[k] = fmaf(dpre, k < f ? fr[k] : 0.f, acc_w[k]);
                    if (d_feat != nullptr) {
#pragma unroll
                        for (int k = 0; k < FP; ++k) {
                            const float v = rwk_group_sum(dpre * wr[k], dpad);
                            if (local < nodes && c == 0 && k < f) d_feat[(base + local) * f + k] = v;
                        }
                    }
                }
            }
        }
        __syncthreads();                                     // the last tile has been read: lds becomes the sum buffer
    } else {
    // uniform trip count so that the optional group sum sees every lane; the loads of four rounds are issued
    // together (unconditional, at clamped indices), then the rounds are added in order
    const long long per_round = (long long)grid * slots;
    const long long rounds = (n + per_round - 1) / per_round;
    const int cc = own ? c : d - 1;
    for (long long r0 = 0; r0 < rounds; r0 += 4) {
        float xv[4], dv[4], fr[4][FP];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long node = (r0 + u) * per_round + (long long)block * slots + slot;
            const long long nc = node < n ? node : n - 1;
            xv[u] = x0[nc * d + cc];
            dv[u] = dx0[nc * d + cc];
#pragma unroll
            for (int k = 0; k < FP; ++k) fr[u][k] = feat[nc * f + (k < f ? k : f - 1)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (r0 + u >= rounds) break;                  // uniform
            const long long node = (r0 + u) * per_round + (long long)block * slots + slot;
            const bool valid = node < n && own;
            const float dpre = valid ? (SIGMOID ? dv[u] * xv[u] * (1.0f - xv[u]) : dv[u]) : 0.f;
            acc_b += dpre;
#pragma unroll
            for (int k = 0; k < FP; ++k) acc_w[k] = fmaf(dpre, (valid && k < f) ? fr[u][k] : 0.f, acc_w[k]);
            if (d_feat != nullptr) {
#pragma unroll
                for (int k = 0; k < FP; ++k) {
                    const float v = rwk_group_sum(dpre * wr[k], dpad);
                    if (node < n && c == 0 && k < f) d_feat[node * f + k] = v;
                }
            }
        }
    }
    }
#pragma unroll
    for (int k = 0; k < FP; ++k) lds[k * kBlock + tid] = acc_w[k];
    lds[FP * kBlock + tid] = acc_b;
    __syncthreads();
    const int n_out = d * f + d;
    float *out = partials + (size_t)block * n_out;
    for (int o = tid; o < n_out; o += kBlock) {
        int cc, row;
        if (o < d * f) { cc = o / f; row = o % f; } else { cc = o - d * f; row = FP; }
        float acc = 0.f;
        for (int q = 0; q < slots; ++q) acc += lds[row * kBlock + q * dpad + cc];
        out[o] = acc;
    }
}

