// Fused random-walk kernel, vector-ALU form (small hidden_dim: the contraction over D is a
// handful of FMAs per element, not a dense GEMM).
//
//   K[g, i*H+h] = c_g * sum_{p in g} sum_s keep[i,h,s,n] * (Z0[h,s].X0[n]) * (Zi[h,s].Xi[n])
//
// Thread layout: a "stream" is stream_len consecutive sorted positions p; HP = pow2 >= H lanes
// serve one stream, lane j owning hidden graph h = j.  Each lane walks its stream sequentially,
// so in the forward pass NO cross-lane or cross-workgroup reduction happens inside the kernel:
// a lane keeps L running sums for the graph it is currently inside and writes them to slot
// (stream + graph) of k_partials whenever the graph id changes; rwk_fused_reduce then adds the
// slots of each graph in increasing stream order (sorted segmented reduction, no atomics).
// The (H*S) x N intermediates (zx, t, the dropout mask) live in registers only.
//
// Backward recomputes zx and t, forms dzx / dt in registers, reduces the per-node dX rows over
// the HP lanes of the stream with a fixed DPP butterfly and writes them once; the gradients of
// Z accumulate per lane in LDS and are reduced per workgroup in a fixed order into dz_partials.
//
// HBM bytes per call (roofline figure): forward  L*N*D*4 + N*4 (+ N*4 with perm) + slots*L*H*4;
//                                        backward 2*L*N*D*4 + N*4 + G*L*H*4 + partials.
#include "rwk_common.h"
#include "rwk_fused_mfma.h"
#include "rwk_fused_reg.h"

namespace {

constexpr size_t kLdsBudget = 150 * 1024;   // of the 160 KiB per CU, leave room for the runtime

struct FusedPlan {
    int hp;          // lanes per stream (power of two >= h)
    int block;       // threads per workgroup
    int s_chunk;     // hidden nodes handled per launch (passes over s when LDS is short)
    int n_streams;
    int n_blocks;
    size_t lds_bytes;
};

inline int pow2_ceil(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}
inline int pad4(int d) { return (d + 3) & ~3; }

// backward = true adds the per-thread dZ accumulators to the LDS bill.
inline bool make_plan(int64_t n, int64_t d, int64_t h, int64_t s, int64_t max_step, int64_t stream_len,
                      bool backward, FusedPlan *plan) {
    if (h > 64 || max_step > 4 || d > 64) return false;
    const int dp = pad4((int)d);
    const int hp = pow2_ceil((int)h);
    plan->hp = hp;
    plan->n_streams = (int)rwk_ceil_div(n, stream_len);
    const int min_block = hp > 64 ? hp : 64;
    auto lds_floats = [&](int sc, int block) {
        size_t floats = (size_t)max_step * h * sc * dp;
        if (backward) floats += (size_t)max_step * sc * dp * block;
        return floats;
    };
    auto accept = [&](int sc, int block) {
        plan->block = block;
        plan->s_chunk = sc;
        plan->lds_bytes = lds_floats(sc, block) * sizeof(float);
        plan->n_blocks = (int)rwk_ceil_div(plan->n_streams, block / hp);
        return true;
    };
    // the whole s range with the largest workgroup that fits; else the smallest workgroup and
    // several launches over slices of s
    for (int block = 256; block >= min_block; block >>= 1)
        if (lds_floats((int)s, block) * sizeof(float) <= kLdsBudget) return accept((int)s, block);
    for (int sc = (int)s - 1; sc >= 1; --sc)
        if (lds_floats(sc, min_block) * sizeof(float) <= kLdsBudget) return accept(sc, min_block);
    return false;
}

struct DropParams {
    uint32_t t0, t1;
};

__device__ inline RwkDropCfg load_drop_cfg(const unsigned long long *rng, DropParams dp) {
    RwkDropCfg cfg;
    const unsigned long long seed = rng[0], off = rng[1];
    cfg.t0 = dp.t0;
    cfg.t1 = dp.t1;
    cfg.seed_lo = (uint32_t)seed;
    cfg.key_hi = (uint32_t)(seed >> 32) ^ (uint32_t)(off >> 32);
    cfg.off_lo = (uint32_t)off;
    return cfg;
}

template <int DP>
__device__ inline void load_row(const float *__restrict__ base, int node, int d, bool valid, float (&out)[DP]) {
    if (!valid) {
#pragma unroll
        for (int k = 0; k < DP; ++k) out[k] = 0.f;
        return;
    }
    const float *row = base + (size_t)node * d;
    if (d == DP) {
#pragma unroll
        for (int k = 0; k < DP; k += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(row + k);
            out[k] = v.x; out[k + 1] = v.y; out[k + 2] = v.z; out[k + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < DP; ++k) out[k] = k < d ? row[k] : 0.f;
    }
}

template <int DP>
__device__ inline float dot_lds(const float *__restrict__ zrow, const float (&xr)[DP]) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < DP; k += 4) {
        const float4 zv = *reinterpret_cast<const float4 *>(zrow + k);
        acc = fmaf(zv.x, xr[k], acc);
        acc = fmaf(zv.y, xr[k + 1], acc);
        acc = fmaf(zv.z, xr[k + 2], acc);
        acc = fmaf(zv.w, xr[k + 3], acc);
    }
    return acc;
}

// Z rows of steps 0..LT-1 for hidden nodes [s_begin, s_end) into LDS, padded to DP floats.
template <int DP, int LT>
__device__ inline void stage_z(const float *__restrict__ z, float *zs, int h, int s, int d, int s_begin, int s_cnt) {
    const int total = LT * h * s_cnt * DP;
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const int k = idx % DP;
        int r = idx / DP;
        const int ss = r % s_cnt; r /= s_cnt;
        const int hh = r % h;
        const int i = r / h;
        zs[idx] = k < d ? z[(((size_t)i * h + hh) * s + s_begin + ss) * d + k] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <int DP, int LT, bool DROP>
__global__ void __launch_bounds__(256)
fused_forward_kernel(const float *__restrict__ x, const float *__restrict__ z, const int *__restrict__ gid,
                     const int *__restrict__ perm, int n, int d, int h, int s, int stream_len, int hp,
                     int s_begin, int s_cnt, int accumulate, DropParams dparams,
                     const unsigned long long *__restrict__ rng, float *__restrict__ k_partials) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *zs = lds;                                   // [LT][h][s_cnt][DP]
    stage_z<DP, LT>(z, zs, h, s, d, s_begin, s_cnt);
    __syncthreads();

    const int lane_h = threadIdx.x % hp;
    const int hh = lane_h < h ? lane_h : h - 1;         // idle lanes read a valid row, write nothing
    const int c = blockIdx.x * (blockDim.x / hp) + threadIdx.x / hp;
    const long long start = (long long)c * stream_len;
    if (start >= n) return;
    const int end = (int)(start + stream_len < n ? start + stream_len : n);
    RwkDropCfg cfg;
    if (DROP) cfg = load_drop_cfg(rng, dparams);
    const int lh = LT * h;
    const size_t step_stride = (size_t)n * d;

    float acc[LT];
#pragma unroll
    for (int i = 0; i < LT; ++i) acc[i] = 0.f;
    int g_cur = gid[start];

    for (int p = (int)start; p < end; ++p) {
        const int g = gid[p];
        if (g != g_cur) {
            if (lane_h < h) {
                float *dst = k_partials + (size_t)(c + g_cur) * lh + lane_h;
#pragma unroll
                for (int i = 0; i < LT; ++i) dst[i * h] = accumulate ? dst[i * h] + acc[i] : acc[i];
            }
#pragma unroll
            for (int i = 0; i < LT; ++i) acc[i] = 0.f;
            g_cur = g;
        }
        const int node = perm ? perm[p] : p;
        float xr[LT][DP];
#pragma unroll
        for (int i = 0; i < LT; ++i) load_row<DP>(x + i * step_stride, node, d, true, xr[i]);

        uint32_t bits = 0;
        int cur_call = -1;
        float node_acc[LT];                              // this node's sum over s, added to acc once
#pragma unroll
        for (int i = 0; i < LT; ++i) node_acc[i] = 0.f;
        for (int ss = 0; ss < s_cnt; ++ss) {
            const float *z0 = zs + ((size_t)(0 * h + hh) * s_cnt + ss) * DP;
            const float zx = dot_lds<DP>(z0, xr[0]);
#pragma unroll
            for (int i = 0; i < LT; ++i) {
                float t = zx;
                if (i > 0) t = dot_lds<DP>(zs + ((size_t)(i * h + hh) * s_cnt + ss) * DP, xr[i]);
                float prod = zx * t;
                if (DROP) {
                    const int e = (s_begin + ss) * LT + i;
                    const int call = e / 15;
                    if (call != cur_call) {
                        bits = rwk_keep_bits15(cfg, (uint32_t)node, (uint32_t)lane_h, (uint32_t)call);
                        cur_call = call;
                    }
                    prod = ((bits >> (e - call * 15)) & 1u) ? prod : 0.f;
                }
                node_acc[i] += prod;
            }
        }
#pragma unroll
        for (int i = 0; i < LT; ++i) acc[i] += node_acc[i];
    }
    if (lane_h < h) {
        float *dst = k_partials + (size_t)(c + g_cur) * lh + lane_h;
#pragma unroll
        for (int i = 0; i < LT; ++i) dst[i * h] = accumulate ? dst[i * h] + acc[i] : acc[i];
    }
}

// K[g, col] = c_g * sum over the streams that touch graph g of their slot, in stream order.
__global__ void __launch_bounds__(256)
fused_reduce_kernel(const float *__restrict__ k_partials, const int *__restrict__ seg_ptr,
                    const float *__restrict__ counts_f32, int n_graphs, int lh, int stream_len, float scale,
                    int normalize, float *__restrict__ k) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n_graphs * lh) return;
    const int g = (int)(t / lh), col = (int)(t % lh);
    const int c_first = seg_ptr[g] / stream_len;
    const int c_last = (seg_ptr[g + 1] - 1) / stream_len;
    // slots are added in stream order; sixteen loads are requested at a time (at clamped indices) so that their
    // latencies overlap: a graph of a few hundred nodes is covered by one or two such groups
    const float *src = k_partials + (size_t)g * lh + col;
    float acc = 0.f;
    for (int c0 = c_first; c0 <= c_last; c0 += 16) {
        float a[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int c = c0 + u;
            a[u] = src[(size_t)(c <= c_last ? c : c_last) * lh];
            if (c > c_last) a[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += a[u];
    }
    acc *= scale;
    if (normalize) acc /= counts_f32[g];
    k[t] = acc;
}

// ------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------
template <int DP, int LT, bool DROP>
__global__ void __launch_bounds__(256)
fused_backward_kernel(const float *__restrict__ x, const float *__restrict__ z, const int *__restrict__ gid,
                      const int *__restrict__ perm, const float *__restrict__ d_k,
                      const float *__restrict__ counts_f32, int n, int d, int h, int s, int stream_len, int hp,
                      int s_begin, int s_cnt, int accumulate_dx, float scale, int normalize, DropParams dparams,
                      const unsigned long long *__restrict__ rng, float *__restrict__ dx,
                      float *__restrict__ dz_partials) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int block = blockDim.x;
    float *zs = lds;                                   // [LT][h][s_cnt][DP]
    float *dzacc = zs + LT * h * s_cnt * DP;           // [LT][s_cnt][DP][block]  (column = thread)
    stage_z<DP, LT>(z, zs, h, s, d, s_begin, s_cnt);
    const int acc_rows = LT * s_cnt * DP;
    for (int idx = threadIdx.x; idx < acc_rows * block; idx += block) dzacc[idx] = 0.f;
    __syncthreads();

    const int tid = threadIdx.x;
    const int lane_h = tid % hp;
    const int hh = lane_h < h ? lane_h : h - 1;
    const bool own_h = lane_h < h;
    const int c = blockIdx.x * (block / hp) + tid / hp;
    const long long start_ll = (long long)c * stream_len;
    const bool has_stream = start_ll < n;
    const int start = has_stream ? (int)start_ll : 0;
    const int end = has_stream ? (int)(start_ll + stream_len < n ? start_ll + stream_len : n) : 0;
    RwkDropCfg cfg;
    if (DROP) cfg = load_drop_cfg(rng, dparams);
    const int lh = LT * h;
    const size_t step_stride = (size_t)n * d;

    float dp_cur[LT];
#pragma unroll
    for (int i = 0; i < LT; ++i) dp_cur[i] = 0.f;
    int g_cur = -1;

    // every lane of the wavefront runs all stream_len iterations: the DPP butterfly below needs
    // all of them; lanes past their stream's end carry zeros.
    for (int it = 0; it < stream_len; ++it) {
        const int p = start + it;
        const bool valid = has_stream && p < end;
        int node = 0;
        if (valid) {
            const int g = gid[p];
            node = perm ? perm[p] : p;
            if (g != g_cur) {
                g_cur = g;
                float f = scale;
                if (normalize) f /= counts_f32[g];
#pragma unroll
                for (int i = 0; i < LT; ++i) dp_cur[i] = own_h ? d_k[(size_t)g * lh + i * h + lane_h] * f : 0.f;
            }
        }
        float xr[LT][DP], dxr[LT][DP];
#pragma unroll
        for (int i = 0; i < LT; ++i) {
            load_row<DP>(x + i * step_stride, node, d, valid, xr[i]);
#pragma unroll
            for (int k = 0; k < DP; ++k) dxr[i][k] = 0.f;
        }
        if (valid && own_h) {
            uint32_t bits = 0;
            int cur_call = -1;
            for (int ss = 0; ss < s_cnt; ++ss) {
                const float *z0 = zs + ((size_t)(0 * h + hh) * s_cnt + ss) * DP;
                const float zx = dot_lds<DP>(z0, xr[0]);
                float w[LT];
#pragma unroll
                for (int i = 0; i < LT; ++i) {
                    w[i] = dp_cur[i];
                    if (DROP) {
                        const int e = (s_begin + ss) * LT + i;
                        const int call = e / 15;
                        if (call != cur_call) {
                            bits = rwk_keep_bits15(cfg, (uint32_t)node, (uint32_t)lane_h, (uint32_t)call);
                            cur_call = call;
                        }
                        w[i] = ((bits >> (e - call * 15)) & 1u) ? w[i] : 0.f;
                    }
                }
                // step 0: k += keep * zx * zx  =>  d zx = 2 * w0 * zx
                float dzx = 2.f * w[0] * zx;
#pragma unroll
                for (int i = 1; i < LT; ++i) {
                    const float *zi = zs + ((size_t)(i * h + hh) * s_cnt + ss) * DP;
                    const float t = dot_lds<DP>(zi, xr[i]);
                    dzx = fmaf(w[i], t, dzx);
                    const float dt = w[i] * zx;
                    float *acc_row = dzacc + (size_t)((i * s_cnt + ss) * DP) * block + tid;
#pragma unroll
                    for (int k = 0; k < DP; ++k) {
                        dxr[i][k] = fmaf(dt, zi[k], dxr[i][k]);
                        acc_row[(size_t)k * block] += dt * xr[i][k];
                    }
                }
                float *acc_row0 = dzacc + (size_t)((0 * s_cnt + ss) * DP) * block + tid;
#pragma unroll
                for (int k = 0; k < DP; ++k) {
                    dxr[0][k] = fmaf(dzx, z0[k], dxr[0][k]);
                    acc_row0[(size_t)k * block] += dzx * xr[0][k];
                }
            }
        }
        // sum the dX rows over the hidden graphs (the hp lanes of this stream) and store once
#pragma unroll
        for (int i = 0; i < LT; ++i) {
#pragma unroll
            for (int k = 0; k < DP; ++k) dxr[i][k] = rwk_group_sum(dxr[i][k], hp);
        }
        if (valid && lane_h == 0) {
#pragma unroll
            for (int i = 0; i < LT; ++i) {
                float *row = dx + i * step_stride + (size_t)node * d;
                if (d == DP) {
#pragma unroll
                    for (int k = 0; k < DP; k += 4) {
                        float4 v = make_float4(dxr[i][k], dxr[i][k + 1], dxr[i][k + 2], dxr[i][k + 3]);
                        float4 *dst = reinterpret_cast<float4 *>(row + k);
                        if (accumulate_dx) {
                            const float4 o = *dst;
                            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                        }
                        *dst = v;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < DP; ++k)
                        if (k < d) row[k] = accumulate_dx ? row[k] + dxr[i][k] : dxr[i][k];
                }
            }
        }
    }

    // per-workgroup partial of dZ: add the threads that own the same hidden graph in thread order
    __syncthreads();
    const int streams_per_block = block / hp;
    float *out = dz_partials + (size_t)blockIdx.x * LT * h * s * d;
    for (int idx = tid; idx < acc_rows * h; idx += block) {
        const int hq = idx % h;
        const int row = idx / h;                        // (i * s_cnt + ss) * DP + k
        const int k = row % DP;
        const int ss = (row / DP) % s_cnt;
        const int i = row / (DP * s_cnt);
        if (k >= d) continue;
        const float *src = dzacc + (size_t)row * block + hq;
        float acc = 0.f;
        for (int q = 0; q < streams_per_block; ++q) acc += src[q * hp];
        out[(((size_t)i * h + hq) * s + s_begin + ss) * d + k] = acc;
    }
}

// ------------------------------------------------------------------------------------------
// mask dump (test hook)
// ------------------------------------------------------------------------------------------
__global__ void dropout_mask_kernel(const unsigned long long *__restrict__ rng, int n, int h, int s, int max_step,
                                    DropParams dparams, uint8_t *__restrict__ keep) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n * h) return;
    const int node = (int)(t % n), hq = (int)(t / n);
    const RwkDropCfg cfg = load_drop_cfg(rng, dparams);
    const int samples = s * max_step;
    for (int call = 0; call * 15 < samples; ++call) {
        const uint32_t bits = rwk_keep_bits15(cfg, (uint32_t)node, (uint32_t)hq, (uint32_t)call);
        for (int j = 0; j < 15 && call * 15 + j < samples; ++j) {
            const int e = call * 15 + j;
            const int ss = e / max_step, i = e % max_step;
            keep[(((size_t)i * h + hq) * s + ss) * n + node] = (uint8_t)((bits >> j) & 1u);
        }
    }
}

// ------------------------------------------------------------------------------------------
// dispatch
// ------------------------------------------------------------------------------------------
struct FwdArgs {
    const float *x, *z; const int *gid, *perm; int n, d, h, s, stream_len; DropParams dparams;
    const unsigned long long *rng; float *k_partials; hipStream_t stream;
};

template <int DP, int LT, bool DROP>
int launch_forward(const FwdArgs &a, const FusedPlan &plan) {
    auto kern = fused_forward_kernel<DP, LT, DROP>;
    if (plan.lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)plan.lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    for (int s0 = 0; s0 < a.s; s0 += plan.s_chunk) {
        const int cnt = a.s - s0 < plan.s_chunk ? a.s - s0 : plan.s_chunk;
        const size_t lds = (size_t)LT * a.h * cnt * DP * sizeof(float);
        hipLaunchKernelGGL(kern, dim3((unsigned)plan.n_blocks), dim3((unsigned)plan.block), lds, a.stream, a.x, a.z,
                           a.gid, a.perm, a.n, a.d, a.h, a.s, a.stream_len, plan.hp, s0, cnt, s0 > 0 ? 1 : 0,
                           a.dparams, a.rng, a.k_partials);
        RWK_LAUNCH_CHECK();
    }
    return RWK_OK;
}

struct BwdArgs {
    const float *x, *z; const int *gid, *perm; const float *d_k, *counts_f32; int n, d, h, s, stream_len;
    float scale; int normalize; DropParams dparams; const unsigned long long *rng; float *dx, *dz_partials;
    hipStream_t stream;
};

template <int DP, int LT, bool DROP>
int launch_backward(const BwdArgs &a, const FusedPlan &plan) {
    auto kern = fused_backward_kernel<DP, LT, DROP>;
    if (plan.lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)plan.lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    for (int s0 = 0; s0 < a.s; s0 += plan.s_chunk) {
        const int cnt = a.s - s0 < plan.s_chunk ? a.s - s0 : plan.s_chunk;
        const size_t lds = ((size_t)LT * a.h * cnt * DP + (size_t)LT * cnt * DP * plan.block) * sizeof(float);
        hipLaunchKernelGGL(kern, dim3((unsigned)plan.n_blocks), dim3((unsigned)plan.block), lds, a.stream, a.x, a.z,
                           a.gid, a.perm, a.d_k, a.counts_f32, a.n, a.d, a.h, a.s, a.stream_len, plan.hp, s0, cnt,
                           s0 > 0 ? 1 : 0, a.scale, a.normalize, a.dparams, a.rng, a.dx, a.dz_partials);
        RWK_LAUNCH_CHECK();
    }
    return RWK_OK;
}

template <int DP, int LT>
int dispatch_forward_drop(const FwdArgs &a, const FusedPlan &plan, bool drop) {
    return drop ? launch_forward<DP, LT, true>(a, plan) : launch_forward<DP, LT, false>(a, plan);
}
template <int DP>
int dispatch_forward_l(const FwdArgs &a, const FusedPlan &plan, int l, bool drop) {
    switch (l) {
        case 1: return dispatch_forward_drop<DP, 1>(a, plan, drop);
        case 2: return dispatch_forward_drop<DP, 2>(a, plan, drop);
        case 3: return dispatch_forward_drop<DP, 3>(a, plan, drop);
        case 4: return dispatch_forward_drop<DP, 4>(a, plan, drop);
    }
    return RWK_E_UNSUPPORTED;
}
template <int DP, int LT>
int dispatch_backward_drop(const BwdArgs &a, const FusedPlan &plan, bool drop) {
    return drop ? launch_backward<DP, LT, true>(a, plan) : launch_backward<DP, LT, false>(a, plan);
}
template <int DP>
int dispatch_backward_l(const BwdArgs &a, const FusedPlan &plan, int l, bool drop) {
    switch (l) {
        case 1: return dispatch_backward_drop<DP, 1>(a, plan, drop);
        case 2: return dispatch_backward_drop<DP, 2>(a, plan, drop);
        case 3: return dispatch_backward_drop<DP, 3>(a, plan, drop);
        case 4: return dispatch_backward_drop<DP, 4>(a, plan, drop);
    }
    return RWK_E_UNSUPPORTED;
}

inline int dp_bucket(int64_t d) {
    const int p = pad4((int)d);
    if (p <= 4) return 4;
    if (p <= 8) return 8;
    if (p <= 16) return 16;
    if (p <= 32) return 32;
    return 64;
}

// make_plan must see the padded width the kernel will really use
inline bool plan_for(int64_t n, int64_t d, int64_t h, int64_t s, int64_t l, int64_t stream_len, bool backward,
                     FusedPlan *plan) {
    if (d > 64 || d <= 0) return false;
    return make_plan(n, dp_bucket(d), h, s, l, stream_len, backward, plan);
}

inline bool sizes_ok(int64_t n, int64_t d, int64_t h, int64_t s, int64_t l, int64_t stream_len) {
    return n > 0 && n < 0x7FFFFFFFll && d > 0 && h > 0 && s > 0 && l > 0 && stream_len > 0 &&
           stream_len <= (1 << 20);
}

}  // namespace

extern "C" {

int64_t rwk_fused_n_streams(int64_t n, int64_t stream_len) {
    if (n <= 0 || stream_len <= 0) return 0;
    return rwk_ceil_div(n, stream_len);
}

int64_t rwk_fused_k_partials_size(int64_t n, int64_t stream_len, int64_t n_graphs, int64_t h, int64_t max_step) {
    if (n <= 0 || stream_len <= 0 || n_graphs <= 0) return 0;
    return (rwk_ceil_div(n, stream_len) + n_graphs) * h * max_step;
}

int64_t rwk_fused_dz_partials_count(int64_t n, int64_t stream_len, int64_t d, int64_t h, int64_t s,
                                    int64_t max_step) {
    FusedPlan plan;
    if (!sizes_ok(n, d, h, s, max_step, stream_len)) return 0;
    if (rwk_reg::supported(d, h, s, max_step)) return rwk_reg::backward_blocks(n, stream_len, h);
    if (rwk_mfma::supported(d, h, s, max_step)) return rwk_mfma::backward_blocks(n, stream_len);
    if (!plan_for(n, d, h, s, max_step, stream_len, true, &plan)) return 0;
    return plan.n_blocks;
}

int64_t rwk_fused_backward_workspace(int64_t n, int64_t d, int64_t h, int64_t s, int64_t max_step) {
    if (n <= 0 || d <= 0 || h <= 0 || s <= 0 || max_step <= 0) return 0;
    if (rwk_reg::supported(d, h, s, max_step)) return 0;
    if (rwk_mfma::supported(d, h, s, max_step)) return rwk_mfma::dx_partials_size(n, d, h, max_step);
    return 0;
}

int rwk_fused_path(int64_t d, int64_t h, int64_t s, int64_t max_step) {
    if (rwk_reg::supported(d, h, s, max_step)) return 1;
    if (rwk_mfma::supported(d, h, s, max_step)) return 2;
    return 0;
}

int64_t rwk_fused_stream_len(int64_t n, int64_t d, int64_t h, int64_t s, int64_t max_step, int backward) {
    if (n <= 0 || d <= 0 || h <= 0 || s <= 0 || max_step <= 0) return 0;
    if (rwk_mfma::supported(d, h, s, max_step) && !rwk_reg::supported(d, h, s, max_step))
        return rwk_mfma::stream_len(n, h, backward != 0);
    return rwk_reg::stream_len(n, h, s, max_step, backward != 0);
}

int64_t rwk_fused_keep_words(int64_t n, int64_t d, int64_t h, int64_t s, int64_t max_step) {
    if (n <= 0 || rwk_reg::supported(d, h, s, max_step)) return 0;
    return rwk_mfma::keep_words(n, d, h, s, max_step);
}

int rwk_fused_forward(const float *x, const float *z, const int32_t *gid, const int32_t *perm, int64_t n,
                      int64_t d, int64_t h, int64_t s, int64_t max_step, int64_t stream_len, float drop_p,
                      const uint64_t *rng, float *k_partials, rwk_stream_t stream) {
    return rwk_fused_forward_bits(x, z, gid, perm, n, d, h, s, max_step, stream_len, drop_p, rng, k_partials, nullptr,
                                  stream);
}

int rwk_fused_forward_bits(const float *x, const float *z, const int32_t *gid, const int32_t *perm, int64_t n,
                           int64_t d, int64_t h, int64_t s, int64_t max_step, int64_t stream_len, float drop_p,
                           const uint64_t *rng, float *k_partials, uint64_t *keep_out, rwk_stream_t stream) {
    if (keep_out != nullptr && rwk_fused_keep_words(n, d, h, s, max_step) == 0) return RWK_E_UNSUPPORTED;
    RWK_REQUIRE_PTR(x);
    RWK_REQUIRE_PTR(z);
    RWK_REQUIRE_PTR(gid);
    RWK_REQUIRE_PTR(k_partials);
    if (!sizes_ok(n, d, h, s, max_step, stream_len) || !(drop_p >= 0.f && drop_p < 1.f)) return RWK_E_SIZE;
    const bool drop = drop_p > 0.f && rng != nullptr;
    if (rwk_reg::supported(d, h, s, max_step)) {
        rwk_reg::Args r{};
        r.x = x; r.z = z; r.gid = gid; r.perm = perm;
        r.n = (int)n; r.d = (int)d; r.h = (int)h; r.s = (int)s; r.max_step = (int)max_step;
        r.stream_len = (int)stream_len; r.drop = drop; r.rng = (const unsigned long long *)rng;
        r.stream = (hipStream_t)stream; r.k_partials = k_partials;
        rwk_drop_thresholds(drop_p, &r.t0, &r.t1);
        return rwk_reg::forward(r);
    }
    if (rwk_mfma::supported(d, h, s, max_step)) {
        rwk_mfma::Args r{};
        r.x = x; r.z = z; r.gid = gid; r.perm = perm;
        r.n = (int)n; r.d = (int)d; r.h = (int)h; r.s = (int)s; r.max_step = (int)max_step;
        r.stream_len = (int)stream_len; r.drop = drop; r.rng = (const unsigned long long *)rng;
        r.stream = (hipStream_t)stream; r.k_partials = k_partials;
        r.keep = (unsigned long long *)keep_out;
        rwk_drop_thresholds(drop_p, &r.t0, &r.t1);
        return rwk_mfma::forward(r);
    }
    FusedPlan plan;
    if (!plan_for(n, d, h, s, max_step, stream_len, false, &plan)) return RWK_E_UNSUPPORTED;
    FwdArgs a{x, z, gid, perm, (int)n, (int)d, (int)h, (int)s, (int)stream_len, {0, 0},
              (const unsigned long long *)rng, k_partials, (hipStream_t)stream};
    rwk_drop_thresholds(drop_p, &a.dparams.t0, &a.dparams.t1);
    switch (dp_bucket(d)) {
        case 4: return dispatch_forward_l<4>(a, plan, (int)max_step, drop);
        case 8: return dispatch_forward_l<8>(a, plan, (int)max_step, drop);
        case 16: return dispatch_forward_l<16>(a, plan, (int)max_step, drop);
        case 32: return dispatch_forward_l<32>(a, plan, (int)max_step, drop);
        default: return dispatch_forward_l<64>(a, plan, (int)max_step, drop);
    }
}

int rwk_fused_reduce(const float *k_partials, const int32_t *seg_ptr, const float *counts_f32, int64_t n_graphs,
                     int64_t h, int64_t max_step, int64_t stream_len, float drop_p, int normalize, float *k,
                     rwk_stream_t stream) {
    RWK_REQUIRE_PTR(k_partials);
    RWK_REQUIRE_PTR(seg_ptr);
    RWK_REQUIRE_PTR(k);
    if (normalize) RWK_REQUIRE_PTR(counts_f32);
    if (n_graphs <= 0 || h <= 0 || max_step <= 0 || stream_len <= 0 || !(drop_p >= 0.f && drop_p < 1.f))
        return RWK_E_SIZE;
    const int64_t total = n_graphs * h * max_step;
    if (total >= 0x7FFFFFFFll * 128) return RWK_E_SIZE;
    const float scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    hipLaunchKernelGGL(fused_reduce_kernel, dim3((unsigned)rwk_ceil_div(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, k_partials, seg_ptr, counts_f32, (int)n_graphs, (int)(h * max_step),
                       (int)stream_len, scale, normalize, k);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

int rwk_fused_backward(const float *x, const float *z, const int32_t *gid, const int32_t *perm, const float *d_k,
                       const float *counts_f32, int64_t n, int64_t d, int64_t h, int64_t s, int64_t max_step,
                       int64_t stream_len, float drop_p, int normalize, const uint64_t *rng, float *dx,
                       float *dz_partials, float *workspace, rwk_stream_t stream) {
    return rwk_fused_backward_bits(x, z, gid, perm, d_k, counts_f32, n, d, h, s, max_step, stream_len, drop_p,
                                   normalize, rng, dx, dz_partials, workspace, nullptr, stream);
}

int rwk_fused_backward_bits(const float *x, const float *z, const int32_t *gid, const int32_t *perm,
                            const float *d_k, const float *counts_f32, int64_t n, int64_t d, int64_t h, int64_t s,
                            int64_t max_step, int64_t stream_len, float drop_p, int normalize, const uint64_t *rng,
                            float *dx, float *dz_partials, float *workspace, const uint64_t *keep_in,
                            rwk_stream_t stream) {
    if (keep_in != nullptr && rwk_fused_keep_words(n, d, h, s, max_step) == 0) return RWK_E_UNSUPPORTED;
    RWK_REQUIRE_PTR(x);
    RWK_REQUIRE_PTR(z);
    RWK_REQUIRE_PTR(gid);
    RWK_REQUIRE_PTR(d_k);
    RWK_REQUIRE_PTR(dx);
    RWK_REQUIRE_PTR(dz_partials);
    if (normalize) RWK_REQUIRE_PTR(counts_f32);
    if (!sizes_ok(n, d, h, s, max_step, stream_len) || !(drop_p >= 0.f && drop_p < 1.f)) return RWK_E_SIZE;
    const bool drop = drop_p > 0.f && rng != nullptr;
    if (rwk_reg::supported(d, h, s, max_step)) {
        rwk_reg::Args r{};
        r.x = x; r.z = z; r.gid = gid; r.perm = perm;
        r.n = (int)n; r.d = (int)d; r.h = (int)h; r.s = (int)s; r.max_step = (int)max_step;
        r.stream_len = (int)stream_len; r.drop = drop; r.rng = (const unsigned long long *)rng;
        r.stream = (hipStream_t)stream; r.d_k = d_k; r.counts_f32 = counts_f32;
        r.scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f; r.normalize = normalize;
        r.dx = dx; r.dz_partials = dz_partials;
        rwk_drop_thresholds(drop_p, &r.t0, &r.t1);
        return rwk_reg::backward(r);
    }
    if (rwk_mfma::supported(d, h, s, max_step)) {
        rwk_mfma::Args r{};
        r.x = x; r.z = z; r.gid = gid; r.perm = perm;
        r.n = (int)n; r.d = (int)d; r.h = (int)h; r.s = (int)s; r.max_step = (int)max_step;
        r.stream_len = (int)stream_len; r.drop = drop; r.rng = (const unsigned long long *)rng;
        r.stream = (hipStream_t)stream; r.d_k = d_k; r.counts_f32 = counts_f32;
        r.scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f; r.normalize = normalize;
        r.dx = dx; r.dz_partials = dz_partials; r.dx_partials = workspace;
        r.keep = (unsigned long long *)keep_in;
        if (rwk_mfma::dx_partials_size(n, d, h, max_step) > 0 && workspace == nullptr) return RWK_E_NULL;
        rwk_drop_thresholds(drop_p, &r.t0, &r.t1);
        return rwk_mfma::backward(r);
    }
    FusedPlan plan;
    if (!plan_for(n, d, h, s, max_step, stream_len, true, &plan)) return RWK_E_UNSUPPORTED;
    BwdArgs a{x, z, gid, perm, d_k, counts_f32, (int)n, (int)d, (int)h, (int)s, (int)stream_len,
              drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f, normalize, {0, 0},
              (const unsigned long long *)rng, dx, dz_partials, (hipStream_t)stream};
    rwk_drop_thresholds(drop_p, &a.dparams.t0, &a.dparams.t1);
    switch (dp_bucket(d)) {
        case 4: return dispatch_backward_l<4>(a, plan, (int)max_step, drop);
        case 8: return dispatch_backward_l<8>(a, plan, (int)max_step, drop);
        case 16: return dispatch_backward_l<16>(a, plan, (int)max_step, drop);
        case 32: return dispatch_backward_l<32>(a, plan, (int)max_step, drop);
        default: return dispatch_backward_l<64>(a, plan, (int)max_step, drop);
    }
}

int rwk_fused_backward_bn_supported(int64_t d, int64_t h, int64_t s, int64_t max_step) {
    return rwk_reg::bn_fold_supported(d, h, s, max_step) ? 1 : 0;
}

int rwk_fused_backward_bn(const float *x, const float *z, const int32_t *gid, const int32_t *perm,
                          const float *counts_f32, int64_t n, int64_t d, int64_t h, int64_t s, int64_t max_step,
                          int64_t stream_len, float drop_p, int normalize, const uint64_t *rng, float *dx,
                          float *dz_partials, const float *k, const float *dy, const float *mean, const float *invstd,
                          const float *bn_w, const float *head_partials, int64_t n_head_partials, int64_t head_nv,
                          int64_t o_gamma, const float *d_loss, int64_t n_graphs, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(x);
    RWK_REQUIRE_PTR(z);
    RWK_REQUIRE_PTR(gid);
    RWK_REQUIRE_PTR(dx);
    RWK_REQUIRE_PTR(dz_partials);
    RWK_REQUIRE_PTR(k);
    RWK_REQUIRE_PTR(dy);
    RWK_REQUIRE_PTR(mean);
    RWK_REQUIRE_PTR(invstd);
    RWK_REQUIRE_PTR(bn_w);
    RWK_REQUIRE_PTR(head_partials);
    RWK_REQUIRE_PTR(d_loss);
    if (normalize) RWK_REQUIRE_PTR(counts_f32);
    if (!sizes_ok(n, d, h, s, max_step, stream_len) || !(drop_p >= 0.f && drop_p < 1.f)) return RWK_E_SIZE;
    if (!rwk_reg::bn_fold_supported(d, h, s, max_step)) return RWK_E_UNSUPPORTED;
    const int64_t c = max_step * h;
    if (n_graphs <= 0 || n_head_partials <= 0 || n_head_partials > 0x7FFFFFFF || head_nv <= 0 ||
        head_nv > 0x7FFFFFFF || o_gamma < 0 || o_gamma + 2 * c > head_nv)
        return RWK_E_SIZE;
    rwk_reg::Args r{};
    r.x = x; r.z = z; r.gid = gid; r.perm = perm;
    r.n = (int)n; r.d = (int)d; r.h = (int)h; r.s = (int)s; r.max_step = (int)max_step;
    r.stream_len = (int)stream_len; r.drop = drop_p > 0.f && rng != nullptr; r.rng = (const unsigned long long *)rng;
    r.stream = (hipStream_t)stream; r.d_k = nullptr; r.counts_f32 = counts_f32;
    r.scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f; r.normalize = normalize;
    r.dx = dx; r.dz_partials = dz_partials;
    r.bn = rwk_reg::BnFold{k, dy, mean, invstd, bn_w, head_partials, d_loss, (int)n_head_partials, (int)head_nv,
                           (int)o_gamma, 1.f / (float)n_graphs};
    rwk_drop_thresholds(drop_p, &r.t0, &r.t1);
    return rwk_reg::backward(r);
}

int rwk_fused_keep_s_supported(int64_t d, int64_t h, int64_t s, int64_t max_step) {
    return rwk_reg::keep_s_supported(d, h, s, max_step) ? 1 : 0;
}

int rwk_fused_forward_keep_s(const float *x, const float *z, const int32_t *gid, const int32_t *perm, int64_t n,
                             int64_t d, int64_t h, int64_t s, int64_t max_step, int64_t stream_len, float drop_p,
                             const uint64_t *rng, float *k_partials, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(x);
    RWK_REQUIRE_PTR(z);
    RWK_REQUIRE_PTR(gid);
    RWK_REQUIRE_PTR(k_partials);
    if (!sizes_ok(n, d, h, s, max_step, stream_len) || !(drop_p >= 0.f && drop_p < 1.f)) return RWK_E_SIZE;
    if (!rwk_reg::keep_s_supported(d, h, s, max_step)) return RWK_E_UNSUPPORTED;
    rwk_reg::Args r{};
    r.x = x; r.z = z; r.gid = gid; r.perm = perm;
    r.n = (int)n; r.d = (int)d; r.h = (int)h; r.s = (int)s; r.max_step = (int)max_step;
    r.stream_len = (int)stream_len; r.drop = drop_p > 0.f && rng != nullptr; r.rng = (const unsigned long long *)rng;
    r.stream = (hipStream_t)stream; r.k_partials = k_partials; r.keep_s = true;
    rwk_drop_thresholds(drop_p, &r.t0, &r.t1);
    return rwk_reg::forward(r);
}

int rwk_fused_backward_keep_s(const float *x, const float *z, const int32_t *gid, const int32_t *perm,
                              const float *d_k, int64_t n, int64_t d, int64_t h, int64_t s, int64_t max_step,
                              int64_t stream_len, float drop_p, const uint64_t *rng, float *dx, float *dz_partials,
                              rwk_stream_t stream) {
    RWK_REQUIRE_PTR(x);
    RWK_REQUIRE_PTR(z);
    RWK_REQUIRE_PTR(gid);
    RWK_REQUIRE_PTR(d_k);
    RWK_REQUIRE_PTR(dx);
    RWK_REQUIRE_PTR(dz_partials);
    if (!sizes_ok(n, d, h, s, max_step, stream_len) || !(drop_p >= 0.f && drop_p < 1.f)) return RWK_E_SIZE;
    if (!rwk_reg::keep_s_supported(d, h, s, max_step)) return RWK_E_UNSUPPORTED;
    rwk_reg::Args r{};
    r.x = x; r.z = z; r.gid = gid; r.perm = perm;
    r.n = (int)n; r.d = (int)d; r.h = (int)h; r.s = (int)s; r.max_step = (int)max_step;
    r.stream_len = (int)stream_len; r.drop = drop_p > 0.f && rng != nullptr; r.rng = (const unsigned long long *)rng;
    r.stream = (hipStream_t)stream; r.d_k = d_k; r.counts_f32 = nullptr;
    r.scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f; r.normalize = 0;
    r.dx = dx; r.dz_partials = dz_partials; r.keep_s = true;
    rwk_drop_thresholds(drop_p, &r.t0, &r.t1);
    return rwk_reg::backward(r);
}

int rwk_dropout_mask(const uint64_t *rng, int64_t n, int64_t h, int64_t s, int64_t max_step, float drop_p,
                     uint8_t *keep, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(rng);
    RWK_REQUIRE_PTR(keep);
    if (n <= 0 || h <= 0 || s <= 0 || max_step <= 0 || n * h >= 0x7FFFFFFFll * 64) return RWK_E_SIZE;
    DropParams dp;
    rwk_drop_thresholds(drop_p, &dp.t0, &dp.t1);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)rwk_ceil_div(n * h, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const unsigned long long *)rng, (int)n, (int)h, (int)s, (int)max_step,
                       dp, keep);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

int rwk_version(void) { return RWK_VERSION; }

const char *rwk_error_string(int code) {
    switch (code) {
        case RWK_OK: return "ok";
        case RWK_E_NULL: return "a required pointer is null";
        case RWK_E_SIZE: return "a size argument is out of range";
        case RWK_E_UNSUPPORTED: return "hyper-parameters not supported by the kernels (need H <= 64, D <= 64, max_step <= 4)";
        case RWK_E_ALIGN: return "a pointer is misaligned";
        case RWK_E_STATIC_SIZE:
            return "the input exceeds the library's fixed device scratch (rwk_reduce_rows_static: more than "
                   "RWK_REDUCE_ROWS_STATIC_MAX_V columns); use the entry point that takes a workspace";
        case RWK_E_DEVICE: return "the stream belongs to another device than the current one";
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown error";
}

}  // extern "C"
