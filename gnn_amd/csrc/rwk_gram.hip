// Random-walk-kernel features WITHOUT dropout (evaluation mode, or dropout 0) through per-graph Gram matrices.
//
// With every product kept, the sum over the nodes of a graph moves inside the contraction (rwgnn2/model.py:45-64
// with the identity mask):
//
//   K[g, i*H+h] = sum_{n in g} sum_s (Z0[h,s].X0[n]) (Zi[h,s].Xi[n])
//               = sum_s Z0[h,s]^T C[g,i] Zi[h,s],      C[g,i] = sum_{n in g} X0[n] (x) Xi[n]     (D x D)
//
// so a node costs D*D multiply-adds per step instead of 2*H*S*D: 16 against 640 at main.py's defaults (16 hidden
// graphs of 5 nodes, hidden_dim 4).  The validation and test loops of the reference run exactly this forward every
// epoch (rwgnn2/main.py:110-117,148-151).  hidden_dim <= 4 (rows padded to four floats).
//
// One workgroup per graph.  Thread t serves node slot q = t / 4 (sixty-four nodes in flight, two rows per lane
// requested before the first is used) and row a = t % 4 of C; the slots are added by a fixed xor butterfly inside a
// wavefront and the four wavefronts in order through LDS, so results are bitwise reproducible.  (One wavefront per
// graph, the first version, spent 25 us on 1024 graphs of 300 nodes: nineteen dependent trips to memory per lane.)
// Backward, given dK:  dC[g,i] = sum_h dK[g,iH+h] sum_s Z0[h,s] (x) Zi[h,s];  dX0[n] += dC[g,i] Xi[n],
// dXi[n] += dC[g,i]^T X0[n]  (one launch);  dM[i,h] = sum_g dK[g,iH+h] C[g,i] and from it dZ (a second launch, one
// workgroup per hidden graph, graphs added in a fixed strided order).
#include "rwk_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / RWK_WAVE;
constexpr int kSlots = kBlock / 4;

__device__ inline float4 load_row4(const float *__restrict__ base, int node, int d) {
    const float *row = base + (size_t)node * d;
    if (d == 4) return *reinterpret_cast<const float4 *>(row);
    float4 v;
    v.x = row[0];
    v.y = d > 1 ? row[1] : 0.f;
    v.z = d > 2 ? row[2] : 0.f;
    v.w = 0.f;
    return v;
}

__device__ inline float pick(const float4 &v, int a) { return a == 0 ? v.x : a == 1 ? v.y : a == 2 ? v.z : v.w; }

// sum over the sixteen node slots of a wavefront (lanes that differ in bits 2..5), fixed order
__device__ inline float slot_sum(float v) {
    v += __shfl_xor(v, 4, RWK_WAVE);
    v += __shfl_xor(v, 8, RWK_WAVE);
    v += __shfl_xor(v, 16, RWK_WAVE);
    v += __shfl_xor(v, 32, RWK_WAVE);
    return v;
}

template <int LT>
__device__ inline void gram_accumulate(const float4 &x0, const float4 (&xi)[LT], int a, float (&acc)[LT][4]) {
    const float x0a = pick(x0, a);
#pragma unroll
    for (int i = 0; i < LT; ++i) {
        const float4 r = i == 0 ? x0 : xi[i];
        acc[i][0] = fmaf(x0a, r.x, acc[i][0]);
        acc[i][1] = fmaf(x0a, r.y, acc[i][1]);
        acc[i][2] = fmaf(x0a, r.z, acc[i][2]);
        acc[i][3] = fmaf(x0a, r.w, acc[i][3]);
    }
}

template <int LT>
__global__ void __launch_bounds__(kBlock)
gram_forward_kernel(const float *__restrict__ x, const float *__restrict__ z, const int *__restrict__ seg_ptr,
                    const int *__restrict__ perm, const float *__restrict__ counts_f32, int n, int d, int h, int s,
                    int normalize, float *__restrict__ k, float *__restrict__ c_out) {
    __shared__ float part[kWaves][LT][16];
    __shared__ float c_lds[LT][16];
    const int g = blockIdx.x;
    const int wave = threadIdx.x / RWK_WAVE, lane = threadIdx.x % RWK_WAVE;
    const int q = threadIdx.x >> 2, a = threadIdx.x & 3;
    const size_t step_stride = (size_t)n * d;
    float acc[LT][4];
#pragma unroll
    for (int i = 0; i < LT; ++i)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[i][b] = 0.f;
    const int p0 = seg_ptr[g], p1 = seg_ptr[g + 1];
    for (int p = p0 + q; p < p1; p += 2 * kSlots) {
        // two nodes per trip, every row requested before the first is used
        const int pb = p + kSlots;
        const bool second = pb < p1;
        const int node_a = perm ? perm[p] : p;
        const int node_b = second ? (perm ? perm[pb] : pb) : node_a;
        float4 xa[LT], xb[LT];
#pragma unroll
        for (int i = 0; i < LT; ++i) {
            xa[i] = load_row4(x + i * step_stride, node_a, d);
            xb[i] = load_row4(x + i * step_stride, node_b, d);
        }
        gram_accumulate<LT>(xa[0], xa, a, acc);
        if (second) gram_accumulate<LT>(xb[0], xb, a, acc);
    }
#pragma unroll
    for (int i = 0; i < LT; ++i)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[i][b] = slot_sum(acc[i][b]);
    if (lane < 4) {
#pragma unroll
        for (int i = 0; i < LT; ++i)
#pragma unroll
            for (int b = 0; b < 4; ++b) part[wave][i][a * 4 + b] = acc[i][b];
    }
    __syncthreads();
    if (threadIdx.x < LT * 16) {
        const int i = threadIdx.x >> 4, ab = threadIdx.x & 15;
        float v = part[0][i][ab];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) v += part[w][i][ab];
        c_lds[i][ab] = v;
        if (c_out) c_out[((size_t)g * LT + i) * 16 + ab] = v;
    }
    __syncthreads();
    const float cnt = normalize ? counts_f32[g] : 1.f;
    const int lh = LT * h;
    for (int idx = threadIdx.x; idx < lh; idx += kBlock) {
        const int i = idx / h, hh = idx - i * h;
        const float *cm = c_lds[i];
        float sum = 0.f;
        for (int ss = 0; ss < s; ++ss) {
            const float4 z0 = load_row4(z, hh * s + ss, d);
            const float4 zi = i == 0 ? z0 : load_row4(z + (size_t)i * h * s * d, hh * s + ss, d);
            float r = 0.f;
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) {
                float t = cm[aa * 4 + 0] * zi.x;
                t = fmaf(cm[aa * 4 + 1], zi.y, t);
                t = fmaf(cm[aa * 4 + 2], zi.z, t);
                t = fmaf(cm[aa * 4 + 3], zi.w, t);
                r = fmaf(pick(z0, aa), t, r);
            }
            sum += r;
        }
        k[(size_t)g * lh + idx] = normalize ? sum / cnt : sum;
    }
}

// dC of the graph and the dX rows of its nodes.
template <int LT>
__global__ void __launch_bounds__(kBlock)
gram_backward_kernel(const float *__restrict__ x, const float *__restrict__ z, const int *__restrict__ seg_ptr,
                     const int *__restrict__ perm, const float *__restrict__ counts_f32,
                     const float *__restrict__ d_k, int n, int d, int h, int s, int normalize,
                     float *__restrict__ dx) {
    constexpr int E = LT * 16;                                  // entries (i, a, b) of dC
    constexpr int GROUPS = kBlock / E;                          // hidden graphs are dealt to this many thread groups
    __shared__ float part[GROUPS][E];
    __shared__ float dc[E];
    const int g = blockIdx.x;
    const int q = threadIdx.x >> 2, a = threadIdx.x & 3;
    const size_t step_stride = (size_t)n * d;
    const int lh = LT * h;
    const float f = normalize ? 1.f / counts_f32[g] : 1.f;
    {
        // dC[i][a][b] = f * sum_h dK[g, iH+h] sum_s Z0[h,s,a] Zi[h,s,b]; thread (group, entry) adds the hidden
        // graphs group, group + GROUPS, ...; the groups are then added in order
        const int e = threadIdx.x % E, grp = threadIdx.x / E;
        if (grp < GROUPS) {
            const int i = e >> 4, aa = (e >> 2) & 3, bb = e & 3;
            const float *z0 = z, *zi = z + (size_t)i * h * s * d;
            float sum = 0.f;
            if (aa < d && bb < d) {
                for (int hh = grp; hh < h; hh += GROUPS) {
                    float m = 0.f;
                    for (int ss = 0; ss < s; ++ss)
                        m = fmaf(z0[((size_t)hh * s + ss) * d + aa], zi[((size_t)hh * s + ss) * d + bb], m);
                    sum = fmaf(d_k[(size_t)g * lh + i * h + hh], m, sum);
                }
            }
            part[grp][e] = sum;
        }
    }
    __syncthreads();
    if (threadIdx.x < E) {
        float v = part[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < GROUPS; ++w) v += part[w][threadIdx.x];
        dc[threadIdx.x] = v * f;
    }
    __syncthreads();
    float row_a[LT][4], col_a[LT][4];                           // dC[i][a][.] and dC[i][.][a]
#pragma unroll
    for (int i = 0; i < LT; ++i)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            row_a[i][b] = dc[i * 16 + a * 4 + b];
            col_a[i][b] = dc[i * 16 + b * 4 + a];
        }
    const int p0 = seg_ptr[g], p1 = seg_ptr[g + 1];
    for (int p = p0 + q; p < p1; p += kSlots) {
        const int node = perm ? perm[p] : p;
        float4 xr[LT];
#pragma unroll
        for (int i = 0; i < LT; ++i) xr[i] = load_row4(x + i * step_stride, node, d);
        const float4 x0 = xr[0];
        // step 0 enters twice (both factors are X0): dC[0] X0 + dC[0]^T X0
        float g0 = row_a[0][0] * x0.x;
        g0 = fmaf(row_a[0][1], x0.y, g0);
        g0 = fmaf(row_a[0][2], x0.z, g0);
        g0 = fmaf(row_a[0][3], x0.w, g0);
        g0 = fmaf(col_a[0][0], x0.x, g0);
        g0 = fmaf(col_a[0][1], x0.y, g0);
        g0 = fmaf(col_a[0][2], x0.z, g0);
        g0 = fmaf(col_a[0][3], x0.w, g0);
#pragma unroll
        for (int i = 1; i < LT; ++i) {
            const float4 xi = xr[i];
            g0 = fmaf(row_a[i][0], xi.x, g0);
            g0 = fmaf(row_a[i][1], xi.y, g0);
            g0 = fmaf(row_a[i][2], xi.z, g0);
            g0 = fmaf(row_a[i][3], xi.w, g0);
            float gi = col_a[i][0] * x0.x;
            gi = fmaf(col_a[i][1], x0.y, gi);
            gi = fmaf(col_a[i][2], x0.z, gi);
            gi = fmaf(col_a[i][3], x0.w, gi);
            if (a < d) dx[i * step_stride + (size_t)node * d + a] = gi;
        }
        if (a < d) dx[(size_t)node * d + a] = g0;
    }
}

// dM[i,h] = sum_g dK'[g, iH+h] C[g,i] and dZ of hidden graph h = blockIdx.x:
//   dZ0[h,s,a] = sum_i sum_b dM[i,h,a,b] Zi[h,s,b] + sum_a' dM[0,h,a',a] Z0[h,s,a'],
//   dZi[h,s,b] = sum_a dM[i,h,a,b] Z0[h,s,a]  (i >= 1).
// Thread (slice, entry) adds the graphs slice, slice + SLICES, ...; the slices are then added in order.
template <int LT>
__global__ void __launch_bounds__(kBlock)
gram_dm_dz_kernel(const float *__restrict__ d_k, const float *__restrict__ c_in, const float *__restrict__ counts_f32,
                  const float *__restrict__ z, int n_graphs, int d, int h, int s, int normalize,
                  float *__restrict__ dz) {
    constexpr int E = LT * 16;
    constexpr int SLICES = kBlock / E;
    __shared__ float part[SLICES][E];
    __shared__ float dm[E];
    const int hh = blockIdx.x;
    const int lh = LT * h;
    {
        const int e = threadIdx.x % E, slice = threadIdx.x / E;
        if (slice < SLICES) {
            const int i = e >> 4, ab = e & 15;
            float sum = 0.f;
            for (int g = slice; g < n_graphs; g += SLICES) {
                float w = d_k[(size_t)g * lh + i * h + hh];
                if (normalize) w *= 1.f / counts_f32[g];
                sum = fmaf(w, c_in[((size_t)g * LT + i) * 16 + ab], sum);
            }
            part[slice][e] = sum;
        }
    }
    __syncthreads();
    if (threadIdx.x < E) {
        float v = part[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < SLICES; ++w) v += part[w][threadIdx.x];
        dm[threadIdx.x] = v;
    }
    __syncthreads();
    const size_t step = (size_t)h * s * d;
    for (int t = threadIdx.x; t < s * d; t += kBlock) {
        const int kk = t % d, ss = t / d;
        const size_t at = ((size_t)hh * s + ss) * d;
        const float *z0 = z + at;
        float g0 = 0.f;
        for (int i = 0; i < LT; ++i) {
            const float *m = dm + i * 16;
            const float *zi = z + i * step + at;
            for (int b = 0; b < d; ++b) g0 = fmaf(m[kk * 4 + b], zi[b], g0);
            float gi = 0.f;
            for (int aa = 0; aa < d; ++aa) gi = fmaf(m[aa * 4 + kk], z0[aa], gi);
            if (i == 0) g0 += gi; else dz[i * step + at + kk] = gi;
        }
        dz[at + kk] = g0;
    }
}

inline bool gram_ok(int64_t d, int64_t h, int64_t s, int64_t l) {
    return d >= 1 && d <= 4 && h >= 1 && h <= 64 && s >= 1 && s <= 64 && l >= 1 && l <= 4;
}

}  // namespace

extern "C" {

int rwk_gram_supported(int64_t d, int64_t h, int64_t s, int64_t max_step) { return gram_ok(d, h, s, max_step) ? 1 : 0; }

int rwk_gram_forward(const float *x, const float *z, const int32_t *seg_ptr, const int32_t *perm,
                     const float *counts_f32, int64_t n, int64_t n_graphs, int64_t d, int64_t h, int64_t s,
                     int64_t max_step, int normalize, float *k, float *c_out, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(x);
    RWK_REQUIRE_PTR(z);
    RWK_REQUIRE_PTR(seg_ptr);
    RWK_REQUIRE_PTR(k);
    if (normalize) RWK_REQUIRE_PTR(counts_f32);
    if (n <= 0 || n >= 0x7FFFFFFFll || n_graphs <= 0 || n_graphs > n) return RWK_E_SIZE;
    if (!gram_ok(d, h, s, max_step)) return RWK_E_UNSUPPORTED;
    if (d == 4 && (((uintptr_t)x | (uintptr_t)z) & 15u)) return RWK_E_ALIGN;
    const dim3 grid((unsigned)n_graphs);
#define RWK_GRAM_FWD(LT)                                                                                           \
    hipLaunchKernelGGL(gram_forward_kernel<LT>, grid, dim3(kBlock), 0, (hipStream_t)stream, x, z, seg_ptr, perm,   \
                       counts_f32, (int)n, (int)d, (int)h, (int)s, normalize, k, c_out)
    switch (max_step) {
        case 1: RWK_GRAM_FWD(1); break;
        case 2: RWK_GRAM_FWD(2); break;
        case 3: RWK_GRAM_FWD(3); break;
        default: RWK_GRAM_FWD(4); break;
    }
#undef RWK_GRAM_FWD
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

int rwk_gram_backward(const float *x, const float *z, const int32_t *seg_ptr, const int32_t *perm,
                      const float *counts_f32, const float *d_k, const float *c_in, int64_t n, int64_t n_graphs,
                      int64_t d, int64_t h, int64_t s, int64_t max_step, int normalize, float *dx, float *dz,
                      rwk_stream_t stream) {
    RWK_REQUIRE_PTR(x);
    RWK_REQUIRE_PTR(z);
    RWK_REQUIRE_PTR(seg_ptr);
    RWK_REQUIRE_PTR(d_k);
    RWK_REQUIRE_PTR(c_in);
    RWK_REQUIRE_PTR(dx);
    RWK_REQUIRE_PTR(dz);
    if (normalize) RWK_REQUIRE_PTR(counts_f32);
    if (n <= 0 || n >= 0x7FFFFFFFll || n_graphs <= 0 || n_graphs > n) return RWK_E_SIZE;
    if (!gram_ok(d, h, s, max_step)) return RWK_E_UNSUPPORTED;
    if (d == 4 && (((uintptr_t)x | (uintptr_t)z) & 15u)) return RWK_E_ALIGN;
#define RWK_GRAM_BWD(LT)                                                                                           \
    do {                                                                                                           \
        hipLaunchKernelGGL(gram_backward_kernel<LT>, dim3((unsigned)n_graphs), dim3(kBlock), 0, (hipStream_t)stream, \
                           x, z, seg_ptr, perm, counts_f32, d_k, (int)n, (int)d, (int)h, (int)s, normalize, dx);   \
        hipLaunchKernelGGL(gram_dm_dz_kernel<LT>, dim3((unsigned)h), dim3(kBlock), 0, (hipStream_t)stream, d_k,    \
                           c_in, counts_f32, z, (int)n_graphs, (int)d, (int)h, (int)s, normalize, dz);             \
    } while (0)
    switch (max_step) {
        case 1: RWK_GRAM_BWD(1); break;
        case 2: RWK_GRAM_BWD(2); break;
        case 3: RWK_GRAM_BWD(3); break;
        default: RWK_GRAM_BWD(4); break;
    }
#undef RWK_GRAM_BWD
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

}  // extern "C"
