// Distance-biased transition operator of the distance-biased walk models as a sparse operator
// (rwgnn2/rwgnn_dis_bias6.py:109-128 hop distances capped at max_distance, :222-250 transition probabilities):
//
//   P = (1 - c) * rownorm(W) + c / n_g,   W[i][j] = f(dist(i, j)) for i != j, 0 on the diagonal,
//
// per graph of n_g nodes, dist = hop distance with every pair further apart than max_distance (or unconnected) at
// max_distance + 1.  All those far pairs share one weight w_far = f(max_distance + 1), so
//
//   W = w_far * (1 1^T - I) + Y,   Y[i][j] = f(dist) - w_far  for 1 <= dist(i, j) <= max_distance,
//
// a rank-one term plus a sparse matrix over the max_distance-hop neighbourhoods: no dense n_g x n_g matrix.  Here:
// the neighbourhood search that builds Y's structure (one wavefront per source node, visited / frontier / next
// bitmaps of its graph in LDS, level by level; a counting launch and a filling launch), the per-graph column sums
// and the element-wise combination; Y x itself is the CSR SpMM of rwk_spmm.hip.
#include "rwk_common.h"

namespace {

constexpr int kWave = 64;
constexpr int kBlock = 256;

// exclusive prefix sum of v over the 64 lanes; total = sum over all lanes
__device__ inline int wave_exclusive_scan(int v, int lane, int *total) {
    int incl = v;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const int t = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += t;
    }
    *total = __shfl(incl, kWave - 1, kWave);
    return incl - v;
}

// One workgroup of one wavefront per source node i.  Level d: every node of the frontier (nodes at distance d - 1)
// marks its unvisited neighbours in `next`; the set bits of `next` are the nodes at distance d, emitted in
// ascending node order.  Entries of a row are therefore ordered by (distance, node).  Edges that leave the graph
// of i (none in a block-diagonal batch) are ignored.
template <bool FILL>
__global__ void __launch_bounds__(kWave)
khop_kernel(const int *__restrict__ crow, const int *__restrict__ col, const int *__restrict__ gid,
            const int *__restrict__ seg_ptr, int words_cap, int max_distance, int *__restrict__ row_count,
            int *__restrict__ level_count, const int *__restrict__ row_start, int *__restrict__ out_col,
            int *__restrict__ out_dist) {
    extern __shared__ unsigned int khop_bits[];
    unsigned int *visited = khop_bits, *frontier = khop_bits + words_cap, *next = khop_bits + 2 * words_cap;
    const int i = blockIdx.x, lane = threadIdx.x;
    const int g = gid[i];
    const int lo = seg_ptr[g], ng = seg_ptr[g + 1] - lo;
    int words = (ng + 31) >> 5;
    if (words > words_cap) words = words_cap;              // the host sized words_cap for the largest graph
    for (int w = lane; w < words; w += kWave) {
        visited[w] = 0u;
        frontier[w] = 0u;
        next[w] = 0u;
    }
    __syncthreads();
    if (lane == 0) {
        const int b = i - lo;
        if ((unsigned)b < (unsigned)ng && (b >> 5) < words) {
            visited[b >> 5] = 1u << (b & 31);
            frontier[b >> 5] = 1u << (b & 31);
        }
    }
    __syncthreads();
    int total = 0;
    const long long base = FILL ? (long long)row_start[i] : 0;
    int d = 1;
    for (; d <= max_distance; ++d) {
        for (int w = lane; w < words; w += kWave) {
            unsigned int f = frontier[w];
            while (f) {
                const int b = __ffs(f) - 1;
                f &= f - 1u;
                const int u = lo + w * 32 + b;
                const int e1 = crow[u + 1];
                for (int e = crow[u]; e < e1; ++e) {
                    const int v = col[e] - lo;
                    if ((unsigned)v < (unsigned)ng && (v >> 5) < words && !((visited[v >> 5] >> (v & 31)) & 1u))
                        atomicOr(&next[v >> 5], 1u << (v & 31));
                }
            }
        }
        __syncthreads();
        int level = 0;
        for (int w0 = 0; w0 < words; w0 += kWave) {
            const int w = w0 + lane;
            const unsigned int nb = w < words ? next[w] : 0u;
            int chunk;
            const int before = wave_exclusive_scan(__popc(nb), lane, &chunk);
            if (FILL) {
                long long pos = base + total + level + before;
                unsigned int t = nb;
                while (t) {
                    const int b = __ffs(t) - 1;
                    t &= t - 1u;
                    out_col[pos] = lo + w * 32 + b;
                    out_dist[pos] = d;
                    ++pos;
                }
            }
            level += chunk;
            if (w < words) {
                visited[w] |= nb;
                frontier[w] = nb;
                next[w] = 0u;
            }
        }
        if (!FILL && lane == 0) level_count[(size_t)i * max_distance + d - 1] = level;
        total += level;
        __syncthreads();
        if (level == 0) {                                  // same value in every lane: nothing further is reachable
            ++d;
            break;
        }
    }
    if (!FILL) {
        for (int dd = d + lane; dd <= max_distance; dd += kWave) level_count[(size_t)i * max_distance + dd - 1] = 0;
        if (lane == 0) row_count[i] = total;
    }
}

// out[g][c] = sum of x[r][c] over the rows r of graph g, in a fixed order: thread (slot, c) adds rows slot,
// slot + slots, ..., then the slots are added in slot order
__global__ void __launch_bounds__(kBlock)
segment_sum_kernel(const float *__restrict__ x, const int *__restrict__ seg_ptr, int d, int dpad,
                   float *__restrict__ out) {
    __shared__ float red[kBlock];
    const int g = blockIdx.x, tid = threadIdx.x;
    const int lo = seg_ptr[g], hi = seg_ptr[g + 1];
    const int c = tid % dpad, slot = tid / dpad, slots = kBlock / dpad;
    float acc = 0.f;
    if (c < d && lo < hi)
        for (int r0 = lo + slot; r0 < hi; r0 += 8 * slots) {       // eight loads in flight, added in row order
            float a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + u * slots;
                a[u] = x[(size_t)(r < hi ? r : hi - 1) * d + c];
                if (r >= hi) a[u] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += a[u];
        }
    red[tid] = acc;
    __syncthreads();
    if (slot == 0 && c < d) {
        float s = 0.f;
        for (int q = 0; q < slots; ++q) s += red[q * dpad + c];
        out[(size_t)g * d + c] = s;
    }
}

// out[i][c] = a_i * (w_far * (s1[g][c] - x[i][c]) + y[i][c]) + graph_scale[g] * s2[g][c],  g = gid[i], a_i = 1 if
// row_scale is null
__global__ void __launch_bounds__(kBlock)
distance_bias_combine_kernel(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ s1,
                             const float *__restrict__ s2, const int *__restrict__ gid,
                             const float *__restrict__ row_scale, const float *__restrict__ graph_scale, float w_far,
                             long long total, int d, float *__restrict__ out) {
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= total) return;
    const long long i = t / d;
    const int c = (int)(t - i * d);
    const int g = gid[i];
    const float inner = fmaf(w_far, s1[(size_t)g * d + c] - x[t], y[t]);
    const float a = row_scale != nullptr ? row_scale[i] : 1.f;
    out[t] = fmaf(a, inner, graph_scale[g] * s2[(size_t)g * d + c]);
}

__global__ void __launch_bounds__(kBlock)
row_scale_kernel(const float *__restrict__ x, const float *__restrict__ scale, long long total, int d,
                 float *__restrict__ out) {
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= total) return;
    out[t] = x[t] * scale[t / d];
}

inline int pow2_ceil_int(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

constexpr size_t kKhopLdsLimit = 150 * 1024;

template <bool FILL>
int khop_launch(const int *crow, const int *col, const int *gid, const int *seg_ptr, int64_t n,
                int64_t max_graph_nodes, int64_t max_distance, int *row_count, int *level_count, const int *row_start,
                int *out_col, int *out_dist, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(crow);
    RWK_REQUIRE_PTR(gid);
    RWK_REQUIRE_PTR(seg_ptr);
    if (n <= 0 || n >= 0x7FFFFFFFll || max_graph_nodes <= 0 || max_graph_nodes > n || max_distance < 1 ||
        max_distance > 16)
        return RWK_E_SIZE;
    const int words_cap = (int)((max_graph_nodes + 31) / 32);
    const size_t lds = (size_t)3 * words_cap * sizeof(unsigned int);
    if (lds > kKhopLdsLimit) return RWK_E_UNSUPPORTED;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)khop_kernel<FILL>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((khop_kernel<FILL>), dim3((unsigned)n), dim3(kWave), lds, (hipStream_t)stream, crow, col, gid,
                       seg_ptr, words_cap, (int)max_distance, row_count, level_count, row_start, out_col, out_dist);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

}  // namespace

extern "C" {

int rwk_khop_count(const int32_t *crow, const int32_t *col, const int32_t *gid, const int32_t *seg_ptr, int64_t n,
                   int64_t max_graph_nodes, int64_t max_distance, int32_t *row_count, int32_t *level_count,
                   rwk_stream_t stream) {
    RWK_REQUIRE_PTR(row_count);
    RWK_REQUIRE_PTR(level_count);
    return khop_launch<false>(crow, col, gid, seg_ptr, n, max_graph_nodes, max_distance, row_count, level_count,
                              nullptr, nullptr, nullptr, stream);
}

int rwk_khop_fill(const int32_t *crow, const int32_t *col, const int32_t *gid, const int32_t *seg_ptr, int64_t n,
                  int64_t max_graph_nodes, int64_t max_distance, const int32_t *row_start, int32_t *out_col,
                  int32_t *out_dist, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(row_start);
    RWK_REQUIRE_PTR(out_col);
    RWK_REQUIRE_PTR(out_dist);
    return khop_launch<true>(crow, col, gid, seg_ptr, n, max_graph_nodes, max_distance, nullptr, nullptr, row_start,
                             out_col, out_dist, stream);
}

int rwk_segment_sum(const float *x, const int32_t *seg_ptr, int64_t n_graphs, int64_t d, float *out,
                    rwk_stream_t stream) {
    RWK_REQUIRE_PTR(x);
    RWK_REQUIRE_PTR(seg_ptr);
    RWK_REQUIRE_PTR(out);
    if (n_graphs <= 0 || n_graphs >= 0x7FFFFFFFll || d <= 0) return RWK_E_SIZE;
    if (d > kBlock) return RWK_E_UNSUPPORTED;
    hipLaunchKernelGGL(segment_sum_kernel, dim3((unsigned)n_graphs), dim3(kBlock), 0, (hipStream_t)stream, x, seg_ptr,
                       (int)d, pow2_ceil_int((int)d), out);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

int rwk_distance_bias_combine(const float *x, const float *y, const float *s1, const float *s2, const int32_t *gid,
                              const float *row_scale, const float *graph_scale, float w_far, int64_t n, int64_t d,
                              float *out, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(x);
    RWK_REQUIRE_PTR(y);
    RWK_REQUIRE_PTR(s1);
    RWK_REQUIRE_PTR(s2);
    RWK_REQUIRE_PTR(gid);
    RWK_REQUIRE_PTR(graph_scale);
    RWK_REQUIRE_PTR(out);
    if (n <= 0 || d <= 0 || d > 0x7FFFFFFFll || n * d >= 0x7FFFFFFFll * kBlock) return RWK_E_SIZE;
    const long long total = (long long)n * d;
    hipLaunchKernelGGL(distance_bias_combine_kernel, dim3((unsigned)rwk_ceil_div(total, kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, x, y, s1, s2, gid, row_scale, graph_scale, w_far, total, (int)d, out);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

int rwk_row_scale(const float *x, const float *scale, int64_t n, int64_t d, float *out, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(x);
    RWK_REQUIRE_PTR(scale);
    RWK_REQUIRE_PTR(out);
    if (n <= 0 || d <= 0 || d > 0x7FFFFFFFll || n * d >= 0x7FFFFFFFll * kBlock) return RWK_E_SIZE;
    const long long total = (long long)n * d;
    hipLaunchKernelGGL(row_scale_kernel, dim3((unsigned)rwk_ceil_div(total, kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, x, scale, total, (int)d, out);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

}  // extern "C"
