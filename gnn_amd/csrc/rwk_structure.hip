// Graph-structure kernels: COO inspection, CSR construction, symmetry test, segment pointers.
// Integer work only; every result is independent of thread scheduling (integer atomics are used
// for counters and maxima only).
#include "rwk_common.h"

namespace {

constexpr int kBlock = 256;

__global__ void coo_inspect_kernel(const int64_t *__restrict__ rows, const int64_t *__restrict__ cols,
                                   int64_t nnz, int64_t n, int32_t *flags) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const int64_t r = rows[e], c = cols[e];
    if (r < 0 || r >= n || c < 0 || c >= n) atomicAdd(&flags[1], 1);
    if (e > 0) {
        const int64_t pr = rows[e - 1], pc = cols[e - 1];
        if (r < pr || (r == pr && c <= pc)) atomicAdd(&flags[0], 1);
    }
}

// first index e in [0, nnz) with key[e] >= target
__device__ inline int64_t lower_bound_i64(const int64_t *__restrict__ key, int64_t nnz, int64_t target) {
    int64_t lo = 0, hi = nnz;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (key[mid] < target) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ void csr_from_sorted_coo_kernel(const int64_t *__restrict__ rows, const int64_t *__restrict__ cols,
                                           const float *__restrict__ vals, int64_t nnz, int64_t n,
                                           int32_t *__restrict__ crow, int32_t *__restrict__ col,
                                           float *__restrict__ val) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t <= n) crow[t] = (int32_t)lower_bound_i64(rows, nnz, t);
    if (t < nnz) {
        col[t] = (int32_t)cols[t];
        val[t] = vals[t];
    }
}

__global__ void csr_count_asymmetric_kernel(const int32_t *__restrict__ crow, const int32_t *__restrict__ col,
                                            const float *__restrict__ val, int64_t n, int32_t *flag) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int bad = 0;
    for (int32_t j = crow[r]; j < crow[r + 1]; ++j) {
        const int32_t c = col[j];
        if (c < 0 || (int64_t)c >= n) {      // the caller may not have read the range check's verdict yet
            ++bad;
            continue;
        }
        int32_t lo = crow[c], hi = crow[c + 1];
        while (lo < hi) {
            const int32_t mid = (lo + hi) >> 1;
            if (col[mid] < (int32_t)r) lo = mid + 1; else hi = mid;
        }
        const bool found = lo < crow[c + 1] && col[lo] == (int32_t)r &&
                           __float_as_uint(val[lo]) == __float_as_uint(val[j]);
        bad += found ? 0 : 1;
    }
    if (bad) atomicAdd(flag, bad);
}

__global__ void indicator_inspect_kernel(const int64_t *__restrict__ gi, int64_t n, int32_t *flags) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int64_t g = gi[p];
    if (g < 0) {
        atomicAdd(&flags[2], 1);
    } else {
        const int32_t g32 = g > 0x7FFFFFFF ? 0x7FFFFFFF : (int32_t)g;
        atomicMax(&flags[1], g32);
    }
    if (p > 0) {
        const int64_t prev = gi[p - 1];
        if (g < prev) atomicAdd(&flags[0], 1);
        if (g > prev + 1) atomicAdd(&flags[3], 1);
    } else if (g != 0) {
        atomicAdd(&flags[3], 1);
    }
}

__global__ void segments_build_kernel(const int64_t *__restrict__ ids, int64_t n, int64_t n_graphs,
                                      int32_t *__restrict__ gid32, int32_t *__restrict__ seg_ptr,
                                      int64_t *__restrict__ counts, float *__restrict__ counts_f32,
                                      int32_t *empty) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) gid32[t] = (int32_t)ids[t];
    if (t < n_graphs) {
        const int64_t a = lower_bound_i64(ids, n, t);
        const int64_t b = lower_bound_i64(ids, n, t + 1);
        seg_ptr[t] = (int32_t)a;
        counts[t] = b - a;
        counts_f32[t] = (float)(b - a);
        if (b == a) atomicAdd(empty, 1);
        if (t == n_graphs - 1) seg_ptr[n_graphs] = (int32_t)b;
    }
}

inline bool fits_i32(int64_t v) { return v >= 0 && v < 0x7FFFFFFFll; }

// Batch assembly: node v of the batch is node src_node[v] of the data set; its stored entries are copied to
// positions out_ptr[v].. of the batch's COO list with the column ids moved by col_delta[v] (new first node of
// its graph minus old first node).  Rows keep their stored (column-sorted) order, so the list comes out
// row-major sorted like scipy's lil -> coo conversion.
__global__ void __launch_bounds__(kBlock)
batch_gather_coo_kernel(const int64_t *__restrict__ ds_row_ptr, const int32_t *__restrict__ ds_col,
                        const float *__restrict__ ds_val, const int64_t *__restrict__ src_node,
                        const int64_t *__restrict__ col_delta, const int64_t *__restrict__ out_ptr, int64_t n,
                        int64_t *__restrict__ rows, int64_t *__restrict__ cols, float *__restrict__ vals) {
    const int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (v >= n) return;
    const int64_t sv = src_node[v];
    const int64_t beg = ds_row_ptr[sv], end = ds_row_ptr[sv + 1];
    const int64_t delta = col_delta[v];
    int64_t o = out_ptr[v];
    for (int64_t e = beg; e < end; ++e, ++o) {
        rows[o] = v;
        cols[o] = (int64_t)ds_col[e] + delta;
        vals[o] = ds_val[e];
    }
}

// One launch from a list of graphs of the data set to everything a training step needs of the batch: the COO list
// (the reference's interface), features, graph_indicator, labels, and the CSR / compact columns / segment arrays
// the kernels run on.  A node finds its graph by binary search in the batch's node offsets (the host holds the
// per-graph sizes, so the offsets, n and nnz are known without asking the device); inside a graph the data set's
// own row pointer is already the prefix sum a CSR needs.
struct BatchBuildArgs {
    const int64_t *ds_row_ptr;
    const int32_t *ds_col;
    const float *ds_val;
    const float *ds_feat;
    const int64_t *ds_y;
    const int64_t *table;      // (4, b+1): new node start, data-set node start, new entry start, data-set graph id
    int64_t b, n, nnz, n_feat;
    int64_t *coo_rows, *coo_cols;
    float *coo_vals;
    float *feat;
    int64_t *gi, *y;
    int32_t *crow, *col;
    float *val;
    int16_t *col16;
    int32_t *gid, *seg_ptr;
    int64_t *counts;
    float *counts_f32;
    int64_t node_blocks;
    int feat_vec4;
};

// last g in [lo, hi) with start[g] <= v
__device__ inline int64_t graph_of_node(const int64_t *__restrict__ start, int64_t lo, int64_t hi, int64_t v) {
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (start[mid] <= v) lo = mid; else hi = mid;
    }
    return lo;
}

// The graphs of the nodes v0 .. v1 (of one workgroup) form a range g0 .. g1: the first wavefront finds its ends in
// the table - lanes 0-31 for v0, lanes 32-63 for v1, each half probing 32 evenly spaced entries per round and
// keeping the interval behind the last entry <= v (three rounds for 4096 graphs; a two-way search by one lane made
// twelve dependent trips to the L2, half of what the whole launch took) - the workgroup copies that slice of the
// node offsets to LDS, and every thread then searches its own node there.
constexpr int kSliceMax = 512;

__device__ inline int64_t graph_of_node_block(const int64_t *__restrict__ new_start, int64_t b, int64_t v0, int64_t v1,
                                              int64_t v, bool active, int64_t *slice, int64_t *ends) {
    if (threadIdx.x < 64) {
        const int half = threadIdx.x >> 5, l = threadIdx.x & 31;
        const int64_t want = half ? v1 : v0;
        int64_t lo = 0, hi = b;                              // new_start[lo] <= want < new_start[hi]
        while (hi - lo > 1) {                                // both halves loop until both are done (ballot below)
            const int64_t step = (hi - lo + 31) / 32;
            const int64_t p = lo + l * step;
            const bool ok = p < hi && new_start[p < hi ? p : lo] <= want;      // a prefix of the lanes (lane 0 always)
            const unsigned long long m = __ballot(ok);
            const int cnt = __popc((unsigned)(half ? (m >> 32) : m));
            lo += (int64_t)(cnt - 1) * step;
            hi = lo + step < hi ? lo + step : hi;
        }
        if (l == 0) ends[half] = lo;
    }
    __syncthreads();
    const int64_t g0 = ends[0], g1 = ends[1];
    const int64_t len = g1 - g0 + 1;
    if (len <= kSliceMax) {
        for (int64_t i = threadIdx.x; i < len; i += kBlock) slice[i] = new_start[g0 + i];
        __syncthreads();
        return active ? g0 + graph_of_node(slice, 0, len, v) : 0;
    }
    return active ? graph_of_node(new_start, g0, g1 + 1, v) : 0;
}

__global__ void __launch_bounds__(kBlock) batch_build_kernel(const BatchBuildArgs a) {
    __shared__ int64_t slice[kSliceMax];
    __shared__ int64_t ends[2];
    const int64_t *__restrict__ new_start = a.table;
    const int64_t *__restrict__ old_start = a.table + (a.b + 1);
    const int64_t *__restrict__ new_entry = a.table + 2 * (a.b + 1);
    const int64_t *__restrict__ graph_id = a.table + 3 * (a.b + 1);
    if ((int64_t)blockIdx.x >= a.node_blocks) {
        // wide feature rows: one element (16 bytes when rows are whole float4s) per thread, coalesced along the row
        const int64_t q = a.feat_vec4 ? (a.n_feat >> 2) : a.n_feat;
        const int64_t e0 = ((int64_t)blockIdx.x - a.node_blocks) * kBlock;
        const int64_t e = e0 + threadIdx.x;
        const int64_t total = a.n * q;
        const int64_t e_last = (e0 + kBlock - 1 < total ? e0 + kBlock - 1 : total - 1);
        const bool active = e < total;
        const int64_t v = active ? e / q : 0, k = e - v * q;
        const int64_t g = graph_of_node_block(new_start, a.b, e0 / q, e_last / q, v, active, slice, ends);
        if (!active) return;
        const int64_t src = (old_start[g] + (v - new_start[g])) * q + k;
        if (a.feat_vec4)
            reinterpret_cast<float4 *__restrict__>(a.feat)[e] = reinterpret_cast<const float4 *__restrict__>(a.ds_feat)[src];
        else
            a.feat[e] = a.ds_feat[src];
        return;
    }
    const int64_t t0 = (int64_t)blockIdx.x * kBlock;
    const int64_t t = t0 + threadIdx.x;
    if (t < a.b) {
        const int64_t s0 = new_start[t], s1 = new_start[t + 1];
        if (a.seg_ptr) {
            a.seg_ptr[t] = (int32_t)s0;
            if (t == a.b - 1) a.seg_ptr[a.b] = (int32_t)s1;
        }
        if (a.counts) a.counts[t] = s1 - s0;
        if (a.counts_f32) a.counts_f32[t] = (float)(s1 - s0);
        if (a.y) a.y[t] = a.ds_y[graph_id[t]];
    }
    const bool active = t < a.n;
    const int64_t t_last = t0 + kBlock - 1 < a.n ? t0 + kBlock - 1 : a.n - 1;
    const int64_t g = graph_of_node_block(new_start, a.b, t0, t_last, t, active, slice, ends);
    if (!active) return;
    const int64_t first_new = new_start[g], first_old = old_start[g];
    const int64_t src = first_old + (t - first_new);
    const int64_t *__restrict__ row_ptr = a.ds_row_ptr;
    const int32_t *__restrict__ ds_col = a.ds_col;
    const float *__restrict__ ds_val = a.ds_val;
    const int64_t beg = row_ptr[src], end = row_ptr[src + 1];
    int64_t o = new_entry[g] + (beg - row_ptr[first_old]);
    const int64_t delta = first_new - first_old;
    if (a.crow) {
        a.crow[t] = (int32_t)o;
        if (t == a.n - 1) a.crow[a.n] = (int32_t)(o + (end - beg));
    }
    if (a.gid) a.gid[t] = (int32_t)g;
    if (a.gi) a.gi[t] = g;
    if (a.feat && a.n_feat <= 4)
        for (int64_t k = 0; k < a.n_feat; ++k) a.feat[t * a.n_feat + k] = a.ds_feat[src * a.n_feat + k];
    const int64_t base = t & ~(int64_t)255;
    int64_t *__restrict__ coo_rows = a.coo_rows;
    int64_t *__restrict__ coo_cols = a.coo_cols;
    float *__restrict__ coo_vals = a.coo_vals;
    int32_t *__restrict__ col = a.col;
    float *__restrict__ val = a.val;
    int16_t *__restrict__ col16 = a.col16;
    // four entries of the row per pass, their loads requested together (an entry past the row's end is read at the
    // pass's first index and not stored): one trip to memory per pass instead of one per entry
    const bool want_val = coo_vals != nullptr || val != nullptr;
    for (int64_t e = beg; e < end; e += 4, o += 4) {
        int32_t cc[4];
        float ww[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t eu = e + u < end ? e + u : e;
            cc[u] = ds_col[eu];
            ww[u] = want_val ? ds_val[eu] : 1.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (e + u < end) {
                const int64_t c = (int64_t)cc[u] + delta;
                if (coo_rows) {
                    coo_rows[o + u] = t;
                    coo_cols[o + u] = c;
                    coo_vals[o + u] = ww[u];
                }
                if (col) col[o + u] = (int32_t)c;
                if (val) val[o + u] = ww[u];
                if (col16) col16[o + u] = (int16_t)(c - base);
            }
        }
    }
}

}  // namespace

extern "C" {

int rwk_batch_build(const int64_t *ds_row_ptr, const int32_t *ds_col, const float *ds_val, const float *ds_feat,
                    int64_t n_feat, const int64_t *ds_y, const int64_t *table, int64_t b, int64_t n, int64_t nnz,
                    int64_t *coo_rows, int64_t *coo_cols, float *coo_vals, float *feat, int64_t *gi, int64_t *y,
                    int32_t *crow, int32_t *col, float *val, int16_t *col16, int32_t *gid, int32_t *seg_ptr,
                    int64_t *counts, float *counts_f32, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(ds_row_ptr);
    RWK_REQUIRE_PTR(table);
    if (b <= 0 || n <= 0 || b > n || !fits_i32(n) || !fits_i32(nnz) || n_feat < 0) return RWK_E_SIZE;
    if (nnz > 0) {
        RWK_REQUIRE_PTR(ds_col);
        RWK_REQUIRE_PTR(ds_val);
    }
    if (feat != nullptr) RWK_REQUIRE_PTR(ds_feat);
    if (y != nullptr) RWK_REQUIRE_PTR(ds_y);
    if ((coo_rows == nullptr) != (coo_cols == nullptr) || (coo_rows == nullptr) != (coo_vals == nullptr))
        return RWK_E_NULL;
    BatchBuildArgs a{ds_row_ptr, ds_col, ds_val, ds_feat, ds_y, table, b, n, nnz, n_feat, coo_rows, coo_cols,
                     coo_vals, feat, gi, y, crow, col, val, col16, gid, seg_ptr, counts, counts_f32, 0, 0};
    a.node_blocks = rwk_ceil_div(n, kBlock);
    a.feat_vec4 = (n_feat % 4 == 0) && ((((uintptr_t)feat | (uintptr_t)ds_feat) & 15u) == 0);
    const int64_t feat_blocks =
        (feat != nullptr && n_feat > 4) ? rwk_ceil_div(a.feat_vec4 ? n * (n_feat / 4) : n * n_feat, kBlock) : 0;
    if (a.node_blocks + feat_blocks >= 0x7FFFFFFFll) return RWK_E_SIZE;
    hipLaunchKernelGGL(batch_build_kernel, dim3((unsigned)(a.node_blocks + feat_blocks)), dim3(kBlock), 0,
                       (hipStream_t)stream, a);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

int rwk_batch_gather_coo(const int64_t *ds_row_ptr, const int32_t *ds_col, const float *ds_val,
                         const int64_t *src_node, const int64_t *col_delta, const int64_t *out_ptr, int64_t n,
                         int64_t *rows, int64_t *cols, float *vals, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(ds_row_ptr);
    RWK_REQUIRE_PTR(src_node);
    RWK_REQUIRE_PTR(col_delta);
    RWK_REQUIRE_PTR(out_ptr);
    if (n <= 0 || n >= 0x7FFFFFFFll) return RWK_E_SIZE;
    hipLaunchKernelGGL(batch_gather_coo_kernel, dim3((unsigned)rwk_ceil_div(n, kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, ds_row_ptr, ds_col, ds_val, src_node, col_delta, out_ptr, n, rows, cols,
                       vals);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

int rwk_coo_inspect(const int64_t *rows, const int64_t *cols, int64_t nnz, int64_t n, int32_t *flags,
                    rwk_stream_t stream) {
    RWK_REQUIRE_PTR(flags);
    if (!fits_i32(nnz) || !fits_i32(n) || n == 0) return RWK_E_SIZE;
    if (nnz == 0) return RWK_OK;
    RWK_REQUIRE_PTR(rows);
    RWK_REQUIRE_PTR(cols);
    hipLaunchKernelGGL(coo_inspect_kernel, dim3((unsigned)rwk_ceil_div(nnz, kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, rows, cols, nnz, n, flags);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

int rwk_csr_from_sorted_coo(const int64_t *rows, const int64_t *cols, const float *vals, int64_t nnz,
                            int64_t n, int32_t *crow, int32_t *col, float *val, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(crow);
    if (!fits_i32(nnz) || !fits_i32(n) || n == 0) return RWK_E_SIZE;
    if (nnz > 0) {
        RWK_REQUIRE_PTR(rows);
        RWK_REQUIRE_PTR(cols);
        RWK_REQUIRE_PTR(vals);
        RWK_REQUIRE_PTR(col);
        RWK_REQUIRE_PTR(val);
    }
    const int64_t work = (n + 1) > nnz ? (n + 1) : nnz;
    hipLaunchKernelGGL(csr_from_sorted_coo_kernel, dim3((unsigned)rwk_ceil_div(work, kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, rows, cols, vals, nnz, n, crow, col, val);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

int rwk_csr_count_asymmetric(const int32_t *crow, const int32_t *col, const float *val, int64_t n,
                             int32_t *flag, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(crow);
    RWK_REQUIRE_PTR(flag);
    if (!fits_i32(n) || n == 0) return RWK_E_SIZE;
    hipLaunchKernelGGL(csr_count_asymmetric_kernel, dim3((unsigned)rwk_ceil_div(n, kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, crow, col, val, n, flag);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

int rwk_indicator_inspect(const int64_t *gi, int64_t n, int32_t *flags, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(gi);
    RWK_REQUIRE_PTR(flags);
    if (!fits_i32(n) || n == 0) return RWK_E_SIZE;
    hipLaunchKernelGGL(indicator_inspect_kernel, dim3((unsigned)rwk_ceil_div(n, kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, gi, n, flags);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

int rwk_segments_build(const int64_t *ids_sorted, int64_t n, int64_t n_graphs, int32_t *gid32,
                       int32_t *seg_ptr, int64_t *counts, float *counts_f32, int32_t *empty,
                       rwk_stream_t stream) {
    RWK_REQUIRE_PTR(ids_sorted);
    RWK_REQUIRE_PTR(gid32);
    RWK_REQUIRE_PTR(seg_ptr);
    RWK_REQUIRE_PTR(counts);
    RWK_REQUIRE_PTR(counts_f32);
    RWK_REQUIRE_PTR(empty);
    if (!fits_i32(n) || n == 0 || n_graphs <= 0 || n_graphs > n) return RWK_E_SIZE;
    const int64_t work = n > n_graphs ? n : n_graphs;
    hipLaunchKernelGGL(segments_build_kernel, dim3((unsigned)rwk_ceil_div(work, kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, ids_sorted, n, n_graphs, gid32, seg_ptr, counts, counts_f32, empty);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

}  // extern "C"
