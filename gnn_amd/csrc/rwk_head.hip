// Classifier head of RW_NN in four launches instead of ~35:
//   BatchNorm1d(C) -> Linear(C, P) -> ReLU -> Dropout(p) -> Linear(P, NC) -> log_softmax
// (rwgnn2/model.py:68-72 and their autograd backward).  G rows (graphs) of C = max_step * hidden_graphs
// kernel features; everything is tiny next to the random-walk layer, so the kernels are plain vector-ALU code
// whose only aim is to be few launches, deterministic (fixed summation orders, no atomics in this file) and exact
// fp32.
//
//   forward, training:  head_stats_kernel   per-block (count, mean, M2) of each column, merged by the pairwise-variance identity
//                       head_apply_kernel   normalise, both linear layers, dropout, log-softmax; running stats
//   backward:           head_bwd_tile_kernel  d(logits), d(pre-activations), dY; per-block partial parameter sums
//                       head_bwd_finish_kernel  parameter gradients, batch-norm input gradient
//
// Rows are processed in tiles of kRows; a block walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...
#include "rwk_common.h"
#include "rwk_head_partials.h"

namespace {

// Diagnostic build only (-DRWK_STAMP): cycle shares of the segments of head_apply_kernel (slots 0..7) and
// head_bwd_tile_kernel (slots 8..15) and
// head_train_kernel (slots 16..31) for wavefront 0 of workgroup 1.
#ifdef RWK_STAMP
__device__ unsigned long long rwk_head_stamp_buf[32];
#define HEAD_STAMP_DECL() unsigned long long st_last = __builtin_readcyclecounter()
#define HEAD_STAMP_AT(k)                                                               \
    do {                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                             \
        const unsigned long long now_ = __builtin_readcyclecounter();                  \
        __builtin_amdgcn_sched_barrier(0);                                             \
        if (blockIdx.x == 1 && threadIdx.x == 0) rwk_head_stamp_buf[k] = now_ - st_last; \
        st_last = now_;                                                                \
    } while (0)
#else
#define HEAD_STAMP_DECL()
#define HEAD_STAMP_AT(k)
#endif

constexpr int kBlock = 256;
constexpr int kRows = 8;

// i / d and i % d for 0 <= i < 2^22, 1 <= d <= 2^14 in a handful of instructions (a 32-bit integer division by a
// run-time value costs about forty): the float quotient is off by at most one, which the remainder corrects.
struct FastDiv {
    int d;
    float inv;
    __device__ explicit FastDiv(int d_) : d(d_), inv(1.0f / (float)d_) {}
    __device__ inline int div(int i) const {
        int q = (int)((float)i * inv);
        const int r = i - q * d;
        q += (r >= d ? 1 : 0) - (r < 0 ? 1 : 0);
        return q;
    }
    __device__ inline int mod(int i) const { return i - div(i) * d; }
};
// (1 - momentum) * running + momentum * batch, with the rounding steps spelled out so that every kernel that
// updates the running statistics produces the same bits (left to itself the compiler contracts either product)
__device__ inline float running_update(float running, float batch, float momentum) {
    return fmaf(momentum, batch, (1.f - momentum) * running);
}
constexpr int kMaxBlocks = 256;
constexpr int kMaxC = 256, kMaxNC = 32, kMaxPC = 8192;
constexpr uint32_t kHeadCall = 0xFFFFFFFFu;       // Philox counter word no other mask of the model uses

__host__ __device__ inline int head_blocks(long long g) {
    const long long t = (g + kRows - 1) / kRows;
    return (int)(t < kMaxBlocks ? t : kMaxBlocks);
}

// Workgroups of the statistics launch.  Every workgroup of the apply launch merges all of their partials for
// every column (a round trip to L2 per 64 of them and per 32 columns), so there are fewer of them than row
// tiles: the statistics pass itself only reads K once more however it is cut.
__host__ __device__ inline int stats_stride(int c) { return 4 + 2 * c; }   // floats per block partial
constexpr int kMaxStatsBlocks = 64;
constexpr int kMaxFusedStatsBlocks = 128;    // block partials the fused K reduction (kreduce_stats_kernel) leaves
__host__ __device__ inline int head_stats_blocks(long long g) {
    const int b = head_blocks(g);
    return b < kMaxStatsBlocks ? b : kMaxStatsBlocks;
}

// ---------------------------------------------------------------------------------------------------
// batch statistics: partials[b] = {count, 3 unused, mean[C], M2[C]} over the rows of block b's tiles (the padding
// keeps the two vectors 16-byte aligned for widths that are multiples of four)
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
head_stats_kernel(const float *__restrict__ k, long long g, int c, float *__restrict__ partials,
                  unsigned long long *rng_state, unsigned long long *rng_used) {
    __shared__ float red[kBlock];
    __shared__ float s_mean[kMaxC];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0 && rng_state != nullptr && rng_used != nullptr) {
        const unsigned long long seed = rng_state[0], off = rng_state[1];
        rng_used[0] = seed;
        rng_used[1] = off;
        rng_state[1] = off + 1ull;
    }
    const int tpc = kBlock / c;                    // row lanes per column (c <= 256)
    const int col = tid % c, rl = tid / c;
    const bool lane_ok = rl < tpc;
    const long long n_tiles = (g + kRows - 1) / kRows;
    // pass 1: column sums
    float sum = 0.f;
    long long count = 0;
    for (long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const long long r0 = t * kRows;
        const int nr = (int)(g - r0 < kRows ? g - r0 : kRows);
        count += nr;
        if (lane_ok)
            for (int r = rl; r < nr; r += tpc) sum += k[(r0 + r) * c + col];
    }
    red[tid] = lane_ok ? sum : 0.f;
    __syncthreads();
    if (tid < c) {
        float s = 0.f;
        for (int q = 0; q < tpc; ++q) s += red[q * c + tid];
        s_mean[tid] = count > 0 ? s / (float)count : 0.f;
    }
    __syncthreads();
    // pass 2: sum of squared deviations from the block mean
    const float mu = s_mean[col];
    float m2 = 0.f;
    for (long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const long long r0 = t * kRows;
        const int nr = (int)(g - r0 < kRows ? g - r0 : kRows);
        if (lane_ok)
            for (int r = rl; r < nr; r += tpc) {
                const float dv = k[(r0 + r) * c + col] - mu;
                m2 = fmaf(dv, dv, m2);
            }
    }
    red[tid] = lane_ok ? m2 : 0.f;
    __syncthreads();
    float *out = partials + (size_t)blockIdx.x * stats_stride(c);
    if (tid == 0) out[0] = (float)count;
    if (tid < c) {
        float s = 0.f;
        for (int q = 0; q < tpc; ++q) s += red[q * c + tid];
        out[4 + tid] = s_mean[tid];
        out[4 + c + tid] = s;
    }
}

// The per-graph slot reduction of the random-walk layer (K[g, col] = scale * sum of the slots of graph g, in stream
// order: what fused_reduce_kernel computes, added in the same order) and the batch statistics of the K it writes,
// in one launch: block b owns rows [b * rows_per_block, (b + 1) * rows_per_block) and leaves
// partials[b] = {count, mean[C], M2[C]} of them.  A thread owns one column and every (256 / C)-th row of the block;
// it works on two of its rows at a time so that their segment bounds, and then sixteen slots of each, are in flight
// together.  With at most two rows per thread (the usual case) the values stay in registers for the second pass; otherwise
// the second pass reads back what the same thread stored.
constexpr int kKrRows = 2, kKrSlots = 16;

__global__ void __launch_bounds__(kBlock)
kreduce_stats_kernel(const float *__restrict__ k_partials, const int *__restrict__ seg_ptr,
                     const float *__restrict__ counts_f32, long long g, int c, int stream_len, float scale,
                     int normalize, int rows_per_block, float *k, float *__restrict__ partials,
                     unsigned long long *rng_state, unsigned long long *rng_used) {
    __shared__ float red[kBlock];
    __shared__ float s_mean[kMaxC];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0 && rng_state != nullptr && rng_used != nullptr) {
        const unsigned long long seed = rng_state[0], off = rng_state[1];
        rng_used[0] = seed;
        rng_used[1] = off;
        rng_state[1] = off + 1ull;
    }
    const int tpc = kBlock / c;                    // row lanes per column (c <= 256)
    const int col = tid % c, rl = tid / c;
    const bool lane_ok = rl < tpc;
    const long long r_lo = (long long)blockIdx.x * rows_per_block;
    const long long r_hi = r_lo + rows_per_block < g ? r_lo + rows_per_block : g;
    const long long count = r_hi > r_lo ? r_hi - r_lo : 0;
    const int chunk_rows = kKrRows * tpc;
    const bool one_chunk = rows_per_block <= chunk_rows;
    float first_vals[kKrRows] = {0.f, 0.f};
    float sum = 0.f;
    for (long long rb = r_lo; rb < r_hi; rb += chunk_rows) {          // uniform over the workgroup
        long long rows[kKrRows];
        bool ok[kKrRows];
        int cur[kKrRows], last[kKrRows];
#pragma unroll
        for (int q = 0; q < kKrRows; ++q) {
            const long long r = rb + rl + (long long)q * tpc;
            ok[q] = lane_ok && r < r_hi;
            rows[q] = ok[q] ? r : r_lo;                               // a row of this block either way
            cur[q] = seg_ptr[rows[q]] / stream_len;
            last[q] = (seg_ptr[rows[q] + 1] - 1) / stream_len;
        }
        float acc[kKrRows] = {0.f, 0.f};
        bool more = true;
        while (more) {
            // A slot past the row's last one is loaded at a clamped index and enters the sum times 0.0f (a slot of the
            // row times 1.0f: fmaf(a, 1, acc) is acc + a, the same bits).  With a select on the loaded value instead
            // the compiler waited for most of the loads one by one (vmcnt(0) behind each, to reuse its register): a
            // dozen trips to the L2 in a row where one was meant.
            float a[kKrRows][kKrSlots], m[kKrRows][kKrSlots];
#pragma unroll
            for (int q = 0; q < kKrRows; ++q) {
                const float *src = k_partials + (size_t)rows[q] * c + col;
#pragma unroll
                for (int u = 0; u < kKrSlots; ++u) {
                    const int cs = cur[q] + u;
                    m[q][u] = cs <= last[q] ? 1.f : 0.f;
                    a[q][u] = src[(size_t)(cs <= last[q] ? cs : last[q]) * c];
                }
            }
            more = false;
#pragma unroll
            for (int q = 0; q < kKrRows; ++q) {
#pragma unroll
                for (int u = 0; u < kKrSlots; ++u) acc[q] = fmaf(a[q][u], m[q][u], acc[q]);
                cur[q] += kKrSlots;
                more = more || cur[q] <= last[q];
            }
        }
#pragma unroll
        for (int q = 0; q < kKrRows; ++q) {
            float v = acc[q] * scale;
            if (normalize) v /= counts_f32[rows[q]];
            if (ok[q]) {
                k[rows[q] * c + col] = v;
                sum += v;
                if (one_chunk) first_vals[q] = v;
            }
        }
    }
    red[tid] = lane_ok ? sum : 0.f;
    __syncthreads();
    if (tid < c) {
        float sacc = 0.f;
        for (int q = 0; q < tpc; ++q) sacc += red[q * c + tid];
        s_mean[tid] = count > 0 ? sacc / (float)count : 0.f;
    }
    __syncthreads();
    const float mu = s_mean[col];
    float m2 = 0.f;
    if (one_chunk) {
#pragma unroll
        for (int q = 0; q < kKrRows; ++q) {
            const long long r = r_lo + rl + (long long)q * tpc;
            if (lane_ok && r < r_hi) {
                const float dv = first_vals[q] - mu;
                m2 = fmaf(dv, dv, m2);
            }
        }
    } else if (lane_ok) {
        for (long long rb = r_lo + rl; rb < r_hi; rb += chunk_rows) {
            float v[kKrRows];
#pragma unroll
            for (int q = 0; q < kKrRows; ++q) {
                const long long r = rb + (long long)q * tpc;
                v[q] = k[(r < r_hi ? r : r_lo + rl) * c + col];
            }
#pragma unroll
            for (int q = 0; q < kKrRows; ++q) {
                if (rb + (long long)q * tpc < r_hi) {
                    const float dv = v[q] - mu;
                    m2 = fmaf(dv, dv, m2);
                }
            }
        }
    }
    red[tid] = lane_ok ? m2 : 0.f;
    __syncthreads();
    float *out = partials + (size_t)blockIdx.x * stats_stride(c);
    if (tid == 0) out[0] = (float)count;
    if (tid < c) {
        float sacc = 0.f;
        for (int q = 0; q < tpc; ++q) sacc += red[q * c + tid];
        out[4 + tid] = s_mean[tid];
        out[4 + c + tid] = sacc;
    }
}

// The same reduction when a block owns few rows (at most kKrSeqRows: batches of up to 512 graphs).  A row's slots are
// consecutive rows of c floats in k_partials, so the whole workgroup copies them to LDS as one coalesced stream, every
// load in flight at once, and the thread of a column then adds them from LDS in slot order - the same additions in the
// same order as kreduce_stats_kernel, which walks a row's slots sixteen at a time with one thread per (row, column): a
// batch of 64 power-law graphs of 3000 nodes (150 slots per graph) took it ten dependent round trips, 56 us.  The
// block's rows are taken one after the other; the thread of a column keeps their values for the statistics (folded
// into the many-rows kernel, with eight rows per block in sequence, this form was slower on every batch).
constexpr int kKrRowChunk = 8192;            // floats of LDS per chunk of slots
constexpr int kKrRowFlight = 16;             // loads in flight per thread and round
constexpr int kKrSeqRows = 4;                // rows per block up to which this kernel is used

__global__ void __launch_bounds__(kBlock)
kreduce_stats_row_kernel(const float *__restrict__ k_partials, const int *__restrict__ seg_ptr,
                         const float *__restrict__ counts_f32, long long g, int c, int stream_len, float scale,
                         int normalize, int rows_per_block, float *k, float *__restrict__ partials,
                         unsigned long long *rng_state, unsigned long long *rng_used) {
    __shared__ float chunk[kKrRowChunk];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0 && rng_state != nullptr && rng_used != nullptr) {
        const unsigned long long seed = rng_state[0], off = rng_state[1];
        rng_used[0] = seed;
        rng_used[1] = off;
        rng_state[1] = off + 1ull;
    }
    const long long r_lo = (long long)blockIdx.x * rows_per_block;
    const long long r_hi = r_lo + rows_per_block < g ? r_lo + rows_per_block : g;
    if (r_lo >= r_hi) return;
    const int slots_per_chunk = kKrRowChunk / c;
    float vals[kKrSeqRows];
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < kKrSeqRows; ++q) {
        vals[q] = 0.f;
        const long long row = r_lo + q;
        if (row >= r_hi) continue;                                       // uniform over the workgroup
        const int first = seg_ptr[row] / stream_len, last = (seg_ptr[row + 1] - 1) / stream_len;
        const int n_slots = last - first + 1;
        const float *src = k_partials + ((size_t)row + first) * c;     // slot `first` of this row; the next ones follow
        float acc = 0.f;
        for (int s0 = 0; s0 < n_slots; s0 += slots_per_chunk) {          // uniform over the workgroup
            const int n_ch = n_slots - s0 < slots_per_chunk ? n_slots - s0 : slots_per_chunk;
            const int total = n_ch * c;
            const float *from = src + (size_t)s0 * c;
            for (int e0 = tid; e0 < total; e0 += kKrRowFlight * kBlock) {
                float v[kKrRowFlight];
#pragma unroll
                for (int u = 0; u < kKrRowFlight; ++u) {
                    const int e = e0 + u * kBlock;
                    v[u] = from[e < total ? e : total - 1];
                }
#pragma unroll
                for (int u = 0; u < kKrRowFlight; ++u) {
                    const int e = e0 + u * kBlock;
                    if (e < total) chunk[e] = v[u];
                }
            }
            __syncthreads();
            if (tid < c)
                for (int sl = 0; sl < n_ch; ++sl) acc += chunk[sl * c + tid];
            __syncthreads();
        }
        if (tid < c) {
            float v = acc * scale;
            if (normalize) v /= counts_f32[row];
            k[row * c + tid] = v;
            vals[q] = v;
            sum += v;
        }
    }
    if (tid < c) {
        const long long count = r_hi - r_lo;
        const float mu = sum / (float)count;
        float m2 = 0.f;
#pragma unroll
        for (int q = 0; q < kKrSeqRows; ++q) {
            if (r_lo + q < r_hi) {
                const float dv = vals[q] - mu;
                m2 = fmaf(dv, dv, m2);
            }
        }
        float *out = partials + (size_t)blockIdx.x * stats_stride(c);
        if (tid == 0) out[0] = (float)count;
        out[4 + tid] = mu;
        out[4 + c + tid] = m2;
    }
}

// Batch statistics of one column from the per-block partials (count, mean, M2 about the block mean), computed by
// 8 cooperating lanes (lane8 = 0..7 of an aligned group) with the pairwise-variance identity applied once over all
// blocks:  mean = sum_b n_b mean_b / N,  M2 = sum_b (M2_b + n_b (mean_b - mean)^2).  Lane l adds blocks l, l+8, ...
// in order, the eight sums are added in a butterfly whose operand pairs are the same on every lane, so every lane
// (and every workgroup) ends with the same bits.  Two divisions per column; the partials are fetched eight blocks
// at a time (24 loads in flight) - a load per step would cost a round trip to L2 each.
constexpr int kChunk = 16;       // blocks per lane and chunk: 128 blocks (the most the fused K reduction leaves) in one
struct StatsChunk {
    float n[kChunk], mean[kChunk], m2[kChunk];
    __device__ inline void load(const float *__restrict__ partials, int n_blocks, int c, int col, int b0) {
#pragma unroll
        for (int u = 0; u < kChunk; ++u) {
            const int b = b0 + 8 * u;
            const float *p = partials + (size_t)(b < n_blocks ? b : n_blocks - 1) * stats_stride(c);
            n[u] = p[0];
            mean[u] = p[4 + col];
            m2[u] = p[4 + c + col];
            if (b >= n_blocks) {                          // counts as an empty block
                n[u] = 0.f;
                m2[u] = 0.f;
            }
        }
    }
};

// total over an aligned group of 8 lanes, the same bits on each of them (partners at lane distance 1, 2, then the
// other quad: data-parallel-primitive moves, no trip through LDS)
__device__ inline float sum8(float v) { return rwk_group_sum(v, 8); }

__device__ inline float merge_stats8(const float *__restrict__ partials, int n_blocks, int c, int col, int lane8,
                                     const StatsChunk &first, float *var) {
    float sn = 0.f, sm = 0.f;
#pragma unroll
    for (int u = 0; u < kChunk; ++u) {
        sn += first.n[u];
        sm = fmaf(first.n[u], first.mean[u], sm);
    }
    for (int b0 = lane8 + 8 * kChunk; b0 < n_blocks; b0 += 8 * kChunk) {
        StatsChunk next;
        next.load(partials, n_blocks, c, col, b0);
#pragma unroll
        for (int u = 0; u < kChunk; ++u) {
            sn += next.n[u];
            sm = fmaf(next.n[u], next.mean[u], sm);
        }
    }
    sn = sum8(sn);
    sm = sum8(sm);
    const float mean = sn > 0.f ? sm / sn : 0.f;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < kChunk; ++u) {
        const float dv = first.mean[u] - mean;
        q += fmaf(first.n[u] * dv, dv, first.m2[u]);
    }
    for (int b0 = lane8 + 8 * kChunk; b0 < n_blocks; b0 += 8 * kChunk) {
        StatsChunk next;
        next.load(partials, n_blocks, c, col, b0);
#pragma unroll
        for (int u = 0; u < kChunk; ++u) {
            const float dv = next.mean[u] - mean;
            q += fmaf(next.n[u] * dv, dv, next.m2[u]);
        }
    }
    q = sum8(q);
    *var = sn > 0.f ? q / sn : 0.f;
    return mean;
}

// The first kBlock * U elements of a parameter array, fetched into registers so that the loads of several arrays
// are in flight together before the first of them is stored to LDS.
template <int U>
struct ParamChunk {
    float v[U];
    __device__ inline void load(const float *__restrict__ src, int n, int tid) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = tid + u * kBlock;
            v[u] = src[i < n ? i : n - 1];
        }
    }
    // dst[(i / w) * ws + i % w] = src[i]; elements beyond the chunk are copied by a plain loop
    __device__ inline void store_rows(const float *__restrict__ src, int n, const FastDiv &w, int ws, float *dst,
                                      int tid) const {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = tid + u * kBlock;
            const int r = w.div(i);
            if (i < n) dst[r * ws + (i - r * w.d)] = v[u];
        }
        for (int i = tid + U * kBlock; i < n; i += kBlock) {
            const int r = w.div(i);
            dst[r * ws + (i - r * w.d)] = src[i];
        }
    }
    __device__ inline void store_flat(const float *__restrict__ src, int n, float *dst, int tid) const {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = tid + u * kBlock;
            if (i < n) dst[i] = v[u];
        }
        for (int i = tid + U * kBlock; i < n; i += kBlock) dst[i] = src[i];
    }
};

struct HeadDims {
    long long g;
    int c, p, nc;
};

// log sum exp of one row of scores
__device__ inline float row_lse(const float *__restrict__ x, int nc) {
    float mx = x[0];
    for (int q = 1; q < nc; ++q) mx = fmaxf(mx, x[q]);
    float se = 0.f;
    for (int q = 0; q < nc; ++q) se += expf(x[q] - mx);
    return mx + logf(se);
}

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
head_apply_kernel(const float *__restrict__ k, HeadDims dm, const float *__restrict__ bn_w,
                  const float *__restrict__ bn_b, float *running_mean, float *running_var,
                  long long *num_batches_tracked, float momentum, float eps, int training,
                  const float *__restrict__ partials, int n_partials, const float *__restrict__ w1,
                  const float *__restrict__ b1, const float *__restrict__ w2, const float *__restrict__ b2,
                  float drop_p, const unsigned long long *__restrict__ rng, float *__restrict__ mean_out,
                  float *__restrict__ invstd_out, float *__restrict__ h1d_out, float *__restrict__ out,
                  const long long *__restrict__ labels, float *__restrict__ loss_partials) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int c = dm.c, p = dm.p, nc = dm.nc;
    const int cs = c + 1, ps = p + 1;
    float nll = 0.f;                         // labels given: -log p(label) of this thread's rows, over the tiles
    float *s_w1 = lds;                       // [p][cs]
    float *s_w2 = s_w1 + p * cs;             // [nc][ps]
    float *s_b1 = s_w2 + nc * ps;            // [p]
    float *s_b2 = s_b1 + p;                  // [nc]
    float *s_scale = s_b2 + nc;              // [c]
    float *s_shift = s_scale + c;            // [c]
    float *s_y = s_shift + c;                // [kRows][cs]
    float *s_h = s_y + kRows * cs;           // [kRows][ps]
    float *s_o = s_h + kRows * ps;           // [kRows][nc]
    const int tid = threadIdx.x;
    const FastDiv by_c(c), by_p(p), by_nc(nc);
    HEAD_STAMP_DECL();

    // all parameter loads and the first chunk of batch-statistics partials in flight together
    ParamChunk<4> r_w1;
    ParamChunk<1> r_w2, r_b1, r_b2;
    ParamChunk<2> r_k;                                        // first elements of this block's first tile
    r_w1.load(w1, p * c, tid);
    r_w2.load(w2, nc * p, tid);
    r_b1.load(b1, p, tid);
    r_b2.load(b2, nc, tid);
    {
        const long long r0 = (long long)blockIdx.x * kRows;   // blockIdx.x < number of tiles
        const int nr = (int)(dm.g - r0 < kRows ? dm.g - r0 : kRows);
        r_k.load(k + r0 * c, nr * c, tid);
    }
    const int lane8 = tid & 7;
    int c0 = 0;
    do {                                                      // uniform trip count: the shuffles need all lanes
        const int col_raw = c0 + (tid >> 3);
        const int col = col_raw < c ? col_raw : c - 1;
        StatsChunk first;
        if (training) first.load(partials, n_partials, c, col, lane8);
        const float rm = running_mean[col], rv = running_var[col], ga = bn_w[col], be = bn_b[col];
        if (c0 == 0) {
            r_w1.store_rows(w1, p * c, by_c, cs, s_w1, tid);
            r_w2.store_rows(w2, nc * p, by_p, ps, s_w2, tid);
            r_b1.store_flat(b1, p, s_b1, tid);
            r_b2.store_flat(b2, nc, s_b2, tid);
        }
        float mean = rm, var = rv;
        if (training) mean = merge_stats8(partials, n_partials, c, col, lane8, first, &var);
        if (lane8 == 0 && col_raw < c) {
            if (training && blockIdx.x == 0) {
                // running estimates as torch.nn.BatchNorm1d keeps them: unbiased variance, momentum update
                const float unbiased = dm.g > 1 ? var * ((float)dm.g / (float)(dm.g - 1)) : var;
                running_mean[col] = running_update(rm, mean, momentum);
                running_var[col] = running_update(rv, unbiased, momentum);
                if (col == 0 && num_batches_tracked != nullptr) *num_batches_tracked += 1;
            }
            const float invstd = 1.f / sqrtf(var + eps);
            if (blockIdx.x == 0 && mean_out != nullptr) {
                mean_out[col] = mean;
                invstd_out[col] = invstd;
            }
            const float sc = ga * invstd;
            s_scale[col] = sc;
            s_shift[col] = be - mean * sc;
        }
        c0 += kBlock / 8;
    } while (c0 < c);
    __syncthreads();
    HEAD_STAMP_AT(0);

    const bool drop = training && drop_p > 0.f;
    uint32_t seed_lo = 0, key_hi = 0, off_lo = 0, keep_thr = 0xFFFFFFFFu;
    float keep_scale = 1.f;
    if (drop) {
        const unsigned long long seed = rng[0], off = rng[1];
        seed_lo = (uint32_t)seed;
        key_hi = (uint32_t)(seed >> 32) ^ (uint32_t)(off >> 32);
        off_lo = (uint32_t)off;
        const double q = 1.0 - (double)drop_p;
        const double thr = q * 4294967296.0;
        keep_thr = thr >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)thr;
        keep_scale = 1.f / (1.f - drop_p);
    }

    const long long n_tiles = (dm.g + kRows - 1) / kRows;
    for (long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const long long r0 = t * kRows;
        const int nr = (int)(dm.g - r0 < kRows ? dm.g - r0 : kRows);
        const bool first_tile = t == blockIdx.x;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * kBlock;
            const int r = by_c.div(i), col = i - r * c;
            if (first_tile && i < nr * c) s_y[r * cs + col] = fmaf(r_k.v[u], s_scale[col], s_shift[col]);
        }
        for (int i = tid + (first_tile ? 2 * kBlock : 0); i < nr * c; i += kBlock) {
            const int r = by_c.div(i), col = i - r * c;
            s_y[r * cs + col] = fmaf(k[(r0 + r) * c + col], s_scale[col], s_shift[col]);
        }
        __syncthreads();
        HEAD_STAMP_AT(1);
        for (int i = tid; i < nr * p; i += kBlock) {
            const int r = by_p.div(i), j = i - r * p;
            float acc = s_b1[j];
            const float *yr = s_y + r * cs, *wr = s_w1 + j * cs;
#pragma unroll 8
            for (int col = 0; col < c; ++col) acc = fmaf(yr[col], wr[col], acc);
            acc = acc > 0.f ? acc : 0.f;
            if (drop) {
                const RwkPhilox4 rr = rwk_philox4x32_10((uint32_t)(r0 + r), (uint32_t)(j >> 2), kHeadCall, off_lo,
                                                        seed_lo, key_hi);
                const uint32_t wv[4] = {rr.x, rr.y, rr.z, rr.w};
                acc = wv[j & 3] < keep_thr ? acc * keep_scale : 0.f;
            }
            s_h[r * ps + j] = acc;
            if (h1d_out != nullptr) h1d_out[(r0 + r) * p + j] = acc;
        }
        __syncthreads();
        HEAD_STAMP_AT(2);
        for (int i = tid; i < nr * nc; i += kBlock) {
            const int r = by_nc.div(i), q = i - r * nc;
            float acc = s_b2[q];
            const float *hr = s_h + r * ps, *wr = s_w2 + q * ps;
#pragma unroll 8
            for (int j = 0; j < p; ++j) acc = fmaf(hr[j], wr[j], acc);
            s_o[r * nc + q] = acc;
        }
        __syncthreads();
        HEAD_STAMP_AT(3);
        if (tid < nr) {
            const float *o = s_o + tid * nc;
            float mx = o[0];
            for (int q = 1; q < nc; ++q) mx = fmaxf(mx, o[q]);
            float se = 0.f;
            for (int q = 0; q < nc; ++q) se += expf(o[q] - mx);
            const float lse = mx + logf(se);
            for (int q = 0; q < nc; ++q) out[(r0 + tid) * nc + q] = o[q] - lse;
            if (labels != nullptr) {
                // cross-entropy of the stored log-probabilities x = o - lse, as the stand-alone kernel computes it
                float mx2 = o[0] - lse;
                for (int q = 1; q < nc; ++q) mx2 = fmaxf(mx2, o[q] - lse);
                float se2 = 0.f;
                for (int q = 0; q < nc; ++q) se2 += expf((o[q] - lse) - mx2);
                long long cls = labels[r0 + tid];
                cls = cls < 0 ? 0 : (cls >= nc ? nc - 1 : cls);
                nll += (mx2 + logf(se2)) - (o[cls] - lse);
            }
        }
        __syncthreads();
        HEAD_STAMP_AT(4);
    }
    if (labels != nullptr) {                 // this block's share of the loss sum: its kRows row lanes in order
        if (tid < kRows) s_y[tid] = nll;
        __syncthreads();
        if (tid == 0) {
            float acc = 0.f;
            for (int r = 0; r < kRows; ++r) acc += s_y[r];
            loss_partials[blockIdx.x] = acc;
        }
    }
}

// parameter-gradient vector of one block: [dW2 (nc*p) | db2 (nc) | dW1 (p*c) | db1 (p) | dgamma (c) | dbeta (c)]
__host__ __device__ inline int head_nv(int c, int p, int nc) { return nc * p + nc + p * c + p + 2 * c; }

// ---------------------------------------------------------------------------------------------------
// backward, first launch
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
head_bwd_tile_kernel(const float *__restrict__ k, HeadDims dm, const float *__restrict__ bn_w,
                     const float *__restrict__ bn_b, const float *__restrict__ mean, const float *__restrict__ invstd,
                     const float *__restrict__ w1, const float *__restrict__ w2, const float *__restrict__ h1d,
                     const float *__restrict__ out, const float *__restrict__ d_out, float keep_scale,
                     float *__restrict__ dy, float *__restrict__ partials, const long long *__restrict__ labels,
                     const float *__restrict__ d_loss, const float *__restrict__ loss_partials, int n_loss_partials,
                     float *__restrict__ loss_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int c = dm.c, p = dm.p, nc = dm.nc;
    const int cs = c + 1, ps = p + 1;
    const int nv = head_nv(c, p, nc);
    float *s_w1 = lds;                       // [p][cs]
    float *s_w2 = s_w1 + p * cs;             // [nc][ps]
    float *s_xh = s_w2 + nc * ps;            // [kRows][cs]  normalised input
    float *s_dy = s_xh + kRows * cs;         // [kRows][cs]
    float *s_h = s_dy + kRows * cs;          // [kRows][ps]  hidden activations after dropout
    float *s_dp = s_h + kRows * ps;          // [kRows][ps]  gradient of the first layer's pre-activation
    float *s_dl = s_dp + kRows * ps;         // [kRows][nc]  gradient of the logits
    float *s_g = s_dl + kRows * nc;          // [nv]
    float *s_ga = s_g + nv;                  // [c] batch-norm weight
    float *s_be = s_ga + c;                  // [c] batch-norm bias
    const int tid = threadIdx.x;
    const FastDiv by_c(c), by_p(p);
    HEAD_STAMP_DECL();
    // all parameter loads in flight together; their LDS stores follow the first tile's own loads (below), so that
    // one round trip to memory covers both
    ParamChunk<4> r_w1;
    ParamChunk<1> r_w2, r_ga, r_be;
    r_w1.load(w1, p * c, tid);
    r_w2.load(w2, nc * p, tid);
    r_ga.load(bn_w, c, tid);
    r_be.load(bn_b, c, tid);
    for (int i = tid; i < nv; i += kBlock) s_g[i] = 0.f;

    const int o_b2 = nc * p, o_w1 = o_b2 + nc, o_b1 = o_w1 + p * c, o_ga = o_b1 + p, o_be = o_ga + c;
    const long long n_tiles = (dm.g + kRows - 1) / kRows;
    for (long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const long long r0 = t * kRows;
        const int nr = (int)(dm.g - r0 < kRows ? dm.g - r0 : kRows);
        // the tile's loads first (two elements per thread of each array; the rest, for wide layers, in plain
        // loops), then the stores
        ParamChunk<2> t_k, t_h;
        float t_mu[2], t_is[2];
        t_k.load(k + r0 * c, nr * c, tid);
        t_h.load(h1d + r0 * p, nr * p, tid);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * kBlock;
            const int col = by_c.mod(i < nr * c ? i : nr * c - 1);
            t_mu[u] = mean[col];
            t_is[u] = invstd[col];
        }
        if (tid < nr && labels == nullptr) {             // log-softmax backward, one row per thread
            const float *go = d_out + (r0 + tid) * nc, *lo = out + (r0 + tid) * nc;
            float s = 0.f;
            for (int q = 0; q < nc; ++q) s += go[q];
            for (int q = 0; q < nc; ++q) s_dl[tid * nc + q] = go[q] - expf(lo[q]) * s;
        }
        if (tid < nr && labels != nullptr) {
            // the same with d_out = d(mean cross-entropy)/d(out) formed in place: d_loss / G * (softmax(out) - onehot)
            const float *lo = out + (r0 + tid) * nc;
            const float lse = row_lse(lo, nc);
            const float scale = d_loss[0] / (float)dm.g;
            long long cls = labels[r0 + tid];
            cls = cls < 0 ? 0 : (cls >= nc ? nc - 1 : cls);       // as the forward launch does
            float s = 0.f;
            for (int q = 0; q < nc; ++q) s += scale * (expf(lo[q] - lse) - (q == cls ? 1.f : 0.f));
            for (int q = 0; q < nc; ++q) {
                const float go = scale * (expf(lo[q] - lse) - (q == cls ? 1.f : 0.f));
                s_dl[tid * nc + q] = go - expf(lo[q]) * s;
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * kBlock;
            const int rc = by_c.div(i), rp = by_p.div(i);
            if (i < nr * c) s_xh[rc * cs + (i - rc * c)] = (t_k.v[u] - t_mu[u]) * t_is[u];
            if (i < nr * p) s_h[rp * ps + (i - rp * p)] = t_h.v[u];
        }
        for (int i = tid + 2 * kBlock; i < nr * c; i += kBlock) {
            const int r = by_c.div(i), col = i - r * c;
            s_xh[r * cs + col] = (k[(r0 + r) * c + col] - mean[col]) * invstd[col];
        }
        for (int i = tid + 2 * kBlock; i < nr * p; i += kBlock) {
            const int r = by_p.div(i);
            s_h[r * ps + (i - r * p)] = h1d[r0 * p + i];
        }
        if (t == blockIdx.x) {
            r_w1.store_rows(w1, p * c, by_c, cs, s_w1, tid);
            r_w2.store_rows(w2, nc * p, by_p, ps, s_w2, tid);
            r_ga.store_flat(bn_w, c, s_ga, tid);
            r_be.store_flat(bn_b, c, s_be, tid);
        }
        __syncthreads();
        HEAD_STAMP_AT(8);
        for (int i = tid; i < nr * p; i += kBlock) {
            const int r = by_p.div(i), j = i - r * p;
            float acc = 0.f;
#pragma unroll 8
            for (int q = 0; q < nc; ++q) acc = fmaf(s_dl[r * nc + q], s_w2[q * ps + j], acc);
            s_dp[r * ps + j] = s_h[r * ps + j] > 0.f ? acc * keep_scale : 0.f;
        }
        __syncthreads();
        HEAD_STAMP_AT(9);
        for (int i = tid; i < nr * c; i += kBlock) {
            const int r = by_c.div(i), col = i - r * c;
            float acc = 0.f;
#pragma unroll 8
            for (int j = 0; j < p; ++j) acc = fmaf(s_dp[r * ps + j], s_w1[j * cs + col], acc);
            s_dy[r * cs + col] = acc;
            dy[(r0 + r) * c + col] = acc;
        }
        __syncthreads();
        HEAD_STAMP_AT(10);
        // every thread owns the gradient elements e = tid, tid + 256, ...: rows are added in row order
        for (int e = tid; e < nv; e += kBlock) {
            float acc = 0.f;
            if (e < o_b2) {
                const int q = by_p.div(e), j = e - q * p;
#pragma unroll 8
                for (int r = 0; r < nr; ++r) acc = fmaf(s_dl[r * nc + q], s_h[r * ps + j], acc);
            } else if (e < o_w1) {
                const int q = e - o_b2;
#pragma unroll 8
                for (int r = 0; r < nr; ++r) acc += s_dl[r * nc + q];
            } else if (e < o_b1) {
                const int j = by_c.div(e - o_w1), col = (e - o_w1) - j * c;
                const float ga = s_ga[col], be = s_be[col];
#pragma unroll 8
                for (int r = 0; r < nr; ++r) acc = fmaf(s_dp[r * ps + j], fmaf(s_xh[r * cs + col], ga, be), acc);
            } else if (e < o_ga) {
                const int j = e - o_b1;
#pragma unroll 8
                for (int r = 0; r < nr; ++r) acc += s_dp[r * ps + j];
            } else if (e < o_be) {
                const int col = e - o_ga;
#pragma unroll 8
                for (int r = 0; r < nr; ++r) acc = fmaf(s_dy[r * cs + col], s_xh[r * cs + col], acc);
            } else {
                const int col = e - o_be;
#pragma unroll 8
                for (int r = 0; r < nr; ++r) acc += s_dy[r * cs + col];
            }
            s_g[e] += acc;
        }
        __syncthreads();
        HEAD_STAMP_AT(11);
    }
    float *dst = partials + (size_t)blockIdx.x * nv;
    for (int e = tid; e < nv; e += kBlock) dst[e] = s_g[e];
    HEAD_STAMP_AT(12);
    if (loss_out != nullptr && blockIdx.x == 0 && tid < 64) {
        // mean loss from the forward launch's per-block sums: lane l adds blocks l, l + 64, ... in order (at most
        // kMaxBlocks = 256: four loads in flight), then a butterfly whose operand pairs are fixed
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int b = tid + 64 * u;
            v[u] = loss_partials[b < n_loss_partials ? b : n_loss_partials - 1];
            if (b >= n_loss_partials) v[u] = 0.f;
        }
        float acc = ((v[0] + v[1]) + v[2]) + v[3];
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (tid == 0) loss_out[0] = acc / (float)dm.g;
    }
}

// ---------------------------------------------------------------------------------------------------
// training step: the forward pass and the per-tile part of the backward pass in one launch
// ---------------------------------------------------------------------------------------------------
// When the head's output goes straight into the mean cross-entropy and a backward pass will follow (the training
// step), everything the backward pass does per tile of rows - d(logits) from the log-probabilities and the labels,
// d(pre-activations), dY, this block's partial parameter sums - needs nothing that the forward pass of the same tile
// has not just left in LDS.  Doing it there saves a launch, the re-load of the parameters and of the tile, and the
// store and re-load of the hidden activations.  The gradient of the loss value itself (d_loss, known only when the
// backward pass runs) is a common factor: everything here is computed for d_loss = 1 and head_bwd_finish_kernel
// multiplies by it.  Expressions and orders of addition are those of head_apply_kernel and head_bwd_tile_kernel.
__global__ void __launch_bounds__(kBlock)
head_train_kernel(const float *__restrict__ k, HeadDims dm, const float *__restrict__ bn_w,
                  const float *__restrict__ bn_b, float *running_mean, float *running_var,
                  long long *num_batches_tracked, float momentum, float eps, const float *__restrict__ partials,
                  int n_partials, const float *__restrict__ w1, const float *__restrict__ b1,
                  const float *__restrict__ w2, const float *__restrict__ b2, float drop_p,
                  const unsigned long long *__restrict__ rng, float *__restrict__ mean_out,
                  float *__restrict__ invstd_out, float *__restrict__ out, const long long *__restrict__ labels,
                  float *__restrict__ loss_partials, float *__restrict__ dy, float *__restrict__ grad_partials,
                  float *__restrict__ loss_pending) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int c = dm.c, p = dm.p, nc = dm.nc;
    const int cs = c + 1, ps = p + 1;
    const int nv = head_nv(c, p, nc);
    float nll = 0.f;
    HEAD_STAMP_DECL();
    // the loss value is finished by the backward launch: until then it reads as not-a-number, never as stale memory
    if (loss_pending != nullptr && blockIdx.x == 0 && threadIdx.x == 0) loss_pending[0] = __builtin_nanf("");
    float *s_w1 = lds;                       // [p][cs]
    float *s_w2 = s_w1 + p * cs;             // [nc][ps]
    float *s_b1 = s_w2 + nc * ps;            // [p]
    float *s_b2 = s_b1 + p;                  // [nc]
    float *s_scale = s_b2 + nc;              // [c]  gamma * invstd
    float *s_shift = s_scale + c;            // [c]  beta - mean * gamma * invstd
    float *s_mu = s_shift + c;               // [c]
    float *s_is = s_mu + c;                  // [c]
    float *s_ga = s_is + c;                  // [c]
    float *s_be = s_ga + c;                  // [c]
    float *s_y = s_be + c;                   // [kRows][cs]  batch-norm output, later dY
    float *s_xh = s_y + kRows * cs;          // [kRows][cs]  normalised input
    float *s_h = s_xh + kRows * cs;          // [kRows][ps]  hidden activations after dropout
    float *s_dp = s_h + kRows * ps;          // [kRows][ps]  gradient of the first layer's pre-activation
    float *s_o = s_dp + kRows * ps;          // [kRows][nc]  logits
    float *s_dl = s_o + kRows * nc;          // [kRows][nc]  gradient of the logits
    float *s_g = s_dl + kRows * nc;          // [nv]
    float *s_one = s_g + nv;                 // [1]   the constant 1.0f (second factor of the plain column sums)
    const int tid = threadIdx.x;
    const FastDiv by_c(c), by_p(p), by_nc(nc);

    ParamChunk<4> r_w1;
    ParamChunk<1> r_w2, r_b1, r_b2;
    ParamChunk<2> r_k;
    r_w1.load(w1, p * c, tid);
    r_w2.load(w2, nc * p, tid);
    r_b1.load(b1, p, tid);
    r_b2.load(b2, nc, tid);
    {
        const long long r0 = (long long)blockIdx.x * kRows;
        const int nr = (int)(dm.g - r0 < kRows ? dm.g - r0 : kRows);
        r_k.load(k + r0 * c, nr * c, tid);
    }
    // generator state and the first tile's labels: fetched with the parameters, so that neither costs a round trip of
    // its own in the middle of the tile
    const bool drop = drop_p > 0.f;
    unsigned long long rng_seed = 0, rng_off = 0;
    if (drop) {
        rng_seed = rng[0];
        rng_off = rng[1];
    }
    long long first_label = 0;
    {
        const long long r = (long long)blockIdx.x * kRows + tid;
        if (tid < kRows && r < dm.g) first_label = labels[r];
    }
    for (int i = tid; i < nv; i += kBlock) s_g[i] = 0.f;
    if (tid == 0) s_one[0] = 1.f;
    HEAD_STAMP_AT(25);
    const int lane8 = tid & 7;
    int c0 = 0;
    do {                                                      // uniform trip count: the shuffles need all lanes
        const int col_raw = c0 + (tid >> 3);
        const int col = col_raw < c ? col_raw : c - 1;
        StatsChunk first;
        first.load(partials, n_partials, c, col, lane8);
        const float rm = running_mean[col], rv = running_var[col], ga = bn_w[col], be = bn_b[col];
        HEAD_STAMP_AT(26);
        if (c0 == 0) {
            r_w1.store_rows(w1, p * c, by_c, cs, s_w1, tid);
            r_w2.store_rows(w2, nc * p, by_p, ps, s_w2, tid);
            r_b1.store_flat(b1, p, s_b1, tid);
            r_b2.store_flat(b2, nc, s_b2, tid);
        }
        HEAD_STAMP_AT(27);
        float var;
        const float mean = merge_stats8(partials, n_partials, c, col, lane8, first, &var);
        HEAD_STAMP_AT(28);
        if (lane8 == 0 && col_raw < c) {
            if (blockIdx.x == 0) {
                const float unbiased = dm.g > 1 ? var * ((float)dm.g / (float)(dm.g - 1)) : var;
                running_mean[col] = running_update(rm, mean, momentum);
                running_var[col] = running_update(rv, unbiased, momentum);
                if (col == 0 && num_batches_tracked != nullptr) *num_batches_tracked += 1;
            }
            const float invstd = 1.f / sqrtf(var + eps);
            if (blockIdx.x == 0) {
                mean_out[col] = mean;
                invstd_out[col] = invstd;
            }
            const float sc = ga * invstd;
            s_scale[col] = sc;
            s_shift[col] = be - mean * sc;
            s_mu[col] = mean;
            s_is[col] = invstd;
            s_ga[col] = ga;
            s_be[col] = be;
        }
        c0 += kBlock / 8;
    } while (c0 < c);
    HEAD_STAMP_AT(29);
    __syncthreads();
    HEAD_STAMP_AT(16);

    uint32_t seed_lo = 0, key_hi = 0, off_lo = 0, keep_thr = 0xFFFFFFFFu;
    float keep_scale = 1.f;
    if (drop) {
        const unsigned long long seed = rng_seed, off = rng_off;
        seed_lo = (uint32_t)seed;
        key_hi = (uint32_t)(seed >> 32) ^ (uint32_t)(off >> 32);
        off_lo = (uint32_t)off;
        const double q = 1.0 - (double)drop_p;
        const double thr = q * 4294967296.0;
        keep_thr = thr >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)thr;
        keep_scale = 1.f / (1.f - drop_p);
    }
    const int o_b2 = nc * p, o_w1 = o_b2 + nc, o_b1 = o_w1 + p * c, o_ga = o_b1 + p, o_be = o_ga + c;
    const float inv_g = 1.f / (float)dm.g;          // d(mean loss) / d(row loss), for d_loss = 1

    const long long n_tiles = (dm.g + kRows - 1) / kRows;
    for (long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const long long r0 = t * kRows;
        const int nr = (int)(dm.g - r0 < kRows ? dm.g - r0 : kRows);
        const bool first_tile = t == blockIdx.x;
        if (nr < kRows) {                                     // uniform: the parameter sums below run over all kRows rows
            for (int i = tid + nr * cs; i < kRows * cs; i += kBlock) s_y[i] = s_xh[i] = 0.f;
            for (int i = tid + nr * ps; i < kRows * ps; i += kBlock) s_h[i] = s_dp[i] = 0.f;
            for (int i = tid + nr * nc; i < kRows * nc; i += kBlock) s_dl[i] = 0.f;
        }
        // ---- forward (head_apply_kernel) ----
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * kBlock;
            const int r = by_c.div(i), col = i - r * c;
            if (first_tile && i < nr * c) {
                s_y[r * cs + col] = fmaf(r_k.v[u], s_scale[col], s_shift[col]);
                s_xh[r * cs + col] = (r_k.v[u] - s_mu[col]) * s_is[col];
            }
        }
        for (int i = tid + (first_tile ? 2 * kBlock : 0); i < nr * c; i += kBlock) {
            const int r = by_c.div(i), col = i - r * c;
            const float kv = k[(r0 + r) * c + col];
            s_y[r * cs + col] = fmaf(kv, s_scale[col], s_shift[col]);
            s_xh[r * cs + col] = (kv - s_mu[col]) * s_is[col];
        }
        __syncthreads();
        HEAD_STAMP_AT(17);
        for (int i = tid; i < nr * p; i += kBlock) {
            const int r = by_p.div(i), j = i - r * p;
            float acc = s_b1[j];
            const float *yr = s_y + r * cs, *wr = s_w1 + j * cs;
#pragma unroll 8
            for (int col = 0; col < c; ++col) acc = fmaf(yr[col], wr[col], acc);
            acc = acc > 0.f ? acc : 0.f;
            if (drop) {
                const RwkPhilox4 rr = rwk_philox4x32_10((uint32_t)(r0 + r), (uint32_t)(j >> 2), kHeadCall, off_lo,
                                                        seed_lo, key_hi);
                const uint32_t wv[4] = {rr.x, rr.y, rr.z, rr.w};
                acc = wv[j & 3] < keep_thr ? acc * keep_scale : 0.f;
            }
            s_h[r * ps + j] = acc;
        }
        __syncthreads();
        HEAD_STAMP_AT(18);
        for (int i = tid; i < nr * nc; i += kBlock) {
            const int r = by_nc.div(i), q = i - r * nc;
            float acc = s_b2[q];
            const float *hr = s_h + r * ps, *wr = s_w2 + q * ps;
#pragma unroll 8
            for (int j = 0; j < p; ++j) acc = fmaf(hr[j], wr[j], acc);
            s_o[r * nc + q] = acc;
        }
        __syncthreads();
        HEAD_STAMP_AT(19);
        if (tid < nr) {
            float *o = s_o + tid * nc;
            float mx = o[0];
            for (int q = 1; q < nc; ++q) mx = fmaxf(mx, o[q]);
            float se = 0.f;
            for (int q = 0; q < nc; ++q) se += expf(o[q] - mx);
            const float lse = mx + logf(se);
            for (int q = 0; q < nc; ++q) {
                o[q] -= lse;                                 // the stored log-probabilities
                out[(r0 + tid) * nc + q] = o[q];
            }
            // cross-entropy of the stored log-probabilities and its gradient (head_bwd_tile_kernel, d_loss = 1)
            const float lse2 = row_lse(o, nc);
            long long cls = first_tile ? first_label : labels[r0 + tid];
            cls = cls < 0 ? 0 : (cls >= nc ? nc - 1 : cls);  // the host side has checked the range (see head.py)
            nll += lse2 - o[cls];
            float sgo = 0.f;
            for (int q = 0; q < nc; ++q) sgo += inv_g * (expf(o[q] - lse2) - (q == cls ? 1.f : 0.f));
            for (int q = 0; q < nc; ++q) {
                const float go = inv_g * (expf(o[q] - lse2) - (q == cls ? 1.f : 0.f));
                s_dl[tid * nc + q] = go - expf(o[q]) * sgo;
            }
        }
        __syncthreads();
        HEAD_STAMP_AT(20);
        // ---- backward, per-tile part (head_bwd_tile_kernel) ----
        for (int i = tid; i < nr * p; i += kBlock) {
            const int r = by_p.div(i), j = i - r * p;
            float acc = 0.f;
#pragma unroll 8
            for (int q = 0; q < nc; ++q) acc = fmaf(s_dl[r * nc + q], s_w2[q * ps + j], acc);
            s_dp[r * ps + j] = s_h[r * ps + j] > 0.f ? acc * keep_scale : 0.f;
        }
        __syncthreads();
        HEAD_STAMP_AT(21);
        for (int i = tid; i < nr * c; i += kBlock) {
            const int r = by_c.div(i), col = i - r * c;
            float acc = 0.f;
#pragma unroll 8
            for (int j = 0; j < p; ++j) acc = fmaf(s_dp[r * ps + j], s_w1[j * cs + col], acc);
            s_y[r * cs + col] = acc;                         // dY: the batch-norm output is no longer needed
            dy[(r0 + r) * c + col] = acc;
        }
        __syncthreads();
        HEAD_STAMP_AT(22);
        // this tile's share of the parameter sums.  Every element is sum_r A[r] * (B[r] * ga + be) over the tile's rows
        // (plain column sums take B = 1): four elements at a time, their 64 LDS reads in flight together, rows in order.
        // d w1 (p x c of the nv elements) in register tiles of 2 rows x 4 columns of w1: the eight values of each of
        // the two d(pre-activation) columns and of the four normalised-input columns are read once for eight
        // elements (6 LDS reads per element instead of 16; this phase was 30-45 % of the kernel at 64 and 128
        // columns).  Same products, added in the same (row) order: same bits as the element-wise form below.
        const bool w1_tiled = (p & 1) == 0 && (c & 3) == 0;
        if (w1_tiled) {
            const int cq = c >> 2, n_tiles = (p >> 1) * cq;
            const FastDiv by_cq(cq);
            for (int q = tid; q < n_tiles; q += kBlock) {
                const int j2 = by_cq.div(q), j0 = 2 * j2, c0 = 4 * (q - j2 * cq);
                float dpv[2][kRows], yv[4][kRows];
#pragma unroll
                for (int r = 0; r < kRows; ++r) {
                    dpv[0][r] = s_dp[r * ps + j0];
                    dpv[1][r] = s_dp[r * ps + j0 + 1];
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) yv[cc][r] = s_xh[r * cs + c0 + cc];
                }
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const float ga = s_ga[c0 + cc], be = s_be[c0 + cc];
                    float a0 = 0.f, a1 = 0.f;
#pragma unroll
                    for (int r = 0; r < kRows; ++r) {
                        const float y = fmaf(yv[cc][r], ga, be);
                        a0 = fmaf(dpv[0][r], y, a0);
                        a1 = fmaf(dpv[1][r], y, a1);
                    }
                    s_g[o_w1 + j0 * c + c0 + cc] += a0;
                    s_g[o_w1 + (j0 + 1) * c + c0 + cc] += a1;
                }
            }
        }
        const int w1_skip = w1_tiled ? p * c : 0;            // the remaining elements, numbered without the d w1 block
        const int nv_rest = nv - w1_skip;
        for (int e0 = tid; e0 < nv_rest; e0 += 4 * kBlock) {
            // operands as offsets into the workgroup's LDS block (pointers chosen at run time among several arrays
            // would lose their address space and be read through the flat path)
            int oa[4], ob[4], sa[4], sb[4];
            float ga4[4], be4[4], acc[4];
            const int f_dl = (int)(s_dl - lds), f_h = (int)(s_h - lds), f_one = (int)(s_one - lds),
                      f_dp = (int)(s_dp - lds), f_xh = (int)(s_xh - lds), f_y = (int)(s_y - lds);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i_raw = e0 + u * kBlock;
                const int i_c = i_raw < nv_rest ? i_raw : nv_rest - 1;
                const int e = i_c < o_w1 ? i_c : i_c + w1_skip;
                ga4[u] = 1.f; be4[u] = 0.f; acc[u] = 0.f;
                if (e < o_b2) {                               // d w2[q][j] = sum_r dl[r][q] h[r][j]
                    const int q = by_p.div(e), j = e - q * p;
                    oa[u] = f_dl + q; sa[u] = nc; ob[u] = f_h + j; sb[u] = ps;
                } else if (e < o_w1) {                        // d b2[q]
                    oa[u] = f_dl + (e - o_b2); sa[u] = nc; ob[u] = f_one; sb[u] = 0;
                } else if (e < o_b1) {                        // d w1[j][col] = sum_r dp[r][j] y[r][col]
                    const int j = by_c.div(e - o_w1), col = (e - o_w1) - j * c;
                    oa[u] = f_dp + j; sa[u] = ps; ob[u] = f_xh + col; sb[u] = cs;
                    ga4[u] = s_ga[col]; be4[u] = s_be[col];
                } else if (e < o_ga) {                        // d b1[j]
                    oa[u] = f_dp + (e - o_b1); sa[u] = ps; ob[u] = f_one; sb[u] = 0;
                } else if (e < o_be) {                        // d gamma[col] = sum_r dY[r][col] xhat[r][col]
                    const int col = e - o_ga;
                    oa[u] = f_y + col; sa[u] = cs; ob[u] = f_xh + col; sb[u] = cs;
                } else {                                      // d beta[col]
                    oa[u] = f_y + (e - o_be); sa[u] = cs; ob[u] = f_one; sb[u] = 0;
                }
            }
            float av[4][kRows], bv[4][kRows];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < kRows; ++r) {
                    av[u][r] = lds[oa[u] + r * sa[u]];
                    bv[u][r] = lds[ob[u] + r * sb[u]];
                }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int r = 0; r < kRows; ++r) acc[u] = fmaf(av[u][r], fmaf(bv[u][r], ga4[u], be4[u]), acc[u]);
                const int i = e0 + u * kBlock;
                if (i < nv_rest) s_g[i < o_w1 ? i : i + w1_skip] += acc[u];
            }
        }
        __syncthreads();
        HEAD_STAMP_AT(23);
    }
    float *dst = grad_partials + (size_t)blockIdx.x * nv;
    for (int e = tid; e < nv; e += kBlock) dst[e] = s_g[e];
    if (tid < kRows) s_o[tid] = nll;                         // this block's share of the loss sum, row lanes in order
    __syncthreads();
    if (tid == 0) {
        float acc = 0.f;
        for (int r = 0; r < kRows; ++r) acc += s_o[r];
        loss_partials[blockIdx.x] = acc;
    }
    HEAD_STAMP_AT(24);
}

// ---------------------------------------------------------------------------------------------------
// training step at the reference's default widths (32 features = 16 hidden graphs x 2 steps, penultimate_dim 32,
// at most kT32MaxNC classes): one wavefront per tile of 8 rows and no barrier-separated phases.
//
// head_train_kernel above spends its time between phases (nine barriers per tile, every operand through LDS and a
// run-time index split per element: 31 k cycles for 8 x 32 x 32 multiply-adds each way).  Here lane (r, s) =
// (l >> 3, l & 7) owns columns 4s..4s+3 of row r of the tile and keeps in registers the two slices of fc1's weight
// it needs (all 32 rows x its 4 columns for the forward product, its 4 rows x all 32 columns for dY), the row's
// values travel between lanes by data-parallel-primitive moves, and only the sums over the tile's rows go through
// LDS (after one wait, no barrier: a workgroup is one wavefront).  Same expressions as head_train_kernel; the inner
// products are added in another (fixed) order: per lane over its 4 columns, then over the 8 lanes.
// ---------------------------------------------------------------------------------------------------
constexpr int kT32 = 32, kT32MaxNC = 4, kT32Threads = 64;

// Sums of 32 per-lane values over the 8 lanes of an aligned group, scattered: lane s ends with the totals of values
// 4s..4s+3.  Three halving steps - partner 7-s, then 3-s within the quad, then s^1 - in each of which a lane keeps
// the half its final values lie in and adds the partner's copy of that half: 28 additions for 32 sums.
__device__ inline void reduce_scatter8(const float (&v)[32], int s, float (&out)[4]) {
    const bool up_a = s >= 4, up_b = (s & 3) >= 2, up_c = (s & 1) != 0;
    float a[16], b[8];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const float keep = up_a ? v[m + 16] : v[m], give = up_a ? v[m] : v[m + 16];
        a[m] = keep + rwk_dpp<0x141>(give);              // row_half_mirror
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const float keep = up_b ? a[m + 8] : a[m], give = up_b ? a[m] : a[m + 8];
        b[m] = keep + rwk_dpp<0x1B>(give);               // quad_perm [3,2,1,0]
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float keep = up_c ? b[m + 4] : b[m], give = up_c ? b[m] : b[m + 4];
        out[m] = keep + rwk_dpp<0xB1>(give);             // quad_perm [1,0,3,2]
    }
}

// Totals over the 8 lanes that share l & 7 (lane distances 8, 16, 32), the same bits on each of them, for N values
// stage by stage.  Distance 8 is a rotation within the 16-lane row; 16 and 32 are the gfx950 lane-swap instructions
// (two copies of the value swap their odd rows / upper half with the other copy's even rows / lower half, so one
// copy holds the lane's own value and the other its partner's): vector-ALU moves, where a shuffle is a trip through
// the LDS crossbar.  own + partner, as a shuffle butterfly adds them.
__device__ inline float sum_swap16(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ inline float sum_swap32(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
template <int N>
__device__ inline void sum_rows8(float (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += rwk_dpp<0x128>(v[i]);    // row_ror:8: lane l ^ 8 of the 16-lane row
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = sum_swap16(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = sum_swap32(v[i]);
}

__device__ inline float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

template <int kSB>                                       // statistics blocks per lane: 8 or 16 (n_partials <= 8 kSB)
__global__ void __launch_bounds__(kT32Threads)
head_train32_kernel(const float *__restrict__ k, HeadDims dm, const float *__restrict__ bn_w,
                    const float *__restrict__ bn_b, float *running_mean, float *running_var,
                    long long *num_batches_tracked, float momentum, float eps, const float *__restrict__ partials,
                    int n_partials, const float *__restrict__ w1, const float *__restrict__ b1,
                    const float *__restrict__ w2, const float *__restrict__ b2, float drop_p,
                    const unsigned long long *__restrict__ rng, float *__restrict__ mean_out,
                    float *__restrict__ invstd_out, float *__restrict__ out, const long long *__restrict__ labels,
                    float *__restrict__ loss_partials, float *__restrict__ dy, float *__restrict__ grad_partials,
                    float *__restrict__ loss_pending) {
    constexpr int C = kT32, P = kT32, NCM = kT32MaxNC;
    constexpr int WS = C + 4;                            // row stride of the staged fc1.weight: 16-byte rows, lanes 4 rows
                                                         // apart at most two to a bank group
    __shared__ __attribute__((aligned(16))) float s_dl[kRows][NCM], s_h[kRows][P], s_dp[kRows][P], s_dyt[kRows][C],
        s_xh[kRows][C], s_w1[P][WS];
    HEAD_STAMP_DECL();
    const int nc = dm.nc;
    const int l = threadIdx.x, r = l >> 3, s = l & 7;
    if (loss_pending != nullptr && blockIdx.x == 0 && l == 0) loss_pending[0] = __builtin_nanf("");
    const long long n_tiles = (dm.g + kRows - 1) / kRows;

    // ---- everything the first tile needs, requested before the first wait ----
    long long t = blockIdx.x;
    long long row = t * kRows + r;
    bool valid = row < dm.g;
    float4 kv = ld4(k + (valid ? row : dm.g - 1) * C + 4 * s);
    long long label = labels[valid ? row : dm.g - 1];
    const bool drop = drop_p > 0.f;
    unsigned long long rng_seed = 0, rng_off = 0;
    if (drop) {
        rng_seed = rng[0];
        rng_off = rng[1];
    }
    // batch statistics: lane (r, s) adds blocks r, r + 8, ... for its 4 columns
    float sn_b[kSB];
    float4 mean_b[kSB], m2_b[kSB];
    const int stride = stats_stride(C);
#pragma unroll
    for (int u = 0; u < kSB; ++u) {
        const int b = r + 8 * u;
        const float *pb = partials + (size_t)(b < n_partials ? b : n_partials - 1) * stride;
        sn_b[u] = pb[0];
        mean_b[u] = ld4(pb + 4 + 4 * s);
        m2_b[u] = ld4(pb + 4 + C + 4 * s);
    }
    const float4 gav = ld4(bn_w + 4 * s), bev = ld4(bn_b + 4 * s), rmv = ld4(running_mean + 4 * s),
                 rvv = ld4(running_var + 4 * s);
    float4 w1_own[P * C / 4 / kT32Threads];              // this lane's share of the copy of fc1.weight to LDS
#pragma unroll
    for (int u = 0; u < P * C / 4 / kT32Threads; ++u) w1_own[u] = ld4(w1 + 4 * (l + kT32Threads * u));
    const float4 b1v = ld4(b1 + 4 * s);
    const float ga_c = bn_w[l & 31], be_c = bn_b[l & 31];      // for the parameter sums: column l & 31
    float4 w2v[NCM];
    float b2v[NCM];
#pragma unroll
    for (int q = 0; q < NCM; ++q) {
        const int qc = q < nc ? q : nc - 1;
        w2v[q] = ld4(w2 + qc * P + 4 * s);
        b2v[q] = b2[qc];
    }

    HEAD_STAMP_AT(16);
    // ---- batch statistics (merge_stats8's additions: blocks in order per lane, then the same butterfly) ----
    float sn = 0.f;
    float sm[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < kSB; ++u) {
        const bool ok = r + 8 * u < n_partials;
        const float nb = ok ? sn_b[u] : 0.f;
        sn_b[u] = nb;
        sn += nb;
        sm[0] = fmaf(nb, mean_b[u].x, sm[0]);
        sm[1] = fmaf(nb, mean_b[u].y, sm[1]);
        sm[2] = fmaf(nb, mean_b[u].z, sm[2]);
        sm[3] = fmaf(nb, mean_b[u].w, sm[3]);
        if (!ok) m2_b[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    HEAD_STAMP_AT(17);
    float mean[4], var[4];
    {
        float t5[5] = {sn, sm[0], sm[1], sm[2], sm[3]};
        sum_rows8(t5);
        sn = t5[0];
#pragma unroll
        for (int e = 0; e < 4; ++e) mean[e] = sn > 0.f ? t5[1 + e] / sn : 0.f;
    }
    {
        float qq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < kSB; ++u) {
            const float mb[4] = {mean_b[u].x, mean_b[u].y, mean_b[u].z, mean_b[u].w};
            const float m2[4] = {m2_b[u].x, m2_b[u].y, m2_b[u].z, m2_b[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dv = mb[e] - mean[e];
                qq[e] += fmaf(sn_b[u] * dv, dv, m2[e]);
            }
        }
        sum_rows8(qq);
#pragma unroll
        for (int e = 0; e < 4; ++e) var[e] = sn > 0.f ? qq[e] / sn : 0.f;
    }
    const float ga[4] = {gav.x, gav.y, gav.z, gav.w}, be[4] = {bev.x, bev.y, bev.z, bev.w};
    float invstd[4], sc[4], shift[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        invstd[e] = 1.f / sqrtf(var[e] + eps);
        sc[e] = ga[e] * invstd[e];
        shift[e] = be[e] - mean[e] * sc[e];
    }
    if (blockIdx.x == 0 && r == 0) {
        const float rm[4] = {rmv.x, rmv.y, rmv.z, rmv.w}, rv[4] = {rvv.x, rvv.y, rvv.z, rvv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float unbiased = dm.g > 1 ? var[e] * ((float)dm.g / (float)(dm.g - 1)) : var[e];
            running_mean[4 * s + e] = running_update(rm[e], mean[e], momentum);
            running_var[4 * s + e] = running_update(rv[e], unbiased, momentum);
            mean_out[4 * s + e] = mean[e];
            invstd_out[4 * s + e] = invstd[e];
        }
        if (s == 0 && num_batches_tracked != nullptr) *num_batches_tracked += 1;
    }

    HEAD_STAMP_AT(18);
    uint32_t seed_lo = 0, key_hi = 0, off_lo = 0, keep_thr = 0xFFFFFFFFu;
    float keep_scale = 1.f;
    if (drop) {
        seed_lo = (uint32_t)rng_seed;
        key_hi = (uint32_t)(rng_seed >> 32) ^ (uint32_t)(rng_off >> 32);
        off_lo = (uint32_t)rng_off;
        const double q = 1.0 - (double)drop_p;
        const double thr = q * 4294967296.0;
        keep_thr = thr >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)thr;
        keep_scale = 1.f / (1.f - drop_p);
    }
    // fc1's weight to LDS: the lane's two slices (all rows x columns 4s..4s+3 for the forward product, rows 4s..4s+3 x
    // all columns for dY) are read from there, 16 bytes a time, when they are needed (in registers they are 256 of
    // them and push the kernel into the accumulation registers)
#pragma unroll
    for (int u = 0; u < P * C / 4 / kT32Threads; ++u) {
        const int i4 = l + kT32Threads * u;              // 16-byte piece i4: row i4 / 8, columns 4 (i4 % 8) ..
        *reinterpret_cast<float4 *>(&s_w1[i4 >> 3][4 * (i4 & 7)]) = w1_own[u];
    }
    __syncthreads();
    const float inv_g = 1.f / (float)dm.g;
    const int o_b2 = nc * P, o_w1 = o_b2 + nc, o_b1 = o_w1 + P * C, o_ga = o_b1 + P, o_be = o_ga + C;
    const int nv = o_be + C;

    // this lane's parameter sums over the tiles it sees: fc1.weight[16 (l >> 5) + u][l & 31], u < 16; fc2.weight
    // element l + 64u, u < 2; fc1.bias (lanes < 32) or fc2.bias (lanes 32..32+nc-1); gamma (lanes < 32) or beta
    float g_w1[16], g_w2[2] = {0.f, 0.f}, g_b = 0.f, g_gb = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) g_w1[u] = 0.f;
    float nll_sum = 0.f;                                 // lane 0: this block's rows, in order

    for (; t < n_tiles; t += gridDim.x) {
        if (t != (long long)blockIdx.x) {                // later tiles of a batch of more than 8 x 256 rows
            row = t * kRows + r;
            valid = row < dm.g;
            kv = ld4(k + (valid ? row : dm.g - 1) * C + 4 * s);
            label = labels[valid ? row : dm.g - 1];
        }
        HEAD_STAMP_AT(19);
        // ---- forward ----
        const float kk[4] = {kv.x, kv.y, kv.z, kv.w};
        float y[4], xh[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            y[e] = fmaf(kk[e], sc[e], shift[e]);
            xh[e] = (kk[e] - mean[e]) * invstd[e];
        }
        float part[P];
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const float4 wf = *reinterpret_cast<const float4 *>(&s_w1[j][4 * s]);      // fc1.weight[j][4s..4s+3]
            part[j] = fmaf(y[3], wf.w, fmaf(y[2], wf.z, fmaf(y[1], wf.y, y[0] * wf.x)));
        }
        HEAD_STAMP_AT(20);
        float h[4];
        reduce_scatter8(part, s, h);                     // hidden units 4s..4s+3 of row r
        {
            const float bb[4] = {b1v.x, b1v.y, b1v.z, b1v.w};
            uint32_t wv[4] = {0, 0, 0, 0};
            if (drop) {
                const RwkPhilox4 rr = rwk_philox4x32_10((uint32_t)row, (uint32_t)s, kHeadCall, off_lo, seed_lo, key_hi);
                wv[0] = rr.x; wv[1] = rr.y; wv[2] = rr.z; wv[3] = rr.w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = h[e] + bb[e];
                a = a > 0.f ? a : 0.f;
                if (drop) a = wv[e] < keep_thr ? a * keep_scale : 0.f;
                h[e] = a;
            }
        }
        HEAD_STAMP_AT(21);
        float o[NCM];
#pragma unroll
        for (int q = 0; q < NCM; ++q) {
            float a = fmaf(h[3], w2v[q].w, fmaf(h[2], w2v[q].z, fmaf(h[1], w2v[q].y, h[0] * w2v[q].x)));
            o[q] = rwk_group_sum(a, 8) + b2v[q];
        }
        // log-softmax, the cross-entropy of the stored log-probabilities and its gradient for d_loss = 1
        // (head_train_kernel's expressions; every lane of the row computes them)
        HEAD_STAMP_AT(22);
        float dl[NCM];
        {
            float mx = o[0];
#pragma unroll
            for (int q = 1; q < NCM; ++q) if (q < nc) mx = fmaxf(mx, o[q]);
            float se = 0.f;
#pragma unroll
            for (int q = 0; q < NCM; ++q) if (q < nc) se += expf(o[q] - mx);
            const float lse = mx + logf(se);
#pragma unroll
            for (int q = 0; q < NCM; ++q) o[q] -= lse;
            if (valid && s < nc) {
                float mine = o[0];
#pragma unroll
                for (int q = 1; q < NCM; ++q) mine = s == q ? o[q] : mine;
                out[row * nc + s] = mine;
            }
            float mx2 = o[0];
#pragma unroll
            for (int q = 1; q < NCM; ++q) if (q < nc) mx2 = fmaxf(mx2, o[q]);
            float se2 = 0.f;
#pragma unroll
            for (int q = 0; q < NCM; ++q) if (q < nc) se2 += expf(o[q] - mx2);
            const float lse2 = mx2 + logf(se2);
            long long cls = label;
            cls = cls < 0 ? 0 : (cls >= nc ? nc - 1 : cls);  // the host side has checked the range (see head.py)
            float o_cls = o[0];
#pragma unroll
            for (int q = 1; q < NCM; ++q) o_cls = cls == q ? o[q] : o_cls;
            const float nll = valid ? lse2 - o_cls : 0.f;
            // rows of the tile in order, on lane 0 (the value of lane 8 r' is row r')
#pragma unroll
            for (int rr = 0; rr < kRows; ++rr) nll_sum += __shfl(nll, 8 * rr, 64);
            float go[NCM], sgo = 0.f;
#pragma unroll
            for (int q = 0; q < NCM; ++q) {
                go[q] = q < nc ? inv_g * (expf(o[q] - lse2) - (q == cls ? 1.f : 0.f)) : 0.f;
                sgo += go[q];
            }
#pragma unroll
            for (int q = 0; q < NCM; ++q) dl[q] = (valid && q < nc) ? go[q] - expf(o[q]) * sgo : 0.f;
        }
        HEAD_STAMP_AT(23);
        // ---- backward, per-tile part ----
        float dp[4];
        {
            const float w2e[NCM][4] = {{w2v[0].x, w2v[0].y, w2v[0].z, w2v[0].w}, {w2v[1].x, w2v[1].y, w2v[1].z, w2v[1].w},
                                       {w2v[2].x, w2v[2].y, w2v[2].z, w2v[2].w}, {w2v[3].x, w2v[3].y, w2v[3].z, w2v[3].w}};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = 0.f;
#pragma unroll
                for (int q = 0; q < NCM; ++q) a = fmaf(dl[q], w2e[q][e], a);
                dp[e] = h[e] > 0.f ? a * keep_scale : 0.f;
            }
        }
        float dyv[4];
        {
            float pd[C];
#pragma unroll
            for (int cg = 0; cg < C / 4; ++cg) {
                float4 wb[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) wb[jj] = *reinterpret_cast<const float4 *>(&s_w1[4 * s + jj][4 * cg]);
                pd[4 * cg + 0] = fmaf(dp[3], wb[3].x, fmaf(dp[2], wb[2].x, fmaf(dp[1], wb[1].x, dp[0] * wb[0].x)));
                pd[4 * cg + 1] = fmaf(dp[3], wb[3].y, fmaf(dp[2], wb[2].y, fmaf(dp[1], wb[1].y, dp[0] * wb[0].y)));
                pd[4 * cg + 2] = fmaf(dp[3], wb[3].z, fmaf(dp[2], wb[2].z, fmaf(dp[1], wb[1].z, dp[0] * wb[0].z)));
                pd[4 * cg + 3] = fmaf(dp[3], wb[3].w, fmaf(dp[2], wb[2].w, fmaf(dp[1], wb[1].w, dp[0] * wb[0].w)));
            }
            reduce_scatter8(pd, s, dyv);                 // dY[r][4s..4s+3]
        }
        HEAD_STAMP_AT(24);
        if (valid) *reinterpret_cast<float4 *>(dy + row * C + 4 * s) = make_float4(dyv[0], dyv[1], dyv[2], dyv[3]);
        // ---- this tile's share of the parameter sums: the rows meet in LDS (rows past the batch are zero there) ----
        __syncthreads();                                 // the previous tile's readers are done (one wavefront: a wait)
        {
            const float z = valid ? 1.f : 0.f;
            *reinterpret_cast<float4 *>(&s_h[r][4 * s]) = make_float4(h[0] * z, h[1] * z, h[2] * z, h[3] * z);
            *reinterpret_cast<float4 *>(&s_dp[r][4 * s]) = make_float4(dp[0] * z, dp[1] * z, dp[2] * z, dp[3] * z);
            *reinterpret_cast<float4 *>(&s_dyt[r][4 * s]) = make_float4(dyv[0] * z, dyv[1] * z, dyv[2] * z, dyv[3] * z);
            *reinterpret_cast<float4 *>(&s_xh[r][4 * s]) = make_float4(xh[0] * z, xh[1] * z, xh[2] * z, xh[3] * z);
            if (s < NCM) {
                float mine = dl[0];
#pragma unroll
                for (int q = 1; q < NCM; ++q) mine = s == q ? dl[q] : mine;
                s_dl[r][s] = mine;
            }
        }
        __syncthreads();
        HEAD_STAMP_AT(25);
        {
            const int col = l & 31;
            float yv[kRows], dyc[kRows], xhc[kRows];
#pragma unroll
            for (int rr = 0; rr < kRows; ++rr) {
                xhc[rr] = s_xh[rr][col];
                dyc[rr] = s_dyt[rr][col];
            }
#pragma unroll
            for (int rr = 0; rr < kRows; ++rr) yv[rr] = fmaf(xhc[rr], ga_c, be_c);
            {
                float a16[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) a16[u] = 0.f;
#pragma unroll
                for (int rr = 0; rr < kRows; ++rr) {
#pragma unroll
                    for (int u4 = 0; u4 < 4; ++u4) {
                        const float4 d4 = *reinterpret_cast<const float4 *>(&s_dp[rr][16 * (l >> 5) + 4 * u4]);
                        a16[4 * u4 + 0] = fmaf(d4.x, yv[rr], a16[4 * u4 + 0]);
                        a16[4 * u4 + 1] = fmaf(d4.y, yv[rr], a16[4 * u4 + 1]);
                        a16[4 * u4 + 2] = fmaf(d4.z, yv[rr], a16[4 * u4 + 2]);
                        a16[4 * u4 + 3] = fmaf(d4.w, yv[rr], a16[4 * u4 + 3]);
                    }
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) g_w1[u] += a16[u];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {                // fc2.weight[q][j], element l + 64u
                const int e = l + 64 * u;
                const int q = (e >> 5) < nc ? (e >> 5) : nc - 1, j = e & 31;
                float a = 0.f;
#pragma unroll
                for (int rr = 0; rr < kRows; ++rr) a = fmaf(s_dl[rr][q], s_h[rr][j], a);
                g_w2[u] += a;
            }
            {                                            // fc1.bias[l] on lanes < 32, fc2.bias[l - 32] on the others
                const int q = (l - 32) < nc ? (l & 31) : nc - 1;
                float a = 0.f;
#pragma unroll
                for (int rr = 0; rr < kRows; ++rr) a += l < 32 ? s_dp[rr][col] : s_dl[rr][q < NCM ? q : 0];
                g_b += a;
            }
            {                                            // gamma[col] on lanes < 32, beta[col] on the others
                float a = 0.f;
#pragma unroll
                for (int rr = 0; rr < kRows; ++rr) a = l < 32 ? fmaf(dyc[rr], xhc[rr], a) : a + dyc[rr];
                g_gb += a;
            }
        }
    }
    HEAD_STAMP_AT(26);
    // ---- this block's partial row, in head_train_kernel's layout ----
    float *dst = grad_partials + (size_t)blockIdx.x * nv;
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (l + 64 * u < o_b2) dst[l + 64 * u] = g_w2[u];
    if (l >= 32 && l - 32 < nc) dst[o_b2 + (l - 32)] = g_b;
#pragma unroll
    for (int u = 0; u < 16; ++u) dst[o_w1 + (16 * (l >> 5) + u) * C + (l & 31)] = g_w1[u];
    if (l < 32) {
        dst[o_b1 + l] = g_b;
        dst[o_ga + l] = g_gb;
    } else {
        dst[o_be + (l - 32)] = g_gb;
    }
    if (l == 0) loss_partials[blockIdx.x] = nll_sum;
    HEAD_STAMP_AT(27);
}

// Workgroups of the finishing launch: at least the tile count (each finishes the dK rows of its tiles), and enough
// of them that a workgroup's slice of the parameter gradients is at most the 32 elements whose partial rows it fetches
// in one round - with few graphs (8 tiles for a batch of 64) a workgroup otherwise walks hundreds of elements, a round
// trip per 32 (13 us on such a batch).  An element's sum does not depend on which workgroup forms it.
inline int finish_grid(int tiles, int c, int p, int nc) {
    const int by_elements = (head_nv(c, p, nc) + kBlock / 8 - 1) / (kBlock / 8);
    const int want = by_elements < 256 ? by_elements : 256;
    return tiles > want ? tiles : want;
}

// ---------------------------------------------------------------------------------------------------
// backward, second launch: parameter gradients (block partials added in block order) and dK
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
head_bwd_finish_kernel(const float *__restrict__ k, HeadDims dm, const float *__restrict__ bn_w,
                       const float *__restrict__ mean, const float *__restrict__ invstd,
                       const float *__restrict__ dy, const float *__restrict__ partials, int n_partials, int training,
                       float *grads, float *__restrict__ d_k, const float *__restrict__ d_loss,
                       const float *__restrict__ loss_partials, float *__restrict__ loss_out, int cols_ready) {
    // cols_ready: an earlier launch (this kernel with d_k = nullptr) has written grads; this one only forms dK and
    // takes the gamma / beta column sums from grads instead of adding the block partials up again.
    __shared__ float s_dga[kMaxC], s_dbe[kMaxC];
    const int c = dm.c, p = dm.p, nc = dm.nc;
    const int nv = head_nv(c, p, nc);
    const int tid = threadIdx.x;
    const int o_ga = nc * p + nc + p * c + p;
    // d_loss not null: dy and the partials were computed for d(loss) = 1 (head_train_kernel) and take the factor here
    const float dl = d_loss != nullptr ? d_loss[0] : 1.f;
    // mean loss from the forward launch's per-block sums, by the first wavefront of the last block: lane l adds blocks
    // l, l + 64, ... in order (at most kMaxBlocks = 256), then a butterfly whose operand pairs are fixed.  Requested
    // here with everything else, finished at the end of the kernel.
    const bool loss_wave = loss_out != nullptr && blockIdx.x == gridDim.x - 1 && tid < 64;
    float loss_v[4] = {0.f, 0.f, 0.f, 0.f};
    if (loss_wave) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int b = tid + 64 * u;
            loss_v[u] = loss_partials[b < n_partials ? b : n_partials - 1];
        }
    }
    auto finish_loss = [&]() {
        if (!loss_wave) return;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (tid + 64 * u >= n_partials) loss_v[u] = 0.f;
        float acc = ((loss_v[0] + loss_v[1]) + loss_v[2]) + loss_v[3];
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (tid == 0) loss_out[0] = acc / (float)dm.g;
    };
    // Everything below depends on loads that do not depend on each other: the block partials of this block's slice of
    // the parameter gradients, those of the gamma / beta columns, and the first tile's rows.  They are all requested
    // before the first of them is waited for (one round trip instead of eight); what does not fit the prefetch
    // (more than 128 partials, more than 32 elements or 64 columns per block, later tiles) is loaded where it is used.
    const int lane8 = tid & 7, grp = tid >> 3;
    const int per = (nv + gridDim.x - 1) / gridDim.x;
    const int e_lo = blockIdx.x * per, e_hi = e_lo + per < nv ? e_lo + per : nv;
    const int n_cols2 = d_k != nullptr ? 2 * c : 0;
    const long long n_tiles = (dm.g + kRows - 1) / kRows;
    RwkFinPartials pre_e, pre_c[2];
    float ready_col[kMaxC * 2 / kBlock];
    if (cols_ready) {
#pragma unroll
        for (int u = 0; u < kMaxC * 2 / kBlock; ++u) {
            const int q = tid + u * kBlock;
            ready_col[u] = grads[o_ga + (q < 2 * c ? q : 2 * c - 1)];
        }
    }
    if (!cols_ready && e_lo < e_hi)
        pre_e.load(partials, n_partials, nv, e_lo + grp < e_hi ? e_lo + grp : e_hi - 1, lane8);
#pragma unroll
    for (int it = 0; it < 2; ++it)
        if (!cols_ready && it * (kBlock / 8) < n_cols2) {
            const int q = it * (kBlock / 8) + grp;
            pre_c[it].load(partials, n_partials, nv, o_ga + (q < 2 * c ? q : 2 * c - 1), lane8);
        }
    float t_dy = 0.f, t_k = 0.f, t_sc = 0.f, t_mu = 0.f, t_is = 0.f;
    const FastDiv by_c(c);
    if (d_k != nullptr && (long long)blockIdx.x < n_tiles) {
        const long long r0 = (long long)blockIdx.x * kRows;
        const int nr = (int)(dm.g - r0 < kRows ? dm.g - r0 : kRows);
        const int i = tid < nr * c ? tid : 0;
        const int col = by_c.mod(i);
        t_dy = dy[r0 * c + i];
        t_k = k[r0 * c + i];
        t_is = invstd[col];
        t_sc = bn_w[col] * t_is;
        t_mu = mean[col];
    }
    for (int e0 = e_lo; e0 < (cols_ready ? e_lo : e_hi); e0 += kBlock / 8) {   // uniform trip count: the butterfly needs all lanes
        const int e = e0 + grp;
        const int ec = e < e_hi ? e : e_hi - 1;
        float acc = e0 == e_lo ? pre_e.sum(partials, n_partials, nv, ec, lane8)
                               : rwk_sum_partials8(partials, n_partials, nv, ec, lane8);
        acc = rwk_group_sum(acc, 8);
        if (lane8 == 0 && e < e_hi) grads[e] = acc * dl;
    }
    if (d_k == nullptr) {
        finish_loss();
        return;
    }
    if (cols_ready) {
#pragma unroll
        for (int u = 0; u < kMaxC * 2 / kBlock; ++u) {
            const int q = tid + u * kBlock;
            if (q < c) s_dga[q] = ready_col[u];
            else if (q < 2 * c) s_dbe[q - c] = ready_col[u];
        }
    }
    for (int c0 = 0; c0 < (cols_ready ? 0 : 2 * c); c0 += kBlock / 8) {
        const int q = c0 + grp;                              // q < c: dgamma column, else dbeta column
        const int qc = q < 2 * c ? q : 2 * c - 1;
        float acc;
        if (c0 == 0) acc = pre_c[0].sum(partials, n_partials, nv, o_ga + qc, lane8);
        else if (c0 == kBlock / 8) acc = pre_c[1].sum(partials, n_partials, nv, o_ga + qc, lane8);
        else acc = rwk_sum_partials8(partials, n_partials, nv, o_ga + qc, lane8);
        acc = rwk_group_sum(acc, 8);
        if (lane8 == 0 && q < 2 * c) {
            if (q < c) s_dga[q] = acc * dl; else s_dbe[q - c] = acc * dl;
        }
    }
    __syncthreads();
    const float inv_g = 1.f / (float)dm.g;
    for (long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const long long r0 = t * kRows;
        const int nr = (int)(dm.g - r0 < kRows ? dm.g - r0 : kRows);
        for (int i = tid; i < nr * c; i += kBlock) {
            const int col = by_c.mod(i);
            const long long at = r0 * c + i;
            const bool pre = t == (long long)blockIdx.x && i == tid;
            const float is = pre ? t_is : invstd[col];
            const float sc = pre ? t_sc : bn_w[col] * is;
            d_k[at] = rwk_bn_input_grad(pre ? t_dy : dy[at], pre ? t_k : k[at], pre ? t_mu : mean[col], is, sc,
                                        s_dga[col], s_dbe[col], inv_g, dl, training != 0);
        }
    }
    finish_loss();
}

// One launch while the gamma / beta columns fit the kernel's prefetch (2 c <= 64) or nearly so.  Wider heads on
// batches with many block partials take two: every
// workgroup of the single launch adds up all 2 c columns of all block partials before it can form a dK row
// (c = 128, 256 partials: 256 KB from L2 and sixteen dependent trips per workgroup, 22 us on 4096 graphs); split, the
// first launch forms each parameter gradient once (32 elements per workgroup, one trip) and the second reads the
// 2 c column sums it needs from there.  Same sums in the same order either way.
inline void launch_finish(hipStream_t stream, const float *k, const HeadDims &dm, const float *bn_w, const float *mean,
                          const float *invstd, const float *dy, const float *partials, int blocks, int training,
                          float *grads, float *d_k, const float *d_loss, const float *loss_partials, float *loss) {
    // dependent trips the single launch makes beyond its prefetch: a round per 32 further columns, each a trip per 64
    // block partials; a second launch costs about three of them
    const int extra_trips = ((2 * dm.c - 1) / (kBlock / 8) - 1) * ((blocks + 63) / 64);
    if (d_k == nullptr || extra_trips <= 6) {
        hipLaunchKernelGGL(head_bwd_finish_kernel, dim3(finish_grid(blocks, dm.c, dm.p, dm.nc)), dim3(kBlock), 0, stream,
                           k, dm, bn_w, mean, invstd, dy, partials, blocks, training, grads, d_k, d_loss, loss_partials,
                           loss, 0);
        return;
    }
    const int by_elements = (head_nv(dm.c, dm.p, dm.nc) + kBlock / 8 - 1) / (kBlock / 8);
    hipLaunchKernelGGL(head_bwd_finish_kernel, dim3(by_elements), dim3(kBlock), 0, stream, k, dm, bn_w, mean, invstd, dy,
                       partials, blocks, training, grads, (float *)nullptr, d_loss, loss_partials, loss, 0);
    hipLaunchKernelGGL(head_bwd_finish_kernel, dim3(blocks), dim3(kBlock), 0, stream, k, dm, bn_w, mean, invstd, dy,
                       partials, blocks, training, grads, d_k, d_loss, (const float *)nullptr, (float *)nullptr, 1);
}

// ---------------------------------------------------------------------------------------------------
// Mean cross-entropy of (G, NC) scores against int64 class labels, as F.cross_entropy with default arguments
// (rwgnn2/main.py:85 applies it to the model's log-probabilities): loss = mean_r (logsumexp(x_r) - x_r[y_r]).
// One workgroup, rows strided over its threads, partial sums added in a fixed tree order.
// ---------------------------------------------------------------------------------------------------
constexpr int kCeBlock = 1024;

__global__ void __launch_bounds__(kCeBlock)
cross_entropy_forward_kernel(const float *__restrict__ x, const long long *__restrict__ y, long long g, int nc,
                             float *__restrict__ loss) {
    __shared__ float red[kCeBlock];
    float acc = 0.f;
    for (long long r = threadIdx.x; r < g; r += kCeBlock) {
        const float *row = x + r * nc;
        long long cls = y[r];
        cls = cls < 0 ? 0 : (cls >= nc ? nc - 1 : cls);           // the host side checks the range once per label tensor
        acc += row_lse(row, nc) - row[cls];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = kCeBlock / 2; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = red[0] / (float)g;
}

__global__ void __launch_bounds__(kBlock)
cross_entropy_backward_kernel(const float *__restrict__ x, const long long *__restrict__ y, long long g, int nc,
                              const float *__restrict__ d_loss, float *__restrict__ d_x) {
    const long long r = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (r >= g) return;
    const float *row = x + r * nc;
    const float lse = row_lse(row, nc);
    const float scale = d_loss[0] / (float)g;
    long long cls = y[r];
    cls = cls < 0 ? 0 : (cls >= nc ? nc - 1 : cls);               // as the forward kernel does
    for (int q = 0; q < nc; ++q) d_x[r * nc + q] = scale * (expf(row[q] - lse) - (q == cls ? 1.f : 0.f));
}

size_t fwd_lds(int c, int p, int nc) {
    return sizeof(float) * ((size_t)p * (c + 1) + (size_t)nc * (p + 1) + p + nc + 2 * c + (size_t)kRows * (c + 1) +
                            (size_t)kRows * (p + 1) + (size_t)kRows * nc);
}
size_t bwd_lds(int c, int p, int nc) {
    return sizeof(float) * ((size_t)p * (c + 1) + (size_t)nc * (p + 1) + 2 * (size_t)kRows * (c + 1) +
                            2 * (size_t)kRows * (p + 1) + (size_t)kRows * nc + head_nv(c, p, nc) + 2 * (size_t)c);
}

size_t train_lds(int c, int p, int nc) {
    return sizeof(float) * ((size_t)p * (c + 1) + (size_t)nc * (p + 1) + p + nc + 6 * (size_t)c +
                            2 * (size_t)kRows * (c + 1) + 2 * (size_t)kRows * (p + 1) + 2 * (size_t)kRows * nc +
                            head_nv(c, p, nc) + 1);
}

bool dims_ok(int64_t g, int64_t c, int64_t p, int64_t nc) {
    return g >= 1 && c >= 1 && c <= kMaxC && p >= 1 && nc >= 1 && nc <= kMaxNC && p * c <= kMaxPC &&
           g < (int64_t)1 << 40;
}

template <typename K>
int allow_lds(K kern, size_t lds) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    return RWK_OK;
}

}  // namespace

extern "C" {

#ifdef RWK_STAMP
int rwk_debug_read_head_stamps(unsigned long long *host32) {
    return (int)hipMemcpyFromSymbol(host32, HIP_SYMBOL(rwk_head_stamp_buf), sizeof(unsigned long long) * 32);
}
#endif

int rwk_head_supported(int64_t g, int64_t c, int64_t p, int64_t nc) {
    return dims_ok(g, c, p, nc) && bwd_lds((int)c, (int)p, (int)nc) <= 150 * 1024 ? 1 : 0;
}

int64_t rwk_head_n_blocks(int64_t g) { return g >= 1 ? head_blocks(g) : 0; }

// room for the block partials of whichever launch writes them: the head's own statistics launch or the K reduction
int64_t rwk_head_stats_size(int64_t g, int64_t c) {
    if (g < 1) return 0;
    const int64_t own = head_blocks(g), reduction = rwk_fused_reduce_stats_blocks(g, c);
    return (own > reduction ? own : reduction) * stats_stride((int)c);
}

int64_t rwk_head_grad_size(int64_t c, int64_t p, int64_t nc) { return head_nv((int)c, (int)p, (int)nc); }

int rwk_head_forward_loss(const float *k, int64_t g, int64_t c, int64_t p, int64_t nc, const float *bn_w,
                          const float *bn_b, float *running_mean, float *running_var, int64_t *num_batches_tracked,
                          float momentum, float eps, int training, float *stats, const float *w1, const float *b1,
                          const float *w2, const float *b2, float drop_p, uint64_t *rng_state, uint64_t *rng_used,
                          float *mean_out, float *invstd_out, float *h1d_out, float *out, const int64_t *labels,
                          float *loss_partials, rwk_stream_t stream_);

int rwk_head_forward(const float *k, int64_t g, int64_t c, int64_t p, int64_t nc, const float *bn_w,
                     const float *bn_b, float *running_mean, float *running_var, int64_t *num_batches_tracked,
                     float momentum, float eps, int training, float *stats, const float *w1, const float *b1,
                     const float *w2, const float *b2, float drop_p, uint64_t *rng_state, uint64_t *rng_used,
                     float *mean_out, float *invstd_out, float *h1d_out, float *out, rwk_stream_t stream_) {
    return rwk_head_forward_loss(k, g, c, p, nc, bn_w, bn_b, running_mean, running_var, num_batches_tracked, momentum,
                                 eps, training, stats, w1, b1, w2, b2, drop_p, rng_state, rng_used, mean_out,
                                 invstd_out, h1d_out, out, nullptr, nullptr, stream_);
}

// n_stats_ready > 0: `stats` already holds that many block partials (rwk_fused_reduce_stats wrote them and drew
// the generator state into rng_used), so the statistics launch is skipped
static int head_forward_impl(const float *k, int64_t g, int64_t c, int64_t p, int64_t nc, const float *bn_w,
                             const float *bn_b, float *running_mean, float *running_var, int64_t *num_batches_tracked,
                             float momentum, float eps, int training, float *stats, int n_stats_ready, const float *w1,
                             const float *b1, const float *w2, const float *b2, float drop_p, uint64_t *rng_state,
                             uint64_t *rng_used, float *mean_out, float *invstd_out, float *h1d_out, float *out,
                             const int64_t *labels, float *loss_partials, rwk_stream_t stream_);

int rwk_head_forward_loss(const float *k, int64_t g, int64_t c, int64_t p, int64_t nc, const float *bn_w,
                          const float *bn_b, float *running_mean, float *running_var, int64_t *num_batches_tracked,
                          float momentum, float eps, int training, float *stats, const float *w1, const float *b1,
                          const float *w2, const float *b2, float drop_p, uint64_t *rng_state, uint64_t *rng_used,
                          float *mean_out, float *invstd_out, float *h1d_out, float *out, const int64_t *labels,
                          float *loss_partials, rwk_stream_t stream_) {
    return head_forward_impl(k, g, c, p, nc, bn_w, bn_b, running_mean, running_var, num_batches_tracked, momentum, eps,
                             training, stats, 0, w1, b1, w2, b2, drop_p, rng_state, rng_used, mean_out, invstd_out,
                             h1d_out, out, labels, loss_partials, stream_);
}

int rwk_head_apply_loss(const float *k, int64_t g, int64_t c, int64_t p, int64_t nc, const float *bn_w,
                        const float *bn_b, float *running_mean, float *running_var, int64_t *num_batches_tracked,
                        float momentum, float eps, const float *stats, int64_t n_stats, const float *w1,
                        const float *b1, const float *w2, const float *b2, float drop_p, const uint64_t *rng_used,
                        float *mean_out, float *invstd_out, float *h1d_out, float *out, const int64_t *labels,
                        float *loss_partials, rwk_stream_t stream_) {
    RWK_REQUIRE_PTR(stats);
    if (n_stats < 1 || n_stats > kMaxFusedStatsBlocks) return RWK_E_SIZE;
    if (drop_p > 0.f) RWK_REQUIRE_PTR(rng_used);
    return head_forward_impl(k, g, c, p, nc, bn_w, bn_b, running_mean, running_var, num_batches_tracked, momentum, eps,
                             1, const_cast<float *>(stats), (int)n_stats, w1, b1, w2, b2, drop_p,
                             const_cast<uint64_t *>(rng_used), const_cast<uint64_t *>(rng_used), mean_out, invstd_out,
                             h1d_out, out, labels, loss_partials, stream_);
}

// At most kMaxFusedStatsBlocks blocks of at least eight rows.  (Sixteen rows per block at 32 columns - two per thread,
// half the partials for the head to merge - made the reduction itself 3 us slower: 6.4 -> 9.8 us at 1024 rows.)
int64_t rwk_fused_reduce_stats_rows(int64_t g, int64_t c) {
    if (g < 1 || c < 1) return 0;
    // one row per block while the blocks stay within what the head merges (kreduce_stats_row_kernel: every slot of a
    // graph in flight at once, whatever the graph's size), else at least eight
    if (g <= kMaxFusedStatsBlocks) return 1;
    // beyond that as few rows per block as keep the blocks within that number: the reduction is latency-bound, more
    // workgroups with fewer rows each end sooner (256 power-law graphs: 44 us at eight rows per block)
    return (g + kMaxFusedStatsBlocks - 1) / kMaxFusedStatsBlocks;
}

int64_t rwk_fused_reduce_stats_blocks(int64_t g, int64_t c) {
    if (g < 1 || c < 1) return 0;
    const int64_t rows = rwk_fused_reduce_stats_rows(g, c);
    return (g + rows - 1) / rows;
}

int rwk_fused_reduce_stats(const float *k_partials, const int32_t *seg_ptr, const float *counts_f32, int64_t n_graphs,
                           int64_t h, int64_t max_step, int64_t stream_len, float drop_p, int normalize, float *k,
                           float *stats, uint64_t *rng_state, uint64_t *rng_used, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(k_partials);
    RWK_REQUIRE_PTR(seg_ptr);
    RWK_REQUIRE_PTR(k);
    RWK_REQUIRE_PTR(stats);
    if (normalize) RWK_REQUIRE_PTR(counts_f32);
    if ((rng_state == nullptr) != (rng_used == nullptr)) return RWK_E_NULL;
    if (n_graphs <= 0 || h <= 0 || max_step <= 0 || stream_len <= 0 || !(drop_p >= 0.f && drop_p < 1.f))
        return RWK_E_SIZE;
    const int64_t c = h * max_step;
    if (c > kMaxC || n_graphs >= (int64_t)1 << 40) return RWK_E_UNSUPPORTED;
    const float scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const int64_t rows = rwk_fused_reduce_stats_rows(n_graphs, c);
    if (rows <= kKrSeqRows)
        hipLaunchKernelGGL(kreduce_stats_row_kernel, dim3((unsigned)rwk_fused_reduce_stats_blocks(n_graphs, c)),
                           dim3(kBlock), 0, (hipStream_t)stream, k_partials, seg_ptr, counts_f32, (long long)n_graphs,
                           (int)c, (int)stream_len, scale, normalize, (int)rows, k, stats,
                           (unsigned long long *)rng_state, (unsigned long long *)rng_used);
    else
        hipLaunchKernelGGL(kreduce_stats_kernel, dim3((unsigned)rwk_fused_reduce_stats_blocks(n_graphs, c)),
                           dim3(kBlock), 0, (hipStream_t)stream, k_partials, seg_ptr, counts_f32, (long long)n_graphs,
                           (int)c, (int)stream_len, scale, normalize, (int)rows, k, stats,
                           (unsigned long long *)rng_state, (unsigned long long *)rng_used);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

static int head_forward_impl(const float *k, int64_t g, int64_t c, int64_t p, int64_t nc, const float *bn_w,
                             const float *bn_b, float *running_mean, float *running_var, int64_t *num_batches_tracked,
                             float momentum, float eps, int training, float *stats, int n_stats_ready, const float *w1,
                             const float *b1, const float *w2, const float *b2, float drop_p, uint64_t *rng_state,
                             uint64_t *rng_used, float *mean_out, float *invstd_out, float *h1d_out, float *out,
                             const int64_t *labels, float *loss_partials, rwk_stream_t stream_) {
    if ((labels == nullptr) != (loss_partials == nullptr)) return RWK_E_NULL;
    RWK_REQUIRE_PTR(k);
    RWK_REQUIRE_PTR(bn_w);
    RWK_REQUIRE_PTR(bn_b);
    RWK_REQUIRE_PTR(running_mean);
    RWK_REQUIRE_PTR(running_var);
    RWK_REQUIRE_PTR(w1);
    RWK_REQUIRE_PTR(b1);
    RWK_REQUIRE_PTR(w2);
    RWK_REQUIRE_PTR(b2);
    RWK_REQUIRE_PTR(out);
    if (!rwk_head_supported(g, c, p, nc)) return RWK_E_UNSUPPORTED;
    if (drop_p < 0.f || drop_p >= 1.f) return RWK_E_SIZE;
    const bool drop = training && drop_p > 0.f;
    if (training) {
        RWK_REQUIRE_PTR(stats);
        RWK_REQUIRE_PTR(mean_out);
        RWK_REQUIRE_PTR(invstd_out);
    }
    if (drop) {
        RWK_REQUIRE_PTR(rng_state);
        RWK_REQUIRE_PTR(rng_used);
    }
    hipStream_t stream = (hipStream_t)stream_;
    const int blocks = head_blocks(g);
    const HeadDims dm{(long long)g, (int)c, (int)p, (int)nc};
    if (training && n_stats_ready == 0) {
        hipLaunchKernelGGL(head_stats_kernel, dim3(head_stats_blocks(g)), dim3(kBlock), 0, stream, k, (long long)g, (int)c, stats,
                           drop ? (unsigned long long *)rng_state : nullptr,
                           drop ? (unsigned long long *)rng_used : nullptr);
        RWK_LAUNCH_CHECK();
    }
    const int n_stats = n_stats_ready > 0 ? n_stats_ready : head_stats_blocks(g);
    const size_t lds = fwd_lds((int)c, (int)p, (int)nc);
    const int rc = allow_lds(head_apply_kernel, lds);
    if (rc != RWK_OK) return rc;
    hipLaunchKernelGGL(head_apply_kernel, dim3(blocks), dim3(kBlock), lds, stream, k, dm, bn_w, bn_b, running_mean,
                       running_var, (long long *)num_batches_tracked, momentum, eps, training, stats, n_stats, w1, b1,
                       w2, b2, drop_p, (const unsigned long long *)rng_used, mean_out, invstd_out, h1d_out, out,
                       (const long long *)labels, loss_partials);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

int rwk_head_backward_loss(const float *k, int64_t g, int64_t c, int64_t p, int64_t nc, const float *bn_w,
                           const float *bn_b, const float *mean, const float *invstd, int training, const float *w1,
                           const float *w2, const float *h1d, const float *out, const float *d_out, float drop_p,
                           float *dy, float *partials, float *grads, float *d_k, const int64_t *labels,
                           const float *d_loss, const float *loss_partials, float *loss, rwk_stream_t stream_);

int rwk_head_backward(const float *k, int64_t g, int64_t c, int64_t p, int64_t nc, const float *bn_w,
                      const float *bn_b, const float *mean, const float *invstd, int training, const float *w1,
                      const float *w2, const float *h1d, const float *out, const float *d_out, float drop_p,
                      float *dy, float *partials, float *grads, float *d_k, rwk_stream_t stream_) {
    RWK_REQUIRE_PTR(d_out);
    return rwk_head_backward_loss(k, g, c, p, nc, bn_w, bn_b, mean, invstd, training, w1, w2, h1d, out, d_out, drop_p,
                                  dy, partials, grads, d_k, nullptr, nullptr, nullptr, nullptr, stream_);
}

int rwk_head_backward_loss(const float *k, int64_t g, int64_t c, int64_t p, int64_t nc, const float *bn_w,
                           const float *bn_b, const float *mean, const float *invstd, int training, const float *w1,
                           const float *w2, const float *h1d, const float *out, const float *d_out, float drop_p,
                           float *dy, float *partials, float *grads, float *d_k, const int64_t *labels,
                           const float *d_loss, const float *loss_partials, float *loss, rwk_stream_t stream_) {
    if (labels != nullptr) {
        RWK_REQUIRE_PTR(d_loss);
        if ((loss == nullptr) != (loss_partials == nullptr)) return RWK_E_NULL;
    } else {
        RWK_REQUIRE_PTR(d_out);
        if (loss != nullptr || loss_partials != nullptr) return RWK_E_NULL;
    }
    RWK_REQUIRE_PTR(k);
    RWK_REQUIRE_PTR(bn_w);
    RWK_REQUIRE_PTR(bn_b);
    RWK_REQUIRE_PTR(mean);
    RWK_REQUIRE_PTR(invstd);
    RWK_REQUIRE_PTR(w1);
    RWK_REQUIRE_PTR(w2);
    RWK_REQUIRE_PTR(h1d);
    RWK_REQUIRE_PTR(out);
    RWK_REQUIRE_PTR(dy);
    RWK_REQUIRE_PTR(partials);
    RWK_REQUIRE_PTR(grads);
    if (!rwk_head_supported(g, c, p, nc)) return RWK_E_UNSUPPORTED;
    if (drop_p < 0.f || drop_p >= 1.f) return RWK_E_SIZE;
    hipStream_t stream = (hipStream_t)stream_;
    const int blocks = head_blocks(g);
    const HeadDims dm{(long long)g, (int)c, (int)p, (int)nc};
    const float keep_scale = (training && drop_p > 0.f) ? 1.f / (1.f - drop_p) : 1.f;
    const size_t lds = bwd_lds((int)c, (int)p, (int)nc);
    const int rc = allow_lds(head_bwd_tile_kernel, lds);
    if (rc != RWK_OK) return rc;
    hipLaunchKernelGGL(head_bwd_tile_kernel, dim3(blocks), dim3(kBlock), lds, stream, k, dm, bn_w, bn_b, mean, invstd,
                       w1, w2, h1d, out, d_out, keep_scale, dy, partials, (const long long *)labels, d_loss,
                       loss_partials, blocks, loss);
    RWK_LAUNCH_CHECK();
    launch_finish(stream, k, dm, bn_w, mean, invstd, dy, partials, blocks, training, grads, d_k,
                  (const float *)nullptr, (const float *)nullptr, (float *)nullptr);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

int rwk_head_train_supported(int64_t g, int64_t c, int64_t p, int64_t nc) {
    return rwk_head_supported(g, c, p, nc) && g >= 2 && train_lds((int)c, (int)p, (int)nc) <= 150 * 1024 ? 1 : 0;
}

int rwk_head_train_forward(const float *k, int64_t g, int64_t c, int64_t p, int64_t nc, const float *bn_w,
                           const float *bn_b, float *running_mean, float *running_var, int64_t *num_batches_tracked,
                           float momentum, float eps, float *stats, int64_t n_stats_ready, const float *w1,
                           const float *b1, const float *w2, const float *b2, float drop_p, uint64_t *rng_state,
                           uint64_t *rng_used, float *mean_out, float *invstd_out, float *out, const int64_t *labels,
                           float *loss_partials, float *dy, float *partials, float *loss, rwk_stream_t stream_) {
    RWK_REQUIRE_PTR(k);
    RWK_REQUIRE_PTR(bn_w);
    RWK_REQUIRE_PTR(bn_b);
    RWK_REQUIRE_PTR(running_mean);
    RWK_REQUIRE_PTR(running_var);
    RWK_REQUIRE_PTR(stats);
    RWK_REQUIRE_PTR(w1);
    RWK_REQUIRE_PTR(b1);
    RWK_REQUIRE_PTR(w2);
    RWK_REQUIRE_PTR(b2);
    RWK_REQUIRE_PTR(mean_out);
    RWK_REQUIRE_PTR(invstd_out);
    RWK_REQUIRE_PTR(out);
    RWK_REQUIRE_PTR(labels);
    RWK_REQUIRE_PTR(loss_partials);
    RWK_REQUIRE_PTR(dy);
    RWK_REQUIRE_PTR(partials);
    if (!rwk_head_train_supported(g, c, p, nc)) return RWK_E_UNSUPPORTED;
    if (drop_p < 0.f || drop_p >= 1.f) return RWK_E_SIZE;
    if (n_stats_ready < 0 || n_stats_ready > kMaxFusedStatsBlocks) return RWK_E_SIZE;
    const bool drop = drop_p > 0.f;
    if (drop) {
        RWK_REQUIRE_PTR(rng_used);
        if (n_stats_ready == 0) RWK_REQUIRE_PTR(rng_state);
    }
    hipStream_t stream = (hipStream_t)stream_;
    const HeadDims dm{(long long)g, (int)c, (int)p, (int)nc};
    if (n_stats_ready == 0) {
        hipLaunchKernelGGL(head_stats_kernel, dim3(head_stats_blocks(g)), dim3(kBlock), 0, stream, k, (long long)g,
                           (int)c, stats, drop ? (unsigned long long *)rng_state : nullptr,
                           drop ? (unsigned long long *)rng_used : nullptr);
        RWK_LAUNCH_CHECK();
    }
    const int n_stats = n_stats_ready > 0 ? (int)n_stats_ready : head_stats_blocks(g);
    const size_t lds = train_lds((int)c, (int)p, (int)nc);
    const int rc = allow_lds(head_train_kernel, lds);
    if (rc != RWK_OK) return rc;
    // the reference's default widths with every vector operand 16-byte aligned: the one-wavefront-per-tile kernel
    const auto aligned16 = [](const void *q) { return ((uintptr_t)q & 15u) == 0; };
    if (c == kT32 && p == kT32 && nc <= kT32MaxNC && aligned16(k) && aligned16(bn_w) && aligned16(bn_b) &&
        aligned16(running_mean) && aligned16(running_var) && aligned16(stats) && aligned16(w1) && aligned16(b1) &&
        aligned16(w2) && aligned16(dy)) {
        auto kern = n_stats <= 64 ? head_train32_kernel<8> : head_train32_kernel<16>;
        hipLaunchKernelGGL(kern, dim3(head_blocks(g)), dim3(kT32Threads), 0, stream, k, dm, bn_w, bn_b,
                           running_mean, running_var, (long long *)num_batches_tracked, momentum, eps,
                           (const float *)stats, n_stats, w1, b1, w2, b2, drop_p, (const unsigned long long *)rng_used,
                           mean_out, invstd_out, out, (const long long *)labels, loss_partials, dy, partials, loss);
        RWK_LAUNCH_CHECK();
        return RWK_OK;
    }
    hipLaunchKernelGGL(head_train_kernel, dim3(head_blocks(g)), dim3(kBlock), lds, stream, k, dm, bn_w, bn_b,
                       running_mean, running_var, (long long *)num_batches_tracked, momentum, eps,
                       (const float *)stats, n_stats, w1, b1, w2, b2, drop_p, (const unsigned long long *)rng_used,
                       mean_out, invstd_out, out, (const long long *)labels, loss_partials, dy, partials, loss);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

int rwk_head_train_backward(const float *k, int64_t g, int64_t c, int64_t p, int64_t nc, const float *bn_w,
                            const float *mean, const float *invstd, const float *dy, const float *partials,
                            const float *d_loss, const float *loss_partials, float *loss, float *grads, float *d_k,
                            rwk_stream_t stream_) {
    RWK_REQUIRE_PTR(k);
    RWK_REQUIRE_PTR(bn_w);
    RWK_REQUIRE_PTR(mean);
    RWK_REQUIRE_PTR(invstd);
    RWK_REQUIRE_PTR(dy);
    RWK_REQUIRE_PTR(partials);
    RWK_REQUIRE_PTR(d_loss);
    RWK_REQUIRE_PTR(grads);
    if ((loss == nullptr) != (loss_partials == nullptr)) return RWK_E_NULL;
    if (!rwk_head_train_supported(g, c, p, nc)) return RWK_E_UNSUPPORTED;
    const int blocks = head_blocks(g);
    const HeadDims dm{(long long)g, (int)c, (int)p, (int)nc};
    launch_finish((hipStream_t)stream_, k, dm, bn_w, mean, invstd, dy, partials, blocks, 1, grads, d_k, d_loss,
                  loss_partials, loss);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

int rwk_cross_entropy_forward(const float *scores, const int64_t *labels, int64_t g, int64_t nc, float *loss,
                              rwk_stream_t stream) {
    RWK_REQUIRE_PTR(scores);
    RWK_REQUIRE_PTR(labels);
    RWK_REQUIRE_PTR(loss);
    if (g <= 0 || nc <= 0 || nc > 0x7FFFFFFF) return RWK_E_SIZE;
    hipLaunchKernelGGL(cross_entropy_forward_kernel, dim3(1), dim3(kCeBlock), 0, (hipStream_t)stream, scores,
                       (const long long *)labels, (long long)g, (int)nc, loss);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

int rwk_cross_entropy_backward(const float *scores, const int64_t *labels, int64_t g, int64_t nc,
                               const float *d_loss, float *d_scores, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(scores);
    RWK_REQUIRE_PTR(labels);
    RWK_REQUIRE_PTR(d_loss);
    RWK_REQUIRE_PTR(d_scores);
    if (g <= 0 || nc <= 0 || nc > 0x7FFFFFFF) return RWK_E_SIZE;
    hipLaunchKernelGGL(cross_entropy_backward_kernel, dim3((unsigned)rwk_ceil_div(g, kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, scores, (const long long *)labels, (long long)g, (int)nc, d_loss,
                       d_scores);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

}  // extern "C"
