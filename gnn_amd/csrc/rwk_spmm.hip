// CSR SpMM  y = A x (+ y)  for narrow dense operands (d = 1..64 floats per row).
//
// One "group" of VEC*R lanes owns one output row: VEC lanes cover the d floats of the row as
// float4 chunks (VEC = ceil(d/4)), R lanes split the row's stored entries round-robin and are
// summed with a fixed butterfly at the end, so the result does not depend on scheduling.
// R is chosen by the caller from the degree histogram: R = 1 (row per lane group, the right
// shape for molecular / protein graphs of degree 2..8), larger R for dense rows, and rows above
// `long_threshold` are taken out of the packed pass and given a whole wavefront each.
//
// HBM traffic per call (the figure the roofline uses): nnz*(4+4) + (n+1)*4 + 2*n*d*4 bytes.
#include "rwk_common.h"
#include "rwk_reduce_body.h"

namespace {

constexpr int kBlock = 256;

// VEC: float4 chunks per row (d padded to 4*VEC); ALIGNED: d == 4*VEC so rows are 16-byte aligned.
template <int VEC, bool ALIGNED>
__global__ void __launch_bounds__(kBlock)
spmm_csr_kernel(const int32_t *__restrict__ crow, const int32_t *__restrict__ col,
                const float *__restrict__ val, int n_rows_or_list, int d, const float *__restrict__ x,
                float *__restrict__ y, int accumulate, int r_lanes, const int32_t *__restrict__ row_list,
                int skip_above) {
    const int group_lanes = VEC * r_lanes;              // power of two, <= 64
    const long long gtid = (long long)blockIdx.x * kBlock + threadIdx.x;
    const int group = (int)(gtid / group_lanes);
    const int q = (int)(gtid % group_lanes);
    const int sub = q % VEC;                            // which float4 of the row
    const int r = q / VEC;                              // which share of the entries
    const bool active = group < n_rows_or_list;
    int row = 0, beg = 0, end = 0;
    if (active) {
        row = row_list ? row_list[group] : group;
        beg = crow[row];
        end = crow[row + 1];
    }
    const bool skipped = active && !row_list && skip_above > 0 && (end - beg) > skip_above;
    if (skipped) end = beg;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // Aligned rows, four entries of the lane's share per pass: their columns and values are requested together,
    // then the four x rows, then the sums in the order of the entries (a one-entry loop waits for each column before
    // it can ask for the x row behind it, two trips to memory per entry).  An entry past the row's end is loaded at
    // the pass's first index and enters the sum with the value 0.0f (a branch around it would put its loads behind
    // a wait).  Unaligned rows (four scalar loads per x row, fewer in the row's last chunk) measured no better
    // that way and keep the one-entry loop.
    if constexpr (ALIGNED) {
        constexpr int U = 4;
        for (int j = beg + r; j < end; j += U * r_lanes) {
            int c[U];
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int jj = j + u * r_lanes;
                const bool ok = jj < end;
                const int jc = ok ? jj : j;
                c[u] = col[jc];
                v[u] = val[jc] * (ok ? 1.f : 0.f);
            }
            float4 xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) xv[u] = *reinterpret_cast<const float4 *>(x + (size_t)c[u] * d + 4 * sub);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc.x = fmaf(v[u], xv[u].x, acc.x);
                acc.y = fmaf(v[u], xv[u].y, acc.y);
                acc.z = fmaf(v[u], xv[u].z, acc.z);
                acc.w = fmaf(v[u], xv[u].w, acc.w);
            }
        }
    } else {
        for (int j = beg + r; j < end; j += r_lanes) {
            const int c = col[j];
            const float v = val[j];
            const float *xr = x + (size_t)c * d;
            const int b = 4 * sub;
            float4 xv;
            xv.x = b + 0 < d ? xr[b + 0] : 0.f;
            xv.y = b + 1 < d ? xr[b + 1] : 0.f;
            xv.z = b + 2 < d ? xr[b + 2] : 0.f;
            xv.w = b + 3 < d ? xr[b + 3] : 0.f;
            acc.x = fmaf(v, xv.x, acc.x);
            acc.y = fmaf(v, xv.y, acc.y);
            acc.z = fmaf(v, xv.z, acc.z);
            acc.w = fmaf(v, xv.w, acc.w);
        }
    }
    // sum the R shares: lanes q and q ^ (VEC*k) hold the same float4 chunk
    for (int off = VEC; off < group_lanes; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off, 64);
        acc.y += __shfl_xor(acc.y, off, 64);
        acc.z += __shfl_xor(acc.z, off, 64);
        acc.w += __shfl_xor(acc.w, off, 64);
    }
    if (!active || skipped || r != 0) return;
    float *yr = y + (size_t)row * d;
    const int b = 4 * sub;
    if (ALIGNED) {
        float4 *yp = reinterpret_cast<float4 *>(yr + b);
        if (accumulate) {
            const float4 o = *yp;
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
        *yp = acc;
    } else {
        const float a[4] = {acc.x, acc.y, acc.z, acc.w};
        for (int k = 0; k < 4; ++k)
            if (b + k < d) yr[b + k] = accumulate ? yr[b + k] + a[k] : a[k];
    }
}

// ---------------------------------------------------------------------------------------------------
// What the tile kernel gathers and what it does with a finished row.  PlainOp is the SpMM itself: x rows in,
// y = A x (+ y) out.  ProjectGradOp lets the last transposed propagation step of the backward pass ride in the
// launch of the input projection's backward pass, which would otherwise run as a kernel of its own right after it.
// (The forward counterpart - x1 = A x0 with the neighbours' x0 rows recomputed from their feature rows inside the
// projection's launch - was built and measured: five sigmoid rows per node instead of one cost more than the
// launch it saved, 14.0 against 6.0 + 6.2 us on 1024 protein-sized graphs.)
// ---------------------------------------------------------------------------------------------------
struct PlainOp {
    const float *__restrict__ x;
    float *__restrict__ y;
    int d, accumulate;

    template <bool ALIGNED>
    __device__ inline float4 fetch(int c, int sub) const {
        const float *xr = x + (size_t)c * d;
        if (ALIGNED) return *reinterpret_cast<const float4 *>(xr + 4 * sub);
        float t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int b = 4 * sub + j;
            const float w = xr[b < d ? b : d - 1];
            t[j] = b < d ? w : 0.f;
        }
        return make_float4(t[0], t[1], t[2], t[3]);
    }

    template <bool ALIGNED>
    __device__ inline void finish(int /*block*/, int row, int sub, bool has_row, float4 acc, float * /*lds*/) const {
        if (!has_row) return;
        float *yr = y + (size_t)row * d;
        const int b = 4 * sub;
        if (ALIGNED) {
            float4 *yp = reinterpret_cast<float4 *>(yr + b);
            if (accumulate) {
                const float4 o = *yp;
                acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
            }
            *yp = acc;
        } else {
            const float a[4] = {acc.x, acc.y, acc.z, acc.w};
            for (int k = 0; k < 4; ++k)
                if (b + k < d) yr[b + k] = accumulate ? yr[b + k] + a[k] : a[k];
        }
    }
};

// hidden_dim 4, one lane per row; the matrix is A^T.  The finished row A^T dX_1 completes dX_0, whose only reader
// is the projection's backward pass (autograd of model.py:43): d_pre = dX_0 x0 (1 - x0), this row's terms of
// dW = d_pre^T features and db = sum d_pre, added over the workgroup's 256 rows (each wavefront in a fixed
// butterfly order, then the four wavefront sums in index order) into one partial row [dW (4, f) | db (4)].
template <int FP>
struct ProjectGradOp {
    const float *__restrict__ dx0, *__restrict__ dx1, *__restrict__ x0, *__restrict__ feat, *__restrict__ w;
    int f;
    float *__restrict__ partials, *__restrict__ d_feat;

    template <bool ALIGNED>
    __device__ inline float4 fetch(int c, int /*sub*/) const { return reinterpret_cast<const float4 *>(dx1)[c]; }

    template <bool ALIGNED>
    __device__ inline void finish(int block, int row, int /*sub*/, bool has_row, float4 acc, float *lds) const {
        float dpre[4] = {0.f, 0.f, 0.f, 0.f}, fr[FP];
#pragma unroll
        for (int k = 0; k < FP; ++k) fr[k] = 0.f;
        if (has_row) {
            const float4 g0 = reinterpret_cast<const float4 *>(dx0)[row];
            const float4 xv = reinterpret_cast<const float4 *>(x0)[row];
#pragma unroll
            for (int k = 0; k < FP; ++k)
                if (k < f) fr[k] = feat[(size_t)row * f + k];
            // dX_0 = (the fused kernel's own share) + A^T dX_1, as the accumulating SpMM forms it
            const float gt[4] = {g0.x + acc.x, g0.y + acc.y, g0.z + acc.z, g0.w + acc.w};
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) dpre[c] = gt[c] * xs[c] * (1.0f - xs[c]);
            if (d_feat != nullptr) {
#pragma unroll
                for (int k = 0; k < FP; ++k) {
                    if (k < f) {
                        float v = 0.f;
#pragma unroll
                        for (int c = 0; c < 4; ++c) v = fmaf(dpre[c], w[c * f + k], v);
                        d_feat[(size_t)row * f + k] = v;
                    }
                }
            }
        }
        const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int k = 0; k < FP; ++k) {
                const float t = rwk_group_sum(dpre[c] * fr[k], 64);
                if (lane == 0) lds[(c * FP + k) * 4 + wave] = t;
            }
            const float t = rwk_group_sum(dpre[c], 64);
            if (lane == 0) lds[(4 * FP + c) * 4 + wave] = t;
        }
        __syncthreads();
        const int n_out = 4 * f + 4;
        float *out = partials + (size_t)block * n_out;
        for (int o = tid; o < n_out; o += kBlock) {
            const int r = o < 4 * f ? (o / f) * FP + o % f : 4 * FP + (o - 4 * f);
            const float *p = lds + r * 4;
            out[o] = ((p[0] + p[1]) + p[2]) + p[3];
        }
    }
};

// ---------------------------------------------------------------------------------------------------
// Tile kernel for sparse rows (mean degree well below 64): a workgroup owns 256/VEC consecutive rows,
// hence one contiguous range of stored entries.  Phase 1 walks that range entry-parallel: column and
// value are read coalesced, each entry's x row is gathered by VEC lanes (every gather of a pass is in
// flight at once) and parked in LDS.  Phase 2 walks it row-parallel: the row's lanes add its entries
// from LDS in stored order (an fmaf chain, as a sequential CSR loop would).  Rows with more than
// kLongRow entries in the tile (power-law hubs) are summed by a wavefront each instead, in a fixed
// order (strided lane sums, then a butterfly).  y is written coalesced, one 16-byte chunk per lane.
constexpr int kTileBytes = 16 * 1024;
constexpr int kLongRow = 64;

//
// Compact structure (COLT = int16_t): a column is stored as its distance from the first row of the 256-row
// block its row lies in (block-diagonal batches of graphs below 32 k nodes fit 16 bits), and UNIT says that
// every stored value is exactly 1.0f (unweighted graphs), so no value array is read: 2 bytes per stored entry
// instead of 8.  acc + t is what fmaf(1.0f, t, acc) returns, so the results are the same bits.
// ---------------------------------------------------------------------------------------------------
constexpr int kColBlock = 256;

// bid / nblocks: this workgroup's index among the SpMM workgroups of the launch (they come first in a merged launch,
// so the dispatch order, and with it the XCD a workgroup lands on, is that of a stand-alone launch)
template <int VEC, bool ALIGNED, typename COLT, bool UNIT, typename OP>
__device__ inline void spmm_tile_body(int bid, int nblocks, const int32_t *__restrict__ crow,
                                      const COLT *__restrict__ col, const float *__restrict__ val, int n,
                                      const OP &op) {
    constexpr int RPW = kBlock / VEC;                   // rows per workgroup = entry slots per pass
    constexpr int ET = kTileBytes / (16 * VEC);         // entries per LDS tile
    constexpr int PASSES = ET / RPW;
    constexpr int GP = PASSES < 4 ? PASSES : 4;         // passes whose loads are issued together
    __shared__ float4 s_x[ET * VEC];
    __shared__ float s_val[UNIT ? 1 : ET];
    __shared__ float4 s_red[kBlock];
    __shared__ int s_crow[RPW + 1];
    __shared__ int s_long[RPW];
    __shared__ int s_nlong;

    const int tid = threadIdx.x;
    const int sub = tid % VEC, slot = tid / VEC;
    // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2: give every XCD one contiguous
    // eighth of the row blocks, so that neighbouring blocks (rows of the same graph, which gather the same
    // x rows) meet in one L2.
    const int per_xcd = nblocks / 8;
    const int lb = bid < per_xcd * 8 ? (bid % 8) * per_xcd + bid / 8 : bid;
    const int r0 = lb * RPW;
    const int col_base = sizeof(COLT) == 2 ? (r0 & ~(kColBlock - 1)) : 0;
    const int nrows = n - r0 < RPW ? n - r0 : RPW;
    // the block's RPW + 1 row pointers: one per thread and the last one, both requested before either is stored (as a
    // loop over i = tid, tid + 256 the second round - one thread's load - was a trip to memory of its own)
    {
        const int cv = crow[r0 + (tid <= nrows ? tid : nrows)];
        const int cl = crow[r0 + nrows];
        if (tid <= nrows) s_crow[tid] = cv;
        if (tid == 0) {
            s_crow[nrows] = cl;
            s_nlong = 0;
        }
    }
    __syncthreads();
    const int e0 = s_crow[0], e1 = s_crow[nrows];
    const bool has_row = slot < nrows;
    const int beg = has_row ? s_crow[slot] : 0;
    const int end = has_row ? s_crow[slot + 1] : 0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);

    // columns (and values) of the next tile's first pass group, requested behind the current group's gathers: a
    // workgroup with several tiles (hub rows) then waits one trip per further tile instead of two.  (In this form the
    // compiler also keeps the four column loads of a group together and the four gathers behind them; in the plain
    // loop it had peeled the first pass off as a dependent load pair of its own: 241 against 215 us on 19.7 M rows.)
    int cn[GP];
    float vn[GP];
    bool have_next = false;
    for (int cb = e0; cb < e1; cb += ET) {
        const int ce = cb + ET < e1 ? cb + ET : e1;
        // phase 1, four entry passes at a time: unconditional loads at clamped indices, so that the four
        // column/value loads and then the four gathers are issued back to back
#pragma unroll
        for (int g = 0; g < PASSES; g += GP) {
            if (cb + g * RPW >= ce) break;               // uniform
            int c[GP];
            float v[GP];
            if (g == 0 && have_next) {
#pragma unroll
                for (int k = 0; k < GP; ++k) { c[k] = cn[k]; v[k] = vn[k]; }
            } else {
#pragma unroll
                for (int k = 0; k < GP; ++k) {
                    const int e = cb + slot + (g + k) * RPW;
                    const int ec = e < ce ? e : ce - 1;
                    c[k] = (int)col[ec] + col_base;
                    v[k] = UNIT ? 1.f : val[ec];
                }
            }
            float4 xv[GP];
#pragma unroll
            for (int k = 0; k < GP; ++k) xv[k] = op.template fetch<ALIGNED>(c[k], sub);
            const bool last_group = g + GP >= PASSES || cb + (g + GP) * RPW >= ce;
            have_next = false;
            if (last_group && ce < e1) {                  // uniform
                const int nb = ce, ne = nb + ET < e1 ? nb + ET : e1;
#pragma unroll
                for (int k = 0; k < GP; ++k) {
                    const int e = nb + slot + k * RPW;
                    const int ec = e < ne ? e : ne - 1;
                    cn[k] = (int)col[ec] + col_base;
                    vn[k] = UNIT ? 1.f : val[ec];
                }
                have_next = true;
            }
#pragma unroll
            for (int k = 0; k < GP; ++k) {
                const int e = cb + slot + (g + k) * RPW;
                if (e < ce) {
                    s_x[(e - cb) * VEC + sub] = xv[k];
                    if (!UNIT && sub == 0) s_val[e - cb] = v[k];
                }
            }
        }
        __syncthreads();
        // phase 2
        const int lo = beg > cb ? beg : cb;
        const int hi = end < ce ? end : ce;
        const bool is_long = hi - lo > kLongRow;
        if (!is_long) {
            for (int e = lo; e < hi; ++e) {
                const float4 t = s_x[(e - cb) * VEC + sub];
                if (UNIT) {
                    acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
                } else {
                    const float w = s_val[e - cb];
                    acc.x = fmaf(w, t.x, acc.x);
                    acc.y = fmaf(w, t.y, acc.y);
                    acc.z = fmaf(w, t.z, acc.z);
                    acc.w = fmaf(w, t.w, acc.w);
                }
            }
        } else if (sub == 0) {
            s_long[atomicAdd(&s_nlong, 1)] = slot;       // LDS counter; the order of the list does not matter
        }
        __syncthreads();
        const int n_long = s_nlong;
        if (n_long > 0) {                                // uniform over the workgroup
            // a wavefront per long row: lane (el, sub) adds entries el, el + 64/VEC, ... of the row's part in this
            // tile, the lane sums of a chunk are added in a fixed butterfly (no barrier per row: the whole workgroup
            // with a tree through LDS took nine of them per row, 1 us each on power-law batches)
            constexpr int EL = RWK_WAVE / VEC;
            const int wave = tid >> 6, el = (tid & 63) / VEC;
            for (int k = wave; k < n_long; k += kBlock / RWK_WAVE) {
                const int rs = s_long[k];
                const int lo_k = s_crow[rs] > cb ? s_crow[rs] : cb;
                const int hi_k = s_crow[rs + 1] < ce ? s_crow[rs + 1] : ce;
                float4 part = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int e = lo_k + el; e < hi_k; e += EL) {
                    const float w = UNIT ? 1.f : s_val[e - cb];
                    const float4 t = s_x[(e - cb) * VEC + sub];
                    part.x = fmaf(w, t.x, part.x);
                    part.y = fmaf(w, t.y, part.y);
                    part.z = fmaf(w, t.z, part.z);
                    part.w = fmaf(w, t.w, part.w);
                }
#pragma unroll
                for (int off = VEC; off < RWK_WAVE; off <<= 1) {
                    part.x += __shfl_xor(part.x, off, 64);
                    part.y += __shfl_xor(part.y, off, 64);
                    part.z += __shfl_xor(part.z, off, 64);
                    part.w += __shfl_xor(part.w, off, 64);
                }
                if (el == 0) s_red[rs * VEC + sub] = part;
            }
            __syncthreads();
            if (is_long) {
                const float4 t = s_red[tid];
                acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
            }
        }
        if (tid == 0) s_nlong = 0;
        __syncthreads();
    }

    // s_red is free again: every use above is followed by a workgroup barrier
    op.template finish<ALIGNED>(lb, r0 + slot, sub, has_row, acc, reinterpret_cast<float *>(s_red));
}

template <int VEC, bool ALIGNED, typename COLT, bool UNIT>
__global__ void __launch_bounds__(kBlock)
spmm_tile_kernel(const int32_t *__restrict__ crow, const COLT *__restrict__ col, const float *__restrict__ val,
                 int n, int d, const float *__restrict__ x, float *__restrict__ y, int accumulate) {
    spmm_tile_body<VEC, ALIGNED, COLT, UNIT>(blockIdx.x, gridDim.x, crow, col, val, n, PlainOp{x, y, d, accumulate});
}

// The tile SpMM and the first level of an independent row reduction in one launch: workgroups [0, spmm_blocks)
// run the SpMM, the rest the reduction (red_bx column blocks x 32 row blocks).
template <int VEC, bool ALIGNED, typename COLT, bool UNIT>
__global__ void __launch_bounds__(kBlock)
spmm_tile_reduce_kernel(const int32_t *__restrict__ crow, const COLT *__restrict__ col, const float *__restrict__ val,
                        int n, int d, const float *__restrict__ x, float *__restrict__ y, int accumulate,
                        int spmm_blocks, const float *__restrict__ red_in, long long red_p, long long red_v,
                        int red_rows_per_block, int red_bx, float *__restrict__ red_out) {
    if ((int)blockIdx.x < spmm_blocks) {
        spmm_tile_body<VEC, ALIGNED, COLT, UNIT>(blockIdx.x, spmm_blocks, crow, col, val, n,
                                                 PlainOp{x, y, d, accumulate});
    } else {
        const int r = (int)blockIdx.x - spmm_blocks;
        reduce_rows_level1_body(r % red_bx, r / red_bx, red_in, red_p, red_v, red_rows_per_block, red_out);
    }
}

// The last transposed propagation step inside the projection's backward pass (tile body with ProjectGradOp) and the
// first level of the dZ partial-sum reduction in one launch.
template <typename COLT, bool UNIT, int FP>
__global__ void __launch_bounds__(kBlock)
spmm_project_backward_kernel(const int32_t *__restrict__ crow, const COLT *__restrict__ col,
                             const float *__restrict__ val, int n, int spmm_blocks, ProjectGradOp<FP> op,
                             const float *__restrict__ red_in, long long red_p, long long red_v, int red_rows_per_block,
                             int red_bx, float *__restrict__ red_out) {
    if ((int)blockIdx.x < spmm_blocks) {
        spmm_tile_body<1, true, COLT, UNIT>(blockIdx.x, spmm_blocks, crow, col, val, n, op);
    } else {
        const int r = (int)blockIdx.x - spmm_blocks;
        reduce_rows_level1_body(r % red_bx, r / red_bx, red_in, red_p, red_v, red_rows_per_block, red_out);
    }
}

struct ReduceJob {
    const float *in;
    long long p, v;
    float *out32;
};

template <int VEC, typename COLT>
int launch_tile(const int32_t *crow, const COLT *col, const float *val, int64_t n, int64_t d, const float *x,
                float *y, int accumulate, hipStream_t stream, const ReduceJob *job) {
    const bool aligned = (d == 4 * VEC) && ((((uintptr_t)x | (uintptr_t)y) & 15u) == 0);
    const unsigned spmm_blocks = (unsigned)rwk_ceil_div(n, kBlock / VEC);
    if (job == nullptr) {
        const dim3 grid(spmm_blocks);
#define RWK_TILE(AL, UN)                                                                                        \
    hipLaunchKernelGGL((spmm_tile_kernel<VEC, AL, COLT, UN>), grid, dim3(kBlock), 0, stream, crow, col, val, (int)n, \
                       (int)d, x, y, accumulate)
        if (val == nullptr) { if (aligned) RWK_TILE(true, true); else RWK_TILE(false, true); }
        else { if (aligned) RWK_TILE(true, false); else RWK_TILE(false, false); }
#undef RWK_TILE
    } else {
        const int red_bx = (int)rwk_ceil_div(job->v, 64);
        const int rows_per_block = (int)rwk_ceil_div(job->p, kReduceMaxFinalRows);
        const dim3 grid(spmm_blocks + (unsigned)red_bx * kReduceMaxFinalRows);
#define RWK_TILE(AL, UN)                                                                                        \
    hipLaunchKernelGGL((spmm_tile_reduce_kernel<VEC, AL, COLT, UN>), grid, dim3(kBlock), 0, stream, crow, col, val, \
                       (int)n, (int)d, x, y, accumulate, (int)spmm_blocks, job->in, job->p, job->v, rows_per_block, \
                       red_bx, job->out32)
        if (val == nullptr) { if (aligned) RWK_TILE(true, true); else RWK_TILE(false, true); }
        else { if (aligned) RWK_TILE(true, false); else RWK_TILE(false, false); }
#undef RWK_TILE
    }
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

template <typename COLT>
int launch_tile_by_width(const int32_t *crow, const COLT *col, const float *val, int64_t n, int64_t d, const float *x,
                         float *y, int accumulate, hipStream_t stream, const ReduceJob *job = nullptr) {
    const int vec = (int)((d + 3) / 4);
    if (vec == 1) return launch_tile<1, COLT>(crow, col, val, n, d, x, y, accumulate, stream, job);
    if (vec == 2) return launch_tile<2, COLT>(crow, col, val, n, d, x, y, accumulate, stream, job);
    if (vec <= 4) return launch_tile<4, COLT>(crow, col, val, n, d, x, y, accumulate, stream, job);
    if (vec <= 8) return launch_tile<8, COLT>(crow, col, val, n, d, x, y, accumulate, stream, job);
    return launch_tile<16, COLT>(crow, col, val, n, d, x, y, accumulate, stream, job);
}

// col16[e] = col[e] - (row & ~255); flags[0] counts entries that do not fit, flags[1] values other than 1.0f
__global__ void __launch_bounds__(kBlock)
csr_compact_kernel(const int32_t *__restrict__ crow, const int32_t *__restrict__ col, const float *__restrict__ val,
                   int n, int16_t *__restrict__ col16, int32_t *__restrict__ flags) {
    const int row = blockIdx.x * kBlock + threadIdx.x;
    if (row >= n) return;
    const int base = row & ~(kColBlock - 1);
    int bad = 0, weighted = 0;
    for (int e = crow[row]; e < crow[row + 1]; ++e) {
        const int dlt = col[e] - base;
        if (dlt < -32768 || dlt > 32767) ++bad;
        col16[e] = (int16_t)dlt;
        if (val[e] != 1.0f) ++weighted;
    }
    if (bad) atomicAdd(&flags[0], bad);
    if (weighted) atomicAdd(&flags[1], weighted);
}

template <int VEC>
int launch(const int32_t *crow, const int32_t *col, const float *val, int64_t n, int64_t d, const float *x,
           float *y, int accumulate, int r_lanes, const int32_t *long_rows, int64_t n_long,
           int32_t long_threshold, hipStream_t stream) {
    const bool aligned = (d == 4 * VEC) && ((((uintptr_t)x | (uintptr_t)y) & 15u) == 0);
    if (r_lanes * VEC > 64) r_lanes = 64 / VEC;
    const int skip = (long_rows && n_long > 0) ? long_threshold : 0;
    {
        const int64_t threads = n * VEC * r_lanes;
        const dim3 grid((unsigned)rwk_ceil_div(threads, kBlock));
        if (aligned)
            hipLaunchKernelGGL((spmm_csr_kernel<VEC, true>), grid, dim3(kBlock), 0, stream, crow, col, val, (int)n,
                               (int)d, x, y, accumulate, r_lanes, (const int32_t *)nullptr, skip);
        else
            hipLaunchKernelGGL((spmm_csr_kernel<VEC, false>), grid, dim3(kBlock), 0, stream, crow, col, val, (int)n,
                               (int)d, x, y, accumulate, r_lanes, (const int32_t *)nullptr, skip);
        RWK_LAUNCH_CHECK();
    }
    if (skip > 0) {
        const int rl = 64 / VEC;                       // a whole wavefront per long row
        const int64_t threads = n_long * 64;
        const dim3 grid((unsigned)rwk_ceil_div(threads, kBlock));
        if (aligned)
            hipLaunchKernelGGL((spmm_csr_kernel<VEC, true>), grid, dim3(kBlock), 0, stream, crow, col, val,
                               (int)n_long, (int)d, x, y, accumulate, rl, long_rows, 0);
        else
            hipLaunchKernelGGL((spmm_csr_kernel<VEC, false>), grid, dim3(kBlock), 0, stream, crow, col, val,
                               (int)n_long, (int)d, x, y, accumulate, rl, long_rows, 0);
        RWK_LAUNCH_CHECK();
    }
    return RWK_OK;
}

}  // namespace

extern "C" int rwk_spmm_csr(const int32_t *crow, const int32_t *col, const float *val, int64_t n, int64_t d,
                            const float *x, float *y, int accumulate, int32_t row_lanes,
                            const int32_t *long_rows, int64_t n_long, int32_t long_threshold,
                            rwk_stream_t stream_) {
    RWK_REQUIRE_PTR(crow);
    RWK_REQUIRE_PTR(x);
    RWK_REQUIRE_PTR(y);
    if (n <= 0 || n >= 0x7FFFFFFFll || d <= 0 || n_long < 0 || n_long > n) return RWK_E_SIZE;
    if (d > 64) return RWK_E_UNSUPPORTED;
    if (row_lanes < 0 || row_lanes > 64 || (row_lanes & (row_lanes - 1))) return RWK_E_SIZE;
    if (n * 64 >= (int64_t)0x7FFFFFFF * kBlock) return RWK_E_SIZE;
    hipStream_t stream = (hipStream_t)stream_;
    const int vec = (int)((d + 3) / 4);
    if (row_lanes == 0) {                               // sparse rows: the tile kernel (hubs handled inside)
        return launch_tile_by_width<int32_t>(crow, col, val, n, d, x, y, accumulate, stream);
    }
    if (vec == 1) return launch<1>(crow, col, val, n, d, x, y, accumulate, row_lanes, long_rows, n_long, long_threshold, stream);
    if (vec == 2) return launch<2>(crow, col, val, n, d, x, y, accumulate, row_lanes, long_rows, n_long, long_threshold, stream);
    if (vec <= 4) return launch<4>(crow, col, val, n, d, x, y, accumulate, row_lanes, long_rows, n_long, long_threshold, stream);
    if (vec <= 8) return launch<8>(crow, col, val, n, d, x, y, accumulate, row_lanes, long_rows, n_long, long_threshold, stream);
    return launch<16>(crow, col, val, n, d, x, y, accumulate, row_lanes, long_rows, n_long, long_threshold, stream);
}

extern "C" int rwk_csr_compact(const int32_t *crow, const int32_t *col, const float *val, int64_t n, int16_t *col16,
                               int32_t *flags, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(crow);
    RWK_REQUIRE_PTR(col);
    RWK_REQUIRE_PTR(val);
    RWK_REQUIRE_PTR(col16);
    RWK_REQUIRE_PTR(flags);
    if (n <= 0 || n >= 0x7FFFFFFFll) return RWK_E_SIZE;
    hipLaunchKernelGGL(csr_compact_kernel, dim3((unsigned)rwk_ceil_div(n, kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, crow, col, val, (int)n, col16, flags);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

extern "C" int rwk_spmm_csr_compact(const int32_t *crow, const int16_t *col16, const float *val, int64_t n, int64_t d,
                                    const float *x, float *y, int accumulate, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(crow);
    RWK_REQUIRE_PTR(col16);
    RWK_REQUIRE_PTR(x);
    RWK_REQUIRE_PTR(y);
    if (n <= 0 || n >= 0x7FFFFFFFll || d <= 0) return RWK_E_SIZE;
    if (d > 64) return RWK_E_UNSUPPORTED;
    return launch_tile_by_width<int16_t>(crow, col16, val, n, d, x, y, accumulate, (hipStream_t)stream);
}

extern "C" int rwk_spmm_tile_reduce_rows(const int32_t *crow, const int32_t *col, const int16_t *col16, const float *val,
                                         int64_t n, int64_t d, const float *x, float *y, int accumulate,
                                         const float *red_in, int64_t red_p, int64_t red_v, float *red_out32,
                                         rwk_stream_t stream) {
    RWK_REQUIRE_PTR(crow);
    RWK_REQUIRE_PTR(x);
    RWK_REQUIRE_PTR(y);
    RWK_REQUIRE_PTR(red_in);
    RWK_REQUIRE_PTR(red_out32);
    if (col == nullptr && col16 == nullptr) return RWK_E_NULL;
    if (n <= 0 || n >= 0x7FFFFFFFll || d <= 0) return RWK_E_SIZE;
    if (d > 64) return RWK_E_UNSUPPORTED;
    if (red_p <= kReduceMaxFinalRows || red_v <= 0 || red_v >= 0x7FFFFFFFll) return RWK_E_SIZE;
    if (rwk_ceil_div(n, 16) + rwk_ceil_div(red_v, 64) * kReduceMaxFinalRows >= 0x7FFFFFFFll) return RWK_E_SIZE;
    const ReduceJob job{red_in, (long long)red_p, (long long)red_v, red_out32};
    if (col16 != nullptr)
        return launch_tile_by_width<int16_t>(crow, col16, val, n, d, x, y, accumulate, (hipStream_t)stream, &job);
    return launch_tile_by_width<int32_t>(crow, col, val, n, d, x, y, accumulate, (hipStream_t)stream, &job);
}

// FP = padded input width the projection bodies are instantiated for
#define RWK_BY_FP(f, M)            \
    do {                           \
        if ((f) == 1) { M(1); }    \
        else if ((f) <= 4) { M(4); } \
        else { M(16); }            \
    } while (0)

extern "C" int64_t rwk_spmm_project_backward_blocks(int64_t n) { return n > 0 ? rwk_ceil_div(n, kBlock) : 0; }

extern "C" int rwk_spmm_project_backward(const float *dx0, const float *dx1, const float *x0, const float *features,
                                         const float *weight, int64_t n, int64_t f, const int32_t *crow_t,
                                         const int32_t *col_t, const int16_t *col16_t, const float *val_t,
                                         float *partials, float *d_features, const float *red_in, int64_t red_p,
                                         int64_t red_v, float *red_out32, rwk_stream_t stream_) {
    RWK_REQUIRE_PTR(dx0);
    RWK_REQUIRE_PTR(dx1);
    RWK_REQUIRE_PTR(x0);
    RWK_REQUIRE_PTR(features);
    RWK_REQUIRE_PTR(weight);
    RWK_REQUIRE_PTR(crow_t);
    RWK_REQUIRE_PTR(partials);
    if (col_t == nullptr && col16_t == nullptr) return RWK_E_NULL;
    if (n <= 0 || f <= 0 || n >= 0x7FFFFFFFll / 4) return RWK_E_SIZE;
    if (f > 16) return RWK_E_UNSUPPORTED;
    if ((((uintptr_t)dx0 | (uintptr_t)dx1 | (uintptr_t)x0) & 15u) != 0) return RWK_E_ALIGN;
    const int spmm_blocks = (int)rwk_ceil_div(n, kBlock);
    int red_bx = 0, rows_per_block = 0;
    if (red_in != nullptr) {
        RWK_REQUIRE_PTR(red_out32);
        if (red_p <= kReduceMaxFinalRows || red_v <= 0 || red_v >= 0x7FFFFFFFll * 64) return RWK_E_SIZE;
        red_bx = (int)rwk_ceil_div(red_v, 64);
        rows_per_block = (int)rwk_ceil_div(red_p, kReduceMaxFinalRows);
    }
    const dim3 grid((unsigned)(spmm_blocks + red_bx * kReduceMaxFinalRows));
    hipStream_t stream = (hipStream_t)stream_;
#define RWK_SPB_K(COLT, COLP, UN, FP)                                                                                \
    hipLaunchKernelGGL((spmm_project_backward_kernel<COLT, UN, FP>), grid, dim3(kBlock), 0, stream, crow_t, COLP, val_t, \
                       (int)n, spmm_blocks,                                                                           \
                       (ProjectGradOp<FP>{dx0, dx1, x0, features, weight, (int)f, partials, d_features}), red_in,     \
                       (long long)red_p, (long long)red_v, rows_per_block, red_bx > 0 ? red_bx : 1, red_out32)
#define RWK_SPB(FP)                                                              \
    do {                                                                         \
        if (col16_t != nullptr) {                                                \
            if (val_t == nullptr) RWK_SPB_K(int16_t, col16_t, true, FP);         \
            else RWK_SPB_K(int16_t, col16_t, false, FP);                         \
        } else {                                                                 \
            if (val_t == nullptr) RWK_SPB_K(int32_t, col_t, true, FP);           \
            else RWK_SPB_K(int32_t, col_t, false, FP);                           \
        }                                                                        \
    } while (0)
    RWK_BY_FP(f, RWK_SPB);
#undef RWK_SPB
#undef RWK_SPB_K
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}
#undef RWK_BY_FP
