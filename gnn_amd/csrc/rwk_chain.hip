// The propagation chain of the layer, X_i = A X_{i-1} (rwgnn2/model.py:55) and its backward dX_{i-1} += A^T dX_i, for
// block-diagonal batches of LARGE graphs (thousands of nodes, power-law hubs) at hidden_dim 4: one workgroup per
// graph, the graph - its rows AND its structure - in LDS for the whole chain.
//
// As separate launches every step of the chain is an SpMM whose workgroups wait three dependent trips to memory
// (row pointers, columns, gathered rows) and whose launch does nothing until the previous step has drained: on 64
// graphs of 3000 nodes and four steps the three forward products and the three transposed ones are 70 of the
// training step's 175 us, each launch at 14 % of the HBM rate.  A node's neighbours are nodes of its own graph, so a
// workgroup that holds its graph in LDS walks the whole chain without leaving the chip: 16 bytes per node for the
// rows of the current step, 4 for the row pointers, 2 per stored entry for the columns (as node numbers within the
// graph), loaded once.  The rows travel to memory once per step (forward, for the fused kernels and the backward
// pass) or not at all (backward: only the projection's gradient sums leave).  Entries beyond the LDS share of the
// columns are read from memory (same sums, slower).
//
// A step reads the rows of the previous one from the ONE row buffer and keeps its results elsewhere until every
// read is done (a barrier), then writes them back (another barrier): a thread's own rows in registers (at most four
// per thread), the rows of the cooperative classes in a staging block.  Three classes of rows, by stored entries:
//   <= 16   the row's own thread, an fmaf chain in stored order (the order of the tile SpMM: same bits);
//   <= 128  a group of 16 lanes: lane j adds entries j, j + 16, ... in order, the lane sums in a fixed butterfly;
//   above   a wavefront, the same with 64 lanes.
// The cooperative rows are listed once per launch, in thread order, and the other rows are dealt to the threads sorted
// by length class (graph_lists: a workgroup-wide prefix sum, not an atomic counter), so neither which rows fit the lists
// - the others fall back to a thread - nor anything else depends on scheduling: all sums in fixed orders, bitwise
// reproducible.
#include "rwk_common.h"
#include "rwk_hidden_body.h"
#include "rwk_reduce_body.h"

namespace {

constexpr int kShortRow = 16;
constexpr int kGroupRow = 128;
constexpr int kRpt = 4;                           // rows per thread (graphs of at most 4 * THREADS nodes)
constexpr int kCapA = 768, kCapB = 256;           // cooperative rows listed per graph (16-lane groups, wavefronts)
constexpr size_t kLdsBudget = 150 * 1024;

__device__ inline float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

__device__ inline void f4_fma(float w, const float4 &r, float4 &acc) {
    acc.x = fmaf(w, r.x, acc.x);
    acc.y = fmaf(w, r.y, acc.y);
    acc.z = fmaf(w, r.z, acc.z);
    acc.w = fmaf(w, r.w, acc.w);
}

__device__ inline void f4_add(float4 &a, const float4 &b) {
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
}

// LDS sizes of one graph's view, shared by host and device
struct ChainLayout {
    int max_nodes, cap_a, cap_b, cap_cols, extra_floats, threads;
    __host__ __device__ size_t bytes() const {
        return (size_t)max_nodes * 16 + (size_t)(cap_a + cap_b) * 16 + (size_t)(max_nodes + 1) * 4 +
               (size_t)(4 * (threads / RWK_WAVE) + 4) * 4 + (size_t)extra_floats * 4 +
               (size_t)(cap_a + cap_b + max_nodes) * 2 + (size_t)cap_cols * 2;
    }
};

inline int chain_threads(int64_t max_nodes) { return max_nodes > 1024 ? 1024 : 256; }

// extra_floats: what the kernel keeps beside the view (projection weights; wavefront sums)
inline bool chain_layout(int64_t max_nodes, int extra_floats, ChainLayout *out) {
    if (max_nodes < 1 || max_nodes > 4096) return false;
    ChainLayout l;
    l.max_nodes = (int)max_nodes;
    l.threads = chain_threads(max_nodes);
    l.cap_a = max_nodes < kCapA ? (int)max_nodes : kCapA;
    l.cap_b = max_nodes < kCapB ? (int)max_nodes : kCapB;
    l.extra_floats = extra_floats;
    l.cap_cols = 0;
    const size_t fixed = l.bytes() + 64;          // (alignment slack)
    if (fixed + 2 * 4 * (size_t)max_nodes > kLdsBudget) return false;       // room for at least four entries per node
    size_t cols = (kLdsBudget - fixed) / 2;
    const size_t want = (size_t)max_nodes * 16;   // mean degree 16 and the workgroups of small graphs share a CU
    if (cols > want) cols = want;
    l.cap_cols = (int)(cols & ~(size_t)7);
    *out = l;
    return true;
}

struct GraphView {
    float4 *x;                 // [max_nodes] rows of the current step
    float4 *stage;             // [cap_a + cap_b] results of the listed rows, until every read of x is done
    int *crow;                 // [max_nodes + 1] row pointers relative to the graph's first entry
    int *misc;                 // [4 * waves] wavefront totals of the prefix sum (seven 16-bit counters in two words)
    float *extra;
    unsigned short *list_a, *list_b;
    unsigned short *perm;      // [max_nodes] the rows summed by their own thread, sorted by length class
    unsigned short *cols;
    int cap_a, cap_b, cap_cols;
    const int *gcol;
    const float *gval;
    int p0, ng, eb, n_a, n_b, n_own;
};

template <int THREADS>
__device__ inline GraphView graph_view(float *lds, const ChainLayout &l, const int *gcol, const float *gval) {
    GraphView v;
    v.x = reinterpret_cast<float4 *>(lds);
    v.stage = v.x + l.max_nodes;
    v.crow = reinterpret_cast<int *>(v.stage + l.cap_a + l.cap_b);
    v.misc = v.crow + l.max_nodes + 1;
    v.extra = reinterpret_cast<float *>(v.misc + 4 * (THREADS / RWK_WAVE) + 4);
    v.list_a = reinterpret_cast<unsigned short *>(v.extra + l.extra_floats);
    v.list_b = v.list_a + l.cap_a;
    v.perm = v.list_b + l.cap_b;
    v.cols = v.perm + l.max_nodes;
    v.cap_a = l.cap_a; v.cap_b = l.cap_b; v.cap_cols = l.cap_cols;
    v.gcol = gcol; v.gval = gval;
    v.p0 = v.ng = v.eb = v.n_a = v.n_b = v.n_own = 0;
    return v;
}

// Row pointers and columns of graph rows [p0, p0 + ng) into LDS; the caller's barrier follows.  One workgroup has a
// whole CU to itself and nothing else to hide a trip to memory behind, so what counts is bytes in flight per thread:
// every row pointer of a thread is requested before the first is used, the columns travel as 16-byte loads, four per
// thread at a time (one 4-byte load per thread and trip took 10 us for the 40 KB of a 3000-node graph).
template <int THREADS>
__device__ inline void graph_load(GraphView &v, const int *__restrict__ crow, int p0, int ng) {
    const int tid = threadIdx.x;
    v.p0 = p0; v.ng = ng;
    int t[kRpt + 1];
#pragma unroll
    for (int r = 0; r <= kRpt; ++r) {
        const int i = tid + r * THREADS;
        t[r] = crow[p0 + (i <= ng ? i : ng)];
    }
    v.eb = crow[p0];
    const int gnnz = crow[p0 + ng] - v.eb;
#pragma unroll
    for (int r = 0; r <= kRpt; ++r) {
        const int i = tid + r * THREADS;
        if (i <= ng) v.crow[i] = t[r] - v.eb;
    }
    const int n_lds = gnnz < v.cap_cols ? gnnz : v.cap_cols;
    // entries [s, s + 4 nfull): whole 16-byte words of the column array; up to three entries before and after them
    const int mis = (int)(((uintptr_t)v.gcol >> 2) & 3);
    const int end = v.eb + n_lds;
    int s = v.eb + ((4 - ((mis + v.eb) & 3)) & 3);
    if (s > end) s = end;
    const int nfull = (end - s) >> 2;
    const int4 *base = reinterpret_cast<const int4 *>(v.gcol + s);
    for (int q0 = tid; q0 < nfull; q0 += 4 * THREADS) {
        int4 w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = q0 + u * THREADS;
            w[u] = base[q < nfull ? q : nfull - 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = q0 + u * THREADS;
            if (q < nfull) {
                unsigned short *d = v.cols + (s - v.eb) + 4 * q;
                d[0] = (unsigned short)(w[u].x - p0);
                d[1] = (unsigned short)(w[u].y - p0);
                d[2] = (unsigned short)(w[u].z - p0);
                d[3] = (unsigned short)(w[u].w - p0);
            }
        }
    }
    const int head = s - v.eb, tail0 = s + 4 * nfull;
    if (tid < head) v.cols[tid] = (unsigned short)(v.gcol[v.eb + tid] - p0);
    if (tid < end - tail0) v.cols[tail0 - v.eb + tid] = (unsigned short)(v.gcol[tail0 + tid] - p0);
}

__device__ inline int col_at(const GraphView &v, int e) {
    return e < v.cap_cols ? (int)v.cols[e] : v.gcol[v.eb + e] - v.p0;
}

// sum over the entries [e0, e1) taken with stride `stride` from e0 + first: four entries in flight
__device__ inline float4 entries_sum(const GraphView &v, int first, int e1, int stride) {
    float4 acc = f4_zero();
    for (int e = first; e < e1; e += 4 * stride) {
        int c[4];
        float w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int at = e + u * stride < e1 ? e + u * stride : e1 - 1;
            c[u] = col_at(v, at);
            w[u] = v.gval ? v.gval[v.eb + at] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (e + u * stride < e1) f4_fma(w[u], v.x[c[u]], acc);
    }
    return acc;
}

template <int W>
__device__ inline float4 lanes_sum(float4 p) {
    p.x = rwk_group_sum(p.x, W);
    p.y = rwk_group_sum(p.y, W);
    p.z = rwk_group_sum(p.z, W);
    p.w = rwk_group_sum(p.w, W);
    return p;
}

// Who sums which row.  Rows of more than kShortRow entries go to the two lists of cooperative rows (in thread order; a
// row the lists have no room for stays with a thread).  All other rows are dealt to the threads SORTED by the number of
// four-entry batches their sum takes (longest first): slot q of the order is summed by thread q % THREADS in its round
// q / THREADS, so the 64 rows a wavefront sums in one round are of one length class - in natural order most lanes of
// a wavefront idle while the longest of its 64 rows finishes (power-law graphs: most rows have one or two entries;
// a step of a 3000-node graph took 6 us that way).  Positions come from one workgroup-wide prefix sum over seven
// counters (no atomic counter: the order does not depend on scheduling).  Barriers inside.
template <int THREADS>
__device__ inline void graph_lists(GraphView &v) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int kWaves = THREADS / RWK_WAVE;
    // counters, 16 bits each: lo = own rows of 4, 3, 2, 1 batches; hi = own rows without entries, list A, list B
    int key[kRpt];
    unsigned long long lo = 0, hi = 0;
#pragma unroll
    for (int r = 0; r < kRpt; ++r) {
        const int local = tid + r * THREADS;
        key[r] = -1;
        if (local < v.ng) {
            const int len = v.crow[local + 1] - v.crow[local];
            key[r] = len > kGroupRow ? 6 : (len > kShortRow ? 5 : 4 - ((len + 3) >> 2));
            if (key[r] < 4) lo += 1ull << (16 * key[r]);
            else hi += 1ull << (16 * (key[r] - 4));
        }
    }
    unsigned long long ilo = lo, ihi = hi;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long tl = __shfl_up(ilo, off, 64), th = __shfl_up(ihi, off, 64);
        if (lane >= off) { ilo += tl; ihi += th; }
    }
    if (lane == 63) {
        v.misc[4 * wave + 0] = (int)(unsigned)ilo;
        v.misc[4 * wave + 1] = (int)(unsigned)(ilo >> 32);
        v.misc[4 * wave + 2] = (int)(unsigned)ihi;
        v.misc[4 * wave + 3] = (int)(unsigned)(ihi >> 32);
    }
    __syncthreads();
    unsigned long long blo = ilo - lo, bhi = ihi - hi, tlo = 0, thi = 0;
#pragma unroll
    for (int q = 0; q < kWaves; ++q) {
        const unsigned long long wl = (unsigned)v.misc[4 * q] | ((unsigned long long)(unsigned)v.misc[4 * q + 1] << 32);
        const unsigned long long wh = (unsigned)v.misc[4 * q + 2] | ((unsigned long long)(unsigned)v.misc[4 * q + 3] << 32);
        if (q < wave) { blo += wl; bhi += wh; }
        tlo += wl; thi += wh;
    }
    int before[7], total[7];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        before[k] = (int)((blo >> (16 * k)) & 0xFFFF);
        total[k] = (int)((tlo >> (16 * k)) & 0xFFFF);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        before[4 + k] = (int)((bhi >> (16 * k)) & 0xFFFF);
        total[4 + k] = (int)((thi >> (16 * k)) & 0xFFFF);
    }
    const int over_a = total[5] > v.cap_a ? total[5] - v.cap_a : 0;
    const int over_b = total[6] > v.cap_b ? total[6] - v.cap_b : 0;
    int base[5];
    base[0] = over_a + over_b;                     // the rows the lists had no room for come first: they are the longest
#pragma unroll
    for (int k = 1; k < 5; ++k) base[k] = base[k - 1] + total[k - 1];
#pragma unroll
    for (int r = 0; r < kRpt; ++r) {
        const int local = tid + r * THREADS;
        const int k = key[r];
        if (k < 0) continue;
        int at = 0;
#pragma unroll
        for (int q = 0; q < 7; ++q)
            if (q == k) at = before[q]++;
        if (k < 5) {
            int b0 = 0;
#pragma unroll
            for (int q = 0; q < 5; ++q)
                if (q == k) b0 = base[q];
            v.perm[b0 + at] = (unsigned short)local;
        } else if (k == 5) {
            if (at < v.cap_a) v.list_a[at] = (unsigned short)local;
            else v.perm[at - v.cap_a] = (unsigned short)local;
        } else {
            if (at < v.cap_b) v.list_b[at] = (unsigned short)local;
            else v.perm[over_a + at - v.cap_b] = (unsigned short)local;
        }
    }
    v.n_a = total[5] - over_a;
    v.n_b = total[6] - over_b;
    v.n_own = base[4] + total[4];
    __syncthreads();
}

// the rows this thread sums itself (-1: none), constant over the steps of a launch
template <int THREADS>
__device__ inline void graph_own_rows(const GraphView &v, int (&loc)[kRpt]) {
#pragma unroll
    for (int r = 0; r < kRpt; ++r) {
        const int slot = threadIdx.x + r * THREADS;
        loc[r] = slot < v.n_own ? (int)v.perm[slot] : -1;
    }
}

// One product step on the view: res[r] = (A x)[loc[r]] (+ own(row)), listed rows into the staging block.
// OWN: float4 operator()(int local) - the addend of a row (the step's own gradient share in the backward chain).
template <int THREADS, bool ADD, class OWN>
__device__ inline void graph_step(const GraphView &v, const int (&loc)[kRpt], float4 (&res)[kRpt], const OWN &own) {
    const int tid = threadIdx.x;
    float4 o[kRpt];
#pragma unroll
    for (int r = 0; r < kRpt; ++r) {              // requested together, used after the row's sum
        o[r] = f4_zero();
        if (ADD && loc[r] >= 0) o[r] = own(loc[r]);
    }
#pragma unroll
    for (int r = 0; r < kRpt; ++r) {
        if (loc[r] < 0) continue;
        const int e0 = v.crow[loc[r]], e1 = v.crow[loc[r] + 1];
        float4 acc = entries_sum(v, e0, e1, 1);
        if (ADD) f4_add(acc, o[r]);               // as the accumulating SpMM adds it: (sum over the row) + y
        res[r] = acc;
    }
    {
        const int gid = tid >> 4, j = tid & 15;
        for (int k = gid; k < v.n_a; k += THREADS / 16) {
            const int local = v.list_a[k];
            float4 o = f4_zero();
            if (ADD && j == 0) o = own(local);
            const int e0 = v.crow[local], e1 = v.crow[local + 1];
            float4 acc = lanes_sum<16>(entries_sum(v, e0 + j, e1, 16));
            if (ADD) f4_add(acc, o);
            if (j == 0) v.stage[k] = acc;
        }
    }
    {
        const int wave = tid >> 6, lane = tid & 63;
        for (int k = wave; k < v.n_b; k += THREADS / RWK_WAVE) {
            const int local = v.list_b[k];
            float4 o = f4_zero();
            if (ADD && lane == 0) o = own(local);
            const int e0 = v.crow[local], e1 = v.crow[local + 1];
            float4 acc = lanes_sum<64>(entries_sum(v, e0 + lane, e1, 64));
            if (ADD) f4_add(acc, o);
            if (lane == 0) v.stage[v.cap_a + k] = acc;
        }
    }
}

// every read of x is done: the step's results become the rows.  STORE: void operator()(int local, float4) - what else
// happens to a finished row (the forward chain writes it to memory).
template <int THREADS, class STORE>
__device__ inline void graph_commit(const GraphView &v, const int (&loc)[kRpt], const float4 (&res)[kRpt],
                                    const STORE &store) {
    const int tid = threadIdx.x;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRpt; ++r) {
        if (loc[r] < 0) continue;
        v.x[loc[r]] = res[r];
        store(loc[r], res[r]);
    }
    for (int k = tid; k < v.n_a; k += THREADS) {
        const int local = v.list_a[k];
        const float4 t = v.stage[k];
        v.x[local] = t;
        store(local, t);
    }
    for (int k = tid; k < v.n_b; k += THREADS) {
        const int local = v.list_b[k];
        const float4 t = v.stage[v.cap_a + k];
        v.x[local] = t;
        store(local, t);
    }
    __syncthreads();
}

struct NoOwn {
    __device__ inline float4 operator()(int) const { return f4_zero(); }
};
struct NoStore {
    __device__ inline void operator()(int, const float4 &) const {}
};

struct ChainForwardArgs {
    const float *feat;
    int f;
    const float *fc_w, *fc_b;
    const int *crow, *col;
    const float *val;
    const int *seg_ptr;
    int n, n_graphs, steps;
    ChainLayout layout;
    float *xs;                         // (steps, n, 4)
    const float *adj_hidden, *feat_hidden;
    int h, s;
    float *ah, *z;
    unsigned long long *rng_state, *rng_used;
};

struct RowStore {
    float4 *xg;
    int p0;
    __device__ inline void operator()(int local, const float4 &v) const { xg[p0 + local] = v; }
};

// Workgroups [0, h): the hidden graphs (their body is written for 256 threads: the other wavefronts leave);
// workgroups [h, h + n_graphs): one graph each.
template <int THREADS>
__global__ void __launch_bounds__(THREADS) chain_forward_kernel(const ChainForwardArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < a.h) {
        if (tid >= kHiddenBlock) return;
        hidden_forward_body((int)blockIdx.x, lds, a.adj_hidden, a.feat_hidden, a.h, a.s, 4, a.steps, a.ah, a.z,
                            a.rng_state, a.rng_used);
        return;
    }
    GraphView v = graph_view<THREADS>(lds, a.layout, a.col, a.val);
    float *fcs = v.extra;                                      // [4 * f + 4]
    const int g = (int)blockIdx.x - a.h;
    const int p0 = a.seg_ptr[g], ng = a.seg_ptr[g + 1] - p0;
    if (a.steps > 1) graph_load<THREADS>(v, a.crow, p0, ng);
    v.p0 = p0; v.ng = ng;
    for (int i = tid; i < 4 * a.f + 4; i += THREADS) fcs[i] = i < 4 * a.f ? a.fc_w[i] : a.fc_b[i - 4 * a.f];
    __syncthreads();
    float4 *x0g = reinterpret_cast<float4 *>(a.xs);
    float f0[kRpt];
#pragma unroll
    for (int r = 0; r < kRpt; ++r) {              // the first feature of every row of the thread, requested together
        const int local = tid + r * THREADS;
        f0[r] = a.feat[(size_t)(p0 + (local < ng ? local : ng - 1)) * a.f];
    }
#pragma unroll
    for (int r = 0; r < kRpt; ++r) {
        const int local = tid + r * THREADS;
        if (local >= ng) continue;
        const size_t node = (size_t)(p0 + local);
        const float *frow = a.feat + node * a.f;
        float acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = fmaf(f0[r], fcs[c * a.f], fcs[4 * a.f + c]);
        for (int k = 1; k < a.f; ++k) {
            const float fv = frow[k];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = fmaf(fv, fcs[c * a.f + k], acc[c]);
        }
        const float4 t = make_float4(1.0f / (1.0f + expf(-acc[0])), 1.0f / (1.0f + expf(-acc[1])),
                                     1.0f / (1.0f + expf(-acc[2])), 1.0f / (1.0f + expf(-acc[3])));
        v.x[local] = t;
        x0g[node] = t;
    }
    if (a.steps <= 1) return;
    graph_lists<THREADS>(v);                                   // (its barriers also publish the projected rows)
    int loc[kRpt];
    graph_own_rows<THREADS>(v, loc);
    for (int i = 1; i < a.steps; ++i) {
        float4 res[kRpt];
        graph_step<THREADS, false>(v, loc, res, NoOwn{});
        graph_commit<THREADS>(v, loc, res, RowStore{reinterpret_cast<float4 *>(a.xs) + (size_t)i * a.n, p0});
    }
}

// ---------------------------------------------------------------------------------------------------------
// Backward: dX_{i-1} += A^T dX_i for i = steps-1 .. 1 (autograd of model.py:55) and, on the finished dX_0, the input
// projection's backward (autograd of model.py:43: d_pre = dX_0 x0 (1 - x0), dW = d_pre^T features, db = sum d_pre,
// optionally d features = d_pre W), per graph.  dx holds every step's own share as the fused backward kernel left it;
// nothing of the chain is written back - its only reader is the next step of the chain, in LDS.  Output: one partial
// row [dW (4, f) | db (4)] per graph, summed by the caller in a fixed order (rwk_hidden_backward_reduce and its
// variants, rwk_reduce_rows).  Trailing workgroups run the first level of an independent row reduction (the dZ
// partial sums), as the launches this one replaces did.
// ---------------------------------------------------------------------------------------------------------
struct ChainBackwardArgs {
    const float *dx;                   // (steps, n, 4)
    const float *x0, *feat, *w;
    int f;
    const int *crow, *col;             // CSR of A^T
    const float *val;
    const int *seg_ptr;
    int n, n_graphs, steps;
    ChainLayout layout;
    float *partials;                   // (n_graphs, 4 f + 4)
    float *d_feat;                     // (n, f) or null
    const float *red_in;
    long long red_p, red_v;
    int red_rows_per_block, red_bx;
    float *red_out;
};

struct OwnShare {
    const float4 *g;                   // the step's own rows, already offset to the graph's first node
    __device__ inline float4 operator()(int local) const { return g[local]; }
};

template <int THREADS, int FP>
__global__ void __launch_bounds__(THREADS) chain_backward_kernel(const ChainBackwardArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= a.n_graphs) {
        if (tid >= 256) return;                                 // the reduction's body is written for 256 threads
        const int r = (int)blockIdx.x - a.n_graphs;
        reduce_rows_level1_body(r % a.red_bx, r / a.red_bx, a.red_in, a.red_p, a.red_v, a.red_rows_per_block,
                                a.red_out);
        return;
    }
    constexpr int kWaves = THREADS / RWK_WAVE;
    GraphView v = graph_view<THREADS>(lds, a.layout, a.col, a.val);
    float *red = v.extra;                                      // [4 * FP + 4][kWaves]
    const int g = (int)blockIdx.x;
    const int p0 = a.seg_ptr[g], ng = a.seg_ptr[g + 1] - p0;
    const float4 *dxg = reinterpret_cast<const float4 *>(a.dx);
    graph_load<THREADS>(v, a.crow, p0, ng);
    {
        const float4 *last = dxg + (size_t)(a.steps - 1) * a.n + p0;
        float4 t[kRpt];
#pragma unroll
        for (int r = 0; r < kRpt; ++r) {
            const int local = tid + r * THREADS;
            t[r] = last[local < ng ? local : ng - 1];
        }
#pragma unroll
        for (int r = 0; r < kRpt; ++r) {
            const int local = tid + r * THREADS;
            if (local < ng) v.x[local] = t[r];
        }
    }
    __syncthreads();
    graph_lists<THREADS>(v);
    int loc[kRpt];
    graph_own_rows<THREADS>(v, loc);
    for (int i = a.steps - 1; i >= 1; --i) {
        float4 res[kRpt];
        graph_step<THREADS, true>(v, loc, res, OwnShare{dxg + (size_t)(i - 1) * a.n + p0});
        graph_commit<THREADS>(v, loc, res, NoStore{});
    }
    // v.x: the complete dX_0 of this graph.  Projection backward, a thread's rows in row order.
    float acc_w[4][FP], acc_b[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        acc_b[c] = 0.f;
#pragma unroll
        for (int k = 0; k < FP; ++k) acc_w[c][k] = 0.f;
    }
    const float4 *x0g = reinterpret_cast<const float4 *>(a.x0);
    float4 xr[kRpt];
    float f0[kRpt];
#pragma unroll
    for (int r = 0; r < kRpt; ++r) {              // requested together
        const int local = tid + r * THREADS;
        const size_t node = (size_t)(p0 + (local < ng ? local : ng - 1));
        xr[r] = x0g[node];
        f0[r] = a.feat[node * a.f];
    }
#pragma unroll
    for (int r = 0; r < kRpt; ++r) {
        const int local = tid + r * THREADS;
        if (local >= ng) continue;
        const size_t node = (size_t)(p0 + local);
        const float4 gt = v.x[local];
        const float4 xv = xr[r];
        float fr[FP];
        fr[0] = f0[r];
#pragma unroll
        for (int k = 1; k < FP; ++k) fr[k] = k < a.f ? a.feat[node * a.f + k] : 0.f;
        const float gv[4] = {gt.x, gt.y, gt.z, gt.w};
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        float dpre[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            dpre[c] = gv[c] * xs[c] * (1.0f - xs[c]);
            acc_b[c] += dpre[c];
#pragma unroll
            for (int k = 0; k < FP; ++k) acc_w[c][k] = fmaf(dpre[c], fr[k], acc_w[c][k]);
        }
        if (a.d_feat != nullptr) {
#pragma unroll
            for (int k = 0; k < FP; ++k) {
                if (k < a.f) {
                    float t = 0.f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) t = fmaf(dpre[c], a.w[c * a.f + k], t);
                    a.d_feat[node * a.f + k] = t;
                }
            }
        }
    }
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int k = 0; k < FP; ++k) {
            const float t = rwk_group_sum(acc_w[c][k], 64);
            if (lane == 0) red[(c * FP + k) * kWaves + wave] = t;
        }
        const float t = rwk_group_sum(acc_b[c], 64);
        if (lane == 0) red[(4 * FP + c) * kWaves + wave] = t;
    }
    __syncthreads();
    const int n_out = 4 * a.f + 4;
    float *out = a.partials + (size_t)g * n_out;
    for (int o = tid; o < n_out; o += THREADS) {
        const int r = o < 4 * a.f ? (o / a.f) * FP + o % a.f : 4 * FP + (o - 4 * a.f);
        const float *p = red + r * kWaves;
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < kWaves; ++q) acc += p[q];
        out[o] = acc;
    }
}

inline bool forward_layout(int64_t f, int64_t max_nodes, int64_t s, ChainLayout *l, size_t *bytes) {
    if (f < 1 || f > 16 || s < 1) return false;
    if (!chain_layout(max_nodes, (int)(4 * f + 4), l)) return false;
    const size_t hidden = hidden_forward_lds(s, 4);
    *bytes = l->bytes() > hidden ? l->bytes() : hidden;
    return *bytes <= kLdsBudget;
}

inline bool backward_layout(int64_t f, int64_t max_nodes, ChainLayout *l, size_t *bytes) {
    if (f < 1 || f > 16) return false;
    const int fp = f == 1 ? 1 : f <= 4 ? 4 : 16;
    if (!chain_layout(max_nodes, (4 * fp + 4) * (chain_threads(max_nodes) / RWK_WAVE), l)) return false;
    *bytes = l->bytes();
    return *bytes <= kLdsBudget;
}

template <typename K>
int allow_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
    }
    return RWK_OK;
}

}  // namespace

extern "C" {

int64_t rwk_graph_chain_forward_lds(int64_t f, int64_t max_nodes, int64_t s) {
    ChainLayout l;
    size_t bytes = 0;
    return forward_layout(f, max_nodes, s, &l, &bytes) ? (int64_t)bytes : 0;
}

int64_t rwk_graph_chain_backward_lds(int64_t f, int64_t max_nodes) {
    ChainLayout l;
    size_t bytes = 0;
    return backward_layout(f, max_nodes, &l, &bytes) ? (int64_t)bytes : 0;
}

int rwk_graph_chain_forward(const float *features, const float *weight, const float *bias, int64_t n, int64_t f,
                            const int32_t *crow, const int32_t *col, const float *val, const int32_t *seg_ptr,
                            int64_t n_graphs, int64_t max_nodes, int64_t max_step, float *xs, const float *adj_hidden,
                            const float *features_hidden, int64_t h, int64_t s, float *ah, float *z,
                            uint64_t *rng_state, uint64_t *rng_used, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(features);
    RWK_REQUIRE_PTR(weight);
    RWK_REQUIRE_PTR(bias);
    RWK_REQUIRE_PTR(crow);
    RWK_REQUIRE_PTR(col);
    RWK_REQUIRE_PTR(seg_ptr);
    RWK_REQUIRE_PTR(xs);
    RWK_REQUIRE_PTR(features_hidden);
    RWK_REQUIRE_PTR(ah);
    RWK_REQUIRE_PTR(z);
    if (s > 1) RWK_REQUIRE_PTR(adj_hidden);
    if (n <= 0 || n >= 0x7FFFFFFFll / 4 || n_graphs <= 0 || n_graphs > n || h <= 0 || h > 65535 || max_step <= 0)
        return RWK_E_SIZE;
    if (((uintptr_t)xs & 15u) != 0) return RWK_E_ALIGN;
    ChainLayout l;
    size_t bytes = 0;
    if (!forward_layout(f, max_nodes, s, &l, &bytes)) return RWK_E_UNSUPPORTED;
    ChainForwardArgs a{features, (int)f, weight, bias, crow, col, val, seg_ptr, (int)n, (int)n_graphs, (int)max_step,
                       l, xs, adj_hidden, features_hidden, (int)h, (int)s, ah, z, (unsigned long long *)rng_state,
                       (unsigned long long *)rng_used};
    const dim3 grid((unsigned)(n_graphs + h));
    int rc;
    if (l.threads == 1024) {
        if ((rc = allow_lds(chain_forward_kernel<1024>, bytes)) != RWK_OK) return rc;
        hipLaunchKernelGGL(chain_forward_kernel<1024>, grid, dim3(1024), bytes, (hipStream_t)stream, a);
    } else {
        if ((rc = allow_lds(chain_forward_kernel<256>, bytes)) != RWK_OK) return rc;
        hipLaunchKernelGGL(chain_forward_kernel<256>, grid, dim3(256), bytes, (hipStream_t)stream, a);
    }
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

int rwk_graph_chain_backward(const float *dx, const float *x0, const float *features, const float *weight, int64_t n,
                             int64_t f, const int32_t *crow_t, const int32_t *col_t, const float *val_t,
                             const int32_t *seg_ptr, int64_t n_graphs, int64_t max_nodes, int64_t max_step,
                             float *partials, float *d_features, const float *red_in, int64_t red_p, int64_t red_v,
                             float *red_out32, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(dx);
    RWK_REQUIRE_PTR(x0);
    RWK_REQUIRE_PTR(features);
    RWK_REQUIRE_PTR(weight);
    RWK_REQUIRE_PTR(crow_t);
    RWK_REQUIRE_PTR(col_t);
    RWK_REQUIRE_PTR(seg_ptr);
    RWK_REQUIRE_PTR(partials);
    if (n <= 0 || n >= 0x7FFFFFFFll / 4 || n_graphs <= 0 || n_graphs > n || max_step < 2 || max_step > 64)
        return RWK_E_SIZE;
    if ((((uintptr_t)dx | (uintptr_t)x0) & 15u) != 0) return RWK_E_ALIGN;
    ChainLayout l;
    size_t bytes = 0;
    if (!backward_layout(f, max_nodes, &l, &bytes)) return RWK_E_UNSUPPORTED;
    int red_bx = 1, rows_per_block = 0, red_blocks = 0;
    if (red_in != nullptr) {
        RWK_REQUIRE_PTR(red_out32);
        if (red_p <= kReduceMaxFinalRows || red_v <= 0 || red_v >= 0x7FFFFFFFll * 64) return RWK_E_SIZE;
        red_bx = (int)rwk_ceil_div(red_v, 64);
        rows_per_block = (int)rwk_ceil_div(red_p, kReduceMaxFinalRows);
        red_blocks = red_bx * kReduceMaxFinalRows;
    }
    ChainBackwardArgs a{dx, x0, features, weight, (int)f, crow_t, col_t, val_t, seg_ptr, (int)n, (int)n_graphs,
                        (int)max_step, l, partials, d_features, red_in, (long long)red_p, (long long)red_v,
                        rows_per_block, red_bx, red_out32};
    const dim3 grid((unsigned)(n_graphs + red_blocks));
    const bool big = l.threads == 1024;
    int rc;
#define RWK_CB(T, FP)                                                                                       \
    do {                                                                                                    \
        if ((rc = allow_lds(chain_backward_kernel<T, FP>, bytes)) != RWK_OK) return rc;                     \
        hipLaunchKernelGGL((chain_backward_kernel<T, FP>), grid, dim3(T), bytes, (hipStream_t)stream, a);   \
    } while (0)
    if (f == 1) { if (big) RWK_CB(1024, 1); else RWK_CB(256, 1); }
    else if (f <= 4) { if (big) RWK_CB(1024, 4); else RWK_CB(256, 4); }
    else { if (big) RWK_CB(1024, 16); else RWK_CB(256, 16); }
#undef RWK_CB
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

}  // extern "C"
