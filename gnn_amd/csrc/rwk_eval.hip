// RW_NN.forward in evaluation mode (rwgnn2/model.py:31-72 under model.eval(): no dropout, batch-norm on its running
// statistics) as ONE launch, one workgroup per graph.
//
// A batch is block-diagonal: every neighbour of a node lies in the node's own graph.  Without dropout the layer
// factorises through the graph's Gram matrices (rwk_gram.hip), and an evaluation-mode head treats every row of K on
// its own.  So the workgroup of graph g can do everything for g without leaving the chip: project its nodes
// (X0 = sigmoid(features W^T + b), model.py:43) into LDS, propagate them there (X_i = A X_{i-1}, model.py:55, the
// graph's CSR rows gathered from LDS), build the hidden graphs' Z_i = Ah^i Z0 (model.py:39-42,56; a few thousand
// multiply-adds, recomputed per workgroup), form C[g,i] = sum_n X0[n] (x) X_i[n] and K[g, i*H+h] = sum_s Z0[h,s]^T
// C[g,i] Z_i[h,s] (model.py:45-67), and run the head on that one row (model.py:68-72).  The reference's validation and
// test loops (rwgnn2/main.py:110-117,148-151) call exactly this every epoch; as separate launches (projection +
// hidden graphs, SpMM, Gram, head) it took 28 us on 1024 graphs of 300 nodes, of which 20 were launch boundaries and
// dependent round trips of four tiny kernels.
//
// hidden_dim <= 4; graphs of at most max_nodes nodes, with everything fitting the LDS budget (rwk_eval_forward_lds).
// Same operations in the same order as the stand-alone kernels where that is cheap (projection, hidden-graph chain,
// neighbour sums in stored order); all sums in fixed orders: bitwise reproducible.
#include "rwk_common.h"
#include "rwk_hidden_body.h"

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / RWK_WAVE;
constexpr int kSlots = kBlock / 4;
constexpr size_t kLdsBudget = 150 * 1024;

struct EvalArgs {
    const float *feat;        // (n, f) raw features, or (n, d) pre-activations when fc_w is null
    int f;
    const float *fc_w, *fc_b;
    const float *adj_hidden, *feat_hidden;
    const int *crow, *col;
    const float *val;         // null: every stored value is 1
    const int *seg_ptr;
    const float *counts_f32;
    int n, d, h, s, normalize, max_nodes;
    const float *bn_w, *bn_b, *rmean, *rvar;
    float eps;
    const float *w1, *b1, *w2, *b2;
    int p, nc;
    float *k_out, *out;
    // Not null: the graphs are read from the device-resident data set of rwk_batch_build (feat = its feature rows,
    // col = its columns, unweighted), graph g of the launch being the entry g of the batch's (4, b+1) offset table;
    // no batch is assembled.
    const long long *table, *ds_row_ptr;
    int n_graphs;
};

__device__ inline float pick4(const float4 &v, int a) { return a == 0 ? v.x : a == 1 ? v.y : a == 2 ? v.z : v.w; }

__device__ inline float slot_sum16(float v) {
    v += __shfl_xor(v, 4, RWK_WAVE);
    v += __shfl_xor(v, 8, RWK_WAVE);
    v += __shfl_xor(v, 16, RWK_WAVE);
    v += __shfl_xor(v, 32, RWK_WAVE);
    return v;
}

// GATHER: the instance that reads the graphs from the data set (EvalArgs::table); the other carries none of that code
template <int LT, bool GATHER>
__global__ void __launch_bounds__(kBlock) eval_forward_kernel(const EvalArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int h = a.h, s = a.s, d = a.d;
    float *zs = lds;                                   // [LT][h][s][4]
    float *xs = zs + LT * h * s * 4;                   // [LT][max_nodes][4]  (16-byte rows: before the odd-sized ah)
    float *ah = xs + (size_t)LT * a.max_nodes * 4;     // [h][s][s]
    float *part = ah + h * s * s;                      // [kWaves][LT][16]
    float *cm = part + kWaves * LT * 16;               // [LT][16]
    float *krow = cm + LT * 16;                        // [LT * h]
    float *h1 = krow + LT * h;                         // [p]
    float *o = h1 + a.p;                               // [nc]
    float *bn = o + a.nc;                              // [4][LT * h]: running mean, running var, gamma, beta
    float *w1s = bn + 4 * LT * h;                      // [p][LT * h]
    float *b1s = w1s + (size_t)a.p * LT * h;           // [p]
    float *w2s = b1s + a.p;                            // [nc][p]
    float *b2s = w2s + (size_t)a.nc * a.p;             // [nc]
    float *fcs = b2s + a.nc;                           // [d * f + d]: projection weights and bias
    const int tid = threadIdx.x;
    const int g = blockIdx.x;
    // p0: the graph's first node in the arrays the rows are read from (the batch's, or the data set's)
    long long p0;
    int ng;
    constexpr bool gather = GATHER;
    if (gather) {
        ng = (int)(a.table[g + 1] - a.table[g]);
        p0 = a.table[(long long)a.n_graphs + 1 + g];
    } else {
        p0 = a.seg_ptr[g];
        ng = a.seg_ptr[g + 1] - (int)p0;
    }
    const long long *__restrict__ rp64 = gather ? a.ds_row_ptr + p0 : nullptr;
    const int *__restrict__ rp32 = gather ? nullptr : a.crow + p0;

    // ---- hidden graphs: Ah = sym(triu(relu(adj_hidden))), Z_i = Ah Z_{i-1}
    const int n_pairs = s * (s - 1) / 2;
    for (int i = tid; i < h * s * s; i += kBlock) ah[i] = 0.f;
    for (int i = tid; i < h * s * 4; i += kBlock) {
        const int k = i & 3, r = i >> 2;               // r = hh * s + ss
        zs[i] = k < d ? a.feat_hidden[(size_t)r * d + k] : 0.f;
    }
    // row pointers of this thread's first two nodes: requested now, used after the projection (the thread that
    // projects node p0 + tid (+ 256) is the one that later gathers its neighbours)
    int e_beg[2] = {0, 0}, e_end[2] = {0, 0};
    float f_first[2] = {0.f, 0.f};             // first feature of the thread's first two nodes (projected input only)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int local = tid + r * kBlock;
        if (local < ng) {
            e_beg[r] = gather ? (int)rp64[local] : rp32[local];
            e_end[r] = gather ? (int)rp64[local + 1] : rp32[local + 1];
            if (a.fc_w != nullptr) f_first[r] = a.feat[(size_t)(p0 + local) * a.f];
        }
    }
    if (a.fc_w != nullptr)
        for (int i = tid; i < d * a.f + d; i += kBlock) fcs[i] = i < d * a.f ? a.fc_w[i] : a.fc_b[i - d * a.f];
    if (a.out != nullptr) {
        // the head's parameters travel to LDS now, with every other first load of the workgroup: read where they are
        // used (row j of fc1 by thread j, one element per step of its sum) they cost a trip to the L2 per element
        const int lh0 = LT * h;
        for (int i = tid; i < lh0; i += kBlock) {
            bn[i] = a.rmean[i];
            bn[lh0 + i] = a.rvar[i];
            bn[2 * lh0 + i] = a.bn_w[i];
            bn[3 * lh0 + i] = a.bn_b[i];
        }
        for (int i = tid; i < a.p * lh0; i += kBlock) w1s[i] = a.w1[i];
        for (int i = tid; i < a.p; i += kBlock) b1s[i] = a.b1[i];
        for (int i = tid; i < a.nc * a.p; i += kBlock) w2s[i] = a.w2[i];
        for (int i = tid; i < a.nc; i += kBlock) b2s[i] = a.b2[i];
    }
    __syncthreads();
    for (int i = tid; i < h * n_pairs; i += kBlock) {
        const int hh = i / n_pairs, k = i - hh * n_pairs;
        int r, c;
        triu_pair(k, s, &r, &c);
        const float w = fmaxf(a.adj_hidden[i], 0.f);
        ah[(hh * s + r) * s + c] = w;
        ah[(hh * s + c) * s + r] = w;
    }
    // ---- X0 of the graph's nodes
    float4 *x0s = reinterpret_cast<float4 *>(xs);
    for (int local = tid, r = 0; local < ng; local += kBlock, ++r) {
        const size_t node = (size_t)(p0 + local);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.fc_w != nullptr) {
            // weights from LDS (one broadcast read per term); the sum runs over the input features in order, as in
            // the stand-alone projection kernel
            const float *frow = a.feat + node * a.f;
            float acc[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = c < d ? fcs[d * a.f + c] : 0.f;
            for (int k = 0; k < a.f; ++k) {
                const float fv = (k == 0 && r < 2) ? (r == 0 ? f_first[0] : f_first[1]) : frow[k];
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < d) acc[c] = fmaf(fv, fcs[c * a.f + k], acc[c]);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < d) v[c] = 1.0f / (1.0f + expf(-acc[c]));
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < d) v[c] = 1.0f / (1.0f + expf(-a.feat[node * d + c]));
        }
        x0s[local] = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    // ---- the two chains, step by step: Z_i = Ah Z_{i-1} and X_i = A X_{i-1} (rows of the graph, gathered from LDS)
#pragma unroll
    for (int i = 1; i < LT; ++i) {
        const float *zp = zs + (size_t)(i - 1) * h * s * 4;
        float *zn = zs + (size_t)i * h * s * 4;
        for (int e = tid; e < h * s * 4; e += kBlock) {
            const int k = e & 3, r = e >> 2;           // r = hh * s + row
            const int hh = r / s;
            const float *arow = ah + (size_t)r * s;
            float acc = 0.f;
            for (int c = 0; c < s; ++c) acc = fmaf(arow[c], zp[((hh * s + c) << 2) + k], acc);
            zn[e] = acc;
        }
        const float4 *xp = reinterpret_cast<const float4 *>(xs) + (size_t)(i - 1) * a.max_nodes;
        float4 *xn = reinterpret_cast<float4 *>(xs) + (size_t)i * a.max_nodes;
        for (int local = tid, r = 0; local < ng; local += kBlock, ++r) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const int e0 = r == 0 ? e_beg[0] : r == 1 ? e_beg[1] : (gather ? (int)rp64[local] : rp32[local]);
            const int e1 = r == 0 ? e_end[0] : r == 1 ? e_end[1] : (gather ? (int)rp64[local + 1] : rp32[local + 1]);
            // four stored entries per trip, their columns (and values) requested together; entries past the row's end
            // enter with weight zero (fmaf(0, r, acc) is acc), so the sum runs in stored order whatever the length
            for (int e = e0; e < e1; e += 4) {
                int c[4];
                float w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int at = e + u < e1 ? e + u : e1 - 1;
                    c[u] = a.col[at] - (int)p0;
                    w[u] = a.val ? a.val[at] : 1.f;
                    if (e + u >= e1) w[u] = 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 r = xp[c[u]];
                    acc.x = fmaf(w[u], r.x, acc.x);
                    acc.y = fmaf(w[u], r.y, acc.y);
                    acc.z = fmaf(w[u], r.z, acc.z);
                    acc.w = fmaf(w[u], r.w, acc.w);
                }
            }
            xn[local] = acc;
        }
        __syncthreads();
    }
    // ---- Gram matrices of the graph: thread (slot q, row a)
    {
        const int q = tid >> 2, ra = tid & 3;
        const int wave = tid / RWK_WAVE, lane = tid % RWK_WAVE;
        float acc[LT][4];
#pragma unroll
        for (int i = 0; i < LT; ++i)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[i][b] = 0.f;
        for (int local = q; local < ng; local += kSlots) {
            const float4 x0 = x0s[local];
            const float x0a = pick4(x0, ra);
#pragma unroll
            for (int i = 0; i < LT; ++i) {
                const float4 r = i == 0 ? x0 : (reinterpret_cast<const float4 *>(xs) + (size_t)i * a.max_nodes)[local];
                acc[i][0] = fmaf(x0a, r.x, acc[i][0]);
                acc[i][1] = fmaf(x0a, r.y, acc[i][1]);
                acc[i][2] = fmaf(x0a, r.z, acc[i][2]);
                acc[i][3] = fmaf(x0a, r.w, acc[i][3]);
            }
        }
#pragma unroll
        for (int i = 0; i < LT; ++i)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[i][b] = slot_sum16(acc[i][b]);
        if (lane < 4) {
#pragma unroll
            for (int i = 0; i < LT; ++i)
#pragma unroll
                for (int b = 0; b < 4; ++b) part[(wave * LT + i) * 16 + ra * 4 + b] = acc[i][b];
        }
    }
    __syncthreads();
    if (tid < LT * 16) {
        float v = part[tid];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) v += part[w * LT * 16 + tid];
        cm[tid] = v;
    }
    __syncthreads();
    // ---- K row of the graph
    const int lh = LT * h;
    const float cnt = a.normalize ? (gather ? (float)ng : a.counts_f32[g]) : 1.f;
    for (int idx = tid; idx < lh; idx += kBlock) {
        const int i = idx / h, hh = idx - i * h;
        const float *c = cm + i * 16;
        const float4 *z0r = reinterpret_cast<const float4 *>(zs) + (size_t)hh * s;
        const float4 *zir = reinterpret_cast<const float4 *>(zs) + (size_t)(i * h + hh) * s;
        float sum = 0.f;
        for (int ss = 0; ss < s; ++ss) {
            const float4 z0 = z0r[ss], zi = zir[ss];
            float r = 0.f;
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) {
                float t = c[aa * 4 + 0] * zi.x;
                t = fmaf(c[aa * 4 + 1], zi.y, t);
                t = fmaf(c[aa * 4 + 2], zi.z, t);
                t = fmaf(c[aa * 4 + 3], zi.w, t);
                r = fmaf(pick4(z0, aa), t, r);
            }
            sum += r;
        }
        if (a.normalize) sum /= cnt;
        if (a.k_out) a.k_out[(size_t)g * lh + idx] = sum;
        if (a.out) krow[idx] = (sum - bn[idx]) * (1.f / sqrtf(bn[lh + idx] + a.eps)) * bn[2 * lh + idx] + bn[3 * lh + idx];
    }
    if (a.out == nullptr) return;
    __syncthreads();
    // ---- head on the row: fc1, ReLU, fc2, log-softmax (model.py:68-72 in evaluation mode)
    // fc1: eight lanes per output, each adding every eighth term, then a fixed xor butterfly over the eight (one lane
    // walking the whole row waited for an LDS read per term)
    for (int j0 = 0; j0 < a.p; j0 += kBlock / 8) {
        const int j = j0 + (tid >> 3), part8 = tid & 7;
        float acc = 0.f;
        if (j < a.p) {
            const float *w = w1s + (size_t)j * lh;
            for (int c = part8; c < lh; c += 8) acc = fmaf(krow[c], w[c], acc);
        }
        acc += __shfl_xor(acc, 1, RWK_WAVE);
        acc += __shfl_xor(acc, 2, RWK_WAVE);
        acc += __shfl_xor(acc, 4, RWK_WAVE);
        if (j < a.p && part8 == 0) h1[j] = fmaxf(acc + b1s[j], 0.f);
    }
    __syncthreads();
    for (int qq = tid; qq < a.nc; qq += kBlock) {
        float acc = b2s[qq];
        const float *w = w2s + (size_t)qq * a.p;
        for (int j = 0; j < a.p; ++j) acc = fmaf(h1[j], w[j], acc);
        o[qq] = acc;
    }
    __syncthreads();
    for (int qq = tid; qq < a.nc; qq += kBlock) {
        float mx = o[0];
        for (int t = 1; t < a.nc; ++t) mx = fmaxf(mx, o[t]);
        float se = 0.f;
        for (int t = 0; t < a.nc; ++t) se += expf(o[t] - mx);
        a.out[(size_t)g * a.nc + qq] = o[qq] - (mx + logf(se));
    }
}

// ---------------------------------------------------------------------------------------------------------
// Training forward, first launch: the node features of EVERY step (X_0 = sigmoid(features W^T + b), X_i = A X_{i-1},
// model.py:43,55) and the hidden graphs (model.py:39-42,56) in one launch, for block-diagonal batches.  The
// workgroup of graph g projects its nodes into LDS and gathers every propagation step from there (a node's
// neighbours are nodes of its own graph), writing each X_i row to memory once for the fused kernels and the backward
// pass; the first h workgroups run the hidden-graph kernel's body.  As separate launches (projection +
// hidden graphs, then one SpMM per step) this was 5.8 + 5.6 us at two steps on 1024 graphs of 300 nodes; the SpMM's
// launch did nothing but wait for rows the projection had just written.  Same operations in the same order as those
// launches for rows below the tile SpMM's hub length (an fmaf chain in stored order): same bits.
// ---------------------------------------------------------------------------------------------------------
constexpr int kXChainColsPerNode = 8;     // LDS reserved for a graph's columns: this many 16-bit entries per node
constexpr int kXChainColsPre = 6;         // column loads per thread in flight under the projection (1536 entries)

struct XChainArgs {
    const float *feat;
    int f;
    const float *fc_w, *fc_b;
    const int *crow, *col;
    const float *val;
    const int *seg_ptr;
    int n, n_graphs, max_nodes, steps;
    float *xs;                         // (steps, n, 4)
    const float *adj_hidden, *feat_hidden;
    int h, s;
    float *ah, *z;
    unsigned long long *rng_state, *rng_used;
    int cols_cap;                      // 16-bit column entries the launch's LDS has room for (0: none)
    // has_gather: the batch is not assembled yet.  The workgroup of batch graph g reads the graph's rows from the
    // device-resident data set (feat, crow = its 64-bit row pointers, col = its columns; unweighted) and leaves the
    // batch's structure behind for the launches that follow - what rwk_batch_build would have written beforehand.
    rwk_x_chain_gather_t gather;
    int has_gather;
};

// GATHER: the instance that reads the data set (XChainArgs::gather); the other one carries none of that code
template <bool GATHER>
__global__ void __launch_bounds__(kBlock) x_chain_hidden_kernel(const XChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // the hidden graphs' workgroups come first: dispatched behind a thousand graph workgroups they would start when
    // those end, and the launch would take both durations
    if ((int)blockIdx.x < a.h) {
        hidden_forward_body((int)blockIdx.x, lds, a.adj_hidden, a.feat_hidden, a.h, a.s, 4, a.steps, a.ah, a.z,
                            a.rng_state, a.rng_used);
        return;
    }
    float4 *xa = reinterpret_cast<float4 *>(lds);              // [max_nodes]: the previous step's rows
    float4 *xb = xa + a.max_nodes;                             // [max_nodes]: this step's
    float *fcs = reinterpret_cast<float *>(xb + a.max_nodes);  // [4 * f + 4]
    // [kXChainColsPerNode * max_nodes] columns of the graph's rows, as distances from its first node
    unsigned short *cols = reinterpret_cast<unsigned short *>(fcs + 4 * a.f + 4);
    const int tid = threadIdx.x;
    // XCD k takes the k-th contiguous eighth of the graphs (workgroups are dealt round-robin to the 8 XCDs): the X
    // rows written here are read by the fused kernels, which give XCD k the k-th eighth of the nodes
    int g = (int)blockIdx.x - a.h;
    {
        const int per_xcd = a.n_graphs / 8;
        if (g < per_xcd * 8) g = (((int)blockIdx.x) % 8) * per_xcd + g / 8;
    }
    const rwk_x_chain_gather_t *gt = &a.gather;
    constexpr bool gather = GATHER;
    // gather: p0 = the graph's first node in the batch, q0 = in the data set; its entries start at eb in the batch
    int p0, ng;
    long long q0 = 0;
    int eb = 0;
    if (gather) {
        const long long *__restrict__ tb = reinterpret_cast<const long long *>(gt->table);
        const long long b1 = (long long)a.n_graphs + 1;
        const long long s0 = tb[g], s1 = tb[g + 1];
        q0 = tb[b1 + g];
        eb = (int)tb[2 * b1 + g];
        p0 = (int)s0;
        ng = (int)(s1 - s0);
        if (tid == 0) {
            gt->seg_ptr[g] = p0;
            gt->counts_f32[g] = (float)ng;
            gt->y[g] = gt->ds_y[tb[3 * b1 + g]];
            if (g == a.n_graphs - 1) gt->seg_ptr[a.n_graphs] = p0 + ng;
        }
    } else {
        p0 = a.seg_ptr[g];
        ng = a.seg_ptr[g + 1] - p0;
    }
    const float *__restrict__ featp = gather ? gt->ds_feat + (size_t)q0 * a.f : a.feat + (size_t)p0 * a.f;
    const long long *__restrict__ rp64 = gather ? reinterpret_cast<const long long *>(gt->ds_row_ptr) + q0 : nullptr;
    int e_beg[2] = {0, 0}, e_end[2] = {0, 0};
    long long r_beg[2] = {0, 0}, r_end[2] = {0, 0}, r_first = 0, r_last = 0;
    float f_first[2] = {0.f, 0.f};             // the first feature of the thread's first two nodes, requested with the
#pragma unroll                                 // row pointers (in the loop below each cost a trip of its own)
    for (int r = 0; r < 2; ++r) {
        const int local = tid + r * kBlock;
        if (local < ng) f_first[r] = featp[(size_t)local * a.f];
        if (local < ng && a.steps > 1) {
            if (gather) {
                r_beg[r] = rp64[local];
                r_end[r] = rp64[local + 1];
            } else {
                e_beg[r] = a.crow[p0 + local];
                e_end[r] = a.crow[p0 + local + 1];
            }
        }
    }
    // The stored entries of the graph's rows are one contiguous range: with its two ends (requested here, with the row
    // pointers) the whole range of columns is fetched coalesced while the projection below runs, and every
    // propagation step reads its columns from LDS.  Read row by row from memory they were one to three more
    // dependent trips per step (four entries a trip, and a wavefront waits for its longest row).  Unweighted graphs
    // below 65536 nodes whose entries fit the reserved space; others read memory as before.
    int ge0 = 0, ge1 = 0;
    if (gather) {
        if (a.steps > 1 && ng > 0) {
            r_first = rp64[0];
            r_last = rp64[ng];
        }
    } else if (a.steps > 1 && a.val == nullptr) {
        ge0 = a.crow[p0];
        ge1 = a.crow[p0 + ng];
    }
    for (int i = tid; i < 4 * a.f + 4; i += kBlock) fcs[i] = i < 4 * a.f ? a.fc_w[i] : a.fc_b[i - 4 * a.f];
    __syncthreads();
    // entries are numbered as in the batch from here on; colp[e] is entry e's column, a node number that starts at
    // `sub` for the graph's first node (the batch's numbering, or the data set's)
    const int *__restrict__ colp = a.col;
    long long sub = p0;
    if (gather) {
        ge0 = eb;
        ge1 = eb + (int)(r_last - r_first);
        colp = gt->ds_col + (r_first - eb);
        sub = q0;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            e_beg[r] = eb + (int)(r_beg[r] - r_first);
            e_end[r] = eb + (int)(r_end[r] - r_first);
        }
    }
    const int n_e = ge1 - ge0;
    const bool cols_in_lds = n_e > 0 && n_e <= a.cols_cap && ng <= 65536;
    int pre_c[kXChainColsPre];
    if (cols_in_lds) {
#pragma unroll
        for (int u = 0; u < kXChainColsPre; ++u) {
            const int i = tid + u * kBlock;
            pre_c[u] = colp[ge0 + (i < n_e ? i : n_e - 1)];
        }
    }
    float4 *x0g = reinterpret_cast<float4 *>(a.xs);
    for (int local = tid, r = 0; local < ng; local += kBlock, ++r) {
        const size_t node = (size_t)(p0 + local);
        const float *frow = featp + (size_t)local * a.f;
        float acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = fcs[4 * a.f + c];
        for (int k = 0; k < a.f; ++k) {
            const float fv = (k == 0 && r < 2) ? (r == 0 ? f_first[0] : f_first[1]) : frow[k];
            if (gather) gt->feat[node * a.f + k] = fv;
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = fmaf(fv, fcs[c * a.f + k], acc[c]);
        }
        const float4 v = make_float4(1.0f / (1.0f + expf(-acc[0])), 1.0f / (1.0f + expf(-acc[1])),
                                     1.0f / (1.0f + expf(-acc[2])), 1.0f / (1.0f + expf(-acc[3])));
        xa[local] = v;
        x0g[node] = v;
        if (gather) gt->gid[node] = g;
    }
    if (cols_in_lds) {
#pragma unroll
        for (int u = 0; u < kXChainColsPre; ++u) {
            const int i = tid + u * kBlock;
            if (i < n_e) cols[i] = (unsigned short)(pre_c[u] - sub);
        }
        for (int i = tid + kXChainColsPre * kBlock; i < n_e; i += kBlock) cols[i] = (unsigned short)(colp[ge0 + i] - sub);
    }
    __syncthreads();
    for (int i = 1; i < a.steps; ++i) {
        float4 *xg = reinterpret_cast<float4 *>(a.xs) + (size_t)i * a.n;
        // gather: the first step also writes the rows' pointers and compact columns (rwk_csr_compact's form: the
        // distance from the first row of the row's 256-row block)
        const bool emit = gather && i == 1;
        for (int local = tid, r = 0; local < ng; local += kBlock, ++r) {
            const int node = p0 + local;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int e0, e1;
            if (r < 2) {
                e0 = r == 0 ? e_beg[0] : e_beg[1];
                e1 = r == 0 ? e_end[0] : e_end[1];
            } else if (gather) {
                e0 = eb + (int)(rp64[local] - r_first);
                e1 = eb + (int)(rp64[local + 1] - r_first);
            } else {
                e0 = a.crow[node];
                e1 = a.crow[node + 1];
            }
            const int cbase = p0 - (node & ~255);          // compact column = (column within the graph) + cbase
            if (emit) {
                gt->crow[node] = e0;
                if (node == a.n - 1) gt->crow[a.n] = e1;
            }
            if (cols_in_lds) {
                // acc + x is what fmaf(1.0f, x, acc) returns: the same sum in the same (stored) order
                for (int e = e0 - ge0; e < e1 - ge0; ++e) {
                    const int cl = cols[e];
                    const float4 rr = xa[cl];
                    if (emit) gt->col16[ge0 + e] = (short)(cl + cbase);
                    acc.x += rr.x; acc.y += rr.y; acc.z += rr.z; acc.w += rr.w;
                }
                xb[local] = acc;
                xg[node] = acc;
                continue;
            }
            for (int e = e0; e < e1; e += 4) {
                int c[4];
                float w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int at = e + u < e1 ? e + u : e1 - 1;
                    c[u] = (int)(colp[at] - sub);
                    w[u] = a.val ? a.val[at] : 1.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (e + u < e1) {                          // uniform enough: rows of 3.7 entries on average
                        const float4 rr = xa[c[u]];
                        if (emit) gt->col16[e + u] = (short)(c[u] + cbase);
                        acc.x = fmaf(w[u], rr.x, acc.x);
                        acc.y = fmaf(w[u], rr.y, acc.y);
                        acc.z = fmaf(w[u], rr.z, acc.z);
                        acc.w = fmaf(w[u], rr.w, acc.w);
                    }
                }
            }
            xb[local] = acc;
            xg[node] = acc;
        }
        __syncthreads();
        float4 *t = xa; xa = xb; xb = t;
    }
}

inline size_t x_chain_cols_bytes(int64_t max_nodes) {
    return (size_t)kXChainColsPerNode * max_nodes * sizeof(unsigned short);
}

inline size_t eval_lds(int64_t h, int64_t s, int64_t l, int64_t max_nodes, int64_t p, int64_t nc) {
    const int64_t lh = l * h;
    return (size_t)(l * h * s * 4 + h * s * s + l * max_nodes * 4 + kWaves * l * 16 + l * 16 + lh + p + nc +
                    (p > 0 ? 4 * lh + p * lh + p + nc * p + nc : 0) + 4 * 16 + 4) *
           sizeof(float);
}

int launch_eval(const EvalArgs &a, int64_t bytes, int64_t n_graphs, int64_t max_step, rwk_stream_t stream) {
    const dim3 grid((unsigned)n_graphs);
#define RWK_EVAL(LT)                                                                                          \
    do {                                                                                                      \
        auto kern = a.table != nullptr ? eval_forward_kernel<LT, true> : eval_forward_kernel<LT, false>;      \
        if (bytes > 64 * 1024) {                                                                              \
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                               (int)bytes);                                                   \
            if (e != hipSuccess) return (int)e;                                                               \
        }                                                                                                     \
        hipLaunchKernelGGL(kern, grid, dim3(kBlock), (size_t)bytes, (hipStream_t)stream, a);                  \
    } while (0)
    switch (max_step) {
        case 1: RWK_EVAL(1); break;
        case 2: RWK_EVAL(2); break;
        case 3: RWK_EVAL(3); break;
        default: RWK_EVAL(4); break;
    }
#undef RWK_EVAL
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

int launch_x_chain(XChainArgs a, int64_t bytes, int64_t f, int64_t max_nodes, int64_t n_graphs, int64_t h,
                   rwk_stream_t stream) {
    auto kern = a.has_gather ? x_chain_hidden_kernel<true> : x_chain_hidden_kernel<false>;
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
    }
    const size_t rows = (size_t)(2 * max_nodes * 4 + 4 * f + 4) * sizeof(float);
    if (rows + x_chain_cols_bytes(max_nodes) <= kLdsBudget) a.cols_cap = (int)(kXChainColsPerNode * max_nodes);
    hipLaunchKernelGGL(kern, dim3((unsigned)(n_graphs + h)), dim3(kBlock), (size_t)bytes, (hipStream_t)stream, a);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

}  // namespace

extern "C" {

int64_t rwk_x_chain_lds(int64_t f, int64_t max_nodes, int64_t s) {
    if (f < 1 || f > 16 || max_nodes < 1 || s < 1) return 0;
    const size_t rows = (size_t)(2 * max_nodes * 4 + 4 * f + 4) * sizeof(float);
    const size_t with_cols = rows + x_chain_cols_bytes(max_nodes);
    const size_t graph = with_cols <= kLdsBudget ? with_cols : rows;   // the columns' space only when it fits
    const size_t hidden = hidden_forward_lds(s, 4);
    const size_t bytes = graph > hidden ? graph : hidden;
    return bytes <= kLdsBudget ? (int64_t)bytes : 0;
}

int rwk_x_chain_hidden_forward(const float *features, const float *weight, const float *bias, int64_t n, int64_t f,
                               const int32_t *crow, const int32_t *col, const float *val, const int32_t *seg_ptr,
                               int64_t n_graphs, int64_t max_nodes, int64_t max_step, float *xs,
                               const float *adj_hidden, const float *features_hidden, int64_t h, int64_t s, float *ah,
                               float *z, uint64_t *rng_state, uint64_t *rng_used, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(features);
    RWK_REQUIRE_PTR(weight);
    RWK_REQUIRE_PTR(bias);
    RWK_REQUIRE_PTR(crow);
    RWK_REQUIRE_PTR(seg_ptr);
    RWK_REQUIRE_PTR(xs);
    RWK_REQUIRE_PTR(features_hidden);
    RWK_REQUIRE_PTR(ah);
    RWK_REQUIRE_PTR(z);
    if (s > 1) RWK_REQUIRE_PTR(adj_hidden);
    if (n <= 0 || n >= 0x7FFFFFFFll / 4 || n_graphs <= 0 || n_graphs > n || h <= 0 || h > 65535 || max_step <= 0)
        return RWK_E_SIZE;
    if (((uintptr_t)xs & 15u) != 0) return RWK_E_ALIGN;
    const int64_t bytes = rwk_x_chain_lds(f, max_nodes, s);
    if (bytes == 0) return RWK_E_UNSUPPORTED;
    XChainArgs a{features, (int)f, weight, bias, crow, col, val, seg_ptr, (int)n, (int)n_graphs, (int)max_nodes,
                 (int)max_step, xs, adj_hidden, features_hidden, (int)h, (int)s, ah, z,
                 (unsigned long long *)rng_state, (unsigned long long *)rng_used, 0, {}, 0};
    return launch_x_chain(a, bytes, f, max_nodes, n_graphs, h, stream);
}

int rwk_x_chain_hidden_forward_gather(const rwk_x_chain_gather_t *gather, const float *weight, const float *bias,
                                      int64_t n, int64_t f, int64_t n_graphs, int64_t max_nodes, int64_t max_step,
                                      float *xs, const float *adj_hidden, const float *features_hidden, int64_t h,
                                      int64_t s, float *ah, float *z, uint64_t *rng_state, uint64_t *rng_used,
                                      rwk_stream_t stream) {
    RWK_REQUIRE_PTR(gather);
    RWK_REQUIRE_PTR(weight);
    RWK_REQUIRE_PTR(bias);
    RWK_REQUIRE_PTR(xs);
    RWK_REQUIRE_PTR(features_hidden);
    RWK_REQUIRE_PTR(ah);
    RWK_REQUIRE_PTR(z);
    if (s > 1) RWK_REQUIRE_PTR(adj_hidden);
    if (!gather->table || !gather->ds_row_ptr || !gather->ds_col || !gather->ds_feat || !gather->ds_y || !gather->feat ||
        !gather->y || !gather->crow || !gather->col16 || !gather->gid || !gather->seg_ptr || !gather->counts_f32)
        return RWK_E_NULL;
    if (n <= 0 || n >= 0x7FFFFFFFll / 4 || n_graphs <= 0 || n_graphs > n || h <= 0 || h > 65535 || max_step < 2)
        return RWK_E_SIZE;                        // (the first propagation step writes the rows' structure)
    if (((uintptr_t)xs & 15u) != 0) return RWK_E_ALIGN;
    const int64_t bytes = rwk_x_chain_lds(f, max_nodes, s);
    if (bytes == 0) return RWK_E_UNSUPPORTED;
    XChainArgs a{nullptr, (int)f, weight, bias, nullptr, nullptr, nullptr, nullptr, (int)n, (int)n_graphs,
                 (int)max_nodes, (int)max_step, xs, adj_hidden, features_hidden, (int)h, (int)s, ah, z,
                 (unsigned long long *)rng_state, (unsigned long long *)rng_used, 0, *gather, 1};
    return launch_x_chain(a, bytes, f, max_nodes, n_graphs, h, stream);
}

int64_t rwk_eval_forward_lds(int64_t d, int64_t h, int64_t s, int64_t max_step, int64_t f, int64_t max_nodes,
                             int64_t p, int64_t nc) {
    if (d < 1 || d > 4 || h < 1 || h > 64 || s < 1 || s > 64 || max_step < 1 || max_step > 4) return 0;
    if (f < 1 || f > 16 || max_nodes < 1 || p < 0 || nc < 0 || p > 4096 || nc > 4096) return 0;
    const size_t bytes = eval_lds(h, s, max_step, max_nodes, p, nc);
    return bytes <= kLdsBudget ? (int64_t)bytes : 0;
}

int rwk_eval_forward(const float *feat, int64_t f, const float *fc_w, const float *fc_b, const float *adj_hidden,
                     const float *feat_hidden, const int32_t *crow, const int32_t *col, const float *val,
                     const int32_t *seg_ptr, const float *counts_f32, int64_t n, int64_t n_graphs, int64_t max_nodes,
                     int64_t d, int64_t h, int64_t s, int64_t max_step, int normalize, const float *bn_w,
                     const float *bn_b, const float *running_mean, const float *running_var, float eps,
                     const float *w1, const float *b1, const float *w2, const float *b2, int64_t p, int64_t nc,
                     float *k_out, float *out, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(feat);
    RWK_REQUIRE_PTR(feat_hidden);
    RWK_REQUIRE_PTR(crow);
    RWK_REQUIRE_PTR(seg_ptr);
    if (s > 1) RWK_REQUIRE_PTR(adj_hidden);
    if (fc_w != nullptr) RWK_REQUIRE_PTR(fc_b);
    if (normalize) RWK_REQUIRE_PTR(counts_f32);
    if (k_out == nullptr && out == nullptr) return RWK_E_NULL;
    if (out != nullptr) {
        RWK_REQUIRE_PTR(bn_w);
        RWK_REQUIRE_PTR(bn_b);
        RWK_REQUIRE_PTR(running_mean);
        RWK_REQUIRE_PTR(running_var);
        RWK_REQUIRE_PTR(w1);
        RWK_REQUIRE_PTR(b1);
        RWK_REQUIRE_PTR(w2);
        RWK_REQUIRE_PTR(b2);
        if (p < 1 || nc < 1) return RWK_E_SIZE;
    } else {
        p = 0;
        nc = 0;
    }
    if (n <= 0 || n >= 0x7FFFFFFFll || n_graphs <= 0 || n_graphs > n || max_nodes <= 0) return RWK_E_SIZE;
    const int64_t f_eff = fc_w != nullptr ? f : 1;
    const int64_t bytes = rwk_eval_forward_lds(d, h, s, max_step, f_eff, max_nodes, p, nc);
    if (bytes == 0) return RWK_E_UNSUPPORTED;
    EvalArgs a{feat, (int)f, fc_w, fc_b, adj_hidden, feat_hidden, crow, col, val, seg_ptr, counts_f32, (int)n, (int)d,
               (int)h, (int)s, normalize, (int)max_nodes, bn_w, bn_b, running_mean, running_var, eps, w1, b1, w2, b2,
               (int)p, (int)nc, k_out, out, nullptr, nullptr, (int)n_graphs};
    return launch_eval(a, bytes, n_graphs, max_step, stream);
}

int rwk_eval_forward_gather(const int64_t *table, const int64_t *ds_row_ptr, const int32_t *ds_col, const float *ds_feat,
                            int64_t f, const float *fc_w, const float *fc_b, const float *adj_hidden,
                            const float *feat_hidden, int64_t n_graphs, int64_t max_nodes, int64_t d, int64_t h,
                            int64_t s, int64_t max_step, int normalize, const float *bn_w, const float *bn_b,
                            const float *running_mean, const float *running_var, float eps, const float *w1,
                            const float *b1, const float *w2, const float *b2, int64_t p, int64_t nc, float *k_out,
                            float *out, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(table);
    RWK_REQUIRE_PTR(ds_row_ptr);
    RWK_REQUIRE_PTR(ds_feat);
    RWK_REQUIRE_PTR(feat_hidden);
    RWK_REQUIRE_PTR(fc_w);
    RWK_REQUIRE_PTR(fc_b);
    if (s > 1) RWK_REQUIRE_PTR(adj_hidden);
    if (k_out == nullptr && out == nullptr) return RWK_E_NULL;
    if (out != nullptr) {
        RWK_REQUIRE_PTR(bn_w);
        RWK_REQUIRE_PTR(bn_b);
        RWK_REQUIRE_PTR(running_mean);
        RWK_REQUIRE_PTR(running_var);
        RWK_REQUIRE_PTR(w1);
        RWK_REQUIRE_PTR(b1);
        RWK_REQUIRE_PTR(w2);
        RWK_REQUIRE_PTR(b2);
        if (p < 1 || nc < 1) return RWK_E_SIZE;
    } else {
        p = 0;
        nc = 0;
    }
    if (n_graphs <= 0 || n_graphs >= 0x7FFFFFFFll || max_nodes <= 0) return RWK_E_SIZE;
    const int64_t bytes = rwk_eval_forward_lds(d, h, s, max_step, f, max_nodes, p, nc);
    if (bytes == 0) return RWK_E_UNSUPPORTED;
    EvalArgs a{ds_feat, (int)f, fc_w, fc_b, adj_hidden, feat_hidden, nullptr, ds_col, nullptr, nullptr, nullptr, 0,
               (int)d, (int)h, (int)s, normalize, (int)max_nodes, bn_w, bn_b, running_mean, running_var, eps, w1, b1,
               w2, b2, (int)p, (int)nc, k_out, out, reinterpret_cast<const long long *>(table),
               reinterpret_cast<const long long *>(ds_row_ptr), (int)n_graphs};
    return launch_eval(a, bytes, n_graphs, max_step, stream);
}

}  // extern "C"
