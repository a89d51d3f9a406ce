// The hidden-graph backward (autograd of rwgnn2/model.py:39-42,56) and the final sum of the projection's
// per-workgroup gradient rows in one launch: both are a few microseconds of latency-bound work at the end of the
// backward pass, neither depends on the other, and launched one after the other each pays its own launch, ramp
// and drain.  (The launches that do the first propagation step inside the projection's kernels are in rwk_spmm.hip,
// next to the tile kernel they are built from.)
#include "rwk_hidden_body.h"
#include "rwk_project_body.h"
#include "rwk_reduce_body.h"
#include "rwk_head_partials.h"

namespace {

constexpr int kBlock = 256;
static_assert(kHiddenBlock == kBlock && kProjectBlock == kBlock, "the merged launches use one workgroup size");
constexpr size_t kLdsLimit = 160 * 1024;
// Column sums of a (p, v) matrix with v <= 64 by one workgroup of 256 threads: 256 / vp lanes per column (vp = v
// rounded up to a power of two), lane l adds rows l, l + lanes, ... in order (sixteen loads in flight), then the lanes of
// a column are added in a fixed halving tree.  lds: 256 floats.
struct NoSumTail {
    __device__ inline void column(int /*col*/, float /*sum*/) const {}
};

template <typename TAIL = NoSumTail>
__device__ inline void reduce_rows_small256_body(float *red, const float *__restrict__ in, int p, int v, int vp,
                                                 float *__restrict__ out, const TAIL &tail = TAIL()) {
    const int tid = threadIdx.x;
    const int colv = tid % vp, lane = tid / vp, lanes = kBlock / vp;
    const int cc = colv < v ? colv : v - 1;
    float acc = 0.f;
    constexpr int kFlight = 16;      // loads in flight per round (a lane's rows are still added in row order)
    for (int r0 = lane; r0 < p; r0 += kFlight * lanes) {
        float a[kFlight];
#pragma unroll
        for (int u = 0; u < kFlight; ++u) {
            const int r = r0 + u * lanes;
            a[u] = in[(size_t)(r < p ? r : p - 1) * v + cc];
            if (r >= p) a[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < kFlight; ++u) acc += a[u];
    }
    red[tid] = acc;
    __syncthreads();
    for (int off = lanes / 2; off >= 1; off >>= 1) {
        if (lane < off) red[tid] += red[tid + off * vp];
        __syncthreads();
    }
    if (lane == 0 && colv < v) {
        out[colv] = red[colv];
        tail.column(colv, red[colv]);
    }
}

__global__ void __launch_bounds__(kBlock)
hidden_backward_reduce_kernel(const float *__restrict__ adj_hidden, const float *__restrict__ ah_in,
                              const float *__restrict__ z, const float *__restrict__ dz_partials, int n_partials,
                              int n_hidden, int s, int d, int max_step, float *__restrict__ d_adj_hidden,
                              float *__restrict__ d_feat_hidden, int n_slices, const float *__restrict__ red_in,
                              int red_p, int red_v, int red_vp, float *__restrict__ red_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if ((int)blockIdx.x < n_hidden)
        hidden_backward_body(blockIdx.x, lds, adj_hidden, ah_in, z, dz_partials, n_partials, n_hidden, s, d, max_step,
                             d_adj_hidden, d_feat_hidden, n_slices);
    else
        reduce_rows_small256_body(lds, red_in, red_p, red_v, red_vp, red_out);
}

// ---------------------------------------------------------------------------------------------------
// The same launch with the optimiser's update inside: the last launch of a single-process training step.
// Adam (torch.optim.Adam's arithmetic, rwgnn2/main.py:79,87; the update of rwk_optim.hip's adam_step_kernel) is
// element-wise, so the thread that finishes a gradient element applies it at once: hidden graph h's workgroup
// updates its own rows of adj_hidden and features_hidden, the column-sum workgroup the input projection's weight
// and bias, and further workgroups the remaining parameters (the head's), whose gradients an earlier launch left
// in memory.  One launch less per step.  Every workgroup keeps its own step counter, as the stand-alone kernels do.
// ---------------------------------------------------------------------------------------------------
constexpr int kTailChunks = 16;
struct AdamTailArgs {
    float *param, *exp_avg, *exp_avg_sq, *grad_out;      // flat buffers; grad_out may be null
    const float *lr;
    float beta1, beta2, eps, weight_decay;
    long long *counters;
    long long off_adj, off_feat, off_red;                // flat index of adj_hidden[0], features_hidden[0], column 0
    long long rest_off, n;                               // the chunked remainder covers [rest_off, n)
    // Two slots of {step count t as int64, the two bias corrections of step t as two floats in an int64}; step t uses
    // slot t & 1.  Written for the NEXT step (into the other slot, which nobody reads during this launch) by the
    // column-sum workgroup, which ends well before the hidden-graph workgroups.  May be null.
    long long *corr_slot;
    const float *src[kTailChunks];
    long long off[kTailChunks];                          // flat index each chunk starts at; unused: LLONG_MAX
};

// The head's parameter gradients summed here instead of by the head's own finishing launch (partials == nullptr:
// off, the chunk sources are finished gradient vectors).  The chunk sources then point into row 0 of `partials`
// (n_partials rows of nv floats, the per-workgroup sums of the head's training kernel for d(loss) = 1); an element is
// served by eight lanes, which add its rows in the order of the head's finishing kernel, and takes the factor d_loss.
struct HeadSumArgs {
    const float *partials;
    int n_partials, nv;
    const float *d_loss;
    float *grads_out;                 // nv floats, or null
    const float *loss_partials;       // n_partials per-workgroup sums of -log p(label), or null
    float *loss_out;                  // their sum / n_graphs
    long long n_graphs;
};
constexpr int kHeadSumPerBlock = kBlock / 8;

__device__ inline double pow_int(double b, long long t) {
    double r = 1.0;
    while (t > 0) {
        if (t & 1) r *= b;
        b *= b;
        t >>= 1;
    }
    return r;
}

__device__ inline void adam_element(const AdamTailArgs &a, const float *corr, long long i, float g) {
    if (a.grad_out != nullptr) a.grad_out[i] = g;
    if (a.param == nullptr) return;      // gradient only: the update follows the data-parallel all-reduce
    const float p = a.param[i];
    if (a.weight_decay != 0.f) g = fmaf(a.weight_decay, p, g);
    const float m = fmaf(a.beta1, a.exp_avg[i], (1.f - a.beta1) * g);
    const float v = fmaf(a.beta2, a.exp_avg_sq[i], (1.f - a.beta2) * g * g);
    a.exp_avg[i] = m;
    a.exp_avg_sq[i] = v;
    const float denom = sqrtf(v) / corr[1] + a.eps;
    a.param[i] = p - (a.lr[0] / corr[0]) * (m / denom);
}

struct HiddenAdamTail {
    const AdamTailArgs &a;
    const float *corr;
    __device__ inline void features_hidden(long long index, float g) const { adam_element(a, corr, a.off_feat + index, g); }
    __device__ inline void adj_hidden(long long index, float g) const { adam_element(a, corr, a.off_adj + index, g); }
};
struct SumAdamTail {
    const AdamTailArgs &a;
    const float *corr;
    __device__ inline void column(int col, float sum) const { adam_element(a, corr, a.off_red + col, sum); }
};

__global__ void __launch_bounds__(kBlock)
hidden_backward_reduce_adam_kernel(const float *adj_hidden /* aliases a.param */, const float *__restrict__ ah_in,
                                   const float *__restrict__ z, const float *__restrict__ dz_partials, int n_partials,
                                   int n_hidden, int s, int d, int max_step, float *__restrict__ d_adj_hidden,
                                   float *__restrict__ d_feat_hidden, int n_slices, const float *__restrict__ red_in,
                                   int red_p, int red_v, int red_vp, float *__restrict__ red_out, AdamTailArgs a,
                                   HeadSumArgs hs) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float corr[2];
    // The bias corrections are double-precision power loops and a root: 0.6 us in one thread, on the critical path of
    // every workgroup (the others wait for it at the first barrier).  The previous step's launch left them in
    // corr_slot, computed by a workgroup that had time to spare; a slot of another step count (first step, another
    // launch shape in between) is ignored and the values are computed here, with the same operations.
    if (threadIdx.x == 0 && a.param != nullptr) {
        // both slots are requested together with the counter (a slot address that waited for t would cost the round
        // trip the hand-over is meant to save)
        long long sl[4] = {0, 0, 0, 0};
        if (a.corr_slot != nullptr) {
#pragma unroll
            for (int q = 0; q < 4; ++q) sl[q] = a.corr_slot[q];
        }
        const long long t = a.counters[blockIdx.x] + 1;
        a.counters[blockIdx.x] = t;
        bool have = false;
        if (a.corr_slot != nullptr) {
            const long long t_slot = (t & 1) ? sl[2] : sl[0];
            const long long packed = (t & 1) ? sl[3] : sl[1];
            if (t_slot == t) {
                corr[0] = __int_as_float((int)(packed & 0xFFFFFFFFll));
                corr[1] = __int_as_float((int)(packed >> 32));
                have = true;
            }
        }
        if (!have) {
            corr[0] = (float)(1.0 - pow_int((double)a.beta1, t));
            corr[1] = (float)sqrt(1.0 - pow_int((double)a.beta2, t));
        }
    }
    const int b = blockIdx.x;
    if (b < n_hidden) {
        // the body's own barriers stand between the write of corr and its first use
        hidden_backward_body(b, lds, adj_hidden, ah_in, z, dz_partials, n_partials, n_hidden, s, d, max_step,
                             d_adj_hidden, d_feat_hidden, n_slices, HiddenAdamTail{a, corr});
    } else if (b == n_hidden) {
        reduce_rows_small256_body(lds, red_in, red_p, red_v, red_vp, red_out, SumAdamTail{a, corr});
        // the next step's bias corrections, for every workgroup of the next launch (read there before anything here is
        // overwritten: launches on one stream do not overlap)
        if (threadIdx.x == 0 && a.param != nullptr && a.corr_slot != nullptr) {
            const long long t1 = a.counters[blockIdx.x] + 1;          // this workgroup's counter already holds t
            const float c0 = (float)(1.0 - pow_int((double)a.beta1, t1));
            const float c1 = (float)sqrt(1.0 - pow_int((double)a.beta2, t1));
            a.corr_slot[2 * (t1 & 1) + 1] = (long long)(unsigned int)__float_as_int(c0) | ((long long)__float_as_int(c1) << 32);
            a.corr_slot[2 * (t1 & 1)] = t1;
        }
    } else if (hs.partials == nullptr) {
        const long long i = a.rest_off + (long long)(b - n_hidden - 1) * kBlock + threadIdx.x;
        const long long ic = i < a.n ? i : a.n - 1;
        const float *src = a.src[0];
        long long base = a.off[0];
#pragma unroll
        for (int k = 1; k < kTailChunks; ++k) {           // compare-select chain: a dynamic index would go to scratch
            const bool ge = ic >= a.off[k];
            src = ge ? a.src[k] : src;
            base = ge ? a.off[k] : base;
        }
        const float g = src ? src[ic - base] : 0.f;
        __syncthreads();
        if (i < a.n) adam_element(a, corr, i, g);
    } else {
        const int tid = threadIdx.x, lane8 = tid & 7, grp = tid >> 3;
        const long long i = a.rest_off + (long long)(b - n_hidden - 1) * kHeadSumPerBlock + grp;
        const long long ic = i < a.n ? i : a.n - 1;
        const float *src = a.src[0];
        long long base = a.off[0];
#pragma unroll
        for (int k = 1; k < kTailChunks; ++k) {
            const bool ge = ic >= a.off[k];
            src = ge ? a.src[k] : src;
            base = ge ? a.off[k] : base;
        }
        const int e = src ? (int)(src - hs.partials) + (int)(ic - base) : 0;      // column of the partial rows
        RwkFinPartials pre;
        pre.load(hs.partials, hs.n_partials, hs.nv, e, lane8);
        // mean loss, by the first wavefront of the last workgroup: lane l adds workgroups l, l + 64, ... in order, then
        // a butterfly whose operand pairs are fixed (the head's finishing kernel, operation for operation)
        const bool loss_wave = hs.loss_out != nullptr && blockIdx.x == gridDim.x - 1 && tid < 64;
        float loss_v[4] = {0.f, 0.f, 0.f, 0.f};
        if (loss_wave) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = tid + 64 * u;
                loss_v[u] = hs.loss_partials[q < hs.n_partials ? q : hs.n_partials - 1];
            }
        }
        const float dl = hs.d_loss[0];
        float acc = pre.sum(hs.partials, hs.n_partials, hs.nv, e, lane8);
        acc = rwk_group_sum(acc, 8);
        const float g = src ? acc * dl : 0.f;
        __syncthreads();
        if (lane8 == 0 && i < a.n) {
            if (hs.grads_out != nullptr && src) hs.grads_out[e] = g;
            adam_element(a, corr, i, g);
        }
        if (loss_wave) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (tid + 64 * u >= hs.n_partials) loss_v[u] = 0.f;
            float lacc = ((loss_v[0] + loss_v[1]) + loss_v[2]) + loss_v[3];
            for (int off = 32; off >= 1; off >>= 1) lacc += __shfl_xor(lacc, off, 64);
            if (tid == 0) hs.loss_out[0] = lacc / (float)hs.n_graphs;
        }
    }
}

template <typename K>
int allow_lds(K kern, size_t lds) {
    if (lds > kLdsLimit) return RWK_E_UNSUPPORTED;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    return RWK_OK;
}

}  // namespace

extern "C" {

int rwk_hidden_backward_reduce(const float *adj_hidden, const float *ah, const float *z, const float *dz_partials,
                               int64_t n_partials, int64_t h, int64_t s, int64_t d, int64_t max_step,
                               float *d_adj_hidden, float *d_features_hidden, const float *red_in, int64_t red_p,
                               int64_t red_v, float *red_out, rwk_stream_t stream_) {
    RWK_REQUIRE_PTR(ah);
    RWK_REQUIRE_PTR(z);
    RWK_REQUIRE_PTR(dz_partials);
    RWK_REQUIRE_PTR(d_features_hidden);
    RWK_REQUIRE_PTR(red_in);
    RWK_REQUIRE_PTR(red_out);
    if (h <= 0 || s <= 0 || d <= 0 || max_step <= 0 || n_partials <= 0 || h > 65535) return RWK_E_SIZE;
    if (red_p <= 0 || red_p > (1 << 20) || red_v <= 0) return RWK_E_SIZE;
    if (red_v > 64) return RWK_E_UNSUPPORTED;
    if (s > 1) {
        RWK_REQUIRE_PTR(adj_hidden);
        RWK_REQUIRE_PTR(d_adj_hidden);
    }
    const int64_t n_slices = hidden_backward_slices(max_step * s * d, n_partials);
    size_t lds = hidden_backward_lds(s, d, max_step, n_slices);
    if (lds < kBlock * sizeof(float)) lds = kBlock * sizeof(float);
    const int rc = allow_lds(hidden_backward_reduce_kernel, lds);
    if (rc != RWK_OK) return rc;
    hipLaunchKernelGGL(hidden_backward_reduce_kernel, dim3((unsigned)(h + 1)), dim3(kBlock), lds, (hipStream_t)stream_,
                       adj_hidden, ah, z, dz_partials, (int)n_partials, (int)h, (int)s, (int)d, (int)max_step,
                       d_adj_hidden, d_features_hidden, (int)n_slices, red_in, (int)red_p, (int)red_v,
                       project_pow2_ceil((int)red_v), red_out);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

/* rwk_hidden_backward_reduce with the Adam update of every parameter inside (see hidden_backward_reduce_adam_kernel).
 * The flat buffers hold n parameters: adj_hidden at off_adj, features_hidden at off_feat, the red_v column sums
 * (fc.weight | fc.bias) at off_red, and n_rest further tensors back to back from rest_off to n whose gradients are
 * rest_srcs[k] (rest_sizes[k] floats each, NULL: zeros).  step_counters: rwk_step_tail_n_counters() int64, all at the
 * same step count.  grad_out (may be NULL) receives the flat gradient. */
int64_t rwk_step_tail_n_counters(int64_t h, int64_t n, int64_t rest_off) {
    if (h <= 0 || n <= 0 || rest_off < 0 || rest_off > n) return 0;
    return h + 1 + rwk_ceil_div(n - rest_off, kBlock);
}

int64_t rwk_step_tail_n_counters_head(int64_t h, int64_t n, int64_t rest_off) {
    if (h <= 0 || n <= 0 || rest_off < 0 || rest_off > n) return 0;
    return h + 1 + rwk_ceil_div(n - rest_off, kHeadSumPerBlock);
}

static int step_tail_launch(const float *adj_hidden, const float *ah, const float *z, const float *dz_partials,
                            int64_t n_partials, int64_t h, int64_t s, int64_t d, int64_t max_step,
                            float *d_adj_hidden, float *d_features_hidden, const float *red_in, int64_t red_p,
                            int64_t red_v, float *red_out, float *param, float *exp_avg, float *exp_avg_sq,
                            float *grad_out, int64_t n, int64_t off_adj, int64_t off_feat, int64_t off_red,
                            const float *const *rest_srcs, const int64_t *rest_sizes, int64_t n_rest,
                            int64_t rest_off, const float *lr, float beta1, float beta2, float eps,
                            float weight_decay, int64_t *step_counters, int64_t *corr_slot, const HeadSumArgs &hs,
                            rwk_stream_t stream_) {
    RWK_REQUIRE_PTR(adj_hidden);
    RWK_REQUIRE_PTR(ah);
    RWK_REQUIRE_PTR(z);
    RWK_REQUIRE_PTR(dz_partials);
    RWK_REQUIRE_PTR(d_adj_hidden);
    RWK_REQUIRE_PTR(d_features_hidden);
    RWK_REQUIRE_PTR(red_in);
    RWK_REQUIRE_PTR(red_out);
    if (param != nullptr) {
        RWK_REQUIRE_PTR(exp_avg);
        RWK_REQUIRE_PTR(exp_avg_sq);
        RWK_REQUIRE_PTR(lr);
        RWK_REQUIRE_PTR(step_counters);
    } else {
        RWK_REQUIRE_PTR(grad_out);       // gradient-only launch: the flat gradient is all it leaves
    }
    if (h <= 0 || s <= 1 || d <= 0 || max_step <= 1 || n_partials <= 0 || h > 65535) return RWK_E_SIZE;
    if (red_p <= 0 || red_p > (1 << 20) || red_v <= 0) return RWK_E_SIZE;
    if (red_v > 64) return RWK_E_UNSUPPORTED;
    if (n_rest < 0 || n_rest > kTailChunks) return RWK_E_UNSUPPORTED;
    if (n_rest > 0) {
        RWK_REQUIRE_PTR(rest_srcs);
        RWK_REQUIRE_PTR(rest_sizes);
    }
    const int64_t n_adj = h * (s * (s - 1) / 2), n_feat = h * s * d;
    if (n <= 0 || off_adj < 0 || off_feat < 0 || off_red < 0 || rest_off < 0 || rest_off > n ||
        off_adj + n_adj > rest_off || off_feat + n_feat > rest_off || off_red + red_v > rest_off)
        return RWK_E_SIZE;
    // the three in-kernel ranges must tile [0, rest_off) exactly: no parameter left out, none updated twice
    if (n_adj + n_feat + red_v != rest_off) return RWK_E_SIZE;
    if ((off_adj < off_feat + n_feat && off_feat < off_adj + n_adj) || (off_adj < off_red + red_v && off_red < off_adj + n_adj) ||
        (off_feat < off_red + red_v && off_red < off_feat + n_feat))
        return RWK_E_SIZE;
    AdamTailArgs a;
    a.param = param; a.exp_avg = exp_avg; a.exp_avg_sq = exp_avg_sq; a.grad_out = grad_out; a.lr = lr;
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
    a.counters = (long long *)step_counters;
    a.corr_slot = (long long *)corr_slot;
    a.off_adj = off_adj; a.off_feat = off_feat; a.off_red = off_red; a.rest_off = rest_off; a.n = n;
    long long off = rest_off;
    for (int k = 0; k < kTailChunks; ++k) {
        const bool used = k < n_rest;
        if (used && rest_sizes[k] < 0) return RWK_E_SIZE;
        a.src[k] = used ? rest_srcs[k] : nullptr;
        a.off[k] = used ? off : 0x7FFFFFFFFFFFFFFFll;
        if (used) off += rest_sizes[k];
    }
    if (off != n) return RWK_E_SIZE;
    const int64_t n_slices = hidden_backward_slices(max_step * s * d, n_partials);
    size_t lds = hidden_backward_lds(s, d, max_step, n_slices);
    if (lds < kBlock * sizeof(float)) lds = kBlock * sizeof(float);
    const int rc = allow_lds(hidden_backward_reduce_adam_kernel, lds);
    if (rc != RWK_OK) return rc;
    const unsigned grid = (unsigned)(hs.partials != nullptr ? rwk_step_tail_n_counters_head(h, n, rest_off)
                                                            : rwk_step_tail_n_counters(h, n, rest_off));
    hipLaunchKernelGGL(hidden_backward_reduce_adam_kernel, dim3(grid), dim3(kBlock), lds, (hipStream_t)stream_,
                       adj_hidden, ah, z, dz_partials, (int)n_partials, (int)h, (int)s, (int)d, (int)max_step,
                       d_adj_hidden, d_features_hidden, (int)n_slices, red_in, (int)red_p, (int)red_v,
                       project_pow2_ceil((int)red_v), red_out, a, hs);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

int rwk_hidden_backward_reduce_adam(const float *adj_hidden, const float *ah, const float *z, const float *dz_partials,
                                    int64_t n_partials, int64_t h, int64_t s, int64_t d, int64_t max_step,
                                    float *d_adj_hidden, float *d_features_hidden, const float *red_in, int64_t red_p,
                                    int64_t red_v, float *red_out, float *param, float *exp_avg, float *exp_avg_sq,
                                    float *grad_out, int64_t n, int64_t off_adj, int64_t off_feat, int64_t off_red,
                                    const float *const *rest_srcs, const int64_t *rest_sizes, int64_t n_rest,
                                    int64_t rest_off, const float *lr, float beta1, float beta2, float eps,
                                    float weight_decay, int64_t *step_counters, int64_t *corr_slot, rwk_stream_t stream_) {
    return step_tail_launch(adj_hidden, ah, z, dz_partials, n_partials, h, s, d, max_step, d_adj_hidden,
                            d_features_hidden, red_in, red_p, red_v, red_out, param, exp_avg, exp_avg_sq, grad_out, n,
                            off_adj, off_feat, off_red, rest_srcs, rest_sizes, n_rest, rest_off, lr, beta1, beta2, eps,
                            weight_decay, step_counters, corr_slot, HeadSumArgs{}, stream_);
}

int rwk_hidden_backward_reduce_adam_head(const float *adj_hidden, const float *ah, const float *z,
                                         const float *dz_partials, int64_t n_partials, int64_t h, int64_t s, int64_t d,
                                         int64_t max_step, float *d_adj_hidden, float *d_features_hidden,
                                         const float *red_in, int64_t red_p, int64_t red_v, float *red_out,
                                         float *param, float *exp_avg, float *exp_avg_sq, float *grad_out, int64_t n,
                                         int64_t off_adj, int64_t off_feat, int64_t off_red,
                                         const int64_t *rest_columns, const int64_t *rest_sizes, int64_t n_rest,
                                         int64_t rest_off, const float *lr, float beta1, float beta2, float eps,
                                         float weight_decay, int64_t *step_counters, int64_t *corr_slot,
                                         const float *head_partials,
                                         int64_t n_head_partials, int64_t head_nv, const float *d_loss,
                                         float *head_grads_out, const float *loss_partials, float *loss_out,
                                         int64_t n_graphs, rwk_stream_t stream_) {
    RWK_REQUIRE_PTR(head_partials);
    RWK_REQUIRE_PTR(d_loss);
    RWK_REQUIRE_PTR(rest_columns);
    RWK_REQUIRE_PTR(rest_sizes);
    if ((loss_out == nullptr) != (loss_partials == nullptr)) return RWK_E_NULL;
    if (n_head_partials <= 0 || n_head_partials > 256 || head_nv <= 0 || head_nv > 0x7FFFFFFF || n_graphs <= 0)
        return RWK_E_SIZE;
    if (n_rest <= 0 || n_rest > kTailChunks) return RWK_E_UNSUPPORTED;
    const float *srcs[kTailChunks];
    for (int64_t k = 0; k < n_rest; ++k) {
        if (rest_columns[k] < 0 || rest_sizes[k] < 0 || rest_columns[k] + rest_sizes[k] > head_nv) return RWK_E_SIZE;
        srcs[k] = head_partials + rest_columns[k];
    }
    const HeadSumArgs hs{head_partials, (int)n_head_partials, (int)head_nv, d_loss, head_grads_out, loss_partials,
                         loss_out, (long long)n_graphs};
    return step_tail_launch(adj_hidden, ah, z, dz_partials, n_partials, h, s, d, max_step, d_adj_hidden,
                            d_features_hidden, red_in, red_p, red_v, red_out, param, exp_avg, exp_avg_sq, grad_out, n,
                            off_adj, off_feat, off_red, srcs, rest_sizes, n_rest, rest_off, lr, beta1, beta2, eps,
                            weight_decay, step_counters, corr_slot, hs, stream_);
}

}  // extern "C"
