// Input projection  X0 = sigmoid(features W^T + b)  (rwgnn2/model.py:43) for narrow inputs, its
// backward, and the generic deterministic row reduction the backward kernels share.
//
// With F = 1 (node degree as the only feature, utils.py:24) the projection is N x 1 times 1 x D:
// a GEMM library spends ~0.7 ms on the N-long weight-gradient reduction, the kernels here stream it
// at HBM rate: forward reads N*F and writes N*D floats; backward reads 2*N*D + N*F floats and
// writes per-workgroup partial sums of (dW | db) which rwk_reduce_rows adds in a fixed order.
#include <mutex>

#include "rwk_project_body.h"
#include "rwk_reduce_body.h"

namespace {

constexpr int kBlock = kProjectBlock;

// one thread per (node, output channel); 32-bit index arithmetic (n * d < 2^31 is checked by the launcher)
template <bool SIGMOID>
__global__ void __launch_bounds__(kBlock)
project_forward_kernel(const float *__restrict__ feat, const float *__restrict__ w, const float *__restrict__ b,
                       int n, int f, int d, float *__restrict__ x0) {
    const unsigned t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= (unsigned)n * (unsigned)d) return;
    const unsigned node = t / (unsigned)d;
    const int c = (int)(t - node * (unsigned)d);
    float acc = b[c];
    const float *fr = feat + (size_t)node * f;
    const float *wr = w + (size_t)c * f;
    for (int k = 0; k < f; ++k) acc = fmaf(fr[k], wr[k], acc);
    x0[t] = SIGMOID ? 1.0f / (1.0f + expf(-acc)) : acc;
}

// hidden_dim up to 64 with the weights in registers: thread (slot, c) keeps row c of W and walks `rows` nodes.
// The workgroup's feature rows are one contiguous range of memory: they are copied to LDS with coalesced loads
// (rows padded to FP floats) and every thread reads its node's row from there.  Reading them from memory per
// thread costs one address per lane for a value that dpad lanes share: at hidden_dim 64 the address unit, not
// HBM, then bounds the kernel; the general kernel above is worse still (W[c * f + k] per output: 64 cache lines
// per wavefront load at f = 16).
constexpr int kRowTileFloats = 4096;                   // LDS tile: (nodes of a workgroup) x FP floats, 16 KB
constexpr int kRowNodesMax = 32;                      // amortises the strided load of the weight row

inline int project_rows_per_slot(int slots, int fp) {
    int rows = kRowTileFloats / (slots * fp);
    return rows > kRowNodesMax ? kRowNodesMax : (rows < 1 ? 1 : rows);
}

template <int FP, bool SIGMOID>
__global__ void __launch_bounds__(kBlock)
project_forward_rows_kernel(const float *__restrict__ feat, const float *__restrict__ w, const float *__restrict__ b,
                            int n, int f, int d, int dpad, int rows, float *__restrict__ x0) {
    __shared__ __attribute__((aligned(16))) float tile[kRowTileFloats];
    const int c = threadIdx.x % dpad, slot = threadIdx.x / dpad, slots = kBlock / dpad;
    const bool own = c < d;
    const int cc = own ? c : d - 1;
    const long long first = (long long)blockIdx.x * slots * rows;          // first node of the workgroup
    const long long left = (long long)n - first;
    const int nodes = (int)(left < (long long)slots * rows ? left : (long long)slots * rows);
    const float inv_f = 1.0f / (float)f;
    for (int i = threadIdx.x; i < nodes * f; i += kBlock) {
        const int r = (int)(((float)i + 0.5f) * inv_f);                     // i / f, exact for i < 2^17
        tile[r * FP + (i - r * f)] = feat[first * f + i];
    }
    float wr[FP];
#pragma unroll
    for (int k = 0; k < FP; ++k) wr[k] = k < f ? w[(size_t)cc * f + k] : 0.f;
    const float bias = b[cc];
    __syncthreads();
    for (int u = 0; u < rows; ++u) {
        const int local = slot * rows + u;
        if (local >= nodes) break;
        float fr[FP];                                     // whole padded row: the padding is never used below
#pragma unroll
        for (int k = 0; k < FP; ++k) fr[k] = tile[local * FP + k];
        float acc = bias;
#pragma unroll
        for (int k = 0; k < FP; ++k) acc = k < f ? fmaf(fr[k], wr[k], acc) : acc;
        if (own) x0[(first + local) * d + c] = SIGMOID ? 1.0f / (1.0f + expf(-acc)) : acc;
    }
}

template <int FP, bool SIGMOID>
__global__ void __launch_bounds__(kBlock)
project_forward4_kernel(const float *__restrict__ feat, const float *__restrict__ w, const float *__restrict__ b,
                        int n, int f, float *__restrict__ x0) {
    project_forward4_body<FP, SIGMOID>(blockIdx.x, feat, w, b, n, f, x0);
}

template <int FP, bool SIGMOID>
__global__ void __launch_bounds__(kBlock)
project_backward_kernel(const float *__restrict__ dx0, const float *__restrict__ x0,
                        const float *__restrict__ feat, const float *__restrict__ w, long long n, int f, int d,
                        int dpad, float *__restrict__ partials, float *__restrict__ d_feat) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [FP + 1][kBlock]
    project_backward_body<FP, SIGMOID>(blockIdx.x, gridDim.x, lds, dx0, x0, feat, w, n, f, d, dpad, partials, d_feat);
}

__global__ void __launch_bounds__(kBlock)
reduce_rows_level1_kernel(const float *__restrict__ in, long long p, long long v, int rows_per_block,
                          float *__restrict__ out) {
    reduce_rows_level1_body(blockIdx.x, blockIdx.y, in, p, v, rows_per_block, out);
}

// Level 2: out[col] = sum of p <= 32 rows in index order.
__global__ void __launch_bounds__(kBlock)
reduce_rows_final_kernel(const float *__restrict__ in, int p, long long v, float *__restrict__ out) {
    const long long col = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (col >= v) return;
    float acc = 0.f;
    int r = 0;
    for (; r + 7 < p; r += 8) {                         // eight rows in flight, added in row order
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = in[(long long)(r + u) * v + col];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += a[u];
    }
    for (; r < p; ++r) acc += in[(long long)r * v + col];
    out[col] = acc;
}

constexpr int kMaxFinalRows = kReduceMaxFinalRows;

// Few columns (v <= 64) and up to a few thousand rows in ONE launch: a workgroup of 1024 threads gives every
// column 1024/vp lanes (vp = v rounded up to a power of two); lane l adds rows l, l + lanes, ... in order (four
// loads in flight), then the lanes of a column are added in a fixed halving tree through LDS.
constexpr int kSmallBlock = 1024;
constexpr int kSmallMaxV = 64;

__global__ void __launch_bounds__(kSmallBlock)
reduce_rows_small_kernel(const float *__restrict__ in, int p, int v, int vp, float *__restrict__ out) {
    __shared__ float red[kSmallBlock];
    const int tid = threadIdx.x;
    const int col = tid % vp, lane = tid / vp, lanes = kSmallBlock / vp;
    const int cc = col < v ? col : v - 1;
    float acc = 0.f;
    for (int r0 = lane; r0 < p; r0 += 4 * lanes) {
        float a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + u * lanes;
            a[u] = in[(size_t)(r < p ? r : p - 1) * v + cc];
            if (r >= p) a[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += a[u];
    }
    red[tid] = acc;
    __syncthreads();
    for (int off = lanes / 2; off >= 1; off >>= 1) {
        if (lane < off) red[tid] += red[tid + off * vp];
        __syncthreads();
    }
    if (lane == 0 && col < v) out[col] = red[col];
}

inline int pow2_ceil(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

}  // namespace

extern "C" {

int64_t rwk_reduce_rows_level1(int64_t p) {
    if (p <= kMaxFinalRows) return p;
    return kMaxFinalRows;
}

int64_t rwk_reduce_rows_workspace(int64_t p, int64_t v) {
    return p > kMaxFinalRows ? kMaxFinalRows * v : 0;
}

int rwk_reduce_rows_to32(const float *in, int64_t p, int64_t v, float *out32, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(in);
    RWK_REQUIRE_PTR(out32);
    if (p <= kMaxFinalRows || v <= 0 || v >= 0x7FFFFFFFll * 64) return RWK_E_SIZE;
    const int rows_per_block = (int)rwk_ceil_div(p, kMaxFinalRows);
    // with rows_per_block = ceil(p/32) some trailing row blocks may be empty: they write zeros
    hipLaunchKernelGGL(reduce_rows_level1_kernel, dim3((unsigned)rwk_ceil_div(v, 64), (unsigned)kMaxFinalRows),
                       dim3(kBlock), 0, (hipStream_t)stream, in, (long long)p, (long long)v, rows_per_block, out32);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

int rwk_reduce_rows(const float *in, int64_t p, int64_t v, float *out, float *workspace, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(in);
    RWK_REQUIRE_PTR(out);
    if (p <= 0 || v <= 0 || v >= 0x7FFFFFFFll * 64) return RWK_E_SIZE;
    if (v <= kSmallMaxV && p > kMaxFinalRows && p <= 1 << 20) {
        hipLaunchKernelGGL(reduce_rows_small_kernel, dim3(1), dim3(kSmallBlock), 0, (hipStream_t)stream, in, (int)p,
                           (int)v, pow2_ceil((int)v), out);
        RWK_LAUNCH_CHECK();
        return RWK_OK;
    }
    const float *src = in;
    int rows = (int)p;
    if (p > kMaxFinalRows) {
        RWK_REQUIRE_PTR(workspace);
        const int rc = rwk_reduce_rows_to32(in, p, v, workspace, stream);
        if (rc != RWK_OK) return rc;
        src = workspace;
        rows = kMaxFinalRows;
    }
    hipLaunchKernelGGL(reduce_rows_final_kernel, dim3((unsigned)rwk_ceil_div(v, kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, src, rows, (long long)v, out);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------
// rwk_reduce_rows for callers that cannot allocate: the 32 x v level-1 sums live in a file-scope device
// array instead of a caller-owned workspace.  Every GPU the library runs on holds its own copy of the array
// (2 MiB) at its own address until the process ends; the address is looked up once per device, with that
// device current, and only that address (never the array's name) is handed to kernels.
// ---------------------------------------------------------------------------------------
constexpr long long kStaticMaxV = RWK_REDUCE_ROWS_STATIC_MAX_V;
static __device__ __align__(256) float scratch[kMaxFinalRows * kStaticMaxV];

namespace {
constexpr int kMaxDevices = 64;
std::mutex scratch_mutex;
float *scratch_ptr[kMaxDevices] = {};        // device address of `scratch` on each GPU, nullptr = not looked up yet

// The array's copy on the device the launches will run on: the device that owns `stream` (the current device for
// the null stream).  That device must be the current one - the address is looked up with it current, and so are the
// launches issued - otherwise RWK_E_DEVICE: no kernel ever gets another GPU's address.
int scratch_for_stream_device(hipStream_t stream, float **out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (stream != nullptr) {
        hipDevice_t owner;
        e = hipStreamGetDevice(stream, &owner);
        if (e != hipSuccess) return (int)e;
        if ((int)owner != dev) return RWK_E_DEVICE;
    }
    if (dev < 0 || dev >= kMaxDevices) return RWK_E_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(scratch_mutex);
    if (scratch_ptr[dev] == nullptr) {
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return (int)e;
        if (p == nullptr) return RWK_E_NULL;
        scratch_ptr[dev] = static_cast<float *>(p);
    }
    *out = scratch_ptr[dev];
    return RWK_OK;
}
}  // namespace

extern "C" int rwk_reduce_rows_static(const float *in, int64_t p, int64_t v, float *out, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(in);
    RWK_REQUIRE_PTR(out);
    if (p <= 0 || v <= 0) return RWK_E_SIZE;
    float *ws = nullptr;
    if (rwk_reduce_rows_workspace(p, v) > 0) {
        // every element of the 32 x v level-1 block is written by the first launch (row blocks without rows write
        // zeros), so nothing of an earlier call's contents is read and nothing needs clearing
        if (v > kStaticMaxV) return RWK_E_STATIC_SIZE;
        const int rc = scratch_for_stream_device((hipStream_t)stream, &ws);
        if (rc != RWK_OK) return rc;
    }
    return rwk_reduce_rows(in, p, v, out, ws, stream);
}

template <bool SIGMOID>
static int project_forward_impl(const float *features, const float *weight, const float *bias, int64_t n, int64_t f,
                                int64_t d, float *x0, rwk_stream_t stream) {
    RWK_REQUIRE_PTR(features);
    RWK_REQUIRE_PTR(weight);
    RWK_REQUIRE_PTR(bias);
    RWK_REQUIRE_PTR(x0);
    if (n <= 0 || f <= 0 || d <= 0 || n * d >= 0x7FFFFFFFll) return RWK_E_SIZE;
    hipStream_t s = (hipStream_t)stream;
    if (d == 4 && f <= 16 && ((uintptr_t)x0 & 15u) == 0) {
        const dim3 grid((unsigned)rwk_ceil_div(n, kBlock));
#define RWK_PF(FP)                                                                                                \
    hipLaunchKernelGGL((project_forward4_kernel<FP, SIGMOID>), grid, dim3(kBlock), 0, s, features, weight, bias, (int)n, \
                       (int)f, x0)
        if (f == 1) RWK_PF(1);
        else if (f == 2) RWK_PF(2);
        else if (f <= 4) RWK_PF(4);
        else if (f <= 8) RWK_PF(8);
        else RWK_PF(16);
#undef RWK_PF
    } else if (d <= 64 && f <= 16) {
        const int dpad = pow2_ceil((int)d), slots = kBlock / dpad;
        const int fp = f == 1 ? 1 : f == 2 ? 2 : f <= 4 ? 4 : f <= 8 ? 8 : 16;
        const int rows = project_rows_per_slot(slots, fp);
        const dim3 grid((unsigned)rwk_ceil_div(n, (int64_t)slots * rows));
#define RWK_PR(FP)                                                                                                \
    hipLaunchKernelGGL((project_forward_rows_kernel<FP, SIGMOID>), grid, dim3(kBlock), 0, s, features, weight, bias, \
                       (int)n, (int)f, (int)d, dpad, rows, x0)
        if (f == 1) RWK_PR(1);
        else if (f == 2) RWK_PR(2);
        else if (f <= 4) RWK_PR(4);
        else if (f <= 8) RWK_PR(8);
        else RWK_PR(16);
#undef RWK_PR
    } else {
        hipLaunchKernelGGL((project_forward_kernel<SIGMOID>), dim3((unsigned)rwk_ceil_div(n * d, kBlock)), dim3(kBlock),
                           0, s, features, weight, bias, (int)n, (int)f, (int)d, x0);
    }
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

extern "C" {

int rwk_project_forward(const float *features, const float *weight, const float *bias, int64_t n, int64_t f,
                        int64_t d, float *x0, rwk_stream_t stream) {
    return project_forward_impl<true>(features, weight, bias, n, f, d, x0, stream);
}

int rwk_linear_forward(const float *x, const float *weight, const float *bias, int64_t n, int64_t f, int64_t d,
                       float *y, rwk_stream_t stream) {
    return project_forward_impl<false>(x, weight, bias, n, f, d, y, stream);
}

int64_t rwk_project_backward_blocks(int64_t n, int64_t d) {
    if (n <= 0 || d <= 0 || d > 64) return 0;
    const int64_t slots = kBlock / pow2_ceil((int)d);
    int64_t blocks = rwk_ceil_div(n, slots * 8);       // about 8 nodes per thread
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    return blocks;
}

}  // extern "C"

template <bool SIGMOID>
static int project_backward_impl(const float *dx0, const float *x0, const float *features, const float *weight,
                                 int64_t n, int64_t f, int64_t d, float *partials, float *d_features,
                                 rwk_stream_t stream) {
    RWK_REQUIRE_PTR(dx0);
    RWK_REQUIRE_PTR(x0);
    RWK_REQUIRE_PTR(features);
    RWK_REQUIRE_PTR(weight);
    RWK_REQUIRE_PTR(partials);
    if (n <= 0 || f <= 0 || d <= 0) return RWK_E_SIZE;
    if (f > 16 || d > 64) return RWK_E_UNSUPPORTED;
    const int dpad = pow2_ceil((int)d);
    const unsigned blocks = (unsigned)rwk_project_backward_blocks(n, d);
    hipStream_t s = (hipStream_t)stream;
#define RWK_PB(FP)                                                                                              \
    hipLaunchKernelGGL((project_backward_kernel<FP, SIGMOID>), dim3(blocks), dim3(kBlock),                      \
                       (FP + 1) * kBlock * sizeof(float), s, dx0, x0, features, weight, (long long)n, (int)f,   \
                       (int)d, dpad, partials, d_features)
    if (f == 1) RWK_PB(1);
    else if (f == 2) RWK_PB(2);
    else if (f <= 4) RWK_PB(4);
    else if (f <= 8) RWK_PB(8);
    else RWK_PB(16);
#undef RWK_PB
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

extern "C" {

int rwk_project_backward(const float *dx0, const float *x0, const float *features, const float *weight,
                         int64_t n, int64_t f, int64_t d, float *partials, float *d_features,
                         rwk_stream_t stream) {
    return project_backward_impl<true>(dx0, x0, features, weight, n, f, d, partials, d_features, stream);
}

int rwk_linear_backward(const float *d_y, const float *x, const float *weight, int64_t n, int64_t f, int64_t d,
                        float *partials, float *d_x, rwk_stream_t stream) {
    // the body reads its second argument only for the activation's derivative: any valid (n, d) buffer will do
    return project_backward_impl<false>(d_y, d_y, x, weight, n, f, d, partials, d_x, stream);
}

}  // extern "C"
