// Micro-benchmark: what one dependent kernel boundary costs inside a replayed HIP graph on gfx950.
// A chain of `chain` kernels on one stream is captured once and replayed; every kernel reads what the previous
// one wrote (so the chain is a true dependence), with `blocks` workgroups of 256 threads and `bytes` written per
// kernel.  Reported: time per kernel of the replayed chain (events around `reps` replays), for
//   * a trivial kernel (one workgroup, 4 bytes): the floor a launch boundary costs by itself,
//   * 128 / 1200 workgroups and 0.1-16 MB written: how the floor moves with the grid and with the dirty data the
//     end-of-kernel write-back has to push out.
// The same chains are also timed as plain stream launches (no graph).
// Build: hipcc --offload-arch=gfx950 -O3 scratch/launch_floor/launch_floor.hip -o scratch/launch_floor/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            std::fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            std::exit(1);                                                                 \
        }                                                                                 \
    } while (0)

// every thread copies `per_thread` floats (stride = grid size, coalesced) from src to dst and adds one
__global__ void __launch_bounds__(256) hop_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                  int per_thread) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = 0; k < per_thread; ++k, i += stride) dst[i] = src[i] + 1.f;
}

__global__ void __launch_bounds__(64) trivial_kernel(const float *__restrict__ src, float *__restrict__ dst) {
    if (threadIdx.x == 0) dst[0] = src[0] + 1.f;
}

struct Case {
    const char *name;
    int blocks, per_thread;     // blocks == 0: the trivial kernel
};

static void enqueue_chain(const Case &c, int chain, float *a, float *b, hipStream_t s) {
    for (int k = 0; k < chain; ++k) {
        const float *src = (k & 1) ? b : a;
        float *dst = (k & 1) ? a : b;
        if (c.blocks == 0) hipLaunchKernelGGL(trivial_kernel, dim3(1), dim3(64), 0, s, src, dst);
        else hipLaunchKernelGGL(hop_kernel, dim3(c.blocks), dim3(256), 0, s, src, dst, c.per_thread);
    }
}

int main() {
    const int chain = 10, reps = 200;
    const Case cases[] = {
        {"trivial (1 workgroup, 4 B)", 0, 0},
        {"128 workgroups, 128 KB", 128, 1},
        {"128 workgroups, 1 MB", 128, 8},
        {"1200 workgroups, 1.2 MB", 1200, 1},
        {"1200 workgroups, 4.9 MB", 1200, 4},
        {"1200 workgroups, 19.7 MB", 1200, 16},
    };
    float *a, *b;
    const size_t n = (size_t)1200 * 256 * 16;
    CHECK(hipMalloc(&a, n * sizeof(float)));
    CHECK(hipMalloc(&b, n * sizeof(float)));
    CHECK(hipMemset(a, 0, n * sizeof(float)));
    CHECK(hipMemset(b, 0, n * sizeof(float)));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::printf("%-30s %14s %14s\n", "kernel", "graph us/kernel", "stream us/kernel");
    for (const Case &c : cases) {
        hipGraph_t graph;
        hipGraphExec_t exec;
        CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        enqueue_chain(c, chain, a, b, s);
        CHECK(hipStreamEndCapture(s, &graph));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int r = 0; r < 20; ++r) CHECK(hipGraphLaunch(exec, s));
        CHECK(hipStreamSynchronize(s));
        CHECK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) CHECK(hipGraphLaunch(exec, s));
        CHECK(hipEventRecord(e1, s));
        CHECK(hipStreamSynchronize(s));
        float ms_graph = 0.f;
        CHECK(hipEventElapsedTime(&ms_graph, e0, e1));
        for (int r = 0; r < 20; ++r) enqueue_chain(c, chain, a, b, s);
        CHECK(hipStreamSynchronize(s));
        CHECK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) enqueue_chain(c, chain, a, b, s);
        CHECK(hipEventRecord(e1, s));
        CHECK(hipStreamSynchronize(s));
        float ms_stream = 0.f;
        CHECK(hipEventElapsedTime(&ms_stream, e0, e1));
        std::printf("%-30s %14.2f %14.2f\n", c.name, ms_graph * 1e3 / (reps * chain), ms_stream * 1e3 / (reps * chain));
        CHECK(hipGraphExecDestroy(exec));
        CHECK(hipGraphDestroy(graph));
    }
    // one long graph: 100 trivial kernels per replay (what the boundary costs when the replay's own start is amortised)
    {
        const Case c = cases[0];
        hipGraph_t graph;
        hipGraphExec_t exec;
        CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        enqueue_chain(c, 100, a, b, s);
        CHECK(hipStreamEndCapture(s, &graph));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int r = 0; r < 5; ++r) CHECK(hipGraphLaunch(exec, s));
        CHECK(hipStreamSynchronize(s));
        CHECK(hipEventRecord(e0, s));
        for (int r = 0; r < 50; ++r) CHECK(hipGraphLaunch(exec, s));
        CHECK(hipEventRecord(e1, s));
        CHECK(hipStreamSynchronize(s));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("%-30s %14.2f\n", "trivial, 100 per replay", ms * 1e3 / (50 * 100));
    }
    return 0;
}
