// Fused random-walk kernels, register-resident form for hidden_dim <= 4.
//
// Same thread layout and the same results as the generic kernels in rwk_fused.hip (a stream of
// stream_len sorted positions per group of HP lanes, lane j = hidden graph j), but the hidden-graph
// size S and the step count L are template parameters, so
//   * the lane's L*S rows of Z live in registers for the whole kernel (no LDS reads in the loop),
//   * the gradient of Z accumulates in L*S*4 registers per lane (no LDS read-modify-write per node),
//   * every dropout sample index is a compile-time constant: one Philox4x32-7 call per 15 samples,
//     each sample one byte compare.
// LDS is touched once, at the end of the backward kernel, to add the lanes that own the same hidden
// graph in a fixed order.
#include "rwk_fused_reg.h"
#include "rwk_head_partials.h"

namespace rwk_reg {
namespace {

constexpr int kBlock = 256;
constexpr int DP = 4;
// Waves per SIMD the backward kernel is compiled for: L*S*4 dZ accumulators + L*S*4 rows of Z must fit.
// 4 (<= 128 VGPRs, four 256-thread workgroups per CU) up to L*S = 10, then 2, then 1.
constexpr int bwd_waves_per_simd(int ls) { return ls <= 10 ? 3 : (ls <= 20 ? 2 : 1); }
// instances whose backward kernel can take the head's batch-norm backward (BnFold)
constexpr bool bn_fold_built(int lt, int st) { return lt * st <= 10; }
// The forward kernel keeps L*S*4 rows of Z, the prefetched slot and the running sums: four workgroups per CU up to
// L*S = 10, three up to 20 (168 VGPRs), two up to 32, else one; the keep-S instances carry S times the running
// sums and are compiled one tier lower.
constexpr int fwd_waves_per_simd(int ls, bool keep_s = false) {
    return keep_s ? fwd_waves_per_simd(2 * ls) : (ls <= 10 ? 4 : (ls <= 20 ? 3 : (ls <= 32 ? 2 : 1)));
}

// Diagnostic build only (-DRWK_STAMP): cycle shares of the backward kernel's segments for the wavefronts of one
// workgroup; slot 7 holds the wavefront's lifetime in 100 MHz ticks.
#ifdef RWK_STAMP
__device__ unsigned long long rwk_reg_stamp_buf[4 * 8];
__device__ unsigned long long rwk_reg_stamp_blocks[2 * 2048];   // start / end of each workgroup, 100 MHz ticks
#define RWK_STAMP_DECL()                                                              \
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};                          \
    unsigned long long st_last = __builtin_readcyclecounter();                        \
    const unsigned long long st_rt0 = __builtin_amdgcn_s_memrealtime()
#define RWK_STAMP_AT(k)                                                      \
    do {                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                   \
        const unsigned long long now_ = __builtin_readcyclecounter();        \
        __builtin_amdgcn_sched_barrier(0);                                   \
        st_sum[k] += now_ - st_last;                                         \
        st_last = now_;                                                      \
    } while (0)
#define RWK_STAMP_FLUSH()                                                    \
    do {                                                                     \
        const unsigned long long rt1_ = __builtin_amdgcn_s_memrealtime();    \
        st_sum[7] = rt1_ - st_rt0;                                           \
        if (threadIdx.x == 0 && blockIdx.x < 2048) {                         \
            rwk_reg_stamp_blocks[2 * blockIdx.x] = st_rt0;                   \
            rwk_reg_stamp_blocks[2 * blockIdx.x + 1] = rt1_;                 \
        }                                                                    \
        if (blockIdx.x == 300 && (threadIdx.x & 63) == 0)                    \
            for (int k_ = 0; k_ < 8; ++k_) rwk_reg_stamp_buf[(threadIdx.x >> 6) * 8 + k_] = st_sum[k_]; \
    } while (0)
#else
#define RWK_STAMP_DECL()
#define RWK_STAMP_AT(k)
#define RWK_STAMP_FLUSH()
#endif

struct DropCfg {
    uint32_t t0, t1, seed_lo, key_hi, off_lo;
};

__device__ inline DropCfg load_cfg(const unsigned long long *rng, uint32_t t0, uint32_t t1) {
    const unsigned long long seed = rng[0], off = rng[1];
    return DropCfg{t0, t1, (uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(off >> 32), (uint32_t)off};
}

// The keep decisions of one (node, hidden graph) stream: NC calls of 15 one-byte samples each.
// keep(e) <=> byte < t0 + (tie byte < t1), which equals  byte < t0 || (byte == t0 && tie).
template <int NC>
struct Keep {
    uint32_t w[NC][4];
    uint32_t thr[NC];
    __device__ inline void draw(const DropCfg &c, uint32_t node, uint32_t hq) {
#pragma unroll
        for (int call = 0; call < NC; ++call) {
            const RwkPhilox4 r = rwk_philox_mask(node, hq, (uint32_t)call, c.off_lo, c.seed_lo, c.key_hi);
            w[call][0] = r.x; w[call][1] = r.y; w[call][2] = r.z; w[call][3] = r.w;
            thr[call] = c.t0 + ((r.w >> 24) < c.t1 ? 1u : 0u);
        }
    }
    __device__ inline bool at(int e) const {        // e is a constant after unrolling
        const int call = e / 15, j = e % 15;
        return ((w[call][j >> 2] >> (8 * (j & 3))) & 0xFFu) < thr[call];
    }
};

__device__ inline void load_x(const float *__restrict__ row, int d, float (&out)[DP]) {
    if (d == DP) {
        const float4 v = *reinterpret_cast<const float4 *>(row);
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    } else {
#pragma unroll
        for (int k = 0; k < DP; ++k) {
            const float v = row[k < d ? k : d - 1];     // unconditional load, then select
            out[k] = k < d ? v : 0.f;
        }
    }
}

__device__ inline float dot4(const float (&a)[DP], const float (&b)[DP]) {
    float acc = a[0] * b[0];
    acc = fmaf(a[1], b[1], acc);
    acc = fmaf(a[2], b[2], acc);
    acc = fmaf(a[3], b[3], acc);
    return acc;
}


// A SIMD serves the wavefront of highest priority first and the oldest among equals.  With equal priorities the
// wavefronts of the single round of workgroups finish one after the other, and the last one runs alone at half the
// SIMD's issue rate.  Lowering a wavefront's priority as it advances through its stream (3 for the first half of
// its positions, 2 up to three quarters, 1 up to seven eighths, then 0) makes the wavefronts that are behind go
// first at each of these marks, so they enter the last eighth together and end close to each other.
// Only for a launch whose workgroups are all resident at once (prio != 0): when workgroups run in several rounds,
// a SIMD that serves its oldest wavefront first frees slots for the next round soonest, and the same scheme cost
// 5 % (forward) to 13 % (backward) on 19.7 M nodes.
struct ProgressPriority {
    int next, total, level;          // position of the next drop (never reached when priorities are off)
    __device__ inline ProgressPriority(int prio, int total_) : next(0x7fffffff), total(total_), level(3) {
        if (prio) {
            next = total >> 1;
            __builtin_amdgcn_s_setprio(3);
        }
    }
    // one scalar compare per position; the drops themselves are the rare path
    __device__ inline void at(int done) {
        if (__builtin_expect(done == next, 0)) {
            --level;
            if (level == 2) { __builtin_amdgcn_s_setprio(2); next = total - (total >> 2); }
            else if (level == 1) { __builtin_amdgcn_s_setprio(1); next = total - (total >> 3); }
            else { __builtin_amdgcn_s_setprio(0); next = 0x7fffffff; }
        }
    }
};

template <int LT, int ST>
__device__ inline void load_z(const float *__restrict__ z, int h, int hh, int d, float (&zr)[LT][ST][DP]) {
    // every load is unconditional (a load under a per-element condition makes the compiler branch
    // around it and wait for each one separately: L*S*4 serial round trips to L2)
    if (d == DP) {
#pragma unroll
        for (int i = 0; i < LT; ++i)
#pragma unroll
            for (int ss = 0; ss < ST; ++ss) {
                const float4 v = *reinterpret_cast<const float4 *>(z + (((size_t)i * h + hh) * ST + ss) * DP);
                zr[i][ss][0] = v.x; zr[i][ss][1] = v.y; zr[i][ss][2] = v.z; zr[i][ss][3] = v.w;
            }
    } else {
#pragma unroll
        for (int i = 0; i < LT; ++i)
#pragma unroll
            for (int ss = 0; ss < ST; ++ss)
#pragma unroll
                for (int k = 0; k < DP; ++k) {
                    const float v = z[(((size_t)i * h + hh) * ST + ss) * d + (k < d ? k : d - 1)];
                    zr[i][ss][k] = k < d ? v : 0.f;
                }
    }
}

// One tile = hp consecutive positions of each of the workgroup's kBlock/hp streams = kBlock node slots,
// slot t = (stream t / hp, position t % hp): thread t loads the rows of its own slot (hp lanes read
// hp consecutive 16-byte rows: coalesced), the tile is then read back through LDS by every lane of the
// stream.  node[] holds the node index of the slot, or -1 past the end of the stream.
template <int LT>
struct Tile {
    float4 x[LT][kBlock];
    int gid[kBlock];
    int node[kBlock];
};

// A thread's own slot of a tile, fetched into registers one tile ahead of its use (the loads stay in flight while
// the previous tile is being worked on) and written to LDS afterwards.  Slots past the end of a stream carry zero
// rows (so they contribute nothing), node = -1, and the graph id of the stream's last position (so they never
// trigger a flush): the loops below need no validity branch.  Every load is unconditional, from a clamped
// position: a select on a value makes the wavefront wait for its load where the select stands, so validity is
// applied when the slot is written to LDS.
template <int LT>
struct Slot {
    float x[LT][DP];
    int gid, node;
    bool valid;
};

template <int LT>
__device__ inline void fetch_slot(Slot<LT> &sl, const float *__restrict__ x, const int *__restrict__ gid,
                                  const int *__restrict__ perm, int p, int end, bool has_stream, int d,
                                  size_t step_stride) {
    sl.valid = has_stream && p < end;
    const int pc = has_stream ? (sl.valid ? p : end - 1) : 0;         // always a position of the batch
    sl.gid = gid[pc];
    sl.node = perm ? perm[pc] : pc;
#pragma unroll
    for (int i = 0; i < LT; ++i) load_x(x + i * step_stride + (size_t)sl.node * d, d, sl.x[i]);
}

template <int LT>
__device__ inline void put_slot(Tile<LT> &tile, const Slot<LT> &sl, bool has_stream) {
    const int t = threadIdx.x;
    tile.node[t] = sl.valid ? sl.node : -1;
    tile.gid[t] = has_stream ? sl.gid : -1;
#pragma unroll
    for (int i = 0; i < LT; ++i)
        tile.x[i][t] = sl.valid ? make_float4(sl.x[i][0], sl.x[i][1], sl.x[i][2], sl.x[i][3])
                                : make_float4(0.f, 0.f, 0.f, 0.f);
}

// An even number of steps: the rows of steps 2p and 2p+1 of a slot interleaved, tile.x[2p][t] = (x_2p[0], x_2p+1[0],
// x_2p[1], x_2p+1[1]) and tile.x[2p + 1][t] = (x_2p[2], x_2p+1[2], x_2p[3], x_2p+1[3]), so that a 16-byte LDS read yields
// two (step 2p, step 2p+1) operand pairs of the packed multiply-adds as they stand (row-major rows cost a register move
// per pair).
template <int LT>
__device__ inline void put_slot_pairs(Tile<LT> &tile, const Slot<LT> &sl, bool has_stream) {
    static_assert(LT % 2 == 0, "pairs of steps");
    const int t = threadIdx.x;
    tile.node[t] = sl.valid ? sl.node : -1;
    tile.gid[t] = has_stream ? sl.gid : -1;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int p = 0; p < LT / 2; ++p) {
        tile.x[2 * p][t] = sl.valid ? make_float4(sl.x[2 * p][0], sl.x[2 * p + 1][0], sl.x[2 * p][1], sl.x[2 * p + 1][1]) : zero;
        tile.x[2 * p + 1][t] = sl.valid ? make_float4(sl.x[2 * p][2], sl.x[2 * p + 1][2], sl.x[2 * p][3], sl.x[2 * p + 1][3]) : zero;
    }
}

// The loads a kernel starts with - the first tile's slot and the thread's rows of Z - for hidden_dim 4 without a
// permutation (the caller checks both once), as one stretch of code without a branch: every load is requested before
// the first of them is waited for.  (fetch_slot + load_z, each with its "hidden_dim == 4 ?" and "permutation ?"
// branches, compile to a wait at the end of every branch: graph id, wait, x rows and Z, wait - a trip to memory more
// than needed in front of every workgroup of a launch that is one round of workgroups.  The same loads element by
// element, valid for every hidden_dim, were tried: four times the load instructions for the same bytes cost the
// vector cache more than the trip saved, forward 17.4 -> 21.9 us.)
template <int LT, int ST>
__device__ inline void first_loads4(Slot<LT> &sl, float (&zr)[LT][ST][DP], const float *__restrict__ x,
                                    const float *__restrict__ z, const int *__restrict__ gid, int p, int end,
                                    bool has_stream, size_t step_stride, int h, int hh) {
    sl.valid = has_stream && p < end;
    const int pc = has_stream ? (sl.valid ? p : end - 1) : 0;         // always a position of the batch
    sl.gid = gid[pc];
    sl.node = pc;
#pragma unroll
    for (int i = 0; i < LT; ++i) {
        const float4 v = *reinterpret_cast<const float4 *>(x + i * step_stride + (size_t)pc * DP);
        sl.x[i][0] = v.x; sl.x[i][1] = v.y; sl.x[i][2] = v.z; sl.x[i][3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < LT; ++i)
#pragma unroll
        for (int ss = 0; ss < ST; ++ss) {
            const float4 v = *reinterpret_cast<const float4 *>(z + (((size_t)i * h + hh) * ST + ss) * DP);
            zr[i][ss][0] = v.x; zr[i][ss][1] = v.y; zr[i][ss][2] = v.z; zr[i][ss][3] = v.w;
        }
}

typedef float rwk_f2 __attribute__((ext_vector_type(2)));

// HPT: lanes per stream as a compile-time constant (0 = use the run-time value hp_arg)
// KS (keep S): the sums over the hidden nodes are not taken; slot layout [(i * h + lane) * ST + s], L*H*ST floats
// per slot, i.e. one feature per (step, hidden graph, hidden node) - the output of rwgnn/rwgnn_pyg.py:138-149.
template <int LT, int ST, bool DROP, int HPT, bool KS = false>
__global__ void __launch_bounds__(kBlock, fwd_waves_per_simd(LT * ST, KS))
forward_kernel(const float *__restrict__ x, const float *__restrict__ z, const int *__restrict__ gid,
               const int *__restrict__ perm, int n, int d, int h, int stream_len, int hp_arg, int prio,
               uint32_t t0, uint32_t t1, const unsigned long long *__restrict__ rng, float *__restrict__ k_partials) {
    constexpr int NC = (LT * ST + 14) / 15;
    const int hp = HPT ? HPT : hp_arg;
    __shared__ Tile<LT> tiles[2];                       // tile t in tiles[t & 1]; the other one is being filled
    const int tid = threadIdx.x;
    const int lane_h = tid % hp;
    const int hh = lane_h < h ? lane_h : h - 1;
    const int group0 = tid - lane_h;                    // first slot of this lane's stream in the tile
    const int c = rwk_xcd_block(blockIdx.x, gridDim.x) * (kBlock / hp) + tid / hp;
    const long long start_ll = (long long)c * stream_len;
    const bool has_stream = start_ll < n;
    const int start = has_stream ? (int)start_ll : 0;
    const int end = has_stream ? (int)(start_ll + stream_len < n ? start_ll + stream_len : n) : 0;
    const size_t step_stride = (size_t)n * d;
    Slot<LT> slot;
    float zr[LT][ST][DP];
    if (d == DP && perm == nullptr) {
        first_loads4<LT, ST>(slot, zr, x, z, gid, start + lane_h, end, has_stream, step_stride, h, hh);
    } else {
        fetch_slot<LT>(slot, x, gid, perm, start + lane_h, end, has_stream, d, step_stride);
        load_z<LT, ST>(z, h, hh, d, zr);
    }
    DropCfg cfg;
    if (DROP) cfg = load_cfg(rng, t0, t1);
    const int lh = LT * h;

    constexpr int SK = KS ? ST : 1;                 // accumulators per step
    float acc[LT][SK];
#pragma unroll
    for (int i = 0; i < LT; ++i)
#pragma unroll
        for (int q = 0; q < SK; ++q) acc[i][q] = 0.f;
    int g_cur = -1;
    auto flush = [&](int g) {
        float *dst = k_partials + ((size_t)(c + g) * lh + lane_h) * SK;
#pragma unroll
        for (int i = 0; i < LT; ++i)
#pragma unroll
            for (int q = 0; q < SK; ++q) dst[(size_t)i * h * SK + q] = acc[i][q];
    };

    constexpr bool kPairs = LT % 2 == 0 && !KS;     // (step 2p, step 2p+1) operand pairs, see put_slot_pairs
    constexpr int NP = kPairs ? LT / 2 : 1;
    rwk_f2 zp[NP][kPairs ? ST : 1][DP];
    if constexpr (kPairs) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int ss = 0; ss < ST; ++ss)
#pragma unroll
                for (int k = 0; k < DP; ++k) zp[p][ss][k] = rwk_f2{zr[2 * p][ss][k], zr[2 * p + 1][ss][k]};
        put_slot_pairs<LT>(tiles[0], slot, has_stream);
    } else {
        put_slot<LT>(tiles[0], slot, has_stream);
    }
    __syncthreads();
    ProgressPriority priority(prio, stream_len);
    int buf = 0;
    for (int base = 0; base < stream_len; base += hp, buf ^= 1) {
        const Tile<LT> &tile = tiles[buf];
        const bool more = base + hp < stream_len;
        if (more) fetch_slot<LT>(slot, x, gid, perm, start + base + hp + lane_h, end, has_stream, d, step_stride);
        const int jmax = stream_len - base < hp ? stream_len - base : hp;
        for (int j = 0; j < jmax; ++j) {
            priority.at(base + j);
            const int node = tile.node[group0 + j];
            const int g = tile.gid[group0 + j];
            if (g != g_cur) {
                if (g_cur >= 0 && lane_h < h) flush(g_cur);
#pragma unroll
                for (int i = 0; i < LT; ++i)
#pragma unroll
                    for (int q = 0; q < SK; ++q) acc[i][q] = 0.f;
                g_cur = g;
            }
            Keep<NC> keep;
            if (DROP) keep.draw(cfg, (uint32_t)node, (uint32_t)lane_h);
            if constexpr (kPairs) {
                // the same multiplications and additions in the same order as the general form below, two steps at a
                // time: (zx, t_1) = sum_k (Z_0[ss][k], Z_1[ss][k]) * (x_0[k], x_1[k]), products (zx * zx, zx * t_1), ...
                rwk_f2 xp[NP][DP], node2[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const float4 va = tile.x[2 * p][group0 + j], vb = tile.x[2 * p + 1][group0 + j];
                    xp[p][0] = rwk_f2{va.x, va.y}; xp[p][1] = rwk_f2{va.z, va.w};
                    xp[p][2] = rwk_f2{vb.x, vb.y}; xp[p][3] = rwk_f2{vb.z, vb.w};
                    node2[p] = rwk_f2{0.f, 0.f};
                }
#pragma unroll
                for (int ss = 0; ss < ST; ++ss) {
                    rwk_f2 zt[NP];
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        zt[p] = zp[p][ss][0] * xp[p][0];
                        zt[p] = __builtin_elementwise_fma(zp[p][ss][1], xp[p][1], zt[p]);
                        zt[p] = __builtin_elementwise_fma(zp[p][ss][2], xp[p][2], zt[p]);
                        zt[p] = __builtin_elementwise_fma(zp[p][ss][3], xp[p][3], zt[p]);
                    }
                    const rwk_f2 zx2 = rwk_f2{zt[0].x, zt[0].x};
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        rwk_f2 prod = zt[p] * zx2;
                        if (DROP) {
                            prod.x = keep.at(ss * LT + 2 * p) ? prod.x : 0.f;
                            prod.y = keep.at(ss * LT + 2 * p + 1) ? prod.y : 0.f;
                        }
                        node2[p] += prod;
                    }
                }
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    acc[2 * p][0] += node2[p].x;
                    acc[(2 * p + 1) % LT][0] += node2[p].y;
                }
                continue;
            }
            float xr[LT][DP];
#pragma unroll
            for (int i = 0; i < LT; ++i) {
                const float4 v = tile.x[i][group0 + j];
                xr[i][0] = v.x; xr[i][1] = v.y; xr[i][2] = v.z; xr[i][3] = v.w;
            }
            float node_acc[LT][SK];
#pragma unroll
            for (int i = 0; i < LT; ++i)
#pragma unroll
                for (int q = 0; q < SK; ++q) node_acc[i][q] = 0.f;
#pragma unroll
            for (int ss = 0; ss < ST; ++ss) {
                const float zx = dot4(zr[0][ss], xr[0]);
#pragma unroll
                for (int i = 0; i < LT; ++i) {
                    const float t = i == 0 ? zx : dot4(zr[i][ss], xr[i]);
                    float prod = zx * t;
                    if (DROP) prod = keep.at(ss * LT + i) ? prod : 0.f;
                    node_acc[i][KS ? ss : 0] += prod;
                }
            }
#pragma unroll
            for (int i = 0; i < LT; ++i)
#pragma unroll
                for (int q = 0; q < SK; ++q) acc[i][q] += node_acc[i][q];
        }
        if (more) {
            // its readers all passed the previous barrier
            if constexpr (kPairs) put_slot_pairs<LT>(tiles[buf ^ 1], slot, has_stream);
            else put_slot<LT>(tiles[buf ^ 1], slot, has_stream);
            __syncthreads();
        }
    }
    if (g_cur >= 0 && lane_h < h) flush(g_cur);
}

// KS: d_k holds one value per (step, hidden graph, hidden node), layout as the forward kernel's slots.
template <int LT, int ST, bool DROP, int HPT, bool KS = false>
__global__ void __launch_bounds__(kBlock, bwd_waves_per_simd(LT * ST))
backward_kernel(const float *__restrict__ x, const float *__restrict__ z, const int *__restrict__ gid,
                const int *__restrict__ perm, const float *__restrict__ d_k, const float *__restrict__ counts_f32,
                int n, int d, int h, int stream_len, int hp_arg, int prio, float scale, int normalize,
                uint32_t t0, uint32_t t1, const unsigned long long *__restrict__ rng, float *__restrict__ dx,
                float *__restrict__ dz_partials, BnFold bn) {
    constexpr int NC = (LT * ST + 14) / 15;
    const int hp = HPT ? HPT : hp_arg;
    constexpr int V = LT * ST * DP;               // gradient values per lane
    constexpr int CH = V < 16 ? V : 16;           // values staged per LDS round of the final reduction
    __shared__ Tile<LT> tiles[2];                 // tile t in tiles[t & 1]; the other one is being filled
    __shared__ float red[CH][kBlock];
    __shared__ float s_bn[64];                    // folded batch-norm backward: [dgamma sums | dbeta sums], times d_loss

    RWK_STAMP_DECL();
    const int tid = threadIdx.x;
    // folded batch-norm backward (never with KS): the 2c column sums of the head's partial rows, requested before
    // anything else is waited for and added in the order of the head's own finishing kernel (8 lanes per column,
    // rows lane8, lane8 + 8, ..., then the eight lanes)
    // Compiled into the instances of the three-workgroups-per-CU tier only (L*S <= 10): the wider ones have no
    // registers to spare (32 more live values in the prologue made the L*S = 20 instance spill).
    constexpr bool kFoldBuilt = !KS && bn_fold_built(LT, ST);
    const bool fold = kFoldBuilt && bn.partials != nullptr;
    RwkFinPartials bn_pre[kFoldBuilt ? 2 : 1];
    const int bn_c = LT * h;
    // The partial rows are loaded whether or not the fold is on (without it, from z[0]): under a condition the block
    // ends in a wait for the values the compiler moves where the two paths meet, and everything behind it - the
    // first slot, Z - is requested a trip to memory later.
    const bool fast = d == DP && perm == nullptr;
    if constexpr (kFoldBuilt) {
        const float *bp = fold ? bn.partials : z;
        const int bn_n = fold ? bn.n_partials : 1, bn_nv = fold ? bn.nv : 0;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int q = it * (kBlock / 8) + (tid >> 3);
            bn_pre[it].load(bp, bn_n, bn_nv, fold ? bn.o_ga + (q < 2 * bn_c ? q : 2 * bn_c - 1) : 0, tid & 7);
        }
    }
    const int lane_h = tid % hp;
    const int hh = lane_h < h ? lane_h : h - 1;
    const bool own_h = lane_h < h;
    const int group0 = tid - lane_h;
    const int c = rwk_xcd_block(blockIdx.x, gridDim.x) * (kBlock / hp) + tid / hp;
    const long long start_ll = (long long)c * stream_len;
    const bool has_stream = start_ll < n;
    const int start = has_stream ? (int)start_ll : 0;
    const int end = has_stream ? (int)(start_ll + stream_len < n ? start_ll + stream_len : n) : 0;
    const size_t step_stride = (size_t)n * d;
    Slot<LT> slot;
    float zr[LT][ST][DP];
    if (fast) {
        first_loads4<LT, ST>(slot, zr, x, z, gid, start + lane_h, end, has_stream, step_stride, h, hh);
    } else {
        fetch_slot<LT>(slot, x, gid, perm, start + lane_h, end, has_stream, d, step_stride);
        load_z<LT, ST>(z, h, hh, d, zr);
    }
    DropCfg cfg;
    if (DROP) cfg = load_cfg(rng, t0, t1);
    const int lh = LT * h;

    float dz[V];
#pragma unroll
    for (int v = 0; v < V; ++v) dz[v] = 0.f;
    // two steps: the arithmetic runs on (step 0, step 1) pairs, one packed instruction for two of the scalar ones
    // (same operations on the same operands: same bits); dz2[ss][k] = (dZ_0[ss][k], dZ_1[ss][k]) is unpacked into
    // dz[] after the loop
    // (four steps only with sixteen lanes per stream: the run-time lane count's instance has no registers to spare)
    constexpr bool kPairs = !KS && (LT == 2 || (LT == 4 && HPT == 16));
    constexpr int NP = kPairs ? LT / 2 : 1;
    // ... and with sixteen lanes per stream the tile holds the rows interleaved (put_slot_pairs): the operand pairs
    // come out of the 16-byte LDS reads as they stand, and the dX totals are parked in the same interleaved places
    constexpr bool kPairTile = kPairs && HPT == 16 && (LT == 2 || LT == 4);
    rwk_f2 zp[NP][kPairs ? ST : 1][DP], dz2[NP][kPairs ? ST : 1][DP];
    if constexpr (kPairs) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int ss = 0; ss < ST; ++ss)
#pragma unroll
                for (int k = 0; k < DP; ++k) {
                    zp[p][ss][k] = rwk_f2{zr[2 * p][ss][k], zr[2 * p + 1][ss][k]};
                    dz2[p][ss][k] = rwk_f2{0.f, 0.f};
                }
    }
    constexpr int SK = KS ? ST : 1;
    float dp_cur[LT][SK];
#pragma unroll
    for (int i = 0; i < LT; ++i)
#pragma unroll
        for (int q = 0; q < SK; ++q) dp_cur[i][q] = 0.f;
    int g_cur = -1;

    // where this lane puts its dX total of the halving reduction: float (idx % 4) of row (idx / 4) of slot group0 + j
    int red_off = 0;
    if (HPT == 16 && (LT == 2 || LT == 4)) {
        const int idx = rwk_halving_index<(LT == 4 ? 16 : 8)>(lane_h);
        red_off = (((idx / DP) % LT) * kBlock + group0) * DP + idx % DP;      // in floats from &tile.x[0][0]
        // interleaved rows (put_slot_pairs): value (i, k) is float (k % 2) * 2 + i % 2 of tile.x[2 (i / 2) + k / 2][slot]
        if (kPairTile) {
            const int i = (idx / DP) % LT, k = idx % DP;
            red_off = ((2 * (i / 2) + k / 2) * kBlock + group0) * DP + (k % 2) * 2 + i % 2;
        }
    }
    float bn_dl = 1.f;
    if constexpr (kFoldBuilt) if (fold) {
        bn_dl = bn.d_loss[0];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int q = it * (kBlock / 8) + (tid >> 3);
            float acc = bn_pre[it].sum(bn.partials, bn.n_partials, bn.nv, bn.o_ga + (q < 2 * bn_c ? q : 2 * bn_c - 1),
                                       tid & 7);
            acc = rwk_group_sum(acc, 8);
            if ((tid & 7) == 0 && q < 2 * bn_c) s_bn[q < bn_c ? q : 32 + (q - bn_c)] = acc * bn_dl;
        }
    }
    if constexpr (kPairTile) put_slot_pairs<LT>(tiles[0], slot, has_stream);
    else put_slot<LT>(tiles[0], slot, has_stream);
    __syncthreads();
    ProgressPriority priority(prio, stream_len);
    int buf = 0;
    for (int base = 0; base < stream_len; base += hp, buf ^= 1) {
        Tile<LT> &tile = tiles[buf];
        float *red_dst = reinterpret_cast<float *>(&tile.x[0][0]) + red_off;
        RWK_STAMP_AT(0);
        const int my_node = tile.node[tid];              // the slot this lane loaded and later stores
        const bool more = base + hp < stream_len;
        if (more) fetch_slot<LT>(slot, x, gid, perm, start + base + hp + lane_h, end, has_stream, d, step_stride);
        RWK_STAMP_AT(1);
        // every lane of the wavefront runs all hp iterations: the DPP butterfly needs them all
        const int jmax = stream_len - base < hp ? stream_len - base : hp;
        for (int j = 0; j < jmax; ++j) {
            priority.at(base + j);
            const int node = tile.node[group0 + j];
            const int g = tile.gid[group0 + j];
            float xr[LT][DP], dxr[LT][DP];
#pragma unroll
            for (int i = 0; i < LT; ++i) {
                const float4 v = tile.x[i][group0 + j];
                if constexpr (kPairTile) {      // with i = 2p + q: v = (x_2p[2q], x_2p+1[2q], x_2p[2q+1], x_2p+1[2q+1])
                    constexpr int kOdd = LT > 1 ? 1 : 0;
                    const int lo = i - i % 2, hi = lo + kOdd, q = i % 2;
                    xr[lo][2 * q] = v.x; xr[hi][2 * q] = v.y; xr[lo][2 * q + 1] = v.z; xr[hi][2 * q + 1] = v.w;
                } else {
                    xr[i][0] = v.x; xr[i][1] = v.y; xr[i][2] = v.z; xr[i][3] = v.w;
                }
#pragma unroll
                for (int k = 0; k < DP; ++k) dxr[i][k] = 0.f;
            }
            if (g != g_cur) {                             // first node of a graph: fetch its dK row
                g_cur = g;
                float f = scale;
                if (normalize) f /= counts_f32[g];
                if (kFoldBuilt && fold) {
                    // this lane's elements of d(loss)/d(K[g]) from the head's intermediates (every load independent)
                    float e_dy[LT], e_k[LT], e_mu[LT], e_is[LT], e_w[LT];
#pragma unroll
                    for (int i = 0; i < LT; ++i) {
                        const int col = i * h + hh;
                        e_dy[i] = bn.dy[(size_t)g * lh + col];
                        e_k[i] = bn.k[(size_t)g * lh + col];
                        e_mu[i] = bn.mean[col];
                        e_is[i] = bn.invstd[col];
                        e_w[i] = bn.bn_w[col];
                    }
#pragma unroll
                    for (int i = 0; i < LT; ++i) {
                        const int col = i * h + hh;
                        const float dk = rwk_bn_input_grad(e_dy[i], e_k[i], e_mu[i], e_is[i], e_w[i] * e_is[i],
                                                           s_bn[col], s_bn[32 + col], bn.inv_g, bn_dl, true);
                        dp_cur[i][0] = own_h ? dk * f : 0.f;
                    }
                } else {
                    // every lane loads (a lane without a hidden graph at its neighbour's address) and the unused values
                    // are dropped afterwards: under a per-lane condition each of the LT * SK loads sat in a block of
                    // its own with a wait behind it - ten trips to the L2 in a row at two steps and five hidden nodes
                    const int lane_c = own_h ? lane_h : h - 1;
                    float t[LT][SK];
#pragma unroll
                    for (int i = 0; i < LT; ++i)
#pragma unroll
                        for (int q = 0; q < SK; ++q) t[i][q] = d_k[((size_t)g * lh + i * h + lane_c) * SK + q];
#pragma unroll
                    for (int i = 0; i < LT; ++i)
#pragma unroll
                        for (int q = 0; q < SK; ++q) dp_cur[i][q] = own_h ? t[i][q] * f : 0.f;
                }
            }
            // no validity branch: a slot past the end of the stream has x = 0, so zx = t = 0 and every
            // product below is zero
            if constexpr (kPairs) {
                Keep<NC> keep;
                if (DROP) keep.draw(cfg, (uint32_t)node, (uint32_t)lane_h);
                rwk_f2 xp[NP][DP], dx2[NP][DP];
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int k = 0; k < DP; ++k) {
                        xp[p][k] = rwk_f2{xr[2 * p][k], xr[(2 * p + 1) % LT][k]};
                        dx2[p][k] = rwk_f2{0.f, 0.f};
                    }
#pragma unroll
                for (int ss = 0; ss < ST; ++ss) {
                    rwk_f2 zt[NP];                                                  // (zx, t_1), (t_2, t_3), ...
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        zt[p] = zp[p][ss][0] * xp[p][0];
                        zt[p] = __builtin_elementwise_fma(zp[p][ss][1], xp[p][1], zt[p]);
                        zt[p] = __builtin_elementwise_fma(zp[p][ss][2], xp[p][2], zt[p]);
                        zt[p] = __builtin_elementwise_fma(zp[p][ss][3], xp[p][3], zt[p]);
                    }
                    float w[LT];
#pragma unroll
                    for (int i = 0; i < LT; ++i) {
                        w[i] = dp_cur[i][0];
                        if (DROP) w[i] = keep.at(ss * LT + i) ? w[i] : 0.f;
                    }
                    const float zx = zt[0].x;
                    float dzx = 2.f * w[0] * zx;          // step 0 contributes keep * zx^2
#pragma unroll
                    for (int i = 1; i < LT; ++i) dzx = fmaf(w[i], i % 2 ? zt[i / 2].y : zt[i / 2].x, dzx);
                    rwk_f2 cf[NP];                                                  // (d zx, d t_1), (d t_2, d t_3), ...
                    cf[0] = rwk_f2{dzx, w[1 % LT] * zx};
#pragma unroll
                    for (int p = 1; p < NP; ++p) cf[p] = rwk_f2{w[2 * p], w[(2 * p + 1) % LT]} * rwk_f2{zx, zx};
#pragma unroll
                    for (int p = 0; p < NP; ++p)
#pragma unroll
                        for (int k = 0; k < DP; ++k) {
                            dx2[p][k] = __builtin_elementwise_fma(cf[p], zp[p][ss][k], dx2[p][k]);
                            dz2[p][ss][k] = __builtin_elementwise_fma(cf[p], xp[p][k], dz2[p][ss][k]);
                        }
                }
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int k = 0; k < DP; ++k) {
                        dxr[2 * p][k] = dx2[p][k].x;
                        dxr[(2 * p + 1) % LT][k] = dx2[p][k].y;
                    }
            } else {
                Keep<NC> keep;
                if (DROP) keep.draw(cfg, (uint32_t)node, (uint32_t)lane_h);
#pragma unroll
                for (int ss = 0; ss < ST; ++ss) {
                    const float zx = dot4(zr[0][ss], xr[0]);
                    float w[LT];
#pragma unroll
                    for (int i = 0; i < LT; ++i) {
                        w[i] = dp_cur[i][KS ? ss : 0];
                        if (DROP) w[i] = keep.at(ss * LT + i) ? w[i] : 0.f;
                    }
                    float dzx = 2.f * w[0] * zx;          // step 0 contributes keep * zx^2
#pragma unroll
                    for (int i = 1; i < LT; ++i) {
                        const float t = dot4(zr[i][ss], xr[i]);
                        dzx = fmaf(w[i], t, dzx);
                        const float dt = w[i] * zx;
#pragma unroll
                        for (int k = 0; k < DP; ++k) {
                            dxr[i][k] = fmaf(dt, zr[i][ss][k], dxr[i][k]);
                            dz[(i * ST + ss) * DP + k] = fmaf(dt, xr[i][k], dz[(i * ST + ss) * DP + k]);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < DP; ++k) {
                        dxr[0][k] = fmaf(dzx, zr[0][ss][k], dxr[0][k]);
                        dz[ss * DP + k] = fmaf(dzx, xr[0][k], dz[ss * DP + k]);
                    }
                }
            }
            // sum the dX rows over the hidden graphs (the hp lanes of this stream); the totals go where the x
            // rows of slot j were: the lanes of a stream share a wavefront, so all of them have read those rows
            if (HPT == 16 && (LT == 2 || LT == 4)) {
                // halving reduction: lane l ends with the total of value (i, k) = (idx / 4, idx % 4) and stores it
                constexpr int V = (LT == 2 || LT == 4) ? LT * DP : 8;
                float vals[V];
#pragma unroll
                for (int e = 0; e < V; ++e) vals[e] = dxr[(e / DP) % LT][e % DP];
                const float tot = rwk_halving_sum16<V>(vals, lane_h);
                if (V == 16 || (lane_h & 1) == 0) red_dst[4 * j] = tot;
            } else {
#pragma unroll
                for (int i = 0; i < LT; ++i) {
                    float tot[DP];
#pragma unroll
                    for (int k = 0; k < DP; ++k) tot[k] = rwk_group_sum(dxr[i][k], hp);
                    if (lane_h == j) tile.x[i][group0 + j] = make_float4(tot[0], tot[1], tot[2], tot[3]);
                }
            }
        }
        RWK_STAMP_AT(2);
        if (my_node >= 0) {                               // coalesced: hp lanes store hp consecutive rows
#pragma unroll
            for (int i = 0; i < LT; ++i) {
                float4 mv = tile.x[i][tid];               // written by a lane of this wavefront
                if constexpr (kPairTile) {                // de-interleave: row i = float i % 2 of every pair
                    const float4 ma = tile.x[i - i % 2][tid], mb = tile.x[(i - i % 2 + 1) % LT][tid];
                    mv = i % 2 == 0 ? make_float4(ma.x, ma.z, mb.x, mb.z) : make_float4(ma.y, ma.w, mb.y, mb.w);
                }
                const float mine_i[DP] = {mv.x, mv.y, mv.z, mv.w};
                float *row = dx + i * step_stride + (size_t)my_node * d;
                if (d == DP) {
                    *reinterpret_cast<float4 *>(row) = mv;
                } else {
#pragma unroll
                    for (int k = 0; k < DP; ++k)
                        if (k < d) row[k] = mine_i[k];
                }
            }
        }
        if (more) {
            // its readers all passed the previous barrier
            if constexpr (kPairTile) put_slot_pairs<LT>(tiles[buf ^ 1], slot, has_stream);
            else put_slot<LT>(tiles[buf ^ 1], slot, has_stream);
            __syncthreads();
        }
    }

    RWK_STAMP_AT(3);
    if constexpr (kPairs) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int ss = 0; ss < ST; ++ss)
#pragma unroll
                for (int k = 0; k < DP; ++k) {
                    dz[(2 * p * ST + ss) * DP + k] = dz2[p][ss][k].x;
                    dz[(((2 * p + 1) % LT) * ST + ss) * DP + k] = dz2[p][ss][k].y;
                }
    }
    // per-workgroup partial of dZ: threads that own the same hidden graph are added in thread order
    const int spb = kBlock / hp;
    float *out = dz_partials + (size_t)rwk_xcd_block(blockIdx.x, gridDim.x) * LT * h * ST * d;   // the row of the logical block
#pragma unroll
    for (int cb = 0; cb < V; cb += CH) {
        __syncthreads();
#pragma unroll
        for (int vv = 0; vv < CH; ++vv)
            if (cb + vv < V) red[vv][tid] = dz[cb + vv];
        __syncthreads();
        for (int o = tid; o < CH * h; o += kBlock) {
            const int hq = o % h, vv = o / h;
            const int v = cb + vv;
            if (v >= V) continue;
            const int k = v % DP, ss = (v / DP) % ST, i = v / (DP * ST);
            if (k >= d) continue;
            float acc = 0.f;
            for (int q = 0; q < spb; ++q) acc += red[vv][q * hp + hq];
            out[(((size_t)i * h + hq) * ST + ss) * d + k] = acc;
        }
    }
    RWK_STAMP_AT(4);
    RWK_STAMP_FLUSH();
}

inline int pow2_ceil(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// keep-S instances: sixteen-lane layout, at most 20 (step, hidden node) accumulators per lane
constexpr bool keep_s_built(int lt, int st) { return st > 1 && lt * st <= 20; }

template <int LT, int ST>
int launch_fwd(const Args &a) {
    const int hp = pow2_ceil(a.h);
    const int n_streams = (int)rwk_ceil_div(a.n, a.stream_len);
    const dim3 grid((unsigned)rwk_ceil_div(n_streams, kBlock / hp));
    const int prio = grid.x <= 256u * (unsigned)fwd_waves_per_simd(LT * ST);      // all workgroups resident at once
#define RWK_FWD(DROPV, HPV)                                                                                       \
    hipLaunchKernelGGL((forward_kernel<LT, ST, DROPV, HPV>), grid, dim3(kBlock), 0, a.stream, a.x, a.z, a.gid, a.perm, \
                       a.n, a.d, a.h, a.stream_len, hp, prio, a.t0, a.t1, a.rng, a.k_partials)
    if (a.keep_s) {
        if constexpr (keep_s_built(LT, ST)) {
            if (hp != 16) return RWK_E_UNSUPPORTED;
            if (a.drop)
                hipLaunchKernelGGL((forward_kernel<LT, ST, true, 16, true>), grid, dim3(kBlock), 0, a.stream, a.x, a.z,
                                   a.gid, a.perm, a.n, a.d, a.h, a.stream_len, hp, prio, a.t0, a.t1, a.rng, a.k_partials);
            else
                hipLaunchKernelGGL((forward_kernel<LT, ST, false, 16, true>), grid, dim3(kBlock), 0, a.stream, a.x, a.z,
                                   a.gid, a.perm, a.n, a.d, a.h, a.stream_len, hp, prio, a.t0, a.t1, a.rng, a.k_partials);
        } else {
            return RWK_E_UNSUPPORTED;
        }
    } else if (hp == 16) { if (a.drop) RWK_FWD(true, 16); else RWK_FWD(false, 16); }
    else { if (a.drop) RWK_FWD(true, 0); else RWK_FWD(false, 0); }
#undef RWK_FWD
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

template <int LT, int ST>
int launch_bwd(const Args &a) {
    const int hp = pow2_ceil(a.h);
    const dim3 grid((unsigned)backward_blocks(a.n, a.stream_len, a.h));
    const int prio = grid.x <= 256u * (unsigned)bwd_waves_per_simd(LT * ST);      // all workgroups resident at once
#define RWK_BWD(DROPV, HPV)                                                                                        \
    hipLaunchKernelGGL((backward_kernel<LT, ST, DROPV, HPV>), grid, dim3(kBlock), 0, a.stream, a.x, a.z, a.gid, a.perm, \
                       a.d_k, a.counts_f32, a.n, a.d, a.h, a.stream_len, hp, prio, a.scale, a.normalize, a.t0, a.t1, a.rng, \
                       a.dx, a.dz_partials, a.bn)
    if (a.keep_s) {
        if constexpr (keep_s_built(LT, ST)) {
            if (hp != 16) return RWK_E_UNSUPPORTED;
            if (a.drop)
                hipLaunchKernelGGL((backward_kernel<LT, ST, true, 16, true>), grid, dim3(kBlock), 0, a.stream, a.x, a.z,
                                   a.gid, a.perm, a.d_k, a.counts_f32, a.n, a.d, a.h, a.stream_len, hp, prio, a.scale,
                                   a.normalize, a.t0, a.t1, a.rng, a.dx, a.dz_partials, a.bn);
            else
                hipLaunchKernelGGL((backward_kernel<LT, ST, false, 16, true>), grid, dim3(kBlock), 0, a.stream, a.x,
                                   a.z, a.gid, a.perm, a.d_k, a.counts_f32, a.n, a.d, a.h, a.stream_len, hp, prio, a.scale,
                                   a.normalize, a.t0, a.t1, a.rng, a.dx, a.dz_partials, a.bn);
        } else {
            return RWK_E_UNSUPPORTED;
        }
    } else if (hp == 16) { if (a.drop) RWK_BWD(true, 16); else RWK_BWD(false, 16); }
    else { if (a.drop) RWK_BWD(true, 0); else RWK_BWD(false, 0); }
#undef RWK_BWD
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

template <int LT>
int by_s(const Args &a, bool bwd) {
    switch (a.s) {
        case 1: return bwd ? launch_bwd<LT, 1>(a) : launch_fwd<LT, 1>(a);
        case 3: return bwd ? launch_bwd<LT, 3>(a) : launch_fwd<LT, 3>(a);
        case 4: return bwd ? launch_bwd<LT, 4>(a) : launch_fwd<LT, 4>(a);
        case 5: return bwd ? launch_bwd<LT, 5>(a) : launch_fwd<LT, 5>(a);
        case 6: return bwd ? launch_bwd<LT, 6>(a) : launch_fwd<LT, 6>(a);
        case 8: return bwd ? launch_bwd<LT, 8>(a) : launch_fwd<LT, 8>(a);
        case 10: return bwd ? launch_bwd<LT, 10>(a) : launch_fwd<LT, 10>(a);
    }
    return RWK_E_UNSUPPORTED;
}

int by_l(const Args &a, bool bwd) {
    switch (a.max_step) {
        case 1: return by_s<1>(a, bwd);
        case 2: return by_s<2>(a, bwd);
        case 3: return by_s<3>(a, bwd);
        case 4: return by_s<4>(a, bwd);
    }
    return RWK_E_UNSUPPORTED;
}

}  // namespace

bool supported(int64_t d, int64_t h, int64_t s, int64_t max_step) {
    if (d < 1 || d > DP || h < 1 || h > 64 || max_step < 1 || max_step > 4) return false;
    return s == 1 || s == 3 || s == 4 || s == 5 || s == 6 || s == 8 || s == 10;
}

bool bn_fold_supported(int64_t d, int64_t h, int64_t s, int64_t max_step) {
    // the 2c column sums fit the kernel's 64 slots, and the instance carries the code
    return supported(d, h, s, max_step) && max_step * h <= 32 && bn_fold_built((int)max_step, (int)s);
}

bool keep_s_supported(int64_t d, int64_t h, int64_t s, int64_t max_step) {
    return supported(d, h, s, max_step) && pow2_ceil((int)h) == 16 && keep_s_built((int)max_step, (int)s);
}

int64_t backward_blocks(int64_t n, int64_t stream_len, int64_t h) {
    const int hp = pow2_ceil((int)h);
    return rwk_ceil_div(rwk_ceil_div(n, stream_len), kBlock / hp);
}

// Workgroups of kBlock threads hold kBlock/hp streams; a CU keeps fwd_waves_per_simd() forward workgroups resident
// and bwd_waves_per_simd() backward ones.  Aim at a whole number m of full rounds over the 256 CUs with about
// 20 positions per stream: a last round that is mostly empty costs as much as a full one.
int64_t stream_len(int64_t n, int64_t h, int64_t s, int64_t max_step, bool bwd) {
    const int hp = pow2_ceil((int)h);
    const int64_t resident = bwd ? bwd_waves_per_simd((int)(s * max_step)) : fwd_waves_per_simd((int)(s * max_step));
    const int64_t per_round = (int64_t)(kBlock / hp) * 256 * resident;
    int64_t m = (n + per_round * 10) / (per_round * 20);
    if (m < 1) m = 1;
    const int64_t len = rwk_ceil_div(n, per_round * m);
    return len < 8 ? 8 : len;
}

#ifdef RWK_STAMP
extern "C" int rwk_debug_read_reg_stamps(unsigned long long *host32) {
    return (int)hipMemcpyFromSymbol(host32, HIP_SYMBOL(rwk_reg_stamp_buf), sizeof(unsigned long long) * 32);
}
extern "C" int rwk_debug_read_reg_blocks(unsigned long long *host4096) {
    return (int)hipMemcpyFromSymbol(host4096, HIP_SYMBOL(rwk_reg_stamp_blocks), sizeof(unsigned long long) * 4096);
}
#endif

int forward(const Args &a) { return by_l(a, false); }
int backward(const Args &a) { return by_l(a, true); }

}  // namespace rwk_reg
