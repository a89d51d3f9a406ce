// Fused random-walk kernels on the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, bit-for-bit an
// fmaf chain) for hidden_dim > 4.
//
// Tile algebra.  One MFMA computes D(16x16) += A(16x4) B(4x16); lane l supplies A[l&15][l>>4] and
// B[l>>4][l&15] and receives D[4*(l>>4)+r][l&15], r = 0..3.  With q = l>>4 and c = l&15:
//
//   T = Z X^T  (16 hidden graphs x 16 nodes):  A = Z_i[h = c][d],  B = X_i[node = c][d]; the four d's of one
//       16-byte chunk (d = 16*tc + 4*q + e, e = 0..3) are the k-slots of four consecutive MFMAs, so each
//       operand is one 16-byte load per four MFMAs.  The lane ends up with T[h = 4q+r][node = c].
//   dX = G^T Z (16 nodes x 16 d's): the accumulator registers of G are used directly as the A operand: in
//       the MFMA for register r, k-slot q stands for hidden graph 4q+r, so B = Z_i[h = 4q+r][16*tc + c].
//   dZ = G X   (16 hidden graphs x 16 d's): needs G transposed (node on the k index): one 16x16 transpose
//       through LDS, then A = G^T register r (k-slot q = node 4q+r), B = X_i[node = 4q+r][16*tc + c].
//
// Rows of a tile are 16 hidden graphs at a fixed hidden node s, so the sum over s is a sum of accumulator
// tiles in registers.  A workgroup owns one block of 16 hidden graphs (blockIdx.y) and keeps their Z rows of
// all steps in LDS.  Columns are 16 independent streams of consecutive sorted positions, as in the
// vector-ALU kernels: partial sums go to slot (stream + graph) of k_partials, no atomics in the forward.  The
// backward has ONE kind of float atomic: with more than one block of 16 hidden graphs, the first block stores its
// share of dX and the later blocks add theirs at the L2 (global_atomic_add_f32 without return value).  Every dX
// element has one owner lane that issues its store and its adds in block order, so the sum is formed in a fixed
// order and the result is bitwise reproducible (the same bits as load + add + store).
#include "rwk_fused_mfma.h"

namespace rwk_mfma {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kFwdBlock = 512;     // 8 wavefronts, each with its own 16 streams
constexpr int kBwdBlock = 512;     // 8 wavefronts sharing 16 streams
constexpr int kBwdWaves = 8;
constexpr int kSMax = 12;          // hidden nodes per hidden graph (dZ accumulators are unrolled over them)
constexpr int kGS = 20;            // row stride (floats) of a 16 x 16 gradient tile in LDS
constexpr size_t kLdsBudget = 150 * 1024;

struct DropCfg {
    uint32_t t0, t1, seed_lo, key_hi, off_lo;
};

__device__ inline DropCfg load_cfg(const unsigned long long *rng, uint32_t t0, uint32_t t1) {
    const unsigned long long seed = rng[0], off = rng[1];
    return DropCfg{t0, t1, (uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(off >> 32), (uint32_t)off};
}

// keep bits of the (node, hidden graph) stream: bit e <-> sample e = s*L + i  (n_calls*15 <= 45 samples)
__device__ inline unsigned long long keep_bits(const DropCfg &c, uint32_t node, uint32_t hq, int n_calls) {
    unsigned long long bits = 0;
#pragma unroll
    for (int call = 0; call < 3; ++call) {
        if (call < n_calls) {
            const RwkPhilox4 r = rwk_philox_mask(node, hq, (uint32_t)call, c.off_lo, c.seed_lo, c.key_hi);
            const uint32_t thr = c.t0 + ((r.w >> 24) < c.t1 ? 1u : 0u);
            const uint32_t w[4] = {r.x, r.y, r.z, r.w};
            uint32_t m = 0;
#pragma unroll
            for (int j = 0; j < 15; ++j) m |= (((w[j >> 2] >> (8 * (j & 3))) & 0xFFu) < thr ? 1u : 0u) << j;
            bits |= (unsigned long long)m << (15 * call);
        }
    }
    return bits;
}

// v if bit `bit` of kk is set, else +0: the bit is spread to a mask (one signed bit-field extract) and ANDed in - two
// instructions and no trip through the condition register, against shift, and, compare, select.
__device__ inline float keep_or_zero(float v, uint32_t kk, int bit) {
    const int m = __builtin_amdgcn_sbfe((int)kk, bit, 1);
    return __int_as_float(__float_as_int(v) & m);
}

// Rows are 16-byte aligned (hidden_dim is a multiple of 4 on this path): one 16-byte load per piece.  A piece
// that starts beyond d is read from the row's last piece instead and must be masked by the caller.
__device__ inline float4 load4_raw(const float *__restrict__ row, int d, int off) {
    return *reinterpret_cast<const float4 *>(row + (off < d ? off : d - 4));
}
__device__ inline float4 load4(const float *__restrict__ row, int d, int off) {
    const float4 v = load4_raw(row, d, off);
    return off < d ? v : make_float4(0.f, 0.f, 0.f, 0.f);
}

// Row layout of the 16-row operand tiles in LDS (Z rows of 16 hidden graphs, x rows of 16 nodes).
// hidden_dim 49..64 (NT = 4): rows of 64 floats without padding, the 16-byte slot c of row r stored at slot
// c ^ r.  The lane groups a ds_read_b128 is served in ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32:
// one 16-byte slot of the 256-byte bank row per lane) then read 16 different slots when lane (q, col) reads slot
// 4 tc + q of row col, and the scalar reads of [row 4q + r][16 tc + col] (32 lanes per group, 32 banks) fall on
// 32 different banks.  With the 4-float row padding used for the narrower tiles, two lanes of every b128 group
// share a slot (measured: half of the kernels' LDS cycles were bank conflicts).
template <int NT>
constexpr int row_stride() { return NT == 4 ? 64 : 16 * NT + 4; }
template <int NT>
__device__ inline int roff(int row, int f) {                  // float offset of element f of row `row` (0..15)
    if constexpr (NT == 4) return row * 64 + ((((f >> 2) ^ row) & 15) << 2) + (f & 3);
    else return row * (16 * NT + 4) + f;
}
// A 16 x 16 gradient tile G[node][hidden graph] of the pipelined backward kernel: node n in row 4 (n % 4) + n / 4,
// the 16-byte slot c of that row at slot c ^ (n % 4).  Conflict-free for the 16-byte accesses of lane (q, col) to
// G[col][4q..4q+3] and for the scalar reads of G[4q + r][col].
__device__ inline int g_row(int n16) { return ((n16 & 3) << 2) | (n16 >> 2); }
__device__ inline int goff(int n, int hgraph) {
    return g_row(n) * 16 + ((((hgraph >> 2) ^ n) & 3) << 2) + (hgraph & 3);
}

// Z rows of hidden graphs [16*hb, 16*hb+16), all steps and hidden nodes, into LDS: [i][s][16 rows, see roff].  16-byte pieces
// (hidden_dim is a multiple of 4 on this path), four of a thread's loads in flight at a time.
template <int LT, int NT>
__device__ inline void stage_z(const float *__restrict__ z, float *zs, int h, int s, int d, int hb, int block) {
    constexpr int DPAD = 16 * NT, ZS = row_stride<NT>(), PPR = DPAD / 4;         // pieces per row
    const int total = LT * s * 16 * PPR;
    for (int idx0 = threadIdx.x; idx0 < total; idx0 += 4 * block) {
        float4 v[4];
        int dst[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = idx0 + u * block;
            const int ic = idx < total ? idx : total - 1;
            const int k4 = ic % PPR;
            int r = ic / PPR;
            const int hl = r % 16; r /= 16;
            const int ss = r % s;
            const int i = r / s;
            const int hq = 16 * hb + hl;
            v[u] = load4_raw(z + (((size_t)i * h + (hq < h ? hq : h - 1)) * s + ss) * d, d, 4 * k4);
            ok[u] = hq < h && 4 * k4 < d;
            dst[u] = idx < total ? (i * s + ss) * 16 * ZS + roff<NT>(hl, 4 * k4) : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (dst[u] >= 0)
                *reinterpret_cast<float4 *>(zs + dst[u]) = ok[u] ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// Diagnostic build only (-DRWK_STAMP): per-wavefront cycle shares of the backward step's segments.
#ifdef RWK_STAMP
__device__ unsigned long long rwk_stamp_buf[8 * 8];
#define RWK_STAMP_DECL() unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0
#define RWK_STAMP_AT(k)                                                      \
    do {                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                   \
        const unsigned long long now_ = __builtin_readcyclecounter();        \
        __builtin_amdgcn_sched_barrier(0);                                   \
        if ((k) != 0) st_sum[k] += now_ - st_last;                           \
        else if (st_last != 0) st_sum[0] += now_ - st_last;                  \
        st_last = now_;                                                      \
    } while (0)
#define RWK_STAMP_FLUSH()                                                    \
    do {                                                                     \
        if (blockIdx.x == 3 && blockIdx.y == 0 && (threadIdx.x & 63) == 0)   \
            for (int k_ = 0; k_ < 8; ++k_) rwk_stamp_buf[(threadIdx.x >> 6) * 8 + k_] = st_sum[k_]; \
    } while (0)
#else
#define RWK_STAMP_DECL()
#define RWK_STAMP_AT(k)
#define RWK_STAMP_FLUSH()
#endif

// pipelined kernel: elapsed cycles since the previous stamp go to bucket k; bucket 6 / 7 = whole kernel in
// shader cycles / 100 MHz ticks
#ifdef RWK_STAMP
#define RWK_PSTAMP_DECL()                                                        \
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};                     \
    unsigned long long st_last = __builtin_readcyclecounter();                   \
    const unsigned long long st_c0 = st_last, st_rt0 = __builtin_amdgcn_s_memrealtime()
#define RWK_PSTAMP(k)                                                            \
    do {                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                       \
        const unsigned long long now_ = __builtin_readcyclecounter();            \
        __builtin_amdgcn_sched_barrier(0);                                       \
        st_sum[k] += now_ - st_last;                                             \
        st_last = now_;                                                          \
    } while (0)
#define RWK_PSTAMP_FLUSH()                                                       \
    do {                                                                         \
        st_sum[6] = __builtin_readcyclecounter() - st_c0;                        \
        st_sum[7] = __builtin_amdgcn_s_memrealtime() - st_rt0;                   \
        if (blockIdx.x == 3 && blockIdx.y == 0 && (threadIdx.x & 63) == 0)       \
            for (int k_ = 0; k_ < 8; ++k_) rwk_stamp_buf[(threadIdx.x >> 6) * 8 + k_] = st_sum[k_]; \
    } while (0)
#else
#define RWK_PSTAMP_DECL()
#define RWK_PSTAMP(k)
#define RWK_PSTAMP_FLUSH()
#endif

#define RWK_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int LT, int NT, bool DROP>
__global__ void __launch_bounds__(kFwdBlock)
forward_kernel(const float *__restrict__ x, const float *__restrict__ z, const int *__restrict__ gid,
               const int *__restrict__ perm, int n, int d, int h, int s, int stream_len, int n_calls, uint32_t t0,
               uint32_t t1, const unsigned long long *__restrict__ rng, float *__restrict__ k_partials,
               unsigned long long *__restrict__ keep_out) {
    constexpr int ZS = row_stride<NT>();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *zs = lds;
    const int hb = blockIdx.y;
    stage_z<LT, NT>(z, zs, h, s, d, hb, kFwdBlock);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, col = lane & 15;
    const int c = (blockIdx.x * (kFwdBlock / 64) + wave) * 16 + col;       // this column's stream
    const long long start_ll = (long long)c * stream_len;
    const bool has_stream = start_ll < n;
    const int start = has_stream ? (int)start_ll : 0;
    const int end = has_stream ? (int)(start_ll + stream_len < n ? start_ll + stream_len : n) : 0;
    DropCfg cfg;
    if (DROP) cfg = load_cfg(rng, t0, t1);
    const int lh = LT * h;
    const size_t step_stride = (size_t)n * d;
    const int hq0 = 16 * hb + 4 * q;                                       // hidden graph of register 0

    float acc[LT][4];
#pragma unroll
    for (int i = 0; i < LT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    int g_cur = -1;

    for (int t = 0; t < stream_len; ++t) {        // uniform: the MFMAs need every lane of the wavefront
        const int p = start + t;
        const bool valid = has_stream && p < end;
        int node = 0, g = g_cur;
        if (valid) {
            node = perm ? perm[p] : p;
            g = gid[p];
        }
        if (g != g_cur) {
            if (g_cur >= 0) {
#pragma unroll
                for (int i = 0; i < LT; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (hq0 + r < h) k_partials[(size_t)(c + g_cur) * lh + i * h + hq0 + r] = acc[i][r];
            }
#pragma unroll
            for (int i = 0; i < LT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
            g_cur = g;
        }
        float4 xv[LT][NT];
#pragma unroll
        for (int i = 0; i < LT; ++i)
#pragma unroll
            for (int tc = 0; tc < NT; ++tc) {
                const float4 v = load4(x + i * step_stride + (size_t)node * d, d, 16 * tc + 4 * q);
                xv[i][tc] = valid ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        unsigned long long kb[4] = {0, 0, 0, 0};
        if (DROP) {
#pragma unroll
            for (int r = 0; r < 4; ++r) kb[r] = keep_bits(cfg, (uint32_t)node, (uint32_t)(hq0 + r), n_calls);
            // left for the backward pass, which then does not run the generator again
            // (32-bit words when the S * L samples of a pair fit them)
            if (keep_out != nullptr && valid) {
                const bool k32 = s * LT <= 32;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (hq0 + r < h) {
                        const size_t idx = (size_t)node * h + hq0 + r;
                        if (k32) reinterpret_cast<uint32_t *>(keep_out)[idx] = (uint32_t)kb[r];
                        else keep_out[idx] = kb[r];
                    }
                }
            }
        }
        float node_acc[LT][4];
#pragma unroll
        for (int i = 0; i < LT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) node_acc[i][r] = 0.f;
        // The Z operands of a group of 4 LT MFMAs are requested one group ahead (two register sets, chosen by the
        // group's parity at compile time): their LDS round trip passes under the previous group's MFMAs.
        float4 za[2][LT];
        auto load_za = [&](int ss, int tc, float4 (&dst)[LT]) {
#pragma unroll
            for (int i = 0; i < LT; ++i)
                dst[i] = *reinterpret_cast<const float4 *>(zs + (i * s + ss) * 16 * ZS + roff<NT>(col, 16 * tc + 4 * q));
        };
        load_za(0, 0, za[0]);
        for (int ss = 0; ss < s; ++ss) {
            // the LT tiles advance together: a dependent MFMA issues after 40 cycles, an independent one after 32
            f32x4 tt[LT];
#pragma unroll
            for (int i = 0; i < LT; ++i) tt[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tc = 0; tc < NT; ++tc) {
                constexpr bool kEven = (NT & 1) == 0;         // then group tc of every hidden node has parity tc & 1
                const int cur = kEven ? (tc & 1) : 0;
                float4 (&a)[LT] = za[cur];
                if (tc + 1 < NT) load_za(ss, tc + 1, za[cur ^ 1]);
                else load_za(ss + 1 < s ? ss + 1 : ss, 0, za[cur ^ 1]);      // last one: a harmless repeat
#pragma unroll
                for (int i = 0; i < LT; ++i) tt[i] = RWK_MFMA(a[i].x, xv[i][tc].x, tt[i]);
#pragma unroll
                for (int i = 0; i < LT; ++i) tt[i] = RWK_MFMA(a[i].y, xv[i][tc].y, tt[i]);
#pragma unroll
                for (int i = 0; i < LT; ++i) tt[i] = RWK_MFMA(a[i].z, xv[i][tc].z, tt[i]);
#pragma unroll
                for (int i = 0; i < LT; ++i) tt[i] = RWK_MFMA(a[i].w, xv[i][tc].w, tt[i]);
                if (!kEven) {
#pragma unroll
                    for (int i = 0; i < LT; ++i) za[0][i] = za[1][i];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                uint32_t kk = 0;
                if (DROP) kk = (uint32_t)(kb[r] >> (ss * LT));
#pragma unroll
                for (int i = 0; i < LT; ++i) {
                    float prod = tt[0][r] * tt[i][r];
                    if (DROP) prod = keep_or_zero(prod, kk, i);
                    node_acc[i][r] += prod;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < LT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][r] += node_acc[i][r];
    }
    if (g_cur >= 0) {
#pragma unroll
        for (int i = 0; i < LT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (hq0 + r < h) k_partials[(size_t)(c + g_cur) * lh + i * h + hq0 + r] = acc[i][r];
    }
}

// ------------------------------------------------------------------------------------------------
// backward
//
// A workgroup of 8 wavefronts owns 16 streams and one block of 16 hidden graphs.  Per position step:
//   stage    the 16 x rows of every step i (fetched one step ahead into registers) go to LDS, with the node
//            and graph ids and, once per (node, hidden graph), the dropout keep bits;
//   phase A  wavefront w recomputes T_i = Z_i[ss] X_i^T for the hidden nodes ss = w, w+8 and turns them into
//            the gradient tiles G_i[ss] (w.r.t. T_i; G_0 also carries the zx factor), written to LDS as
//            [node][hidden graph];
//   phase B  wavefront w owns the (step i, 16-column chunk tc) items w, w+8, ...: dX_i[:, tc] = sum_ss
//            G_i[ss]^T Z_i[ss][:, tc] (stored straight from the accumulators) and dZ_i[ss][:, tc] += G_i[ss]
//            X_i[:, tc] (kept in registers over the whole stream), both operands of G read from LDS.
// Two barriers per step; x rows and keep bits are double-buffered.
// ------------------------------------------------------------------------------------------------
template <int LT, int NT, bool DROP, bool PERM>
__global__ void __launch_bounds__(kBwdBlock)
backward_kernel(const float *__restrict__ x, const float *__restrict__ z, const int *__restrict__ gid,
                const int *__restrict__ perm, const float *__restrict__ d_k, const float *__restrict__ counts_f32,
                int n, int d, int h, int s, int stream_len, int n_calls, float scale, int normalize, uint32_t t0,
                uint32_t t1, const unsigned long long *__restrict__ rng, int n_hb, float *__restrict__ dx_out,
                float *__restrict__ dz_partials) {
    constexpr int DPAD = 16 * NT, ZS = row_stride<NT>();
    constexpr int ITEMS = LT * NT, IPW = (ITEMS + kBwdWaves - 1) / kBwdWaves;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *zs = lds;                                          // [LT][s][16][ZS]
    float *xt = zs + LT * s * 16 * ZS;                        // [2][LT][16 nodes][ZS]
    float *gs = xt + 2 * LT * 16 * ZS;                        // [LT][s][16 nodes][kGS]
    unsigned long long *kbits = reinterpret_cast<unsigned long long *>(gs + LT * s * 16 * kGS);   // [2][256]
    int *t_node = reinterpret_cast<int *>(kbits + 2 * 256);   // [2][16]
    float *dks = reinterpret_cast<float *>(t_node + 32);      // [2][LT][16 hidden graphs][16 slots]: scaled dK

    int hb = 0;                                               // block of 16 hidden graphs being processed
    const int tid = threadIdx.x;

    const int lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, col = lane & 15;
    DropCfg cfg;
    if (DROP) cfg = load_cfg(rng, t0, t1);
    const int lh = LT * h;
    const size_t step_stride = (size_t)n * d;
    int hq0 = 4 * q;

    f32x4 dz[IPW][kSMax];
    // stream slot `slot` covers sorted positions first .. first + count - 1 (count = 0: no stream)
    auto slot_first = [&](int slot) -> int {
        const long long cs = (long long)(blockIdx.x * 16 + slot) * stream_len;
        return cs < n ? (int)cs : 0;
    };
    auto slot_count = [&](int slot) -> int {
        const long long cs = (long long)(blockIdx.x * 16 + slot) * stream_len;
        if (cs >= n) return 0;
        return (int)(cs + stream_len < n ? stream_len : n - cs);
    };

    constexpr int kChunks = LT * 16 * (DPAD / 4);             // 16-byte pieces of the x rows of one step
    constexpr int kLoads = (kChunks + kBwdBlock - 1) / kBwdBlock;
    // Everything fetched for step t+1 stays untouched in registers until the stage of step t+1: any arithmetic
    // on a loaded value (even a select) would make the wavefront wait for the load right where it was issued.
    float4 pre[kLoads];                                       // raw x pieces (rows at clamped positions)
    bool pre_ok[kLoads];
    int x_first[kLoads], x_count[kLoads];                     // stream of the row this thread fetches pieces of
    const float *x_base[kLoads];                              // x + step offset + piece offset
#pragma unroll
    for (int k = 0; k < kLoads; ++k) {
        const int idx = tid + k * kBwdBlock;
        const int ch = idx % (DPAD / 4);
        const int row = (idx / (DPAD / 4)) % 16;
        const int i = (idx / (DPAD / 4) / 16) % LT;
        x_first[k] = slot_first(row);
        x_count[k] = idx < kChunks ? slot_count(row) : 0;
        x_base[k] = x + i * step_stride + (4 * ch < d ? 4 * ch : d - 4);
    }
    // Thread tid < 256 serves the pair (slot tid & 15, hidden graph tid >> 4): node id for the keep bits and the
    // dK entries of the slot's graph.  Graph ids are fetched two steps ahead, so that the dK loads that depend
    // on them are issued without waiting.
    const int my_slot = tid & 15, my_hl = tid >> 4;
    int my_hq = my_hl;
    int my_hqc = my_hq < h ? my_hq : h - 1;
    int pre_node = 0;                                         // raw
    bool pre_node_ok = false;
    float pre_dk[LT], pre_cnt = 1.f;                          // raw
    bool pre_dk_ok = false;
    int nxt_gid = 0;                                          // raw graph id of my_slot one step after `pre`
    bool nxt_gid_ok = false;
    int gc_cur = 0;                                           // clamped graph id the next prefetch uses
    bool gc_ok = false;
    const int my_first = slot_first(my_slot), my_count = slot_count(my_slot);
    auto load_gid = [&](int t) {
        nxt_gid_ok = t < my_count;
        nxt_gid = gid[nxt_gid_ok ? my_first + t : 0];
    };
    auto prefetch = [&](int t) {
#pragma unroll
        for (int k = 0; k < kLoads; ++k) {
            pre_ok[k] = t < x_count[k];
            const int pc = pre_ok[k] ? x_first[k] + t : 0;
            const int node = PERM ? perm[pc] : pc;
            pre[k] = *reinterpret_cast<const float4 *>(x_base[k] + (size_t)node * d);
        }
        {
            pre_node_ok = t < my_count;
            const int pc = pre_node_ok ? my_first + t : 0;
            pre_node = PERM ? perm[pc] : pc;
            const int gc = gc_cur;                            // graph of my_slot at step t (fetched a step ago)
            pre_dk_ok = gc_ok && my_hq < h && tid < 256;
            if (normalize) pre_cnt = counts_f32[gc];
#pragma unroll
            for (int i = 0; i < LT; ++i) pre_dk[i] = d_k[(size_t)gc * lh + i * h + my_hqc];
            load_gid(t + 1);
        }
    };
    RWK_STAMP_DECL();
    // The blocks of 16 hidden graphs are taken one after the other by the same workgroup, so that their shares of
    // dX are added in block order by the lane that owns the element (no partial copies, no second launch).
    for (hb = 0; hb < n_hb; ++hb) {
    hq0 = 16 * hb + 4 * q;
    my_hq = 16 * hb + my_hl;
    my_hqc = my_hq < h ? my_hq : h - 1;
    __syncthreads();                                          // readers of the previous block's LDS contents are done
    stage_z<LT, NT>(z, zs, h, s, d, hb, kBwdBlock);
#pragma unroll
    for (int it = 0; it < IPW; ++it)
#pragma unroll
        for (int ss = 0; ss < kSMax; ++ss) dz[it][ss] = f32x4{0.f, 0.f, 0.f, 0.f};
    load_gid(0);
    gc_cur = nxt_gid_ok ? nxt_gid : 0;
    gc_ok = nxt_gid_ok;
    prefetch(0);
    gc_cur = nxt_gid_ok ? nxt_gid : 0;                        // graph at step 1, for the prefetch issued in step 0
    gc_ok = nxt_gid_ok;

    for (int t = 0; t < stream_len; ++t) {
        RWK_STAMP_AT(0);
        const int buf = t & 1;
        // stage step t
        float *xb = xt + buf * LT * 16 * ZS;
#pragma unroll
        for (int k = 0; k < kLoads; ++k) {
            const int idx = tid + k * kBwdBlock;
            if (idx < kChunks) {
                const int ch = idx % (DPAD / 4);
                const int row = (idx / (DPAD / 4)) % 16;
                const int i = idx / (DPAD / 4) / 16;
                const float4 v = pre[k];
                const bool ok = pre_ok[k];
                const bool keep = ok && 4 * ch < d;
                float4 o;
                o.x = keep ? v.x : 0.f;
                o.y = keep ? v.y : 0.f;
                o.z = keep ? v.z : 0.f;
                o.w = keep ? v.w : 0.f;
                *reinterpret_cast<float4 *>(xb + i * 16 * ZS + roff<NT>(row, 4 * ch)) = o;
            }
        }
        const int st_node = pre_node_ok ? pre_node : -1;
        if (tid < 16) t_node[buf * 16 + tid] = st_node;
        if (tid < 256) {
            const float f = normalize ? scale / pre_cnt : scale;
#pragma unroll
            for (int i = 0; i < LT; ++i)
                dks[((buf * LT + i) * 16 + my_hl) * 16 + my_slot] = pre_dk_ok ? pre_dk[i] * f : 0.f;
            if (DROP)
                kbits[buf * 256 + tid] = keep_bits(cfg, (uint32_t)(st_node >= 0 ? st_node : 0), (uint32_t)my_hq,
                                                   n_calls);
        }
        RWK_STAMP_AT(1);
        __syncthreads();                                      // step t staged; phase B of step t-1 finished
        RWK_STAMP_AT(2);
        if (t + 1 < stream_len) prefetch(t + 1);
        RWK_STAMP_AT(3);

        // ---------------- phase A ----------------
        if (wave < s) {                                       // this wavefront has hidden nodes to do
            float4 xv[LT][NT];                                // B of the recompute: X_i[node = col][16tc + 4q ..]
#pragma unroll
            for (int i = 0; i < LT; ++i)
#pragma unroll
                for (int tc = 0; tc < NT; ++tc)
                    xv[i][tc] = *reinterpret_cast<const float4 *>(xb + i * 16 * ZS + roff<NT>(col, 16 * tc + 4 * q));
            unsigned long long kb[4] = {0, 0, 0, 0};
            if (DROP) {
#pragma unroll
                for (int r = 0; r < 4; ++r) kb[r] = kbits[buf * 256 + (4 * q + r) * 16 + col];
            }
            float dp_cur[LT][4];                              // scaled dK of (node col's graph, hidden graph 4q + r)
#pragma unroll
            for (int i = 0; i < LT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) dp_cur[i][r] = dks[((buf * LT + i) * 16 + 4 * q + r) * 16 + col];
            for (int ss = wave; ss < s; ss += kBwdWaves) {
                // the LT tiles advance together, so that consecutive MFMAs never depend on each other
                f32x4 tt[LT];
#pragma unroll
                for (int i = 0; i < LT; ++i) tt[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tc = 0; tc < NT; ++tc) {
                    float4 a[LT];
#pragma unroll
                    for (int i = 0; i < LT; ++i)
                        a[i] = *reinterpret_cast<const float4 *>(zs + (i * s + ss) * 16 * ZS + roff<NT>(col, 16 * tc + 4 * q));
#pragma unroll
                    for (int i = 0; i < LT; ++i) tt[i] = RWK_MFMA(a[i].x, xv[i][tc].x, tt[i]);
#pragma unroll
                    for (int i = 0; i < LT; ++i) tt[i] = RWK_MFMA(a[i].y, xv[i][tc].y, tt[i]);
#pragma unroll
                    for (int i = 0; i < LT; ++i) tt[i] = RWK_MFMA(a[i].z, xv[i][tc].z, tt[i]);
#pragma unroll
                    for (int i = 0; i < LT; ++i) tt[i] = RWK_MFMA(a[i].w, xv[i][tc].w, tt[i]);
                }
                // gradient tiles: rows h = 4q + r, column = node; stored as [node][h]
                float gt[LT][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    uint32_t kk = 0;
                    if (DROP) kk = (uint32_t)(kb[r] >> (ss * LT));
                    float w[LT];
#pragma unroll
                    for (int i = 0; i < LT; ++i) {
                        w[i] = dp_cur[i][r];
                        if (DROP) w[i] = ((kk >> i) & 1u) ? w[i] : 0.f;
                    }
                    float a0 = 2.f * w[0] * tt[0][r];        // step 0 contributes keep * zx^2
#pragma unroll
                    for (int i = 1; i < LT; ++i) {
                        a0 = fmaf(w[i], tt[i][r], a0);
                        gt[i][r] = w[i] * tt[0][r];
                    }
                    gt[0][r] = a0;                            // step 0: everything flows through Z0, X0
                }
#pragma unroll
                for (int i = 0; i < LT; ++i)
                    *reinterpret_cast<float4 *>(gs + ((i * s + ss) * 16 + col) * kGS + 4 * q) =
                        make_float4(gt[i][0], gt[i][1], gt[i][2], gt[i][3]);
            }
        }
        RWK_STAMP_AT(4);
        __syncthreads();                                      // G tiles of step t complete
        RWK_STAMP_AT(5);
        // Take the graph id fetched after the first barrier out of its load register HERE, while only that
        // prefetch is outstanding: after the dX stores below, the in-order memory counter could tell it apart
        // from them only by waiting for every store.
        gc_cur = nxt_gid_ok ? nxt_gid : 0;
        gc_ok = nxt_gid_ok;
        asm volatile("" : "+v"(gc_cur));
        RWK_STAMP_AT(6);

        // ---------------- phase B ----------------
#pragma unroll
        for (int it = 0; it < IPW; ++it) {
            const int item = wave + it * kBwdWaves;
            if (item < ITEMS) {                               // wave-uniform
                const int i = item / NT, tc = item % NT;
                float xq[4];                                  // B of dZ: X_i[node = 4q + r][16tc + col]
#pragma unroll
                for (int r = 0; r < 4; ++r) xq[r] = xb[i * 16 * ZS + roff<NT>(4 * q + r, 16 * tc + col)];
                int nd[4];                                    // nodes of the dX rows this lane will store
#pragma unroll
                for (int r = 0; r < 4; ++r) nd[r] = t_node[buf * 16 + 4 * q + r];
                f32x4 dxa = {0.f, 0.f, 0.f, 0.f};
                // operands of hidden node ss: ga = G_i[ss][node = col][h = 4q..4q+3] (A of dX),
                // zb[r] = Z_i[ss][h = 4q + r][16tc + col] (B of dX), gb[r] = G_i[ss][h = col][node = 4q + r] (A of dZ)
                float4 ga;
                float zb[4], gb[4];
                auto load_ops = [&](int ss, float4 &oga, float (&ozb)[4], float (&ogb)[4]) {
                    const float *gtile = gs + (i * s + ss) * 16 * kGS;
                    const float *ztile = zs + (i * s + ss) * 16 * ZS;
                    oga = *reinterpret_cast<const float4 *>(gtile + col * kGS + 4 * q);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ozb[r] = ztile[roff<NT>(4 * q + r, 16 * tc + col)];
                        ogb[r] = gtile[(4 * q + r) * kGS + col];
                    }
                };
                load_ops(0, ga, zb, gb);
#pragma unroll
                for (int ss = 0; ss < kSMax; ++ss) {
                    if (ss < s) {                             // uniform
                        // the next hidden node's operands are requested before this one's MFMAs are issued
                        // (clamped index: the last request is a harmless repeat)
                        float4 nga;
                        float nzb[4], ngb[4];
                        load_ops(ss + 1 < s ? ss + 1 : s - 1, nga, nzb, ngb);
                        dxa = RWK_MFMA(ga.x, zb[0], dxa);
                        dz[it][ss] = RWK_MFMA(gb[0], xq[0], dz[it][ss]);
                        dxa = RWK_MFMA(ga.y, zb[1], dxa);
                        dz[it][ss] = RWK_MFMA(gb[1], xq[1], dz[it][ss]);
                        dxa = RWK_MFMA(ga.z, zb[2], dxa);
                        dz[it][ss] = RWK_MFMA(gb[2], xq[2], dz[it][ss]);
                        dxa = RWK_MFMA(ga.w, zb[3], dxa);
                        dz[it][ss] = RWK_MFMA(gb[3], xq[3], dz[it][ss]);
                        ga = nga;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            zb[r] = nzb[r];
                            gb[r] = ngb[r];
                        }
                    }
                }
                RWK_STAMP_AT(7);
                // the lane holds dX_i[node = 4q + r][16tc + col]
                const int k = 16 * tc + col;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (nd[r] >= 0 && k < d) {
                        // first block of hidden graphs: store; later blocks: add at the L2, in block order (the
                        // element's owner lane issues them one after the other)
                        float *dst = dx_out + (size_t)i * step_stride + (size_t)nd[r] * d + k;
                        if (hb > 0)
                            (void)__hip_atomic_fetch_add(dst, dxa[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else
                            *dst = dxa[r];
                    }
                }
            }
        }
    }

    RWK_STAMP_FLUSH();
    // dZ: the lane holds dZ_i[ss][h = 4q + r][16tc + col] of its items
    float *out = dz_partials + (size_t)blockIdx.x * LT * h * s * d;
#pragma unroll
    for (int it = 0; it < IPW; ++it) {
        const int item = wave + it * kBwdWaves;
        if (item >= ITEMS) continue;
        const int i = item / NT, tc = item % NT;
        const int k = 16 * tc + col;
#pragma unroll
        for (int ss = 0; ss < kSMax; ++ss) {
            if (ss >= s) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int hq = hq0 + r;
                if (hq < h && k < d) out[(((size_t)i * h + hq) * s + ss) * d + k] = dz[it][ss][r];
            }
        }
    }
    }   // hidden-graph blocks
}

// ------------------------------------------------------------------------------------------------
// backward, pipelined: the same phases, but phase A runs one position step ahead of phase B and the staging two
// steps ahead, so a step has ONE barrier and a wavefront that waits (for LDS, for its turn on the matrix pipe)
// leaves the pipe to wavefronts in the other phase:
//
//   interval t:   phase A(t+1) -> G tiles, buffer (t+1)&1      stage(t+2) -> x rows, buffer (t+2)%3; keep bits,
//                 phase B(t)   <- G tiles, buffer t&1                        dK and node ids of step t+2
//                 barrier;  global loads of step t+3 are issued (consumed by stage(t+3) an interval later)
//
// Costs LDS: three x-row buffers and two G buffers (G tiles without padding, see goff).  The staging of step t+2
// stands before phase B: its loads have had all of phase A to arrive, and phase B's dX stores are not yet in the in-order
// memory counter the wait for those loads looks at.
// ------------------------------------------------------------------------------------------------

template <int LT, int NT, bool DROP, bool PERM, bool KEEP>
__global__ void __launch_bounds__(kBwdBlock)
backward_pipe_kernel(const float *__restrict__ x, const float *__restrict__ z, const int *__restrict__ gid,
                     const int *__restrict__ perm, const float *__restrict__ d_k,
                     const float *__restrict__ counts_f32, int n, int d, int h, int s, int stream_len, int n_calls,
                     float scale, int normalize, uint32_t t0, uint32_t t1,
                     const unsigned long long *__restrict__ rng, int n_hb, float *__restrict__ dx_out,
                     float *__restrict__ dz_partials, const unsigned long long *__restrict__ keep_in) {
    constexpr int DPAD = 16 * NT, ZS = row_stride<NT>();
    constexpr int ITEMS = LT * NT, IPW = (ITEMS + kBwdWaves - 1) / kBwdWaves;
    constexpr int XB = LT * 16 * ZS;                          // floats of one x-row buffer
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *zs = lds;                                          // [LT][s][16][ZS]
    float *xt = zs + LT * s * 16 * ZS;                        // [3][LT][16 nodes][ZS]
    float *gs = xt + 3 * XB;                                  // [2][LT][s][16 rows][16 hidden graphs]
    unsigned long long *kbits = reinterpret_cast<unsigned long long *>(gs + 2 * LT * s * 256);    // [2][256]
    int *t_node = reinterpret_cast<int *>(kbits + 2 * 256);   // [3][16] (+ 16 unused)
    // scaled dK [2][LT][16][16 slots] and keep bits [2][16][16 slots]: hidden graph hl in row g_row(hl), so that
    // the reads of rows 4q + r by lane (q, col) spread over the banks
    float *dks = reinterpret_cast<float *>(t_node + 64);
    const int gs_buf = LT * s * 256;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, col = lane & 15;
    DropCfg cfg;
    if (DROP && !KEEP) cfg = load_cfg(rng, t0, t1);
    const int lh = LT * h;
    const size_t step_stride = (size_t)n * d;
    int hq0 = 4 * q;

    f32x4 dz[IPW][kSMax];
    auto slot_first = [&](int slot) -> int {
        const long long cs = (long long)(blockIdx.x * 16 + slot) * stream_len;
        return cs < n ? (int)cs : 0;
    };
    auto slot_count = [&](int slot) -> int {
        const long long cs = (long long)(blockIdx.x * 16 + slot) * stream_len;
        if (cs >= n) return 0;
        return (int)(cs + stream_len < n ? stream_len : n - cs);
    };

    constexpr int kChunks = LT * 16 * (DPAD / 4);             // 16-byte pieces of the x rows of one step
    constexpr int kLoads = (kChunks + kBwdBlock - 1) / kBwdBlock;
    // Everything fetched for a later step stays untouched in registers until that step is staged: any arithmetic
    // on a loaded value (even a select) would make the wavefront wait for the load right where it was issued.
    float4 pre[kLoads];
    bool pre_ok[kLoads];
    int x_first[kLoads], x_count[kLoads];
    const float *x_base[kLoads];
#pragma unroll
    for (int k = 0; k < kLoads; ++k) {
        const int idx = tid + k * kBwdBlock;
        const int ch = idx % (DPAD / 4);
        const int row = (idx / (DPAD / 4)) % 16;
        const int i = (idx / (DPAD / 4) / 16) % LT;
        x_first[k] = slot_first(row);
        x_count[k] = idx < kChunks ? slot_count(row) : 0;
        x_base[k] = x + i * step_stride + (4 * ch < d ? 4 * ch : d - 4);
    }
    // Thread tid < 256 serves the pair (slot tid & 15, hidden graph tid >> 4): node id for the keep bits and the
    // dK entries of the slot's graph.  Graph ids are fetched one step ahead of the dK loads that depend on them.
    const int my_slot = tid & 15, my_hl = tid >> 4;
    int my_hq = my_hl;
    int my_hqc = my_hq < h ? my_hq : h - 1;
    int pre_node = 0;
    bool pre_node_ok = false;
    unsigned long long pre_kb = 0;
    float pre_dk[LT], pre_cnt = 1.f;
    bool pre_dk_ok = false;
    int nxt_gid = 0;
    bool nxt_gid_ok = false;
    int gc_cur = 0;
    bool gc_ok = false;
    const int my_first = slot_first(my_slot), my_count = slot_count(my_slot);
    // The keep bits of the 256 (slot, hidden graph) pairs of a step are drawn by four wavefronts, 64 pairs each
    // (pair p: slot p & 15 = this thread's own slot, hidden graph p >> 4).  With more than 8 hidden nodes,
    // wavefronts 0 and 1 carry two of them in phase A, i.e. SIMDs 0 and 1 three and SIMDs 2 and 3 two: the
    // generator (about the cost of one hidden node's MFMAs per wavefront) then goes to wavefronts 2, 3, 6, 7, which
    // evens out the SIMDs; otherwise one wavefront per SIMD.
    int kb_pair = -1;
    if (s > kBwdWaves) {
        if (wave & 2) kb_pair = 64 * ((wave & 1) + 2 * (wave >> 2)) + lane;
    } else if (wave < 4) {
        kb_pair = tid;
    }
    int hb_cur = 0;
    auto load_gid = [&](int t) {
        nxt_gid_ok = t < my_count;
        nxt_gid = gid[nxt_gid_ok ? my_first + t : 0];
    };
    auto prefetch = [&](int t) {
#pragma unroll
        for (int k = 0; k < kLoads; ++k) {
            pre_ok[k] = t < x_count[k];
            const int pc = pre_ok[k] ? x_first[k] + t : 0;
            const int node = PERM ? perm[pc] : pc;
            pre[k] = *reinterpret_cast<const float4 *>(x_base[k] + (size_t)node * d);
        }
        pre_node_ok = t < my_count;
        const int pc = pre_node_ok ? my_first + t : 0;
        pre_node = PERM ? perm[pc] : pc;
        // keep bits the forward pass left (pair (my_slot, my_hl), threads 0..255)
        if (KEEP && tid < 256) {
            const size_t idx = (size_t)pre_node * h + my_hqc;
            if (s * LT <= 32) pre_kb = reinterpret_cast<const uint32_t *>(keep_in)[idx];
            else pre_kb = keep_in[idx];
        }
        const int gc = gc_cur;                                // graph of my_slot at step t (fetched a step ago)
        pre_dk_ok = gc_ok && my_hq < h && tid < 256;
        if (normalize) pre_cnt = counts_f32[gc];
#pragma unroll
        for (int i = 0; i < LT; ++i) pre_dk[i] = d_k[(size_t)gc * lh + i * h + my_hqc];
        load_gid(t + 1);
    };
    // what prefetch(t) fetched goes to LDS: x rows -> xb, node ids -> tn[16], keep bits and scaled dK -> parity par
    int kb_node = 0;                                          // node of this thread's pair in the step last staged
    auto draw_bits = [&](int par) {
        if (DROP && kb_pair >= 0)
            kbits[par * 256 + g_row(kb_pair >> 4) * 16 + (kb_pair & 15)] =
                keep_bits(cfg, (uint32_t)kb_node, (uint32_t)(16 * hb_cur + (kb_pair >> 4)), n_calls);
    };
    auto stage = [&](float *xb, int *tn, int par, bool draw) {
#pragma unroll
        for (int k = 0; k < kLoads; ++k) {
            const int idx = tid + k * kBwdBlock;
            if (idx < kChunks) {
                const int ch = idx % (DPAD / 4);
                const int row = (idx / (DPAD / 4)) % 16;
                const int i = idx / (DPAD / 4) / 16;
                const float4 v = pre[k];
                const bool keep = pre_ok[k] && 4 * ch < d;
                float4 o;
                o.x = keep ? v.x : 0.f;
                o.y = keep ? v.y : 0.f;
                o.z = keep ? v.z : 0.f;
                o.w = keep ? v.w : 0.f;
                *reinterpret_cast<float4 *>(xb + i * 16 * ZS + roff<NT>(row, 4 * ch)) = o;
            }
        }
        const int st_node = pre_node_ok ? pre_node : -1;
        if (tid < 16) tn[tid] = st_node;
        if (tid < 256) {
            const float f = normalize ? scale / pre_cnt : scale;
#pragma unroll
            for (int i = 0; i < LT; ++i)
                dks[((par * LT + i) * 16 + g_row(my_hl)) * 16 + my_slot] = pre_dk_ok ? pre_dk[i] * f : 0.f;
        }
        kb_node = st_node >= 0 ? st_node : 0;
        if (KEEP) {
            if (tid < 256) kbits[par * 256 + g_row(my_hl) * 16 + my_slot] = pre_node_ok && my_hq < h ? pre_kb : 0ull;
        } else if (draw) {
            draw_bits(par);
        }
        // the graph id that came with this fetch is the one the next prefetch's dK loads use
        gc_cur = nxt_gid_ok ? nxt_gid : 0;
        gc_ok = nxt_gid_ok;
        asm volatile("" : "+v"(gc_cur));
    };
    // phase A of one step: G tiles of the hidden nodes wave, wave + 8, ... from the x rows in xb -> gbuf.
    // The Z operands of a group of 4 LT MFMAs are requested one group ahead (two register sets, chosen by the
    // group's parity at compile time), so that their LDS round trip passes under the previous group's MFMAs.
    auto phase_a = [&](const float *xb, int par, float *gbuf) {
        if (wave >= s) return;
        // Requests in the order of first use (LDS answers in order, and all eight wavefronts start the phase
        // together): the first group's Z operands, then the x rows column group by column group.
        float4 za[2][LT];
        auto load_za = [&](int ss, int tc, float4 (&dst)[LT]) {
#pragma unroll
            for (int i = 0; i < LT; ++i)
                dst[i] = *reinterpret_cast<const float4 *>(zs + (i * s + ss) * 16 * ZS + roff<NT>(col, 16 * tc + 4 * q));
        };
        load_za(wave, 0, za[0]);
        float4 xv[LT][NT];                                    // B of the recompute: X_i[node = col][16tc + 4q ..]
#pragma unroll
        for (int tc = 0; tc < NT; ++tc)
#pragma unroll
            for (int i = 0; i < LT; ++i)
                xv[i][tc] = *reinterpret_cast<const float4 *>(xb + i * 16 * ZS + roff<NT>(col, 16 * tc + 4 * q));
        __builtin_amdgcn_sched_barrier(0);                    // keep that order: the requests below are needed later
        unsigned long long kb[4] = {0, 0, 0, 0};
        if (DROP) {
#pragma unroll
            for (int r = 0; r < 4; ++r) kb[r] = kbits[par * 256 + (4 * r + q) * 16 + col];     // row g_row(4q + r)
        }
        float dp_cur[LT][4];                                  // scaled dK of (node col's graph, hidden graph 4q + r)
#pragma unroll
        for (int i = 0; i < LT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) dp_cur[i][r] = dks[((par * LT + i) * 16 + 4 * r + q) * 16 + col];
        for (int ss = wave; ss < s; ss += kBwdWaves) {
            const int ss_next = ss + kBwdWaves < s ? ss + kBwdWaves : ss;       // last one: a harmless repeat
            f32x4 tt[LT];
#pragma unroll
            for (int i = 0; i < LT; ++i) tt[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tc = 0; tc < NT; ++tc) {
                constexpr bool kEven = (NT & 1) == 0;         // then group tc of every hidden node has parity tc & 1
                const int cur = kEven ? (tc & 1) : 0;
                float4 (&a)[LT] = za[cur];
                if (tc + 1 < NT) load_za(ss, tc + 1, za[cur ^ 1]);
                else load_za(ss_next, 0, za[cur ^ 1]);
#pragma unroll
                for (int i = 0; i < LT; ++i) tt[i] = RWK_MFMA(a[i].x, xv[i][tc].x, tt[i]);
#pragma unroll
                for (int i = 0; i < LT; ++i) tt[i] = RWK_MFMA(a[i].y, xv[i][tc].y, tt[i]);
#pragma unroll
                for (int i = 0; i < LT; ++i) tt[i] = RWK_MFMA(a[i].z, xv[i][tc].z, tt[i]);
#pragma unroll
                for (int i = 0; i < LT; ++i) tt[i] = RWK_MFMA(a[i].w, xv[i][tc].w, tt[i]);
                if (!kEven) {
#pragma unroll
                    for (int i = 0; i < LT; ++i) za[0][i] = za[1][i];
                }
            }
            float gt[LT][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                uint32_t kk = 0;
                if (DROP) kk = (uint32_t)(kb[r] >> (ss * LT));
                float w[LT];
#pragma unroll
                for (int i = 0; i < LT; ++i) {
                    w[i] = dp_cur[i][r];
                    if (DROP) w[i] = keep_or_zero(w[i], kk, i);
                }
                float a0 = 2.f * w[0] * tt[0][r];            // step 0 contributes keep * zx^2
#pragma unroll
                for (int i = 1; i < LT; ++i) {
                    a0 = fmaf(w[i], tt[i][r], a0);
                    gt[i][r] = w[i] * tt[0][r];
                }
                gt[0][r] = a0;
            }
#pragma unroll
            for (int i = 0; i < LT; ++i)
                *reinterpret_cast<float4 *>(gbuf + (i * s + ss) * 256 + goff(col, 4 * q)) =
                    make_float4(gt[i][0], gt[i][1], gt[i][2], gt[i][3]);
        }
    };
    // phase B of one step: dX rows stored, dZ accumulated
    auto phase_b = [&](const float *xb, const int *tn, const float *gbuf, int hb) {
#pragma unroll
        for (int it = 0; it < IPW; ++it) {
            const int item = wave + it * kBwdWaves;
            if (item < ITEMS) {                               // wave-uniform
                const int i = item / NT, tc = item % NT;
                float xq[4];                                  // B of dZ: X_i[node = 4q + r][16tc + col]
#pragma unroll
                for (int r = 0; r < 4; ++r) xq[r] = xb[i * 16 * ZS + roff<NT>(4 * q + r, 16 * tc + col)];
                int nd[4];                                    // nodes of the dX rows this lane will store
#pragma unroll
                for (int r = 0; r < 4; ++r) nd[r] = tn[4 * q + r];
                f32x4 dxa = {0.f, 0.f, 0.f, 0.f};
                // operands of hidden node ss: ga = G_i[ss][node = col][h = 4q..4q+3] (A of dX),
                // zb[r] = Z_i[ss][h = 4q + r][16tc + col] (B of dX), gb[r] = G_i[ss][node = 4q + r][h = col] (A of dZ)
                // Two operand sets, chosen by the hidden node's parity at compile time: the set of hidden node
                // ss + 1 is requested before the MFMAs of hidden node ss are issued and first touched after them.
                struct Ops {
                    float4 ga;
                    float zb[4], gb[4];
                };
                Ops ops[2];
                auto load_ops = [&](int ss, Ops &o) {
                    const float *gtile = gbuf + (i * s + ss) * 256;
                    const float *ztile = zs + (i * s + ss) * 16 * ZS;
                    o.ga = *reinterpret_cast<const float4 *>(gtile + goff(col, 4 * q));
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        o.zb[r] = ztile[roff<NT>(4 * q + r, 16 * tc + col)];
                        o.gb[r] = gtile[goff(4 * q + r, col)];
                    }
                };
                load_ops(0, ops[0]);
#pragma unroll
                for (int ss = 0; ss < kSMax; ++ss) {
                    if (ss < s) {                             // uniform
                        // after the last hidden node this reads what follows the tile in LDS: unused
                        load_ops(ss + 1, ops[(ss + 1) & 1]);
                        const Ops &o = ops[ss & 1];
                        dxa = RWK_MFMA(o.ga.x, o.zb[0], dxa);
                        dz[it][ss] = RWK_MFMA(o.gb[0], xq[0], dz[it][ss]);
                        dxa = RWK_MFMA(o.ga.y, o.zb[1], dxa);
                        dz[it][ss] = RWK_MFMA(o.gb[1], xq[1], dz[it][ss]);
                        dxa = RWK_MFMA(o.ga.z, o.zb[2], dxa);
                        dz[it][ss] = RWK_MFMA(o.gb[2], xq[2], dz[it][ss]);
                        dxa = RWK_MFMA(o.ga.w, o.zb[3], dxa);
                        dz[it][ss] = RWK_MFMA(o.gb[3], xq[3], dz[it][ss]);
                        // scheduling: one or two of the next hidden node's nine operand requests after each MFMA
                        // (all nine in a row after the first MFMA delayed the second one)
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                const int k = 16 * tc + col;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (nd[r] >= 0 && k < d) {
                        // The first block of hidden graphs stores its share of dX, the later ones add theirs at
                        // the L2 (an add without return value: nothing to wait for, where a load of the sum so far
                        // cost a memory round trip per step).  Every element is owned by one lane, which issues
                        // its store and its adds in block order, so the sum is formed in a fixed order.
                        float *dst = dx_out + (size_t)i * step_stride + (size_t)nd[r] * d + k;
                        if (hb > 0)
                            (void)__hip_atomic_fetch_add(dst, dxa[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else
                            *dst = dxa[r];
                    }
                }
            }
        }
    };

    RWK_PSTAMP_DECL();
    // With more than 8 hidden nodes the keep bits are drawn by both wavefronts of SIMDs 2 and 3.  Drawn at the same
    // place of the step they would share the vector ALU while the matrix pipe idles (and the younger wavefront
    // gets its turn only when the older one is through): wavefronts 6 and 7 draw theirs after phase B instead, so
    // that on each of the two SIMDs one wavefront issues MFMAs while the other runs the generator.
    const bool late_bits = DROP && !KEEP && s > kBwdWaves && __builtin_amdgcn_readfirstlane(wave) >= 6;
    // The blocks of 16 hidden graphs are taken one after the other by the same workgroup, so that their shares of
    // dX are added in block order by the lane that owns the element (no partial copies, no second launch).
    for (int hb = 0; hb < n_hb; ++hb) {
        hq0 = 16 * hb + 4 * q;
        hb_cur = hb;
        my_hq = 16 * hb + my_hl;
        my_hqc = my_hq < h ? my_hq : h - 1;
        __syncthreads();                                      // readers of the previous block's LDS contents are done
        stage_z<LT, NT>(z, zs, h, s, d, hb, kBwdBlock);
#pragma unroll
        for (int it = 0; it < IPW; ++it)
#pragma unroll
            for (int ss = 0; ss < kSMax; ++ss) dz[it][ss] = f32x4{0.f, 0.f, 0.f, 0.f};
        load_gid(0);
        gc_cur = nxt_gid_ok ? nxt_gid : 0;
        gc_ok = nxt_gid_ok;
        prefetch(0);
        // x-row buffers in rotation: xa holds the step phase A works on, xb the one phase B works on, xs is filled
        float *xs_ = xt, *xa_ = xt + XB, *xb_ = xt + 2 * XB;
        int *ns_ = t_node, *na_ = t_node + 16, *nb_ = t_node + 32;
        auto rotate = [&]() {
            float *tx = xb_; xb_ = xa_; xa_ = xs_; xs_ = tx;
            int *tn = nb_; nb_ = na_; na_ = ns_; ns_ = tn;
        };
        stage(xs_, ns_, 0, true);                             // step 0
        rotate();                                             // step 0 is phase A's
        if (1 < stream_len) prefetch(1);
        __syncthreads();                                      // Z rows and step 0 staged
        phase_a(xa_, 0, gs);
        if (1 < stream_len) stage(xs_, ns_, 1, true);         // step 1
        rotate();                                             // step 0 is phase B's, step 1 phase A's
        __syncthreads();
        if (2 < stream_len) prefetch(2);
        RWK_PSTAMP(5);
        for (int t = 0; t < stream_len; ++t) {
            const int par = t & 1;
            if (t + 1 < stream_len) phase_a(xa_, par ^ 1, gs + (par ^ 1) * gs_buf);
            RWK_PSTAMP(0);
            if (t + 2 < stream_len) stage(xs_, ns_, par, !late_bits);
            RWK_PSTAMP(1);
            phase_b(xb_, nb_, gs + par * gs_buf, hb);
            if (late_bits && t + 2 < stream_len) draw_bits(par);
            RWK_PSTAMP(2);
            rotate();
            __syncthreads();
            RWK_PSTAMP(3);
            if (t + 3 < stream_len) prefetch(t + 3);
            RWK_PSTAMP(4);
        }

        // dZ: the lane holds dZ_i[ss][h = 4q + r][16tc + col] of its items
        float *out = dz_partials + (size_t)blockIdx.x * LT * h * s * d;
#pragma unroll
        for (int it = 0; it < IPW; ++it) {
            const int item = wave + it * kBwdWaves;
            if (item >= ITEMS) continue;
            const int i = item / NT, tc = item % NT;
            const int k = 16 * tc + col;
#pragma unroll
            for (int ss = 0; ss < kSMax; ++ss) {
                if (ss >= s) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int hq = hq0 + r;
                    if (hq < h && k < d) out[(((size_t)i * h + hq) * s + ss) * d + k] = dz[it][ss][r];
                }
            }
        }
        RWK_PSTAMP(5);
    }   // hidden-graph blocks
    RWK_PSTAMP_FLUSH();
}

inline int nt_of(int d) { return (d + 15) / 16; }

inline int row_stride_of(int nt) { return nt == 4 ? 64 : 16 * nt + 4; }
inline size_t fwd_lds(int l, int s, int nt) { return (size_t)l * s * 16 * row_stride_of(nt) * sizeof(float); }
inline size_t bwd_lds(int l, int s, int nt) {
    const int zs = row_stride_of(nt);
    return ((size_t)l * s * 16 * zs + 2 * (size_t)l * 16 * zs + (size_t)l * s * 16 * kGS) * sizeof(float) +
           2 * 256 * sizeof(unsigned long long) + 32 * sizeof(int) + 2 * (size_t)l * 256 * sizeof(float);
}

template <int LT, int NT>
int launch_fwd(const Args &a) {
    const int hb = (a.h + 15) / 16;
    const int n_streams = (int)rwk_ceil_div(a.n, a.stream_len);
    const dim3 grid((unsigned)rwk_ceil_div(n_streams, 16 * (kFwdBlock / 64)), (unsigned)hb);
    const size_t lds = fwd_lds(LT, a.s, NT);
    const int n_calls = (a.s * a.max_step + 14) / 15;
#define RWK_F(DROPV)                                                                                            \
    {                                                                                                           \
        auto kern = forward_kernel<LT, NT, DROPV>;                                                              \
        if (lds > 64 * 1024) {                                                                                  \
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                               (int)lds);                                                       \
            if (e != hipSuccess) return (int)e;                                                                 \
        }                                                                                                       \
        hipLaunchKernelGGL(kern, grid, dim3(kFwdBlock), lds, a.stream, a.x, a.z, a.gid, a.perm, a.n, a.d, a.h, a.s, \
                           a.stream_len, n_calls, a.t0, a.t1, a.rng, a.k_partials, a.keep);                     \
    }
    if (a.drop) RWK_F(true) else RWK_F(false)
#undef RWK_F
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

inline size_t bwd_pipe_lds(int l, int s, int nt) {
    const int zs = row_stride_of(nt);
    return ((size_t)l * s * 16 * zs + 3 * (size_t)l * 16 * zs + 2 * (size_t)l * s * 256) * sizeof(float) +
           2 * 256 * sizeof(unsigned long long) + 64 * sizeof(int) + 2 * (size_t)l * 256 * sizeof(float);
}
// the pipelined kernel when its buffers fit the 160 KB of LDS (with a little room left for the runtime)
inline bool pipelined(int l, int s, int nt) { return bwd_pipe_lds(l, s, nt) + 256 <= 160 * 1024; }

template <int LT, int NT, bool DROPV, bool KEEP>
int launch_pipe(const Args &a, dim3 grid, size_t lds, int n_calls, int hb) {
    auto kern = a.perm ? backward_pipe_kernel<LT, NT, DROPV, true, KEEP> : backward_pipe_kernel<LT, NT, DROPV, false, KEEP>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(kBwdBlock), lds, a.stream, a.x, a.z, a.gid, a.perm, a.d_k, a.counts_f32, a.n,
                       a.d, a.h, a.s, a.stream_len, n_calls, a.scale, a.normalize, a.t0, a.t1, a.rng, hb, a.dx,
                       a.dz_partials, (const unsigned long long *)a.keep);
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

template <int LT, int NT>
int launch_bwd(const Args &a) {
    const int hb = (a.h + 15) / 16;
    const dim3 grid((unsigned)backward_blocks(a.n, a.stream_len), 1u);
    const bool pipe = pipelined(LT, a.s, NT);
    const size_t lds = pipe ? bwd_pipe_lds(LT, a.s, NT) : bwd_lds(LT, a.s, NT);
    const int n_calls = (a.s * a.max_step + 14) / 15;
    float *dx_target = a.dx;
#define RWK_B(DROPV)                                                                                            \
    {                                                                                                           \
        if (pipe) {                                                                                             \
            return (DROPV && a.keep != nullptr) ? launch_pipe<LT, NT, DROPV, DROPV>(a, grid, lds, n_calls, hb)  \
                                                : launch_pipe<LT, NT, DROPV, false>(a, grid, lds, n_calls, hb); \
        } else {                                                                                                \
            auto kern = a.perm ? backward_kernel<LT, NT, DROPV, true> : backward_kernel<LT, NT, DROPV, false>;  \
            if (lds > 64 * 1024) {                                                                              \
                hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                   (int)lds);                                                   \
                if (e != hipSuccess) return (int)e;                                                             \
            }                                                                                                   \
            hipLaunchKernelGGL(kern, grid, dim3(kBwdBlock), lds, a.stream, a.x, a.z, a.gid, a.perm, a.d_k,      \
                               a.counts_f32, a.n, a.d, a.h, a.s, a.stream_len, n_calls, a.scale, a.normalize, a.t0, \
                               a.t1, a.rng, hb, dx_target, a.dz_partials);                                      \
        }                                                                                                       \
    }
    if (a.drop) RWK_B(true) else RWK_B(false)
#undef RWK_B
    RWK_LAUNCH_CHECK();
    return RWK_OK;
}

template <int LT>
int by_nt(const Args &a, bool bwd) {
    switch (nt_of(a.d)) {
        case 1: return bwd ? launch_bwd<LT, 1>(a) : launch_fwd<LT, 1>(a);
        case 2: return bwd ? launch_bwd<LT, 2>(a) : launch_fwd<LT, 2>(a);
        case 3:
        case 4: return bwd ? launch_bwd<LT, 4>(a) : launch_fwd<LT, 4>(a);
    }
    return RWK_E_UNSUPPORTED;
}

int by_l(const Args &a, bool bwd) {
    switch (a.max_step) {
        case 1: return by_nt<1>(a, bwd);
        case 2: return by_nt<2>(a, bwd);
        case 3: return by_nt<3>(a, bwd);
        case 4: return by_nt<4>(a, bwd);
    }
    return RWK_E_UNSUPPORTED;
}

}  // namespace

bool supported(int64_t d, int64_t h, int64_t s, int64_t max_step) {
    if (d <= 4 || d > 64 || (d & 3) != 0 || h < 1 || h > 64 || max_step < 1 || max_step > 4) return false;
    if (s < 1 || s > kSMax || s * max_step > 45) return false;
    int nt = nt_of((int)d);
    if (nt == 3) nt = 4;
    return bwd_lds((int)max_step, (int)s, nt) + 256 <= kLdsBudget;
}

int64_t backward_blocks(int64_t n, int64_t stream_len) { return rwk_ceil_div(rwk_ceil_div(n, stream_len), 16); }

int64_t keep_words(int64_t n, int64_t d, int64_t h, int64_t s, int64_t max_step) {
    if (!supported(d, h, s, max_step)) return 0;
    int nt = nt_of((int)d);
    if (nt == 3) nt = 4;
    if (!pipelined((int)max_step, (int)s, nt)) return 0;      // the unpipelined backward draws its own bits
    return s * max_step <= 32 ? (n * h + 1) / 2 : n * h;      // one 32-bit or one 64-bit word per pair
}

int64_t dx_partials_size(int64_t n, int64_t d, int64_t h, int64_t max_step) {
    (void)n; (void)d; (void)h; (void)max_step;
    return 0;                                                 // the blocks of hidden graphs are added in place
}

// Forward workgroups hold 128 streams for one block of 16 hidden graphs, one workgroup per CU (the Z slab
// takes most of the LDS): aim at m full rounds of 256 workgroups with about 16 positions per stream, and up
// to 64 in very large batches, so that the per-workgroup dZ partial rows stay small next to the X rows read.
// Backward workgroups hold 16 streams for all blocks of hidden graphs and pay a fixed price per block (Z rows to
// LDS, pipeline fill, the dZ partial rows): m full rounds of 256 workgroups with 24 to 48 positions per stream.
int64_t stream_len(int64_t n, int64_t h, bool bwd) {
    if (bwd) {
        const int64_t per_round = 16 * 256;
        int64_t m = (n + per_round * 16) / (per_round * 32);
        if (m < 1) m = 1;
        const int64_t len = rwk_ceil_div(n, per_round * m);
        return len < 4 ? 4 : len;
    }
    const int64_t hb = (h + 15) / 16;
    const int64_t per_round = 128 * 256 / hb;
    int64_t target = n / (per_round * 4);
    target = target < 16 ? 16 : (target > 64 ? 64 : target);
    int64_t m = (n + per_round * target / 2) / (per_round * target);
    if (m < 1) m = 1;
    const int64_t len = rwk_ceil_div(n, per_round * m);
    return len < 4 ? 4 : len;
}

#ifdef RWK_STAMP
extern "C" int rwk_debug_read_stamps(unsigned long long *host64) {
    return (int)hipMemcpyFromSymbol(host64, HIP_SYMBOL(rwk_stamp_buf), sizeof(unsigned long long) * 64);
}
#endif

int forward(const Args &a) { return by_l(a, false); }
int backward(const Args &a) { return by_l(a, true); }

}  // namespace rwk_mfma
