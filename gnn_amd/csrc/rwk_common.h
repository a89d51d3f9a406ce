// Shared device/host helpers for librwk (gfx950 only: 64-lane wavefronts, DPP, LDS).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rwk.h"

#define RWK_WAVE 64

#define RWK_REQUIRE_PTR(p) \
    do {                   \
        if ((p) == nullptr) return RWK_E_NULL; \
    } while (0)

#define RWK_LAUNCH_CHECK()                       \
    do {                                         \
        hipError_t e__ = hipGetLastError();      \
        if (e__ != hipSuccess) return (int)e__;  \
    } while (0)

static inline int64_t rwk_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------
// Philox4x32-R (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3").  The paper's default is
// ten rounds; seven is the smallest round count it reports as passing the whole BigCrush battery.  The head's
// dropout (G x P samples per step) uses ten.  The random-walk layer's mask - one call per (node, hidden graph) and
// step of training, a third of the vector instructions of the fused kernels at ten rounds - uses seven
// (RWK_MASK_ROUNDS); tests/test_oracle_golden.py holds the known-answer vectors of both and the statistics of the
// resulting keep mask (rates and pairwise correlations within five standard deviations on 10^7 samples).
// ---------------------------------------------------------------------------------------
#define RWK_MASK_ROUNDS 7

struct RwkPhilox4 {
    uint32_t x, y, z, w;
};

// a ^ b ^ c: one v_bitop3_b32 on the device (the compiler leaves the round's two-step XOR as two instructions)
__host__ __device__ static inline uint32_t rwk_xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
}

template <int ROUNDS>
__host__ __device__ static inline RwkPhilox4 rwk_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                            uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = rwk_xor3((uint32_t)(p1 >> 32), c1, k0);
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = rwk_xor3((uint32_t)(p0 >> 32), c3, k1);
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return RwkPhilox4{c0, c1, c2, c3};
}

__host__ __device__ static inline RwkPhilox4 rwk_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                                               uint32_t c3, uint32_t k0, uint32_t k1) {
    return rwk_philox4x32<10>(c0, c1, c2, c3, k0, k1);
}

// the random-walk layer's mask generator
__host__ __device__ static inline RwkPhilox4 rwk_philox_mask(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                             uint32_t k0, uint32_t k1) {
    return rwk_philox4x32<RWK_MASK_ROUNDS>(c0, c1, c2, c3, k0, k1);
}

// Dropout keep-mask specification (shared by forward, backward, the mask dump and the oracle):
//   one Philox stream per (node n, hidden graph h); element (step i, hidden node s) is sample
//   e = s*L + i of that stream; sample e is byte (e % 15) of call (e / 15), where a call is
//   philox4x32 with RWK_MASK_ROUNDS rounds (counter = {n, h, call, offset_lo}, key = {seed_lo, seed_hi ^ offset_hi}) and byte j
//   is bits [8*(j&3), 8*(j&3)+8) of output word j>>2.  Byte 15 of the call breaks ties:
//   keep <=> b < T0 || (b == T0 && byte15 < T1), with T0 = floor(q*256), T1 = floor((q*256-T0)*256),
//   q = 1 - p: the keep probability is q rounded down to a multiple of 2^-16.
struct RwkDropCfg {
    uint32_t t0;      // 0..256
    uint32_t t1;      // 0..255
    uint32_t seed_lo, key_hi, off_lo;
};

static inline void rwk_drop_thresholds(float drop_p, uint32_t *t0, uint32_t *t1) {
    double q = 1.0 - (double)drop_p;
    if (q < 0.0) q = 0.0;
    if (q > 1.0) q = 1.0;
    double a = q * 256.0;
    uint32_t a0 = (uint32_t)a;
    if (a0 > 256u) a0 = 256u;
    double rem = (a - (double)a0) * 256.0;
    uint32_t a1 = (uint32_t)rem;
    if (a1 > 255u) a1 = 255u;
    *t0 = a0;
    *t1 = a1;
}

// 15 keep bits (bit j <-> byte j) of one call.
__device__ static inline uint32_t rwk_keep_bits15(const RwkDropCfg &cfg, uint32_t n, uint32_t h, uint32_t call) {
    const RwkPhilox4 r = rwk_philox_mask(n, h, call, cfg.off_lo, cfg.seed_lo, cfg.key_hi);
    const uint32_t tie = (r.w >> 24) < cfg.t1 ? 1u : 0u;
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
    uint32_t bits = 0;
#pragma unroll
    for (int j = 0; j < 15; ++j) {
        const uint32_t b = (w[j >> 2] >> (8 * (j & 3))) & 0xFFu;
        const uint32_t k = (b < cfg.t0) ? 1u : ((b == cfg.t0) ? tie : 0u);
        bits |= k << j;
    }
    return bits;
}

// ---------------------------------------------------------------------------------------
// Sum over aligned groups of `width` consecutive lanes (width a power of two <= 64); every
// lane of the group receives the total.  Fixed butterfly order => bitwise reproducible.
// Lane distances 1 and 2 use quad-permute DPP; 4 and 8 use the half-row / row mirror DPP modes
// (after the quad steps every quad is uniform, so a mirror pairs each quad with its partner);
// 16 and 32 go through a cross-row permute.
// ---------------------------------------------------------------------------------------
template <int CTRL>
__device__ static inline float rwk_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

// Sums of V = 8 or 16 per-lane values over the 16 lanes of a DPP row at about a third of the cost of V
// butterflies: at every step a lane keeps one half of its values and adds the same half received from its
// partner (15-l, then 7-l within 8 lanes, then 3-l within 4, then l^1), so the work halves each time.  Lane l
// (0..15 within the row) ends with the total of value number rwk_halving_index<V>(l); for V = 8 both lanes of a
// pair hold it.  Fixed order => bitwise reproducible.
template <int V>
__device__ static inline int rwk_halving_index(int l) {
    const int a = l >= 8, b = (l & 7) >= 4, c = (l & 3) >= 2, d = l & 1;
    return V == 16 ? 8 * a + 4 * b + 2 * c + d : 4 * a + 2 * b + c;
}

template <int V>
__device__ static inline float rwk_halving_sum16(const float (&v)[V], int l) {
    static_assert(V == 8 || V == 16, "8 or 16 values");
    const bool up_a = l >= 8, up_b = (l & 7) >= 4, up_c = (l & 3) >= 2, up_d = (l & 1) != 0;
    float a[V / 2], b[V / 4], c[V / 8];
#pragma unroll
    for (int m = 0; m < V / 2; ++m) {
        const float keep = up_a ? v[m + V / 2] : v[m], give = up_a ? v[m] : v[m + V / 2];
        a[m] = keep + rwk_dpp<0x140>(give);           // row_mirror
    }
#pragma unroll
    for (int m = 0; m < V / 4; ++m) {
        const float keep = up_b ? a[m + V / 4] : a[m], give = up_b ? a[m] : a[m + V / 4];
        b[m] = keep + rwk_dpp<0x141>(give);           // row_half_mirror
    }
#pragma unroll
    for (int m = 0; m < V / 8; ++m) {
        const float keep = up_c ? b[m + V / 8] : b[m], give = up_c ? b[m] : b[m + V / 8];
        c[m] = keep + rwk_dpp<0x1B>(give);            // quad_perm [3,2,1,0]
    }
    if (V == 16) {
        const float keep = up_d ? c[1] : c[0], give = up_d ? c[0] : c[1];
        return keep + rwk_dpp<0xB1>(give);            // quad_perm [1,0,3,2]
    }
    return c[0] + rwk_dpp<0xB1>(c[0]);
}

__device__ static inline float rwk_group_sum(float v, int width) {
    if (width >= 2) v += rwk_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
    if (width >= 4) v += rwk_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
    if (width >= 8) v += rwk_dpp<0x141>(v);  // row_half_mirror: lane i <-> 7-i within 8 lanes
    if (width >= 16) v += rwk_dpp<0x140>(v); // row_mirror:      lane i <-> 15-i within 16 lanes
    if (width >= 32) v += __shfl_xor(v, 16, 64);
    if (width >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

// Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Every kernel of the step that walks the
// nodes gives XCD k the k-th contiguous eighth of them (the fused kernels by their streams, the per-graph launches by
// their graphs, the tile SpMM by its row blocks), so that rows one kernel wrote or read are met in the same L2 by the next.
__device__ static inline int rwk_xcd_block(int bid, int nblocks) {
    const int per_xcd = nblocks / 8;
    return bid < per_xcd * 8 ? (bid % 8) * per_xcd + bid / 8 : bid;
}

