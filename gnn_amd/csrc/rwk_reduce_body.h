// Device body of the first level of the deterministic row reduction, shared by rwk_project.hip (stand-alone
// kernel) and rwk_spmm.hip (merged launch with the tile SpMM).
#pragma once

#include "rwk_common.h"

constexpr int kReduceMaxFinalRows = 32;

// Level 1: workgroup (bx, by) of 256 threads owns 64 columns and the rows [by*rows_per_block, (by+1)*rows_per_block);
// its 4 wavefronts ("slices") take every 4th row of that range, then the slices are added in slice order.
__device__ inline void reduce_rows_level1_body(int bx, int by, const float *__restrict__ in, long long p, long long v,
                                               int rows_per_block, float *__restrict__ out) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const long long col = (long long)bx * 64 + lane;
    const long long r0 = (long long)by * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > p) r1 = p;
    float acc = 0.f;
    if (col < v) {
        long long r = r0 + slice;
        for (; r + 12 < r1; r += 16) {                  // four rows in flight, added in row order
            const float a0 = in[r * v + col], a1 = in[(r + 4) * v + col], a2 = in[(r + 8) * v + col],
                        a3 = in[(r + 12) * v + col];
            acc += a0;
            acc += a1;
            acc += a2;
            acc += a3;
        }
        for (; r < r1; r += 4) acc += in[r * v + col];
    }
    part[slice][lane] = acc;
    __syncthreads();
    if (slice == 0 && col < v)
        out[(long long)by * v + col] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}
