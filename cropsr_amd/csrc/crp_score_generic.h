// crp_score_generic.h -- the on-target score in the other two accumulation orders
// the reference produces (device code, slow path, a handful of rows per CSV).
//
// The reference scores a batch with two numpy matmuls (CROPSR.py:305,311).  With
// the pinned BLAS (OpenBLAS 0.3.29, SkylakeX runtime core, one thread) the
// floating-point sum of one row depends on WHERE the row sits in its batch of n:
//
//   CRP_ORDER_BODY4  rows 0 .. 4*floor(n/4)-1, and the last row when n%4 is 1 or 3
//                    (dgemv_t 4x4 / 4x1 kernels): four lanes by (flat index mod 4),
//                    combined (l0+l2)+(l1+l3).  This is the fast path (crp_score.h).
//   CRP_ORDER_TAIL2  rows 4*floor(n/4) and +1 when n%4 is 2 or 3 (dgemv_t 4x2
//                    kernel): two lanes by (flat index mod 2), combined l0+l1.
//   CRP_ORDER_DOT1   the only row of a batch of one (numpy calls ddot): the
//                    SkylakeX ddot kernel -- 4x8 lanes over blocks of 32, folded
//                    to 4x4, one more block of 16, ((a0+a1)+a2)+a3, halves, hadd,
//                    then a scalar tail.
//
// All three verified row-for-row against the real reference (tests/golden,
// DESIGN.md "Accumulation orders").  Products are one-hot x weight, hence exact;
// only the order of the additions differs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dense_weights.inc"

#define CRP_ORDER_BODY4 0
#define CRP_ORDER_TAIL2 1
#define CRP_ORDER_DOT1 2

// one-hot x weight of flat index i; code[p] in 0..3 (A,T,C,G) or -1
__device__ __forceinline__ double crp_term1(const int8_t *code, int i)
{
    return (code[i >> 2] == (i & 3) ? 1.0 : 0.0) * CRP_W1[i];
}
__device__ __forceinline__ double crp_term2(const int8_t *code, int i)
{
    const int p = i >> 4;
    return ((code[p] == ((i >> 2) & 3) && code[p + 1] == (i & 3)) ? 1.0 : 0.0) * CRP_W2[i];
}

template <bool SECOND>
__device__ inline double crp_sum_tail2(const int8_t *code)
{
    constexpr int N = SECOND ? 464 : 120;
    double l0 = 0.0, l1 = 0.0;
    for (int i = 0; i < N; i += 2) {
        l0 += SECOND ? crp_term2(code, i) : crp_term1(code, i);
        l1 += SECOND ? crp_term2(code, i + 1) : crp_term1(code, i + 1);
    }
    return l0 + l1;
}

template <bool SECOND>
__device__ inline double crp_sum_dot1(const int8_t *code)
{
    constexpr int N = SECOND ? 464 : 120;
    constexpr int N16 = N & -16, N32 = N16 & ~31;
    double wide[4][8];
    for (int q = 0; q < 4; ++q)
        for (int j = 0; j < 8; ++j) wide[q][j] = 0.0;
    int i = 0;
    for (; i < N32; i += 32)
        for (int q = 0; q < 4; ++q)
            for (int j = 0; j < 8; ++j) {
                const int k = i + 8 * q + j;
                wide[q][j] += SECOND ? crp_term2(code, k) : crp_term1(code, k);
            }
    double acc[4][4];
    for (int q = 0; q < 4; ++q)
        for (int j = 0; j < 4; ++j) acc[q][j] = wide[q][j] + wide[q][j + 4];
    for (; i < N16; i += 16)
        for (int q = 0; q < 4; ++q)
            for (int j = 0; j < 4; ++j) {
                const int k = i + 4 * q + j;
                acc[q][j] += SECOND ? crp_term2(code, k) : crp_term1(code, k);
            }
    double v[4];
    for (int j = 0; j < 4; ++j) v[j] = ((acc[0][j] + acc[1][j]) + acc[2][j]) + acc[3][j];
    double d = (v[0] + v[2]) + (v[1] + v[3]);
    for (; i < N; ++i) d += SECOND ? crp_term2(code, i) : crp_term1(code, i);
    return d;
}
