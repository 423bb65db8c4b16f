// crp_score.h -- in-register on-target score of one 30-mer (device code, gfx950).
//
// Replaces rs1_score (reference CROPSR.py:285-313) for one row: the row arrives
// as four 30-bit masks (bit p set in mX <=> scoring character t[p] == X) instead
// of the reference's (n,120)/(n,464) one-hot float64 matrices.
//
// Bit-exactness (see gen_score_terms.py for the derivation):
//   * the two matmuls become eight ordered chains of gated FMAs, one chain per
//     OpenBLAS dgemv lane, combined as (A+C)+(T+G);
//   * exp() is glibc 2.35's table-driven algorithm (N = 128) with the FMA
//     placement of its x86-64 FMA variant -- the exp the reference's numpy call
//     resolves to on hosts without AVX-512 (tests/golden `libm` environment);
//   * 1/(1+e) is the IEEE-correct f64 division (no fast-math anywhere).
// The translation unit must be compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "score_terms.inc"

// 2^(k/128) table; staged into LDS by every kernel that scores.
__device__ const uint64_t CRP_EXP_TAB[256] = {
#include "exp_table.inc"
};

// Pre-scaled weights in evaluation order, as literals: two s_mov_b32 per term on the scalar ALU,
// hidden under the VALU work.  (Wide scalar loads from constant memory and uniform ds_read_b64
// from LDS were both measured slower)
static constexpr double CRP_WS_LIT[CRP_WS_COUNT] = CRP_WS_TABLE;
#define CRP_WS(i) (CRP_WS_LIT[i])

// Chain-prefix tables of the PAM-variant scorer (gen_score_terms.py, "table form"): the value of a
// chain after its first k terms, for every pattern of the k gate bits.  Staged into LDS by the
// emit kernel; `off` is a byte offset inside the chain's table.
// (the table's data, CRP_SCORE_TAB_DATA, is part of the image crp_kernels.hip stages: TabsImage)
__device__ __forceinline__ double crp_tab_at(const double *tab, uint32_t base_bytes, uint32_t off_bytes)
{
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(tab) + base_bytes + off_bytes);
}

#define CRP_TERM(acc, copy, bit, wc) \
    acc = __builtin_fma(__hiloint2double((int)((copy) & (1u << (bit))), 0), (wc), acc);
// second order: both bases tested in the gate (three-input AND = one v_bitop3_b32)
#define CRP_TERM2(acc, copy1, copy2, bit, wc) \
    acc = __builtin_fma(__hiloint2double((int)((copy1) & (copy2) & (1u << (bit))), 0), (wc), acc);

// exp(x) for |x| < 512.  `tab` points at a copy of CRP_EXP_TAB (LDS).
__device__ __forceinline__ double crp_exp(double x, const uint64_t *tab)
{
    const double inv_ln2_n = 0x1.71547652b82fep0 * 128;
    const double neg_ln2_hi_n = -0x1.62e42fefa0000p-8;
    const double neg_ln2_lo_n = -0x1.cf79abc9e3b3ap-47;
    const double shift = 0x1.8p52;
    const double c2 = 0x1.ffffffffffdbdp-2, c3 = 0x1.555555555543cp-3;
    const double c4 = 0x1.55555cf172b91p-5, c5 = 0x1.1111167a4d017p-7;
    // (glibc returns 1.0 + x early for |x| < 2^-54 to keep the inexact/underflow flags clean; the
    // value is the same on the main path -- tail + r == x, fma(1, x, 1) rounds like 1 + x -- so the
    // test is not needed here)
    const double z = inv_ln2_n * x;
    double kd = z + shift;
    const uint64_t ki = (uint64_t)__double_as_longlong(kd);
    kd -= shift;
    const double r = __builtin_fma(kd, neg_ln2_lo_n, __builtin_fma(kd, neg_ln2_hi_n, x));
    const uint32_t idx = 2u * ((uint32_t)ki & 127u);
    const uint64_t top = ki << (52 - 7);
    const double tail = __longlong_as_double((long long)tab[idx]);
    const uint64_t sbits = tab[idx + 1] + top;
    const double r2 = r * r;
    const double p_lo = __builtin_fma(r, c3, c2);
    const double p_hi = __builtin_fma(r, c5, c4);
    double tmp = __builtin_fma(r2, p_lo, tail + r);
    tmp = __builtin_fma(r2 * r2, p_hi, tmp);
    const double scale = __longlong_as_double((long long)sbits);
    return __builtin_fma(scale, tmp, scale);
}

// 1 / d for 1 <= d < 2^64, correctly rounded: the f64 division sequence of the compiler
// (v_rcp_f64, two Newton steps, quotient, remainder, final fma) without v_div_scale_f64 /
// v_div_fmas_f64 / v_div_fixup_f64, which are the identity when numerator and denominator are
// this far from the exponent range's ends.
__device__ __forceinline__ double crp_recip(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double rem = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(rem, r, r);
}

// pre = -(s1 + s2 + intersect + low_gc)  (CROPSR.py:312), score = 1/(1+exp(pre)) (:313)
// PAM = true: the row is a window found by the PAM scan with guide length 20, whose
// characters 3 and 4 are always 'C' (SURVEY.md A.3); the four terms that test those
// positions are folded into the chain start values / dropped by the generator.
// score_tab: LDS copy of CRP_SCORE_TAB (PAM variant only; may be null otherwise).
template <bool PAM>
__device__ __forceinline__ void crp_score_masks(uint32_t mA, uint32_t mT, uint32_t mC, uint32_t mG,
                                                const uint64_t *exp_tab, const double *score_tab, double &pre,
                                                double &score)
{
    double fA = PAM ? CRP_PAM_INIT_fA : 0.0, fT = PAM ? CRP_PAM_INIT_fT : 0.0;
    double fC = PAM ? CRP_PAM_INIT_fC : 0.0, fG = PAM ? CRP_PAM_INIT_fG : 0.0;
    double sA = PAM ? CRP_PAM_INIT_sA : 0.0, sT = PAM ? CRP_PAM_INIT_sT : 0.0;
    double sC = PAM ? CRP_PAM_INIT_sC : 0.0, sG = PAM ? CRP_PAM_INIT_sG : 0.0;
    if (PAM) {
        CRP_SCORE_BODY_PAM_TABLES(mA, mT, mC, mG)
    } else {
        CRP_SCORE_BODY(mA, mT, mC, mG)
    }
    const double s1 = (fA + fC) + (fT + fG);
    const double s2 = (sA + sC) + (sT + sG);
    pre = (((s1 + s2) + CRP_INTERSECT) + CRP_LOW_GC) * -1.0;
    score = crp_recip(1.0 + crp_exp(pre, exp_tab));
}
