// fr52.h -- EXPERIMENT (not part of the library): BN254-Fr Montgomery multiplication on 5 limbs of 52 bits held in
// doubles, limb products split into exact high and low halves by pairs of v_fma_f64 (Emmart, Zheng, Weems:
// "Faster modular exponentiation using double precision floating point arithmetic on the GPU"), Montgomery radix
// 2^260.  Benchmarked by tools/ubench (profiles/r02_ubench_fp64.txt) and checked on the host by
// tests/cpp/test_fr52.cpp.
//
// Motivation (round-1 verdict): v_fma_f64 issues at the rate of v_mad_u64_u32 (4.3 cycles per wave) and a 52 x 52-bit
// limb product needs two of them where the 32-bit schedule needs 2.6 MAD + 2.6 ADDC for the same bits -- if the
// additions were free.  They are not on gfx950: a limb product is
//     hi = fma_rz(a, b, C1)                  C1 = 2^104: the mantissa of hi holds floor(a*b / 2^52)
//     lo = fma_rz(a, b, C2 - hi)             C2 = 2^104 + 2^52: = 2^52 + (a*b mod 2^52), exact
// (round-toward-zero: the kernel sets the FP64 rounding mode with s_setreg, the host test with fesetround) and both
// halves must be ADDED into column sums.  FP64 adds round, so the sums are taken on the bit patterns as
// 64-bit integers (the exponent fields are removed in bulk at the end) -- and a 64-bit integer add (v_lshl_add_u64,
// or v_add_co + v_addc) is a half-rate instruction like the FMA itself, as is the v_add_f64 that forms C2 - hi.
// Per limb product: 2 FMA + 1 FP add + 2 integer adds = 5 half-rate instructions; 25 products = 125 for the plain
// product alone, against 64 MAD + 50 ADDC = 114 for the 32-bit schedule of this round; the Montgomery half adds
// the m_i = t_i * qinv mod 2^52 chain (two more FMA pairs and integer <-> double conversions per step), 25 more
// split products and a carry propagation: about 300 half-rate instructions against 229.  Measured numbers are in
// profiles/r02_ubench_fp64.txt; the variant is NOT adopted.
#pragma once
#include <math.h>
#include <string.h>
#include "../gkr-mimc_amd/csrc/fr_bn254.h"

#define F52_LIMBS 5
#define F52_BITS 52
#define F52_MASK ((1ULL << 52) - 1)

struct F52 {
    double v[F52_LIMBS];   // integers in [0, 2^52)
};

FR_HD u64 f52_bits(double x) {
    u64 r;
    memcpy(&r, &x, 8);
    return r;
}
// 8 x 32 -> 5 x 52 (value unchanged)
FR_HD F52 f52_from_fr(const Fr& a) {
    u64 w[4];
    for (int k = 0; k < 4; k++) w[k] = (u64)a.v[2 * k] | ((u64)a.v[2 * k + 1] << 32);
    F52 r;
    r.v[0] = (double)(w[0] & F52_MASK);
    r.v[1] = (double)(((w[0] >> 52) | (w[1] << 12)) & F52_MASK);
    r.v[2] = (double)(((w[1] >> 40) | (w[2] << 24)) & F52_MASK);
    r.v[3] = (double)(((w[2] >> 28) | (w[3] << 36)) & F52_MASK);
    r.v[4] = (double)(w[3] >> 16);
    return r;
}
// 5 normalised 52-bit integer limbs (value < 2^256) -> 8 x 32
FR_HD Fr f52_limbs_to_fr(const u64 (&l)[F52_LIMBS]) {
    u64 w[4];
    w[0] = l[0] | (l[1] << 52);
    w[1] = (l[1] >> 12) | (l[2] << 40);
    w[2] = (l[2] >> 24) | (l[3] << 28);
    w[3] = (l[3] >> 36) | (l[4] << 16);
    Fr r;
    for (int k = 0; k < 4; k++) {
        r.v[2 * k] = (u32)w[k];
        r.v[2 * k + 1] = (u32)(w[k] >> 32);
    }
    return r;
}

// one split limb product added into two column sums (64-bit integer sums of bit patterns)
#define F52_C1 20282409603651670423947251286016.0 /* 2^104 */
#define F52_C2 20282409603651674927546878656512.0 /* 2^104 + 2^52 */
FR_HD void f52_mac(u64& col_lo, u64& col_hi, double a, double b) {
    const double hi = fma(a, b, F52_C1);          // round toward zero: bits(2^104) + floor(a*b / 2^52)
    const double lo = fma(a, b, F52_C2 - hi);     // = 2^52 + (a*b mod 2^52), exact: bits(2^52) + low half
    col_hi += f52_bits(hi);                       // sums of bit patterns wrap mod 2^64; the exponent fields are
    col_lo += f52_bits(lo);                       // subtracted in bulk when a column is finished
}
#define F52_BITS_C1 0x4670000000000000ULL /* bits(2^104) */
#define F52_BITS_C3 0x4330000000000000ULL /* bits(2^52)  */
// integer in [0, 2^52) -> double without a conversion instruction: splice it under the exponent of 2^52
FR_HD double f52_to_double(u64 x) {
    const u64 b = F52_BITS_C3 | x;
    double d;
    memcpy(&d, &b, 8);
    return d - 4503599627370496.0;
}

// q and -q^-1 mod 2^52 as 52-bit limbs
#define F52_Q_INIT {0x1f593f0000001ULL, 0x4879b9709143eULL, 0x181585d2833e8ULL, 0xa029b85045b68ULL, 0x30644e72e131ULL}
#define F52_QINV 0x1f593efffffffULL

// Montgomery product a*b / 2^260 (mod q), normalised limbs out; value < a*b/2^260 + q
FR_HD void f52_mont_mul(u64 (&out)[F52_LIMBS], const F52& a, const F52& b) {
    const u64 qi[F52_LIMBS] = F52_Q_INIT;
    double q[F52_LIMBS];
    for (int j = 0; j < F52_LIMBS; j++) q[j] = f52_to_double(qi[j]);
    u64 col[2 * F52_LIMBS + 1];
    for (int k = 0; k < 2 * F52_LIMBS + 1; k++) col[k] = 0;
    int cnt_lo[2 * F52_LIMBS + 1] = {0}, cnt_hi[2 * F52_LIMBS + 1] = {0};
    // plain product: column k gets the low halves of a_i b_j (i + j = k) and the high halves of column k - 1
#pragma unroll
    for (int i = 0; i < F52_LIMBS; i++)
#pragma unroll
        for (int j = 0; j < F52_LIMBS; j++) {
            f52_mac(col[i + j], col[i + j + 1], a.v[i], b.v[j]);
            cnt_lo[i + j]++;
            cnt_hi[i + j + 1]++;
        }
    // Montgomery steps: m_i = (t_i * qinv) mod 2^52, t += m_i * q * 2^(52 i)
    const double qinv = f52_to_double(F52_QINV);
    u64 carry = 0;
#pragma unroll
    for (int i = 0; i < F52_LIMBS; i++) {
        // the finished value of column i (< 2^57): remove the exponent patterns, add the carry of the column below
        const u64 t = col[i] - (u64)cnt_lo[i] * F52_BITS_C3 - (u64)cnt_hi[i] * F52_BITS_C1 + carry;
        const double tl = f52_to_double(t & F52_MASK);
        // m = low 52 bits of tl * qinv: the low half of the split product
        const double mh = fma(tl, qinv, F52_C1);
        const double ml = fma(tl, qinv, F52_C2 - mh);
        const double md = ml - 4503599627370496.0;             // exact: ml = 2^52 + m
        u64 m_q0_lo = 0;
#pragma unroll
        for (int j = 0; j < F52_LIMBS; j++) {
            if (j == 0) {
                f52_mac(m_q0_lo, col[i + 1], md, q[0]);       // the low half cancels t_i by construction
                cnt_hi[i + 1]++;
            } else {
                f52_mac(col[i + j], col[i + j + 1], md, q[j]);
                cnt_lo[i + j]++;
                cnt_hi[i + j + 1]++;
            }
        }
        carry = (t + (m_q0_lo - F52_BITS_C3)) >> 52;          // t + lo(m*q0) is a multiple of 2^52
    }
    // the result limbs: columns 5..9 (+ the spill-over column 10), carries propagated
#pragma unroll
    for (int k = 0; k < F52_LIMBS; k++) {
        const int c = F52_LIMBS + k;
        const u64 t = col[c] - (u64)cnt_lo[c] * F52_BITS_C3 - (u64)cnt_hi[c] * F52_BITS_C1 + carry;
        if (k < F52_LIMBS - 1) {
            out[k] = t & F52_MASK;
            carry = t >> 52;
        } else {
            const u64 top = col[c + 1] - (u64)cnt_hi[c + 1] * F52_BITS_C1;   // high halves of the last column
            out[k] = t + (top << 52);
        }
    }
}
