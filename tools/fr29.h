// fr29.h -- EXPERIMENT (not part of the library): carry-free BN254-Fr multiplication for gfx950 on 9 limbs of
// 29 bits with Montgomery radix 2^261, benchmarked by tools/ubench (profiles/r01_ubench_radix29.txt).
//
// Motivation: on gfx950 every VALU instruction that produces or consumes a carry (v_addc_co_u32,
// v_add_co_u32, v_subb_co_u32 ...) issues at the same half rate as v_mad_u64_u32 (4.3 cycles per wave
// against 2.4 for a plain v_add_u32), so in the 8 x 32-bit schedule of fr_bn254.h the 136 carry instructions
// cost as much as the 136 limb products.  With 29-bit limbs a limb product is below 2^58 and the 18
// products of a column fit a 64-bit accumulator: 162 v_mad_u64_u32 and no carry instruction, plus one mask
// and one 64-bit shift per column.  Tables would stay in the boundary layout (converting is bit re-slicing);
// every product loses a factor 2^5 against gnark-crypto's 2^256 Montgomery form, which can be tracked per
// value and folded into one constant (a full round kernel built this way was bit-exact against the oracle).
//
// Result (round 1): 1035 cycles per product against 1107 for the 32-bit schedule in isolation at full
// occupancy (the 64-bit shifts and adds of the column hand-over are half-rate too, and each column is one
// dependent MAD chain), and NO gain in the round kernel once the format conversions are paid (1.53 ms against
// 1.48 ms per round-0 launch at bN = 24).  Kept for the next round as a measured dead end / starting point.
#pragma once
#include "../gkr-mimc_amd/csrc/fr_bn254.h"

#define F29_LIMBS 9
#define F29_BITS 29
#define F29_MASK 0x1FFFFFFFu
#define F29_QINV 0x0FFFFFFFu   // -q^-1 mod 2^29
#define F29_WIDE 19            // limbs of a wide (un-reduced) sum of products: < 2^554

struct F29 {
    u32 v[F29_LIMBS];
};

#define F29_Q_INIT {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu}

// bit re-slicing 8 x 32 -> 9 x 29 (value unchanged)
FR_HD F29 f29_from_fr(const Fr& a) {
    F29 r;
#pragma unroll
    for (int k = 0; k < F29_LIMBS; k++) {
        const int bit = F29_BITS * k, w = bit >> 5, s = bit & 31;
        u32 x = a.v[w] >> s;
        if (s > 32 - F29_BITS && w + 1 < 8) x |= a.v[w + 1] << (32 - s);
        r.v[k] = x & F29_MASK;
    }
    return r;
}
// 9 x 29 (normalised limbs, value < 2^256) -> 8 x 32
FR_HD Fr f29_to_fr(const F29& a) {
    Fr r;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        const int bit = 32 * w, k = bit / F29_BITS, s = bit % F29_BITS;   // word w starts inside limb k at bit s
        u32 x = a.v[k] >> s;
        x |= a.v[k + 1] << (F29_BITS - s);
        if (2 * F29_BITS - s < 32 && k + 2 < F29_LIMBS) x |= a.v[k + 2] << (2 * F29_BITS - s);
        r.v[w] = x;
    }
    return r;
}

// Montgomery product a*b / 2^261 (mod q): normalised limbs, value < a*b/2^261 + q
FR_HD F29 f29_mont_mul(const F29& a, const F29& b) {
    const u32 q[F29_LIMBS] = F29_Q_INIT;
    u32 m[F29_LIMBS];
    F29 r;
    u64 acc = 0;
#pragma unroll
    for (int c = 0; c < 2 * F29_LIMBS - 1; c++) {
        const int lo = c > F29_LIMBS - 1 ? c - (F29_LIMBS - 1) : 0, hi = c < F29_LIMBS - 1 ? c : F29_LIMBS - 1;
#pragma unroll
        for (int i = lo; i <= hi; i++) acc += (u64)a.v[i] * b.v[c - i];
#pragma unroll
        for (int i = lo; i <= hi; i++)
            if (!(c < F29_LIMBS && i == c)) acc += (u64)m[i] * q[c - i];
        if (c < F29_LIMBS) {
            m[c] = ((u32)acc * F29_QINV) & F29_MASK;
            acc += (u64)m[c] * q[0];
        } else {
            r.v[c - F29_LIMBS] = (u32)acc & F29_MASK;
        }
        acc >>= F29_BITS;
    }
    r.v[F29_LIMBS - 1] = (u32)acc;
    return r;
}

// A (19 normalised 29-bit limbs; the top limb absorbs the growth) += a*b as a plain integer product
FR_HD void f29_mac_wide(u32 (&A)[F29_WIDE], const F29& a, const F29& b) {
    u64 acc = 0;
#pragma unroll
    for (int c = 0; c < 2 * F29_LIMBS - 1; c++) {
        const int lo = c > F29_LIMBS - 1 ? c - (F29_LIMBS - 1) : 0, hi = c < F29_LIMBS - 1 ? c : F29_LIMBS - 1;
        acc += A[c];
#pragma unroll
        for (int i = lo; i <= hi; i++) acc += (u64)a.v[i] * b.v[c - i];
        A[c] = (u32)acc & F29_MASK;
        acc >>= F29_BITS;
    }
    acc += A[17];
    A[17] = (u32)acc & F29_MASK;
    A[18] += (u32)(acc >> F29_BITS);
}

// Montgomery square a*a / 2^261 (mod q) for normalised limbs: the 36 cross products are taken once against
// the doubled operand (45 + 81 limb products instead of 162)
FR_HD F29 f29_mont_sqr(const F29& a) {
    const u32 q[F29_LIMBS] = F29_Q_INIT;
    u32 m[F29_LIMBS], a2[F29_LIMBS];
#pragma unroll
    for (int i = 0; i < F29_LIMBS; i++) a2[i] = a.v[i] << 1;
    F29 r;
    u64 acc = 0;
#pragma unroll
    for (int c = 0; c < 2 * F29_LIMBS - 1; c++) {
        const int lo = c > F29_LIMBS - 1 ? c - (F29_LIMBS - 1) : 0, hi = c < F29_LIMBS - 1 ? c : F29_LIMBS - 1;
#pragma unroll
        for (int i = lo; i <= hi; i++) {
            if (2 * i < c) acc += (u64)a2[i] * a.v[c - i];
            else if (2 * i == c) acc += (u64)a.v[i] * a.v[i];
        }
#pragma unroll
        for (int i = lo; i <= hi; i++)
            if (!(c < F29_LIMBS && i == c)) acc += (u64)m[i] * q[c - i];
        if (c < F29_LIMBS) {
            m[c] = ((u32)acc * F29_QINV) & F29_MASK;
            acc += (u64)m[c] * q[0];
        } else {
            r.v[c - F29_LIMBS] = (u32)acc & F29_MASK;
        }
        acc >>= F29_BITS;
    }
    r.v[F29_LIMBS - 1] = (u32)acc;
    return r;
}

// Montgomery reduction of a wide sum T < 2^548: L (11 normalised limbs) == T / 2^261 (mod q), L < T/2^261 + q
#define F29_REDC_LIMBS (F29_WIDE + 1 - F29_LIMBS)
FR_HD void f29_redc_wide_limbs(u32 (&L)[F29_REDC_LIMBS], const u32 (&A)[F29_WIDE]) {
    const u32 q[F29_LIMBS] = F29_Q_INIT;
    u64 T[F29_WIDE + 1];
#pragma unroll
    for (int j = 0; j < F29_WIDE; j++) T[j] = A[j];
    T[F29_WIDE] = 0;
#pragma unroll
    for (int i = 0; i < F29_LIMBS; i++) {
        const u32 m = ((u32)T[i] * F29_QINV) & F29_MASK;
#pragma unroll
        for (int j = 0; j < F29_LIMBS; j++) T[i + j] += (u64)m * q[j];
        T[i + 1] += T[i] >> F29_BITS;       // T[i] is now 0 mod 2^29
    }
    u64 c = 0;                              // limbs 9..19 hold the quotient in redundant form (each < 2^63)
#pragma unroll
    for (int j = 0; j < F29_REDC_LIMBS; j++) {
        c += T[F29_LIMBS + j];
        L[j] = (u32)c & F29_MASK;
        c >>= F29_BITS;
    }
}
// 11 normalised limbs (value < 2^288) -> 9 x 32-bit words
FR_HD void f29_pack_words9(u32 (&out)[9], const u32 (&L)[F29_REDC_LIMBS]) {
#pragma unroll
    for (int w = 0; w < 9; w++) {
        const int bit = 32 * w, k = bit / F29_BITS, s = bit % F29_BITS;
        u32 x = L[k] >> s;
        if (k + 1 < F29_REDC_LIMBS) x |= L[k + 1] << (F29_BITS - s);
        if (2 * F29_BITS - s < 32 && k + 2 < F29_REDC_LIMBS) x |= L[k + 2] << (2 * F29_BITS - s);
        out[w] = x;
    }
}
FR_HD void f29_redc_wide(u32 (&out)[9], const u32 (&A)[F29_WIDE]) {
    u32 L[F29_REDC_LIMBS];
    f29_redc_wide_limbs(L, A);
    f29_pack_words9(out, L);
}
