// fr_bn254.h -- BN254 scalar-field (Fr) arithmetic for gfx950 lanes: 8 x 32-bit limbs, Montgomery
// form with R = 2^256, bit-compatible with gnark-crypto's fr.Element ([4]uint64 little-endian,
// always canonical in [0,q)); replaces the fr.Mul/Square/Add/Sub calls the reference makes at e.g.
// poly/multilin.go:32-34, circuit/gates/cipher.go:34-40, sumcheck/algo.go:124-125,157,165,183,190.
//
// One lane owns one element.  The multiplication is the generated column schedule in
// fr_mont_gen.inc (v_mad_u64_u32 + v_addc_co_u32 per limb product; no MFMA: this is exact integer
// modular arithmetic, not a contraction).  The same header compiles for the host (portable branch),
// which is how the schedule is unit-tested without a GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define FR_HD __host__ __device__ __forceinline__
#else
#define FR_HD inline
#endif

typedef uint32_t u32;
typedef uint64_t u64;

struct Fr {
    u32 v[8];
};

// q = 21888242871839275222246405745257275088548364400416034343698204186575808495617 (hash/ark.go:7)
#define FRQ0 0xf0000001u
#define FRQ1 0x43e1f593u
#define FRQ2 0x79b97091u
#define FRQ3 0x2833e848u
#define FRQ4 0x8181585du
#define FRQ5 0xb85045b6u
#define FRQ6 0xe131a029u
#define FRQ7 0x30644e72u
#define FR_QINV32 0xefffffffu  // -q^-1 mod 2^32

// Copy of a finished limb out of the 64-bit accumulator pair into a register of its own.  Without a real
// move hipcc keeps every result limb in the low half of its own 64-bit register tuple (twice the VGPRs);
// an identity v_mov_b32_dpp is a move the compiler schedules itself (an inline-asm v_mov would cost a
// padding s_nop per statement).
#if defined(__HIP_DEVICE_COMPILE__)
#define FR_LIMB_COPY(x) ((u32)__builtin_amdgcn_mov_dpp((int)(x), 0xE4, 0xF, 0xF, false))
#else
#define FR_LIMB_COPY(x) (x)
#endif

// portable multiply-accumulate into the 96-bit column accumulator (host branch of the schedule)
#define FR_MADC(acc, ovf, x, y)              \
    do {                                     \
        u64 _p = (u64)(x) * (u64)(y);        \
        u64 _s = (acc) + _p;                 \
        (ovf) += (_s < _p) ? 1u : 0u;        \
        (acc) = _s;                          \
    } while (0)

// multiply-add whose sum is statically known to stay below 2^64 (tools/gen_mont_asm.py bounds every column): no
// carry tracking.  -DFR_CHECK_SKIPS (CPU unit tests) counts a wrap-around in fr_skip_overflows.
#if defined(FR_CHECK_SKIPS) && !defined(__HIP_DEVICE_COMPILE__)
static long fr_skip_overflows = 0;
#define FR_MADN(acc, x, y)                         \
    do {                                           \
        u64 _p = (u64)(x) * (u64)(y);              \
        u64 _s = (acc) + _p;                       \
        if (_s < _p) fr_skip_overflows++;          \
        (acc) = _s;                                \
    } while (0)
#else
#define FR_MADN(acc, x, y) ((acc) += (u64)(x) * (u64)(y))
#endif

FR_HD u32 fr_addc(u32 a, u32 b, u32 cin, u32* cout) {
#if defined(__clang__)
    return __builtin_addc(a, b, cin, cout);
#else
    u64 s = (u64)a + b + cin;
    *cout = (u32)(s >> 32);
    return (u32)s;
#endif
}
FR_HD u32 fr_subb(u32 a, u32 b, u32 bin, u32* bout) {
#if defined(__clang__)
    return __builtin_subc(a, b, bin, bout);
#else
    u64 s = (u64)a - b - bin;
    *bout = (u32)(s >> 63);
    return (u32)s;
#endif
}

FR_HD Fr fr_zero() {
    Fr r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.v[j] = 0;
    return r;
}
// Montgomery form of 1 (2^256 mod q)
FR_HD Fr fr_one() {
    Fr r = {{0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u}};
    return r;
}

// r = t - q if t >= q else t        (t < 2q)
FR_HD Fr fr_reduce_once(const Fr& t) {
    const u32 q[8] = {FRQ0, FRQ1, FRQ2, FRQ3, FRQ4, FRQ5, FRQ6, FRQ7};
    u32 d[8];
    u32 br = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) d[j] = fr_subb(t.v[j], q[j], br, &br);
    Fr r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.v[j] = br ? t.v[j] : d[j];
    return r;
}

// a + b mod q, canonical inputs and output
FR_HD Fr fr_add(const Fr& a, const Fr& b) {
    Fr s;
    u32 c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s.v[j] = fr_addc(a.v[j], b.v[j], c, &c);
    return fr_reduce_once(s);  // a+b < 2q < 2^255: no carry out of limb 7
}

// a - b mod q, canonical inputs and output
FR_HD Fr fr_sub(const Fr& a, const Fr& b) {
    const u32 q[8] = {FRQ0, FRQ1, FRQ2, FRQ3, FRQ4, FRQ5, FRQ6, FRQ7};
    u32 s[8];
    u32 br = 0, c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = fr_subb(a.v[j], b.v[j], br, &br);
    const u32 mask = 0u - br;
    Fr r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.v[j] = fr_addc(s[j], q[j] & mask, c, &c);
    return r;
}

FR_HD Fr fr_dbl(const Fr& a) { return fr_add(a, a); }

// Montgomery product a*b/2^256 mod q, result < a*b/2^256 + q: in [0, 2q) for inputs < 2q (4q^2 + q*2^256 < 2q*2^256),
// below 2.7q for inputs < 3q (q/2^256 = 0.189) -- any inputs whose result stays below 2^256 = 5.29q are exact
FR_HD Fr fr_mont_mul_raw(const Fr& a, const Fr& b) {
    Fr r;
#include "fr_mont_gen.inc"
    return r;
}

// Montgomery square a*a/2^256 mod q: the cross products a_i*a_j are taken once against the doubled operand (100 limb
// products instead of 128, fr_sqr_gen.inc).  PRECONDITION a < 2q (2a must fit eight limbs); the result is the very
// integer fr_mont_mul_raw(a, a) returns (same T, same m_i), in [0, 2q).
FR_HD Fr fr_mont_sqr_raw(const Fr& a) {
    Fr r;
#include "fr_sqr_gen.inc"
    return r;
}

// two independent lazy products at once (instruction streams interleaved on the device: fr_mont2_gen.inc)
FR_HD void fr_mont_mul2_raw(Fr& r0, Fr& r1, const Fr& a0, const Fr& b0, const Fr& a1, const Fr& b1) {
#if defined(__HIP_DEVICE_COMPILE__)
#include "fr_mont2_gen.inc"
#else
    r0 = fr_mont_mul_raw(a0, b0);
    r1 = fr_mont_mul_raw(a1, b1);
#endif
}

// ---- deferred reduction for products that only feed a sum -----------------------------------------
// A (17 limbs, un-reduced, < 2^542) += a*b as a plain integer product (fr_mac_wide_gen.inc); the sum of
// many such products is reduced ONCE by fr_redc_wide instead of once per product.
// PRECONDITION a, b < 3q (lazy Montgomery products or canonical elements at every call site): the generated
// schedule leaves out the carry instructions that this bound on the top limbs makes unnecessary.
#define FR_WIDE_LIMBS 17
FR_HD void fr_mac_wide(u32 (&A)[FR_WIDE_LIMBS], const Fr& a, const Fr& b) {
#include "fr_mac_wide_gen.inc"
}
// Montgomery reduction of the wide sum: out (9 limbs, un-reduced) == A / 2^256 (mod q), out < A/2^256 + q.
// With A < 2^542 the result fits the 9-limb accumulators the block reduction sums exactly.
FR_HD void fr_redc_wide(u32 (&out)[9], const u32 (&A)[FR_WIDE_LIMBS]) {
    const u32 q[8] = {FRQ0, FRQ1, FRQ2, FRQ3, FRQ4, FRQ5, FRQ6, FRQ7};
    u32 T[FR_WIDE_LIMBS];
#pragma unroll
    for (int j = 0; j < FR_WIDE_LIMBS; j++) T[j] = A[j];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u32 m = T[i] * FR_QINV32;
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c += (u64)m * q[j] + T[i + j];      // < 2^64: (2^32-1)^2 + 2*(2^32-1)
            T[i + j] = (u32)c;
            c >>= 32;
        }
#pragma unroll
        for (int j = i + 8; j < FR_WIDE_LIMBS; j++) {
            c += T[j];
            T[j] = (u32)c;
            c >>= 32;
        }
    }
#pragma unroll
    for (int j = 0; j < 9; j++) out[j] = T[8 + j];
}

// canonical residue of a 9-limb value below 16q (a wide sum of at most 64 products after fr_redc_wide:
// 64*q^2/2^256 + q < 13.2q): conditional subtraction of 8q, 4q, 2q, q
FR_HD Fr fr_canon_lt16q(const u32 (&w)[9]) {
    const u32 q[9] = {FRQ0, FRQ1, FRQ2, FRQ3, FRQ4, FRQ5, FRQ6, FRQ7, 0u};
    u32 x[9];
#pragma unroll
    for (int j = 0; j < 9; j++) x[j] = w[j];
#pragma unroll
    for (int s = 3; s >= 0; s--) {
        u32 d[9];
        u32 br = 0;
#pragma unroll
        for (int j = 0; j < 9; j++) {
            const u32 qs = s ? ((q[j] << s) | (j ? (q[j - 1] >> (32 - s)) : 0u)) : q[j];
            d[j] = fr_subb(x[j], qs, br, &br);
        }
#pragma unroll
        for (int j = 0; j < 9; j++) x[j] = br ? x[j] : d[j];
    }
    Fr r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.v[j] = x[j];
    return r;
}

// Product with a launch-wide constant c (the fold challenge of a round), c given by the host as the pair
// ca = c * 2^-128 mod q, cb = c (both canonical, Montgomery form like every element):
//     a * c / 2^256 == (a_lo * ca + a_hi * cb) / 2^128  (mod q),   a = a_lo + 2^128 a_hi,
// so four Montgomery steps suffice (96 limb products instead of 128).  PRECONDITION a < 3q, ca and cb canonical
// (the carry planning of the generated schedule relies on their top limbs); result < 3q.
FR_HD Fr fr_mul_const2_raw(const Fr& a, const Fr& ca, const Fr& cb) {
    Fr r;
#include "fr_mulc2_gen.inc"
    return r;
}
// a + b without reduction (the caller knows the bound of the sum)
FR_HD Fr fr_add_raw(const Fr& a, const Fr& b) {
    Fr s;
    u32 c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s.v[j] = fr_addc(a.v[j], b.v[j], c, &c);
    return s;
}
// canonical residue of t < 4q (< 2^256): subtract 2q, then q, when they fit
FR_HD Fr fr_reduce_lt4q(const Fr& t) {
    const u32 q[8] = {FRQ0, FRQ1, FRQ2, FRQ3, FRQ4, FRQ5, FRQ6, FRQ7};
    Fr x = t;
#pragma unroll
    for (int s = 1; s >= 0; s--) {
        u32 dd[8];
        u32 br = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const u32 qs = s ? ((q[j] << 1) | (j ? (q[j - 1] >> 31) : 0u)) : q[j];
            dd[j] = fr_subb(x.v[j], qs, br, &br);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) x.v[j] = br ? x.v[j] : dd[j];
    }
    return x;
}

// canonical product
FR_HD Fr fr_mul(const Fr& a, const Fr& b) { return fr_reduce_once(fr_mont_mul_raw(a, b)); }
FR_HD Fr fr_sqr(const Fr& a) { return fr_reduce_once(fr_mont_sqr_raw(a)); }

// x^7 as the reference does it: sq, mul, sq, mul (hash/poseidon.go:129-135, circuit/gates/cipher.go:36-40)
// PRECONDITION x < 2q (the squaring schedule)
FR_HD Fr fr_pow7(const Fr& x) {
    Fr t = fr_mont_sqr_raw(x);      // x^2  (< 2q)
    t = fr_mont_mul_raw(t, x);      // x^3
    t = fr_mont_sqr_raw(t);         // x^6
    return fr_mul(t, x);            // x^7, canonical
}

FR_HD bool fr_eq(const Fr& a, const Fr& b) {
    u32 d = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) d |= a.v[j] ^ b.v[j];
    return d == 0;
}
