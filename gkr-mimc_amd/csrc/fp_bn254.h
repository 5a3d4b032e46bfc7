// fp_bn254.h -- BN254 BASE-field (Fp) arithmetic for gfx950 lanes: the coordinates of the G1 points of the
// multi-scalar multiplications (g1.hip.h; gnark-crypto's fp.Element behind bn254.G1Affine / G1Jac, called at the
// reference's prover/gadget/prove.go:76,91,189,202,221).  Same shape as fr_bn254.h -- 8 x 32-bit limbs, Montgomery form
// with R = 2^256, bit-compatible with fp.Element ([4]uint64 little-endian) -- with the generated column schedules of
// tools/gen_mont_asm.py for this modulus (fp_mont_gen.inc, fp_sqr_gen.inc).
//
// LAZY RANGE: every Fp value a kernel holds is in [0, 2p).  Products of such values land in [0, 2p) again
// (4p^2 / 2^256 + p < 1.76p), sums and differences are reduced modulo 2p, a value is zero in the field when it is 0 or p,
// and only what leaves the device is made canonical (fp_canon).  The carry planning of the product relies on the bound.
// Compiles for the host too (portable branch of the schedules): that is how they are unit-tested without a GPU.
#pragma once
#include "fr_bn254.h"

struct Fp {
    u32 v[8];
};

// p = 21888242871839275222246405745257275088696311157297823662689037894645226208583
#define FPQ0 0xd87cfd47u
#define FPQ1 0x3c208c16u
#define FPQ2 0x6871ca8du
#define FPQ3 0x97816a91u
#define FPQ4 0x8181585du
#define FPQ5 0xb85045b6u
#define FPQ6 0xe131a029u
#define FPQ7 0x30644e72u
#define FP_QINV32 0xe4866389u  // -p^-1 mod 2^32

FR_HD Fp fp_zero() {
    Fp r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.v[j] = 0;
    return r;
}
// Montgomery form of 1 (2^256 mod p)
FR_HD Fp fp_one() {
    Fp r = {{0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u}};
    return r;
}

// a * b / 2^256 mod p, inputs and output in [0, 2p)
FR_HD Fp fp_mul(const Fp& a, const Fp& b) {
    Fp r;
#include "fp_mont_gen.inc"
    return r;
}
// a * a / 2^256 mod p (100 limb products instead of 128), input and output in [0, 2p)
FR_HD Fp fp_sqr(const Fp& a) {
    Fp r;
#include "fp_sqr_gen.inc"
    return r;
}

// t - m if t >= m else t, for the multiple m = k*p given by its limbs
FR_HD Fp fp_cond_sub(const Fp& t, const u32 (&m)[8]) {
    u32 d[8];
    u32 br = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) d[j] = fr_subb(t.v[j], m[j], br, &br);
    Fp r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.v[j] = br ? t.v[j] : d[j];
    return r;
}
#define FP_P_LIMBS {FPQ0, FPQ1, FPQ2, FPQ3, FPQ4, FPQ5, FPQ6, FPQ7}
#define FP_2P_LIMBS {0xb0f9fa8eu, 0x7841182du, 0xd0e3951au, 0x2f02d522u, 0x0302b0bbu, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u}

// a + b mod 2p   (a + b < 4p < 2^256)
FR_HD Fp fp_add(const Fp& a, const Fp& b) {
    const u32 p2[8] = FP_2P_LIMBS;
    Fp s;
    u32 c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s.v[j] = fr_addc(a.v[j], b.v[j], c, &c);
    return fp_cond_sub(s, p2);
}
// a - b mod 2p
FR_HD Fp fp_sub(const Fp& a, const Fp& b) {
    const u32 p2[8] = FP_2P_LIMBS;
    u32 s[8];
    u32 br = 0, c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = fr_subb(a.v[j], b.v[j], br, &br);
    const u32 mask = 0u - br;
    Fp r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.v[j] = fr_addc(s[j], p2[j] & mask, c, &c);
    return r;
}
FR_HD Fp fp_dbl(const Fp& a) { return fp_add(a, a); }
FR_HD Fp fp_neg(const Fp& a) { return fp_sub(fp_zero(), a); }
// the canonical residue in [0, p)
FR_HD Fp fp_canon(const Fp& a) {
    const u32 p[8] = FP_P_LIMBS;
    return fp_cond_sub(a, p);
}
// zero in the field: 0 or p
FR_HD bool fp_is_zero(const Fp& a) {
    const u32 p[8] = FP_P_LIMBS;
    u32 z = 0, e = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        z |= a.v[j];
        e |= a.v[j] ^ p[j];
    }
    return z == 0 || e == 0;
}
FR_HD bool fp_eq_raw(const Fp& a, const Fp& b) {
    u32 d = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) d |= a.v[j] ^ b.v[j];
    return d == 0;
}

// ---- Fp2 = Fp[u] / (u^2 + 1): the coordinates of G2 (gnark-crypto's fptower.E2 {A0, A1}) ------------------------------
// Components stay in the lazy range [0, 2p) like every Fp value here.
struct Fp2 {
    Fp a0, a1;
};
FR_HD Fp2 fp2_zero() { return Fp2{fp_zero(), fp_zero()}; }
FR_HD Fp2 fp2_one() { return Fp2{fp_one(), fp_zero()}; }
FR_HD Fp2 fp2_add(const Fp2& a, const Fp2& b) { return Fp2{fp_add(a.a0, b.a0), fp_add(a.a1, b.a1)}; }
FR_HD Fp2 fp2_sub(const Fp2& a, const Fp2& b) { return Fp2{fp_sub(a.a0, b.a0), fp_sub(a.a1, b.a1)}; }
FR_HD Fp2 fp2_dbl(const Fp2& a) { return Fp2{fp_dbl(a.a0), fp_dbl(a.a1)}; }
FR_HD Fp2 fp2_neg(const Fp2& a) { return Fp2{fp_neg(a.a0), fp_neg(a.a1)}; }
FR_HD Fp2 fp2_canon(const Fp2& a) { return Fp2{fp_canon(a.a0), fp_canon(a.a1)}; }
FR_HD bool fp2_is_zero(const Fp2& a) { return fp_is_zero(a.a0) && fp_is_zero(a.a1); }
// Karatsuba: three Fp products
FR_HD Fp2 fp2_mul(const Fp2& a, const Fp2& b) {
    const Fp v0 = fp_mul(a.a0, b.a0), v1 = fp_mul(a.a1, b.a1);
    const Fp s = fp_mul(fp_add(a.a0, a.a1), fp_add(b.a0, b.a1));
    return Fp2{fp_sub(v0, v1), fp_sub(fp_sub(s, v0), v1)};
}
// (a0 + a1)(a0 - a1) + 2 a0 a1 u: two Fp products
FR_HD Fp2 fp2_sqr(const Fp2& a) {
    const Fp c0 = fp_mul(fp_add(a.a0, a.a1), fp_sub(a.a0, a.a1));
    return Fp2{c0, fp_dbl(fp_mul(a.a0, a.a1))};
}
