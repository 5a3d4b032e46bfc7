// ntt.hip.h -- number-theoretic transforms over BN254 Fr on limb planes, for computeH: the H part of Groth16's Krs as the
// reference's prover gadget computes it (prover/gadget/prove.go:308-359; SURVEY section 8 row f4, second half):
//     a, b, c  <- FFTInverse(., DIF, 0)          three inverse FFTs, output in bit-reversed order
//     a, b, c  <- FFT(., DIT, 1)                 three FFTs on the coset u * <g> (u of order 2n, u^2 = g), natural order out
//     a        <- (a * b - c) * (-2)^-1          pointwise (Z = X^n - 1 is -2 on the coset)
//     a        <- FFTInverse(a, DIF, 1)          inverse coset FFT, bit-reversed order out;  then FromMont
// (gnark-crypto's fft.Domain, an un-vendored dependency: the algorithm is restated by the test oracle, "parity
// unpinned").  Radix-2 butterflies, in place, up to three stages per pass held in registers (eight elements per lane): a
// 2^24-point transform is eight passes over HBM instead of twenty-four.  The element-wise factors ride on passes that
// exist anyway: the 1/n of the first inverse transforms and the coset shift u^rev(p) are ONE factor applied when the first
// DIT pass loads; the pointwise step is done by the LAST DIT pass, which transforms the same index group of a, b and c and
// stores only (a*b - c) * (-2)^-1; the final 1/n, the inverse coset shift and FromMont are one factor (kept in regular
// form, so the Montgomery product leaves the Montgomery domain) applied when the last DIF pass stores.
// No MFMA: exact modular arithmetic.  One twiddle table omega^i, i < n/2, serves every stage and both directions
// (omega^-i = -omega^(n/2 - i)).
#pragma once
#include "kernels.hip.h"

struct NttPassArgs {
    Planes d[3];          // the arrays of this launch (blockIdx.y selects; TRIPLE: all three in one lane), in place
    CPlanes tw;           // omega^i, i < n/2 (Montgomery form)
    int logn, s0;         // transform size, first stage of this pass
    int inverse;          // twiddles omega^-i
    int pre, post;        // 0 none | pre 2: x *= tw[e >> 1] * (e odd ? k1 : k0), e = rev(p)   (coset shift and 1/n)
                          //        | post 1: x *= k0 | post 3: x *= inv_tw[e >> 1] * (e odd ? k1 : k0) with k0, k1 in REGULAR form
    Fr k0, k1;
    Fr k2;                // TRIPLE: (-2)^-1
};

__device__ __forceinline__ Fr ntt_twiddle(const CPlanes& tw, int logn, bool inverse, size_t e) {      // e < n/2
    if (!inverse) return ld_fr(tw.lo, tw.hi, e);
    if (e == 0) return fr_one();
    return fr_sub(fr_zero(), ld_fr(tw.lo, tw.hi, ((size_t)1 << (logn - 1)) - e));   // omega^-e = -omega^(n/2 - e)
}
__device__ __forceinline__ size_t ntt_rev(size_t p, int logn) { return logn ? (size_t)(__brevll((unsigned long long)p) >> (64 - logn)) : 0; }

// the 2^R elements of group g of one array through R stages; x[] in, x[] out (canonical elements throughout)
template <int R, bool DIT>
__device__ __forceinline__ void ntt_group(const NttPassArgs& a, const Planes& d, size_t base, int lg_q, Fr (&x)[1 << R]) {
    constexpr int E = 1 << R;
    const size_t q = (size_t)1 << lg_q;
#pragma unroll
    for (int t = 0; t < E; t++) {
        const size_t p = base + (size_t)t * q;
        x[t] = ld_fr(d.lo, d.hi, p);
        if (a.pre == 2) {
            const size_t e = ntt_rev(p, a.logn);
            const Fr f = fr_mul(ld_fr(a.tw.lo, a.tw.hi, e >> 1), (e & 1) ? a.k1 : a.k0);
            x[t] = fr_mul(x[t], f);
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int dist = DIT ? (1 << r) : (1 << (R - 1 - r));                 // in units of q
        const int lg_d = lg_q + (DIT ? r : R - 1 - r);                         // log2 of the butterfly distance
        const int sh = a.logn - 1 - lg_d;                                      // n / (2 d)
#pragma unroll
        for (int t = 0; t < E; t++) {
            if (t & dist) continue;
            const size_t p = base + (size_t)t * q;
            const size_t e = (p & (((size_t)1 << lg_d) - 1)) << sh;
            const Fr w = ntt_twiddle(a.tw, a.logn, a.inverse != 0, e);
            if (DIT) {
                const Fr y = fr_mul(x[t + dist], w);
                const Fr s = fr_add(x[t], y);
                x[t + dist] = fr_sub(x[t], y);
                x[t] = s;
            } else {
                const Fr s = fr_add(x[t], x[t + dist]);
                x[t + dist] = fr_mul(fr_sub(x[t], x[t + dist]), w);
                x[t] = s;
            }
        }
    }
}

template <int R, bool DIT, bool TRIPLE>
__global__ void __launch_bounds__(GKR_BLOCK) k_ntt_pass(NttPassArgs a) {
    constexpr int E = 1 << R;
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ((size_t)1 << (a.logn - R))) return;
    // DIF: the pass covers distances n >> (s0+1) .. n >> (s0+R), element stride q = n >> (s0+R);  DIT: distances 1 << s0 ..,
    // element stride q = 1 << s0.  Group g -> base index: R zero bits inserted at bit log2(q).
    const int lg_q = DIT ? a.s0 : a.logn - a.s0 - R;
    const size_t low = g & (((size_t)1 << lg_q) - 1), high = g >> lg_q;
    const size_t base = (high << (lg_q + R)) | low;
    const size_t q = (size_t)1 << lg_q;
    Fr x[E];
    if (TRIPLE) {
        Fr y[E];
        ntt_group<R, DIT>(a, a.d[0], base, lg_q, x);
        ntt_group<R, DIT>(a, a.d[1], base, lg_q, y);
#pragma unroll
        for (int t = 0; t < E; t++) x[t] = fr_mul(x[t], y[t]);
        ntt_group<R, DIT>(a, a.d[2], base, lg_q, y);
#pragma unroll
        for (int t = 0; t < E; t++) x[t] = fr_mul(fr_sub(x[t], y[t]), a.k2);      // (a*b - c) * (-2)^-1   (prove.go:341-347)
    } else {
        ntt_group<R, DIT>(a, a.d[blockIdx.y], base, lg_q, x);
    }
    const Planes out = a.d[TRIPLE ? 0 : blockIdx.y];
#pragma unroll
    for (int t = 0; t < E; t++) {
        const size_t p = base + (size_t)t * q;
        Fr v = x[t];
        if (a.post == 1) {
            v = fr_mul(v, a.k0);
        } else if (a.post == 3) {
            const size_t e = ntt_rev(p, a.logn);
            const Fr f = fr_mul(ntt_twiddle(a.tw, a.logn, true, e >> 1), (e & 1) ? a.k1 : a.k0);   // regular form: Montgomery x regular
            v = fr_mul(v, f);                                                       // ... and the result leaves Montgomery form
        }
        st_fr(out.lo, out.hi, p, v);
    }
}

// twiddle table: tw[i] = hi[i >> l0] * lo[i & (2^l0 - 1)], the two small tables computed on the host
__global__ void __launch_bounds__(GKR_BLOCK) k_ntt_twiddles(Planes tw, CPlanes lo, CPlanes hi, int l0, size_t n_half) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_half; i += (size_t)gridDim.x * blockDim.x)
        st_fr(tw.lo, tw.hi, i, fr_mul(ld_fr(hi.lo, hi.hi, i >> l0), ld_fr(lo.lo, lo.hi, i & (((size_t)1 << l0) - 1))));
}
// zero padding of an array from n to the domain size
__global__ void __launch_bounds__(GKR_BLOCK) k_ntt_zero(Planes d, size_t from, size_t to) {
    for (size_t i = from + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < to; i += (size_t)gridDim.x * blockDim.x)
        st_fr(d.lo, d.hi, i, fr_zero());
}
