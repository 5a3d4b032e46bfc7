// ntt.hip.h -- number-theoretic transforms over BN254 Fr on limb planes, for computeH: the H part of Groth16's Krs as the
// reference's prover gadget computes it (prover/gadget/prove.go:308-359; SURVEY section 8 row f4, second half):
//     a, b, c  <- FFTInverse(., DIF, 0)          three inverse FFTs, output in bit-reversed order
//     a, b, c  <- FFT(., DIT, 1)                 three FFTs on the coset u * <g> (u of order 2n, u^2 = g), natural order out
//     a        <- (a * b - c) * (-2)^-1          pointwise (Z = X^n - 1 is -2 on the coset)
//     a        <- FFTInverse(a, DIF, 1)          inverse coset FFT, bit-reversed order out;  then FromMont
// (gnark-crypto's fft.Domain, an un-vendored dependency: the algorithm is restated by the test oracle, "parity
// unpinned").  Radix-2 butterflies, in place, up to three stages per pass held in registers (eight elements per lane): a
// 2^24-point transform is six passes over HBM instead of twenty-four (five register passes over the large strides, one
// LDS-tiled pass over the eleven stages whose butterflies stay inside 2048 consecutive elements).  The element-wise factors ride on passes that
// exist anyway: the 1/n of the first inverse transforms and the coset shift u^rev(p) are ONE factor applied when the first
// DIT pass loads; the pointwise step is done by the LAST DIT pass, which transforms the same index group of a, b and c and
// stores only (a*b - c) * (-2)^-1; the final 1/n, the inverse coset shift and FromMont are one factor (kept in regular
// form, so the Montgomery product leaves the Montgomery domain) applied when the last DIF pass stores.
// Measured (gkrhip_bench_compute_h, MI355X): 2^24 points 22.5 ms = 18 passes moving 44 GB at 1.95 TB/s (0.24 of the HBM
// peak): the transforms are bound by integer VALU issue like the sumcheck rounds, not by HBM -- ~16 field products per
// element and transform (12 butterflies + the products that derive a group's twiddles from three loaded ones) is ~2 x 10^9
// products per computeH at the ~10^11 products/s the field arithmetic sustains.  With one twiddle load per butterfly and
// every stage in register passes (the first version) the same computeH took 32 ms.
// No MFMA: exact modular arithmetic.  One twiddle table omega^i, i < n/2, serves every stage and both directions
// (omega^-i = -omega^(n/2 - i)).
#pragma once
#include "kernels.hip.h"

struct NttPassArgs {
    Planes d[3];          // the arrays of this launch (blockIdx.y selects; TRIPLE: all three in one lane), in place
    CPlanes tw;           // omega^i, i < n/2 (Montgomery form)
    CPlanes tw_tile;      // tile kernel: the twiddles of the 2^ltile-point domain, (omega^(n / 2^ltile))^i, i < 2^(ltile-1)
    int logn, s0;         // transform size, first stage of this pass
    int ltile;            // tile kernel: log2 of the tile (the last ltile DIF stages / first ltile DIT stages)
    int inverse;          // twiddles omega^-i
    int pre, post;        // 0 none | pre 2: x *= tw[e >> 1] * (e odd ? k1 : k0), e = rev(p)   (coset shift and 1/n)
                          //        | post 1: x *= k0 | post 3: x *= inv_tw[e >> 1] * (e odd ? k1 : k0) with k0, k1 in REGULAR form
    Fr k0, k1;
    Fr k2;                // TRIPLE: (-2)^-1
    Fr z[4];              // z[k] = zeta^k, zeta = omega^(+-n/8) the primitive 8th root of unity of this direction (z[0] = 1)
};

__device__ __forceinline__ Fr ntt_twiddle(const CPlanes& tw, int logn, bool inverse, size_t e) {      // e < n/2
    if (!inverse) return ld_fr(tw.lo, tw.hi, e);
    if (e == 0) return fr_one();
    return fr_sub(fr_zero(), ld_fr(tw.lo, tw.hi, ((size_t)1 << (logn - 1)) - e));   // omega^-e = -omega^(n/2 - e)
}
__device__ __forceinline__ size_t ntt_rev(size_t p, int logn) { return logn ? (size_t)(__brevll((unsigned long long)p) >> (64 - logn)) : 0; }

// element-wise factors of the first load / last store (p = global position)
__device__ __forceinline__ Fr ntt_pre(const NttPassArgs& a, size_t p, const Fr& x) {
    if (a.pre != 2) return x;
    const size_t e = ntt_rev(p, a.logn);
    return fr_mul(x, fr_mul(ld_fr(a.tw.lo, a.tw.hi, e >> 1), (e & 1) ? a.k1 : a.k0));
}
__device__ __forceinline__ Fr ntt_post(const NttPassArgs& a, size_t p, const Fr& x) {
    if (a.post == 1) return fr_mul(x, a.k0);
    if (a.post == 3) {
        const size_t e = ntt_rev(p, a.logn);
        const Fr f = fr_mul(ntt_twiddle(a.tw, a.logn, true, e >> 1), (e & 1) ? a.k1 : a.k0);   // regular form: Montgomery x regular
        return fr_mul(x, f);                                                                   // ... and the result leaves Montgomery form
    }
    return x;
}

// R radix-2 stages on the 2^R elements x[t] of one group (element t sits at position base + t * 2^lg_q of a transform of
// 2^lgn points whose twiddles are `tw`; low = base mod 2^lg_q).  The twiddle of the pair (t, t + dist) of stage r is
//     DIF:  omega^((low + (t mod 2^(R-1-r)) * q) * n / (2 d_r))  =  W_r * zeta_(R-r)^(t mod 2^(R-1-r)),   W_r = omega^(low << (s0 + r)) = W_0^(2^r)
//     DIT:  omega^((low + (t mod 2^r) * q) * n / (2 d_r))        =  V_r * zeta_(r+1)^(t mod 2^r),          V_r = omega^(low << (lgn - 1 - lg_q - r))
// with zeta_m a primitive 2^m-th root of unity -- launch-wide constants (a.z) -- so a group loads R twiddles (three for eight
// elements) and derives the other four by products with constants.  -DGKR_NTT_LOAD_TW reads all seven distinct entries
// omega^(e_r + k * n / 2^m) instead: measured 23.9 against 22.5 ms at 2^24 points (1.18 against 1.26 ms at 2^20): neither
// the products nor the loads alone bound the passes.
template <int R, bool DIT>
__device__ __forceinline__ void ntt_stages(const NttPassArgs& a, const CPlanes& tw, int lgn, size_t low, int lg_q, Fr (&x)[1 << R]) {
    constexpr int E = 1 << R;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int dist = DIT ? (1 << r) : (1 << (R - 1 - r));                 // in units of q
        const int m = DIT ? r + 1 : R - r;                                     // zeta_m: 2^m-th roots at this stage
        const size_t e = DIT ? (low << (lgn - 1 - lg_q - r)) : (low << (lgn - lg_q - R + r));
        Fr w[E / 2];                                                           // w[k] = omega^e * zeta_m^k = omega^(e + k * n / 2^m), k < 2^(m-1)
        w[0] = ntt_twiddle(tw, lgn, a.inverse != 0, e);
#pragma unroll
        for (int k = 1; k < (1 << (m - 1)); k++) {
#ifndef GKR_NTT_LOAD_TW
            w[k] = fr_mul(w[0], a.z[k << (3 - m)]);          // one load per stage, the rest by products with constants
#else
            w[k] = ntt_twiddle(tw, lgn, a.inverse != 0, e + ((size_t)k << (lgn - m)));      // seven loads per group of eight
#endif
        }
#pragma unroll
        for (int t = 0; t < E; t++) {
            if (t & dist) continue;
            const Fr& wk = w[t & (dist - 1)];
            if (DIT) {
                const Fr y = fr_mul(x[t + dist], wk);
                const Fr s = fr_add(x[t], y);
                x[t + dist] = fr_sub(x[t], y);
                x[t] = s;
            } else {
                const Fr s = fr_add(x[t], x[t + dist]);
                x[t + dist] = fr_mul(fr_sub(x[t], x[t + dist]), wk);
                x[t] = s;
            }
        }
    }
}

// the 2^R elements of group g of one array through R stages; x[] out (canonical elements throughout)
template <int R, bool DIT>
__device__ __forceinline__ void ntt_group(const NttPassArgs& a, const Planes& d, size_t base, size_t low, int lg_q, Fr (&x)[1 << R]) {
    constexpr int E = 1 << R;
    const size_t q = (size_t)1 << lg_q;
#pragma unroll
    for (int t = 0; t < E; t++) {
        const size_t p = base + (size_t)t * q;
        x[t] = ntt_pre(a, p, ld_fr(d.lo, d.hi, p));
    }
    ntt_stages<R, DIT>(a, a.tw, a.logn, low, lg_q, x);
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
        ntt_group<R, DIT>(a, a.d[0], base, low, lg_q, x);
        ntt_group<R, DIT>(a, a.d[1], base, low, lg_q, y);
#pragma unroll
        for (int t = 0; t < E; t++) x[t] = fr_mul(x[t], y[t]);
        ntt_group<R, DIT>(a, a.d[2], base, low, lg_q, y);
#pragma unroll
        for (int t = 0; t < E; t++) x[t] = fr_mul(fr_sub(x[t], y[t]), a.k2);      // (a*b - c) * (-2)^-1   (prove.go:341-347)
    } else {
        ntt_group<R, DIT>(a, a.d[blockIdx.y], base, low, lg_q, x);
    }
    const Planes out = a.d[TRIPLE ? 0 : blockIdx.y];
#pragma unroll
    for (int t = 0; t < E; t++) {
        const size_t p = base + (size_t)t * q;
        const Fr v = ntt_post(a, p, x[t]);
        st_fr(out.lo, out.hi, p, v);
    }
}

// ------------------------------------------------------------------------------------------------
// Tile kernel: the ltile stages whose butterflies stay inside an aligned block of 2^ltile consecutive elements -- the
// LAST stages of a DIF transform, the FIRST of a DIT transform -- in ONE pass over HBM: the block is loaded into LDS with
// fully coalesced accesses (the register passes would touch these stages with strides below a cache line), transformed
// there as a complete 2^ltile-point transform (its twiddles are the small domain's: omega^(n / 2^ltile) generates it) in
// sub-passes of up to three stages, and stored back.  LDS index i lives at i + (i >> 3): a lane that walks its group with
// stride 2^lg_q then meets its neighbours' elements in different banks for every lg_q.
// ------------------------------------------------------------------------------------------------
#define GKR_NTT_LTILE 11
#define GKR_NTT_TILE (1 << GKR_NTT_LTILE)
struct NttTileShared {
    uint4 lo[GKR_NTT_TILE + GKR_NTT_TILE / 8], hi[GKR_NTT_TILE + GKR_NTT_TILE / 8];
};
__device__ __forceinline__ int ntt_sw(int i) { return i + (i >> 3); }
__device__ __forceinline__ Fr ntt_lds_ld(const NttTileShared& sh, int i) {
    const uint4 a = sh.lo[ntt_sw(i)], b = sh.hi[ntt_sw(i)];
    Fr r = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
    return r;
}
__device__ __forceinline__ void ntt_lds_st(NttTileShared& sh, int i, const Fr& x) {
    sh.lo[ntt_sw(i)] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    sh.hi[ntt_sw(i)] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
}
template <int R, bool DIT>
__device__ __forceinline__ void ntt_tile_subpass(const NttPassArgs& a, NttTileShared& sh, int L, int ls0) {
    constexpr int E = 1 << R;
    const int lg_q = DIT ? ls0 : L - ls0 - R;
    for (int g = threadIdx.x; g < (1 << (L - R)); g += blockDim.x) {
        const int low = g & ((1 << lg_q) - 1), high = g >> lg_q;
        const int base = (high << (lg_q + R)) | low;
        Fr x[E];
#pragma unroll
        for (int t = 0; t < E; t++) x[t] = ntt_lds_ld(sh, base + (t << lg_q));
        ntt_stages<R, DIT>(a, a.tw_tile, L, (size_t)low, lg_q, x);
#pragma unroll
        for (int t = 0; t < E; t++) ntt_lds_st(sh, base + (t << lg_q), x[t]);
    }
}
template <bool DIT>
__global__ void __launch_bounds__(GKR_BLOCK) k_ntt_tile(NttPassArgs a) {
    __shared__ NttTileShared sh;
    const int L = a.ltile;
    const size_t tile0 = (size_t)blockIdx.x << L;
    const Planes d = a.d[blockIdx.y];
    for (int i = threadIdx.x; i < (1 << L); i += blockDim.x) ntt_lds_st(sh, i, ntt_pre(a, tile0 + i, ld_fr(d.lo, d.hi, tile0 + i)));
    __syncthreads();
    for (int ls0 = 0; ls0 < L; ls0 += 3) {
        const int R = min(3, L - ls0);
        if (R == 3) ntt_tile_subpass<3, DIT>(a, sh, L, ls0);
        else if (R == 2) ntt_tile_subpass<2, DIT>(a, sh, L, ls0);
        else ntt_tile_subpass<1, DIT>(a, sh, L, ls0);
        __syncthreads();
    }
    for (int i = threadIdx.x; i < (1 << L); i += blockDim.x) {
        const Fr v = ntt_post(a, tile0 + i, ntt_lds_ld(sh, i));
        st_fr(d.lo, d.hi, tile0 + i, v);
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
