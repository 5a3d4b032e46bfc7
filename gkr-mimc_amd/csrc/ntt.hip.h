// ntt.hip.h -- number-theoretic transforms over BN254 Fr on limb planes, for computeH: the H part of Groth16's Krs as the
// reference's prover gadget computes it (prover/gadget/prove.go:308-359):
//     a, b, c  <- FFTInverse(., DIF, 0)          three inverse FFTs, output in bit-reversed order
//     a, b, c  <- FFT(., DIT, 1)                 three FFTs on the coset u * <g> (u of order 2n, u^2 = g), natural order out
//     a        <- (a * b - c) * (-2)^-1          pointwise (Z = X^n - 1 is -2 on the coset)
//     a        <- FFTInverse(a, DIF, 1)          inverse coset FFT, bit-reversed order out;  then FromMont
// (gnark-crypto's fft.Domain, an un-vendored dependency: the algorithm is restated by the test oracle, "parity
// unpinned").  Radix-2 butterflies, in place.
//
// EVERY pass over HBM is an LDS pass (round 4; round 3 held the large strides in registers, three stages per pass, and ran
// at half the field-product rate of the sumcheck kernels: two waves per SIMD, loads, arithmetic and stores of a wave one
// after the other).  A workgroup of 512 lanes owns a TILE of 2048 elements = 2^lrows rows at stride 2^lgQ x 2^lcols
// consecutive elements, loads it into LDS (64 KiB + padding: two workgroups per CU, four waves per SIMD), runs lrows
// butterfly stages on the row index in sub-passes of two stages (four elements per lane in registers: ~110 VGPRs), and
// stores it back.  The contiguous tile (lcols = 0, lgQ = 0) takes the eleven stages whose butterflies stay inside 2048
// consecutive elements -- the LAST stages of a DIF transform, the FIRST of a DIT transform; the larger distances go in
// tiles of up to 128 rows x 16 consecutive elements (256-byte segments per plane and row).  A 2^24-point transform is
// THREE passes over HBM (7 + 6 + 11 stages) instead of six.
// The element-wise factors ride on passes that exist anyway: the 1/n of the first inverse transforms and the coset shift
// u^rev(p) are ONE factor applied when the first DIT pass loads; the pointwise step a*b - c is done by the first
// pass of the last transform when it loads (three arrays in, one out); its (-2)^-1 (round 6), the final 1/n, the inverse coset
// shift and FromMont are one factor (kept in regular form, so the Montgomery product leaves the Montgomery domain) applied when the
// last pass stores.
// No MFMA: exact modular arithmetic.  One twiddle table omega^i, i < n/2, serves every stage and both directions
// (omega^-i = -omega^(n/2 - i)); a group of four elements loads two twiddles and derives the third by a product with a
// launch-wide constant (the 96-limb-product form fr_mul_const2_raw).
#pragma once
#include "kernels.hip.h"

struct NttPassArgs {
    Planes d[3];          // the arrays of this launch (blockIdx.y selects; pre == 3 reads all three and writes d[0]), in place
    CPlanes tw;           // omega^i, i < n/2 (Montgomery form)
    int logn, s0;         // transform size, first stage of this pass
    int lrows, lcols, lgQ;   // tile: 2^lrows rows at stride 2^lgQ, 2^lcols consecutive elements per row
    CPlanes coset;        // pre 2 / post 3: the per-position factor table of the domain (k_ntt_coset_table)
    int pre, post;        // 0 none | pre 2: x *= coset[p] = u^rev(p) / n                       (coset shift and the 1/n of step 1)
                          //        | pre 3: x = d[0][p] * d[1][p] - d[2][p]                    (pointwise step, prove.go:341-347; its (-2)^-1: post 3)
                          //        | post 3: x *= coset[p] = (-2)^-1 u^-rev(p) / n in REGULAR form    (the product leaves Montgomery form)
};

// Twiddles are kept PER STAGE SHIFT s: T_s[j] = omega^(j * 2^s), j <= n / 2^(s+1) (the last entry is -1), T_s starting at
// entry n - n / 2^s + s of one allocation of n + logn + 1 entries.  A stage whose twiddle exponents are multiples of 2^s then
// reads CONSECUTIVE entries for consecutive positions, whatever s is (one table omega^i, i < n/2, read at stride 2^s costs a
// cache line per lane from s = 2 on).  A forward stage loads w = omega^e; an inverse stage loads w' = omega^(n/2 - e) =
// -omega^-e and the butterflies swap the operands of their subtraction instead of negating.
__device__ __forceinline__ Fr ntt_twiddle(const CPlanes& tw, int logn, bool inverse, int shift, size_t j) {      // omega^(+-(j << shift)), j < n / 2^(shift+1)
    const size_t n = (size_t)1 << logn, off = n - (n >> shift) + shift, cnt = n >> (shift + 1);
    return ld_fr(tw.lo, tw.hi, off + (inverse ? cnt - j : j));
}
__device__ __forceinline__ size_t ntt_rev(size_t p, int logn) { return logn ? (size_t)(__brevll((unsigned long long)p) >> (64 - logn)) : 0; }

// Lazy range of the butterflies: every value in LDS and in registers is in [0, 2q); what comes from HBM is canonical and what
// goes back is made canonical (fr_reduce_once).  Twiddles are canonical, so (x - y + 2q) * w < 4q * q stays an exact lazy
// Montgomery product below 2q.
#define FR_2Q_LIMBS {0xe0000002u, 0x87c3eb27u, 0xf372e122u, 0x5067d090u, 0x0302b0bau, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u}
__device__ __forceinline__ Fr ntt_add2q(const Fr& a, const Fr& b) {          // a + b mod 2q
    const u32 q2[8] = FR_2Q_LIMBS;
    const Fr s = fr_add_raw(a, b);
    u32 d[8], br = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) d[j] = fr_subb(s.v[j], q2[j], br, &br);
    Fr r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.v[j] = br ? s.v[j] : d[j];
    return r;
}
__device__ __forceinline__ Fr ntt_sub2q(const Fr& a, const Fr& b) {          // a - b mod 2q
    const u32 q2[8] = FR_2Q_LIMBS;
    u32 s[8], br = 0, c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = fr_subb(a.v[j], b.v[j], br, &br);
    const u32 mask = 0u - br;
    Fr r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.v[j] = fr_addc(s[j], q2[j] & mask, c, &c);
    return r;
}
__device__ __forceinline__ Fr ntt_sub_plus2q(const Fr& a, const Fr& b) {     // a - b + 2q in (0, 4q): no comparison
    const u32 q2[8] = FR_2Q_LIMBS;
    u32 s[8], br = 0, c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = fr_subb(a.v[j], b.v[j], br, &br);
    Fr r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.v[j] = fr_addc(s[j], q2[j], c, &c);
    return r;
}

// element-wise factors of the first load / last store (p = global position); loads return values below 2q
__device__ __forceinline__ Fr ntt_load(const NttPassArgs& a, const Planes& d, size_t p) {
    if (a.pre == 3) {
        const Fr x = ld_fr(a.d[0].lo, a.d[0].hi, p), y = ld_fr(a.d[1].lo, a.d[1].hi, p), z = ld_fr(a.d[2].lo, a.d[2].hi, p);
        return ntt_sub2q(fr_mont_mul_raw(x, y), z);      // x y - c mod 2q; the (-2)^-1 of the step is part of the last store's factor (the transform between is linear)
    }
    const Fr x = ld_fr(d.lo, d.hi, p);
    if (a.pre != 2) return x;
    return fr_mont_mul_raw(x, ld_fr(a.coset.lo, a.coset.hi, p));
}
__device__ __forceinline__ Fr ntt_store_value(const NttPassArgs& a, size_t p, const Fr& x) {      // x < 2q -> canonical
    if (a.post == 3) return fr_reduce_once(fr_mont_mul_raw(x, ld_fr(a.coset.lo, a.coset.hi, p)));   // Montgomery x regular: leaves Montgomery form
    return fr_reduce_once(x);
}

// R <= 2 radix-2 stages on the 2^R elements x[t] of one group (element t sits at position base + t * 2^lg_q of a transform of
// 2^lgn points; low = base mod 2^lg_q).  The twiddle of the pair (t, t + dist) of stage r is
//     DIF:  omega^((low + (t mod 2^(R-1-r)) * q) * n / (2 d_r))  =  W_r * zeta^(t mod 2^(R-1-r)),   W_r = omega^(low << (lgn - lg_q - R + r))
//     DIT:  omega^((low + (t mod 2^r) * q) * n / (2 d_r))        =  V_r * zeta^(t mod 2^r),          V_r = omega^(low << (lgn - 1 - lg_q - r))
// with zeta the primitive 4th root of unity of the direction: omega^(e + n/4) sits a quarter of the stage's table further, so a
// group of four loads its three twiddles, all coalesced (deriving the third by a product with the constant zeta instead:
// 16.3 against 13.8 ms per computeH at 2^24).
// TRIV (round 6): the group sits at position stride 1 with low = 0 -- the FIRST sub-pass of a DIT transform, the LAST of a DIF
// transform.  Every twiddle of such a group is omega^0 except the second one of the two-twiddle stage (the 4th root of unity):
// a butterfly by omega^0 is an addition and a subtraction, whatever the direction (forward: (x + y, (x - y) * 1); inverse: the
// loaded -omega^0 = -1 with the swapped subtraction gives the same pair), so three of the four products of a two-stage group
// -- or the only one of a one-stage group -- and their twiddle loads are not issued: 1.5 of a transform's 24 stages at 2^24.
template <int R, bool DIT, bool INV, bool TRIV = false>
__device__ __forceinline__ void ntt_stages(const NttPassArgs& a, int lgn, size_t low, int lg_q, Fr (&x)[1 << R]) {
    static_assert(R == 1 || R == 2, "sub-passes of one or two stages");
    constexpr int E = 1 << R;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int dist = DIT ? (1 << r) : (1 << (R - 1 - r));                 // in units of q
        const bool two = DIT ? (r == 1) : (R == 2 && r == 0);                  // the stage with two distinct twiddles
        const int shift = DIT ? lgn - 1 - lg_q - r : lgn - lg_q - R + r;       // the twiddle exponents of this stage are (low + ...) << shift
        Fr w[2];
        if (!TRIV) w[0] = ntt_twiddle(a.tw, lgn, INV, shift, low);
        if (two) w[1] = ntt_twiddle(a.tw, lgn, INV, shift, low + ((size_t)1 << (lgn - shift - 2)));      // times the 4th root of unity: a quarter of T_s further
#pragma unroll
        for (int t = 0; t < E; t++) {
            if (t & dist) continue;
            const int wi = two ? (t & (dist - 1)) : 0;
            if (TRIV && wi == 0) {                                       // twiddle omega^0
                const Fr s = ntt_add2q(x[t], x[t + dist]), d = ntt_sub2q(x[t], x[t + dist]);
                x[t] = s;
                x[t + dist] = d;
                continue;
            }
            const Fr& wk = w[wi];
            if (DIT) {
                const Fr y = fr_mont_mul_raw(x[t + dist], wk);           // inverse: -(x_hi * omega^-e)
                const Fr s = ntt_add2q(x[t], y), d = ntt_sub2q(x[t], y);
                x[t] = INV ? d : s;
                x[t + dist] = INV ? s : d;
            } else {
                const Fr s = ntt_add2q(x[t], x[t + dist]);
                const Fr d = INV ? ntt_sub_plus2q(x[t + dist], x[t]) : ntt_sub_plus2q(x[t], x[t + dist]);
                x[t + dist] = fr_mont_mul_raw(d, wk);
                x[t] = s;
            }
        }
    }
}

#define GKR_NTT_LTILE 11
#define GKR_NTT_TILE (1 << GKR_NTT_LTILE)
#define GKR_NTT_WG 512
struct NttTileShared {
    uint4 lo[GKR_NTT_TILE + GKR_NTT_TILE / 8], hi[GKR_NTT_TILE + GKR_NTT_TILE / 8];
};
// LDS index i lives at i + (i >> 3): lanes that walk row groups with a power-of-two stride meet different banks
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
// R stages on the row index of the tile, starting at the pass's local stage ls0; tile_low = the tile's column offset
// (global position bits below lgQ that all its elements share, plus the column)
template <int R, bool DIT, bool INV, bool TRIV = false>
__device__ __forceinline__ void ntt_tile_subpass(const NttPassArgs& a, NttTileShared& sh, int ls0, size_t tile_low) {
    constexpr int E = 1 << R;
    const int lgqr = DIT ? ls0 : a.lrows - ls0 - R;            // log2 of the element stride of a group, in rows
    const int cmask = (1 << a.lcols) - 1;
    const int ngroups = 1 << (a.lrows + a.lcols - R);
    for (int idx = threadIdx.x; idx < ngroups; idx += GKR_NTT_WG) {
        const int c = idx & cmask, g = idx >> a.lcols;
        const int low_r = g & ((1 << lgqr) - 1), high_r = g >> lgqr;
        const int r_base = (high_r << (lgqr + R)) | low_r;
        Fr x[E];
#pragma unroll
        for (int t = 0; t < E; t++) x[t] = ntt_lds_ld(sh, ((r_base + (t << lgqr)) << a.lcols) | c);
        ntt_stages<R, DIT, INV, TRIV>(a, a.logn, ((size_t)low_r << a.lgQ) | tile_low | (size_t)c, a.lgQ + lgqr, x);
#pragma unroll
        for (int t = 0; t < E; t++) ntt_lds_st(sh, ((r_base + (t << lgqr)) << a.lcols) | c, x[t]);
    }
}
// one pass: stages s0 .. s0 + lrows - 1 of the transform on every tile.  blockIdx.x = tile, blockIdx.y = array.
template <bool DIT, bool INV>
__global__ void __launch_bounds__(GKR_NTT_WG, 2) k_ntt_tile(NttPassArgs a) {
    __shared__ NttTileShared sh;
    const int lt = a.lrows + a.lcols;                                     // log2 of the tile (11 unless the transform is smaller)
    const size_t per_hi = (size_t)1 << (a.lgQ - a.lcols);                 // tiles that share the position bits above the rows
    const size_t hi = blockIdx.x / per_hi, cblk = blockIdx.x % per_hi;
    const size_t p0 = (hi << (a.lgQ + a.lrows)) | (cblk << a.lcols);
    const int cmask = (1 << a.lcols) - 1;
    const Planes d = a.d[blockIdx.y];
    for (int i = threadIdx.x; i < (1 << lt); i += GKR_NTT_WG) {
        const size_t p = p0 | ((size_t)(i >> a.lcols) << a.lgQ) | (size_t)(i & cmask);
        ntt_lds_st(sh, i, ntt_load(a, d, p));
    }
    __syncthreads();
    // The contiguous tile (stride 1, one column: the transform's first DIT / last DIF stages) has one sub-pass whose groups sit at
    // element stride 1 -- twiddles omega^0 and the 4th root of unity only (ntt_stages: TRIV).  A DIF tile with an odd number of
    // stages runs its single stage FIRST, so that this last sub-pass is a two-stage one (three of four products saved, not one of two).
    const bool contiguous = a.lgQ == 0 && a.lcols == 0;        // uniform over the launch
    int ls0 = 0;
    if (!DIT && (a.lrows & 1)) {
        if (contiguous && a.lrows == 1) ntt_tile_subpass<1, DIT, INV, true>(a, sh, 0, cblk << a.lcols);
        else ntt_tile_subpass<1, DIT, INV>(a, sh, 0, cblk << a.lcols);
        __syncthreads();
        ls0 = 1;
    }
    for (; ls0 < a.lrows; ls0 += 2) {
        const bool triv = contiguous && (DIT ? ls0 == 0 : ls0 + 2 >= a.lrows);
        if (a.lrows - ls0 >= 2) {
            if (triv) ntt_tile_subpass<2, DIT, INV, true>(a, sh, ls0, cblk << a.lcols);
            else ntt_tile_subpass<2, DIT, INV>(a, sh, ls0, cblk << a.lcols);
        } else {            // (DIT only: an odd number of stages ends with the single one)
            if (triv) ntt_tile_subpass<1, DIT, INV, true>(a, sh, ls0, cblk << a.lcols);
            else ntt_tile_subpass<1, DIT, INV>(a, sh, ls0, cblk << a.lcols);
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < (1 << lt); i += GKR_NTT_WG) {
        const size_t p = p0 | ((size_t)(i >> a.lcols) << a.lgQ) | (size_t)(i & cmask);
        st_fr(d.lo, d.hi, p, ntt_store_value(a, p, ntt_lds_ld(sh, i)));
    }
}

// twiddle tables: omega^i = hi[i >> l0] * lo[i & (2^l0 - 1)] for i < n/2 (the two small tables computed on the host), -1 for
// i = n/2; entry (s, j) of the per-stage layout is omega^(j << s)
GKR_KERNEL void __launch_bounds__(GKR_BLOCK) k_ntt_twiddles(Planes tw, CPlanes lo, CPlanes hi, int l0, int logn) {
    const size_t n = (size_t)1 << logn, total = n + logn;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
        int s = 0;
        while (s + 1 <= logn - 1 && g >= n - (n >> (s + 1)) + (s + 1)) s++;       // the table this entry belongs to
        const size_t j = g - (n - (n >> s) + s), i = j << s;
        // The tables hold n + logn - 1 entries; entry n + logn - 1 is padding (j = 2 of the last table: i = n).  Until round 5 it
        // was computed like the others and read hi[n >> l0], TWICE the length of that table away: harmless where the bytes behind
        // the table are mapped, "Memory access fault by GPU" where they are not -- one run of the default bench line in ~12,
        // located with tools/alloc_trace.c (profiles/r05_anomalies.md (c)).
        Fr w;
        if (i < (n >> 1)) w = fr_mul(ld_fr(hi.lo, hi.hi, i >> l0), ld_fr(lo.lo, lo.hi, i & (((size_t)1 << l0) - 1)));
        else if (i == (n >> 1)) w = fr_sub(fr_zero(), fr_one());
        else w = fr_zero();
        st_fr(tw.lo, tw.hi, g, w);
    }
}
// The per-position factors of the coset transforms, as gnark-crypto's fft.Domain precomputes its CosetTable / CosetTableInv
// (here indexed by the position in the bit-reversed vector the factor is applied to, with the 1/n folded in):
//     fwd[p] = u^rev(p) * k0        inv[p] = u^-rev(p) * k0'         (u^e = omega^(e >> 1) * (e odd ? u : 1); k1 = k0 * u, k1' = k0' / u;
//     k0 = 1/n, k0' = (-2)^-1 / n in regular form: the constant of computeH's pointwise step rides on the last store)
GKR_KERNEL void __launch_bounds__(GKR_BLOCK) k_ntt_coset_table(Planes out, CPlanes tw, int logn, int inverse, Fr k0, Fr k1) {
    const size_t n = (size_t)1 << logn;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        const size_t e = ntt_rev(p, logn);
        Fr w = ld_fr(tw.lo, tw.hi, inverse ? (n >> 1) - (e >> 1) : (e >> 1));      // T_0; inverse: -omega^-(e >> 1)
        if (inverse) w = fr_sub(fr_zero(), w);
        st_fr(out.lo, out.hi, p, fr_mul(w, (e & 1) ? k1 : k0));
    }
}
// zero padding of an array from n to the domain size
GKR_KERNEL void __launch_bounds__(GKR_BLOCK) k_ntt_zero(Planes d, size_t from, size_t to) {
    for (size_t i = from + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < to; i += (size_t)gridDim.x * blockDim.x)
        st_fr(d.lo, d.hi, i, fr_zero());
}
