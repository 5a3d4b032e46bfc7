// g1.hip.h -- BN254 G1 and G2 arithmetic and the kernels of the multi-scalar multiplication sum_i [s_i] P_i on gfx950: what
// gnark-crypto's (*G1Jac).MultiExp / (*G1Affine).MultiExp / (*G2Jac).MultiExp compute for the reference's Groth16 prover
// (prover/gadget/prove.go:76,91 krsNotGkr / KrsPrivNotGkr; :189 bs1, :202 ar, :221 krs2; :277 Bs on G2; SURVEY section 8 row
// f4) and bn254.BatchScalarMultiplicationG1 (prove.go:177).  gnark-crypto is an un-vendored dependency of the reference
// (v0.6.1-0.20220110145513-493bb1c180d9): the RESULT is a group element, so parity is defined by the mathematics -- the
// affine coordinates of the sum are unique -- and pinned by the test oracle's big-integer double-and-add ("parity
// unpinned" against bytes of the Go binary, like computeH).
//
// Pippenger's bucket method laid out for a GPU instead of gnark-crypto's one-goroutine-per-window loop:
//   1. every scalar (regular form, < q < 2^254) is cut into W = ceil(255 / c) SIGNED digits of c bits
//      (d in [-2^(c-1), 2^(c-1)]: half the buckets, the sign negates the point's y), once, into 16-bit planes (k_msm_digits);
//   2. the (window, |digit|) pairs are counting-sorted with every atomic in LDS: a workgroup owns one window of one chunk
//      of the scalars and keeps that window's whole histogram (128 KiB at c = 16) in gfx950's 160 KiB LDS
//      (k_msm_hist, k_msm_totals, k_msm_scan, k_msm_offsets, k_msm_scatter) -- never an atomic on a point; from 2^20 points
//      in two levels (coarse bins, then slices of a bin: k_msm_scatter_coarse, k_msm_refine_*), each scatter staged in LDS;
//   3. ONE LANE PER BUCKET (the buckets of a window handed out in order of size, the top window first: k_msm_order) adds its run of points with
//      mixed additions into an extended-Jacobian (XYZZ) accumulator (k_msm_accumulate: 8 M + 2 S per point, the bulk of the
//      work: W * n additions); the few buckets far above the mean
//      (skewed scalars: the 0/1 wires of a real witness put most points of window 0 into bucket 1) are cut into segments,
//      a workgroup per segment (k_msm_accumulate_big, k_msm_big_combine);
//   4. sum_b (b + 1) B_b per window by running sums over chunks of consecutive buckets, one lane per chunk, the chunk's
//      offset applied by a short double-and-add (k_msm_reduce_chunks), then a workgroup tree per window (k_msm_reduce_windows);
//   5. the W window sums go to the host, which combines them by Horner's rule (c doublings per window) and converts to
//      affine: ~270 group operations, microseconds on a CPU core, milliseconds for a lone GPU lane.
// Coordinates live in the lazy range [0, 2p) of fp_bn254.h.  No MFMA: exact 254-bit modular arithmetic.
// Points in HBM keep gnark's G1Affine image (X, Y: 2 x 32 B Montgomery, infinity = (0, 0)): a lane gathers one point as four
// 16-byte loads of one 64-byte line; buckets and partial sums are limb planes (coalesced across lanes).
// Everything from the point arithmetic down is written once over a coordinate-field policy: Fp for G1, Fp2 for G2.
#pragma once
#include "fp_bn254.h"
#include "kernels.hip.h"

// ---- coordinate fields ------------------------------------------------------------------------------------------------
// The curve code below is written once, over a field policy F: FpF for G1 (coordinates in Fp) and Fp2F for G2 (coordinates
// in Fp2 = Fp[u]/(u^2 + 1), gnark-crypto's fptower.E2; the curve is the sextic twist y^2 = x^3 + 3/(9 + u), whose constant
// the addition formulas never use).  F::W16 = 16-byte words per element in HBM.
__device__ __forceinline__ Fp fp_from4(const uint4& a, const uint4& b) {
    Fp r = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
    return r;
}
// a^(p-2) (Fermat): the exponent is a compile-time constant, the loop is uniform
__device__ __forceinline__ Fp fp_inv(const Fp& a) {
    const u32 e[8] = {FPQ0 - 2u, FPQ1, FPQ2, FPQ3, FPQ4, FPQ5, FPQ6, FPQ7};
    Fp r = fp_one();
    for (int i = 253; i >= 0; i--) {
        r = fp_sqr(r);
        if ((e[i >> 5] >> (i & 31)) & 1u) r = fp_mul(r, a);
    }
    return r;
}
struct FpF {
    typedef Fp T;
    static const int W16 = 2;
    static __device__ __forceinline__ T ld(const uint4* __restrict__ p, size_t stride) { return fp_from4(p[0], p[stride]); }
    static __device__ __forceinline__ void st(uint4* __restrict__ p, size_t stride, const T& x) {
        p[0] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
        p[stride] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
    }
    static __device__ __forceinline__ T zero() { return fp_zero(); }
    static __device__ __forceinline__ T one() { return fp_one(); }
    static __device__ __forceinline__ T mul(const T& a, const T& b) { return fp_mul(a, b); }
    static __device__ __forceinline__ T sqr(const T& a) { return fp_sqr(a); }
    static __device__ __forceinline__ T add(const T& a, const T& b) { return fp_add(a, b); }
    static __device__ __forceinline__ T sub(const T& a, const T& b) { return fp_sub(a, b); }
    static __device__ __forceinline__ T dbl(const T& a) { return fp_dbl(a); }
    static __device__ __forceinline__ T neg(const T& a) { return fp_neg(a); }
    static __device__ __forceinline__ T canon(const T& a) { return fp_canon(a); }
    static __device__ __forceinline__ bool is_zero(const T& a) { return fp_is_zero(a); }
    static __device__ __forceinline__ T inv(const T& a) { return fp_inv(a); }
};
struct Fp2F {
    typedef Fp2 T;
    static const int W16 = 4;
    static __device__ __forceinline__ T ld(const uint4* __restrict__ p, size_t stride) {
        return Fp2{fp_from4(p[0], p[stride]), fp_from4(p[2 * stride], p[3 * stride])};
    }
    static __device__ __forceinline__ void st(uint4* __restrict__ p, size_t stride, const T& x) {
        FpF::st(p, stride, x.a0);
        FpF::st(p + 2 * stride, stride, x.a1);
    }
    static __device__ __forceinline__ T zero() { return fp2_zero(); }
    static __device__ __forceinline__ T one() { return fp2_one(); }
    static __device__ __forceinline__ T mul(const T& a, const T& b) { return fp2_mul(a, b); }
    static __device__ __forceinline__ T sqr(const T& a) { return fp2_sqr(a); }
    static __device__ __forceinline__ T add(const T& a, const T& b) { return fp2_add(a, b); }
    static __device__ __forceinline__ T sub(const T& a, const T& b) { return fp2_sub(a, b); }
    static __device__ __forceinline__ T dbl(const T& a) { return fp2_dbl(a); }
    static __device__ __forceinline__ T neg(const T& a) { return fp2_neg(a); }
    static __device__ __forceinline__ T canon(const T& a) { return fp2_canon(a); }
    static __device__ __forceinline__ bool is_zero(const T& a) { return fp2_is_zero(a); }
    static __device__ __forceinline__ T inv(const T& a) {        // conj(a) / (a0^2 + a1^2)
        const Fp n = fp_inv(fp_add(fp_sqr(a.a0), fp_sqr(a.a1)));
        return Fp2{fp_mul(a.a0, n), fp_neg(fp_mul(a.a1, n))};
    }
};

template <class F>
struct AffT {                  // gnark-crypto's G1Affine / G2Affine image: X, Y Montgomery, infinity = (0, 0)
    typename F::T x, y;
};
template <class F>
struct XyzzT {                 // extended Jacobian: x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2; infinity: ZZ = 0
    typename F::T x, y, zz, zzz;
};
typedef AffT<FpF> G1Aff;
typedef XyzzT<FpF> G1X;
typedef AffT<Fp2F> G2Aff;
typedef XyzzT<Fp2F> G2X;

// point i of an array of affine points (2 * W16 consecutive 16-byte words)
template <class F>
__device__ __forceinline__ AffT<F> ec_ld_aff(const uint4* __restrict__ pts, size_t i) {
    const uint4* p = pts + 2 * F::W16 * i;
    AffT<F> r;
    r.x = F::ld(p, 1);
    r.y = F::ld(p + F::W16, 1);
    return r;
}
template <class F>
__device__ __forceinline__ void ec_st_aff(uint4* __restrict__ pts, size_t i, const AffT<F>& a) {
    uint4* p = pts + 2 * F::W16 * i;
    F::st(p, 1, a.x);
    F::st(p + F::W16, 1, a.y);
}
// XYZZ points as 4 * W16 planes of 16-byte words, plane k of element t at base[k * stride + t]
struct XPlanes {
    uint4* base;
    size_t stride;
};
template <class F>
__device__ __forceinline__ XyzzT<F> ecx_ld(const XPlanes& pl, size_t t) {
    XyzzT<F> r;
    r.x = F::ld(pl.base + t, pl.stride);
    r.y = F::ld(pl.base + (size_t)F::W16 * pl.stride + t, pl.stride);
    r.zz = F::ld(pl.base + (size_t)2 * F::W16 * pl.stride + t, pl.stride);
    r.zzz = F::ld(pl.base + (size_t)3 * F::W16 * pl.stride + t, pl.stride);
    return r;
}
template <class F>
__device__ __forceinline__ void ecx_st(const XPlanes& pl, size_t t, const XyzzT<F>& p) {
    F::st(pl.base + t, pl.stride, p.x);
    F::st(pl.base + (size_t)F::W16 * pl.stride + t, pl.stride, p.y);
    F::st(pl.base + (size_t)2 * F::W16 * pl.stride + t, pl.stride, p.zz);
    F::st(pl.base + (size_t)3 * F::W16 * pl.stride + t, pl.stride, p.zzz);
}

template <class F>
__device__ __forceinline__ bool ec_aff_is_inf(const AffT<F>& a) { return F::is_zero(a.x) && F::is_zero(a.y); }
template <class F>
__device__ __forceinline__ void ecx_set_inf(XyzzT<F>& p) { p.x = F::zero(), p.y = F::zero(), p.zz = F::zero(), p.zzz = F::zero(); }
template <class F>
__device__ __forceinline__ bool ecx_is_inf(const XyzzT<F>& p) { return F::is_zero(p.zz); }

// p = 2 a for an affine a != infinity (mdbl-2008-s-1; y = 0 does not occur in a group of odd prime order)
template <class F>
__device__ __forceinline__ void ecx_dbl_aff(XyzzT<F>& p, const AffT<F>& a) {
    const typename F::T u = F::dbl(a.y), v = F::sqr(u), w = F::mul(u, v), s = F::mul(a.x, v);
    const typename F::T xx = F::sqr(a.x), m = F::add(F::dbl(xx), xx);
    p.x = F::sub(F::sqr(m), F::dbl(s));
    p.y = F::sub(F::mul(m, F::sub(s, p.x)), F::mul(w, a.y));
    p.zz = v;
    p.zzz = w;
}
// p = 2 p (dbl-2008-s-1); infinity stays infinity (ZZ3 = V * 0)
template <class F>
__device__ __forceinline__ void ecx_dbl(XyzzT<F>& p) {
    const typename F::T u = F::dbl(p.y), v = F::sqr(u), w = F::mul(u, v), s = F::mul(p.x, v);
    const typename F::T xx = F::sqr(p.x), m = F::add(F::dbl(xx), xx);
    const typename F::T x3 = F::sub(F::sqr(m), F::dbl(s));
    p.y = F::sub(F::mul(m, F::sub(s, x3)), F::mul(w, p.y));
    p.x = x3;
    p.zz = F::mul(v, p.zz);
    p.zzz = F::mul(w, p.zzz);
}
// p += a (madd-2008-s: 8 M + 2 S), every special case handled: a or p at infinity, a == p (doubling), a == -p
template <class F>
__device__ __forceinline__ void ecx_madd(XyzzT<F>& p, const AffT<F>& a) {
    if (ec_aff_is_inf(a)) return;         // gnark-crypto's g1JacExtended.addMixed skips the (0, 0) encoding the same way
    if (ecx_is_inf(p)) {
        p.x = a.x, p.y = a.y, p.zz = F::one(), p.zzz = F::one();
        return;
    }
    const typename F::T pp_ = F::sub(F::mul(a.x, p.zz), p.x), r = F::sub(F::mul(a.y, p.zzz), p.y);
    if (F::is_zero(pp_)) {
        if (F::is_zero(r)) ecx_dbl_aff(p, a);
        else ecx_set_inf(p);
        return;
    }
    const typename F::T pp = F::sqr(pp_), ppp = F::mul(pp_, pp), q = F::mul(p.x, pp);
    const typename F::T x3 = F::sub(F::sub(F::sqr(r), ppp), F::dbl(q));
    p.y = F::sub(F::mul(r, F::sub(q, x3)), F::mul(p.y, ppp));
    p.x = x3;
    p.zz = F::mul(p.zz, pp);
    p.zzz = F::mul(p.zzz, ppp);
}
// The common case alone, for the hot loop of the bucket accumulation: p += a when neither is infinity and a != +-p;
// returns false and leaves p untouched otherwise (the caller then takes ecx_madd).  Keeping the doubling out of the loop
// body keeps its 6 products out of the loop's instruction stream (and of the instruction cache's working set).
template <class F>
__device__ __forceinline__ bool ecx_madd_fast(XyzzT<F>& p, const AffT<F>& a) {
    const typename F::T pp_ = F::sub(F::mul(a.x, p.zz), p.x), r = F::sub(F::mul(a.y, p.zzz), p.y);
    if (F::is_zero(pp_) || ecx_is_inf(p) || ec_aff_is_inf(a)) return false;
    const typename F::T pp = F::sqr(pp_), ppp = F::mul(pp_, pp), q = F::mul(p.x, pp);
    const typename F::T x3 = F::sub(F::sub(F::sqr(r), ppp), F::dbl(q));
    p.y = F::sub(F::mul(r, F::sub(q, x3)), F::mul(p.y, ppp));
    p.x = x3;
    p.zz = F::mul(p.zz, pp);
    p.zzz = F::mul(p.zzz, ppp);
    return true;
}
// p += q (add-2008-s: 12 M + 2 S), every special case handled
template <class F>
__device__ __forceinline__ void ecx_add(XyzzT<F>& p, const XyzzT<F>& q) {
    if (ecx_is_inf(q)) return;
    if (ecx_is_inf(p)) {
        p = q;
        return;
    }
    const typename F::T u1 = F::mul(p.x, q.zz), s1 = F::mul(p.y, q.zzz);
    const typename F::T pp_ = F::sub(F::mul(q.x, p.zz), u1), r = F::sub(F::mul(q.y, p.zzz), s1);
    if (F::is_zero(pp_)) {
        if (F::is_zero(r)) ecx_dbl(p);
        else ecx_set_inf(p);
        return;
    }
    const typename F::T pp = F::sqr(pp_), ppp = F::mul(pp_, pp), qq = F::mul(u1, pp);
    const typename F::T x3 = F::sub(F::sub(F::sqr(r), ppp), F::dbl(qq));
    p.y = F::sub(F::mul(r, F::sub(qq, x3)), F::mul(s1, ppp));
    p.x = x3;
    p.zz = F::mul(F::mul(p.zz, q.zz), pp);
    p.zzz = F::mul(F::mul(p.zzz, q.zzz), ppp);
}
// affine image (canonical coordinates; infinity -> (0, 0)) of an XYZZ point: one inversion of ZZ * ZZZ
template <class F>
__device__ __forceinline__ AffT<F> ecx_to_aff(const XyzzT<F>& p) {
    AffT<F> a;
    if (ecx_is_inf(p)) {
        a.x = F::zero(), a.y = F::zero();
        return a;
    }
    const typename F::T i = F::inv(F::mul(p.zz, p.zzz));
    a.x = F::canon(F::mul(p.x, F::mul(i, p.zzz)));      // X / ZZ
    a.y = F::canon(F::mul(p.y, F::mul(i, p.zz)));       // Y / ZZZ
    return a;
}

// ------------------------------------------------------------------------------------------------
// scalars -> signed window digits
// ------------------------------------------------------------------------------------------------
struct MsmArgs {
    const uint4* scalars;     // n x 32 B, the image of []fr.Element (4 x u64 little-endian) ...
    const uint4* scalars_hi;  // ... or, when not null, limb planes: scalar i = {scalars[i], scalars_hi[i]} (a device table, e.g. computeH's h)
    const uint4* points;      // n x 64 B (G1) or 128 B (G2): the image of []G1Affine / []G2Affine
    size_t n;
    int c, W;                 // window bits, number of windows
    unsigned int nb;          // buckets per window = 2^(c-1)
    int scalars_mont;         // 1: the scalars are in Montgomery form (MultiExpConfig.ScalarsMont), converted on the fly
    unsigned int* count;      // W * nb bucket sizes
    unsigned int* offset;     // exclusive scan of count
    unsigned int* chist;      // [W][nchunk][nb] chunk histograms, then chunk offsets
    unsigned int* order;      // [W][nb] bucket ids of a window by decreasing size
    unsigned int* tile_sum;   // scan tiles of MSM_SCAN_TILE buckets
    unsigned int ntiles;
    unsigned int nchunk;      // chunks of the scalar vector (one sorting workgroup per window and chunk)
    size_t chunk_len;
    u32 bias[8];              // H: half a window added to every window but the top one
    unsigned int* entries;    // point index | sign << 31, sorted by (window, bucket)
    unsigned int* big;        // [0] = number of segments of big buckets, [1 + k] = bucket id | segment << MSM_LIST_ID_BITS, [big_cap + 1] = a scalar was not below 2^254
    unsigned int big_threshold, big_cap;
    unsigned int seg;         // points per SEGMENT of a big bucket (a power of two): one workgroup sums one segment
    XPlanes bigparts;         // partial sums of the segments of multi-segment buckets, indexed like the segment list
    XPlanes buckets;          // W * nb
    XPlanes parts;            // W * nchunk chunk sums
    XPlanes wins;             // W window sums
    int chunk;                // buckets per lane of k_msm_reduce_chunks (a power of two)
    unsigned short* digits;   // [W][dstride] raw c-bit windows of the biased scalars (k_msm_digits)
    size_t dstride;           // n rounded up to a multiple of eight
    unsigned int* err;        // error word: 1 = a scalar was not below 2^254, 2 = a list overflowed
    // two-level sort (large n): the sorting kernels above run as a COARSE pass that files a point under bucket >> lowbits and
    // keeps the low bits in the entry; the refine kernels finish the order inside every coarse bin
    int lowbits;                       // coarse pass: > 0
    unsigned int* bin_first;           // coarse pass: first slice-list entry of every nonempty bin
    const unsigned int* c_entries;     // refine: the coarse pass's entries (index | sign << ib | low << (ib + 1), ib = 31 - lowbits) ...
    const unsigned int* c_count;       // ... its bin sizes, offsets, first list entries and the slice list (big[] of the coarse pass)
    const unsigned int* c_offset;
    const unsigned int* c_first;
    const unsigned int* slices;        // [0] = number of slices, [1 + k] = bin | slice << MSM_LIST_ID_BITS (slice_len entries each)
    unsigned int slice_cap, slice_len;
    int rbits;                         // refine: the coarse pass's lowbits
    unsigned int* slice_hist;          // [slice][2^rbits] counts, then write cursors
    // fixed-base MSM (k_msm_fb_*): ONE bucket space for all windows; W here is the number of windows of a scalar, the sums run with W = 1
    unsigned int* fb_raw;              // [W][n] digit planes of the fixed-base sort: bucket | sign << 31, 0xffffffff for a zero digit
    unsigned char fb_wb[32];           // fixed-base: bits of window j (255 bits dealt evenly: floor(255 / W) or one more, the wider ones on top) ...
    unsigned short fb_wo[32];          // ... and its first bit: T_j = [2^fb_wo[j]] P
    // the size ordering (k_msm_order) and the bucket sums (k_msm_accumulate) see the fixed-base MSM's one bucket space as acc_W
    // ranges of acc_nb consecutive buckets (a workgroup orders one range in LDS); 0: the windows themselves
    unsigned int acc_W, acc_nb;
};
__device__ __forceinline__ unsigned int msm_acc_windows(const MsmArgs& a) { return a.acc_W ? a.acc_W : (unsigned int)a.W; }
__device__ __forceinline__ unsigned int msm_acc_nb(const MsmArgs& a) { return a.acc_nb ? a.acc_nb : a.nb; }
// entries of the big-bucket list and of the slice list: id (bucket or coarse bin) in the low MSM_LIST_ID_BITS bits, the segment /
// slice number above (at most 2^11 of either per id: host_msm.hip.h sizes seg and slice_len for that).  21 bits: the one bucket
// space of a fixed-base MSM at c = 22 holds 2^21 buckets (20 bits until round 6).
#define MSM_LIST_ID_BITS 21
#define MSM_LIST_ID_MASK ((1u << MSM_LIST_ID_BITS) - 1u)
// bins of the sorting kernels: the buckets themselves, or the coarse bins of a two-level sort
__device__ __forceinline__ unsigned int msm_sort_bins(const MsmArgs& a) { return a.nb >> a.lowbits; }

__device__ __forceinline__ void msm_load_scalar(const MsmArgs& a, size_t i, u32 (&s)[8]) {
    const uint4 lo = a.scalars_hi ? a.scalars[i] : a.scalars[2 * i], hi = a.scalars_hi ? a.scalars_hi[i] : a.scalars[2 * i + 1];
    Fr x = {{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}};
    if (a.scalars_mont) {
        Fr one = fr_zero();
        one.v[0] = 1;
        x = fr_mul(x, one);          // Montgomery product with the plain integer 1: the regular form, canonical
    }
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = x.v[j];
}
// Signed digits without a carry chain: with H = sum_{j < W-1} 2^(c j + c - 1) (half a window added to every window but the
// top one) the plain c-bit windows d'_j of s + H give d_j = d'_j - 2^(c-1) in [-2^(c-1), 2^(c-1) - 1] for j < W - 1 and
// d_(W-1) = d'_(W-1) <= 2^(c-1) (s < 2^254 and c W >= 255), and sum_j d_j 2^(c j) = s.  Every window's digit is then a
// function of s alone, which lets a workgroup own ONE window.  Returns the bucket index |d| - 1 (and the sign), or
// 0xffffffff for a zero digit; MSM_DIGIT_BAD when the top digit leaves the bucket range (a scalar >= 2^254: not an fr.Element).
#define MSM_DIGIT_NONE 0xffffffffu
#define MSM_DIGIT_BAD 0xfffffffeu
__device__ __forceinline__ bool msm_bias_scalar(const MsmArgs& a, u32 (&s)[8]) {      // false: s + H left 256 bits
    u32 cy = 0;
#pragma unroll
    for (int l = 0; l < 8; l++) s[l] = fr_addc(s[l], a.bias[l], cy, &cy);
    return cy == 0;
}
// the plain c-bit window j of the biased scalar
__device__ __forceinline__ u32 msm_window_raw(const MsmArgs& a, const u32 (&sb)[8], int j) {
    const int bit = j * a.c, limb = bit >> 5, sh = bit & 31;      // uniform across the workgroup: selects, not indexed registers
    u32 lo = 0, hi = 0;
#pragma unroll
    for (int l = 0; l < 8; l++) {
        lo = (l == limb) ? sb[l] : lo;
        hi = (l == limb + 1) ? sb[l] : hi;
    }
    return (u32)((((u64)hi << 32) | lo) >> sh) & ((1u << a.c) - 1u);
}
__device__ __forceinline__ u32 msm_digit(const MsmArgs& a, u32 dp, int j, bool* neg) {
    const u32 half = a.nb;
    if (j == a.W - 1) {
        *neg = false;
        return dp == 0 ? MSM_DIGIT_NONE : (dp > half ? MSM_DIGIT_BAD : dp - 1u);
    }
    *neg = dp < half;
    const u32 mag = *neg ? half - dp : dp - half;
    return mag ? mag - 1u : MSM_DIGIT_NONE;
}
// Every scalar is decoded ONCE (Montgomery conversion included) into W planes of 16-bit raw windows, digits[j][i], rows padded
// to a multiple of eight: the sorting kernels then read 2 bytes per point and window instead of the 32-byte scalar (which
// every one of the 2 W workgroups over a chunk used to decode again).  Padding holds the raw value of a zero digit.
GKR_KERNEL void __launch_bounds__(GKR_BLOCK) k_msm_digits(MsmArgs a) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.dstride) return;
    if (i >= a.n) {
        for (int j = 0; j < a.W; j++) a.digits[(size_t)j * a.dstride + i] = (unsigned short)(j == a.W - 1 ? 0u : a.nb);
        return;
    }
    u32 s[8];
    msm_load_scalar(a, i, s);
    bool bad = !msm_bias_scalar(a, s);
    for (int j = 0; j < a.W; j++) {
        const u32 dp = msm_window_raw(a, s, j);
        if (j == a.W - 1 && dp > a.nb) bad = true;
        a.digits[(size_t)j * a.dstride + i] = (unsigned short)dp;
    }
    if (bad) *a.err = 1u;
}
// the eight raw windows of points i .. i + 7 (i a multiple of eight) of window j
__device__ __forceinline__ void msm_load_digits8(const MsmArgs& a, int j, size_t i, u32 (&dp)[8]) {
    const uint4 v = *reinterpret_cast<const uint4*>(a.digits + (size_t)j * a.dstride + i);
    dp[0] = v.x & 0xffffu;
    dp[1] = v.x >> 16;
    dp[2] = v.y & 0xffffu;
    dp[3] = v.y >> 16;
    dp[4] = v.z & 0xffffu;
    dp[5] = v.z >> 16;
    dp[6] = v.w & 0xffffu;
    dp[7] = v.w >> 16;
}

// Counting sort of the (window, bucket) pairs with every atomic in LDS.  Workgroup (j, k) owns window j of the k-th chunk
// of the scalars and keeps the window's whole histogram -- 2^(c-1) words, 128 KiB at c = 16: gfx950's 160 KiB of LDS per CU
// is what makes one pass per window possible -- so the 16 n increments of an MSM never leave the CU.
//   k_msm_hist:      chunk histograms            chist[j][k][b]
//   k_msm_totals / k_msm_scan / k_msm_offsets:   bucket sizes, their exclusive scan, the chunks' write positions
//   k_msm_scatter:   the chunk again, cursors in LDS initialised from chist: entries[cursor[b]++] = index | sign
#define MSM_SORT_THREADS 1024
GKR_KERNEL void __launch_bounds__(MSM_SORT_THREADS) k_msm_hist(MsmArgs a) {
    extern __shared__ unsigned int hist[];
    const int j = blockIdx.x;
    const unsigned int k = blockIdx.y;
    const unsigned int nbs = msm_sort_bins(a);
    for (unsigned int b = threadIdx.x; b < nbs; b += MSM_SORT_THREADS) hist[b] = 0;
    __syncthreads();
    const size_t lo = (size_t)k * a.chunk_len, hi = min(a.n, lo + a.chunk_len);      // chunk_len is a multiple of eight
    for (size_t i = lo + 8 * (size_t)threadIdx.x; i < hi; i += 8 * MSM_SORT_THREADS) {
        u32 dp[8];
        msm_load_digits8(a, j, i, dp);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            bool neg;
            const u32 b = msm_digit(a, dp[r], j, &neg);
            if (i + r < hi && b < MSM_DIGIT_BAD) atomicAdd(&hist[b >> a.lowbits], 1u);
        }
    }
    __syncthreads();
    unsigned int* out = a.chist + ((size_t)j * a.nchunk + k) * nbs;
    for (unsigned int b = threadIdx.x; b < nbs; b += MSM_SORT_THREADS) out[b] = hist[b];
}
// Exclusive scan of the W * nb bucket sizes in tiles of MSM_SCAN_TILE buckets, every access coalesced:
//   k_msm_totals:   count[t] = sum_k chist[.][k][.], tile sums                       (one workgroup per tile)
//   k_msm_scan:     exclusive scan of the tile sums                                   (one workgroup)
//   k_msm_offsets:  offset[t] = tile base + scan inside the tile; chunk offsets chist[j][k][b] <- offset + sum_{k' < k};
//                   the buckets above the threshold listed for k_msm_accumulate_big   (one workgroup per tile)
#define MSM_SCAN_THREADS 1024
#define MSM_SCAN_TILE 2048
__device__ __forceinline__ unsigned int msm_block_scan(unsigned int* sh, unsigned int v) {      // inclusive scan over the workgroup
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < MSM_SCAN_THREADS; d <<= 1) {
        const unsigned int x = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0u;
        __syncthreads();
        sh[threadIdx.x] += x;
        __syncthreads();
    }
    return sh[threadIdx.x];
}
GKR_KERNEL void __launch_bounds__(MSM_SCAN_THREADS) k_msm_totals(MsmArgs a) {
    __shared__ unsigned int sh[MSM_SCAN_THREADS];
    const unsigned int nbs = msm_sort_bins(a);
    const size_t total = (size_t)a.W * nbs;
    unsigned int mine = 0;
#pragma unroll
    for (int h = 0; h < MSM_SCAN_TILE / MSM_SCAN_THREADS; h++) {
        const size_t t = (size_t)blockIdx.x * MSM_SCAN_TILE + h * MSM_SCAN_THREADS + threadIdx.x;
        if (t < total) {
            const size_t j = t / nbs, b = t % nbs;
            unsigned int s = 0;
            for (unsigned int k = 0; k < a.nchunk; k++) s += a.chist[(j * a.nchunk + k) * nbs + b];
            a.count[t] = s;
            mine += s;
        }
    }
    const unsigned int incl = msm_block_scan(sh, mine);
    if (threadIdx.x == MSM_SCAN_THREADS - 1) a.tile_sum[blockIdx.x] = incl;
}
GKR_KERNEL void __launch_bounds__(MSM_SCAN_THREADS) k_msm_scan(MsmArgs a) {      // ntiles <= MSM_SCAN_THREADS * 8 (host checks)
    __shared__ unsigned int sh[MSM_SCAN_THREADS];
    const unsigned int per = (a.ntiles + MSM_SCAN_THREADS - 1) / MSM_SCAN_THREADS;
    const unsigned int lo = min(a.ntiles, per * threadIdx.x), hi = min(a.ntiles, lo + per);
    unsigned int s = 0;
    for (unsigned int t = lo; t < hi; t++) s += a.tile_sum[t];
    unsigned int run = msm_block_scan(sh, s) - s;
    for (unsigned int t = lo; t < hi; t++) {
        const unsigned int c = a.tile_sum[t];
        a.tile_sum[t] = run;
        run += c;
    }
}
GKR_KERNEL void __launch_bounds__(MSM_SCAN_THREADS) k_msm_offsets(MsmArgs a) {
    __shared__ unsigned int sh[MSM_SCAN_THREADS];
    const unsigned int nbs = msm_sort_bins(a);
    const size_t total = (size_t)a.W * nbs;
    const size_t t0 = (size_t)blockIdx.x * MSM_SCAN_TILE + (size_t)threadIdx.x * (MSM_SCAN_TILE / MSM_SCAN_THREADS);   // consecutive buckets per lane
    unsigned int c[MSM_SCAN_TILE / MSM_SCAN_THREADS], mine = 0;
#pragma unroll
    for (int h = 0; h < MSM_SCAN_TILE / MSM_SCAN_THREADS; h++) {
        c[h] = t0 + h < total ? a.count[t0 + h] : 0u;
        mine += c[h];
    }
    unsigned int run = a.tile_sum[blockIdx.x] + msm_block_scan(sh, mine) - mine;
#pragma unroll
    for (int h = 0; h < MSM_SCAN_TILE / MSM_SCAN_THREADS; h++) {
        const size_t t = t0 + h;
        if (t >= total) break;
        a.offset[t] = run;
        if (c[h] > a.big_threshold) {        // a big bucket: one list entry per segment of a.seg points (consecutive entries)
            const unsigned int nseg = (c[h] + a.seg - 1) / a.seg;
            const unsigned int k0 = atomicAdd(&a.big[0], nseg);
            for (unsigned int sg = 0; sg < nseg && k0 + sg < a.big_cap; sg++) a.big[1 + k0 + sg] = (unsigned int)t | (sg << MSM_LIST_ID_BITS);
            if (a.bin_first) a.bin_first[t] = k0;
        }
        const size_t j = t / nbs, b = t % nbs;
        unsigned int r2 = run;
        for (unsigned int k = 0; k < a.nchunk; k++) {
            unsigned int* p = &a.chist[(j * a.nchunk + k) * nbs + b];
            const unsigned int x = *p;
            *p = r2;
            r2 += x;
        }
        run += c[h];
    }
}
GKR_KERNEL void __launch_bounds__(MSM_SORT_THREADS) k_msm_scatter(MsmArgs a) {
    extern __shared__ unsigned int cursor[];
    const int j = blockIdx.x;
    const unsigned int k = blockIdx.y;
    const unsigned int* in = a.chist + ((size_t)j * a.nchunk + k) * a.nb;
    for (unsigned int b = threadIdx.x; b < a.nb; b += MSM_SORT_THREADS) cursor[b] = in[b];
    __syncthreads();
    const size_t lo = (size_t)k * a.chunk_len, hi = min(a.n, lo + a.chunk_len);
    for (size_t i = lo + 8 * (size_t)threadIdx.x; i < hi; i += 8 * MSM_SORT_THREADS) {
        u32 dp[8];
        msm_load_digits8(a, j, i, dp);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            bool neg;
            const u32 b = msm_digit(a, dp[r], j, &neg);
            if (i + r < hi && b < MSM_DIGIT_BAD) a.entries[atomicAdd(&cursor[b], 1u)] = (u32)(i + r) | (neg ? 0x80000000u : 0u);
        }
    }
}
// LDS counter update of a wave: when every active lane of the wave names the same counter (a slice of the bin that holds the
// 0/1 wires of a witness) one lane adds the population count; returns the lane's position
__device__ __forceinline__ unsigned int msm_wave_counter_add(unsigned int* ctr, unsigned int idx) {
    const unsigned int first = __builtin_amdgcn_readfirstlane(idx);
    const unsigned long long same = __ballot(idx == first), act = __ballot(true);
    if (same == act) {
        const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)act, 0u));
        unsigned int base = 0;
        if (rank == 0) base = atomicAdd(&ctr[first], (unsigned int)__popcll(act));
        return __builtin_amdgcn_readfirstlane(base) + rank;
    }
    return atomicAdd(&ctr[idx], 1u);
}
// ---- scatter through LDS ---------------------------------------------------------------------------------------------------
// A lane that writes its 4-byte entry wherever its bin's cursor points costs one memory request per lane (64 lines per wave
// store): the two scatters of the two-level sort were bound by that request rate, not by bytes.  Both therefore sort a batch
// (eight entries per lane) inside LDS first -- count per bin, scan, place -- and write the batch out in order, so that
// consecutive lanes write the consecutive entries of a bin's run.
// T lanes take batches of 8 T entries; at most T bins.
template <int T>
struct MsmStage {
    u32 staged[8 * T];
    unsigned short sbin[8 * T];
    unsigned int cursor[T];      // the workgroup's write position of every bin in the output
    unsigned int lcnt[T];        // entries of the batch per bin, then (output position - staged position) of the bin
    unsigned int lstart[T];
    unsigned int wsum[T / 64];
    unsigned int total;
};
// exclusive scan over the workgroup's T values
template <int T>
__device__ __forceinline__ unsigned int msm_stage_scan(MsmStage<T>& sh, unsigned int v) {
    const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned int y = __shfl_up(x, d, 64);
        if ((int)lane >= d) x += y;
    }
    if (lane == 63) sh.wsum[wave] = x;
    __syncthreads();
    unsigned int base = 0;
    for (unsigned int w = 0; w < wave; w++) base += sh.wsum[w];
    return base + x - v;
}
// One batch: the lane's eight entries e[r] for the bins bin[r] (0xffffffff: none) go to out[] at the bins' cursors.
template <int T>
__device__ __forceinline__ void msm_stage_batch(MsmStage<T>& sh, unsigned int nbins, const u32 (&e)[8], const u32 (&bin)[8], unsigned int* out) {
    unsigned int rank[8];
    if (threadIdx.x < nbins) sh.lcnt[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; r++)
        if (bin[r] != 0xffffffffu) rank[r] = msm_wave_counter_add(sh.lcnt, bin[r]);
    __syncthreads();
    const unsigned int mine = threadIdx.x < nbins ? sh.lcnt[threadIdx.x] : 0u;
    const unsigned int start = msm_stage_scan(sh, mine);
    if (threadIdx.x < nbins) {
        sh.lstart[threadIdx.x] = start;
        sh.lcnt[threadIdx.x] = sh.cursor[threadIdx.x] - start;
        sh.cursor[threadIdx.x] += mine;
        if (threadIdx.x == nbins - 1) sh.total = start + mine;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; r++)
        if (bin[r] != 0xffffffffu) {
            const unsigned int p = sh.lstart[bin[r]] + rank[r];
            sh.staged[p] = e[r];
            sh.sbin[p] = (unsigned short)bin[r];
        }
    __syncthreads();
    const unsigned int total = sh.total;
    for (unsigned int p = threadIdx.x; p < total; p += T) out[sh.lcnt[sh.sbin[p]] + p] = sh.staged[p];
    __syncthreads();
}
// the coarse pass's scatter: workgroup (j, k) files the entries of window j of chunk k under their coarse bins
GKR_KERNEL void __launch_bounds__(MSM_SORT_THREADS) k_msm_scatter_coarse(MsmArgs a) {
    __shared__ MsmStage<MSM_SORT_THREADS> sh;
    const int j = blockIdx.x;
    const unsigned int k = blockIdx.y;
    const unsigned int nbs = msm_sort_bins(a);
    const unsigned int* in = a.chist + ((size_t)j * a.nchunk + k) * nbs;
    if (threadIdx.x < nbs) sh.cursor[threadIdx.x] = in[threadIdx.x];
    __syncthreads();
    const size_t lo = (size_t)k * a.chunk_len, hi = min(a.n, lo + a.chunk_len);
    const int ib = 31 - a.lowbits;
    const u32 lowmask = (1u << a.lowbits) - 1u;
    for (size_t i0 = lo; i0 < hi; i0 += 8 * MSM_SORT_THREADS) {      // a lane takes the eight points of one 16-byte load
        u32 e[8], bin[8], dp[8];
        const size_t i = i0 + 8 * (size_t)threadIdx.x;
        if (i < hi) msm_load_digits8(a, j, i, dp);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            bin[r] = 0xffffffffu;
            e[r] = 0;
            if (i + r < hi) {
                bool neg;
                const u32 b = msm_digit(a, dp[r], j, &neg);
                if (b < MSM_DIGIT_BAD) {
                    bin[r] = b >> a.lowbits;
                    e[r] = (u32)(i + r) | ((neg ? 1u : 0u) << ib) | ((b & lowmask) << (ib + 1));
                }
            }
        }
        msm_stage_batch(sh, nbs, e, bin, a.entries);
    }
}
// ---- second level of the sort -------------------------------------------------------------------------------------------
// Above ~2^20 points the single pass above is bound by its writes: a workgroup scatters 4-byte entries over all 2^(c-1)
// buckets of its window, 2 MiB of open 64-byte lines per workgroup and 64 MiB per XCD against 4 MiB of L2, so nearly every
// entry costs a partial-line write in HBM (2^24 points: 6.4 ms of scatter for 1 GiB of entries).  Two levels keep the open
// lines in L2 at both: the coarse pass writes 2^(c-1-lowbits) streams per workgroup, then every SLICE (slice_len consecutive
// entries of one coarse bin; the bins of skewed scalars are cut into many) is counted, and scattered over the 2^lowbits
// buckets of its bin, whose runs are adjacent in the output.
//   k_msm_refine_count:    slice_hist[slice][low]                                          (one workgroup per slice)
//   k_msm_refine_offsets:  count / offset of every bucket, the slices' write cursors, the big-bucket list (one lane per bucket)
//   k_msm_refine_scatter:  entries[cursor[low]++] = index | sign << 31                     (one workgroup per slice)
#define MSM_REFINE_THREADS 256
#define MSM_REFINE_MAXLOW 128
__device__ __forceinline__ bool msm_slice_range(const MsmArgs& a, unsigned int* bin, unsigned int* lo, unsigned int* hi) {
    if (blockIdx.x >= min(a.slices[0], a.slice_cap)) return false;
    const unsigned int e = a.slices[1 + blockIdx.x];
    *bin = e & MSM_LIST_ID_MASK;
    const unsigned int base = a.c_offset[*bin], cnt = a.c_count[*bin], s0 = (e >> MSM_LIST_ID_BITS) * a.slice_len;
    *lo = base + s0;
    *hi = base + min(cnt, s0 + a.slice_len);
    return true;
}
GKR_KERNEL void __launch_bounds__(MSM_REFINE_THREADS) k_msm_refine_count(MsmArgs a) {
    __shared__ unsigned int hist[MSM_REFINE_MAXLOW];
    unsigned int bin, lo, hi;
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.slices[0] > a.slice_cap) *a.err = 2u;      // cannot happen by the list's sizing
    if (!msm_slice_range(a, &bin, &lo, &hi)) return;
    const unsigned int nlow = 1u << a.rbits;
    if (threadIdx.x < nlow) hist[threadIdx.x] = 0;
    __syncthreads();
    for (unsigned int i0 = lo; i0 < hi; i0 += 4 * MSM_REFINE_THREADS) {      // four loads in flight per lane
        u32 x[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const unsigned int i = i0 + (unsigned int)r * MSM_REFINE_THREADS + threadIdx.x;
            x[r] = i < hi ? a.c_entries[i] : 0u;
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
            if (i0 + (unsigned int)r * MSM_REFINE_THREADS + threadIdx.x < hi) (void)msm_wave_counter_add(hist, x[r] >> (32 - a.rbits));
    }
    __syncthreads();
    if (threadIdx.x < nlow) a.slice_hist[(size_t)blockIdx.x * nlow + threadIdx.x] = hist[threadIdx.x];
}
GKR_KERNEL void __launch_bounds__(GKR_BLOCK) k_msm_refine_offsets(MsmArgs a) {
    __shared__ unsigned int sh[GKR_BLOCK];
    const size_t t = (size_t)blockIdx.x * GKR_BLOCK + threadIdx.x;       // W * nb is a multiple of the block or smaller than it
    const size_t total = (size_t)a.W * a.nb;
    const unsigned int nlow = 1u << a.rbits, low = (unsigned int)t & (nlow - 1u);
    const size_t bin = t >> a.rbits;
    unsigned int cnt = 0, nsl = 0, k0 = 0, cb = 0;
    if (t < total) {
        cb = a.c_count[bin];
        nsl = (cb + a.slice_len - 1) / a.slice_len;
        k0 = nsl ? a.c_first[bin] : 0u;
        for (unsigned int s = 0; s < nsl; s++) cnt += a.slice_hist[(size_t)(k0 + s) * nlow + low];
    }
    sh[threadIdx.x] = cnt;
    __syncthreads();
    for (int d = 1; d < GKR_BLOCK; d <<= 1) {            // inclusive scan over the block
        const unsigned int x = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0u;
        __syncthreads();
        sh[threadIdx.x] += x;
        __syncthreads();
    }
    if (t >= total) return;
    const unsigned int g0 = threadIdx.x & ~(nlow - 1u);                  // first lane of the bucket's bin in this block
    unsigned int run = a.c_offset[bin] + sh[threadIdx.x] - cnt - (g0 ? sh[g0 - 1] : 0u);
    a.count[t] = cnt;
    a.offset[t] = run;
    if (cnt > a.big_threshold) {
        const unsigned int nseg = (cnt + a.seg - 1) / a.seg;
        const unsigned int b0 = atomicAdd(&a.big[0], nseg);
        for (unsigned int sg = 0; sg < nseg && b0 + sg < a.big_cap; sg++) a.big[1 + b0 + sg] = (unsigned int)t | (sg << MSM_LIST_ID_BITS);
    }
    for (unsigned int s = 0; s < nsl; s++) {
        unsigned int* p = &a.slice_hist[(size_t)(k0 + s) * nlow + low];
        const unsigned int x = *p;
        *p = run;
        run += x;
    }
}
GKR_KERNEL void __launch_bounds__(MSM_REFINE_THREADS) k_msm_refine_scatter(MsmArgs a) {
    __shared__ MsmStage<MSM_REFINE_THREADS> sh;
    unsigned int bin, lo, hi;
    if (!msm_slice_range(a, &bin, &lo, &hi)) return;
    const unsigned int nlow = 1u << a.rbits;
    if (threadIdx.x < nlow) sh.cursor[threadIdx.x] = a.slice_hist[(size_t)blockIdx.x * nlow + threadIdx.x];
    __syncthreads();
    const int ib = 31 - a.rbits;
    const u32 idxmask = (1u << ib) - 1u;
    for (unsigned int i0 = lo; i0 < hi; i0 += 8 * MSM_REFINE_THREADS) {
        u32 e[8], low[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const unsigned int i = i0 + (unsigned int)r * MSM_REFINE_THREADS + threadIdx.x;
            low[r] = 0xffffffffu;
            e[r] = 0;
            if (i < hi) {
                const u32 x = a.c_entries[i];
                low[r] = x >> (32 - a.rbits);
                e[r] = (x & idxmask) | (((x >> ib) & 1u) << 31);
            }
        }
        msm_stage_batch(sh, nlow, e, low, a.entries);
    }
}
// Lanes of a wave run for as long as their longest bucket: the buckets of a window are handed to the lanes in order of
// size (largest first), so that the 64 buckets of a wave hold nearly the same number of points (a counting sort of the
// window's 2^(c-1) bucket ids by their counts, one workgroup per window, everything in LDS).  Big buckets sort as empty:
// they belong to k_msm_accumulate_big.
#define MSM_ORDER_BINS 1024
GKR_KERNEL void __launch_bounds__(MSM_SORT_THREADS) k_msm_order(MsmArgs a) {
    __shared__ unsigned int bin[MSM_ORDER_BINS + 1];
    const unsigned int j = blockIdx.x;
    const unsigned int nb = msm_acc_nb(a);
    const unsigned int* cnt = a.count + (size_t)j * nb;
    for (unsigned int i = threadIdx.x; i <= MSM_ORDER_BINS; i += MSM_SORT_THREADS) bin[i] = 0;
    __syncthreads();
    for (unsigned int b = threadIdx.x; b < nb; b += MSM_SORT_THREADS) {
        const unsigned int c = cnt[b];
        atomicAdd(&bin[c > a.big_threshold ? 0u : min(c, (unsigned int)MSM_ORDER_BINS)], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {            // start of every size class, largest class first (a thousand additions)
        unsigned int run = 0;
        for (int i = MSM_ORDER_BINS; i >= 0; i--) {
            const unsigned int c = bin[i];
            bin[i] = run;
            run += c;
        }
    }
    __syncthreads();
    for (unsigned int b = threadIdx.x; b < nb; b += MSM_SORT_THREADS) {
        const unsigned int c = cnt[b];
        const unsigned int pos = atomicAdd(&bin[c > a.big_threshold ? 0u : min(c, (unsigned int)MSM_ORDER_BINS)], 1u);
        a.order[(size_t)j * nb + pos] = b;
    }
}

// ------------------------------------------------------------------------------------------------
// Fixed-base MSM (round 6).  The bases of the reference's MSMs are proving-key vectors, fixed across proofs (prove.go:76,91,189,
// 202,221,277 all name pk.*): with the multiples T_j[i] = [2^(o_j)] P_i (o_j: the first bit of window j) computed ONCE (k_msm_fb_precompute;
// W x the key's size in HBM), window j of scalar i contributes d_ij * T_j[i] and every window shares ONE bucket space -- so the window can be wide
// (c = 20..22: 13 or 12 additions per scalar instead of 16) without paying 2^(c-1) buckets per window in the reduction.  The sort
// then has 21-bit keys and 28-bit table indices, which the counting sort above (16-bit digit planes, 32-bit entries with the low
// bucket bits inside) does not hold: three levels of the same LDS counting sort follow (FbSortArgs).  Bucket sums, big buckets
// and the window sum are the kernels below with W = 1 and the table array as `points`.
// ------------------------------------------------------------------------------------------------
// tables[j * n + i] = [2^(first bit of window j)] points[i], affine, canonical (infinity stays (0, 0)); one lane per point, one
// inversion per entry.  The windows deal the 255 bits evenly -- floor(255 / W) bits or one more, the wider ones on top -- instead
// of W - 1 windows of c bits and a short one: a short top window (254 = 11 * 22 + 12) puts all its entries into 4 096 buckets,
// which then go through the segment path (2^24 points: 1.85 of 22 ms).
struct FbWindows {
    unsigned char wb[32];      // bits of window j
};
template <class F>
__global__ void __launch_bounds__(GKR_BLOCK) k_msm_fb_precompute(const uint4* __restrict__ points, uint4* __restrict__ tables, size_t n, FbWindows fw, int W) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const AffT<F> p = ec_ld_aff<F>(points, i);
    ec_st_aff<F>(tables, i, p);
    XyzzT<F> x;
    if (ec_aff_is_inf(p)) ecx_set_inf(x);
    else x.x = p.x, x.y = p.y, x.zz = F::one(), x.zzz = F::one();
    for (int j = 1; j < W; j++) {
        for (int k = 0; k < fw.wb[j - 1]; k++) ecx_dbl(x);
        ec_st_aff<F>(tables, (size_t)j * n + i, ecx_to_aff(x));
    }
}
// digit planes of every (window, point): a.W windows of a.c bits, a.nb = 2^(c-1) buckets; one lane per scalar
// (the level-1 histograms in the same pass -- a workgroup per chunk, W x nb1 LDS counters -- were measured: 0.56 ms against
// 0.22 + 0.28 at 2^24 points, and 32 workgroups are no launch at 2^20)
GKR_KERNEL void __launch_bounds__(GKR_BLOCK) k_msm_fb_digits(MsmArgs a) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    u32 s[8];
    msm_load_scalar(a, i, s);
    bool bad = !msm_bias_scalar(a, s);
    for (int j = 0; j < a.W; j++) {
        // window j: fb_wb[j] bits from bit fb_wo[j] (uniform over the launch: selects, not indexed registers)
        const int bit = a.fb_wo[j], wb = a.fb_wb[j], limb = bit >> 5, sh = bit & 31;
        u32 lo = 0, hi = 0;
#pragma unroll
        for (int l = 0; l < 8; l++) {
            lo = (l == limb) ? s[l] : lo;
            hi = (l == limb + 1) ? s[l] : hi;
        }
        const u32 dp = (u32)((((u64)hi << 32) | lo) >> sh) & ((1u << wb) - 1u);
        const u32 half = 1u << (wb - 1);
        bool neg = false;
        u32 b;
        if (j == a.W - 1) {
            b = dp == 0 ? MSM_DIGIT_NONE : (dp > half ? MSM_DIGIT_BAD : dp - 1u);
        } else {
            neg = dp < half;
            const u32 mag = neg ? half - dp : dp - half;
            b = mag ? mag - 1u : MSM_DIGIT_NONE;
        }
        if (b == MSM_DIGIT_BAD) bad = true;
        a.fb_raw[(size_t)j * a.n + i] = b < MSM_DIGIT_BAD ? (b | (neg ? 0x80000000u : 0u)) : 0xffffffffu;
    }
    if (bad) *a.err = 1u;
}
// ---- the fixed-base MSM's own sort: three levels of the LDS counting sort above ---------------------------------------------
// (bucket, entry) pairs of ONE bucket space of 2^kb buckets (kb = c - 1 <= 21) and W * n <= 2^31 entries: bucket bits
// bits1 | bits2 | bits3 (<= 7 | 7 | 7).  Level 1 reads the digit planes (window j, chunk k per workgroup, as the coarse pass
// above) and files an entry under the top bits1 bits; levels 2 and 3 are one kernel triple -- count per slice, offsets per
// output bin, staged scatter per slice -- run twice, each resolving up to seven more bits inside bins that fit the L2.  Every
// level writes the 32-bit entry (table index | sign << 31) and, beside it, the bucket bits still to be resolved (16 bits):
// 44 bytes of traffic per entry in all, no atomic outside LDS.  (rocPRIM's radix sort of the same (bucket, entry) pairs, the first
// version: 5.3 ms of a 23.5 ms MSM at 2^24 points against 4.2 of 22.6 -- profiles/r06_msm_fixed_base.txt, commit d5e303d has both.)
struct FbSortArgs {
    // level 1
    const unsigned int* raw;      // [W][n] bucket | sign << 31, 0xffffffff: none (k_msm_fb_digits)
    size_t n, tstride;            // scalars; the tables' window stride
    int W;
    unsigned int nchunk;
    size_t chunk_len;             // a multiple of eight
    int sh1;                      // bin = bucket >> sh1
    unsigned int nb1;             // 2^bits1 <= 1024
    unsigned int* chist;          // [W * nchunk][nb1] counts, then write positions
    // levels 2 and 3: output bin = input bin << bits | ((key >> sh) & (2^bits - 1))
    const unsigned int* e_in;
    const unsigned short* k_in;
    unsigned int* e_out;
    unsigned short* k_out;        // nullptr at the last level
    int sh, bits;
    unsigned int nbins_in;
    const unsigned int *in_count, *in_offset, *in_first;      // per input bin (in_first: its first slice-list entry)
    const unsigned int* slices;   // [0] = number of slices, [1 + k] = bin | slice << id_bits
    unsigned int slice_cap, slice_len;
    int id_bits, next_id_bits;    // id / segment split of the list read and of the list written (a level has few bins and may have many
                                  // slices per bin: 7 + 25 bits at level 2; the big-bucket list keeps MSM_LIST_ID_BITS)
    unsigned int* slice_hist;     // [slice][2^bits] counts, then write positions
    // what a level leaves for the next one (level 1 and 2: the next slice list; level 3: bucket sizes and the big-bucket list)
    unsigned int *out_count, *out_offset, *out_first;
    unsigned int* next_list;
    unsigned int next_cap, next_seg, next_threshold;
    unsigned int* err;
};
template <int T>
struct FbStage {
    u32 se[8 * T];
    unsigned short sk[8 * T], sbin[8 * T];
    unsigned int cursor[T], lcnt[T], lstart[T], wsum[T / 64], total;
};
template <int T>
__device__ __forceinline__ unsigned int fb_stage_scan(FbStage<T>& sh, unsigned int v) {
    const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned int y = __shfl_up(x, d, 64);
        if ((int)lane >= d) x += y;
    }
    if (lane == 63) sh.wsum[wave] = x;
    __syncthreads();
    unsigned int base = 0;
    for (unsigned int w = 0; w < wave; w++) base += sh.wsum[w];
    return base + x - v;
}
// one batch (as msm_stage_batch, with the residual key beside the entry); out_k may be null
template <int T>
__device__ __forceinline__ void fb_stage_batch(FbStage<T>& sh, unsigned int nbins, const u32 (&e)[8], const u32 (&k)[8], const u32 (&bin)[8],
                                               unsigned int* out_e, unsigned short* out_k) {
    unsigned int rank[8];
    for (unsigned int b = threadIdx.x; b < nbins; b += T) sh.lcnt[b] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; r++)
        if (bin[r] != 0xffffffffu) rank[r] = msm_wave_counter_add(sh.lcnt, bin[r]);
    __syncthreads();
    const unsigned int mine = threadIdx.x < nbins ? sh.lcnt[threadIdx.x] : 0u;      // nbins <= T
    const unsigned int start = fb_stage_scan(sh, mine);
    if (threadIdx.x < nbins) {
        sh.lstart[threadIdx.x] = start;
        sh.lcnt[threadIdx.x] = sh.cursor[threadIdx.x] - start;
        sh.cursor[threadIdx.x] += mine;
        if (threadIdx.x == nbins - 1) sh.total = start + mine;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; r++)
        if (bin[r] != 0xffffffffu) {
            const unsigned int p = sh.lstart[bin[r]] + rank[r];
            sh.se[p] = e[r];
            sh.sk[p] = (unsigned short)k[r];
            sh.sbin[p] = (unsigned short)bin[r];
        }
    __syncthreads();
    const unsigned int total = sh.total;
    for (unsigned int p = threadIdx.x; p < total; p += T) {
        const unsigned int o = sh.lcnt[sh.sbin[p]] + p;
        out_e[o] = sh.se[p];
        if (out_k) out_k[o] = sh.sk[p];
    }
    __syncthreads();
}
// level 1: counts of window j, chunk k per top-bits bin
GKR_KERNEL void __launch_bounds__(MSM_SORT_THREADS) k_fb_l1_hist(FbSortArgs a) {
    __shared__ unsigned int hist[1024];
    const unsigned int j = blockIdx.x, k = blockIdx.y;
    for (unsigned int b = threadIdx.x; b < a.nb1; b += MSM_SORT_THREADS) hist[b] = 0;
    __syncthreads();
    const size_t lo = (size_t)k * a.chunk_len, hi = min(a.n, lo + a.chunk_len);
    const unsigned int* raw = a.raw + (size_t)j * a.n;
    for (size_t i = lo + threadIdx.x; i < hi; i += MSM_SORT_THREADS) {
        const unsigned int r = raw[i];
        if (r != 0xffffffffu) (void)msm_wave_counter_add(hist, (r & 0x7fffffffu) >> a.sh1);
    }
    __syncthreads();
    unsigned int* out = a.chist + ((size_t)j * a.nchunk + k) * a.nb1;
    for (unsigned int b = threadIdx.x; b < a.nb1; b += MSM_SORT_THREADS) out[b] = hist[b];
}
// level 1: per bin (one workgroup each) the exclusive scan of its column of the chunk histograms -- a chunk's write position
// inside the bin -- and the bin's size.  (One workgroup walking all W * nchunk chunks per bin, the first version: 2.6 ms of a
// 6.8 ms sort at 2^24 points.)
GKR_KERNEL void __launch_bounds__(MSM_SCAN_THREADS) k_fb_l1_columns(FbSortArgs a) {
    __shared__ unsigned int sh[MSM_SCAN_THREADS];
    const unsigned int b = blockIdx.x, nw = (unsigned int)a.W * a.nchunk;
    const unsigned int per = (nw + MSM_SCAN_THREADS - 1) / MSM_SCAN_THREADS;
    const unsigned int lo = min(nw, per * threadIdx.x), hi = min(nw, lo + per);
    unsigned int mine = 0;
    for (unsigned int kk = lo; kk < hi; kk++) mine += a.chist[(size_t)kk * a.nb1 + b];
    const unsigned int incl = msm_block_scan(sh, mine);
    unsigned int run = incl - mine;
    for (unsigned int kk = lo; kk < hi; kk++) {
        unsigned int* p = &a.chist[(size_t)kk * a.nb1 + b];
        const unsigned int x = *p;
        *p = run;
        run += x;
    }
    if (threadIdx.x == MSM_SCAN_THREADS - 1) a.out_count[b] = incl;
}
// level 1: exclusive scan of the bin sizes and the slice list of level 2 (one workgroup, nb1 <= 1024 bins)
GKR_KERNEL void __launch_bounds__(MSM_SCAN_THREADS) k_fb_l1_offsets(FbSortArgs a) {
    __shared__ unsigned int sh[MSM_SCAN_THREADS];
    const unsigned int b = threadIdx.x;
    const unsigned int cnt = b < a.nb1 ? a.out_count[b] : 0u;
    const unsigned int off = msm_block_scan(sh, cnt) - cnt;
    if (b >= a.nb1) return;
    a.out_offset[b] = off;
    const unsigned int nsl = (cnt + a.next_seg - 1) / a.next_seg;
    const unsigned int k0 = nsl ? atomicAdd(&a.next_list[0], nsl) : 0u;
    for (unsigned int sg = 0; sg < nsl && k0 + sg < a.next_cap; sg++) a.next_list[1 + k0 + sg] = b | (sg << a.next_id_bits);
    a.out_first[b] = k0;
}
// level 1: the chunk again, staged scatter: entry = table index | sign << 31, key = the bucket bits below the bin's
#define FB_L1_SCATTER_THREADS 512      // (its staging area with the keys beside the entries: 38 KB; at most 512 bins -- level 1 has <= 128.  1024 lanes and 76 KB: measured slower, 3.55 against 3.32 ms of sort at 2^24 points)
GKR_KERNEL void __launch_bounds__(FB_L1_SCATTER_THREADS) k_fb_l1_scatter(FbSortArgs a) {
    __shared__ FbStage<FB_L1_SCATTER_THREADS> sh;
    const unsigned int j = blockIdx.x, k = blockIdx.y;
    const unsigned int* in = a.chist + ((size_t)j * a.nchunk + k) * a.nb1;
    if (threadIdx.x < a.nb1) sh.cursor[threadIdx.x] = a.out_offset[threadIdx.x] + in[threadIdx.x];      // the bin's start + the chunk's position inside it
    __syncthreads();
    const size_t lo = (size_t)k * a.chunk_len, hi = min(a.n, lo + a.chunk_len);
    const unsigned int* raw = a.raw + (size_t)j * a.n;
    const u32 kmask = (1u << a.sh1) - 1u;
    for (size_t i0 = lo; i0 < hi; i0 += 8 * FB_L1_SCATTER_THREADS) {
        u32 e[8], kk[8], bin[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {      // (lane-strided: consecutive lanes read consecutive words)
            const size_t i = i0 + (size_t)r * FB_L1_SCATTER_THREADS + threadIdx.x;
            bin[r] = 0xffffffffu;
            e[r] = kk[r] = 0;
            if (i < hi) {
                const unsigned int x = raw[i];
                if (x != 0xffffffffu) {
                    const u32 b = x & 0x7fffffffu;
                    bin[r] = b >> a.sh1;
                    kk[r] = b & kmask;
                    e[r] = (u32)((size_t)j * a.tstride + i) | (x & 0x80000000u);
                }
            }
        }
        fb_stage_batch(sh, a.nb1, e, kk, bin, a.e_out, a.k_out);
    }
}
// levels 2 and 3
#ifndef FB_LV_THREADS
#define FB_LV_THREADS 512      // lanes of a slice's workgroup: batches of 4096 entries, 32 per bin (128-byte runs)
#endif
__device__ __forceinline__ bool fb_slice_range(const FbSortArgs& a, unsigned int* bin, unsigned int* lo, unsigned int* hi) {
    if (blockIdx.x >= min(a.slices[0], a.slice_cap)) return false;
    const unsigned int e = a.slices[1 + blockIdx.x];
    *bin = e & ((1u << a.id_bits) - 1u);
    const unsigned int base = a.in_offset[*bin], cnt = a.in_count[*bin], s0 = (e >> a.id_bits) * a.slice_len;
    *lo = base + s0;
    *hi = base + min(cnt, s0 + a.slice_len);
    return true;
}
GKR_KERNEL void __launch_bounds__(FB_LV_THREADS) k_fb_lv_count(FbSortArgs a) {
    __shared__ unsigned int hist[MSM_REFINE_MAXLOW];
    unsigned int bin, lo, hi;
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.slices[0] > a.slice_cap) *a.err = 2u;      // cannot happen by the list's sizing
    if (!fb_slice_range(a, &bin, &lo, &hi)) return;
    const unsigned int nlow = 1u << a.bits;
    if (threadIdx.x < nlow) hist[threadIdx.x] = 0;
    __syncthreads();
    for (unsigned int i0 = lo; i0 < hi; i0 += 4 * FB_LV_THREADS) {
        u32 x[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const unsigned int i = i0 + (unsigned int)r * FB_LV_THREADS + threadIdx.x;
            x[r] = i < hi ? a.k_in[i] : 0u;
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
            if (i0 + (unsigned int)r * FB_LV_THREADS + threadIdx.x < hi) (void)msm_wave_counter_add(hist, (x[r] >> a.sh) & (nlow - 1u));
    }
    __syncthreads();
    if (threadIdx.x < nlow) a.slice_hist[(size_t)blockIdx.x * nlow + threadIdx.x] = hist[threadIdx.x];
}
GKR_KERNEL void __launch_bounds__(GKR_BLOCK) k_fb_lv_offsets(FbSortArgs a) {
    __shared__ unsigned int sh[GKR_BLOCK];
    const size_t t = (size_t)blockIdx.x * GKR_BLOCK + threadIdx.x;       // nbins_out is a multiple of the block or smaller than it
    const size_t total = (size_t)a.nbins_in << a.bits;
    const unsigned int nlow = 1u << a.bits, low = (unsigned int)t & (nlow - 1u);
    const size_t bin = t >> a.bits;
    unsigned int cnt = 0, nsl = 0, k0 = 0;
    if (t < total) {
        const unsigned int cb = a.in_count[bin];
        nsl = (cb + a.slice_len - 1) / a.slice_len;
        k0 = nsl ? a.in_first[bin] : 0u;
        for (unsigned int s = 0; s < nsl; s++) cnt += a.slice_hist[(size_t)(k0 + s) * nlow + low];
    }
    // exclusive prefix of cnt inside the group of nlow lanes that share the input bin (nlow <= 128, a power of two): a wave scan
    // that restarts at the group's first lane, and for groups of two waves the first wave's total (two barriers, not sixteen)
    const unsigned int wl = threadIdx.x & 63u, seg = min(nlow, 64u), gl = wl & (seg - 1u);
    unsigned int x = cnt;
#pragma unroll
    for (unsigned int d = 1; d < 64; d <<= 1) {
        const unsigned int y = __shfl_up(x, d, 64);
        if (gl >= d) x += y;
    }
    if (wl == 63) sh[threadIdx.x >> 6] = x;
    __syncthreads();
    unsigned int pre = x - cnt;
    if (nlow == 128 && ((threadIdx.x >> 6) & 1u)) pre += sh[(threadIdx.x >> 6) - 1];
    if (t >= total) return;
    unsigned int run = a.in_offset[bin] + pre;
    a.out_count[t] = cnt;
    a.out_offset[t] = run;
    if (cnt > a.next_threshold) {
        const unsigned int nseg = (cnt + a.next_seg - 1) / a.next_seg;
        const unsigned int b0 = atomicAdd(&a.next_list[0], nseg);
        for (unsigned int sg = 0; sg < nseg && b0 + sg < a.next_cap; sg++) a.next_list[1 + b0 + sg] = (unsigned int)t | (sg << a.next_id_bits);
        if (a.out_first) a.out_first[t] = b0;
    }
    for (unsigned int s = 0; s < nsl; s++) {
        unsigned int* p = &a.slice_hist[(size_t)(k0 + s) * nlow + low];
        const unsigned int x = *p;
        *p = run;
        run += x;
    }
}
// The same for a level with FEW input bins of MANY slices (level 2: <= 128 bins of ~200 slices at 2^24 points): one workgroup per
// input bin, the bin's slices dealt to 1024 / 2^bits parts per output bin, so that a lane walks nsl / parts slices instead of all
// (one lane per output bin walking them all, the kernel above: 0.55 ms of a 3.4 ms sort -- 400 dependent loads in a row).
GKR_KERNEL void __launch_bounds__(MSM_SCAN_THREADS) k_fb_lv_offsets_bin(FbSortArgs a) {
    __shared__ unsigned int psum[MSM_SCAN_THREADS];      // [part][low]: the part's count, then its exclusive prefix over the parts
    __shared__ unsigned int base[MSM_REFINE_MAXLOW];     // output bin's first position
    const unsigned int bin = blockIdx.x, nlow = 1u << a.bits, low = threadIdx.x & (nlow - 1u), part = threadIdx.x >> a.bits;
    const unsigned int nparts = MSM_SCAN_THREADS >> a.bits;
    const unsigned int cb = a.in_count[bin], nsl = (cb + a.slice_len - 1) / a.slice_len, k0 = nsl ? a.in_first[bin] : 0u;
    const unsigned int per = (nsl + nparts - 1) / nparts, s_lo = min(nsl, part * per), s_hi = min(nsl, s_lo + per);
    unsigned int mine = 0;
    for (unsigned int sl = s_lo; sl < s_hi; sl++) mine += a.slice_hist[(size_t)(k0 + sl) * nlow + low];
    psum[threadIdx.x] = mine;
    __syncthreads();
    if (part == 0) {                                     // exclusive prefix over the parts of this output bin (<= 8 parts at 128 bins)
        unsigned int run = 0;
        for (unsigned int p = 0; p < nparts; p++) {
            const unsigned int x = psum[p * nlow + low];
            psum[p * nlow + low] = run;
            run += x;
        }
        base[low] = run;                                 // the output bin's size, for now
    }
    __syncthreads();
    // exclusive prefix over the <= 128 output bins of this input bin: the first nlow lanes (part 0: one or two waves), a wave scan and
    // the first wave's total for the second (one lane walking the 128 bins with their stores: 0.2 ms)
    __shared__ unsigned int wtot[2];
    unsigned int c = 0, x = 0;
    if (threadIdx.x < nlow) {
        c = base[low];
        x = c;
        const unsigned int wl = threadIdx.x & 63u;
#pragma unroll
        for (unsigned int d = 1; d < 64; d <<= 1) {
            const unsigned int y = __shfl_up(x, d, 64);
            if (wl >= d) x += y;
        }
        if (wl == 63 || threadIdx.x == nlow - 1) wtot[threadIdx.x >> 6] = x;
    }
    __syncthreads();
    if (threadIdx.x < nlow) {
        const unsigned int off = a.in_offset[bin] + (x - c) + (threadIdx.x >= 64 ? wtot[0] : 0u);
        const size_t t = ((size_t)bin << a.bits) | low;
        a.out_count[t] = c;
        a.out_offset[t] = off;
        if (c > a.next_threshold) {
            const unsigned int nseg = (c + a.next_seg - 1) / a.next_seg;
            const unsigned int b0 = atomicAdd(&a.next_list[0], nseg);
            for (unsigned int sg = 0; sg < nseg && b0 + sg < a.next_cap; sg++) a.next_list[1 + b0 + sg] = (unsigned int)t | (sg << a.next_id_bits);
            if (a.out_first) a.out_first[t] = b0;
        }
        base[low] = off;
    }
    __syncthreads();
    unsigned int run = base[low] + psum[threadIdx.x];
    for (unsigned int sl = s_lo; sl < s_hi; sl++) {
        unsigned int* p = &a.slice_hist[(size_t)(k0 + sl) * nlow + low];
        const unsigned int x = *p;
        *p = run;
        run += x;
    }
}
GKR_KERNEL void __launch_bounds__(FB_LV_THREADS) k_fb_lv_scatter(FbSortArgs a) {
    __shared__ FbStage<FB_LV_THREADS> sh;
    unsigned int bin, lo, hi;
    if (!fb_slice_range(a, &bin, &lo, &hi)) return;
    const unsigned int nlow = 1u << a.bits;
    if (threadIdx.x < nlow) sh.cursor[threadIdx.x] = a.slice_hist[(size_t)blockIdx.x * nlow + threadIdx.x];
    __syncthreads();
    const u32 kmask = (1u << a.sh) - 1u;
    for (unsigned int i0 = lo; i0 < hi; i0 += 8 * FB_LV_THREADS) {
        u32 e[8], kk[8], low[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const unsigned int i = i0 + (unsigned int)r * FB_LV_THREADS + threadIdx.x;
            low[r] = 0xffffffffu;
            e[r] = kk[r] = 0;
            if (i < hi) {
                const u32 key = a.k_in[i];
                low[r] = (key >> a.sh) & (nlow - 1u);
                kk[r] = key & kmask;
                e[r] = a.e_in[i];
            }
        }
        fb_stage_batch(sh, nlow, e, kk, low, a.e_out, a.k_out);
    }
}

// ------------------------------------------------------------------------------------------------
// bucket accumulation
// ------------------------------------------------------------------------------------------------
template <class F>
__device__ __forceinline__ AffT<F> msm_entry_point(const MsmArgs& a, unsigned int e) {
    AffT<F> p = ec_ld_aff<F>(a.points, e & 0x7fffffffu);
    if (e & 0x80000000u) p.y = F::neg(p.y);
    return p;
}
// Register budgets: the G1 loop needs 159 VGPRs (three waves per SIMD; forcing 128 spills and is 8-12 % slower); a G2 point is
// twice as wide (the accumulator alone is 64 registers): one workgroup per CU may take the whole file.
template <class F>
struct MsmTune {
    static const int ACC_MINBLOCKS = 3;
    static const bool PREFETCH = true;
};
template <>
struct MsmTune<Fp2F> {
    static const int ACC_MINBLOCKS = 1;
    static const bool PREFETCH = false;
};
template <class F>
__global__ void __launch_bounds__(GKR_BLOCK, MsmTune<F>::ACC_MINBLOCKS) k_msm_accumulate(MsmArgs a) {
    const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t aW = msm_acc_windows(a), anb = msm_acc_nb(a);
    if (lane >= aW * anb) return;
    // The lane's bucket: the (lane mod nb)-th largest of its window, the TOP window first: a scalar below q < 2^254 leaves the
    // top digit few values (q >> 240 = 12 388 of the 32 768 buckets at c = 16), so its buckets hold 2.6 times the points of the
    // others -- launched last they were the kernel's tail (2^24 points: 26.9 ms, against 22 ms for 16 windows at the rate of
    // the first 14).
    // (fixed-base: the heavy buckets are the LOW ones -- a short top window puts all its entries below 2^(bits of that window) -- so
    // the ranges go in ascending order there, for the same reason: 2^24 points at c = 20, 33.1 ms with the heavy range last)
    const size_t jw = a.acc_W ? lane / anb : (aW - 1) - lane / anb;
    const size_t t = jw * anb + a.order[jw * anb + lane % anb];
    const unsigned int cnt = a.count[t], start = a.offset[t];
    XyzzT<F> acc;
    ecx_set_inf(acc);
    if (cnt && cnt <= a.big_threshold) {
        const unsigned int* ent = a.entries + start;
        if (MsmTune<F>::PREFETCH) {
            // Software pipeline of a lane's chain index -> point -> 10 products: the index is loaded two points ahead and the
            // point one ahead, so neither HBM round trip sits between two additions.  The inner loop is the common case only;
            // the first point of a bucket and the rare doubling / cancellation leave it for the complete addition.
            AffT<F> cur = msm_entry_point<F>(a, ent[0]);
            unsigned int k = 1;
            unsigned int e_nxt = cnt > 1 ? ent[1] : 0u;
            for (;;) {
                bool done = false;
                for (;;) {
                    if (k >= cnt) {
                        done = true;
                        break;
                    }
                    const AffT<F> nxt = msm_entry_point<F>(a, e_nxt);
                    const unsigned int e_nn = k + 1 < cnt ? ent[k + 1] : 0u;
                    if (!ecx_madd_fast(acc, cur)) break;       // e_nxt still names point k: the slow path re-loads it
                    cur = nxt;
                    e_nxt = e_nn;
                    k++;
                }
                if (done) break;
                ecx_madd(acc, cur);
                cur = msm_entry_point<F>(a, e_nxt);
                k++;
                e_nxt = k < cnt ? ent[k] : 0u;
            }
            ecx_madd(acc, cur);
        } else {
            for (unsigned int k = 0; k < cnt; k++) {
                const AffT<F> cur = msm_entry_point<F>(a, ent[k]);
                if (!ecx_madd_fast(acc, cur)) ecx_madd(acc, cur);
            }
        }
    }
    ecx_st<F>(a.buckets, t, acc);        // a big bucket is overwritten by k_msm_accumulate_big (launched after this kernel)
}
// LDS tree over the workgroup's XYZZ partial sums; the result is in sh[0]
template <class F>
struct XyzzShared {
    XyzzT<F> p[GKR_BLOCK];
};
template <class F>
__device__ __forceinline__ void ecx_block_reduce(XyzzShared<F>& sh, XyzzT<F>& mine) {
    sh.p[threadIdx.x] = mine;
    __syncthreads();
    for (int d = GKR_BLOCK / 2; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) {
            XyzzT<F> x = sh.p[threadIdx.x];
            ecx_add(x, sh.p[threadIdx.x + d]);
            sh.p[threadIdx.x] = x;
        }
        __syncthreads();
    }
}
// Buckets far above the mean -- the 0/1 wires of a real witness put a large share of ALL points into bucket 1 of window 0; a
// short top window holds n / 4 points per bucket -- are cut into segments of a.seg points and a workgroup sums a segment
// (one workgroup per bucket, the first version, took 8 ms for four buckets of 2^18 points: 1024 additions in a row per lane).
// A bucket of one segment is finished here; the partial sums of the others are combined by k_msm_big_combine.
template <class F>
__global__ void __launch_bounds__(GKR_BLOCK) k_msm_accumulate_big(MsmArgs a) {
    __shared__ XyzzShared<F> sh;
    const unsigned int nbig = min(a.big[0], a.big_cap);
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.big[0] > a.big_cap) *a.err = 2u;      // the list overflowed (cannot happen by its sizing): an error, not a wrong sum
    for (unsigned int k = blockIdx.x; k < nbig; k += gridDim.x) {
        const unsigned int e = a.big[1 + k];
        const size_t t = e & MSM_LIST_ID_MASK;
        const unsigned int sg = e >> MSM_LIST_ID_BITS;
        const unsigned int cnt = a.count[t], start = a.offset[t];
        const unsigned int lo = sg * a.seg, hi = min(cnt, lo + a.seg);
        XyzzT<F> acc;
        ecx_set_inf(acc);
        for (unsigned int i = lo + threadIdx.x; i < hi; i += GKR_BLOCK) ecx_madd(acc, msm_entry_point<F>(a, a.entries[start + i]));
        ecx_block_reduce(sh, acc);
        if (threadIdx.x == 0) {
            if (cnt <= a.seg) ecx_st<F>(a.buckets, t, sh.p[0]);
            else ecx_st<F>(a.bigparts, k, sh.p[0]);
        }
        __syncthreads();
    }
}
template <class F>
__global__ void __launch_bounds__(GKR_BLOCK) k_msm_big_combine(MsmArgs a) {
    __shared__ XyzzShared<F> sh;
    const unsigned int nbig = min(a.big[0], a.big_cap);
    for (unsigned int k = blockIdx.x; k < nbig; k += gridDim.x) {
        const unsigned int e = a.big[1 + k];
        const size_t t = e & MSM_LIST_ID_MASK;
        const unsigned int cnt = a.count[t];
        if ((e >> MSM_LIST_ID_BITS) != 0 || cnt <= a.seg) continue;          // (uniform over the workgroup) the entry of segment 0 speaks for its bucket
        const unsigned int nseg = (cnt + a.seg - 1) / a.seg;
        XyzzT<F> acc;
        ecx_set_inf(acc);
        for (unsigned int i = threadIdx.x; i < nseg; i += GKR_BLOCK) ecx_add(acc, ecx_ld<F>(a.bigparts, k + i));
        ecx_block_reduce(sh, acc);
        if (threadIdx.x == 0) ecx_st<F>(a.buckets, t, sh.p[0]);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// window sums:  sum_b (b + 1) B_b
// ------------------------------------------------------------------------------------------------
// [k] p for a small k by double-and-add (k < 2^16)
template <class F>
__device__ __forceinline__ XyzzT<F> ecx_mul_small(const XyzzT<F>& p, unsigned int k) {
    XyzzT<F> r;
    ecx_set_inf(r);
    for (int i = 31 - __clz(k | 1u); i >= 0; i--) {
        ecx_dbl(r);
        if ((k >> i) & 1u) ecx_add(r, p);
    }
    return r;
}
// lane (j, ci): buckets [ci * chunk, (ci + 1) * chunk) of window j from the top down: running = sum B, acc = sum of the
// running sums = sum (b - b0 + 1) B_b; the chunk's share of the window sum is acc + b0 * running.  When the chunks of a window
// fill whole workgroups (every window size from 2^11 buckets up) the workgroup sums its lanes' shares at once (LDS tree) and
// k_msm_reduce_windows is left with a handful of terms per window instead of sixteen additions in a row per lane: the window
// sums are a CHAIN of group operations on one lane (~9 us each with one wave per SIMD), 62 of them before, 50 now.
template <class F>
__global__ void __launch_bounds__(GKR_BLOCK) k_msm_reduce_chunks(MsmArgs a) {
    __shared__ XyzzShared<F> sh;
    const unsigned int nchunk = a.nb / a.chunk;
    const bool fused = nchunk % GKR_BLOCK == 0;         // uniform over the launch
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (!fused && g >= (size_t)a.W * nchunk) return;
    const unsigned int j = (unsigned int)(g / nchunk), ci = (unsigned int)(g % nchunk);
    const unsigned int b0 = ci * a.chunk;
    XyzzT<F> running, acc;
    ecx_set_inf(running);
    ecx_set_inf(acc);
    for (int k = a.chunk - 1; k >= 0; k--) {
        ecx_add(running, ecx_ld<F>(a.buckets, (size_t)j * a.nb + b0 + k));
        ecx_add(acc, running);
    }
    if (b0) ecx_add(acc, ecx_mul_small(running, b0));
    if (!fused) {
        ecx_st<F>(a.parts, g, acc);
        return;
    }
    ecx_block_reduce(sh, acc);
    if (threadIdx.x == 0) ecx_st<F>(a.parts, blockIdx.x, sh.p[0]);
}
// one workgroup per window: the sum of its chunk results (of its workgroups' results when k_msm_reduce_chunks summed them)
template <class F>
__global__ void __launch_bounds__(GKR_BLOCK) k_msm_reduce_windows(MsmArgs a) {
    __shared__ XyzzShared<F> sh;
    const unsigned int nchunk = a.nb / a.chunk;
    const unsigned int per = nchunk % GKR_BLOCK == 0 ? nchunk / GKR_BLOCK : nchunk;
    const unsigned int j = blockIdx.x;
    XyzzT<F> acc;
    ecx_set_inf(acc);
    for (unsigned int ci = threadIdx.x; ci < per; ci += GKR_BLOCK) ecx_add(acc, ecx_ld<F>(a.parts, (size_t)j * per + ci));
    ecx_block_reduce(sh, acc);
    if (threadIdx.x == 0) ecx_st<F>(a.wins, j, sh.p[0]);
}

// ------------------------------------------------------------------------------------------------
// bn254.BatchScalarMultiplicationG1 / G2 (base, scalars) (prove.go:177): out[i] = [s_i] base, affine.  One lane per scalar,
// left-to-right double-and-add with mixed additions, then the lane's own inversion (a tenth of its work).
// ------------------------------------------------------------------------------------------------
template <class F>
__global__ void __launch_bounds__(GKR_BLOCK) k_ec_batch_scalar_mul(MsmArgs a, AffT<F> base, uint4* out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    u32 s[8];
    msm_load_scalar(a, i, s);
    XyzzT<F> acc;
    ecx_set_inf(acc);
    for (int limb = 7; limb >= 0; limb--) {
        u32 w = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) w = (j == limb) ? s[j] : w;        // no dynamic register indexing
        for (int b = 31; b >= 0; b--) {
            ecx_dbl(acc);
            if ((w >> b) & 1u) ecx_madd(acc, base);
        }
    }
    ec_st_aff<F>(out, i, ecx_to_aff(acc));
}

// synthetic scalars of the benchmark: pseudo-random canonical values below q, out[i] = limbs of (mix(i, seed))^7
GKR_KERNEL void __launch_bounds__(GKR_BLOCK) k_msm_synth_scalars(uint4* out, size_t n, u32 seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr x = {{(u32)i ^ 0x9df123fu, (u32)(i >> 32) + 0xf45cu, seed, 0x2545f491u, (u32)i * 0x9e3779b9u, 3u, seed ^ 0x5bd1e995u, 0u}};   // < 2^224 < q
        x = fr_pow7(x);
        out[2 * i] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
        out[2 * i + 1] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
    }
}
