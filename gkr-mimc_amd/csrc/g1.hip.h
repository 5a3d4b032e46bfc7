// g1.hip.h -- BN254 G1 arithmetic and the kernels of the multi-scalar multiplication sum_i [s_i] P_i on gfx950: what
// gnark-crypto's (*G1Jac).MultiExp / (*G1Affine).MultiExp compute for the reference's Groth16 prover
// (prover/gadget/prove.go:76,91 krsNotGkr / KrsPrivNotGkr; :189 bs1, :202 ar, :221 krs2) and
// bn254.BatchScalarMultiplicationG1 (prove.go:177).  gnark-crypto is an un-vendored dependency of the reference
// (v0.6.1-0.20220110145513-493bb1c180d9): the RESULT is a group element, so parity is defined by the mathematics -- the
// affine coordinates of the sum are unique -- and pinned by the test oracle's big-integer double-and-add ("parity
// unpinned" against bytes of the Go binary, like computeH).
//
// Pippenger's bucket method laid out for a GPU instead of gnark-crypto's one-goroutine-per-window loop:
//   1. every scalar (regular form, < q < 2^254) is cut into W = ceil(255 / c) SIGNED digits of c bits
//      (d in [-2^(c-1)+1, 2^(c-1)]: half the buckets, the sign negates the point's y);
//   2. the (window, |digit|) pairs are counting-sorted: a histogram (k_msm_count), one exclusive scan (k_msm_scan), a
//      scatter of point indices (k_msm_scatter) -- atomics on 32-bit counters only, never on points;
//   3. ONE LANE PER BUCKET adds its run of points with mixed additions into an extended-Jacobian (XYZZ) accumulator
//      (k_msm_accumulate: 8 M + 2 S per point, the bulk of the work: W * n additions); the few buckets far above the mean
//      (skewed scalars: the 0/1 wires of a real witness put most points of window 0 into bucket 1) are summed by a whole
//      workgroup each (k_msm_accumulate_big);
//   4. sum_b (b + 1) B_b per window by running sums over chunks of consecutive buckets, one lane per chunk, the chunk's
//      offset applied by a short double-and-add (k_msm_reduce_chunks), then a workgroup tree per window (k_msm_reduce_windows);
//   5. the W window sums go to the host, which combines them by Horner's rule (c doublings per window) and converts to
//      affine: ~270 group operations, microseconds on a CPU core, milliseconds for a lone GPU lane.
// Coordinates live in the lazy range [0, 2p) of fp_bn254.h.  No MFMA: exact 254-bit modular arithmetic.
// Points in HBM keep gnark's G1Affine image (X, Y: 2 x 32 B Montgomery, infinity = (0, 0)): a lane gathers one point as four
// 16-byte loads of one 64-byte line; buckets and partial sums are limb planes (coalesced across lanes).
#pragma once
#include "fp_bn254.h"
#include "kernels.hip.h"

struct G1Aff {
    Fp x, y;
};
struct G1X {               // extended Jacobian: x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2; infinity: ZZ = 0
    Fp x, y, zz, zzz;
};

__device__ __forceinline__ Fp fp_from4(const uint4& a, const uint4& b) {
    Fp r = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
    return r;
}
__device__ __forceinline__ G1Aff g1_ld_aff(const uint4* __restrict__ pts, size_t i) {
    const uint4 a = pts[4 * i], b = pts[4 * i + 1], c = pts[4 * i + 2], d = pts[4 * i + 3];
    G1Aff r;
    r.x = fp_from4(a, b);
    r.y = fp_from4(c, d);
    return r;
}
__device__ __forceinline__ void g1_st_aff(uint4* __restrict__ pts, size_t i, const G1Aff& p) {
    pts[4 * i] = make_uint4(p.x.v[0], p.x.v[1], p.x.v[2], p.x.v[3]);
    pts[4 * i + 1] = make_uint4(p.x.v[4], p.x.v[5], p.x.v[6], p.x.v[7]);
    pts[4 * i + 2] = make_uint4(p.y.v[0], p.y.v[1], p.y.v[2], p.y.v[3]);
    pts[4 * i + 3] = make_uint4(p.y.v[4], p.y.v[5], p.y.v[6], p.y.v[7]);
}
// XYZZ points as eight planes of 16-byte words, plane k of element t at base[k * stride + t]
struct G1XPlanes {
    uint4* base;
    size_t stride;
};
__device__ __forceinline__ G1X g1x_ld(const G1XPlanes& pl, size_t t) {
    G1X r;
    r.x = fp_from4(pl.base[t], pl.base[pl.stride + t]);
    r.y = fp_from4(pl.base[2 * pl.stride + t], pl.base[3 * pl.stride + t]);
    r.zz = fp_from4(pl.base[4 * pl.stride + t], pl.base[5 * pl.stride + t]);
    r.zzz = fp_from4(pl.base[6 * pl.stride + t], pl.base[7 * pl.stride + t]);
    return r;
}
__device__ __forceinline__ void g1x_st(const G1XPlanes& pl, size_t t, const G1X& p) {
    const Fp* f[4] = {&p.x, &p.y, &p.zz, &p.zzz};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        pl.base[(2 * k) * pl.stride + t] = make_uint4(f[k]->v[0], f[k]->v[1], f[k]->v[2], f[k]->v[3]);
        pl.base[(2 * k + 1) * pl.stride + t] = make_uint4(f[k]->v[4], f[k]->v[5], f[k]->v[6], f[k]->v[7]);
    }
}

__device__ __forceinline__ bool g1_aff_is_inf(const G1Aff& a) { return fp_is_zero(a.x) && fp_is_zero(a.y); }
__device__ __forceinline__ void g1x_set_inf(G1X& p) { p.x = fp_zero(), p.y = fp_zero(), p.zz = fp_zero(), p.zzz = fp_zero(); }
__device__ __forceinline__ bool g1x_is_inf(const G1X& p) { return fp_is_zero(p.zz); }

// p = 2 a for an affine a != infinity (mdbl-2008-s-1; y = 0 does not occur on a curve of odd prime order)
__device__ __forceinline__ void g1x_dbl_aff(G1X& p, const G1Aff& a) {
    const Fp u = fp_dbl(a.y), v = fp_sqr(u), w = fp_mul(u, v), s = fp_mul(a.x, v);
    const Fp xx = fp_sqr(a.x), m = fp_add(fp_dbl(xx), xx);
    p.x = fp_sub(fp_sqr(m), fp_dbl(s));
    p.y = fp_sub(fp_mul(m, fp_sub(s, p.x)), fp_mul(w, a.y));
    p.zz = v;
    p.zzz = w;
}
// p = 2 p (dbl-2008-s-1); infinity stays infinity (ZZ3 = V * 0)
__device__ __forceinline__ void g1x_dbl(G1X& p) {
    const Fp u = fp_dbl(p.y), v = fp_sqr(u), w = fp_mul(u, v), s = fp_mul(p.x, v);
    const Fp xx = fp_sqr(p.x), m = fp_add(fp_dbl(xx), xx);
    const Fp x3 = fp_sub(fp_sqr(m), fp_dbl(s));
    p.y = fp_sub(fp_mul(m, fp_sub(s, x3)), fp_mul(w, p.y));
    p.x = x3;
    p.zz = fp_mul(v, p.zz);
    p.zzz = fp_mul(w, p.zzz);
}
// p += a (madd-2008-s: 8 M + 2 S), every special case handled: a or p at infinity, a == p (doubling), a == -p
__device__ __forceinline__ void g1x_madd(G1X& p, const G1Aff& a) {
    if (g1_aff_is_inf(a)) return;         // gnark-crypto's g1JacExtended.addMixed skips the (0, 0) encoding the same way
    if (g1x_is_inf(p)) {
        p.x = a.x, p.y = a.y, p.zz = fp_one(), p.zzz = fp_one();
        return;
    }
    const Fp pp_ = fp_sub(fp_mul(a.x, p.zz), p.x), r = fp_sub(fp_mul(a.y, p.zzz), p.y);
    if (fp_is_zero(pp_)) {
        if (fp_is_zero(r)) g1x_dbl_aff(p, a);
        else g1x_set_inf(p);
        return;
    }
    const Fp pp = fp_sqr(pp_), ppp = fp_mul(pp_, pp), q = fp_mul(p.x, pp);
    const Fp x3 = fp_sub(fp_sub(fp_sqr(r), ppp), fp_dbl(q));
    p.y = fp_sub(fp_mul(r, fp_sub(q, x3)), fp_mul(p.y, ppp));
    p.x = x3;
    p.zz = fp_mul(p.zz, pp);
    p.zzz = fp_mul(p.zzz, ppp);
}
// p += q (add-2008-s: 12 M + 2 S), every special case handled
__device__ __forceinline__ void g1x_add(G1X& p, const G1X& q) {
    if (g1x_is_inf(q)) return;
    if (g1x_is_inf(p)) {
        p = q;
        return;
    }
    const Fp u1 = fp_mul(p.x, q.zz), s1 = fp_mul(p.y, q.zzz);
    const Fp pp_ = fp_sub(fp_mul(q.x, p.zz), u1), r = fp_sub(fp_mul(q.y, p.zzz), s1);
    if (fp_is_zero(pp_)) {
        if (fp_is_zero(r)) g1x_dbl(p);
        else g1x_set_inf(p);
        return;
    }
    const Fp pp = fp_sqr(pp_), ppp = fp_mul(pp_, pp), qq = fp_mul(u1, pp);
    const Fp x3 = fp_sub(fp_sub(fp_sqr(r), ppp), fp_dbl(qq));
    p.y = fp_sub(fp_mul(r, fp_sub(qq, x3)), fp_mul(s1, ppp));
    p.x = x3;
    p.zz = fp_mul(fp_mul(p.zz, q.zz), pp);
    p.zzz = fp_mul(fp_mul(p.zzz, q.zzz), ppp);
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
// affine image (canonical coordinates; infinity -> (0, 0)) of an XYZZ point: one inversion of ZZ * ZZZ
__device__ __forceinline__ G1Aff g1x_to_aff(const G1X& p) {
    G1Aff a;
    if (g1x_is_inf(p)) {
        a.x = fp_zero(), a.y = fp_zero();
        return a;
    }
    const Fp i = fp_inv(fp_mul(p.zz, p.zzz));
    a.x = fp_canon(fp_mul(p.x, fp_mul(i, p.zzz)));      // X / ZZ
    a.y = fp_canon(fp_mul(p.y, fp_mul(i, p.zz)));       // Y / ZZZ
    return a;
}

// ------------------------------------------------------------------------------------------------
// scalars -> signed window digits
// ------------------------------------------------------------------------------------------------
struct MsmArgs {
    const uint4* scalars;     // n x 32 B, the image of []fr.Element (4 x u64 little-endian)
    const uint4* points;      // n x 64 B, the image of []G1Affine
    size_t n;
    int c, W;                 // window bits, number of windows
    unsigned int nb;          // buckets per window = 2^(c-1)
    int scalars_mont;         // 1: the scalars are in Montgomery form (MultiExpConfig.ScalarsMont), converted on the fly
    unsigned int* count;      // W * nb bucket sizes
    unsigned int* offset;     // exclusive scan of count
    unsigned int* cursor;     // scatter cursors (a copy of offset)
    unsigned int* entries;    // point index | sign << 31, sorted by (window, bucket)
    unsigned int* big;        // [0] = number of big buckets, [1 + k] = their ids
    unsigned int big_threshold, big_cap;
    G1XPlanes buckets;        // W * nb
    G1XPlanes parts;          // W * nchunk chunk sums
    G1XPlanes wins;           // W window sums
    int chunk;                // buckets per lane of k_msm_reduce_chunks (a power of two)
};

__device__ __forceinline__ void msm_load_scalar(const MsmArgs& a, size_t i, u32 (&s)[8]) {
    const uint4 lo = a.scalars[2 * i], hi = a.scalars[2 * i + 1];
    Fr x = {{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}};
    if (a.scalars_mont) {
        Fr one = fr_zero();
        one.v[0] = 1;
        x = fr_mul(x, one);          // Montgomery product with the plain integer 1: the regular form, canonical
    }
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = x.v[j];
}
// F(window, bucket index = |d| - 1, negative) for every non-zero digit of the scalar.  C is a template parameter so that
// every limb index below is a compile-time constant (the scalar stays in registers).
template <int C, typename F>
__device__ __forceinline__ void msm_digits(const u32 (&s)[8], F&& f) {
    constexpr int W = (255 + C - 1) / C;
    constexpr u32 HALF = 1u << (C - 1);
    u32 carry = 0;
#pragma unroll
    for (int j = 0; j < W; j++) {
        const int bit = j * C, limb = bit >> 5, sh = bit & 31;          // C * (W - 1) < 255: limb <= 7
        u32 d = s[limb] >> sh;
        if (sh + C > 32 && limb < 7) d |= s[limb < 7 ? limb + 1 : 7] << (32 - sh);
        d = (d & ((1u << C) - 1u)) + carry;
        const bool neg = d > HALF;
        carry = neg ? 1u : 0u;
        const u32 mag = neg ? (1u << C) - d : d;
        if (mag) f(j, mag - 1u, neg);
    }
}

template <int C>
__global__ void __launch_bounds__(GKR_BLOCK) k_msm_count(MsmArgs a) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
        u32 s[8];
        msm_load_scalar(a, i, s);
        msm_digits<C>(s, [&](int j, u32 b, bool) { atomicAdd(&a.count[(size_t)j * a.nb + b], 1u); });
    }
}
template <int C>
__global__ void __launch_bounds__(GKR_BLOCK) k_msm_scatter(MsmArgs a) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
        u32 s[8];
        msm_load_scalar(a, i, s);
        msm_digits<C>(s, [&](int j, u32 b, bool neg) {
            const u32 pos = atomicAdd(&a.cursor[(size_t)j * a.nb + b], 1u);
            a.entries[pos] = (u32)i | (neg ? 0x80000000u : 0u);
        });
    }
}
// exclusive scan of the W * nb counts by ONE workgroup (a few hundred thousand words: microseconds), cursors initialised,
// the buckets above the threshold listed for k_msm_accumulate_big
#define MSM_SCAN_THREADS 1024
__global__ void __launch_bounds__(MSM_SCAN_THREADS) k_msm_scan(MsmArgs a) {
    __shared__ unsigned int part[MSM_SCAN_THREADS];
    const size_t total = (size_t)a.W * a.nb;
    const size_t per = (total + MSM_SCAN_THREADS - 1) / MSM_SCAN_THREADS;
    const size_t lo = min(total, per * threadIdx.x), hi = min(total, lo + per);
    unsigned int s = 0;
    for (size_t t = lo; t < hi; t++) s += a.count[t];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < MSM_SCAN_THREADS; d <<= 1) {
        const unsigned int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned int run = part[threadIdx.x] - s;
    for (size_t t = lo; t < hi; t++) {
        const unsigned int cnt = a.count[t];
        a.offset[t] = run;
        a.cursor[t] = run;
        if (cnt > a.big_threshold) {
            const unsigned int k = atomicAdd(&a.big[0], 1u);
            if (k < a.big_cap) a.big[1 + k] = (unsigned int)t;
        }
        run += cnt;
    }
}

// ------------------------------------------------------------------------------------------------
// bucket accumulation
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ G1Aff msm_entry_point(const MsmArgs& a, unsigned int e) {
    G1Aff p = g1_ld_aff(a.points, e & 0x7fffffffu);
    if (e & 0x80000000u) p.y = fp_neg(p.y);
    return p;
}
__global__ void __launch_bounds__(GKR_BLOCK) k_msm_accumulate(MsmArgs a) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)a.W * a.nb) return;
    const unsigned int cnt = a.count[t], start = a.offset[t];
    G1X acc;
    g1x_set_inf(acc);
    if (cnt <= a.big_threshold)
        for (unsigned int k = 0; k < cnt; k++) g1x_madd(acc, msm_entry_point(a, a.entries[start + k]));
    g1x_st(a.buckets, t, acc);        // a big bucket is overwritten by k_msm_accumulate_big (launched after this kernel)
}
// LDS tree over the workgroup's XYZZ partial sums; the result is in sh[0]
struct G1XShared {
    G1X p[GKR_BLOCK];
};
__device__ __forceinline__ void g1x_block_reduce(G1XShared& sh, G1X& mine) {
    sh.p[threadIdx.x] = mine;
    __syncthreads();
    for (int d = GKR_BLOCK / 2; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) {
            G1X x = sh.p[threadIdx.x];
            g1x_add(x, sh.p[threadIdx.x + d]);
            sh.p[threadIdx.x] = x;
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(GKR_BLOCK) k_msm_accumulate_big(MsmArgs a) {
    __shared__ G1XShared sh;
    const unsigned int nbig = min(a.big[0], a.big_cap);
    for (unsigned int k = blockIdx.x; k < nbig; k += gridDim.x) {
        const size_t t = a.big[1 + k];
        const unsigned int cnt = a.count[t], start = a.offset[t];
        G1X acc;
        g1x_set_inf(acc);
        for (unsigned int i = threadIdx.x; i < cnt; i += GKR_BLOCK) g1x_madd(acc, msm_entry_point(a, a.entries[start + i]));
        g1x_block_reduce(sh, acc);
        if (threadIdx.x == 0) g1x_st(a.buckets, t, sh.p[0]);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// window sums:  sum_b (b + 1) B_b
// ------------------------------------------------------------------------------------------------
// [k] p for a small k by double-and-add (k < 2^16)
__device__ __forceinline__ G1X g1x_mul_small(const G1X& p, unsigned int k) {
    G1X r;
    g1x_set_inf(r);
    for (int i = 31 - __clz(k | 1u); i >= 0; i--) {
        g1x_dbl(r);
        if ((k >> i) & 1u) g1x_add(r, p);
    }
    return r;
}
// lane (j, ci): buckets [ci * chunk, (ci + 1) * chunk) of window j from the top down: running = sum B, acc = sum of the
// running sums = sum (b - b0 + 1) B_b; the chunk's share of the window sum is acc + b0 * running
__global__ void __launch_bounds__(GKR_BLOCK) k_msm_reduce_chunks(MsmArgs a) {
    const unsigned int nchunk = a.nb / a.chunk;
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (size_t)a.W * nchunk) return;
    const unsigned int j = (unsigned int)(g / nchunk), ci = (unsigned int)(g % nchunk);
    const unsigned int b0 = ci * a.chunk;
    G1X running, acc;
    g1x_set_inf(running);
    g1x_set_inf(acc);
    for (int k = a.chunk - 1; k >= 0; k--) {
        g1x_add(running, g1x_ld(a.buckets, (size_t)j * a.nb + b0 + k));
        g1x_add(acc, running);
    }
    if (b0) g1x_add(acc, g1x_mul_small(running, b0));
    g1x_st(a.parts, g, acc);
}
// one workgroup per window: the sum of its chunk results
__global__ void __launch_bounds__(GKR_BLOCK) k_msm_reduce_windows(MsmArgs a) {
    __shared__ G1XShared sh;
    const unsigned int nchunk = a.nb / a.chunk;
    const unsigned int j = blockIdx.x;
    G1X acc;
    g1x_set_inf(acc);
    for (unsigned int ci = threadIdx.x; ci < nchunk; ci += GKR_BLOCK) g1x_add(acc, g1x_ld(a.parts, (size_t)j * nchunk + ci));
    g1x_block_reduce(sh, acc);
    if (threadIdx.x == 0) g1x_st(a.wins, j, sh.p[0]);
}

// ------------------------------------------------------------------------------------------------
// bn254.BatchScalarMultiplicationG1(base, scalars) (prove.go:177): out[i] = [s_i] base, affine.  One lane per scalar,
// left-to-right double-and-add with mixed additions, then the lane's own inversion (a tenth of its work).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(GKR_BLOCK) k_g1_batch_scalar_mul(MsmArgs a, G1Aff base, uint4* out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    u32 s[8];
    msm_load_scalar(a, i, s);
    G1X acc;
    g1x_set_inf(acc);
    for (int limb = 7; limb >= 0; limb--) {
        u32 w = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) w = (j == limb) ? s[j] : w;        // no dynamic register indexing
        for (int b = 31; b >= 0; b--) {
            g1x_dbl(acc);
            if ((w >> b) & 1u) g1x_madd(acc, base);
        }
    }
    g1_st_aff(out, i, g1x_to_aff(acc));
}

// synthetic scalars of the benchmark: pseudo-random canonical values below q, out[i] = limbs of (mix(i, seed))^7
__global__ void __launch_bounds__(GKR_BLOCK) k_msm_synth_scalars(uint4* out, size_t n, u32 seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr x = {{(u32)i ^ 0x9df123fu, (u32)(i >> 32) + 0xf45cu, seed, 0x2545f491u, (u32)i * 0x9e3779b9u, 3u, seed ^ 0x5bd1e995u, 0u}};   // < 2^224 < q
        x = fr_pow7(x);
        out[2 * i] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
        out[2 * i + 1] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
    }
}
