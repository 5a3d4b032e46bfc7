// kernels.hip.h -- gfx950 kernels of the GKR/sumcheck hot path.
//
// Device layout of a multilinear table ("bookkeeping table", reference poly/multilin.go:12): two
// limb planes of 16-byte words.  Element i = { lo[i] : limbs 0..3 , hi[i] : limbs 4..7 } (32-bit
// limbs of the Montgomery residue).  One lane owns one element / one index pair, so every global
// access is a coalesced 16 B-per-lane (1 KiB per wave) load or store.  The reference's AoS
// `[]fr.Element` exists only at the C-ABI boundary (k_aos_to_planes / k_planes_to_aos).
//
// Variable order follows the reference: a round binds the TOP index bit, i.e. pairs (i, i+mid)
// (poly/multilin.go:26-35).
#pragma once
#include <hip/hip_runtime.h>

#include "fr_bn254.h"

// The library is built from several translation units (gkr-mimc_amd/build.py): gkrhip.hip holds the host code and every
// non-template kernel; the heavy TEMPLATE kernels are instantiated in units of their own (kern_*.hip; the list is
// kernel_groups.h, `extern template` in gkrhip.hip), which include the same headers with GKR_KERNEL_TU defined -- there a
// non-template kernel becomes a template nobody instantiates (no second definition, no device code).
#ifdef GKR_KERNEL_TU
#define GKR_KERNEL template <int GKR_NEVER_INSTANTIATED = 0> __global__
#else
#define GKR_KERNEL __global__
#endif

#ifndef GKR_BLOCK
#define GKR_BLOCK 256   // threads per workgroup of every grid-shaped kernel (a build parameter for A/B runs: 128 or 256)
#endif
#define GKR_MAX_ARITY 4
#define GKR_MAX_EVALS 9   // cipher gate: degree 8 -> 9 evaluation points (sumcheck/prover.go:95, algo.go:57)
#define GKR_ACC_WORDS 9   // un-reduced 288-bit lane accumulators
// The launch-wide accumulator of the round kernels exists GKR_RACC_SLOTS times (stride GKR_RACC_STRIDE words): workgroup b
// adds into copy b mod 8 and the last workgroup sums the copies.  Hundreds of workgroups finishing together otherwise
// queue their 72 atomic adds on the same 72 addresses (512 adds per address for a round of 2^17 pairs).
#define GKR_RACC_SLOTS 8
#define GKR_RACC_STRIDE 128


// Wave priority of a round kernel: 0 for round 0, rising with the round index (capped at 3).  With several proofs in flight the
// lanes' kernels share SIMDs; the later -- shorter -- rounds of a layer going first ("shortest job first") lets a lane reach its
// next hash sooner while the long round 0 of another lane fills the gaps.  Same-box A/B, three interleaved runs each
// (profiles/r05_prio_rounds.txt): bN = 24 x 5 lanes 85.7 -> 89.0 M hashes/s, GMiMC bN = 22 x 12 104.1 -> 115.0, bN = 22 x 8 71.0 -> 78.1,
// bN = 20 x 24 60.8 -> 61.2; one proof alone unchanged.  Other mappings of the four levels (0233, 0333, 0112, 0223: profiles/
// r05_prio_rounds_mappings.txt) are equal or worse.
__device__ __forceinline__ void round_wave_priority(unsigned int p) {
    if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else if (p >= 3) __builtin_amdgcn_s_setprio(3);
}

struct Planes {
    uint4* lo;
    uint4* hi;
};
struct CPlanes {
    const uint4* lo;
    const uint4* hi;
};

__device__ __forceinline__ Fr ld_fr(const uint4* __restrict__ lo, const uint4* __restrict__ hi, size_t i) {
    const uint4 a = lo[i], b = hi[i];
    Fr r = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
    return r;
}
__device__ __forceinline__ void st_fr(uint4* __restrict__ lo, uint4* __restrict__ hi, size_t i, const Fr& x) {
    lo[i] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    hi[i] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
}

// ------------------------------------------------------------------------------------------------
// boundary: Go []fr.Element (AoS, 32 B per element) <-> limb planes
// ------------------------------------------------------------------------------------------------
// `bad` (host-mapped word) is set when an element is not below q: gnark-crypto keeps fr.Element canonical and the
// lazy-reduction bounds of the round kernels (u < 3q, d < 2q, fold results < 4q) assume exactly that of table entries
// SCALE: every element is multiplied by `factor` on the way (Montgomery product): R^2 turns a regular-form boundary image
// (big.Int words, prover/gadget/hints.go:202-205) into Montgomery form, the plain integer 1 turns Montgomery form into
// regular form -- the hint's conversions fused into the transposition the boundary needs anyway.
template <bool SCALE>
__global__ void __launch_bounds__(GKR_BLOCK) k_aos_to_planes(const uint4* __restrict__ aos, Planes out, size_t n,
                                                             unsigned int* bad, Fr factor) {
    const u32 q[8] = {FRQ0, FRQ1, FRQ2, FRQ3, FRQ4, FRQ5, FRQ6, FRQ7};
    bool any_bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 a = aos[2 * i], b = aos[2 * i + 1];
        const u32 v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        u32 br = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) (void)fr_subb(v[j], q[j], br, &br);
        any_bad |= br == 0;                           // no borrow: v >= q
        if (SCALE) {
            const Fr x = {{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]}};
            const Fr y = fr_mul(x, factor);
            a = make_uint4(y.v[0], y.v[1], y.v[2], y.v[3]);
            b = make_uint4(y.v[4], y.v[5], y.v[6], y.v[7]);
        }
        out.lo[i] = a;
        out.hi[i] = b;
    }
    if (any_bad) atomicOr(bad, 1u);
}
template <bool SCALE>
__global__ void __launch_bounds__(GKR_BLOCK) k_planes_to_aos(CPlanes in, uint4* __restrict__ aos, size_t n, Fr factor) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 a = in.lo[i], b = in.hi[i];
        if (SCALE) {
            const Fr x = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
            const Fr y = fr_mul(x, factor);
            a = make_uint4(y.v[0], y.v[1], y.v[2], y.v[3]);
            b = make_uint4(y.v[4], y.v[5], y.v[6], y.v[7]);
        }
        aos[2 * i] = a;
        aos[2 * i + 1] = b;
    }
}

// ------------------------------------------------------------------------------------------------
// fold: dst[i] = src[i] + r * (src[i+mid] - src[i])   for every table of the instance, same r
// (poly/multilin.go:26-36 applied by sumcheck/algo.go:46-51).  96 algorithmic bytes and one
// modular multiplication per output element per table: the HBM-bound kernel of the path.
// dst may alias src (each lane reads its own pair before writing its own output).
// ------------------------------------------------------------------------------------------------
struct FoldArgs {
    CPlanes src[GKR_MAX_ARITY + 1];
    Planes dst[GKR_MAX_ARITY + 1];
    int ntab;
    size_t mid;
    Fr r;
};
// NTAB is a template parameter so that all 2*NTAB*2 plane loads of an index are issued before the first
// multiplication (more bytes in flight per wave); the data is touched once, hence the non-temporal hints.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Fr ld_fr_nt(const uint4* __restrict__ lo, const uint4* __restrict__ hi, size_t i) {
    const u32x4 a = __builtin_nontemporal_load((const u32x4*)(lo + i)), b = __builtin_nontemporal_load((const u32x4*)(hi + i));
    Fr r = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
    return r;
}
__device__ __forceinline__ void st_fr_nt(uint4* __restrict__ lo, uint4* __restrict__ hi, size_t i, const Fr& x) {
    const u32x4 a = {x.v[0], x.v[1], x.v[2], x.v[3]}, b = {x.v[4], x.v[5], x.v[6], x.v[7]};
    __builtin_nontemporal_store(a, (u32x4*)(lo + i));
    __builtin_nontemporal_store(b, (u32x4*)(hi + i));
}
template <int NTAB>
__global__ void __launch_bounds__(GKR_BLOCK) k_fold(FoldArgs a) {
    const Fr r = a.r;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    // tables outermost: each table is an independent stream, and one table at a time keeps the register
    // footprint (hence the occupancy) of the single-table kernel; two indices per iteration so that four
    // 1 KiB loads per wave are in flight before the first product
#pragma unroll
    for (int t = 0; t < NTAB; t++) {
        const CPlanes src = a.src[t];
        const Planes dst = a.dst[t];
        size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        for (; i + stride < a.mid; i += 2 * stride) {
            const Fr lo0 = ld_fr_nt(src.lo, src.hi, i), hi0 = ld_fr_nt(src.lo, src.hi, i + a.mid);
            const Fr lo1 = ld_fr_nt(src.lo, src.hi, i + stride), hi1 = ld_fr_nt(src.lo, src.hi, i + stride + a.mid);
            st_fr_nt(dst.lo, dst.hi, i, fr_add(lo0, fr_mul(fr_sub(hi0, lo0), r)));
            st_fr_nt(dst.lo, dst.hi, i + stride, fr_add(lo1, fr_mul(fr_sub(hi1, lo1), r)));
        }
        if (i < a.mid) {
            const Fr lo0 = ld_fr_nt(src.lo, src.hi, i), hi0 = ld_fr_nt(src.lo, src.hi, i + a.mid);
            st_fr_nt(dst.lo, dst.hi, i, fr_add(lo0, fr_mul(fr_sub(hi0, lo0), r)));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// eq tables (poly/eq.go:41-89).  eq(q, i) factorises over index bits (the reference's chunked
// builder relies on the same identity, eq.go:74-88): with i = (i_hi << nlo) | i_lo,
//   eq(q, i) = eq(q[0:nhi], i_hi) * eq(q[nhi:], i_lo).
// k_eq_small builds one factor table (<= 2^13 entries) by the reference's doubling recurrence in a
// single workgroup; k_eq_expand writes the full table with one multiplication per element, summing
// over claims for the multi-claim layer (sumcheck/prover.go:128-138: Eq = sum_j rho^j eq(q_j, .),
// rho^j folded into factor table j's seed).
// ------------------------------------------------------------------------------------------------
struct EqSmallArgs {
    Planes out;          // nclaims tables, table j at element offset j * tab_stride
    const Fr* q;         // nclaims * q_stride coordinates (device), this factor uses q[j*q_stride + q_off .. +nbits)
    const Fr* seeds;     // nclaims seeds (multipliers)
    int nbits, q_stride, q_off;
    size_t tab_stride;
};
GKR_KERNEL void __launch_bounds__(1024) k_eq_small(EqSmallArgs a) {
    const int j = blockIdx.x;
    uint4* lo = a.out.lo + (size_t)j * a.tab_stride;
    uint4* hi = a.out.hi + (size_t)j * a.tab_stride;
    const Fr* q = a.q + (size_t)j * a.q_stride + a.q_off;
    const int n = a.nbits;
    if (threadIdx.x == 0) st_fr(lo, hi, 0, a.seeds[j]);
    __syncthreads();
    for (int i = 0; i < n; i++) {
        const Fr r = q[i];
        for (size_t t = threadIdx.x; t < ((size_t)1 << i); t += blockDim.x) {
            const size_t J = t << (n - i);
            const size_t JN = J + ((size_t)1 << (n - 1 - i));
            const Fr cur = ld_fr(lo, hi, J);
            const Fr up = fr_mul(r, cur);          // t[JN] = q_i * t[J]
            st_fr(lo, hi, JN, up);
            st_fr(lo, hi, J, fr_sub(cur, up));     // t[J] -= t[JN]
        }
        __syncthreads();
    }
}

struct EqExpandArgs {
    Planes out;
    CPlanes thi, tlo;    // factor tables: claim j at offset j*hi_stride / j*lo_stride
    size_t hi_stride, lo_stride;
    int nclaims, nlo;    // i_lo has nlo bits
    size_t n;
};
GKR_KERNEL void __launch_bounds__(GKR_BLOCK) k_eq_expand(EqExpandArgs a) {
    const size_t mask = ((size_t)1 << a.nlo) - 1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t ih = i >> a.nlo, il = i & mask;
        // dot product over the claims with deferred reduction: plain 512-bit products accumulate in 17 limbs and
        // are reduced once per 64 claims (fr_mac_wide / fr_redc_wide) instead of once per product
        Fr acc = fr_zero();
        for (int j0 = 0; j0 < a.nclaims; j0 += 64) {
            u32 T[FR_WIDE_LIMBS];
#pragma unroll
            for (int w = 0; w < FR_WIDE_LIMBS; w++) T[w] = 0;
            const int j1 = min(j0 + 64, a.nclaims);
            for (int j = j0; j < j1; j++) {
                const Fr h = ld_fr(a.thi.lo, a.thi.hi, (size_t)j * a.hi_stride + ih);
                const Fr l = ld_fr(a.tlo.lo, a.tlo.hi, (size_t)j * a.lo_stride + il);
                fr_mac_wide(T, h, l);
            }
            u32 red[9];
            fr_redc_wide(red, T);
            acc = fr_add(acc, fr_canon_lt16q(red));
        }
        st_fr(a.out.lo, a.out.hi, i, acc);
    }
}

// ------------------------------------------------------------------------------------------------
// gates (circuit/gates.go:9-21; circuit/gates/cipher.go:25-55, circuit/gates/copy.go:15-22): the descriptor family
//     Eval(xs...) = (sum of the inputs selected by mask + Ark)^POWER,   POWER = 1 or 7
// (identity: mask 1, Ark 0, POWER 1; cipher: (xs[0] + xs[1] + Ark)^7).  POWER and ARITY are compile-time (they fix
// the instruction stream and the register footprint), the mask is a launch-wide runtime value.
// ------------------------------------------------------------------------------------------------
template <int POWER, int ARITY>
__device__ __forceinline__ Fr gate_eval(const Fr* x, const Fr& ark, unsigned mask) {
    Fr s = ark;
#pragma unroll
    for (int k = 0; k < ARITY; k++)
        if ((mask >> k) & 1u) s = fr_add(s, x[k]);
    return POWER == 7 ? fr_pow7(s) : s;
}

// layer assignment: out = Gate(in0, in1, ...)   (circuit/circuit.go:48-64 -> Gate.EvalBatch)
struct AssignArgs {
    CPlanes in[GKR_MAX_ARITY];
    Planes out;
    int arity;
    unsigned mask;
    size_t n;
    Fr ark;
};
template <int POWER, int ARITY>
__global__ void __launch_bounds__(GKR_BLOCK) k_gate_eval_batch(AssignArgs a) {
    const Fr ark = a.ark;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
        Fr x[ARITY];
#pragma unroll
        for (int k = 0; k < ARITY; k++) x[k] = ld_fr(a.in[k].lo, a.in[k].hi, i);
        st_fr(a.out.lo, a.out.hi, i, gate_eval<POWER, ARITY>(x, ark, a.mask));
    }
}

// ------------------------------------------------------------------------------------------------
// per-round partial evaluation (sumcheck/algo.go:54-205):
//   evals[t] = sum_{x < mid} Eq(t,x) * Gate(X_1(t,x), ..., X_n(t,x)),   t = 0 .. deg+1
// with T(t,x) = T[x] + t*(T[x+mid]-T[x]) obtained by repeated addition of the difference exactly as
// the reference does.  One lane per index pair (grid-stride); each lane keeps NEV un-reduced 288-bit
// accumulators; wave reduction by cross-lane adds of 64-bit limb sums, LDS across the waves of the
// block; one "limb-split" partial (NEV x 9 u64 lanes) per block.  k_reduce_partials sums the blocks.
// The limb-split form is an exact integer sum, so it can also be all-reduced across GPUs with a
// plain u64 ncclSum before the single host-side reduction mod q (fr_host.h reduce_limbsplit).
// ------------------------------------------------------------------------------------------------
struct Acc9 {
    u32 w[GKR_ACC_WORDS];
};
__device__ __forceinline__ void acc_add(Acc9& a, const Fr& x) {
    u32 c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) a.w[j] = fr_addc(a.w[j], x.v[j], c, &c);
    a.w[8] += c;
}

// Block reduction of NS un-reduced accumulators (NS*9 32-bit words per lane) to NS*9 exact 64-bit integer
// sums, through LDS: the lanes store CH words at a time transposed ([word][lane], row stride 257 words so
// that the column readers fall on different banks), then 4*CH threads each add the 64 lanes of one wave
// for one word.  (A cross-lane butterfly needs 6 dependent ds_bpermute round trips per word; hipcc
// serialises them, ~20 us per launch for 72 words -- more than the arithmetic of a small round.)
// ATOMIC: the block adds its NW sums into one global accumulator (`out`, NW words shared by all blocks; exact
// integer adds commute) instead of writing a per-block partial
template <int NS, int CH, bool ATOMIC = false>
__device__ __forceinline__ void block_reduce_acc_buf(const Acc9 (&acc)[NS], unsigned long long* __restrict__ out,
                                                     u32 (*tr)[GKR_BLOCK + 1] /* [CH] */,
                                                     unsigned long long (*red)[NS * GKR_ACC_WORDS] /* [GKR_BLOCK/64] */) {
    constexpr int NW = NS * GKR_ACC_WORDS;
    static_assert(NW % CH == 0 && CH * (GKR_BLOCK / 64) <= GKR_BLOCK && NW <= GKR_BLOCK, "chunking");
    const int tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < NW / CH; p++) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            constexpr int dummy = 0;
            (void)dummy;
            const int wi = p * CH + c;
            tr[c][tid] = acc[wi / GKR_ACC_WORDS].w[wi % GKR_ACC_WORDS];
        }
        __syncthreads();
        if (tid < CH * (GKR_BLOCK / 64)) {
            const int c = tid % CH, q = tid / CH;
            unsigned long long s = 0;
#pragma unroll 16
            for (int i = 0; i < 64; i++) s += tr[c][q * 64 + ((i + 8 * q) & 63)];
            red[q][p * CH + c] = s;
        }
        __syncthreads();
    }
    if (tid < NW) {
        unsigned long long s = 0;
#pragma unroll
        for (int q = 0; q < GKR_BLOCK / 64; q++) s += red[q][tid];
        if (ATOMIC) {
            if (s) (void)__hip_atomic_fetch_add(out + (blockIdx.x % GKR_RACC_SLOTS) * GKR_RACC_STRIDE + tid, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            out[tid] = s;
        }
    }
}
template <int NS, int CH, bool ATOMIC = false>
__device__ __forceinline__ void block_reduce_acc(const Acc9 (&acc)[NS], unsigned long long* __restrict__ out) {
    __shared__ u32 tr[CH][GKR_BLOCK + 1];
    __shared__ unsigned long long red[GKR_BLOCK / 64][NS * GKR_ACC_WORDS];
    block_reduce_acc_buf<NS, CH, ATOMIC>(acc, out, tr, red);
}

// Every block has added its sums into the shared accumulator `racc` (nwords words, zero at launch); the
// last-arriving block hands them to the host (host-mapped buffer, then the sequence flag the host polls) and resets
// accumulator and arrival counter for the next launch of the lane.  Agent-scope release by lane 0 after the block's
// atomics have drained; acquire before re-reading.
__device__ __forceinline__ void publish_sums(unsigned long long* racc, unsigned int* counter, unsigned long long* host_out,
                                             unsigned int* host_flag, unsigned int seq, int nwords, unsigned int* s_last) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int prev = atomicAdd(counter, 1u);
        const unsigned int last = (prev == gridDim.x - 1) ? 1u : 0u;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *s_last = last;
    }
    __syncthreads();
    if (*s_last) {
        if ((int)threadIdx.x < nwords) {
            unsigned long long sum = 0;
#pragma unroll
            for (int sl = 0; sl < GKR_RACC_SLOTS; sl++) {
                sum += __hip_atomic_load(racc + sl * GKR_RACC_STRIDE + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                racc[sl * GKR_RACC_STRIDE + threadIdx.x] = 0;
            }
            host_out[threadIdx.x] = sum;
        }
        // (the arrival counter is reset with the accumulators, in front of the ONE fence every writing lane needs anyway; the flag's
        // release store orders lane 0 behind the barrier: a second system-scope fence here cost every round ~1.5 us)
        if (threadIdx.x == 0) *counter = 0;
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

struct PartialEvalArgs {
    CPlanes eq;
    CPlanes x[GKR_MAX_ARITY];
    size_t mid;
    Fr ark;
    unsigned mask;                 // inputs that enter the gate's sum
    unsigned long long* partials;  // [gridDim.x][NEV][9]   (host_flag == nullptr: separate reduction kernel)
    // host_flag != nullptr: the sums go straight to the host, as the fused round kernels hand theirs over
    unsigned long long* racc;      // NEV*9-word accumulator shared by the blocks
    unsigned int* counter;
    unsigned long long* host_out;
    unsigned int* host_flag;
    unsigned int seq;
};

template <int POWER, int ARITY, int NEV>
__global__ void __launch_bounds__(GKR_BLOCK, 2) k_partial_eval(PartialEvalArgs a) {
    Acc9 acc[NEV];
#pragma unroll
    for (int t = 0; t < NEV; t++)
#pragma unroll
        for (int j = 0; j < GKR_ACC_WORDS; j++) acc[t].w[j] = 0;
    const Fr ark = a.ark;
    const size_t mid = a.mid;

    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < mid; i += (size_t)gridDim.x * blockDim.x) {
        Fr e = ld_fr(a.eq.lo, a.eq.hi, i);
        Fr de = fr_sub(ld_fr(a.eq.lo, a.eq.hi, i + mid), e);
        Fr x[ARITY], d[ARITY];
#pragma unroll
        for (int k = 0; k < ARITY; k++) {
            x[k] = ld_fr(a.x[k].lo, a.x[k].hi, i);
            d[k] = fr_sub(ld_fr(a.x[k].lo, a.x[k].hi, i + mid), x[k]);
        }
        // t = 0 .. NEV-1: T(t) = T(t-1) + (T_hi - T_lo).  The loop stays rolled (one gate evaluation of
        // code, I-cache resident); the accumulators rotate through acc[0] so that every index is a
        // compile-time constant (registers, not scratch).  After NEV iterations they are back in place.
#pragma unroll 1
        for (int t = 0; t < NEV; t++) {
            acc_add(acc[0], fr_mul(e, gate_eval<POWER, ARITY>(x, ark, a.mask)));
            const Acc9 first = acc[0];
#pragma unroll
            for (int u = 0; u + 1 < NEV; u++) acc[u] = acc[u + 1];
            acc[NEV - 1] = first;
            e = fr_add(e, de);
#pragma unroll
            for (int k = 0; k < ARITY; k++) x[k] = fr_add(x[k], d[k]);
        }
    }

    if (a.host_flag) {
        __shared__ unsigned int s_last;
        block_reduce_acc<NEV, 27, true>(acc, a.racc);
        publish_sums(a.racc, a.counter, a.host_out, a.host_flag, a.seq, NEV * GKR_ACC_WORDS, &s_last);
    } else {
        block_reduce_acc<NEV, 27>(acc, a.partials + (size_t)blockIdx.x * (NEV * GKR_ACC_WORDS));
    }
}

// sum the per-block partials: out[k] = sum_b partials[b][k],  k < nwords
GKR_KERNEL void __launch_bounds__(GKR_BLOCK) k_reduce_partials(const unsigned long long* __restrict__ partials,
                                                               unsigned long long* __restrict__ out, int nblocks,
                                                               int nwords) {
    __shared__ unsigned long long red[GKR_BLOCK];
    for (int k = blockIdx.x; k < nwords; k += gridDim.x) {
        unsigned long long s = 0;
        for (int b = threadIdx.x; b < nblocks; b += blockDim.x) s += partials[(size_t)b * nwords + k];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int off = GKR_BLOCK / 2; off >= 1; off >>= 1) {
            if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[k] = red[0];
        __syncthreads();
    }
}

// gather element 0 of up to 5 tables into a small AoS buffer (finalClaims, sumcheck/prover.go:79-86)
struct Gather0Args {
    CPlanes t[GKR_MAX_ARITY + 1];
    int ntab;
    uint4* out;  // AoS: 2 x uint4 per element
};
GKR_KERNEL void k_gather0(Gather0Args a) {
    const int t = threadIdx.x;
    if (t < a.ntab) {
        a.out[2 * t] = a.t[t].lo[0];
        a.out[2 * t + 1] = a.t[t].hi[0];
    }
}

// ------------------------------------------------------------------------------------------------
// wire-format helpers of the production caller (prover/gadget/hints.go)
// ------------------------------------------------------------------------------------------------
// bulk Montgomery <-> regular conversion on the boundary (AoS) layout: x <- x * factor / 2^256 mod q.
// factor = 1 (as a plain integer): fr.Element.FromMont / ToBigIntRegular (hints.go:141,231-268);
// factor = R^2: SetBigInt of a reduced value (hints.go:136-137,202-205).
GKR_KERNEL void __launch_bounds__(GKR_BLOCK) k_convert_aos(uint4* __restrict__ data, size_t n, Fr factor) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 a = data[2 * i], b = data[2 * i + 1];
        const Fr x = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
        const Fr y = fr_mul(x, factor);
        data[2 * i] = make_uint4(y.v[0], y.v[1], y.v[2], y.v[3]);
        data[2 * i + 1] = make_uint4(y.v[4], y.v[5], y.v[6], y.v[7]);
    }
}

// hash.Arks (hash/ark.go:232-336), Montgomery limbs
struct ArkLimbs {
    unsigned long long l[4];
};
__device__ __constant__ ArkLimbs d_arks[100] = {
#include "arks_bn254.inc"
};
__device__ __forceinline__ Fr ark_fr(int i) {
    Fr r;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        r.v[2 * k] = (u32)d_arks[i].l[k];
        r.v[2 * k + 1] = (u32)(d_arks[i].l[k] >> 32);
    }
    return r;
}
// batched hash.MimcKeyedPermutation(x[i], key[i]) (hash/mimc.go:31-39): the body of HashHint.Call
// (hints.go:134-145), which the solver invokes once per hash
GKR_KERNEL void __launch_bounds__(GKR_BLOCK) k_mimc_permutation(CPlanes x, CPlanes key, Planes out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr res = ld_fr(x.lo, x.hi, i);
        const Fr k = ld_fr(key.lo, key.hi, i);
#pragma unroll 1
        for (int r = 0; r < 91; r++) res = fr_pow7(fr_add(fr_add(res, k), ark_fr(r)));
        st_fr(out.lo, out.hi, i, res);
    }
}
