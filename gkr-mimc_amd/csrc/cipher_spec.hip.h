// cipher_spec.hip.h -- SPECULATIVE small rounds of the single-point cipher-gate sumcheck (a proof alone on the GPU).
//
// sumcheck.Prove is serial in the rounds (sumcheck/prover.go:70-76): round k's kernel needs r_{k-1}, the hash of round
// k-1's polynomial, and the host needs round k's sums before it can hash again -- so every small round costs the
// host's hash (~34 us) PLUS a kernel's latency and the hand-off (8..25 us) during which the host only waits.
//
// The monomial sums of round k are polynomials of degree 7 in the previous challenge:
//     M_j(r) = sum_x W_k(x) u_x(r)^(7-j) d_x(r)^j,      u_x, d_x linear in r (the tables of round k are the tables of round
//     k-1 folded with r: poly/multilin.go:27-34),
// so eight values M_j(0), ..., M_j(7) determine M_j(r_{k-1}) exactly.  k_cipher_round_spec computes them WITHOUT r_{k-1}:
// it takes r_{k-2} (which the host publishes one hash earlier), folds the tables of round k-2 with it (= the tables of round
// k-1, which it also stores for the next launch), and runs round k once per candidate value rho = 0..7 of r_{k-1}
// (blockIdx.y = rho; the fold with rho is a product by a small constant).  All of it happens while the host hashes round
// k-1; when that hash is done the host interpolates the eight candidates at the true r_{k-1} (Lagrange basis on 0..7, ~130
// host products) and hashes round k at once.  Exact field arithmetic on both sides: the interpolated sums ARE the sums the
// one-pass kernels produce, hence the same coefficients, challenges and transcript (parity-tested with the path forced
// on and off).
//
// Work per candidate lane: 4 folds of U = K + S (the gate only sees the sum, and folding is linear), 2 products by rho, the
// 17..18 products of the monomial schedule -- about one k_cipher_round_lat pair; the eight candidates are extra
// PARALLELISM on a GPU that the small rounds leave idle, not extra latency.  A ninth row of workgroups (blockIdx.y = 8)
// folds K and S separately and stores them (the next launch's input, and the host tail's export).
#pragma once
#include "cipher_round.hip.h"
#include "linear_round.hip.h"

#define GKR_SPEC_CAND 8                                         // candidates rho = 0 .. 7: degree 7 in r
#define GKR_SPEC_SET_WORDS (GKR_RACC_SLOTS * GKR_RACC_STRIDE)   // accumulator words per candidate (striped like the round kernels')
#define GKR_SPEC_OUT_WORDS (GKR_SPEC_CAND * GKR_CR_NSUM * 4)    // 64 canonical elements (4 u64 each) to the host per launch
#define GKR_SPEC_BUF_WORDS 640                                  // one host-mapped result buffer: sums, flag word at 576, diagnostics from 580
#define GKR_SPEC_FLAG_WORD 576
#define GKR_SPEC_DIAG_WORD 580

struct CipherSpecArgs {
    CPlanes k_src, s_src;   // prefolded == 0: the tables of round k-2 (8P entries); else the tables of round k-1 (4P entries)
    Planes k_dst, s_dst;    // the tables of round k-1 (4P entries), written by row 8 (lo == nullptr: not stored)
    CPlanes wt;             // W_k: P entries (pointer already at the level)
    size_t P;               // index pairs of round k (one per lane and candidate)
    Fr r;                   // r_{k-2} (prefolded == 0 and chal == nullptr)
    Fr ark;
    Fr rho[GKR_SPEC_CAND];  // Montgomery forms of 0 .. 7
    unsigned long long* partials;   // GKR_SPEC_CAND sets of GKR_SPEC_SET_WORDS words, zero at launch, reset by the last workgroup
    unsigned int* counter;
    unsigned long long* host_out;   // host-mapped: GKR_SPEC_BUF_WORDS words
    unsigned int seq;
    unsigned int need_m0;
    unsigned int prefolded;
    unsigned long long* tail_tables;   // host-mapped or nullptr: the tables of round k-1 (4P entries of K, then 4P of S, 4 u64 each)
    const unsigned long long* chal;    // challenge slot for r_{k-2} (pre-launched), or nullptr
    unsigned long long* chal_dev;
    unsigned int chal_seq;
};

__device__ __forceinline__ void spec_export(unsigned long long* dst, const Fr& x) {
#pragma unroll
    for (int l = 0; l < 4; l++) dst[l] = (unsigned long long)x.v[2 * l] | ((unsigned long long)x.v[2 * l + 1] << 32);
}
// lo + r * (hi - lo), canonical (poly/multilin.go:32-34)
__device__ __forceinline__ Fr spec_fold(const Fr& lo, const Fr& hi, const Fr& r) {
    return fr_add(lo, fr_reduce_once(fr_mont_mul_raw(fr_sub(hi, lo), r)));
}

// Hand-off of the candidates' sums.  Like publish_sums (kernels.hip.h), but the last workgroup also REDUCES every sum: lane v
// gathers the nine limb-split words of value v = candidate * 8 + j over the accumulator stripes, carries them into
// lo (256 bits) + top * 2^256 and stores lo mod q + top * R mod q as a canonical element (4 u64) -- two products on a
// GPU that is waiting anyway, instead of 56 reductions on the host inside the serial chain.
__device__ __forceinline__ void spec_publish(unsigned long long* racc, unsigned int* counter, unsigned long long* host_out,
                                             unsigned int* host_flag, unsigned int seq, unsigned int nblocks, unsigned int* s_last,
                                             int ncand = GKR_SPEC_CAND, int nsum = GKR_CR_NSUM) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int prev = atomicAdd(counter, 1u);
        const unsigned int last = (prev == nblocks - 1) ? 1u : 0u;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *s_last = last;
    }
    __syncthreads();
    if (*s_last) {
        if ((int)threadIdx.x < ncand * nsum) {
            const int cand = threadIdx.x / nsum, j = threadIdx.x % nsum;
            unsigned long long* base = racc + (size_t)cand * GKR_SPEC_SET_WORDS + (size_t)j * GKR_ACC_WORDS;
            unsigned long long w[GKR_ACC_WORDS];
#pragma unroll
            for (int t = 0; t < GKR_ACC_WORDS; t++) {
                unsigned long long sum = 0;
#pragma unroll
                for (int sl = 0; sl < GKR_RACC_SLOTS; sl++) {
                    sum += __hip_atomic_load(base + sl * GKR_RACC_STRIDE + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    base[sl * GKR_RACC_STRIDE + t] = 0;
                }
                w[t] = sum;
            }
            // value = sum_t w[t] 2^(32 t): carry into eight 32-bit limbs and a top part (< 2^40 for any launch this kernel takes)
            Fr lo;
            unsigned long long c = 0;
#pragma unroll
            for (int t = 0; t < 8; t++) {
                c += w[t] & 0xffffffffull;
                lo.v[t] = (u32)c;
                c = (c >> 32) + (w[t] >> 32);
            }
            c += w[8];                                   // every sum is over < 2^17 lanes of 32-bit words: no overflow
            Fr top = fr_zero();
            top.v[0] = (u32)c;
            top.v[1] = (u32)(c >> 32);
            const Fr r2 = {{0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u}};   // R^2 mod q
            Fr a0, a1;
            fr_mont_mul2_raw(a0, a1, lo, fr_one(), top, r2);          // lo * R / R = lo mod q;  top * R^2 / R = top * 2^256 mod q
            const Fr v = fr_add(fr_reduce_once(a0), fr_reduce_once(a1));
#pragma unroll
            for (int l = 0; l < 4; l++) host_out[4 * threadIdx.x + l] = (unsigned long long)v.v[2 * l] | ((unsigned long long)v.v[2 * l + 1] << 32);
        }
        // (the arrival counter is reset with the accumulators, in front of the ONE fence every writing lane needs anyway; the flag's
        // release store orders lane 0 behind the barrier: a second system-scope fence here cost every round ~1.5 us)
        if (threadIdx.x == 0) *counter = 0;
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

GKR_KERNEL void __launch_bounds__(GKR_BLOCK, 1) k_cipher_round_spec(CipherSpecArgs a) {
    __shared__ unsigned int s_last;
    __builtin_amdgcn_s_setprio(3);
    const size_t P = a.P;
    const size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = x < P;
    const unsigned row = blockIdx.y;                 // 0..7: candidate, 8: the fold-and-store row
    const int nin = a.prefolded ? 4 : 8;
    // the table entries and the weight do not depend on the challenge: requested BEFORE the wait (their latency overlaps it)
    Fr kin[8], sin[8], W = fr_zero();
#pragma unroll
    for (int i = 0; i < 8; i++) kin[i] = sin[i] = fr_zero();
    if (live) {
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (i < nin) {
                kin[i] = ld_fr(a.k_src.lo, a.k_src.hi, x + (size_t)i * P);
                sin[i] = ld_fr(a.s_src.lo, a.s_src.hi, x + (size_t)i * P);
            }
        if (row < GKR_SPEC_CAND) W = ld_fr(a.wt.lo, a.wt.hi, x);
    }
    Fr r = a.r, r_unused = a.r;
    if (!a.prefolded && a.chal && !wait_challenge(a.chal, a.chal_dev, a.chal_seq, r, r_unused, a.host_out + GKR_SPEC_DIAG_WORD)) return;

    Acc9 acc[GKR_CR_NSUM];
#pragma unroll
    for (int t = 0; t < GKR_CR_NSUM; t++)
#pragma unroll
        for (int j = 0; j < GKR_ACC_WORDS; j++) acc[t].w[j] = 0;

    if (row == GKR_SPEC_CAND) {
        // the tables of round k-1: stored for the next launch, exported for the host tail
        if (live && (a.k_dst.lo || a.tail_tables)) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const Fr kf = a.prefolded ? kin[i] : spec_fold(kin[i], kin[i + 4], r);
                const Fr sf = a.prefolded ? sin[i] : spec_fold(sin[i], sin[i + 4], r);
                if (a.k_dst.lo) {
                    st_fr(a.k_dst.lo, a.k_dst.hi, x + (size_t)i * P, kf);
                    st_fr(a.s_dst.lo, a.s_dst.hi, x + (size_t)i * P, sf);
                }
                if (a.tail_tables) {
                    spec_export(a.tail_tables + 4 * (x + (size_t)i * P), kf);
                    spec_export(a.tail_tables + 4 * (4 * P + x + (size_t)i * P), sf);
                }
            }
        }
    } else if (live) {
        // U = K + S at the four entries of round k-1's table this pair is made of (folding is linear: fold the sums)
        Fr f[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const Fr lo = fr_add(kin[i], sin[i]);
            f[i] = a.prefolded ? lo : spec_fold(lo, fr_add(kin[i + 4], sin[i + 4]), r);
        }
        // round k pairs (y, y + 2P) of that table fold with rho into (lo, hi) = entries (x, x + P) of round k's table:
        // lo = f0 + rho (f2 - f0), hi = f1 + rho (f3 - f1);   u = lo + ark,  d = hi - lo
        const Fr rho = a.rho[row];
        const Fr dA = fr_sub(f[1], f[0]), dC = fr_sub(f[3], f[2]);
        Fr g, h;
        fr_mont_mul2_raw(g, h, fr_sub(f[2], f[0]), rho, fr_sub(dC, dA), rho);
        const Fr u = fr_add_raw(fr_add_raw(f[0], fr_reduce_once(g)), a.ark);     // < 3q
        const Fr d = fr_add_raw(dA, fr_reduce_once(h));                          // < 2q
        // the monomial schedule of k_cipher_round_lat (cipher_round.hip.h): lazy products, independent pairs interleaved
        Fr p, r2, A, B, C, D, U4, D4, X0, X1, t, t2;
        fr_mont_mul2_raw(p, r2, u, u, d, d);
        fr_mont_mul2_raw(A, B, p, u, p, d);          // u^3, u^2 d
        fr_mont_mul2_raw(C, D, u, r2, r2, d);        // u d^2, d^3
        fr_mont_mul2_raw(U4, D4, p, p, r2, r2);      // u^4, d^4
        fr_mont_mul2_raw(X0, X1, W, U4, W, D4);
        if (a.need_m0) {
            fr_mont_mul2_raw(t, t2, X0, A, X0, B);
            acc_add_raw(acc[0], t);                  // W u^7
        } else {
            t2 = fr_mont_mul_raw(X0, B);
        }
        acc_add_raw(acc[1], t2);                     // W u^6 d
        fr_mont_mul2_raw(t, t2, X0, C, X0, D);
        acc_add_raw(acc[2], t);                      // W u^5 d^2
        acc_add_raw(acc[3], t2);                     // W u^4 d^3
        fr_mont_mul2_raw(t, t2, X1, A, X1, B);
        acc_add_raw(acc[4], t);                      // W u^3 d^4
        acc_add_raw(acc[5], t2);                     // W u^2 d^5
        fr_mont_mul2_raw(t, t2, X1, C, X1, D);
        acc_add_raw(acc[6], t);                      // W u d^6
        acc_add_raw(acc[7], t2);                     // W d^7
    }
    // exact integer sums per candidate; row 8 contributes zeros (the reduction skips zero words) but takes part in the
    // arrival count: its stores must be complete before the host, or the next launch, is told
    block_reduce_acc<GKR_CR_NSUM, 18, true>(acc, a.partials + (size_t)(row < GKR_SPEC_CAND ? row : 0) * GKR_SPEC_SET_WORDS);
    spec_publish(a.partials, a.counter, a.host_out, (unsigned int*)(a.host_out + GKR_SPEC_FLAG_WORD), a.seq, gridDim.x * gridDim.y, &s_last);
}

// ------------------------------------------------------------------------------------------------------------------
// The same for the single-point round of a LINEAR gate (linear_round.hip.h): its two sums are LINEAR in the previous
// challenge, M_j(r) = (1 - r) M_j(0) + r M_j(1), and the candidates 0 and 1 need no arithmetic of their own: folding the
// tables of round k-1 with 0 or 1 selects their lower or upper half.  Grid rows 0, 1: the candidates; row 2: fold with
// r_{k-2} and store (and export) the tables of round k-1.
// ------------------------------------------------------------------------------------------------------------------
#define GKR_LSPEC_CAND 2
struct LinearSpecArgs {
    CPlanes src[GKR_MAX_ARITY];   // prefolded == 0: the tables of round k-2 (8P entries); else of round k-1 (4P entries)
    Planes dst[GKR_MAX_ARITY];    // the tables of round k-1 (4P entries; lo == nullptr: not stored)
    CPlanes wt;
    size_t P;
    Fr r;
    Fr ark;
    int arity;
    unsigned sum_mask;
    unsigned long long* partials;   // GKR_LSPEC_CAND sets of GKR_SPEC_SET_WORDS words
    unsigned int* counter;
    unsigned long long* host_out;   // host-mapped: GKR_SPEC_BUF_WORDS words (4 canonical elements: M_0(0), M_1(0), M_0(1), M_1(1))
    unsigned int seq;
    unsigned int need_m0;
    unsigned int prefolded;
    unsigned long long* tail_tables;   // host-mapped or nullptr: the tables of round k-1, table t at 4 * t * 4P (4P entries of 4 u64)
    const unsigned long long* chal;
    unsigned long long* chal_dev;
    unsigned int chal_seq;
};

GKR_KERNEL void __launch_bounds__(GKR_BLOCK) k_linear_round_spec(LinearSpecArgs a) {
    __shared__ unsigned int s_last;
    __builtin_amdgcn_s_setprio(3);
    const size_t P = a.P;
    const size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = x < P;
    const unsigned row = blockIdx.y;
    Fr r = a.r, r_unused = a.r;
    if (!a.prefolded && a.chal && !wait_challenge(a.chal, a.chal_dev, a.chal_seq, r, r_unused, a.host_out + GKR_SPEC_DIAG_WORD)) return;
    Acc9 acc[GKR_LR_NSUM];
#pragma unroll
    for (int t = 0; t < GKR_LR_NSUM; t++)
#pragma unroll
        for (int j = 0; j < GKR_ACC_WORDS; j++) acc[t].w[j] = 0;
    // entry y of round k-1's table (4P entries): read, or folded from the entries (y, y + 4P) of round k-2's
    auto entry = [&](int t, size_t y) {
        const Fr lo = ld_fr(a.src[t].lo, a.src[t].hi, y);
        return a.prefolded ? lo : spec_fold(lo, ld_fr(a.src[t].lo, a.src[t].hi, y + 4 * P), r);
    };
    if (row == GKR_LSPEC_CAND) {
        if (live && (a.dst[0].lo || a.tail_tables)) {
            for (int t = 0; t < a.arity; t++)
                for (int i = 0; i < 4; i++) {
                    const Fr f = entry(t, x + (size_t)i * P);
                    if (a.dst[0].lo) st_fr(a.dst[t].lo, a.dst[t].hi, x + (size_t)i * P, f);
                    if (a.tail_tables) spec_export(a.tail_tables + 4 * ((size_t)t * 4 * P + x + (size_t)i * P), f);
                }
        }
    } else if (live) {
        // candidate rho = row: round k's tables are the lower (0) or upper (1) half of round k-1's; this pair is (x, x + P) of it
        const size_t base = x + 2 * P * row;
        Fr u = a.ark, d = fr_zero();
        for (int t = 0; t < a.arity; t++)
            if ((a.sum_mask >> t) & 1u) {
                const Fr lo = entry(t, base), hi = entry(t, base + P);
                u = fr_add(u, lo);
                d = fr_add(d, fr_sub(hi, lo));
            }
        const Fr W = ld_fr(a.wt.lo, a.wt.hi, x);
        if (a.need_m0) acc_add_raw(acc[0], fr_mont_mul_raw(W, u));
        acc_add_raw(acc[1], fr_mont_mul_raw(W, d));
    }
    block_reduce_acc<GKR_LR_NSUM, 18, true>(acc, a.partials + (size_t)(row < GKR_LSPEC_CAND ? row : 0) * GKR_SPEC_SET_WORDS);
    spec_publish(a.partials, a.counter, a.host_out, (unsigned int*)(a.host_out + GKR_SPEC_FLAG_WORD), a.seq, gridDim.x * gridDim.y, &s_last,
                 GKR_LSPEC_CAND, GKR_LR_NSUM);
}
