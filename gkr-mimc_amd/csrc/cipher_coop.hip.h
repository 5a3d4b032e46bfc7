// cipher_coop.hip.h -- the single-point cipher-gate sumcheck round for SMALL rounds, eight lanes per index pair.
//
// A round with at most a few thousand pairs cannot fill the GPU: k_cipher_round_lat runs it with one pair per lane, so
// each SIMD holds a lone wave that issues one instruction every ~6 cycles whatever its dependencies, and the ~27
// field products of a pair (4 folds, 10 monomial products, 8 closing products, the weight) become ~17 us of
// dependent issue -- on the critical path of gkr.Prove ~1000 times per proof (sumcheck/prover.go:70-76 is serial in
// the rounds).  Here the products of ONE pair are dealt to EIGHT lanes, level by level:
//
//     level F (FOLD):  K_lo, K_hi, S_lo, S_hi                       = lo + r * (hi - lo)        4 products (roles 0..3)
//     level A:         u^2, d^2, u d, W u, W d                                                 5 products (roles 0..4)
//     level B:         u^4, d^4, W u^3, W u^2 d, W u d^2, W d^3                                6 products (roles 0..5)
//     level C:         {u^4, d^4} x {W u^3, W u^2 d, W u d^2, W d^3} = W u^(7-j) d^j, j=0..7   8 products (roles 0..7)
//
// with u = K_lo + S_lo + ark, d = (K_hi - K_lo) + (S_hi - S_lo), W = eq(q[k+1:], x) as in cipher_round.hip.h.  The
// operands of a level are the results of earlier levels, exchanged through LDS; every lane runs the same product code
// on its own operands, so there is no divergence inside a wave.  Four dependent products instead of twenty-seven.
// The terms are other REPRESENTATIVES of the same residues than the one-lane kernels produce (a different association
// of the same monomials), and the sums are exact integer sums reduced once on the host -- the same field elements, the
// same transcript (parity-tested against the oracle with the kernel forced on and off).
//
// Workgroup = 256 lanes = 32 pairs x 8 roles, role-major (lane = role * 32 + pair): the lanes of a role read
// consecutive table entries, and waves 2 and 3 (roles 4..7) skip the levels that have no work for them.
#pragma once
#include "cipher_round.hip.h"

#define GKR_COOP_PAIRS 32
static_assert(GKR_BLOCK == 8 * GKR_COOP_PAIRS, "the cooperative round kernel is written for 256-lane workgroups");

struct CoopShared {
    // value slots: 0 K_lo, 1 K_hi, 2 S_lo, 3 S_hi | 4 u^2, 5 d^2, 6 ud, 7 Wu, 8 Wd | 9 u^4, 10 d^4, 11 Wu^3, 12 Wu^2 d, 13 Wu d^2, 14 Wd^3
    u32 v[15][8][GKR_COOP_PAIRS + 1];
    u32 red[GKR_CR_NSUM][GKR_ACC_WORDS][GKR_COOP_PAIRS + 1];
};
__device__ __forceinline__ Fr coop_ld(const CoopShared& sh, int slot, int pl) {
    Fr r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.v[j] = sh.v[slot][j][pl];
    return r;
}
__device__ __forceinline__ void coop_st(CoopShared& sh, int slot, int pl, const Fr& x) {
#pragma unroll
    for (int j = 0; j < 8; j++) sh.v[slot][j][pl] = x.v[j];
}
__device__ __forceinline__ void coop_export(unsigned long long* dst, const Fr& x) {
#pragma unroll
    for (int l = 0; l < 4; l++) dst[l] = (unsigned long long)x.v[2 * l] | ((unsigned long long)x.v[2 * l + 1] << 32);
}

template <bool FOLD>
__global__ void __launch_bounds__(GKR_BLOCK, 1) k_cipher_round_coop(CipherRoundArgs a) {
    __shared__ CoopShared sh;
    __shared__ unsigned int s_last;
    __builtin_amdgcn_s_setprio(3);
    const int role = threadIdx.x / GKR_COOP_PAIRS, pl = threadIdx.x % GKR_COOP_PAIRS;
    const size_t P = a.P;
    // The table entries and the weight do not depend on the challenge: a pre-launched kernel requests those of its first
    // iteration BEFORE it waits for r, so the loads' latency overlaps the host's hash
    const size_t x0 = (size_t)blockIdx.x * GKR_COOP_PAIRS + pl;
    Fr pf_lo = fr_zero(), pf_hi = fr_zero(), pf_W = fr_zero();
    {
        const CPlanes src = role < 2 ? a.k_src : a.s_src;
        const size_t off = (role & 1) ? P : 0;                 // roles 1, 3: the high half of the pair
        if (x0 < P) {
            if (role < 4) {
                pf_lo = ld_fr(src.lo, src.hi, x0 + off);
                if (FOLD) pf_hi = ld_fr(src.lo, src.hi, x0 + off + 2 * P);
            }
            pf_W = ld_fr(a.wt.lo, a.wt.hi, x0);
        }
    }
    Fr ch_r = a.r, ch_rlo = a.r_lo;
    if (FOLD && a.chal && !wait_challenge(a.chal, a.chal_dev, a.chal_seq, ch_r, ch_rlo, a.host_out + 104, a.chal_limit_s)) return;
    Acc9 acc;
#pragma unroll
    for (int j = 0; j < GKR_ACC_WORDS; j++) acc.w[j] = 0;

    for (size_t base = (size_t)blockIdx.x * GKR_COOP_PAIRS; base < P; base += (size_t)gridDim.x * GKR_COOP_PAIRS) {
        const size_t x = base + pl;
        const bool live = x < P;
        const bool first_it = base == (size_t)blockIdx.x * GKR_COOP_PAIRS;
        // ---- level F: the four table entries of the pair, one role each (poly/multilin.go:32-34)
        if (role < 4) {
            Fr v = fr_zero();
            if (live) {
                const CPlanes src = role < 2 ? a.k_src : a.s_src;
                const size_t off = (role & 1) ? P : 0;
                if (FOLD) {
                    const Fr lo = first_it ? pf_lo : ld_fr(src.lo, src.hi, x + off);
                    const Fr hi = first_it ? pf_hi : ld_fr(src.lo, src.hi, x + off + 2 * P);
                    v = fr_reduce_lt4q(fr_add_raw(lo, fr_mul_const2_raw(fr_sub(hi, lo), ch_rlo, ch_r)));
                    const Planes dst = role < 2 ? a.k_dst : a.s_dst;
                    st_fr(dst.lo, dst.hi, x + off, v);
                } else {
                    v = first_it ? pf_lo : ld_fr(src.lo, src.hi, x + off);
                }
                if (a.tail_tables) coop_export(a.tail_tables + 4 * ((size_t)role * P + x), v);      // K: [0, 2P), S: [2P, 4P)
                if (P == 1) coop_export(a.host_out + GKR_CR_WORDS + 4 * role, v);
            }
            coop_st(sh, role, pl, v);
        }
        __syncthreads();
        // ---- u, d, W in every lane
        Fr u, d, W;
        {
            const Fr klo = coop_ld(sh, 0, pl), khi = coop_ld(sh, 1, pl), slo = coop_ld(sh, 2, pl), shi = coop_ld(sh, 3, pl);
            u = fr_add_raw(fr_add_raw(klo, slo), a.ark);                     // < 3q
            d = fr_add_raw(fr_sub(khi, klo), fr_sub(shi, slo));              // < 2q
            W = first_it ? pf_W : (live ? ld_fr(a.wt.lo, a.wt.hi, x) : fr_zero());
        }
        // ---- level A: u^2, d^2, ud, Wu, Wd
        if (role < 5) {
            Fr p, q2;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                p.v[j] = role == 1 ? d.v[j] : (role >= 3 ? W.v[j] : u.v[j]);
                q2.v[j] = (role == 0 || role == 3) ? u.v[j] : d.v[j];
            }
            coop_st(sh, 4 + role, pl, fr_mont_mul_raw(p, q2));
        }
        __syncthreads();
        // ---- level B: u^4 = u^2 u^2, d^4 = d^2 d^2, Wu^3 = Wu u^2, Wu^2 d = Wu ud, Wu d^2 = Wu d^2, Wd^3 = Wd d^2
        if (role < 6) {
            const int sa = role == 0 ? 4 : role == 1 ? 5 : role == 5 ? 8 : 7;
            const int sb = role == 0 ? 4 : role == 3 ? 6 : role == 2 ? 4 : 5;
            coop_st(sh, 9 + role, pl, fr_mont_mul_raw(coop_ld(sh, sa, pl), coop_ld(sh, sb, pl)));
        }
        __syncthreads();
        // ---- level C: role j holds the term of M_j = sum_x W u^(7-j) d^j
        {
            const Fr t = fr_mont_mul_raw(coop_ld(sh, 9 + (role >> 2), pl), coop_ld(sh, 11 + (role & 3), pl));
            if (live && (role != 0 || a.need_m0)) acc_add(acc, t);
        }
        __syncthreads();          // the slots are rewritten by the next iteration
    }

    // ---- sums over the pairs of the workgroup (exact integer sums), one atomic add per word into the launch-wide
    // accumulator, then the hand-off of the fused round kernels
#pragma unroll
    for (int j = 0; j < GKR_ACC_WORDS; j++) sh.red[role][j][pl] = acc.w[j];
    __syncthreads();
    if (threadIdx.x < GKR_CR_WORDS) {
        const int rr = threadIdx.x / GKR_ACC_WORDS, w = threadIdx.x % GKR_ACC_WORDS;
        unsigned long long s = 0;
#pragma unroll 8
        for (int i = 0; i < GKR_COOP_PAIRS; i++) s += sh.red[rr][w][(i + threadIdx.x) % GKR_COOP_PAIRS];
        if (s) (void)__hip_atomic_fetch_add(a.partials + (blockIdx.x % GKR_RACC_SLOTS) * GKR_RACC_STRIDE + threadIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    publish_sums(a.partials, a.counter, a.host_out, a.host_flag, a.seq, GKR_CR_WORDS, &s_last);
}
