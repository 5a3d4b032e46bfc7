// linear_round.hip.h -- the single-point sumcheck round of a LINEAR gate (power-1 descriptors of the gate table:
// identity xs[0], circuit/gates/copy.go:15-22; sums of up to four inputs plus Ark, e.g. the add gate of the GMiMC
// circuits), one launch per round: fold(r_{k-1}) of the
// layer's tables + the round's sums + hand-off to the host, replacing the reference's dispatchPartialEvals /
// dispatchFolding pair (sumcheck/prover.go:70-76,148-190) and its folded Eq table for these layers.
//
// Same construction as cipher_round.hip.h with a degree-1 gate.  With one evaluation point q the round message is
//     P_k(t) = c_k * eq(q_k, t) * S_k(t),   S_k(t) = sum_x W_k(x) * (u(x) + t*d(x)) = M_0 + M_1 t,
//     W_k = eq(q[k+1:], .),  u = gate(tables at x),  d = gate's linear part of (tables at x+mid) - (tables at x),
// so the device returns two sums (one when the round's claim gives c_k*M_0 = claim_k - q_k*c_k*M_1) and the host
// forms the three coefficients poly.InterpolateOnRange would produce from the reference's evaluations at t = 0, 1, 2.
// The kernel is bound by HBM: 4 (fold) or 2 elements read and 2 written per table and pair, two multiplications.
#pragma once
#include "cipher_round.hip.h"

#define GKR_LR_NSUM 2
#define GKR_LR_WORDS (GKR_LR_NSUM * GKR_ACC_WORDS)

struct LinearRoundArgs {
    CPlanes src[GKR_MAX_ARITY];   // FOLD: previous round's tables (4P elements); else this round's (2P)
    Planes dst[GKR_MAX_ARITY];    // FOLD: folded tables (2P elements)
    CPlanes wt, wj;        // eq weights: per lane (2^g entries), per iteration (P >> g entries, HAS_WJ only)
    size_t P;
    unsigned lg_threads;
    Fr r, r_lo;            // previous round's challenge and its image r * 2^-128 (fr_mul_const2_raw)
    Fr ark;                // added once per pair to u (zero for the identity gate)
    int arity;             // tables folded and handed over
    unsigned sum_mask;     // tables that enter the gate's sum (identity: 1; add: 3)
    unsigned long long* racc;
    unsigned int* counter;
    unsigned long long* host_out;   // GKR_LR_WORDS sums, then arity x (lo, hi) tail elements of 4 u64 in the last round
    unsigned int* host_flag;
    unsigned int seq;
    unsigned int need_m0;
    unsigned long long* tail_tables;   // host-mapped, or nullptr: this round's tables (2P entries per table, 4 u64 each)
    const unsigned long long* chal;    // pre-launched round: r, r_lo arrive through this host-mapped slot (cipher_round.hip.h)
    unsigned long long* chal_dev;
    unsigned int chal_seq;
    unsigned int chal_limit_s;     // see CipherRoundArgs
    unsigned int prio;             // wave priority (round_wave_priority): min(round index, 3)
};

template <bool FOLD, bool HAS_WJ>
__global__ void __launch_bounds__(GKR_BLOCK) k_linear_round(Batch<LinearRoundArgs> ba) {
    const LinearRoundArgs& a = ba.inst[blockIdx.z];
    __shared__ unsigned int s_last;
    round_wave_priority(a.prio);
    u32 T0[FR_WIDE_LIMBS], T1[FR_WIDE_LIMBS];      // wide sums of W*u and W*d: reduced once per lane
#pragma unroll
    for (int j = 0; j < FR_WIDE_LIMBS; j++) T0[j] = T1[j] = 0;
    const size_t P = a.P;
    const size_t threads = (size_t)1 << a.lg_threads;
    const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fr ch_r = a.r, ch_rlo = a.r_lo;
    if (FOLD && a.chal && !wait_challenge(a.chal, a.chal_dev, a.chal_seq, ch_r, ch_rlo, a.host_out + 104, a.chal_limit_s)) return;
    if (gtid < threads) {
        const Fr wt = ld_fr(a.wt.lo, a.wt.hi, gtid);
        const size_t iters = P >> a.lg_threads;
        for (size_t j = 0; j < iters; j++) {
            const size_t x = j * threads + gtid;
            Fr u = a.ark, d = fr_zero();
            for (int t = 0; t < a.arity; t++) {
                Fr lo, hi;
                if (FOLD) {
                    const Fr x0 = ld_fr(a.src[t].lo, a.src[t].hi, x), x2 = ld_fr(a.src[t].lo, a.src[t].hi, x + 2 * P);
                    const Fr x1 = ld_fr(a.src[t].lo, a.src[t].hi, x + P), x3 = ld_fr(a.src[t].lo, a.src[t].hi, x + 3 * P);
                    lo = fr_reduce_lt4q(fr_add_raw(x0, fr_mul_const2_raw(fr_sub(x2, x0), ch_rlo, ch_r)));   // poly/multilin.go:32-34
                    hi = fr_reduce_lt4q(fr_add_raw(x1, fr_mul_const2_raw(fr_sub(x3, x1), ch_rlo, ch_r)));
                    st_fr(a.dst[t].lo, a.dst[t].hi, x, lo);
                    st_fr(a.dst[t].lo, a.dst[t].hi, x + P, hi);
                } else {
                    lo = ld_fr(a.src[t].lo, a.src[t].hi, x);
                    hi = ld_fr(a.src[t].lo, a.src[t].hi, x + P);
                }
                if ((a.sum_mask >> t) & 1u) {
                    u = fr_add(u, lo);
                    d = fr_add(d, fr_sub(hi, lo));
                }
                if (a.tail_tables) {   // the host takes over after this round (GKRHIP_HOST_TAIL)
                    unsigned long long* tt = a.tail_tables + 4 * (size_t)t * 2 * P;
#pragma unroll
                    for (int l = 0; l < 4; l++) {
                        tt[4 * x + l] = (unsigned long long)lo.v[2 * l] | ((unsigned long long)lo.v[2 * l + 1] << 32);
                        tt[4 * (x + P) + l] = (unsigned long long)hi.v[2 * l] | ((unsigned long long)hi.v[2 * l + 1] << 32);
                    }
                }
                if (P == 1) {   // last round: the two remaining entries of each table go to the host (final fold there)
                    unsigned long long* tail = a.host_out + GKR_LR_WORDS + 8 * t;
#pragma unroll
                    for (int l = 0; l < 4; l++) {
                        tail[l] = (unsigned long long)lo.v[2 * l] | ((unsigned long long)lo.v[2 * l + 1] << 32);
                        tail[4 + l] = (unsigned long long)hi.v[2 * l] | ((unsigned long long)hi.v[2 * l + 1] << 32);
                    }
                }
            }
            Fr W = wt;
            if (HAS_WJ) W = fr_mont_mul_raw(ld_fr(a.wj.lo, a.wj.hi, j), wt);
            if (a.need_m0) fr_mac_wide(T0, W, u);
            fr_mac_wide(T1, W, d);
        }
    }
    Acc9 acc[GKR_LR_NSUM];
    fr_redc_wide(acc[0].w, T0);
    fr_redc_wide(acc[1].w, T1);
    block_reduce_acc<GKR_LR_NSUM, 18, true>(acc, a.racc);
    publish_sums(a.racc, a.counter, a.host_out, a.host_flag, a.seq, GKR_LR_WORDS, &s_last);
}
