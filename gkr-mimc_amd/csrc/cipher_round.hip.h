// cipher_round.hip.h -- the single-point cipher-gate sumcheck round (the 91 MiMC layers), one launch
// per round:  fold(r_{k-1}) of both tables + the round's sums + cross-block reduction + hand-off to
// the host, replacing the reference's dispatchPartialEvals / dispatchFolding pair
// (sumcheck/prover.go:70-76,148-190; sumcheck/algo.go:46-51,54-205).
//
// Same polynomial, cheaper evaluation.  With one evaluation point q, after k folds the reference's
// bookkeeping table is Eq_k(x) = c_k * eq(q[k:], x), c_k = prod_{i<k} eq(q_i, r_i), so the round
// message is
//     P_k(t) = c_k * eq(q_k, t) * S_k(t),     S_k(t) = sum_x W_k(x) * (u(x) + t*d(x))^7,
//     W_k = eq(q[k+1:], .),  u = K_lo + S_lo + ark,  d = (K_hi - K_lo) + (S_hi - S_lo)
// (K = key table, S = state table, pairs (x, x+mid), gate (K+S+ark)^7: circuit/gates/cipher.go:32-41).
// The device returns the 8 monomial sums M_j = sum_x W_k(x) u^(7-j) d^j, read off the 2 x 4 product table
// {W u^4, W d^4} x {u^3, u^2 d, u d^2, d^3} (18 multiplications per pair instead of the 45 of evaluating at
// t = 0..8; 17 when the round's claim P_k(0)+P_k(1) is known to the host, which then derives c_k*M_0 from it
// instead -- only inside gkr.Prove, where every claim is the previous round's P(r) and therefore consistent by
// construction); the host multiplies by the binomials, by the linear
// factor and by c_k, which yields exactly the coefficients poly.InterpolateOnRange produces from the
// reference's nine evaluations (a polynomial of degree <= 8 is determined by them), hence the same
// Fiat-Shamir challenges and the same transcript.  The Eq table is never materialised or folded:
// W_k factorises over index bits into a per-thread factor Wt (low bits) and a per-iteration factor
// Wj (high bits), both read from tiny "suffix pyramids" of eq tables.
#pragma once
#include "kernels.hip.h"

#define GKR_CR_NSUM 8                          // M_0 .. M_7
#define GKR_CR_WORDS (GKR_CR_NSUM * GKR_ACC_WORDS)

// ------------------------------------------------------------------------------------------------
// Proof groups (host_group.hip.h): the kernels of a layer's rounds take the arguments of up to GKR_GROUP_MAX proofs of the same
// shape that one host thread proves in lock-step -- blockIdx.z selects the proof, every workgroup works for exactly one.  The
// proofs share nothing (tables, accumulators, counters, hand-off words and flags are the proof's own); what they share is the
// launch: at 2^20 entries a proof is ~1 700 launches of mostly tiny kernels, and the GPU's dispatch of such launches from many
// queues -- not its arithmetic -- is what bounds many small proofs in flight (profiles/r04_bn20_plateau.txt).  One proof alone
// is a group of one (inst[0], grid.z = 1).  By value: the whole batch is kernel-argument memory (8 x 400 bytes at most; scalar
// loads at a uniform offset, no scratch).
// ------------------------------------------------------------------------------------------------
#define GKR_GROUP_MAX 8
template <class A>
struct Batch {
    static const int N = sizeof(A) * GKR_GROUP_MAX <= 4096 ? GKR_GROUP_MAX : (int)(4096 / sizeof(A));      // (4 KiB of kernel arguments: a group beyond N takes two launches)
    A inst[N];
};
// ------------------------------------------------------------------------------------------------
// suffix pyramid: level s (s = 0..max_level) is the table eq(q[nc-s .. nc-1], .) of 2^s entries,
// stored at element offset 2^s - 1.  Thread idx computes the running product over its low bits
// (LSB <-> q[nc-1]) and writes level s when idx < 2^s.  max_level muls per thread, no dependencies
// between threads (poly/eq.go:41-59 computes the same values by doubling).
// ------------------------------------------------------------------------------------------------
struct PyramidArgs {
    Planes out;
    Planes out2;      // optional (lo == nullptr: none): every entry times 2^-128, the second image fr_mul_const2_raw takes
    const Fr* q;      // nc coordinates
    int nc, max_level;
    Fr seed;          // level-0 value (1, or the multiplier)
};
__device__ __forceinline__ void eq_suffix_pyramid_body(const PyramidArgs& a, size_t idx, const Fr* __restrict__ q) {
    if (idx >= ((size_t)1 << a.max_level)) return;
    const Fr one = fr_one();
    const Fr two128 = {{0u, 0u, 0u, 0u, 1u, 0u, 0u, 0u}};   // the plain integer 2^128: a Montgomery product with it divides by 2^128
    Fr cur = a.seed;
    if (idx == 0) {
        st_fr(a.out.lo, a.out.hi, 0, cur);
        if (a.out2.lo) st_fr(a.out2.lo, a.out2.hi, 0, fr_mul(cur, two128));
    }
    for (int s = 1; s <= a.max_level; s++) {
        const Fr qc = q[a.nc - s];
        const bool bit = (idx >> (s - 1)) & 1;
        const Fr f = bit ? qc : fr_sub(one, qc);
        cur = fr_mul(cur, f);
        if (idx < ((size_t)1 << s)) {
            st_fr(a.out.lo, a.out.hi, (((size_t)1 << s) - 1) + idx, cur);
            if (a.out2.lo) st_fr(a.out2.lo, a.out2.hi, (((size_t)1 << s) - 1) + idx, fr_mul(cur, two128));
        }
    }
}
GKR_KERNEL void __launch_bounds__(GKR_BLOCK) k_eq_suffix_pyramid(PyramidArgs a) {
    eq_suffix_pyramid_body(a, (size_t)blockIdx.x * blockDim.x + threadIdx.x, a.q);
}
// up to four pyramids of a layer in one launch (blockIdx.y selects; unused slots have max_level < 0)
// The coordinates travel with the launch (qv, for the pyramids whose q is null): up to GKR_PYR_MAXQ of them -- every layer of up to
// 2^30 entries -- instead of a copy to device memory queued in front of the launch, which is a dispatch of its own (a blit kernel:
// two per layer, one in eight of a small proof's dispatches -- and the dispatches are what many small proofs in flight are bound by).
#define GKR_PYR_MAXQ 30
struct PyramidArgs3 {
    PyramidArgs p[4];
    Fr qv[GKR_PYR_MAXQ];
};
// The upper levels of a wide pyramid without the long chains: level s > lo_level of the pyramid over q[.. nc) is
//     level_s[idx] = level_lo[idx mod 2^lo_level] * H_(s - lo_level)[idx >> lo_level],
// H being the (small) pyramid over the next coordinates q[.. nc - lo_level).  k_eq_suffix_pyramids builds the levels up to
// lo_level and H with short chains on few threads; this kernel fills levels lo_level + 1 .. hi_level with ONE product per
// entry.  (Every lane of the one-kernel form walks the whole chain of hi_level products whether it stores or not: 2^17
// lanes x 17 products = 29 us on the critical path of every layer at bN = 24, against ~13 us for the two launches.)
struct PyramidExpandArgs {
    Planes out;          // the wide pyramid (levels <= lo_level already there)
    CPlanes h;           // H pyramid (levels 1 .. hi_level - lo_level)
    int lo_level, hi_level;
};
GKR_KERNEL void __launch_bounds__(GKR_BLOCK) k_eq_pyramid_expand(Batch<PyramidExpandArgs> ba) {
    const PyramidExpandArgs& a = ba.inst[blockIdx.z];
    // a few products per lane in front of a layer's first round: ahead of the other lanes' big rounds on a shared SIMD (same-box
    // A/B, profiles/r05_prio_pyramids.txt: bN = 20 x 24 lanes +0.7 %, bN = 24 x 5 +0.5 %, GMiMC bN = 22 x 12 +1.0 %)
    __builtin_amdgcn_s_setprio(3);
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ((size_t)1 << a.hi_level)) return;
    const size_t lo_mask = ((size_t)1 << a.lo_level) - 1;
    const Fr base = ld_fr(a.out.lo, a.out.hi, lo_mask + (idx & lo_mask));           // level lo_level sits at offset 2^lo_level - 1
    for (int s = a.lo_level + 1; s <= a.hi_level; s++) {
        if (idx >= ((size_t)1 << s)) continue;
        const int t = s - a.lo_level;
        const Fr hv = ld_fr(a.h.lo, a.h.hi, (((size_t)1 << t) - 1) + (idx >> a.lo_level));
        st_fr(a.out.lo, a.out.hi, (((size_t)1 << s) - 1) + idx, fr_mul(base, hv));
    }
}
GKR_KERNEL void __launch_bounds__(GKR_BLOCK) k_eq_suffix_pyramids(Batch<PyramidArgs3> ba) {
    __builtin_amdgcn_s_setprio(3);      // (as k_eq_pyramid_expand)
    const PyramidArgs3& a3 = ba.inst[blockIdx.z];
    const PyramidArgs& p = a3.p[blockIdx.y];
    if (p.max_level < 0) return;
    if (p.q) eq_suffix_pyramid_body(p, (size_t)blockIdx.x * blockDim.x + threadIdx.x, p.q);
    else eq_suffix_pyramid_body(p, (size_t)blockIdx.x * blockDim.x + threadIdx.x, a3.qv);
}

// ------------------------------------------------------------------------------------------------
// the round kernel
// ------------------------------------------------------------------------------------------------
struct CipherRoundArgs {
    CPlanes k_src, s_src;  // FOLD: previous round's tables (4P elements); else this round's (2P)
    Planes k_dst, s_dst;   // FOLD: folded tables (2P elements) are written here (may alias src)
    CPlanes wt;            // per-thread factor: 2^g entries (pointer already at the level)
    CPlanes wj;            // per-iteration factor: P >> g entries (HAS_WJ only)
    CPlanes wj2;           // the same entries times 2^-128 (k_cipher_round_wide with late lane weights)
    size_t P;              // index pairs this round
    unsigned lg_threads;   // log2(threads)
    Fr r;                  // previous round's challenge (FOLD)
    Fr r_lo;               // r * 2^-128 mod q: with r, the split images fr_mul_const2_raw takes
    Fr ark;
    unsigned long long* partials;   // [GKR_CR_WORDS] accumulator shared by the blocks (atomic adds); zero at launch, reset by the last block
    unsigned int* counter;          // arrival counter, zero at launch, reset by the last block
    unsigned long long* host_out;   // host-mapped: GKR_CR_WORDS sums, then 8 x u64 x 4 tail elements
    unsigned int* host_flag;        // host-mapped: set to `seq` when host_out is complete
    unsigned int seq;
    unsigned int need_m0;           // 0: M_0 is derived by the host from the running claim (2 products fewer)
    unsigned long long* tail_tables;   // host-mapped, or nullptr: this round's tables (2P entries of K, then 2P of S, 4 u64
                                       // each) for the host, which runs the remaining (tiny) rounds itself
    // pre-launched round (FOLD only): the kernel is queued BEFORE the host has hashed the previous round; r and r_lo
    // are then not launch arguments but arrive through a host-mapped slot the first lanes poll (see wait_challenge)
    const unsigned long long* chal;    // host-mapped challenge slot (GKR_CHAL_WORDS words), or nullptr: r, r_lo above are valid
    unsigned long long* chal_dev;      // device-memory mailbox of the same shape: workgroup 0 forwards the slot to the others
    unsigned int chal_seq;             // the slot is valid for this launch when the high halves of its words equal chal_seq
    unsigned int chal_limit_s;         // give up after this many seconds without the challenge (0: one second)
    // PRE (k_cipher_round_wide<false, ., true>): the q-independent products of round 0, computed ahead by k_cipher_pre
    CPlanes pre[6];                    // u^4, d^4, u^3, u^2 d, u d^2, d^3 at every pair
    // AHEAD (k_cipher_round_wide<false, true, ., true>): round 0 of the NEXT layer, queued before that layer's last ahead_t
    // coordinates exist -- see the comment above ahead_publish
    unsigned int ahead_t;
    unsigned int prio;                 // wave priority (round_wave_priority, kernels.hip.h): min(round index, 3)
};

// ------------------------------------------------------------------------------------------------
// Challenge hand-over to a pre-launched round kernel.  The host writes 16 words (limbs 0..7: r, 8..15: r * 2^-128), each
// as (seq << 32) | limb -- an aligned 8-byte word is read atomically over PCIe, so every polling lane sees a
// consistent (seq, limb) pair whatever order the host's stores arrive in.  seq = 0xFFFFFFFF: the host gave up
// (error path); a lane also gives up after ~1 s without an answer (the host then runs the layer's rounds once more without
// queueing anything ahead of its challenge: rounds_with_retry in host_sumcheck.hip.h) -- after limit_s seconds where the launch says
// so (the sharded rounds, which have no retry: 20 s).  Returns false when the launch must be abandoned
// (uniformly over the workgroup).  The sixteen limbs come back wave-uniform (SGPRs), like launch arguments.
// ------------------------------------------------------------------------------------------------
#define GKR_CHAL_WORDS 16
#define GKR_CHAL_ABORT 0xFFFFFFFFu
// Workgroup 0 polls host memory (sixteen 8-byte reads per poll over PCIe) and forwards the tagged words to a
// device-memory mailbox; the other workgroups poll the mailbox (L2 atomics, every ~0.5 us) and look at the host slot themselves
// only every 1024th time (~0.5 ms: with 512 workgroups waiting, every 64th time was 26 GB/s of PCIe reads and doubled the
// latency of the rounds of 2^17 / 2^18 pairs).  Hundreds of workgroups polling the host directly saturate the PCIe read path and delay the
// very hand-off the host is waiting for (measured: +12 ms per proof) -- but the mailbox must stay a SHORTCUT, never a
// dependency: workgroup 0 need not be resident (several pre-launched kernels, e.g. of several processes sharing the GPU,
// can each hold part of the machine while their workgroups 0 wait for a slot: measured, a deadlock until the time-out).
// Whoever reads the words from the host forwards them; the writes are idempotent.  ONE workgroup per launch, also of a 2-D grid:
// the speculative launches (nine rows of workgroups) first had one host-polling workgroup per row, and soaks with every path forced
// on for a dozen lanes lost proofs to kernels that did not see, within their time, a challenge the host had published before they
// started; with one polling workgroup per launch that is rare but still happens under that load (never with one proof at a
// time), and it now costs a second, not the proof: the host runs the layer again (rounds_with_retry).
__device__ __forceinline__ bool wait_challenge(const unsigned long long* slot, unsigned long long* mailbox, unsigned int seq, Fr& r,
                                               Fr& r_lo, unsigned long long* diag = nullptr, unsigned int limit_s = 0) {
    __shared__ u32 s_ch[GKR_CHAL_WORDS];
    int bad = 0;
    if (threadIdx.x < GKR_CHAL_WORDS) {
        const unsigned long long t0 = wall_clock64();          // 100 MHz
        const bool first = blockIdx.x == 0 && blockIdx.y == 0;      // (ONE workgroup of the launch, also of a 2-D grid)
        unsigned long long v;
        bool from_host = first;
        for (unsigned it = 0;; it++) {
            // uniform over the sixteen lanes: they count the same iterations until the first of them leaves
            from_host = first || (it & 1023u) == 1023u;
            v = from_host ? __hip_atomic_load(slot + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                          : __hip_atomic_load(mailbox + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u32 s = (u32)(v >> 32);
            if (s == seq) break;
            if (s == GKR_CHAL_ABORT || wall_clock64() - t0 > (unsigned long long)(limit_s ? limit_s : 1u) * 100000000ull) {
                if (s != GKR_CHAL_ABORT) {
                    // The time ran out -- by the wall clock, which also runs while a wave is switched out (more hardware queues than the
                    // chip has slots: the scheduler time-slices them) or stalled behind a slow read: the word this lane last saw may be
                    // older than the verdict.  One more look at the source before giving up.
                    v = __hip_atomic_load(slot + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if ((u32)(v >> 32) == seq) {
                        from_host = true;
                        break;
                    }
                }
                bad = 1;
                if (diag && threadIdx.x == 0) {          // why the launch was abandoned (the host's error message quotes it)
                    diag[0] = ((unsigned long long)blockIdx.x << 32) | (s == GKR_CHAL_ABORT ? 1u : 2u);
                    diag[1] = wall_clock64() - t0;
                    diag[2] = v;
                    diag[3] = seq;
                }
                v = (unsigned long long)GKR_CHAL_ABORT << 32;
                from_host = false;
                break;
            }
            if (first) __builtin_amdgcn_s_sleep(2);
            else __builtin_amdgcn_s_sleep(16);
        }
        if (from_host) __hip_atomic_store(mailbox + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_ch[threadIdx.x] = (u32)v;
    }
    if (__syncthreads_or(bad)) return false;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        r.v[j] = __builtin_amdgcn_readfirstlane(s_ch[j]);
        r_lo.v[j] = __builtin_amdgcn_readfirstlane(s_ch[8 + j]);
    }
    return true;
}

// a += x as an un-reduced 288-bit integer (x < 2^256).  One opaque carry chain: keeps hipcc from
// widening the nine accumulator words to 64-bit pairs and from sinking all eight accumulations to
// the end of the loop body (both of which spill).
__device__ __forceinline__ void acc_add_raw(Acc9& a, const Fr& x) {
    asm volatile(
        "v_add_co_u32 %0, vcc, %0, %9\n\t"
        "v_addc_co_u32 %1, vcc, %1, %10, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %2, %11, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %3, %12, vcc\n\t"
        "v_addc_co_u32 %4, vcc, %4, %13, vcc\n\t"
        "v_addc_co_u32 %5, vcc, %5, %14, vcc\n\t"
        "v_addc_co_u32 %6, vcc, %6, %15, vcc\n\t"
        "v_addc_co_u32 %7, vcc, %7, %16, vcc\n\t"
        "v_addc_co_u32 %8, vcc, 0, %8, vcc"
        : "+v"(a.w[0]), "+v"(a.w[1]), "+v"(a.w[2]), "+v"(a.w[3]), "+v"(a.w[4]), "+v"(a.w[5]), "+v"(a.w[6]),
          "+v"(a.w[7]), "+v"(a.w[8])
        : "v"(x.v[0]), "v"(x.v[1]), "v"(x.v[2]), "v"(x.v[3]), "v"(x.v[4]), "v"(x.v[5]), "v"(x.v[6]), "v"(x.v[7])
        : "vcc");
}

__device__ __forceinline__ void cipher_round_publish(const CipherRoundArgs& a, unsigned int* s_last) {
    publish_sums(a.partials, a.counter, a.host_out, a.host_flag, a.seq, GKR_CR_WORDS, s_last);
}

// sharded prover: after the all-reduce the summed words go to the host the same way (host-mapped buffer + flag)
GKR_KERNEL void __launch_bounds__(128) k_publish_words(const unsigned long long* __restrict__ src, unsigned long long* host_dst,
                                                       int n, unsigned int* host_flag, unsigned int seq) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) host_dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// LAT = latency variant for the small rounds (at most one wave per SIMD is resident, so nothing hides the
// ~10-cycle dependent-issue latency of a single multiplication chain): no scheduling barriers, so hipcc
// interleaves the independent products of the monomial schedule, and a 512-VGPR budget.
template <bool FOLD, bool HAS_WJ, bool LAT>
__device__ __forceinline__ void cipher_round_body(const CipherRoundArgs& a) {
    __shared__ unsigned int s_last;
    if (LAT) __builtin_amdgcn_s_setprio(3);      // a latency-bound round goes first when it shares a SIMD with the look-ahead kernel
    else round_wave_priority(a.prio);
    Acc9 acc[GKR_CR_NSUM];
#pragma unroll
    for (int t = 0; t < GKR_CR_NSUM; t++)
#pragma unroll
        for (int j = 0; j < GKR_ACC_WORDS; j++) acc[t].w[j] = 0;

    const size_t P = a.P;
    const size_t threads = (size_t)1 << a.lg_threads;
    const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fr ch_r = a.r, ch_rlo = a.r_lo;
    if (FOLD && a.chal && !wait_challenge(a.chal, a.chal_dev, a.chal_seq, ch_r, ch_rlo, a.host_out + 104, a.chal_limit_s)) return;
    if (gtid < threads) {
        const Fr wt = ld_fr(a.wt.lo, a.wt.hi, gtid);
        const Fr ark = a.ark;
        const size_t iters = P >> a.lg_threads;
        for (size_t j = 0; j < iters; j++) {
            const size_t x = j * threads + gtid;
            Fr klo, khi, slo, shi;
            if (FOLD) {
                // previous round's tables have 4P entries and pair (y, y+2P); this round pairs (x, x+P)
                const Fr r = ch_r;
                const Fr k0 = ld_fr(a.k_src.lo, a.k_src.hi, x), k2 = ld_fr(a.k_src.lo, a.k_src.hi, x + 2 * P);
                const Fr k1 = ld_fr(a.k_src.lo, a.k_src.hi, x + P), k3 = ld_fr(a.k_src.lo, a.k_src.hi, x + 3 * P);
                const Fr s0 = ld_fr(a.s_src.lo, a.s_src.hi, x), s2 = ld_fr(a.s_src.lo, a.s_src.hi, x + 2 * P);
                const Fr s1 = ld_fr(a.s_src.lo, a.s_src.hi, x + P), s3 = ld_fr(a.s_src.lo, a.s_src.hi, x + 3 * P);
                if (LAT) {
                    Fr f0, f1, f2, f3;
                    fr_mont_mul2_raw(f0, f1, fr_sub(k2, k0), r, fr_sub(k3, k1), r);
                    fr_mont_mul2_raw(f2, f3, fr_sub(s2, s0), r, fr_sub(s3, s1), r);
                    klo = fr_add(k0, fr_reduce_once(f0));
                    khi = fr_add(k1, fr_reduce_once(f1));
                    slo = fr_add(s0, fr_reduce_once(f2));
                    shi = fr_add(s1, fr_reduce_once(f3));
                } else {
                // poly/multilin.go:32-34; the challenge is a launch-wide constant: 96-product multiplication, < 3q
                const Fr ra = ch_rlo;
                klo = fr_reduce_lt4q(fr_add_raw(k0, fr_mul_const2_raw(fr_sub(k2, k0), ra, r)));
                khi = fr_reduce_lt4q(fr_add_raw(k1, fr_mul_const2_raw(fr_sub(k3, k1), ra, r)));
                slo = fr_reduce_lt4q(fr_add_raw(s0, fr_mul_const2_raw(fr_sub(s2, s0), ra, r)));
                shi = fr_reduce_lt4q(fr_add_raw(s1, fr_mul_const2_raw(fr_sub(s3, s1), ra, r)));
                }
                st_fr(a.k_dst.lo, a.k_dst.hi, x, klo);
                st_fr(a.k_dst.lo, a.k_dst.hi, x + P, khi);
                st_fr(a.s_dst.lo, a.s_dst.hi, x, slo);
                st_fr(a.s_dst.lo, a.s_dst.hi, x + P, shi);
            } else {
                klo = ld_fr(a.k_src.lo, a.k_src.hi, x);
                khi = ld_fr(a.k_src.lo, a.k_src.hi, x + P);
                slo = ld_fr(a.s_src.lo, a.s_src.hi, x);
                shi = ld_fr(a.s_src.lo, a.s_src.hi, x + P);
            }
            // lazy sums: u < 3q, d < 2q.  The product chain is closed below 3q (a*b/2^256 + q < 2.7q for a, b < 3q),
            // so neither is reduced
            const Fr u = fr_add_raw(fr_add_raw(klo, slo), ark);
            const Fr d = fr_add_raw(fr_sub(khi, klo), fr_sub(shi, slo));
            Fr W = wt;
            if (HAS_WJ) W = fr_mont_mul_raw(ld_fr(a.wj.lo, a.wj.hi, j), wt);
            // all products below are lazy Montgomery products (below 3q).  The scheduling barriers keep
            // hipcc from interleaving the independent products (which only raises register pressure:
            // the kernel is VALU-bound and each product already saturates the issue slot).
#define GKR_SB() do { if (!LAT) __builtin_amdgcn_sched_barrier(0); } while (0)
            // Monomials through a 2 x 4 product table: {W u^4, W d^4} x {u^3, u^2 d, u d^2, d^3} gives all eight
            // W u^(7-j) d^j with ten products in front of the final ones (u^2, d^2; the four cubics; u^4, d^4; W.).
            Fr p, r2, A, B, C, D, U4, D4, X0, X1, t, t2;
            if (LAT) {
                // latency variant: products issued in independent pairs with interleaved instruction streams
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
            } else {
            p = fr_mont_mul_raw(u, u);   GKR_SB();
            r2 = fr_mont_mul_raw(d, d);  GKR_SB();
            A = fr_mont_mul_raw(p, u);   GKR_SB();   // u^3
            B = fr_mont_mul_raw(p, d);   GKR_SB();   // u^2 d
            C = fr_mont_mul_raw(u, r2);  GKR_SB();   // u d^2
            D = fr_mont_mul_raw(r2, d);  GKR_SB();   // d^3
            U4 = fr_mont_mul_raw(p, p);  GKR_SB();   // u^4
            D4 = fr_mont_mul_raw(r2, r2); GKR_SB();  // d^4
            X0 = fr_mont_mul_raw(W, U4); GKR_SB();
            X1 = fr_mont_mul_raw(W, D4); GKR_SB();
            if (a.need_m0) { t = fr_mont_mul_raw(X0, A); GKR_SB(); acc_add_raw(acc[0], t); GKR_SB(); }  // W u^7
            t = fr_mont_mul_raw(X0, B); GKR_SB(); acc_add_raw(acc[1], t); GKR_SB();  // W u^6 d
            t = fr_mont_mul_raw(X0, C); GKR_SB(); acc_add_raw(acc[2], t); GKR_SB();  // W u^5 d^2
            t = fr_mont_mul_raw(X0, D); GKR_SB(); acc_add_raw(acc[3], t); GKR_SB();  // W u^4 d^3
            t = fr_mont_mul_raw(X1, A); GKR_SB(); acc_add_raw(acc[4], t); GKR_SB();  // W u^3 d^4
            t = fr_mont_mul_raw(X1, B); GKR_SB(); acc_add_raw(acc[5], t); GKR_SB();  // W u^2 d^5
            t = fr_mont_mul_raw(X1, C); GKR_SB(); acc_add_raw(acc[6], t); GKR_SB();  // W u d^6
            t = fr_mont_mul_raw(X1, D); GKR_SB(); acc_add_raw(acc[7], t); GKR_SB();  // W d^7
            }
#undef GKR_SB
            if (a.tail_tables) {
                unsigned long long* tt = a.tail_tables;
#pragma unroll
                for (int l = 0; l < 4; l++) {
                    tt[4 * x + l] = (unsigned long long)klo.v[2 * l] | ((unsigned long long)klo.v[2 * l + 1] << 32);
                    tt[4 * (x + P) + l] = (unsigned long long)khi.v[2 * l] | ((unsigned long long)khi.v[2 * l + 1] << 32);
                    tt[4 * (2 * P + x) + l] = (unsigned long long)slo.v[2 * l] | ((unsigned long long)slo.v[2 * l + 1] << 32);
                    tt[4 * (3 * P + x) + l] = (unsigned long long)shi.v[2 * l] | ((unsigned long long)shi.v[2 * l + 1] << 32);
                }
            }
            if (P == 1) {
                // last round: hand the two remaining entries of each table to the host, which applies
                // the final fold (two scalar multiplications) to obtain finalClaims (prover.go:79-86)
                unsigned long long* tail = a.host_out + GKR_CR_WORDS;
#pragma unroll
                for (int l = 0; l < 4; l++) {
                    tail[0 + l] = (unsigned long long)klo.v[2 * l] | ((unsigned long long)klo.v[2 * l + 1] << 32);
                    tail[4 + l] = (unsigned long long)khi.v[2 * l] | ((unsigned long long)khi.v[2 * l + 1] << 32);
                    tail[8 + l] = (unsigned long long)slo.v[2 * l] | ((unsigned long long)slo.v[2 * l + 1] << 32);
                    tail[12 + l] = (unsigned long long)shi.v[2 * l] | ((unsigned long long)shi.v[2 * l + 1] << 32);
                }
            }
        }
    }

    // ---- block reduction of the limb words (exact integer sums), one partial per block
    block_reduce_acc<GKR_CR_NSUM, 18, true>(acc, a.partials);
    cipher_round_publish(a, &s_last);
}

template <bool FOLD, bool HAS_WJ>
__global__ void __launch_bounds__(GKR_BLOCK, 2) k_cipher_round(Batch<CipherRoundArgs> ba) {
    cipher_round_body<FOLD, HAS_WJ, false>(ba.inst[blockIdx.z]);
}
template <bool FOLD, bool HAS_WJ>
__global__ void __launch_bounds__(GKR_BLOCK, 1) k_cipher_round_lat(Batch<CipherRoundArgs> ba) {
    cipher_round_body<FOLD, HAS_WJ, true>(ba.inst[blockIdx.z]);
}
// (Round 6 measured this round in at most 80 registers -- amdgpu_waves_per_eu(6, 6): 208 registers spilled, 468 B of scratch per lane --
// so that its waves start BESIDE two wide waves instead of waiting for a wide workgroup to retire: bN = 20, 72 in flight in groups of
// 3 83.5 / 83.7 -> 82.9 / 82.7 M hashes/s, on lanes 66.4 / 66.2 -> 64.4 / 64.0: profiles/r06_slim_small_rounds.txt, commit 34d6a6d.  Not kept.)

// ------------------------------------------------------------------------------------------------
// Round 0 AHEAD of its evaluation point.  Round 0 of a layer needs the layer's whole point q -- the challenges of the layer
// proven before it, the last of which the host draws at the very end of that layer (its host-tail rounds: 5 serial hashes,
// ~180 us during which the GPU has nothing to do).  But the weights factorise over index bits, the LAST coordinates
// belonging to the LOWEST bits of the pair index x = (x_hi, y), y = the low t bits:
//     M_j = sum_x eq(q[1:], x) m_j(x) = sum_y eq(q[m-t:], y) * S_j(y),     S_j(y) = sum_{x_hi} eq(q[1:m-t], x_hi) m_j(x_hi, y),
// and S_j needs only the coordinates known when the host tail STARTS.  So the previous layer queues this kernel at that point:
// the wide round-0 kernel with the lane weight over the known bits only (wt indexed by gtid >> t) and an epilogue that sums
// lanes of equal y = gtid mod 2^t instead of all lanes: 7 x 2^t class sums (t <= 6), reduced to canonical elements by the last
// workgroup and handed to the host, which contracts them with eq(q[m-t:], .) once the tail has produced those coordinates
// (7 * 2^t products) -- round 0's kernel, its hand-off and its pyramids leave the critical path of every layer.
// Accumulators: GKR_RACC_SLOTS stripes of (63 << t) words, word (j, w) of class y at ((j * 9 + w) << t) + y.
// ------------------------------------------------------------------------------------------------
#define GKR_AHEAD_TMAX 6
#define GKR_AHEAD_STRIPE (7 * GKR_ACC_WORDS << GKR_AHEAD_TMAX)                 // words per stripe
#define GKR_AHEAD_OUT_WORDS (7 * 4 << GKR_AHEAD_TMAX)                          // canonical elements to the host: S_j(y) at (j * 2^t + y) * 4
#define GKR_AHEAD_FLAG_WORD GKR_AHEAD_OUT_WORDS
#define GKR_AHEAD_BUF_WORDS (GKR_AHEAD_OUT_WORDS + 16)
__device__ __forceinline__ void ahead_publish(unsigned long long* racc, unsigned int* counter, unsigned long long* host_out,
                                              unsigned int* host_flag, unsigned int seq, unsigned int t, unsigned int* s_last) {
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
        const unsigned nval = 7u << t;
        for (unsigned v = threadIdx.x; v < nval; v += blockDim.x) {
            const unsigned j = v >> t, y = v & ((1u << t) - 1);
            unsigned long long w[GKR_ACC_WORDS];
#pragma unroll
            for (int k = 0; k < GKR_ACC_WORDS; k++) {
                unsigned long long* p = racc + (((size_t)j * GKR_ACC_WORDS + k) << t) + y;
                unsigned long long sum = 0;
#pragma unroll
                for (int sl = 0; sl < GKR_RACC_SLOTS; sl++) {
                    sum += __hip_atomic_load(p + (size_t)sl * GKR_AHEAD_STRIPE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    p[(size_t)sl * GKR_AHEAD_STRIPE] = 0;
                }
                w[k] = sum;
            }
            // value = sum_k w[k] 2^(32 k): carry into eight 32-bit limbs and a top part (as spec_publish, cipher_spec.hip.h)
            Fr lo;
            unsigned long long c = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                c += w[k] & 0xffffffffull;
                lo.v[k] = (u32)c;
                c = (c >> 32) + (w[k] >> 32);
            }
            c += w[8];
            Fr top = fr_zero();
            top.v[0] = (u32)c;
            top.v[1] = (u32)(c >> 32);
            const Fr r2 = {{0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u}};   // R^2 mod q
            Fr a0, a1;
            fr_mont_mul2_raw(a0, a1, lo, fr_one(), top, r2);
            const Fr x = fr_add(fr_reduce_once(a0), fr_reduce_once(a1));
#pragma unroll
            for (int l = 0; l < 4; l++) host_out[4 * (size_t)v + l] = (unsigned long long)x.v[2 * l] | ((unsigned long long)x.v[2 * l + 1] << 32);
        }
        // (the arrival counter is reset with the accumulators, in front of the ONE fence every writing lane needs anyway; the flag's
        // release store orders lane 0 behind the barrier: a second system-scope fence here cost every round ~1.5 us)
        if (threadIdx.x == 0) *counter = 0;
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------------------------------------
// Deferred-reduction variant for the large rounds (several pairs per lane, M_0 derived from the claim).
// The seven products W u^(7-j) d^j that only feed the sums M_1..M_7 are accumulated as plain 512-bit
// integers (fr_mac_wide: 79 limb products instead of the 136 of a Montgomery product) into 17-limb
// accumulators, and each lane reduces its seven sums once, after the loop (fr_redc_wide).  Four wide
// accumulators live in VGPRs; three live in LDS (17 words per lane, read and written back around their
// MAC), which keeps the kernel inside the 256-VGPR budget of two waves per SIMD.  The LDS region is
// reused by the block reduction afterwards.
// ------------------------------------------------------------------------------------------------
#ifndef GKR_WIDE_LDS
#define GKR_WIDE_LDS 3
#endif
#ifdef GKR_NO_SQR
#define GKR_SQR(x) fr_mont_mul_raw(x, x)
#else
#define GKR_SQR(x) fr_mont_sqr_raw(x)
#endif
// (the block reduction's transpose buffer AND its per-wave sums alias the accumulators, which are dead by then: 52 224 B per
// workgroup + the challenge words -- three workgroups fit a CU's 160 KiB; with the per-wave sums beside the union it was 54 856 B
// and the third missed by 728 B)
struct WideShared {
    union {
        struct {
            uint4 q[GKR_WIDE_LDS][4][GKR_BLOCK];
            u32 w[GKR_WIDE_LDS][GKR_BLOCK];
        } acc;
        struct {
            u32 tr[18][GKR_BLOCK + 1];
            unsigned long long red[GKR_BLOCK / 64][GKR_CR_WORDS];
        } rd;
    };
};

__device__ __forceinline__ void wide_lds_load(u32 (&T)[FR_WIDE_LIMBS], const WideShared& sh, int slot) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const uint4 x = sh.acc.q[slot][c][threadIdx.x];
        T[4 * c] = x.x; T[4 * c + 1] = x.y; T[4 * c + 2] = x.z; T[4 * c + 3] = x.w;
    }
    T[16] = sh.acc.w[slot][threadIdx.x];
}
__device__ __forceinline__ void wide_lds_store(WideShared& sh, int slot, const u32 (&T)[FR_WIDE_LIMBS]) {
#pragma unroll
    for (int c = 0; c < 4; c++) sh.acc.q[slot][c][threadIdx.x] = make_uint4(T[4 * c], T[4 * c + 1], T[4 * c + 2], T[4 * c + 3]);
    sh.acc.w[slot][threadIdx.x] = T[16];
}

// WT_LATE (rounds with many pairs per lane): the per-lane weight Wt is the same for every pair of a lane, so
// it multiplies the lane's seven reduced sums once after the loop instead of every pair's weight inside it.
// PRE (round 0 only, FOLD = false): u^4, d^4 and the four cubics were computed ahead of time by k_cipher_pre -- they do
// not depend on the layer's evaluation point -- so the launch on the critical path is two products by the launch-wide
// weight and the seven wide MACs, reading 192 bytes per pair instead of computing eight products.
// (Round 6 measured the same kernel built for three waves per SIMD -- __launch_bounds__(256, 3): 168 VGPRs, hipcc spills 84-101 of
// them, ~30 scratch instructions per pair inside the loop; with the LDS trimmed to 52.5 KB three workgroups fit a CU -- and it lost
// everywhere: bN = 20 x 24 / x 56 lanes -8.6 / -9.3 %, GMiMC bN = 22 x 12 -5.5 %, bN = 24 x 5 -6.4 %, one proof alone 2-6 % slower
// (profiles/r06_occupancy_ab.txt, commit 69ae821 has the variant and the switch).  Two waves, nothing spilled, stays.)
template <bool FOLD, bool WT_LATE, bool PRE = false, bool AHEAD = false>
__global__ void __launch_bounds__(GKR_BLOCK, 2) k_cipher_round_wide(Batch<CipherRoundArgs> ba) {
    const CipherRoundArgs& a = ba.inst[blockIdx.z];
    static_assert(!(FOLD && PRE), "the precomputed products exist for round 0 only");
    static_assert(!AHEAD || (WT_LATE && !FOLD), "round 0 ahead of its point: late lane weights, no fold");
    round_wave_priority(a.prio);
    __shared__ WideShared sh;
    __shared__ unsigned int s_last;
    u32 R[GKR_CR_NSUM - 1 - GKR_WIDE_LDS][FR_WIDE_LIMBS];     // M_4 .. M_7
#pragma unroll
    for (int t = 0; t < GKR_CR_NSUM - 1 - GKR_WIDE_LDS; t++)
#pragma unroll
        for (int j = 0; j < FR_WIDE_LIMBS; j++) R[t][j] = 0;
    {
        u32 Z[FR_WIDE_LIMBS];
#pragma unroll
        for (int j = 0; j < FR_WIDE_LIMBS; j++) Z[j] = 0;
#pragma unroll
        for (int s = 0; s < GKR_WIDE_LDS; s++) wide_lds_store(sh, s, Z);   // each lane touches only its own column
    }

    const size_t P = a.P;
    const size_t threads = (size_t)1 << a.lg_threads;
    const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fr ch_r = a.r, ch_rlo = a.r_lo;
    if (FOLD && a.chal && !wait_challenge(a.chal, a.chal_dev, a.chal_seq, ch_r, ch_rlo, a.host_out + 104, a.chal_limit_s)) return;
    if (gtid < threads) {
        const Fr wt = ld_fr(a.wt.lo, a.wt.hi, gtid);
        const Fr negark = fr_sub(fr_zero(), a.ark);      // q - ark (0 for ark = 0), canonical
        const size_t iters = P >> a.lg_threads;
        for (size_t j = 0; j < iters; j++) {
            const size_t x = j * threads + gtid;
            Fr klo, khi, slo, shi;
            Fr p, r2, A, B, C, D, U4, D4, X0, X1;
            if (PRE) {
                U4 = ld_fr_nt(a.pre[0].lo, a.pre[0].hi, x);
                D4 = ld_fr_nt(a.pre[1].lo, a.pre[1].hi, x);
                A = ld_fr_nt(a.pre[2].lo, a.pre[2].hi, x);
                B = ld_fr_nt(a.pre[3].lo, a.pre[3].hi, x);
                C = ld_fr_nt(a.pre[4].lo, a.pre[4].hi, x);
                D = ld_fr_nt(a.pre[5].lo, a.pre[5].hi, x);
            } else if (FOLD) {
                const Fr r = ch_r;
                const Fr k0 = ld_fr(a.k_src.lo, a.k_src.hi, x), k2 = ld_fr(a.k_src.lo, a.k_src.hi, x + 2 * P);
                const Fr k1 = ld_fr(a.k_src.lo, a.k_src.hi, x + P), k3 = ld_fr(a.k_src.lo, a.k_src.hi, x + 3 * P);
                const Fr s0 = ld_fr(a.s_src.lo, a.s_src.hi, x), s2 = ld_fr(a.s_src.lo, a.s_src.hi, x + 2 * P);
                const Fr s1 = ld_fr(a.s_src.lo, a.s_src.hi, x + P), s3 = ld_fr(a.s_src.lo, a.s_src.hi, x + 3 * P);
                // poly/multilin.go:32-34; the challenge is a launch-wide constant: 96-product multiplication, < 3q
                const Fr ra = ch_rlo;
                klo = fr_reduce_lt4q(fr_add_raw(k0, fr_mul_const2_raw(fr_sub(k2, k0), ra, r)));
                khi = fr_reduce_lt4q(fr_add_raw(k1, fr_mul_const2_raw(fr_sub(k3, k1), ra, r)));
                slo = fr_reduce_lt4q(fr_add_raw(s0, fr_mul_const2_raw(fr_sub(s2, s0), ra, r)));
                shi = fr_reduce_lt4q(fr_add_raw(s1, fr_mul_const2_raw(fr_sub(s3, s1), ra, r)));
                st_fr(a.k_dst.lo, a.k_dst.hi, x, klo);
                st_fr(a.k_dst.lo, a.k_dst.hi, x + P, khi);
                st_fr(a.s_dst.lo, a.s_dst.hi, x, slo);
                st_fr(a.s_dst.lo, a.s_dst.hi, x + P, shi);
            } else {
                klo = ld_fr(a.k_src.lo, a.k_src.hi, x);
                khi = ld_fr(a.k_src.lo, a.k_src.hi, x + P);
                slo = ld_fr(a.s_src.lo, a.s_src.hi, x);
                shi = ld_fr(a.s_src.lo, a.s_src.hi, x + P);
            }
            // lazy sums u, d < 2q: slo + ark is taken mod q (as slo - (q - ark)) so that the squarings' operands stay
            // below 2q (fr_mont_sqr_raw doubles them in place); products of operands < 2q stay below 2q
            Fr u, d;
            if (!PRE) {
                u = fr_add_raw(klo, fr_sub(slo, negark));
                d = fr_add_raw(fr_sub(khi, klo), fr_sub(shi, slo));
            }
            Fr W = ld_fr(a.wj.lo, a.wj.hi, j);            // the same element for every lane of the launch
            Fr W2 = W;
            if (WT_LATE) W2 = ld_fr(a.wj2.lo, a.wj2.hi, j);
            else W = fr_mont_mul_raw(W, wt);
#ifdef GKR_NO_SB
#define GKR_SB() do { } while (0)
#else
#define GKR_SB() __builtin_amdgcn_sched_barrier(0)
#endif
            u32 T[FR_WIDE_LIMBS];
            if (PRE) {
                X0 = WT_LATE ? fr_mul_const2_raw(U4, W2, W) : fr_mont_mul_raw(W, U4);  GKR_SB();
                X1 = WT_LATE ? fr_mul_const2_raw(D4, W2, W) : fr_mont_mul_raw(W, D4);  GKR_SB();
            } else {
            // ordered for short lifetimes: u^2 and its dependants first, then d^2 and its dependants
            p = GKR_SQR(u);       GKR_SB();
            U4 = GKR_SQR(p);      GKR_SB();   // u^4
            // with late lane weights W is uniform over the launch's lanes: 96-product multiplication (< 3q)
            X0 = WT_LATE ? fr_mul_const2_raw(U4, W2, W) : fr_mont_mul_raw(W, U4);  GKR_SB();
            A = fr_mont_mul_raw(p, u);    GKR_SB();   // u^3
            B = fr_mont_mul_raw(p, d);    GKR_SB();   // u^2 d
            r2 = GKR_SQR(d);      GKR_SB();
            D4 = GKR_SQR(r2);     GKR_SB();   // d^4
            X1 = WT_LATE ? fr_mul_const2_raw(D4, W2, W) : fr_mont_mul_raw(W, D4);  GKR_SB();
            C = fr_mont_mul_raw(u, r2);   GKR_SB();   // u d^2
            D = fr_mont_mul_raw(r2, d);   GKR_SB();   // d^3
            }
            // {W u^4, W d^4} x {u^3, u^2 d, u d^2, d^3}: the seven closing products are wide MACs; the LDS-resident
            // sums are fetched ahead of their MAC
            // sum M_j (j = 1..7) lives in LDS slot j-1 when j-1 < GKR_WIDE_LDS, else in R[j-1-GKR_WIDE_LDS]
#define GKR_MAC(j, X, Y)                                                                                  \
    do {                                                                                                  \
        if ((j) - 1 < GKR_WIDE_LDS) {                                                                     \
            wide_lds_load(T, sh, (j) - 1); fr_mac_wide(T, X, Y); GKR_SB(); wide_lds_store(sh, (j) - 1, T); GKR_SB(); \
        } else {                                                                                          \
            fr_mac_wide(R[(j) - 1 - GKR_WIDE_LDS < 0 ? 0 : (j) - 1 - GKR_WIDE_LDS], X, Y); GKR_SB();      \
        }                                                                                                 \
    } while (0)
            GKR_MAC(1, X0, B);   // W u^6 d
            GKR_MAC(2, X0, C);   // W u^5 d^2
            GKR_MAC(3, X0, D);   // W u^4 d^3
            GKR_MAC(4, X1, A);   // W u^3 d^4
            GKR_MAC(5, X1, B);   // W u^2 d^5
            GKR_MAC(6, X1, C);   // W u d^6
            GKR_MAC(7, X1, D);   // W d^7
#undef GKR_MAC
#undef GKR_SB
        }
    }

    // ---- one Montgomery reduction per sum and lane, then the usual exact block reduction
    Acc9 acc[GKR_CR_NSUM];
#pragma unroll
    for (int j = 0; j < GKR_ACC_WORDS; j++) acc[0].w[j] = 0;      // M_0: derived by the host from the claim
    // s = lo + top*2^256 (9 limbs) times Wt, as a Montgomery product: s*Wt/2^256 = mont(lo, Wt) + top*Wt
    Fr wtl = fr_zero();
    if (WT_LATE && gtid < threads) wtl = ld_fr(a.wt.lo, a.wt.hi, AHEAD ? (gtid >> a.ahead_t) : gtid);
    auto finish = [&](Acc9& dst, const u32 (&T)[FR_WIDE_LIMBS]) {
        fr_redc_wide(dst.w, T);
        if (WT_LATE) {
            Fr lo;
#pragma unroll
            for (int j = 0; j < 8; j++) lo.v[j] = dst.w[j];
            const u32 top = dst.w[8];
            const Fr m = fr_mont_mul_raw(lo, wtl);                  // < 2q
            u64 c = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                c += (u64)top * wtl.v[j] + m.v[j];
                dst.w[j] = (u32)c;
                c >>= 32;
            }
            dst.w[8] = (u32)c;
        }
    };
#pragma unroll
    for (int s = 0; s < GKR_WIDE_LDS; s++) {
        u32 T[FR_WIDE_LIMBS];
        wide_lds_load(T, sh, s);
        finish(acc[1 + s], T);
    }
#pragma unroll
    for (int t = 0; t < GKR_CR_NSUM - 1 - GKR_WIDE_LDS; t++) finish(acc[1 + GKR_WIDE_LDS + t], R[t]);
    if (AHEAD) {
        // class sums: lanes of equal y = gtid mod 2^t.  Inside a wave the lanes of a class are 2^t apart: a butterfly over the
        // upper lane bits leaves every lane with its class's sum, lanes below 2^t add it to the launch's accumulators (t >= 6:
        // the 64 lanes of a wave are 64 different classes)
        const unsigned t = a.ahead_t, lane = threadIdx.x & 63u;
        const unsigned y = (unsigned)gtid & ((1u << t) - 1);
        unsigned long long* dst = a.partials + (size_t)(blockIdx.x % GKR_RACC_SLOTS) * GKR_AHEAD_STRIPE + y;
        const bool live = gtid < threads;
#pragma unroll
        for (int j = 1; j < GKR_CR_NSUM; j++)
#pragma unroll
            for (int w = 0; w < GKR_ACC_WORDS; w++) {
                unsigned long long v = live ? (unsigned long long)acc[j].w[w] : 0ull;
                for (unsigned off = 32; off >= (1u << t) && off > 0; off >>= 1) v += __shfl_xor(v, (int)off);
                if ((t >= 6 || lane < (1u << t)) && v)
                    (void)__hip_atomic_fetch_add(dst + (((size_t)(j - 1) * GKR_ACC_WORDS + w) << t), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        ahead_publish(a.partials, a.counter, a.host_out, a.host_flag, a.seq, t, &s_last);
        return;
    }
    __syncthreads();                                                // tr aliases the LDS accumulators
    block_reduce_acc_buf<GKR_CR_NSUM, 18, true>(acc, a.partials, sh.rd.tr, sh.rd.red);
    cipher_round_publish(a, &s_last);
}

// ------------------------------------------------------------------------------------------------
// k_cipher_pre: the part of a cipher layer's round 0 that does not depend on the layer's evaluation point.  u = K_lo + S_lo
// + ark and d = (K_hi - K_lo) + (S_hi - S_lo) are functions of the witness tables alone, and so are u^4, d^4 and the four
// cubics -- eight of the ten products in front of the closing MACs.  gkr.Prove knows the NEXT layer's tables while the
// current layer's latency-bound small rounds leave the GPU idle; this kernel fills that idle time (second stream) and the
// next layer's round 0 on the critical path becomes k_cipher_round_wide<false, ., true>.  Same integers as the fused
// kernel computes (same schedules, same lazy bounds), so the sums -- and the transcript -- are identical.
// One pair per lane, many short workgroups: the latency-bound round kernels of the proof find free slots at any time.
// ------------------------------------------------------------------------------------------------
struct CipherPreArgs {
    CPlanes k_src, s_src;   // 2P entries each
    Planes out[6];          // u^4, d^4, u^3, u^2 d, u d^2, d^3 (P entries each)
    size_t P;
    Fr ark;
};
// The launch asks for GKR_PRE_LDS bytes of (unused) dynamic LDS so that ONE of its workgroups fits a CU, beside one workgroup
// of a round kernel (up to 243 VGPRs, 54.5 KB of LDS): without a cap its small waves refill every slot a finished wave frees
// and the round kernels of the proof starve until the look-ahead is done (measured: the whole gain lost).  The kernel's stream
// has normal priority (host_sumcheck.hip.h: a lowest-priority stream left incomplete products under load); one proof alone at
// bN = 24 with 60 KB (two workgroups per CU) / 100 KB: 279.5 / 272.4 ms (lowest priority and 60 KB, as it was: 275.1).
#define GKR_PRE_LDS (100 * 1024)
GKR_KERNEL void __launch_bounds__(GKR_BLOCK, 2) k_cipher_pre(CipherPreArgs a) {
    const size_t P = a.P;
    const Fr negark = fr_sub(fr_zero(), a.ark);
    for (size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x; x < P; x += (size_t)gridDim.x * blockDim.x) {
        const Fr klo = ld_fr(a.k_src.lo, a.k_src.hi, x), khi = ld_fr(a.k_src.lo, a.k_src.hi, x + P);
        const Fr slo = ld_fr(a.s_src.lo, a.s_src.hi, x), shi = ld_fr(a.s_src.lo, a.s_src.hi, x + P);
        const Fr u = fr_add_raw(klo, fr_sub(slo, negark));
        const Fr d = fr_add_raw(fr_sub(khi, klo), fr_sub(shi, slo));
        const Fr p = GKR_SQR(u);
        st_fr_nt(a.out[0].lo, a.out[0].hi, x, GKR_SQR(p));              // u^4
        st_fr_nt(a.out[2].lo, a.out[2].hi, x, fr_mont_mul_raw(p, u));   // u^3
        st_fr_nt(a.out[3].lo, a.out[3].hi, x, fr_mont_mul_raw(p, d));   // u^2 d
        const Fr r2 = GKR_SQR(d);
        st_fr_nt(a.out[1].lo, a.out[1].hi, x, GKR_SQR(r2));             // d^4
        st_fr_nt(a.out[4].lo, a.out[4].hi, x, fr_mont_mul_raw(u, r2));  // u d^2
        st_fr_nt(a.out[5].lo, a.out[5].hi, x, fr_mont_mul_raw(r2, d));  // d^3
    }
}
