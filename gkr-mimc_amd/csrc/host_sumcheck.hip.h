// host_sumcheck.hip.h -- sumcheck.Prove on device tables (sumcheck/prover.go:46-144): the fused single-point
// cipher rounds, the reference-shaped generic rounds, Evaluate.  Included by gkrhip.hip inside its anonymous namespace.
#pragma once
// ---- single-point cipher sumcheck: one fused launch per round (cipher_round.hip.h) -------------------
template <bool FOLD, bool HAS_WJ>
int launch_cipher_round(const CipherRoundArgs& a, int grid, bool lat, bool alone) {
    // a latency-bound launch of at most one workgroup per CU asks for enough (unused) dynamic LDS that two of its
    // workgroups cannot share a CU: the dispatcher otherwise packs some CUs with two lone-wave workgroups and leaves others idle
    // (only for a proof that is alone on the GPU: beside other proofs' kernels the extra LDS would keep the launch waiting)
    const size_t spread = (alone && grid <= cx().n_cu) ? (size_t)64 * 1024 : 0;
    if (lat) GKR_LAUNCH_BATCH((k_cipher_round_lat<FOLD, HAS_WJ>), dim3(grid), dim3(GKR_BLOCK), spread, cx().stream, a);
    else GKR_LAUNCH_BATCH((k_cipher_round<FOLD, HAS_WJ>), dim3(grid), dim3(GKR_BLOCK), 0, cx().stream, a);
    return 0;
}

// ---- hand-off of a fused round's sums ---------------------------------------------------------------------------
// Un-sharded, or sharded over the host shared-memory transport: the kernel writes its sums and tail words into the
// lane's host-mapped buffer and raises the lane's flag; with shared memory the ranks then add the sums on the host
// (576 bytes per round: no extra kernel has to queue behind the big rounds).  Sharded over RCCL: the sums stay on
// the device for ncclAllReduce, the kernel's own completion flag lands in a spare word of the exchange buffer, and
// the reduced words reach the host through coll_publish.
struct RoundTargets {
    unsigned long long* out;
    unsigned int* flag;
    bool on_device;
};
inline RoundTargets round_targets(bool collective) {
    RoundTargets t;
    t.on_device = collective && cx().lc.comm;
    t.out = t.on_device ? cx().lc.d_buf : cx().d_round;
    t.flag = t.on_device ? (unsigned int*)(cx().lc.d_buf + 192) : cx().d_flag;
    return t;
}
// Wait for round `seq`; on return *sums points at nsum words summed over the ranks (scratch: `summed`, nsum + 1 words) and
// cx().h_round + nsum holds this rank's ntail tail words.
int round_collect(bool collective, const RoundTargets& t, unsigned int seq, int nsum, int ntail, unsigned long long* summed,
                  const unsigned long long** sums) {
    *sums = cx().h_round;
    if (t.on_device) {
        CHK(coll_allreduce(cx().lc.d_buf, nsum));     // exact integer sum of limb-split lanes; the tail words are rank-local
        CHK(coll_publish(nsum + ntail, seq));
        const int rcd = wait_flag(seq, nullptr, coll_timeout_ms(), coll_stream());
        if (g_corrupt_collect && rcd == 0) cx().h_round[GKR_ACC_WORDS] ^= 1ull;      // (after the device-side exchange: set the hook on every rank)
        g_corrupt_collect = false;
        return rcd;
    }
    const int rc = wait_flag(seq);
    if (g_corrupt_collect && rc == 0) cx().h_round[GKR_ACC_WORDS] ^= 1ull;           // test_corrupt_sum: this rank's M_1, before the exchange
    g_corrupt_collect = false;
    if (collective && host_exchange()) {
        // The ranks exchange one word more than the sums: a vote.  A rank whose round kernel gave up waiting for its challenge
        // (recoverable: g_chal_timeout) still takes part in the exchange, with zero sums and vote 1; any vote makes EVERY rank
        // leave the layer's rounds at this very exchange with g_chal_timeout set, and rounds_with_retry runs them again in safe
        // mode on all ranks together (the exchanges of the retry line up: every rank restarts the layer).
        if (rc != 0 && !g_chal_timeout) return rc;
        if (rc != 0) memset(summed, 0, sizeof(unsigned long long) * nsum);
        else memcpy(summed, cx().h_round, sizeof(unsigned long long) * nsum);
        summed[nsum] = rc != 0 ? 1ull : 0ull;
        if (cx().lc.tick_lane >= 0) CHK(tick_allreduce(summed, nsum + 1));      // RCCL through the ticker: one communicator for all lanes
        else CHK(shm_allreduce_host(summed, nsum + 1));
        if (summed[nsum]) {
            g_chal_timeout = true;
            return rc != 0 ? rc : fail("the round kernel of %llu peer rank(s) gave up waiting for its challenge: the layer's rounds are run again", summed[nsum]);
        }
        *sums = summed;
        return 0;
    }
    return rc;
}

inline E fold2(const E& lo, const E& hi, const E& r) { return hfr::add(lo, hfr::mul(hfr::sub(hi, lo), r)); }

// ---- the last rounds of a single-point cipher sumcheck on the host ----------------------------------------------
// A round of P <= 2^6 pairs is a launch-latency-bound kernel (a lone wave: ~25 us plus ~8 us of launch and hand-off)
// for a few microseconds of arithmetic, so from GKRHIP_HOST_TAIL pairs on the host runs the rounds itself on the two
// tables the last device round exported (cipher_round.hip.h, tail_tables): same monomial sums, same coefficients,
// same transcript.  K and S hold 2^mm entries (table convention: round binds the top index bit); q[0:mm] are the
// remaining coordinates; seed multiplies every eq weight.  On return K[0], S[0] are the fully folded values.
void host_cipher_rounds(const E& ark, int mm, std::vector<E>& K, std::vector<E>& S, const E* q, const E& seed, E& c, E* proof,
                        E* chal, E* claim, bool* claim_known) {
    static const hfr::u64 binom7[8] = {1, 7, 21, 35, 35, 21, 7, 1};
    for (int k = 0; k < mm; k++) {
        const size_t P = (size_t)1 << (mm - 1 - k);
        // W(x) = seed * eq(q[k+1 : mm], bits(x)), x < P, by the reference's doubling (poly/eq.go:41-59)
        std::vector<E> W(P);
        W[0] = seed;
        for (int i = 0; i < mm - 1 - k; i++) {
            const E& qi = q[k + 1 + i];
            for (size_t t = 0; t < ((size_t)1 << i); t++) {
                const size_t J = t << (mm - 1 - k - i), JN = J + ((size_t)1 << (mm - 2 - k - i));
                W[JN] = hfr::mul(qi, W[J]);
                W[J] = hfr::sub(W[J], W[JN]);
            }
        }
        const bool derive_m0 = claim && *claim_known;
        E M[8];
        for (int j = 0; j < 8; j++) M[j] = hfr::ZERO;
        for (size_t x = 0; x < P; x++) {
            const E u = hfr::add(hfr::add(K[x], S[x]), ark);
            const E d = hfr::add(hfr::sub(K[x + P], K[x]), hfr::sub(S[x + P], S[x]));
            const E u2 = hfr::mul(u, u), d2 = hfr::mul(d, d);
            const E cub[4] = {hfr::mul(u2, u), hfr::mul(u2, d), hfr::mul(u, d2), hfr::mul(d2, d)};
            const E x0 = hfr::mul(W[x], hfr::mul(u2, u2)), x1 = hfr::mul(W[x], hfr::mul(d2, d2));
            for (int j = derive_m0 ? 1 : 0; j < 4; j++) M[j] = hfr::add(M[j], hfr::mul(x0, cub[j]));
            for (int j = 0; j < 4; j++) M[4 + j] = hfr::add(M[4 + j], hfr::mul(x1, cub[j]));
        }
        E csp[8];
        for (int j = derive_m0 ? 1 : 0; j < 8; j++) csp[j] = hfr::mul(c, hfr::mul(M[j], hfr::from_u64(binom7[j])));
        if (derive_m0) {
            E rest = csp[1];
            for (int j = 2; j < 8; j++) rest = hfr::add(rest, csp[j]);
            csp[0] = hfr::sub(*claim, hfr::mul(q[k], rest));
        }
        const E a0 = hfr::sub(hfr::ONE, q[k]);
        const E a1 = hfr::sub(hfr::add(q[k], q[k]), hfr::ONE);
        E* co = proof + (size_t)k * 9;
        co[0] = hfr::mul(a0, csp[0]);
        for (int j = 1; j < 8; j++) co[j] = hfr::add(hfr::mul(a0, csp[j]), hfr::mul(a1, csp[j - 1]));
        co[8] = hfr::mul(a1, csp[7]);
        const double t_h0 = now_ms();
        const E r = hfr::mimc_hash(co, 9);
        cx().prof.host_hash_ms += now_ms() - t_h0;
        chal[k] = r;
        c = hfr::mul(c, hfr::eval_eq(&q[k], &r, 1));
        if (claim) {
            *claim = hfr::eval_univariate(co, 9, r);
            *claim_known = true;
        }
        for (size_t x = 0; x < P; x++) {   // poly/multilin.go:26-36
            K[x] = fold2(K[x], K[x + P], r);
            S[x] = fold2(S[x], S[x + P], r);
        }
        K.resize(P);
        S.resize(P);
        cx().prof.rounds++;
    }
}


// ---- host side of the challenge hand-over to a pre-launched round kernel (wait_challenge, cipher_round.hip.h) -----
inline void chal_publish(unsigned int seq, const E& r, const E& r_lo, int slot = 0) {
    volatile unsigned long long* w = cx().h_chal + (size_t)slot * GKR_CHAL_WORDS;
    const unsigned long long tag = (unsigned long long)seq << 32;
    for (int i = 0; i < 4; i++) {
        w[2 * i] = tag | (unsigned long long)(uint32_t)r.l[i];
        w[2 * i + 1] = tag | (unsigned long long)(uint32_t)(r.l[i] >> 32);
        w[8 + 2 * i] = tag | (unsigned long long)(uint32_t)r_lo.l[i];
        w[8 + 2 * i + 1] = tag | (unsigned long long)(uint32_t)(r_lo.l[i] >> 32);
    }
    __sync_synchronize();
    cx().dbg_pub_seq[slot] = seq;
    cx().dbg_pub_ms[slot] = now_ms();
}
// A pre-launched kernel that will never get its challenge (error return between the launch and the hash) is told
// to leave; the lane's stream is drained so that nothing of the failed call is still running when the caller returns.
struct ChalGuard {
    bool armed = false;
    ~ChalGuard() {
        if (!armed) return;
        volatile unsigned long long* w = cx().h_chal;
        for (int i = 0; i < GKR_CHAL_WORDS * kChalSlots; i++) w[i] = (unsigned long long)GKR_CHAL_ABORT << 32;
        __sync_synchronize();
        (void)hipStreamSynchronize(cx().stream);
        // the abort tags must not outlive the launch they were meant for: the next pre-launched kernel of this lane polls
        // the slot (and the device mailbox the tags were forwarded to) before any new challenge is published
        for (int i = 0; i < GKR_CHAL_WORDS * kChalSlots; i++) w[i] = 0;
        __sync_synchronize();
        (void)hipMemsetAsync(cx().d_chal_dev, 0, sizeof(unsigned long long) * GKR_CHAL_WORDS * kChalSlots, cx().stream);
        (void)hipStreamSynchronize(cx().stream);
    }
};

// Fault injection for the tests, armed only through gkrhip_set_option("test_fail_after_prelaunch" / "test_drop_challenge", k)
// -- never from the environment: a stray variable must not be able to cost a proof.  Each fires once.
//   test_fail_after_prelaunch = k: an error return in round k while a pre-launched kernel is waiting for its challenge;
//   test_drop_challenge = k: the challenge of round k is NOT published -- the kernel waiting for it runs out of time, the round
//   loop fails with g_chal_timeout and rounds_with_retry runs the layer again in safe mode.
//   test_corrupt_sum = k: one bit of a sum of round k (the monomial sum M_1 / the evaluation at t = 1) is flipped on its way
//   from the device to the host -- BEFORE the ranks add their words when the exchange is host-side --, and
//   test_corrupt_tail = 1: one bit of the table entries the last device round of a fused loop hands to the host.  Either
//   stands for any slip of the device side (a race, an incomplete look-ahead, a bad fold): the sumcheck no longer closes,
//   sumcheck_prove_dev notices and runs the layer again (counter "layer_check_failures").
//   test_corrupt_times = n (set after test_corrupt_sum): the flip fires n times instead of once -- twice makes the retry fail too;
//   test_corrupt_skip = j: the first j round loops that reach round k are left alone (j selects the layer of a proof).
std::atomic<int> g_test_fail_round{-1}, g_test_drop_round{-1}, g_test_corrupt_round{-1}, g_test_corrupt_tail{-1}, g_test_corrupt_left{1},
    g_test_corrupt_skip{0};
inline bool test_fire(std::atomic<int>& hook, int k) {
    int want = k;
    return hook.load(std::memory_order_relaxed) == k && hook.compare_exchange_strong(want, -1);
}
inline bool test_fire_corrupt(int k) {
    if (g_test_corrupt_round.load(std::memory_order_relaxed) != k) return false;
    if (g_test_corrupt_skip.load(std::memory_order_relaxed) > 0 && g_test_corrupt_skip.fetch_sub(1) > 0) return false;
    if (g_test_corrupt_left.fetch_sub(1) <= 1) return test_fire(g_test_corrupt_round, k);      // the last time disarms
    return true;
}

// The slow parts of the look-ahead -- the second stream (created on first use: a lane that never looks ahead holds one
// hardware queue, not two) and the six scratch tables (a miss in the arena is a hipMalloc behind the arena's lock) -- are done
// at the START of the round loop, where no kernel of the lane is waiting for the host: a pre-launched or speculative kernel
// gives up after a second without its challenge (20 s on the device-side exchange of a sharded proof), and nothing that can block for an unbounded time (another lane holding the arena's
// lock across its own hipMalloc) belongs between a pre-launch and the publication of its challenge.
int pre_prepare() {
    if (!cx().req_K || !cx().req_S || cx().req_m < 2) return 0;
    const size_t P = (size_t)1 << (cx().req_m - 1);
    if (!cx().aux) {
        // NORMAL priority.  The stream used to have the lowest priority (the look-ahead kernel must not delay a round kernel; its
        // LDS request caps it at one workgroup per CU anyway) -- and with a dozen lanes forced to use it 4 % of the proofs were wrong
        // (57 of 1 440 at bN = 18; 0 of 1 440 at normal priority).  Round 5 reproduced that WITHOUT the library
        // (tools/prio_event_probe.hip, profiles/r05_anomalies.md): with 16 hardware queues, hipStreamWaitEvent on an event recorded in
        // a lowest-priority stream sometimes lets the waiting stream go while one to three XCDs have not finished their share of the
        // producer (16 of 122 036 iterations; 0 of 331 149 at normal priority).  Runtime behaviour, not ours to fix; and whatever still
        // slips through is caught by sumcheck_closes below.  One proof alone: 277.6 ms against 275.1 at bN = 24.
        HIPCHK(hipStreamCreateWithFlags(&cx().aux, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&cx().pre_done, hipEventDisableTiming));
    }
    for (auto& t : cx().pre_t)
        if (t.cap != P) {
            if (t.base) table_release(&t);
            CHK(table_alloc(&t, P));
        }
    return 0;
}
// Queue k_cipher_pre for the layer gkr.Prove announced (cx().req_*) on the lane's look-ahead stream.
int launch_pre() {
    CHK(pre_prepare());                            // (a no-op after the call at the start of the round loop)
    const DevTable* K = cx().req_K;
    const DevTable* S = cx().req_S;
    cx().req_K = cx().req_S = nullptr;
    if (!K || !S || cx().req_m < 2) return 0;
    const size_t P = (size_t)1 << (cx().req_m - 1);
    CipherPreArgs a;
    memset(&a, 0, sizeof a);
    a.k_src = K->cplanes();
    a.s_src = S->cplanes();
    for (int i = 0; i < 6; i++) a.out[i] = cx().pre_t[i].planes();
    a.P = P;
    a.ark = to_dev(cx().req_ark);
    // (GKR_PRE_LDS bytes of unused dynamic LDS cap the look-ahead kernel at one workgroup per CU: it must not crowd out the round kernels)
    {
        static std::once_flag once;       // more than 64 KiB of dynamic LDS needs the attribute
        std::call_once(once, [] { (void)hipFuncSetAttribute((const void*)k_cipher_pre, hipFuncAttributeMaxDynamicSharedMemorySize, GKR_PRE_LDS); });
    }
    hipLaunchKernelGGL(k_cipher_pre, dim3(grid_for(P, 1 << 20)), dim3(GKR_BLOCK), (size_t)GKR_PRE_LDS, cx().aux, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(cx().pre_done, cx().aux));
    cx().pre_K = K->base;
    cx().pre_S = S->base;
    cx().pre_ark = cx().req_ark;
    cx().pre_m = cx().req_m;
    return 0;
}

// A round loop that failed because a waiting kernel's time ran out (g_chal_timeout; ChalGuard drained the stream) is run ONCE
// more in safe mode -- no kernel queued ahead of its challenge, no look-ahead.  The invariant the retry rests on: the round
// loops READ the layer's tables and WRITE only scratch tables, the lane's accumulators and its hand-off buffers.  Workgroups
// have clocks of their own, so some of an abandoned launch may have folded, stored into scratch and added into d_racc /
// d_counter while others left: the scratch tables are rewritten from the layer's tables by the retry's own launches, and the
// accumulators are re-zeroed at the start of the retried loop (racc_dirty is still set from the failed run).  With the running
// values (c, claim) restored the retry therefore produces the same transcript.  Sharded over a host-side exchange (shared
// memory, ticker) the ranks agree on the retry through the vote word of round_collect; the device-side RCCL exchange of a
// single lane has no vote: its kernels wait 20 s and a miss fails the proof on every rank.
template <class F>
int rounds_with_retry(E& c, E& claim, bool& claim_known, F&& run) {
    const E c0 = c, claim0 = claim;
    const bool known0 = claim_known;
    g_chal_timeout = false;
    int rc = run();
    if (rc != 0 && g_chal_timeout && !g_safe_mode && (!shard_view().gamma || host_exchange())) {      // (sharded: the ranks agree on the retry through the vote word of round_collect)
        g_chal_timeout = false;
        c = c0;
        claim = claim0;
        claim_known = known0;
        g_safe_mode = true;
        g_cnt_retries.fetch_add(1, std::memory_order_relaxed);
        rc = run();
        g_safe_mode = false;
    }
    return rc;
}

// ---- speculative small rounds (cipher_spec.hip.h) ------------------------------------------------------------------
// buffers of the lane, allocated the first time a proof takes the path
int spec_ensure() {
    if (cx().h_spec) return 0;
    HIPCHK(hipHostMalloc(&cx().h_spec, sizeof(unsigned long long) * 2 * GKR_SPEC_BUF_WORDS, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer((void**)&cx().d_spec, cx().h_spec, 0));
    memset(cx().h_spec, 0, sizeof(unsigned long long) * 2 * GKR_SPEC_BUF_WORDS);
    HIPCHK(hipMalloc(&cx().d_spec_racc, sizeof(unsigned long long) * GKR_SPEC_CAND * GKR_SPEC_SET_WORDS));
    HIPCHK(hipMemsetAsync(cx().d_spec_racc, 0, sizeof(unsigned long long) * GKR_SPEC_CAND * GKR_SPEC_SET_WORDS, cx().stream));      // stream-ordered: see lane_alloc
    for (int i = 0; i < GKR_SPEC_CAND; i++) cx().spec_pts[i] = hfr::from_u64((hfr::u64)i);
    for (int i = 0; i < GKR_SPEC_CAND; i++) {      // 1 / prod_{j != i} (i - j)
        E den = hfr::ONE;
        for (int j = 0; j < GKR_SPEC_CAND; j++)
            if (j != i) den = hfr::mul(den, hfr::sub(cx().spec_pts[i], cx().spec_pts[j]));
        cx().spec_invden[i] = hfr::pow_q_minus_2(den);
    }
    return 0;
}
// M_j(r), j = j0..7, from the candidates' sums (cand[i * 8 + j] = M_j(i), i = 0..7, canonical elements: the kernel's last
// workgroup reduces them): M_j has degree 7 in r, so the Lagrange basis on the points 0..7 reproduces it exactly
void spec_interpolate(const E* cand, const E& r, int j0, E* M) {
    E d[GKR_SPEC_CAND], pre[GKR_SPEC_CAND], suf[GKR_SPEC_CAND], L[GKR_SPEC_CAND];
    for (int i = 0; i < GKR_SPEC_CAND; i++) d[i] = hfr::sub(r, cx().spec_pts[i]);
    pre[0] = hfr::ONE;
    for (int i = 1; i < GKR_SPEC_CAND; i++) pre[i] = hfr::mul(pre[i - 1], d[i - 1]);
    suf[GKR_SPEC_CAND - 1] = hfr::ONE;
    for (int i = GKR_SPEC_CAND - 2; i >= 0; i--) suf[i] = hfr::mul(suf[i + 1], d[i + 1]);
    for (int i = 0; i < GKR_SPEC_CAND; i++) L[i] = hfr::mul(hfr::mul(pre[i], suf[i]), cx().spec_invden[i]);
    for (int j = j0; j < GKR_CR_NSUM; j++) {
        E acc = hfr::ZERO;
        for (int i = 0; i < GKR_SPEC_CAND; i++)
            acc = hfr::add(acc, hfr::mul(L[i], cand[i * GKR_CR_NSUM + j]));
        M[j] = acc;
    }
}

// ---- round 0 ahead of its point (cipher_round.hip.h, ahead_publish) -------------------------------------------------
// buffers of the lane, allocated the first time a proof takes the path
int ahead_ensure() {
    if (cx().h_ahead) return 0;
    HIPCHK(hipHostMalloc(&cx().h_ahead, sizeof(unsigned long long) * GKR_AHEAD_BUF_WORDS, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer((void**)&cx().d_ahead, cx().h_ahead, 0));
    memset(cx().h_ahead, 0, sizeof(unsigned long long) * GKR_AHEAD_BUF_WORDS);
    HIPCHK(hipMalloc(&cx().d_ahead_racc, sizeof(unsigned long long) * GKR_RACC_SLOTS * GKR_AHEAD_STRIPE));
    HIPCHK(hipMemsetAsync(cx().d_ahead_racc, 0, sizeof(unsigned long long) * GKR_RACC_SLOTS * GKR_AHEAD_STRIPE, cx().stream));      // stream-ordered: see lane_alloc
    HIPCHK(hipMalloc(&cx().d_ahead_counter, 64));
    HIPCHK(hipMemsetAsync(cx().d_ahead_counter, 0, 64, cx().stream));
    return 0;
}
inline volatile unsigned int* ahead_flag() { return (volatile unsigned int*)(cx().h_ahead + GKR_AHEAD_FLAG_WORD); }
// eq(q[0:n], .) over 2^n entries, q[0] <-> the most significant index bit (poly/eq.go:41-59)
inline void eq_table_host(const E* q, int n, std::vector<E>& W) {
    W.assign((size_t)1 << n, hfr::ZERO);
    W[0] = hfr::ONE;
    for (int i = 0; i < n; i++)
        for (size_t t = 0; t < ((size_t)1 << i); t++) {
            const size_t J = t << (n - i), JN = J + ((size_t)1 << (n - 1 - i));
            W[JN] = hfr::mul(q[i], W[J]);
            W[J] = hfr::sub(W[J], W[JN]);
        }
}
// M_j = sum_y eq(q_low, y) S_j(y), j = 1..7 (out[0..6]): the class sums S (S_j(y) at (j - 1) * 2^t + y) contracted with the t
// coordinates that did not exist when they were computed
inline void ahead_contract(const E* S, const E* q_low, int t, E* out) {
    std::vector<E> Wy;
    eq_table_host(q_low, t, Wy);
    for (int j = 0; j < 7; j++) {
        E acc = hfr::ZERO;
        for (size_t y = 0; y < Wy.size(); y++) acc = hfr::add(acc, hfr::mul(Wy[y], S[((size_t)j << t) + y]));
        out[j] = acc;
    }
}
// Queue round 0 of the layer gkr.Prove proves next (cx().nxt_*), whose point is THIS layer's challenges: chal[0 .. k_known]
// exist, the last t = m - 1 - k_known are still to come (this layer's host tail).  Called where the host tail starts; the
// stream is idle from here to the end of the layer.  The products of k_cipher_pre are used when they exist for that layer.
// the coordinates of a layer's pyramids: with the launch when they fit (PyramidArgs3::qv), else staged to cx().d_q; returns what the
// pyramids' q must be (nullptr: qv)
int pyramid_coords(PyramidArgs3& pa3, const E* coords, size_t n, const Fr** qsrc) {
    if (n <= (size_t)GKR_PYR_MAXQ) {
        for (size_t i = 0; i < n; i++) pa3.qv[i] = to_dev(coords[i]);
        *qsrc = nullptr;
        return 0;
    }
    CHK(stage_coords(coords, n));
    *qsrc = cx().d_q;
    return 0;
}
int ahead_launch(int m, const E* chal, int k_known, bool solo) {
    const DevTable* K = cx().nxt_K;
    const DevTable* S = cx().nxt_S;
    const int t = m - 1 - k_known;
    if (!K || !S || cx().req_m != m || t < 1 || t > GKR_AHEAD_TMAX || !cx().wide_mode) return 0;
    const int g_m = round_threads_log2_max(m);
    const int g = std::min(solo ? std::min(g_m + 1, 17) : g_m, m - 2);      // threads = 2^g, at least two pairs per lane
    if (g < t || g < 8) return 0;
    const int lj = m - 1 - g;
    CHK(ahead_ensure());
    for (auto tc : {std::make_pair(&cx().ahead_pyrU, (size_t)2 << lj), std::make_pair(&cx().ahead_pyrU2, (size_t)2 << lj),
                    std::make_pair(&cx().ahead_pyrTh, (size_t)2 << (g - t))})
        if (tc.first->cap != tc.second) {
            if (tc.first->base && cx().ahead_in_flight) {      // an earlier class-sum kernel nobody waited for may still read it
                HIPCHK(hipStreamSynchronize(cx().stream));
                cx().ahead_in_flight = false;
            }
            if (tc.first->base) table_release(tc.first);
            CHK(table_alloc(tc.first, tc.second));
        }
    PyramidArgs3 pa3;
    memset(&pa3, 0, sizeof pa3);
    const Fr* qsrc = nullptr;
    CHK(pyramid_coords(pa3, chal, (size_t)(m - t), &qsrc));
    if (g_arena_check.load(std::memory_order_relaxed)) {      // table_release: everything in front of this point must be done when this layer's scratch goes back
        if (!cx().chk_fence) HIPCHK(hipEventCreateWithFlags(&cx().chk_fence, hipEventDisableTiming));
        HIPCHK(hipEventRecord(cx().chk_fence, cx().stream));
    }
    for (int v = 0; v < 4; v++) pa3.p[v].max_level = -1;
    pa3.p[0].out = cx().ahead_pyrTh.planes();      // lane weight over the known low bits: level g - t = eq(q[m-g .. m-t-1], gtid >> t)
    pa3.p[0].out2 = Planes{nullptr, nullptr};
    pa3.p[0].q = qsrc;
    pa3.p[0].nc = m - t;
    pa3.p[0].max_level = g - t;
    pa3.p[0].seed = to_dev(hfr::ONE);
    pa3.p[1].out = cx().ahead_pyrU.planes();       // iteration weight: level lj = eq(q[1 .. m-g-1], j)
    pa3.p[1].out2 = cx().ahead_pyrU2.planes();
    pa3.p[1].q = qsrc;
    pa3.p[1].nc = m - g;
    pa3.p[1].max_level = lj;
    pa3.p[1].seed = to_dev(hfr::ONE);
    GKR_LAUNCH_BATCH(k_eq_suffix_pyramids, dim3(grid_for((size_t)1 << std::max(g - t, lj), 1 << 20), 2), dim3(GKR_BLOCK), 0, cx().stream, pa3);
    HIPCHK(hipGetLastError());
    CipherRoundArgs a;
    memset(&a, 0, sizeof a);
    a.k_src = K->cplanes();
    a.s_src = S->cplanes();
    const size_t offT = ((size_t)1 << (g - t)) - 1, offU = ((size_t)1 << lj) - 1;
    a.wt = CPlanes{cx().ahead_pyrTh.base + offT, cx().ahead_pyrTh.base + cx().ahead_pyrTh.cap + offT};
    a.wj = CPlanes{cx().ahead_pyrU.base + offU, cx().ahead_pyrU.base + cx().ahead_pyrU.cap + offU};
    a.wj2 = CPlanes{cx().ahead_pyrU2.base + offU, cx().ahead_pyrU2.base + cx().ahead_pyrU2.cap + offU};
    a.P = (size_t)1 << (m - 1);
    a.lg_threads = (unsigned)g;
    a.ark = to_dev(cx().req_ark);
    a.partials = cx().d_ahead_racc;
    a.counter = cx().d_ahead_counter;
    a.host_out = cx().d_ahead;
    a.host_flag = (unsigned int*)(cx().d_ahead + GKR_AHEAD_FLAG_WORD);
    a.seq = cx().ahead_seq = ++cx().seq;
    a.need_m0 = 0;
    a.ahead_t = (unsigned)t;
    const bool pre = cx().pre_K && cx().pre_K == K->base && cx().pre_S == S->base && cx().pre_m == m && cx().pre_ark == cx().req_ark;
    const int grid = (int)(((size_t)1 << g) / GKR_BLOCK);
    if (pre) {
        HIPCHK(hipStreamWaitEvent(cx().stream, cx().pre_done, 0));
        for (int i = 0; i < 6; i++) a.pre[i] = cx().pre_t[i].cplanes();
        cx().pre_K = cx().pre_S = nullptr;             // consumed
        GKR_LAUNCH_BATCH((k_cipher_round_wide<false, true, true, true>), dim3(grid), dim3(GKR_BLOCK), 0, cx().stream, a);
        g_cnt_lookahead.fetch_add(1, std::memory_order_relaxed);
    } else {
        GKR_LAUNCH_BATCH((k_cipher_round_wide<false, true, false, true>), dim3(grid), dim3(GKR_BLOCK), 0, cx().stream, a);
    }
    HIPCHK(hipGetLastError());
    cx().ahead_in_flight = true;
    cx().ahead_K = K->base;
    cx().ahead_S = S->base;
    cx().ahead_ark = cx().req_ark;
    cx().ahead_m = m;
    cx().ahead_t = t;
    cx().ahead_q.assign(chal, chal + (m - t));
    cx().nxt_K = cx().nxt_S = nullptr;
    return 0;
}

// ---- the skeleton the two fused round loops share -------------------------------------------------------------------
// cipher_rounds and linear_rounds differ in their kernels, in how a round's sums become coefficients and in what the host
// tail computes; everything else -- which rounds run on the device, which kernel is queued ahead of its challenge, the
// speculative schedule, the exchange, Fiat-Shamir, the claim, the publication of the challenge -- is this code.
struct InFlight {
    RoundTargets tg;
    unsigned int seq;
    bool derive_m0;
};
struct RoundPlan {
    int m = 0;
    int h_tail = 0;            // the host finishes the rounds with at most 2^h_tail pairs (0: every round on the device)
    int k_export = -1;         // the round whose tables go to the host
    int m_dev = 0;             // rounds on the device
    int k_s = -1;              // first speculative round (-1: none); the speculative rounds are k_s .. k_export
    bool collective = false, alone = false;
    bool pl_on = false;        // pre-launched rounds: the next round's kernel is queued before this round is hashed and polls the challenge slot
    bool pre_on = false;       // look-ahead of the next layer's round 0
    bool sh_tail = false;      // sharded host tail: gather the exported tables, every rank finishes ALL remaining rounds on the host
    bool is_spec(int k) const { return k_s >= 0 && k >= k_s && k <= k_export; }
};
// `tables`: tables the layer folds (what a sharded host tail gathers is tables * 2^(h+1) entries per rank);
// `h_unsharded`: depth of the host tail of an un-sharded layer (GKRHIP_HOST_TAIL; the sharded local rounds exchange
// device-produced words, so there the tail exists only as the gather of round 3: one gather instead of h + 1 exchanged rounds)
inline RoundPlan plan_rounds(int m, bool collective, int gamma_tail, bool* did_gamma, int tables, int h_unsharded) {
    RoundPlan p;
    p.m = m;
    p.collective = collective;
    p.alone = proofs_in_flight_now() <= 1;
    const int hs = cx().host_tail_sharded;
    p.sh_tail = collective && gamma_tail > 0 && did_gamma && hs > 0 && m >= hs + 2 && (cx().lc.comm || cx().lc.shm || cx().lc.tick_lane >= 0) &&
                sharded_tail_pays(tables * (2 << hs), hs);
    const int h_want = collective ? (p.sh_tail ? hs : 0) : h_unsharded;
    p.h_tail = (h_want > 0 && m >= h_want + 2) ? std::min(h_want, kHostTailMax) : 0;
    if (did_gamma) *did_gamma = false;
    p.k_export = p.h_tail ? m - 2 - p.h_tail : -1;
    p.m_dev = p.h_tail ? p.k_export + 1 : m;
    p.pl_on = !g_safe_mode && (cx().prelaunch >= 2 || (cx().prelaunch == 1 && p.alone));
    p.pre_on = !g_safe_mode && (cx().pre_mode >= 2 || (cx().pre_mode == 1 && p.alone));
    // Inside a proof group (host_group.hip.h) nothing is queued that polls for the host: the group's proofs take turns on one thread, a
    // kernel waiting for proof A's challenge would hold the stream while B and C hash (forced on, twelve lanes of bN = 18 in groups of
    // 3 missed 91 challenges in 720 proofs -- each a second of waiting and a layer run again; profiles/r06_soaks_groups.txt)
    if (t_group) p.pl_on = p.pre_on = false;
    // with round 0 running ahead of its point during the host tail (ahead_launch) the look-ahead products only pay from 2^22 entries
    // on: below, the whole round 0 fits the tail (bN = 18 / 20 / 21 alone: 75.9 / 92.9 / 107.7 ms without against 77.0 / 93.8 / 108.9 with
    // the products; bN = 22: 140.0 against 136.4)
    // ... and up to 2^24: at 2^25 the look-ahead kernel (2.7 ms at one workgroup per CU) no longer fits the small rounds it overlaps
    // (bN = 25 alone: 488 ms with, 479 without; bN = 22 / 23 / 24: 133 / 182 / 266 with against 138 / 192 / 291 without: profiles/r06_pre_ab.txt)
    if (cx().pre_mode == 1 && !collective && (m < 22 || m > 24) && p.h_tail > 0 && (cx().ahead_mode >= 2 || (cx().ahead_mode == 1 && p.alone))) p.pre_on = false;
    return p;
}
// Speculative rounds (cipher_spec.hip.h): rounds k_s .. k_export run for the candidate values of r_{k-1} while the host hashes
// round k-1.  They need the pre-launched rounds (the launches they ride behind) and the host tail (their export), and start at
// the first round of at most 2^spec_lg pairs that runs one pair per lane.
// (spec == 1: only while the look-ahead kernel of the next layer is small -- at 2^24 entries it needs the whole idle time of the
// small rounds: bN = 24 measured 279.5 -> 281..285 ms with speculation, bN = 23 200.4 -> 198.3, bN = 22 155 -> 150, bN = 20 109 -> 104)
template <class F>
void plan_speculation(RoundPlan& p, bool export_fits, F&& one_pair_per_lane) {
    if (!((cx().spec >= 2 || (cx().spec == 1 && p.alone && p.m <= cx().spec_max_m)) && !p.collective && p.pl_on && p.h_tail > 0 && export_fits)) return;
    for (int k = 2; k <= p.k_export && p.k_s < 0; k++)
        if (p.m - 1 - k <= cx().spec_lg && one_pair_per_lane(k)) p.k_s = k;
}
// zero accumulators at the start of a round loop: they are zero between launches (the last workgroup of every launch resets
// them); only a call that failed half-way can leave them dirty
int rounds_begin(bool collective) {
    if (cx().racc_dirty) {
        HIPCHK(hipMemsetAsync(cx().d_racc, 0, sizeof(unsigned long long) * kRaccWords * GKR_RACC_SLOTS, cx().stream));
        HIPCHK(hipMemsetAsync(cx().d_counter, 0, sizeof(unsigned int), cx().stream));
        if (cx().d_spec_racc) HIPCHK(hipMemsetAsync(cx().d_spec_racc, 0, sizeof(unsigned long long) * GKR_SPEC_CAND * GKR_SPEC_SET_WORDS, cx().stream));
    }
    cx().racc_dirty = true;                            // until the loop has run to its end
    if (collective) CHK(coll_buffers(256));
    return 0;
}
// the challenge slot of a kernel queued ahead of its challenge (slot 0: the pre-launched round kernels; 1, 2: the speculative ones)
template <class A>
void arm_challenge_wait(A& a, int slot, ChalGuard& guard) {
    a.chal = cx().d_chal + (size_t)slot * GKR_CHAL_WORDS;
    a.chal_dev = cx().d_chal_dev + (size_t)slot * GKR_CHAL_WORDS;
    a.chal_seq = a.seq;
    guard.armed = true;
    cx().dbg_defer_seq = a.seq;
    cx().dbg_defer_ms = now_ms();
}

// The loop.  L supplies the layer's own parts:
//   NCOEF                          coefficients of a round polynomial (9 | 3); NSUM / NTAIL: words exchanged / tail words of a hand-off
//   launch_round(k, deferred, r, derive_m0, &out)   queue round k's kernel (deferred: it takes r_{k-1} from the challenge slot)
//   launch_spec(k), spec_seq[k], spec_flag(k)       the speculative launch of round k and where its result lands
//   coefficients(k, this_spec, derive_m0, sums, co) the round polynomial from the sums (or from the candidates at chal[k-1])
//   finish_round(k, this_spec, r, &r_prev)          tail words of the last round; export + host tail at k == k_export
template <class L>
int run_rounds(L& lp, const RoundPlan& pl, double t_setup0) {
    const int m = pl.m;
    const bool collective = pl.collective;
    const E two128 = {{0, 0, 1, 0}};                 // the plain integer 2^128: a Montgomery product with it divides by 2^128
    E r_prev = hfr::ZERO;
    InFlight cur, nxt;
    bool spec_queued = false;                        // the layer's first speculative launch is in the stream
    {
        const double t_l0 = now_ms();
        cx().prof.setup_ms += t_l0 - t_setup0;
        if (lp.ahead_round0()) {      // round 0's sums were queued by the layer before this one (ahead_launch): nothing to launch
            cur.tg = round_targets(collective);
            cur.seq = cx().ahead_seq;
            cur.derive_m0 = true;
        } else {
            CHK(lp.launch_round(0, false, hfr::ZERO, lp.claim && *lp.claim_known, &cur));
        }
        cx().prof.host_launch_ms += now_ms() - t_l0;
        LAP("rounds: launch round 0");
    }
    bool pre_requested = cx().req_K != nullptr && pl.pre_on;
    for (int k = 0; k < pl.m_dev; k++) {
        const size_t P = lp.n >> (k + 1);
        const double t_l0 = now_ms();
        // round k+1 queued now, behind round k's kernel: its dispatch overlaps the hash below
        const bool have_next = k + 1 < pl.m_dev;
        const bool this_spec = pl.is_spec(k), next_spec = pl.is_spec(k + 1), next2_spec = pl.is_spec(k + 2);
        // (the round before the first speculative one is always pre-launched: the speculative launch rides behind it)
        const bool prelaunched = have_next && !next_spec && pl.pl_on && ((P >> 1) <= ((size_t)1 << cx().prelaunch_lg) || next2_spec);
        // Un-sharded: queued BEFORE waiting for round k (the launch call itself is hidden behind round k's kernel).
        // Sharded: AFTER the exchange of round k -- the kernel then only ever spins for the duration of this rank's own
        // hash, never for a peer that is seconds behind (ranks reach the first exchange of a proof at different times),
        // and over RCCL it stays behind the all-reduce and the publish kernel in stream order.
        if (prelaunched && !collective) {
            CHK(lp.launch_round(k + 1, true, hfr::ZERO, lp.claim != nullptr, &nxt));
            g_cnt_prelaunched.fetch_add(1, std::memory_order_relaxed);
            if (next2_spec) {
                // the first speculative launch of the layer: behind R_{k+1}, whose tables it reads; no challenge of its own
                // (queueing ALL of a layer's speculative launches here was measured equal for one proof and blocked in the
                // runtime with many lanes)
                CHK(lp.launch_spec(k + 2));
                spec_queued = true;
            }
        } else if (next_spec && next2_spec) {
            CHK(lp.launch_spec(k + 2));                          // one by one, two rounds ahead: it polls for r_k
        }
        // the next layer's q-independent round-0 products, on the look-ahead stream, once this layer's rounds are small
        // (from 2^24 entries on one round earlier: 2^24 alone 263.5 -> 260.6 ms with the start at 2^21 pairs, 2^23 181.6 / 181.9 / 180.9 with
        // 20 / 21 / 19: profiles/r06_pre_start_ab.txt)
        if (pre_requested && k >= 1 && (P <= ((size_t)1 << (cx().pre_start_lg + std::max(0, m - 23))) || k == pl.m_dev - 1)) {
            CHK(launch_pre());
            pre_requested = false;
        }
        // test hook: an error return while a kernel is waiting for its challenge -- the guard must tell it to leave, drain
        // the stream and clear the abort tags
        if ((prelaunched || (next_spec && next2_spec)) && test_fire(g_test_fail_round, k))   // (a speculative launch waits for r_k)
            return fail("injected failure after a pre-launch (test hook)");
        const double t_l1 = now_ms();
        unsigned long long summed[GKR_CR_WORDS + 1];
        static_assert(L::NSUM <= GKR_CR_WORDS, "exchange scratch");
        const unsigned long long* sums = nullptr;
        const bool corrupt = test_fire_corrupt(k);
        if (this_spec) {
            CHK(wait_flag(lp.spec_seq[k], lp.spec_flag(k)));
            if (corrupt) cx().h_spec[(size_t)(k & 1) * GKR_SPEC_BUF_WORDS + 4] ^= 1ull;      // candidate 0's M_1
        } else if (k == 0 && lp.ahead_round0()) {
            CHK(wait_flag(cx().ahead_seq, ahead_flag()));
            cx().ahead_in_flight = false;              // its last workgroup has raised the flag: nothing of it is running
            if (corrupt) cx().h_ahead[0] ^= 1ull;                                               // S_1(0)
            g_cnt_ahead.fetch_add(1, std::memory_order_relaxed);
        } else {
            g_corrupt_collect = corrupt;
            CHK(round_collect(collective, cur.tg, cur.seq, L::NSUM, L::NTAIL, summed, &sums));
        }
        const double t_w = now_ms();
        if (prelaunched && collective) {
            CHK(lp.launch_round(k + 1, true, hfr::ZERO, lp.claim != nullptr, &nxt));
            g_cnt_prelaunched.fetch_add(1, std::memory_order_relaxed);
        }
        const bool derive_m0 = this_spec ? lp.claim != nullptr : cur.derive_m0;
        E* co = lp.proof + (size_t)k * L::NCOEF;
        lp.coefficients(k, this_spec, derive_m0, sums, co);
        const double t_h0 = now_ms();
        const E r = hfr::mimc_hash(co, L::NCOEF);
        const double t_h1 = now_ms();
        double t_l2 = t_h1;
        if (next_spec) {
            if (next2_spec) {                                // round k+2's speculative launch folds with r_k
                if (!test_fire(g_test_drop_round, k)) chal_publish(lp.spec_seq[k + 2], r, r, 1 + (k & 1));
                lp.chal_guard.armed = k + 2 < pl.k_export;   // later speculative launches will still wait for theirs
            }
        } else if (prelaunched) {
            if (!test_fire(g_test_drop_round, k)) chal_publish(nxt.seq, r, hfr::mul(r, two128));   // the waiting kernel starts its fold
            lp.chal_guard.armed = spec_queued && pl.k_s < pl.k_export;      // (the speculative launches behind it wait for their own)
            cur = nxt;
        } else if (have_next) {
            CHK(lp.launch_round(k + 1, false, r, lp.claim != nullptr, &nxt));
            cur = nxt;
            t_l2 = now_ms();
        }
        lp.chal[k] = r;
        lp.c = hfr::mul(lp.c, hfr::eval_eq(&lp.q[k], &r, 1));
        r_prev = r;
        if (lp.claim) {   // next round's claim = P_k(r_k)
            *lp.claim = hfr::eval_univariate(co, L::NCOEF, r);
            *lp.claim_known = true;
        }
        CHK(lp.finish_round(k, this_spec, r, &r_prev));
        cx().prof.host_launch_ms += (t_l1 - t_l0) + (t_l2 - t_h1);
        cx().prof.host_wait_ms += t_w - t_l1;
        {
            int lg = 0;
            while (((size_t)1 << lg) < P) lg++;
            cx().prof.wait_lg[lg] += t_w - t_l1;
            cx().prof.cnt_lg[lg]++;
        }
        cx().prof.host_other_ms += t_h0 - t_w;
        cx().prof.host_hash_ms += t_h1 - t_h0;
        cx().prof.rounds++;
    }
    if (pre_requested) CHK(launch_pre());      // no round was small enough: still ahead of the next layer's pyramids
    lp.r_last = r_prev;
    const double t_end0 = now_ms();
    LAP("rounds: the loop");
    // Un-sharded, every kernel this loop queued has been waited for through its own flag, which its LAST workgroup raises after
    // all workgroups have arrived with their stores drained (publish_sums): nothing of the layer is running or has memory
    // traffic in flight, the scratch tables can go back to the arena without a stream synchronisation (19 us per layer).
    if (collective) HIPCHK(hipStreamSynchronize(cx().stream));
    LAP("rounds: final synchronize");
    cx().racc_dirty = false;
    cx().prof.setup_ms += now_ms() - t_end0;
    (void)m;
    return 0;
}

// The rounds of a single-point cipher sumcheck over tables K, S of 2^m entries (m >= 1) and coordinates
// q[0:m].  `seed` multiplies every eq weight (the shard weight; 1 on one GPU); with `collective` the
// monomial sums are all-reduced across ranks before the host reads them.  On return: c has absorbed
// eq(q_k, r_k) of every round, proof/chal hold m rounds, tail = the two remaining entries of each table
// (K_lo, K_hi, S_lo, S_hi) and r_last the last challenge (the caller applies the final fold).
struct CipherLoop {
    static const int NCOEF = 9, NSUM = GKR_CR_WORDS, NTAIL = 16;
    // the call
    const E& ark;
    const int m;
    const DevTable *K, *S;
    const E* q;
    const E& seed;
    const bool collective;
    E& c;
    E *proof, *chal, *tail;
    E& r_last;
    E* claim;                  // running claim, or nullptr
    bool* claim_known;
    const int gamma_tail;
    bool* did_gamma;
    // the loop's state
    const size_t n;
    RoundPlan pl;
    int g_m = 16, g_big = 0, gsplit[2] = {0, 0};
    ScopedTable pyrT, pyrH, pyrU[2], pyrU2[2], ks, ss, ks2, ss2;      // pyrU[0]: split at g_max threads, pyrU[1]: at g_big; pyrU2: times 2^-128; (ks2, ss2): the speculative launches alternate between (ks, ss) and these
    ChalGuard chal_guard;
    std::vector<unsigned int> spec_seq;              // by round
    bool coop_on = false;
    bool use_ahead = false;                          // round 0's class sums are in flight (queued by the layer before)
    bool ahead_round0() const { return use_ahead; }

    CipherLoop(const E& ark_, int m_, const DevTable* K_, const DevTable* S_, const E* q_, const E& seed_, bool collective_, E& c_, E* proof_,
               E* chal_, E* tail_, E& r_last_, E* claim_, bool* claim_known_, int gamma_tail_, bool* did_gamma_)
        : ark(ark_), m(m_), K(K_), S(S_), q(q_), seed(seed_), collective(collective_), c(c_), proof(proof_), chal(chal_), tail(tail_),
          r_last(r_last_), claim(claim_), claim_known(claim_known_), gamma_tail(gamma_tail_), did_gamma(did_gamma_), n((size_t)1 << m_),
          spec_seq((size_t)m_ + 2, 0u) {}

    void release_tables() {      // the stream is idle (run_rounds synchronised it)
        for (ScopedTable* t : {&pyrT, &pyrH, &pyrU[0], &pyrU[1], &pyrU2[0], &pyrU2[1], &ks, &ss, &ks2, &ss2})
            if (t->base) table_release(t);
    }
    // Threads of a round = 2^g.  With other proofs in flight 2^g_max threads (one workgroup per CU) is best: the other lanes'
    // kernels fill the second wave slot.  A proof that is alone on the GPU gets twice the threads for the rounds that still
    // have two pairs per lane, and runs the round with 2^g_big pairs one pair per lane (two waves per SIMD).
    int threads_log2(int k) const {
        const int rem = m - 1 - k;                     // log2(pairs of the round)
        return rem >= g_big ? g_big : std::min(g_m, rem);
    }
    volatile unsigned int* spec_flag(int k) const {
        return (volatile unsigned int*)(cx().h_spec + (size_t)(k & 1) * GKR_SPEC_BUF_WORDS + GKR_SPEC_FLAG_WORD);
    }

    // pyramids (the eq weights, never a table of 2^m entries), scratch tables, accumulators, the plan of the rounds
    int setup() {
        // class sums of this layer's round 0 queued by the layer before (ahead_launch): they are ours if they were computed from
        // these tables, this Ark and the first m - t coordinates of this point; taken or not, they are spent
        use_ahead = cx().ahead_K && cx().ahead_K == K->base && cx().ahead_S == S->base && cx().ahead_m == m && cx().ahead_ark == ark &&
                    !collective && !g_safe_mode && claim && *claim_known && seed == hfr::ONE &&
                    memcmp(q, cx().ahead_q.data(), sizeof(E) * (size_t)(m - cx().ahead_t)) == 0;
        cx().ahead_K = cx().ahead_S = nullptr;
        const bool solo = cx().solo_boost && !collective &&
                          (cx().solo_boost >= 2 || proofs_in_flight_now() <= 1);   // 2: always
        g_m = round_threads_log2_max(m);                // fixed for the layer: the number of proofs in flight may change under it
        g_big = solo ? std::min(g_m + 1, 17) : g_m;
        const int gT = std::max(threads_log2(0), std::min(g_m, m - 1));   // highest level of the per-lane pyramid
        LAP("setup: enter");
        PyramidArgs3 pa3;
        memset(&pa3, 0, sizeof pa3);
        const Fr* qsrc = nullptr;
        CHK(pyramid_coords(pa3, q, (size_t)m, &qsrc));
        LAP("setup: stage_coords");
        gsplit[0] = std::min(g_m, m - 1);
        gsplit[1] = g_big;
        CHK(table_alloc(&pyrT, (size_t)2 << gT));
        CHK(table_alloc(&ks, std::max<size_t>(n / 2, 1)));
        CHK(table_alloc(&ss, std::max<size_t>(n / 2, 1)));
        // the per-lane pyramid (over all of q) and the per-iteration pyramids (one per thread split) in ONE launch
        for (int v = 0; v < 4; v++) pa3.p[v].max_level = -1;
        // the per-lane pyramid in two steps when it is wide (option pyr_split, default 12): levels up to 2^12 entries and the small
        // pyramid H over the next coordinates here, the upper levels by k_eq_pyramid_expand with one product per entry
        const int gLow = (cx().pyr_split > 0 && gT > cx().pyr_split + 1) ? cx().pyr_split : gT;
        pa3.p[0].out = pyrT.planes();
        pa3.p[0].out2 = Planes{nullptr, nullptr};
        pa3.p[0].q = qsrc;
        pa3.p[0].nc = m;
        pa3.p[0].max_level = gLow;
        pa3.p[0].seed = to_dev(seed);
        if (gLow < gT) {
            CHK(table_alloc(&pyrH, (size_t)2 << (gT - gLow)));
            pa3.p[3].out = pyrH.planes();
            pa3.p[3].out2 = Planes{nullptr, nullptr};
            pa3.p[3].q = qsrc;
            pa3.p[3].nc = m - gLow;
            pa3.p[3].max_level = gT - gLow;
            pa3.p[3].seed = to_dev(hfr::ONE);
        }
        int widest = std::max(gLow, gT - gLow);            // the launch covers the widest of its pyramids
        for (int v = 0; v < (g_big != gsplit[0] ? 2 : 1); v++) {
            const int mU = m - 1 - gsplit[v];              // log2(iterations of round 0 at this split)
            CHK(table_alloc(&pyrU[v], (size_t)2 << std::max(mU, 0)));
            CHK(table_alloc(&pyrU2[v], (size_t)2 << std::max(mU, 0)));
            if (mU > 0) {
                PyramidArgs& pa = pa3.p[1 + v];
                pa.out = pyrU[v].planes();
                pa.out2 = pyrU2[v].planes();
                pa.q = qsrc;
                pa.nc = m - gsplit[v];                     // q[0 .. m-g-1]; level L = eq(q[nc-L .. nc-1], .)
                pa.max_level = mU;
                pa.seed = to_dev(hfr::ONE);
                widest = std::max(widest, mU);
            }
        }
        LAP("setup: table allocs");
        GKR_LAUNCH_BATCH(k_eq_suffix_pyramids, dim3(grid_for((size_t)1 << widest, 1 << 20), 4), dim3(GKR_BLOCK), 0, cx().stream, pa3);
        HIPCHK(hipGetLastError());
        if (gLow < gT) {
            PyramidExpandArgs xa;
            xa.out = pyrT.planes();
            xa.h = pyrH.cplanes();
            xa.lo_level = gLow;
            xa.hi_level = gT;
            GKR_LAUNCH_BATCH(k_eq_pyramid_expand, dim3(grid_for((size_t)1 << gT, 1 << 20)), dim3(GKR_BLOCK), 0, cx().stream, xa);
            HIPCHK(hipGetLastError());
        }
        LAP("setup: pyramid launches");
        CHK(rounds_begin(collective));
        pl = plan_rounds(m, collective, gamma_tail, did_gamma, 2, proofs_in_flight_now() <= 1 ? cx().host_tail_solo : cx().host_tail);
        if (pl.pre_on) CHK(pre_prepare());               // nothing of this lane is waiting for the host yet
        LAP("setup: begin, plan, pre_prepare");
        coop_on = cx().coop >= 2 || (cx().coop == 1 && pl.alone);
        plan_speculation(pl, true, [&](int k) { return threads_log2(k) == m - 1 - k; });
        if (pl.k_s >= 0) {
            CHK(spec_ensure());
            if (pl.k_s < pl.k_export) {
                CHK(table_alloc(&ks2, (size_t)4 << (m - 1 - (pl.k_s + 1))));      // round k_s + 1 stores the tables of round k_s: 4 P entries
                CHK(table_alloc(&ss2, (size_t)4 << (m - 1 - (pl.k_s + 1))));
            }
        }
        LAP("setup: speculation tables");
        return 0;
    }

    // queue round k's kernel; `deferred`: r_prev is not known yet, the kernel takes it from the challenge slot
    int launch_round(int k, bool deferred, const E& r_in, bool derive_m0, InFlight* out) {
        const size_t P = n >> (k + 1);
        const int gk = threads_log2(k);
        const int lj = m - 1 - k - gk;                 // log2(iterations)
        CipherRoundArgs a;
        memset(&a, 0, sizeof a);
        const bool fold = k > 0;
        a.prio = (unsigned)std::min(k, 3);             // wave priority rises with the round index (round_wave_priority, kernels.hip.h)
        a.k_src = (k <= 1 ? K : &ks)->cplanes();
        a.s_src = (k <= 1 ? S : &ss)->cplanes();
        a.k_dst = ks.planes();
        a.s_dst = ss.planes();
        const size_t offT = ((size_t)1 << gk) - 1;
        a.wt = CPlanes{pyrT.base + offT, pyrT.base + pyrT.cap + offT};
        if (lj > 0) {
            const DevTable& pu = pyrU[gk == gsplit[0] ? 0 : 1];
            const size_t offU = ((size_t)1 << lj) - 1;
            a.wj = CPlanes{pu.base + offU, pu.base + pu.cap + offU};
            const DevTable& pu2 = pyrU2[gk == gsplit[0] ? 0 : 1];
            a.wj2 = CPlanes{pu2.base + offU, pu2.base + pu2.cap + offU};
        }
        a.P = P;
        a.lg_threads = (unsigned)gk;
        a.ark = to_dev(ark);
        a.partials = cx().d_racc;
        a.counter = cx().d_counter;
        a.tail_tables = k == pl.k_export ? cx().d_tail : nullptr;
        out->tg = round_targets(collective);
        a.host_out = out->tg.out;
        a.host_flag = out->tg.flag;
        a.seq = out->seq = ++cx().seq;
        out->derive_m0 = derive_m0;
        a.need_m0 = derive_m0 ? 0u : 1u;
        if (deferred) {
            arm_challenge_wait(a, 0, chal_guard);
            a.chal_limit_s = collective && !host_exchange() ? 20u : 0u;      // no retry across ranks on the device-side exchange: a generous limit there
        } else {
            const E two128 = {{0, 0, 1, 0}};                 // the plain integer 2^128: mul divides by 2^256
            a.r = to_dev(r_in);
            a.r_lo = to_dev(hfr::mul(r_in, two128));         // r * 2^-128 (fr_mul_const2_raw's second image)
        }
        const int grid = (int)std::max<size_t>(((size_t)1 << gk) / GKR_BLOCK, 1);
        const bool timed = 2 * P >= cx().prof.min_n && !deferred;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        // interleaved-pair (latency) variant for the rounds with one pair per lane; GKRHIP_LAT=0 never, 2 always
        const bool lat = cx().lat_mode == 2 || (cx().lat_mode == 1 && lj == 0);
        const bool wide = cx().wide_mode && lj > 0 && derive_m0 && !lat && k != pl.k_export;   // only the plain kernels export
        const bool late = wide && lj >= cx().wt_late_lj;  // the lane weight multiplies the sums after the loop: 8 products per lane
        // small rounds of a proof that is alone on the GPU: eight lanes per pair (cipher_coop.hip.h)
        const bool coop = coop_on && lj == 0 && P <= ((size_t)1 << cx().coop_lg);
        // round 0 with the q-independent products computed ahead (k_cipher_pre, launched during the previous layer)
        const bool pre = wide && !fold && cx().pre_K && cx().pre_K == K->base && cx().pre_S == S->base && cx().pre_m == m &&
                         cx().pre_ark == ark;
        if (pre) {
            HIPCHK(hipStreamWaitEvent(cx().stream, cx().pre_done, 0));
            for (int i = 0; i < 6; i++) a.pre[i] = cx().pre_t[i].cplanes();
            cx().pre_K = cx().pre_S = nullptr;             // consumed
        }
        if (timed) {
            e0 = prof_event();
            e1 = prof_event();
            HIPCHK(hipEventRecord(e0, cx().stream));
        }
        if (coop) {
            const int cgrid = (int)std::min<size_t>((P + GKR_COOP_PAIRS - 1) / GKR_COOP_PAIRS, (size_t)cx().coop_wgs);
            if (fold) hipLaunchKernelGGL((k_cipher_round_coop<true>), dim3(cgrid), dim3(GKR_BLOCK), 0, cx().stream, a);
            else hipLaunchKernelGGL((k_cipher_round_coop<false>), dim3(cgrid), dim3(GKR_BLOCK), 0, cx().stream, a);
            g_cnt_coop.fetch_add(1, std::memory_order_relaxed);
        } else if (pre) {
            if (late) GKR_LAUNCH_BATCH((k_cipher_round_wide<false, true, true>), dim3(grid), dim3(GKR_BLOCK), 0, cx().stream, a);
            else GKR_LAUNCH_BATCH((k_cipher_round_wide<false, false, true>), dim3(grid), dim3(GKR_BLOCK), 0, cx().stream, a);
            g_cnt_lookahead.fetch_add(1, std::memory_order_relaxed);
        } else if (wide) {
            if (fold) {
                if (late) GKR_LAUNCH_BATCH((k_cipher_round_wide<true, true>), dim3(grid), dim3(GKR_BLOCK), 0, cx().stream, a);
                else GKR_LAUNCH_BATCH((k_cipher_round_wide<true, false>), dim3(grid), dim3(GKR_BLOCK), 0, cx().stream, a);
            } else {
                if (late) GKR_LAUNCH_BATCH((k_cipher_round_wide<false, true>), dim3(grid), dim3(GKR_BLOCK), 0, cx().stream, a);
                else GKR_LAUNCH_BATCH((k_cipher_round_wide<false, false>), dim3(grid), dim3(GKR_BLOCK), 0, cx().stream, a);
            }
        } else if (fold) {
            if (lj > 0) CHK((launch_cipher_round<true, true>(a, grid, lat, pl.alone)));
            else CHK((launch_cipher_round<true, false>(a, grid, lat, pl.alone)));
        } else {
            if (lj > 0) CHK((launch_cipher_round<false, true>(a, grid, lat, pl.alone)));
            else CHK((launch_cipher_round<false, false>(a, grid, lat, pl.alone)));
        }
        HIPCHK(hipGetLastError());
        if (timed) {
            HIPCHK(hipEventRecord(e1, cx().stream));
            cx().prof.peval_ev.emplace_back(e0, e1);
            cx().prof.peval_launches++;
            cx().prof.peval_modmuls += ((derive_m0 ? 17.0 : 18.0) + (lj > 0 && !late ? 1.0 : 0.0) + (fold ? 4.0 : 0.0)) * (double)P;
        }
        return 0;
    }

    // queue the speculative launch of round k (k_s <= k <= k_export).  k == k_s reads the tables R_{k-1} leaves in (ks, ss)
    // and needs no challenge; later rounds read the tables of round k-2, fold them with r_{k-2} (polled from slot
    // 1 + (k & 1); launched with the challenge as an argument instead: measured equal or slower) and store the tables of
    // round k-1 in the other pair of buffers
    int launch_spec(int k) {
        const size_t P = n >> (k + 1);
        const int gk = m - 1 - k;
        const bool pref = k == pl.k_s, odd = ((k - pl.k_s) & 1) != 0, last = k == pl.k_export;
        CipherSpecArgs a;
        memset(&a, 0, sizeof a);
        const ScopedTable& srcK = (pref || odd) ? ks : ks2;
        const ScopedTable& srcS = (pref || odd) ? ss : ss2;
        a.k_src = srcK.cplanes();
        a.s_src = srcS.cplanes();
        if (!pref && !last) {
            a.k_dst = (odd ? ks2 : ks).planes();
            a.s_dst = (odd ? ss2 : ss).planes();
        }
        const size_t offT = ((size_t)1 << gk) - 1;
        a.wt = CPlanes{pyrT.base + offT, pyrT.base + pyrT.cap + offT};
        a.P = P;
        a.ark = to_dev(ark);
        for (int i = 0; i < GKR_SPEC_CAND; i++) a.rho[i] = to_dev(cx().spec_pts[i]);
        a.partials = cx().d_spec_racc;
        a.counter = cx().d_counter;
        a.host_out = cx().d_spec + (size_t)(k & 1) * GKR_SPEC_BUF_WORDS;
        a.seq = spec_seq[k] = ++cx().seq;
        a.need_m0 = claim ? 0u : 1u;
        a.prefolded = pref ? 1u : 0u;
        a.tail_tables = last ? cx().d_tail : nullptr;
        if (!pref) arm_challenge_wait(a, 1 + (k & 1), chal_guard);
        const int gx = (int)std::max<size_t>(P / GKR_BLOCK, 1);
        const bool row8 = !pref || last;             // the fold-and-store row (a plain copy to the host when the tables are folded already)
        hipLaunchKernelGGL(k_cipher_round_spec, dim3(gx, row8 ? GKR_SPEC_CAND + 1 : GKR_SPEC_CAND), dim3(GKR_BLOCK), 0, cx().stream, a);
        HIPCHK(hipGetLastError());
        g_cnt_spec.fetch_add(1, std::memory_order_relaxed);
        return 0;
    }

    // S_k(t) = sum_j C(7,j) M_j t^j ;  P_k(t) = c_k * ((1-q_k) + (2 q_k - 1) t) * S_k(t)
    // csp[j] = c_k * C(7,j) * M_j.  With a known claim, P_k(0) + P_k(1) = claim_k gives
    // c_k*M_0 = claim_k - q_k * sum_{j>=1} csp[j] (the verifier's round check, sumcheck/verifier.go:41-47).
    void coefficients(int k, bool this_spec, bool derive_m0, const unsigned long long* sums, E* co) const {
        static const hfr::u64 binom7[8] = {1, 7, 21, 35, 35, 21, 7, 1};
        E csp[8], Mj[8];
        if (k == 0 && use_ahead) {
            // M_j = sum_y eq(q[m-t:], y) S_j(y): the class sums contracted with the coordinates the layer before drew last
            LAP("ahead: before contraction");
            const int t = cx().ahead_t;
            ahead_contract((const E*)cx().h_ahead, q + (m - t), t, Mj + 1);
            LAP("ahead: contraction");
        } else if (this_spec) {     // the candidates at the true r_{k-1}
            spec_interpolate((const E*)(cx().h_spec + (size_t)(k & 1) * GKR_SPEC_BUF_WORDS), chal[k - 1], derive_m0 ? 1 : 0, Mj);
        } else {
            for (int j = derive_m0 ? 1 : 0; j < 8; j++) Mj[j] = limbs9_to_fr(sums + (size_t)j * GKR_ACC_WORDS);
        }
        for (int j = derive_m0 ? 1 : 0; j < 8; j++) csp[j] = hfr::mul(c, hfr::mul(Mj[j], hfr::from_u64(binom7[j])));
        if (derive_m0) {
            E rest = csp[1];
            for (int j = 2; j < 8; j++) rest = hfr::add(rest, csp[j]);
            csp[0] = hfr::sub(*claim, hfr::mul(q[k], rest));
        }
        const E a0 = hfr::sub(hfr::ONE, q[k]);
        const E a1 = hfr::sub(hfr::add(q[k], q[k]), hfr::ONE);
        co[0] = hfr::mul(a0, csp[0]);
        for (int j = 1; j < 8; j++) co[j] = hfr::add(hfr::mul(a0, csp[j]), hfr::mul(a1, csp[j - 1]));
        co[8] = hfr::mul(a1, csp[7]);
    }

    // after round k's challenge: the tail words of the last device round; at k == k_export the exported tables folded with
    // this round's challenge are the host's starting point (GKRHIP_HOST_TAIL)
    int finish_round(int k, bool this_spec, const E& r, E* r_prev) {
        const size_t P = n >> (k + 1);
        if (k == m - 1) {
            memcpy(tail, cx().h_round + GKR_CR_WORDS, 4 * sizeof(E));  // written by the P == 1 launch
            if (test_fire(g_test_corrupt_tail, 1)) tail[0].l[0] ^= 1ull;
        }
        if (k != pl.k_export) return 0;
        if (test_fire(g_test_corrupt_tail, 1)) cx().h_tail[0] ^= 1ull;
        const E* tt = (const E*)cx().h_tail;
        std::vector<E> Kh(P), Sh(P);
        if (this_spec) {
            // the speculative launch exported the tables of round k-1 (4P entries each): two folds on the host
            const E& r1 = chal[k - 1];
            for (size_t x = 0; x < P; x++) {
                Kh[x] = fold2(fold2(tt[x], tt[x + 2 * P], r1), fold2(tt[x + P], tt[x + 3 * P], r1), r);
                Sh[x] = fold2(fold2(tt[4 * P + x], tt[4 * P + x + 2 * P], r1), fold2(tt[4 * P + x + P], tt[4 * P + x + 3 * P], r1), r);
            }
        } else {
            for (size_t x = 0; x < P; x++) {
                Kh[x] = fold2(tt[x], tt[x + P], r);
                Sh[x] = fold2(tt[2 * P + x], tt[3 * P + x], r);
            }
        }
        const int mm = m - 1 - k;                  // variables left
        // the host tail starts: the GPU has nothing more to do for this layer -- round 0 of the next one, ahead of its last coordinates
        if (!collective && !g_safe_mode && claim && (cx().ahead_mode >= 2 || (cx().ahead_mode == 1 && pl.alone)))
            CHK(ahead_launch(m, chal, k, g_big > g_m));
        const double t_t0 = now_ms(), h_before = cx().prof.host_hash_ms;
        if (pl.sh_tail) {
            // gather every rank's P + P folded entries; global index = local index * world + rank (the shard bits are
            // the LOWEST index bits, bound last), eq weights over all remaining coordinates with seed 1
            const ShardView sv = shard_view();
            std::vector<E> mine(2 * P), all;
            for (size_t x = 0; x < P; x++) {
                mine[x] = Kh[x];
                mine[P + x] = Sh[x];
            }
            CHK(coll_allgather(mine.data(), (int)(2 * P), all));
            std::vector<E> Kg(P * sv.world), Sg(P * sv.world);
            for (int g = 0; g < sv.world; g++)
                for (size_t x = 0; x < P; x++) {
                    Kg[x * sv.world + g] = all[(size_t)g * 2 * P + x];
                    Sg[x * sv.world + g] = all[(size_t)g * 2 * P + P + x];
                }
            host_cipher_rounds(ark, mm + gamma_tail, Kg, Sg, q + k + 1, hfr::ONE, c, proof + (size_t)9 * (k + 1), chal + k + 1, claim,
                               claim_known);
            tail[0] = tail[1] = Kg[0];
            tail[2] = tail[3] = Sg[0];
            *r_prev = chal[m + gamma_tail - 1];
            *did_gamma = true;
        } else {
            host_cipher_rounds(ark, mm, Kh, Sh, q + k + 1, seed, c, proof + (size_t)9 * (k + 1), chal + k + 1, claim, claim_known);
            // hand back in the shape the device path uses: the caller folds (lo, hi) with r_last
            tail[0] = tail[1] = Kh[0];
            tail[2] = tail[3] = Sh[0];
            *r_prev = chal[m - 1];
        }
        cx().prof.tail_ms += (now_ms() - t_t0) - (cx().prof.host_hash_ms - h_before);
        return 0;
    }
};
int cipher_rounds(const E& ark, int m, const DevTable* K, const DevTable* S, const E* q, const E& seed, bool collective,
                  E& c, E* proof, E* chal, E tail[4], E& r_last, E* claim /* running claim, or nullptr */,
                  bool* claim_known, int gamma_tail = 0, bool* did_gamma = nullptr) {
    const double t_setup0 = now_ms();
    CipherLoop lp(ark, m, K, S, q, seed, collective, c, proof, chal, tail, r_last, claim, claim_known, gamma_tail, did_gamma);
    CHK(lp.setup());
    CHK(run_rounds(lp, lp.pl, t_setup0));
    lp.release_tables();      // (scoped tables: an error return releases them too, after draining the stream)
    LAP("rounds: release tables");
    return 0;
}

// host elements -> a small device table (boundary helper for the gathered shard tables)
int small_table(DevTable* t, const std::vector<E>& v) {
    CHK(table_alloc(t, v.size()));
    return upload_table(t, (const uint64_t*)v.data(), v.size());
}

// sumcheck.Prove for the cipher gate with one evaluation point.  bN is the GLOBAL number of variables; K and
// S are this rank's shard (2^(bN-gamma) entries, indices = rank mod world).  Phase 1: the bN-gamma local
// rounds (sums all-reduced); then one element per table per rank is all-gathered and the last gamma
// rounds run redundantly on every rank (phase 2).
int sumcheck_cipher_fast(const E& ark, int bN, const DevTable* K, const DevTable* S, const E* q, E* proof, E* challenges,
                         E* final_claims, const E* trusted_claim, bool track_claim) {
    const ShardView shard = shard_view();
    const int gamma = shard.gamma, m1 = bN - gamma;
    if (m1 < 0) return fail("bN %d is smaller than log2(world) %d", bN, gamma);
    E c = hfr::ONE, tail[4], r_last, kv, sv;
    bool did_gamma = false;      // the sharded host tail has run the rounds over the shard bits as well
    // running claim: known from the start when the caller vouches for it, otherwise from round 1 on
    E claim = trusted_claim ? *trusted_claim : hfr::ZERO;
    bool claim_known = trusted_claim != nullptr;
    E* claim_p = track_claim ? &claim : nullptr;
    if (m1 >= 1) {
        const E seed = gamma ? shard_seed(q + m1, gamma, shard.rank) : hfr::ONE;
        // (see rounds_with_retry: a layer whose pre-launched kernel missed its challenge is run once more, nothing queued ahead)
        CHK(rounds_with_retry(c, claim, claim_known, [&]() {
            return cipher_rounds(ark, m1, K, S, q, seed, gamma > 0 || cx().force_collective, c, proof, challenges, tail, r_last,
                                 claim_p, &claim_known, gamma, &did_gamma);
        }));
        kv = fold2(tail[0], tail[1], r_last);
        sv = fold2(tail[2], tail[3], r_last);
    } else {
        const DevTable* t[2] = {K, S};
        E v[2];
        CHK(gather0(t, 2, v));
        kv = v[0];
        sv = v[1];
    }
    if (gamma > 0 && !did_gamma) {
        const E mine[2] = {kv, sv};
        std::vector<E> all;
        CHK(coll_allgather(mine, 2, all));
        std::vector<E> k2(shard.world), s2(shard.world);
        for (int r = 0; r < shard.world; r++) {
            k2[r] = all[2 * r];
            s2[r] = all[2 * r + 1];
        }
        if (cx().host_tail > 0) {
            // the gathered tables are host data already and the rounds are tiny: no device round trip at all
            host_cipher_rounds(ark, gamma, k2, s2, q + m1, hfr::ONE, c, proof + (size_t)9 * m1, challenges + m1, claim_p, &claim_known);
            kv = k2[0];
            sv = s2[0];
        } else {
            ScopedTable K2, S2;
            CHK(small_table(&K2, k2));
            CHK(small_table(&S2, s2));
            CHK(cipher_rounds(ark, gamma, &K2, &S2, q + m1, hfr::ONE, false, c, proof + (size_t)9 * m1, challenges + m1, tail,
                              r_last, claim_p, &claim_known));
            kv = fold2(tail[0], tail[1], r_last);
            sv = fold2(tail[2], tail[3], r_last);
            table_release(&K2);
            table_release(&S2);
        }
    }
    final_claims[0] = c;
    final_claims[1] = kv;
    final_claims[2] = sv;
    return 0;
}

// The same for a linear gate (host_cipher_rounds' counterpart): T[t] hold 2^mm entries, g.mask selects the tables of the sum.
void host_linear_rounds(const GateDesc& g, const E& ark, int mm, std::vector<std::vector<E>>& T, const E* q, const E& seed, E& c,
                        E* proof, E* chal, E* claim, bool* claim_known) {
    for (int k = 0; k < mm; k++) {
        const size_t P = (size_t)1 << (mm - 1 - k);
        std::vector<E> W(P);
        W[0] = seed;
        for (int i = 0; i < mm - 1 - k; i++) {
            const E& qi = q[k + 1 + i];
            for (size_t t = 0; t < ((size_t)1 << i); t++) {
                const size_t J = t << (mm - 1 - k - i), JN = J + ((size_t)1 << (mm - 2 - k - i));
                W[JN] = hfr::mul(qi, W[J]);
                W[J] = hfr::sub(W[J], W[JN]);
            }
        }
        const bool derive_m0 = claim && *claim_known;
        E m0 = hfr::ZERO, m1 = hfr::ZERO;
        for (size_t x = 0; x < P; x++) {
            E u = ark, d = hfr::ZERO;
            for (int t = 0; t < g.n_in; t++)
                if ((g.mask >> t) & 1u) {
                    u = hfr::add(u, T[t][x]);
                    d = hfr::add(d, hfr::sub(T[t][x + P], T[t][x]));
                }
            if (!derive_m0) m0 = hfr::add(m0, hfr::mul(W[x], u));
            m1 = hfr::add(m1, hfr::mul(W[x], d));
        }
        const E cm1 = hfr::mul(c, m1);
        const E cm0 = derive_m0 ? hfr::sub(*claim, hfr::mul(q[k], cm1)) : hfr::mul(c, m0);
        const E a0 = hfr::sub(hfr::ONE, q[k]);
        const E a1 = hfr::sub(hfr::add(q[k], q[k]), hfr::ONE);
        E* co = proof + (size_t)k * 3;
        co[0] = hfr::mul(a0, cm0);
        co[1] = hfr::add(hfr::mul(a0, cm1), hfr::mul(a1, cm0));
        co[2] = hfr::mul(a1, cm1);
        const E r = hfr::mimc_hash(co, 3);
        chal[k] = r;
        c = hfr::mul(c, hfr::eval_eq(&q[k], &r, 1));
        if (claim) {
            *claim = hfr::eval_univariate(co, 3, r);
            *claim_known = true;
        }
        for (int t = 0; t < g.n_in; t++) {
            for (size_t x = 0; x < P; x++) T[t][x] = fold2(T[t][x], T[t][x + P], r);
            T[t].resize(P);
        }
        cx().prof.rounds++;
    }
}

// ---- single-point sumcheck of a linear gate: one fused launch per round (linear_round.hip.h) ------------------
// Tables X[0..arity) of 2^m entries (m >= 1), coordinates q[0:m]; g.mask selects the tables that enter the gate's sum.
// `seed` multiplies every eq weight (the shard weight; 1 on one GPU); with `collective` the two sums are added over the
// ranks before the host reads them.  On return: c has absorbed eq(q_k, r_k) of every round, proof/chal hold m rounds
// of 3 coefficients, tail[2t], tail[2t+1] = the two remaining entries of table t and r_last the last challenge (the
// caller applies the final fold).  The loop itself is run_rounds (shared with the cipher rounds).
struct LinearLoop {
    static const int NCOEF = 3, NSUM = GKR_LR_WORDS, NTAIL = 8 * GKR_MAX_ARITY;
    const GateDesc& g;
    const E& ark;
    const int m;
    const DevTable* const* X;
    const E* q;
    const E& seed;
    const bool collective;
    E& c;
    E *proof, *chal, *tail;
    E& r_last;
    E* claim;
    bool* claim_known;
    const int gamma_tail;
    bool* did_gamma;
    const size_t n;
    const int arity;
    RoundPlan pl;
    int g_lin = 0;             // log2(max threads): as the cipher rounds (more lanes for these HBM-bound rounds: measured, no gain)
    ScopedTable pyrT, pyrU, scratch[GKR_MAX_ARITY], scratch2[GKR_MAX_ARITY];
    ChalGuard chal_guard;
    std::vector<unsigned int> spec_seq;

    LinearLoop(const GateDesc& g_, const E& ark_, int m_, const DevTable* const* X_, const E* q_, const E& seed_, bool collective_, E& c_,
               E* proof_, E* chal_, E* tail_, E& r_last_, E* claim_, bool* claim_known_, int gamma_tail_, bool* did_gamma_)
        : g(g_), ark(ark_), m(m_), X(X_), q(q_), seed(seed_), collective(collective_), c(c_), proof(proof_), chal(chal_), tail(tail_),
          r_last(r_last_), claim(claim_), claim_known(claim_known_), gamma_tail(gamma_tail_), did_gamma(did_gamma_), n((size_t)1 << m_),
          arity(g_.n_in), spec_seq((size_t)m_ + 2, 0u) {}

    volatile unsigned int* spec_flag(int k) const {
        return (volatile unsigned int*)(cx().h_spec + (size_t)(k & 1) * GKR_SPEC_BUF_WORDS + GKR_SPEC_FLAG_WORD);
    }
    bool ahead_round0() const { return false; }

    void release_tables() {
        table_release(&pyrT);
        table_release(&pyrU);
        for (int t = 0; t < arity; t++) {
            table_release(&scratch[t]);
            if (scratch2[t].base) table_release(&scratch2[t]);
        }
    }
    int setup() {
        g_lin = round_threads_log2_max(m);
        const int gT = std::min(g_lin, m - 1);
        const int mU = m - 1 - gT;
        PyramidArgs3 pa3;
        memset(&pa3, 0, sizeof pa3);
        const Fr* qsrc = nullptr;
        CHK(pyramid_coords(pa3, q, (size_t)m, &qsrc));
        CHK(table_alloc(&pyrT, (size_t)2 << gT));
        CHK(table_alloc(&pyrU, (size_t)2 << std::max(mU, 0)));
        for (int t = 0; t < arity; t++) CHK(table_alloc(&scratch[t], std::max<size_t>(n / 2, 1)));
        for (int v = 0; v < 4; v++) pa3.p[v].max_level = -1;
        pa3.p[0].out = pyrT.planes();
        pa3.p[0].q = qsrc;
        pa3.p[0].nc = m;
        pa3.p[0].max_level = gT;
        pa3.p[0].seed = to_dev(seed);
        if (mU > 0) {
            pa3.p[1].out = pyrU.planes();
            pa3.p[1].q = qsrc;
            pa3.p[1].nc = m - gT;
            pa3.p[1].max_level = mU;
            pa3.p[1].seed = to_dev(hfr::ONE);
        }
        GKR_LAUNCH_BATCH(k_eq_suffix_pyramids, dim3(grid_for((size_t)1 << std::max(gT, mU), 1 << 20), 3), dim3(GKR_BLOCK), 0,
                           cx().stream, pa3);
        HIPCHK(hipGetLastError());
        CHK(rounds_begin(collective));
        pl = plan_rounds(m, collective, gamma_tail, did_gamma, arity, cx().host_tail);
        if (pl.pre_on) CHK(pre_prepare());
        // speculative rounds (k_linear_round_spec): the two sums of a linear gate's round are linear in the previous challenge, so
        // the candidates 0 and 1 -- the lower and the upper half of the previous round's tables -- suffice.  The export must fit
        // the hand-off buffer: 4P entries per table, P = 2^(h+1)
        plan_speculation(pl, (size_t)arity * 4 * 4 * ((size_t)2 << pl.h_tail) <= kTailWords, [&](int k) { return std::min(g_lin, m - 1 - k) == m - 1 - k; });
        if (pl.k_s >= 0) {
            CHK(spec_ensure());
            if (pl.k_s < pl.k_export)
                for (int t = 0; t < arity; t++) CHK(table_alloc(&scratch2[t], (size_t)4 << (m - 1 - (pl.k_s + 1))));
        }
        return 0;
    }

    int launch_round(int k, bool deferred, const E& r_in, bool derive_m0, InFlight* out) {
        const size_t P = n >> (k + 1);
        const int gk = std::min(g_lin, m - 1 - k);
        const int lj = m - 1 - k - gk;
        LinearRoundArgs a;
        memset(&a, 0, sizeof a);
        const bool fold = k > 0;
        a.prio = (unsigned)std::min(k, 3);
        for (int t = 0; t < arity; t++) {
            a.src[t] = (k <= 1 ? X[t] : &scratch[t])->cplanes();
            a.dst[t] = scratch[t].planes();
        }
        const size_t offT = ((size_t)1 << gk) - 1;
        a.wt = CPlanes{pyrT.base + offT, pyrT.base + pyrT.cap + offT};
        if (lj > 0) {
            const size_t offU = ((size_t)1 << lj) - 1;
            a.wj = CPlanes{pyrU.base + offU, pyrU.base + pyrU.cap + offU};
        }
        a.P = P;
        a.lg_threads = (unsigned)gk;
        a.ark = to_dev(ark);
        a.arity = arity;
        a.sum_mask = g.mask;
        a.racc = cx().d_racc;
        a.counter = cx().d_counter;
        a.tail_tables = k == pl.k_export ? cx().d_tail : nullptr;
        out->tg = round_targets(collective);
        a.host_out = out->tg.out;
        a.host_flag = out->tg.flag;
        a.seq = out->seq = ++cx().seq;
        if (deferred) {
            arm_challenge_wait(a, 0, chal_guard);
            a.chal_limit_s = collective && !host_exchange() ? 20u : 0u;      // see CipherLoop::launch_round
        } else {
            const E two128 = {{0, 0, 1, 0}};
            a.r = to_dev(r_in);
            a.r_lo = to_dev(hfr::mul(r_in, two128));
        }
        out->derive_m0 = derive_m0;
        a.need_m0 = derive_m0 ? 0u : 1u;
        const int grid = (int)std::max<size_t>(((size_t)1 << gk) / GKR_BLOCK, 1);
        if (fold) {
            if (lj > 0) GKR_LAUNCH_BATCH((k_linear_round<true, true>), dim3(grid), dim3(GKR_BLOCK), 0, cx().stream, a);
            else GKR_LAUNCH_BATCH((k_linear_round<true, false>), dim3(grid), dim3(GKR_BLOCK), 0, cx().stream, a);
        } else {
            if (lj > 0) GKR_LAUNCH_BATCH((k_linear_round<false, true>), dim3(grid), dim3(GKR_BLOCK), 0, cx().stream, a);
            else GKR_LAUNCH_BATCH((k_linear_round<false, false>), dim3(grid), dim3(GKR_BLOCK), 0, cx().stream, a);
        }
        HIPCHK(hipGetLastError());
        return 0;
    }

    // the speculative launch of round k (see CipherLoop::launch_spec): k == k_s reads the tables R_{k-1} leaves in `scratch`; later
    // rounds read the tables of round k-2, fold them with r_{k-2} (polled from slot 1 + (k & 1)) and store round k-1's
    int launch_spec(int k) {
        const size_t P = n >> (k + 1);
        const int gk = m - 1 - k;
        const bool pref = k == pl.k_s, odd = ((k - pl.k_s) & 1) != 0, last = k == pl.k_export;
        LinearSpecArgs a;
        memset(&a, 0, sizeof a);
        for (int t = 0; t < arity; t++) {
            a.src[t] = ((pref || odd) ? scratch[t] : scratch2[t]).cplanes();
            if (!pref && !last) a.dst[t] = (odd ? scratch2[t] : scratch[t]).planes();
        }
        const size_t offT = ((size_t)1 << gk) - 1;
        a.wt = CPlanes{pyrT.base + offT, pyrT.base + pyrT.cap + offT};
        a.P = P;
        a.ark = to_dev(ark);
        a.arity = arity;
        a.sum_mask = g.mask;
        a.partials = cx().d_spec_racc;
        a.counter = cx().d_counter;
        a.host_out = cx().d_spec + (size_t)(k & 1) * GKR_SPEC_BUF_WORDS;
        a.seq = spec_seq[k] = ++cx().seq;
        a.need_m0 = claim ? 0u : 1u;
        a.prefolded = pref ? 1u : 0u;
        a.tail_tables = last ? cx().d_tail : nullptr;
        if (!pref) arm_challenge_wait(a, 1 + (k & 1), chal_guard);
        const int gx = (int)std::max<size_t>(P / GKR_BLOCK, 1);
        hipLaunchKernelGGL(k_linear_round_spec, dim3(gx, (!pref || last) ? GKR_LSPEC_CAND + 1 : GKR_LSPEC_CAND), dim3(GKR_BLOCK), 0,
                           cx().stream, a);
        HIPCHK(hipGetLastError());
        g_cnt_spec.fetch_add(1, std::memory_order_relaxed);
        return 0;
    }

    // S_k(t) = M_0 + M_1 t;  P_k(t) = c_k * ((1-q_k) + (2 q_k - 1) t) * S_k(t);  c_k*M_0 = claim_k - q_k*c_k*M_1
    void coefficients(int k, bool this_spec, bool derive_m0, const unsigned long long* sums, E* co) const {
        E M0 = hfr::ZERO, M1;
        if (this_spec) {                         // linear in the previous challenge: M(r) = M(0) + r (M(1) - M(0))
            const E* cand = (const E*)(cx().h_spec + (size_t)(k & 1) * GKR_SPEC_BUF_WORDS);      // M_0(0), M_1(0), M_0(1), M_1(1)
            const E& r1 = chal[k - 1];
            M1 = hfr::add(cand[1], hfr::mul(r1, hfr::sub(cand[3], cand[1])));
            if (!derive_m0) M0 = hfr::add(cand[0], hfr::mul(r1, hfr::sub(cand[2], cand[0])));
        } else {
            M1 = limbs9_to_fr(sums + GKR_ACC_WORDS);
            if (!derive_m0) M0 = limbs9_to_fr(sums);
        }
        const E cm1 = hfr::mul(c, M1);
        const E cm0 = derive_m0 ? hfr::sub(*claim, hfr::mul(q[k], cm1)) : hfr::mul(c, M0);
        const E a0 = hfr::sub(hfr::ONE, q[k]);
        const E a1 = hfr::sub(hfr::add(q[k], q[k]), hfr::ONE);
        co[0] = hfr::mul(a0, cm0);
        co[1] = hfr::add(hfr::mul(a0, cm1), hfr::mul(a1, cm0));
        co[2] = hfr::mul(a1, cm1);
    }

    int finish_round(int k, bool this_spec, const E& r, E* r_prev) {
        const size_t P = n >> (k + 1);
        if (k == m - 1) {
            memcpy(tail, cx().h_round + GKR_LR_WORDS, (size_t)2 * arity * sizeof(E));   // written by the P == 1 launch
            if (test_fire(g_test_corrupt_tail, 1)) tail[0].l[0] ^= 1ull;
        }
        if (k != pl.k_export) return 0;
        if (test_fire(g_test_corrupt_tail, 1)) cx().h_tail[0] ^= 1ull;
        const E* tt = (const E*)cx().h_tail;
        std::vector<std::vector<E>> Th(arity, std::vector<E>(P));
        if (this_spec) {                     // the speculative launch exported the tables of round k-1 (4P entries each)
            const E& r1 = chal[k - 1];
            for (int t = 0; t < arity; t++) {
                const E* tb = tt + (size_t)t * 4 * P;
                for (size_t x = 0; x < P; x++)
                    Th[t][x] = fold2(fold2(tb[x], tb[x + 2 * P], r1), fold2(tb[x + P], tb[x + 3 * P], r1), r);
            }
        } else {
            for (int t = 0; t < arity; t++)
                for (size_t x = 0; x < P; x++) Th[t][x] = fold2(tt[(size_t)t * 2 * P + x], tt[(size_t)t * 2 * P + x + P], r);
        }
        // (as CipherLoop::finish_round: round 0 of the next layer, when that is a cipher layer, ahead of its last coordinates)
        if (!collective && !g_safe_mode && claim && (cx().ahead_mode >= 2 || (cx().ahead_mode == 1 && pl.alone)))
            CHK(ahead_launch(m, chal, k, cx().solo_boost && pl.alone));
        if (pl.sh_tail) {
            const ShardView sv = shard_view();
            std::vector<E> mine((size_t)arity * P), all;
            for (int t = 0; t < arity; t++)
                for (size_t x = 0; x < P; x++) mine[(size_t)t * P + x] = Th[t][x];
            CHK(coll_allgather(mine.data(), (int)((size_t)arity * P), all));
            std::vector<std::vector<E>> Tg(arity, std::vector<E>(P * sv.world));
            for (int gr = 0; gr < sv.world; gr++)
                for (int t = 0; t < arity; t++)
                    for (size_t x = 0; x < P; x++) Tg[t][x * sv.world + gr] = all[(size_t)gr * arity * P + (size_t)t * P + x];
            host_linear_rounds(g, ark, m - 1 - k + gamma_tail, Tg, q + k + 1, hfr::ONE, c, proof + (size_t)3 * (k + 1), chal + k + 1, claim,
                               claim_known);
            for (int t = 0; t < arity; t++) tail[2 * t] = tail[2 * t + 1] = Tg[t][0];
            *r_prev = chal[m + gamma_tail - 1];
            *did_gamma = true;
        } else {
            host_linear_rounds(g, ark, m - 1 - k, Th, q + k + 1, seed, c, proof + (size_t)3 * (k + 1), chal + k + 1, claim, claim_known);
            for (int t = 0; t < arity; t++) tail[2 * t] = tail[2 * t + 1] = Th[t][0];
            *r_prev = chal[m - 1];
        }
        return 0;
    }
};
int linear_rounds(const GateDesc& g, const E& ark, int m, const DevTable* const* X, const E* q, const E& seed, bool collective,
                  E& c, E* proof, E* chal, E* tail, E& r_last, E* claim /* running claim, or nullptr */, bool* claim_known,
                  int gamma_tail = 0, bool* did_gamma = nullptr) {
    const double t_setup0 = now_ms();
    LinearLoop lp(g, ark, m, X, q, seed, collective, c, proof, chal, tail, r_last, claim, claim_known, gamma_tail, did_gamma);
    CHK(lp.setup());
    CHK(run_rounds(lp, lp.pl, t_setup0));
    lp.release_tables();
    return 0;
}

// sumcheck.Prove for a linear gate with one evaluation point; same two phases as sumcheck_cipher_fast (the local rounds
// of this rank's shard with the sums added over the ranks, then, when sharded, the gathered per-rank entries and the
// last gamma rounds on every rank).  No Eq table is built or folded.
int sumcheck_linear_fast(const GateDesc& g, const E& ark, int bN, const DevTable* const* X, const E* q, E* proof, E* challenges,
                         E* final_claims, const E* trusted_claim, bool track_claim) {
    const ShardView shard = shard_view();
    const int gamma = shard.gamma, m1 = bN - gamma, arity = g.n_in;
    if (m1 < 0) return fail("bN %d is smaller than log2(world) %d", bN, gamma);
    E c = hfr::ONE, tail[2 * GKR_MAX_ARITY], r_last, v[GKR_MAX_ARITY];
    bool did_gamma = false;
    E claim = trusted_claim ? *trusted_claim : hfr::ZERO;
    bool claim_known = trusted_claim != nullptr;
    E* claim_p = track_claim ? &claim : nullptr;
    if (m1 >= 1) {
        const E seed = gamma ? shard_seed(q + m1, gamma, shard.rank) : hfr::ONE;
        CHK(rounds_with_retry(c, claim, claim_known, [&]() {
            return linear_rounds(g, ark, m1, X, q, seed, gamma > 0 || cx().force_collective, c, proof, challenges, tail, r_last, claim_p,
                                 &claim_known, gamma, &did_gamma);
        }));
        for (int t = 0; t < arity; t++) v[t] = fold2(tail[2 * t], tail[2 * t + 1], r_last);
    } else {
        CHK(gather0(X, arity, v));
    }
    if (gamma > 0 && !did_gamma) {
        std::vector<E> all;
        CHK(coll_allgather(v, arity, all));
        if (cx().host_tail > 0) {
            std::vector<std::vector<E>> Th(arity, std::vector<E>(shard.world));
            for (int t = 0; t < arity; t++)
                for (int r = 0; r < shard.world; r++) Th[t][r] = all[(size_t)r * arity + t];
            host_linear_rounds(g, ark, gamma, Th, q + m1, hfr::ONE, c, proof + (size_t)3 * m1, challenges + m1, claim_p, &claim_known);
            for (int t = 0; t < arity; t++) v[t] = Th[t][0];
        } else {
            ScopedTable x2[GKR_MAX_ARITY];
            const DevTable* X2[GKR_MAX_ARITY];
            for (int t = 0; t < arity; t++) {
                std::vector<E> col(shard.world);
                for (int r = 0; r < shard.world; r++) col[r] = all[(size_t)r * arity + t];
                CHK(small_table(&x2[t], col));
                X2[t] = &x2[t];
            }
            CHK(linear_rounds(g, ark, gamma, X2, q + m1, hfr::ONE, false, c, proof + (size_t)3 * m1, challenges + m1, tail, r_last,
                              claim_p, &claim_known));
            for (int t = 0; t < arity; t++) v[t] = fold2(tail[2 * t], tail[2 * t + 1], r_last);
        }
    }
    final_claims[0] = c;
    for (int t = 0; t < arity; t++) final_claims[1 + t] = v[t];
    return 0;
}

// The reference-shaped rounds (sumcheck/prover.go:70-76) over an Eq table and `arity` tables of 2^m entries:
// partial evaluation at t = 0..deg+1, interpolation, Fiat-Shamir, fold.  eq is folded in place, X is
// read-only (round 0 folds into scratch).  On return `last` = [Eq[0], X_1[0], ...] of this rank.
int generic_rounds(const GateDesc& g, const E& ark, int m, DevTable* eq, const DevTable* const* X, bool collective,
                   E* proof, E* chal, E* last) {
    const size_t n = (size_t)1 << m;
    const int nev = g.power + 2, arity = g.n_in;
    if (cx().racc_dirty) {   // see cipher_rounds
        HIPCHK(hipMemsetAsync(cx().d_racc, 0, sizeof(unsigned long long) * kRaccWords * GKR_RACC_SLOTS, cx().stream));
        HIPCHK(hipMemsetAsync(cx().d_counter, 0, sizeof(unsigned int), cx().stream));
    }
    cx().racc_dirty = true;
    ScopedTable scratch[GKR_MAX_ARITY];
    for (int k = 0; k < arity; k++) CHK(table_alloc(&scratch[k], std::max<size_t>(n / 2, 1)));
    const DevTable* cur[GKR_MAX_ARITY + 1];
    for (int k = 0; k < arity; k++) cur[k] = X[k];
    for (int k = 0; k < m; k++) {
        const size_t mid = n >> (k + 1);
        E evals[GKR_MAX_EVALS];
        CHK(partial_evals(g, eq, cur, mid, ark, evals, nev, collective));
        if (test_fire_corrupt(k)) evals[1].l[0] ^= 1ull;      // (after the exchange: set the hook on every rank)
        E* coeffs = proof + (size_t)k * nev;
        cx().lag->interpolate(coeffs, evals, nev);
        const E r = hfr::mimc_hash(coeffs, (size_t)nev);
        chal[k] = r;
        const DevTable* src[GKR_MAX_ARITY + 1];
        const DevTable* dst[GKR_MAX_ARITY + 1];
        src[0] = eq;
        dst[0] = eq;
        for (int t = 0; t < arity; t++) {
            src[1 + t] = cur[t];
            dst[1 + t] = &scratch[t];
        }
        CHK(launch_fold(src, dst, arity + 1, mid, r));
        for (int t = 0; t < arity; t++) cur[t] = &scratch[t];
    }
    const DevTable* all[GKR_MAX_ARITY + 1];
    all[0] = eq;
    for (int t = 0; t < arity; t++) all[1 + t] = cur[t];
    CHK(gather0(all, arity + 1, last));   // finalClaims (prover.go:79-86)
    cx().racc_dirty = false;
    for (int k = 0; k < arity; k++) table_release(&scratch[k]);
    return 0;
}

// sumcheck.Prove on device-resident tables (sumcheck/prover.go:46-90).  bN = GLOBAL number of variables;
// X = this rank's shards (read-only).  proof: bN*(deg+2), challenges: bN, final: arity+1.
// trust_claims: the caller guarantees that `claims` are the true sums (gkr.Prove: every claim is a previous
// sumcheck's output).  The single-point cipher path then derives one monomial sum per round from the running
// claim instead of computing it.  Entry points that take claims from outside never set it: for them the
// output must be the reference's whatever the claims are (they only feed Fiat-Shamir there).
int sumcheck_prove_once(int gate, const E& ark_in, int arity, int bN, const DevTable* const* X, const E* qprimes, int nq,
                        const E* claims, int nclaims, E* proof, E* challenges, E* final_claims, bool trust_claims, E* rho_out) {
    GateDesc g;
    CHK(gate_resolve(gate, arity, &g));
    const E ark = gate == GKRHIP_GATE_IDENTITY ? hfr::ZERO : ark_in;   // IdentityGate has no Ark
    if (nq < 1) return fail("need at least one evaluation point");
    if (nclaims != nq && nq > 1)  // sumcheck/prover.go:113-115
        return fail("provided a multi-instance %d but the number of claims does not match %d", nq, nclaims);
    const ShardView shard = shard_view();
    const int gamma = shard.gamma, m1 = bN - gamma;
    if (m1 < 0) return fail("bN %d is smaller than log2(world) %d", bN, gamma);
    const int nev = g.power + 2;

    // ---- makeEqTable (prover.go:102-144)
    std::vector<E> seeds(nq, hfr::ONE);
    int nq_used = 1;
    if (nclaims >= 1) {
        LAP("once: gate_resolve etc");
        // (the reference draws it for a single claim too, sumcheck/prover.go:128, and never uses it: nothing observable depends on
        // it, and it is 364 serial products per layer on the critical path)
        const E rho = (nclaims > 1 || nq > 1) ? hfr::mimc_hash(claims, (size_t)nclaims) : hfr::ZERO;
        LAP("once: rho hash");
        *rho_out = rho;
        E mlt = rho;
        for (int j = 1; j < nq; j++) {
            seeds[j] = mlt;
            mlt = hfr::mul(mlt, rho);
        }
        nq_used = nq;
    }
    const E* trusted = (trust_claims && nclaims == 1) ? &claims[0] : nullptr;
    if (gate_is_cipher2(g) && nq_used == 1 && bN >= 1 && !cx().force_generic)
        return sumcheck_cipher_fast(ark, bN, X[0], X[1], qprimes, proof, challenges, final_claims, trusted,
                                    trust_claims && cx().claim_trick);
    // single-point layers of a linear gate (identity, sums of up to four inputs): fused rounds, no Eq table
    if (g.power == 1 && nq_used == 1 && bN >= 1 && !cx().force_generic)
        return sumcheck_linear_fast(g, ark, bN, X, qprimes, proof, challenges, final_claims, trusted,
                                    trust_claims && cx().claim_trick);

    // phase 1: this rank's shard; Eq_local = sum_j seed_j * eq(q_j tail, rank) * eq(q_j[0:m1], .)
    if (gamma > 0)
        for (int j = 0; j < nq_used; j++) seeds[j] = hfr::mul(seeds[j], shard_seed(qprimes + (size_t)j * bN + m1, gamma, shard.rank));
    ScopedTable eq;
    CHK(table_alloc(&eq, (size_t)1 << m1));
    CHK(build_eq(&eq, qprimes, nq_used, bN, m1, seeds.data()));
    E last[GKR_MAX_ARITY + 1];
    CHK(generic_rounds(g, ark, m1, &eq, X, gamma > 0 || cx().force_collective, proof, challenges, last));
    table_release(&eq);
    if (gamma > 0) {
        // phase 2: one entry per table per rank -> tables over the gamma shard bits, same rounds on every rank
        std::vector<E> all;
        CHK(coll_allgather(last, arity + 1, all));
        std::vector<std::vector<E>> cols(arity + 1, std::vector<E>(shard.world));
        for (int r = 0; r < shard.world; r++)
            for (int t = 0; t <= arity; t++) cols[t][r] = all[(size_t)r * (arity + 1) + t];
        ScopedTable eq2, x2[GKR_MAX_ARITY];
        const DevTable* X2[GKR_MAX_ARITY];
        CHK(small_table(&eq2, cols[0]));
        for (int t = 0; t < arity; t++) {
            CHK(small_table(&x2[t], cols[1 + t]));
            X2[t] = &x2[t];
        }
        CHK(generic_rounds(g, ark, gamma, &eq2, X2, false, proof + (size_t)nev * m1, challenges + m1, last));
        table_release(&eq2);
        for (int t = 0; t < arity; t++) table_release(&x2[t]);
    }
    for (int t = 0; t <= arity; t++) final_claims[t] = last[t];
    return 0;
}

// ---- the prover checks what it is about to return ---------------------------------------------------------------------
// The verifier's side of one sumcheck on the prover's own output: sumcheck.Verify's round checks P_i(0) + P_i(1) == expected
// (sumcheck/verifier.go:28-55) with the challenges the prover already hashed -- no second hash --, then testSumcheck's closing
// identity Gate.Eval(finalClaims[1:]) * sum_j rho^j EvalEq(q_j, r) == P_last(r_last) (gkr/verifier.go:93-114,
// poly/eq.go:19-32), and finalClaims[0] against that eq value.  bN * (deg + 9) products, 2 * bN more per point: microseconds.
// Why it is not redundant: inside gkr.Prove the fused round loops DERIVE one sum of every round from the running claim
// (trust_claims), so a wrong device sum -- or a wrong fold, or incomplete look-ahead products -- yields rounds that are
// consistent with each other and a transcript that is simply wrong.  By the soundness of the sumcheck itself such a run
// closes with probability ~ bN * deg / q.  Returns 0 = closes, 1 + i = round i, -1 = the closing identity, -2 = finalClaims[0].
// `claims_are_sums`: the claims are known to be the sums (gkr.Prove); claims from outside only feed Fiat-Shamir
// (sumcheck/prover.go:128) and say nothing about round 0.
int sumcheck_closes(const GateDesc& g, const E& ark, int bN, const E* qprimes, int nq_used, const E* claims, int nclaims,
                    bool claims_are_sums, const E& rho, const E* proof, const E* chal, const E* fin) {
    const int nev = g.power + 2;
    bool have = false;
    E expected = hfr::ZERO;
    if (claims_are_sums && nclaims >= 1) {
        expected = nclaims == 1 ? claims[0] : hfr::eval_univariate(claims, nclaims, rho);      // recombineMultiClaims, verifier.go:58-65
        have = true;
    }
    for (int i = 0; i < bN; i++) {
        const E* p = proof + (size_t)i * nev;
        if (have) {
            E s01 = hfr::add(p[0], p[0]);            // P(0) + P(1) = 2 c_0 + c_1 + ... + c_deg+1
            for (int j = 1; j < nev; j++) s01 = hfr::add(s01, p[j]);
            if (s01 != expected) return 1 + i;
        }
        expected = hfr::eval_univariate(p, nev, chal[i]);
        have = true;
    }
    if (!have) return 0;                             // no round and no claim to hold the final values against
    E eq_eval = hfr::ZERO, w = hfr::ONE;
    for (int j = 0; j < nq_used; j++) {
        eq_eval = hfr::add(eq_eval, hfr::mul(w, hfr::eval_eq(qprimes + (size_t)j * bN, chal, bN)));
        w = hfr::mul(w, rho);
    }
    if (hfr::mul(gate_eval_host(g, ark, fin + 1), eq_eval) != expected) return -1;
    if (fin[0] != eq_eval) return -2;
    return 0;
}

// sumcheck.Prove, checked: a sumcheck that does not close is run ONCE more in safe mode -- nothing queued ahead of its
// challenge, no look-ahead products, every sum of every round computed (no claim trick) -- and checked again; only a second
// failure is an error.  The retry rests on the invariant of rounds_with_retry (the rounds read the layer's tables and write
// scratch), produces the same transcript as any other path, and is deterministic in data every rank of a sharded proof
// holds identically (the exchanged sums, the gathered final values): the ranks reach the same verdict without a vote.
int sumcheck_prove_dev(int gate, const E& ark_in, int arity, int bN, const DevTable* const* X, const E* qprimes, int nq,
                       const E* claims, int nclaims, E* proof, E* challenges, E* final_claims, bool trust_claims = false) {
    E rho = hfr::ZERO;
    CHK(sumcheck_prove_once(gate, ark_in, arity, bN, X, qprimes, nq, claims, nclaims, proof, challenges, final_claims, trust_claims, &rho));
    if (!g_layer_check.load(std::memory_order_relaxed)) return 0;
    GateDesc g;
    CHK(gate_resolve(gate, arity, &g));
    const E ark = gate == GKRHIP_GATE_IDENTITY ? hfr::ZERO : ark_in;
    const int nq_used = nclaims >= 1 ? nq : 1;
    g_cnt_layer_checks.fetch_add(1, std::memory_order_relaxed);
    LAP("checked: before the check");
    int bad = sumcheck_closes(g, ark, bN, qprimes, nq_used, claims, nclaims, trust_claims, rho, proof, challenges, final_claims);
    LAP("checked: sumcheck_closes");
    if (!bad) return 0;
    g_cnt_layer_check_failures.fetch_add(1, std::memory_order_relaxed);
    if (getenv("GKRHIP_TRACE")) fprintf(stderr, "sumcheck (gate %s, bN %d) does not close (%d): running it again in safe mode\n", g.id.c_str(), bN, bad);
    struct SafeScope {
        const bool prev = g_safe_mode;
        SafeScope() { g_safe_mode = true; }
        ~SafeScope() { g_safe_mode = prev; }
    } safe;
    CHK(sumcheck_prove_once(gate, ark_in, arity, bN, X, qprimes, nq, claims, nclaims, proof, challenges, final_claims, false, &rho));
    bad = sumcheck_closes(g, ark, bN, qprimes, nq_used, claims, nclaims, trust_claims, rho, proof, challenges, final_claims);
    if (bad)
        return fail("sumcheck.Prove (gate %s, bN %d): the prover's own check failed twice (%s), the second time in safe mode: "
                    "no proof is returned", g.id.c_str(), bN,
                    bad > 0 ? "a round's P(0) + P(1) is not the running claim" : bad == -1 ? "the final claims do not close the last round" : "finalClaims[0] is not the eq value");
    return 0;
}

template <int POWER, int ARITY>
void launch_gate_eval(const AssignArgs& a) {
    hipLaunchKernelGGL((k_gate_eval_batch<POWER, ARITY>), dim3(grid_for(a.n, cx().max_grid)), dim3(GKR_BLOCK), 0, cx().stream, a);
}
// (off, n): the slice of the tables the launch covers -- every layer of Circuit.Assign is element-wise (circuit/circuit.go:48-64)
int gate_eval_dev(int gate, const E& ark, const DevTable* const* in, int arity, const DevTable* out, size_t n, size_t off = 0) {
    GateDesc g;
    CHK(gate_resolve(gate, arity, &g));
    AssignArgs a;
    memset(&a, 0, sizeof a);
    for (int k = 0; k < arity; k++) a.in[k] = CPlanes{in[k]->base + off, in[k]->base + in[k]->cap + off};
    a.out = Planes{out->base + off, out->base + out->cap + off};
    a.arity = arity;
    a.mask = g.mask;
    a.n = n;
    a.ark = to_dev(gate == GKRHIP_GATE_IDENTITY ? hfr::ZERO : ark);
    switch (g.power * 10 + g.n_in) {
        case 11: launch_gate_eval<1, 1>(a); break;
        case 12: launch_gate_eval<1, 2>(a); break;
        case 13: launch_gate_eval<1, 3>(a); break;
        case 14: launch_gate_eval<1, 4>(a); break;
        case 71: launch_gate_eval<7, 1>(a); break;
        case 72: launch_gate_eval<7, 2>(a); break;
        case 73: launch_gate_eval<7, 3>(a); break;
        case 74: launch_gate_eval<7, 4>(a); break;
        default: return fail("unsupported gate shape (power %d, %d inputs)", g.power, g.n_in);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// MultiLin.Evaluate (poly/multilin.go:59-66) of a table of 2^nc entries: fold chain into scratch.  The
// table is this rank's shard of 2^(nc-gamma) entries; the shard values are all-gathered and the last gamma
// coordinates are applied to the gathered (<= world-entry) table.
int evaluate_dev(const DevTable* t, int nc, const E* coords, E* out) {
    const int gamma = shard_view().gamma, m1 = nc - gamma;
    if (m1 < 0) return fail("Evaluate: %d coordinates for a table sharded over 2^%d ranks", nc, gamma);
    const size_t n = (size_t)1 << m1;
    ScopedTable s;
    CHK(table_alloc(&s, std::max<size_t>(n / 2, 1)));
    const DevTable* cur = t;
    for (int k = 0; k < m1; k++) {
        const size_t mid = n >> (k + 1);
        const DevTable* src[1] = {cur};
        const DevTable* dst[1] = {&s};
        CHK(launch_fold(src, dst, 1, mid, coords[k]));
        cur = &s;
    }
    E v;
    CHK(gather0(&cur, 1, &v));
    table_release(&s);
    if (gamma > 0) {
        std::vector<E> all;
        CHK(coll_allgather(&v, 1, all));
        for (int k = 0; k < gamma; k++) {           // <= world scalar folds
            const size_t mid = all.size() / 2;
            for (size_t i = 0; i < mid; i++) all[i] = fold2(all[i], all[i + mid], coords[m1 + k]);
            all.resize(mid);
        }
        v = all[0];
    }
    *out = v;
    return 0;
}

// synthetic inputs: element j = Montgomery(((i*i) mod 2^64) ^ 0xf45c9df123f), i = j*stride + offset
__global__ void __launch_bounds__(GKR_BLOCK) k_random_fr_array(Planes out, size_t n, unsigned long long stride,
                                                               unsigned long long offset) {
    const Fr r2 = {{0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u}};
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long i = (unsigned long long)j * stride + offset;
        const unsigned long long v = (i * i) ^ 0xf45c9df123fULL;
        Fr x = fr_zero();
        x.v[0] = (u32)v;
        x.v[1] = (u32)(v >> 32);
        st_fr(out.lo, out.hi, j, fr_mul(x, r2));
    }
}
// table[i] = Montgomery(i)  (BenchmarkFolding's table, poly/multilin_test.go:60-63)
__global__ void __launch_bounds__(GKR_BLOCK) k_iota(Planes out, size_t n) {
    const Fr r2 = {{0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u}};
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
        Fr x = fr_zero();
        x.v[0] = (u32)j;
        x.v[1] = (u32)((unsigned long long)j >> 32);
        st_fr(out.lo, out.hi, j, fr_mul(x, r2));
    }
}
