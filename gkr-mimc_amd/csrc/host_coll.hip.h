// host_coll.hip.h -- the per-round exchange of the sharded prover: RCCL (dlopen'd) over xGMI, or a POSIX
// shared-memory transport with the same call sites (several ranks on one GPU, tests).  Included by gkrhip.hip
// inside its anonymous namespace.
#pragma once
// ---- collective over the ranks of one node (RCCL over xGMI), loaded lazily -----------------------------
// The only exchange of the path: an exact integer sum of limb-split lanes (u64), a handful of words per
// round.  world == 1: no-ops.  RCCL is dlopen()ed on gkrhip_comm_init so that single-GPU use neither
// links nor loads it.
struct Coll {
    int world = 1, rank = 0, gamma = 0;
    void* dl = nullptr;
    decltype(&ncclGetUniqueId) p_get_id = nullptr;
    decltype(&ncclCommInitRank) p_init = nullptr;
    decltype(&ncclAllReduce) p_allreduce = nullptr;
    decltype(&ncclCommDestroy) p_destroy = nullptr;
    decltype(&ncclGetErrorString) p_errstr = nullptr;
    // per-lane: LaneColl (RCCL communicator, or the host shared-memory transport used by processes of one
    // node without RCCL, e.cx(). several ranks time-sharing one GPU in the tests)
    std::vector<Ctx*> lanes;               // the lanes that carry a communicator (lane 0 = default lane)
    size_t next_lane = 0;
};
Coll gc;
const size_t kShmSlotWords = 8192;

void shm_barrier() {
    const unsigned gen = cx().lc.shm->gen.load(std::memory_order_acquire);
    if (cx().lc.shm->arrive.fetch_add(1, std::memory_order_acq_rel) == (unsigned)gc.world - 1) {
        cx().lc.shm->arrive.store(0, std::memory_order_relaxed);
        cx().lc.shm->gen.fetch_add(1, std::memory_order_release);
    } else {
        Waiter w;
        while (cx().lc.shm->gen.load(std::memory_order_acquire) == gen) w.step();
    }
}

int coll_load() {
    if (gc.dl) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        gc.dl = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (gc.dl) break;
    }
    if (!gc.dl) return fail("cannot load RCCL (librccl.so.1): %s", dlerror());
    gc.p_get_id = (decltype(gc.p_get_id))dlsym(gc.dl, "ncclGetUniqueId");
    gc.p_init = (decltype(gc.p_init))dlsym(gc.dl, "ncclCommInitRank");
    gc.p_allreduce = (decltype(gc.p_allreduce))dlsym(gc.dl, "ncclAllReduce");
    gc.p_destroy = (decltype(gc.p_destroy))dlsym(gc.dl, "ncclCommDestroy");
    gc.p_errstr = (decltype(gc.p_errstr))dlsym(gc.dl, "ncclGetErrorString");
    if (!gc.p_get_id || !gc.p_init || !gc.p_allreduce || !gc.p_destroy || !gc.p_errstr)
        return fail("RCCL library lacks a required symbol");
    return 0;
}
int coll_buffers(size_t words) {
    if (words <= cx().lc.buf_words) return 0;
    if (cx().lc.d_buf) (void)hipFree(cx().lc.d_buf);
    if (cx().lc.h_buf) (void)hipHostFree(cx().lc.h_buf);
    HIPCHK(hipMalloc(&cx().lc.d_buf, sizeof(unsigned long long) * words));
    HIPCHK(hipHostMalloc(&cx().lc.h_buf, sizeof(unsigned long long) * words, hipHostMallocDefault));
    cx().lc.buf_words = words;
    return 0;
}
#define NCCLCHK(x)                                                                             \
    do {                                                                                       \
        ncclResult_t _r = (x);                                                                 \
        if (_r != ncclSuccess) return fail("%s failed: %s", #x, gc.p_errstr ? gc.p_errstr(_r) : "?"); \
    } while (0)

// in-place sum over ranks of n u64 lanes in device memory, on the library's stream
int coll_allreduce(unsigned long long* d, int n) {
    if (cx().lc.comm) {
        NCCLCHK(gc.p_allreduce(d, d, (size_t)n, ncclUint64, ncclSum, cx().lc.comm, cx().stream));
        return 0;
    }
    if (cx().lc.shm) {
        if ((size_t)n > kShmSlotWords) return fail("shm all-reduce of %d words exceeds the slot", n);
        if (!cx().lc.h_tmp) HIPCHK(hipHostMalloc(&cx().lc.h_tmp, sizeof(unsigned long long) * kShmSlotWords, hipHostMallocDefault));
        HIPCHK(hipMemcpyAsync(cx().lc.h_tmp, d, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost, cx().stream));
        HIPCHK(hipStreamSynchronize(cx().stream));
        memcpy(cx().lc.shm_slots + (size_t)gc.rank * kShmSlotWords, cx().lc.h_tmp, sizeof(unsigned long long) * n);
        shm_barrier();
        for (int i = 0; i < n; i++) {
            unsigned long long s = 0;
            for (int r = 0; r < gc.world; r++) s += cx().lc.shm_slots[(size_t)r * kShmSlotWords + i];
            cx().lc.h_tmp[i] = s;
        }
        shm_barrier();
        HIPCHK(hipMemcpyAsync(d, cx().lc.h_tmp, sizeof(unsigned long long) * n, hipMemcpyHostToDevice, cx().stream));
        HIPCHK(hipStreamSynchronize(cx().stream));
        return 0;
    }
    return 0;
}
// The same sum over ranks for words that are already on the host (the round kernel's host-mapped hand-off):
// slot write, barrier, sum, barrier.  No device round trip at all.
int shm_allreduce_host(unsigned long long* words, int n) {
    if ((size_t)n > kShmSlotWords) return fail("shm all-reduce of %d words exceeds the slot", n);
    memcpy(cx().lc.shm_slots + (size_t)gc.rank * kShmSlotWords, words, sizeof(unsigned long long) * n);
    shm_barrier();
    for (int i = 0; i < n; i++) {
        unsigned long long s = 0;
        for (int r = 0; r < gc.world; r++) s += cx().lc.shm_slots[(size_t)r * kShmSlotWords + i];
        words[i] = s;
    }
    shm_barrier();
    return 0;
}
// all-gather of `cnt` field elements per rank (host values): rank g's elements land in out[g*cnt ..].
// Implemented as an all-reduce of a zero-padded buffer (one contributor per slot: the sum is exact).
int coll_allgather(const E* mine, int cnt, std::vector<E>& out) {
    out.assign((size_t)gc.world * cnt, hfr::ZERO);
    if (gc.world == 1 && !cx().force_collective) {
        for (int i = 0; i < cnt; i++) out[i] = mine[i];
        return 0;
    }
    const size_t words = (size_t)gc.world * cnt * 4;
    CHK(coll_buffers(std::max<size_t>(words, 256)));
    memset(cx().lc.h_buf, 0, words * 8);
    memcpy(cx().lc.h_buf + (size_t)gc.rank * cnt * 4, mine, (size_t)cnt * 32);
    HIPCHK(hipMemcpyAsync(cx().lc.d_buf, cx().lc.h_buf, words * 8, hipMemcpyHostToDevice, cx().stream));
    CHK(coll_allreduce(cx().lc.d_buf, (int)words));
    HIPCHK(hipMemcpyAsync(cx().lc.h_buf, cx().lc.d_buf, words * 8, hipMemcpyDeviceToHost, cx().stream));
    HIPCHK(hipStreamSynchronize(cx().stream));
    memcpy(out.data(), cx().lc.h_buf, words * 8);
    return 0;
}

// eq(q_tail, bits(rank)) with the table convention (q_tail[0] <-> most significant bit): the weight of
// shard `rank` when the hypercube is sharded on its gamma lowest index bits (poly/eq.go:74-88 uses the same
// factorisation for chunks).
E shard_seed(const E* q_tail, int gamma, int rank) {
    E r = hfr::ONE;
    for (int i = 0; i < gamma; i++) {
        const bool bit = (rank >> (gamma - 1 - i)) & 1;
        r = hfr::mul(r, bit ? q_tail[i] : hfr::sub(hfr::ONE, q_tail[i]));
    }
    return r;
}

inline E limbs9_to_fr(const unsigned long long* w) {
    hfr::u64 lanes[8];
    for (int j = 0; j < 8; j++) lanes[j] = w[j];
    const E lo = hfr::reduce_limbsplit(lanes);
    const E hv = {{w[8], 0, 0, 0}};                  // w[8] * 2^256 mod q
    return hfr::add(lo, hfr::mul(hv, hfr::R2));
}

// gather element 0 of up to 5 device tables to the host
int gather0(const DevTable* const* t, int ntab, E* out) {
    Gather0Args ga;
    memset(&ga, 0, sizeof ga);
    for (int i = 0; i < ntab; i++) ga.t[i] = t[i]->cplanes();
    ga.ntab = ntab;
    ga.out = cx().d_small;
    hipLaunchKernelGGL(k_gather0, dim3(1), dim3(64), 0, cx().stream, ga);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(cx().h_small, cx().d_small, 32 * ntab, hipMemcpyDeviceToHost, cx().stream));
    HIPCHK(hipStreamSynchronize(cx().stream));
    memcpy(out, cx().h_small, 32 * ntab);
    return 0;
}

// evals[t] (t < nev) for the current round.  Launches the partial evaluation, the block reduction,
// (all-reduces the limb-split sums across ranks,) copies them to the host and reduces them mod q.
int partial_evals(int gate, int arity, const DevTable* eq, const DevTable* const* x, size_t mid, const E& ark, E* evals,
                  int nev, bool collective) {
    int nblocks = 0;
    const bool direct = !collective;      // un-sharded: the kernel hands the sums to the host itself
    const bool timed = 2 * mid >= cx().prof.min_n;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timed) {
        e0 = prof_event();
        e1 = prof_event();
        HIPCHK(hipEventRecord(e0, cx().stream));
    }
    if (gate == GKRHIP_GATE_CIPHER && arity == 2) {
        CHK((launch_partial_eval_t<GKR_GATE_CIPHER, 2, 9>(eq, x, mid, ark, &nblocks, direct)));
    } else if (gate == GKRHIP_GATE_IDENTITY && arity == 1) {
        CHK((launch_partial_eval_t<GKR_GATE_IDENTITY, 1, 3>(eq, x, mid, ark, &nblocks, direct)));
    } else if (gate == GKRHIP_GATE_IDENTITY && arity == 2) {
        CHK((launch_partial_eval_t<GKR_GATE_IDENTITY, 2, 3>(eq, x, mid, ark, &nblocks, direct)));
    } else if (gate == GKRHIP_GATE_ADD && arity == 2) {
        CHK((launch_partial_eval_t<GKR_GATE_ADD, 2, 3>(eq, x, mid, ark, &nblocks, direct)));
    } else {
        return fail("unsupported gate/arity combination (gate %d, arity %d)", gate, arity);
    }
    HIPCHK(hipGetLastError());
    if (timed) {
        HIPCHK(hipEventRecord(e1, cx().stream));
        cx().prof.peval_ev.emplace_back(e0, e1);
        cx().prof.peval_launches++;
        cx().prof.peval_modmuls += (gate == GKRHIP_GATE_CIPHER ? 45.0 : 3.0) * (double)mid;
    }
    const int nwords = nev * GKR_ACC_WORDS;
    if (direct) {
        CHK(wait_flag(cx().seq));
        for (int t = 0; t < nev; t++) evals[t] = limbs9_to_fr(cx().h_round + (size_t)t * GKR_ACC_WORDS);
        return 0;
    }
    hipLaunchKernelGGL(k_reduce_partials, dim3(nwords), dim3(GKR_BLOCK), 0, cx().stream, cx().d_partials, cx().d_sums, nblocks, nwords);
    HIPCHK(hipGetLastError());
    if (collective) CHK(coll_allreduce(cx().d_sums, nwords));
    HIPCHK(hipMemcpyAsync(cx().h_sums, cx().d_sums, sizeof(unsigned long long) * nwords, hipMemcpyDeviceToHost, cx().stream));
    HIPCHK(hipStreamSynchronize(cx().stream));
    for (int t = 0; t < nev; t++) evals[t] = limbs9_to_fr(cx().h_sums + (size_t)t * GKR_ACC_WORDS);
    return 0;
}

int stage_coords(const E* coords, size_t n) {
    if (n > cx().d_q_cap) {
        if (cx().d_q) HIPCHK(hipFree(cx().d_q));
        cx().d_q_cap = std::max<size_t>(n, 256);
        HIPCHK(hipMalloc(&cx().d_q, sizeof(Fr) * cx().d_q_cap));
    }
    if (n == 0) return 0;
    std::vector<Fr> stage(n);
    for (size_t i = 0; i < n; i++) stage[i] = to_dev(coords[i]);
    HIPCHK(hipMemcpyAsync(cx().d_q, stage.data(), sizeof(Fr) * n, hipMemcpyHostToDevice, cx().stream));
    HIPCHK(hipStreamSynchronize(cx().stream));  // `stage` is pageable host memory
    return 0;
}

// Build Eq = sum_j seed_j * eq(q_j[0:m], .) over 2^m entries (poly/eq.go:41-59 + sumcheck/prover.go:102-144).
// qprimes holds nq points of q_stride coordinates each; only the first m coordinates of each are used.
int build_eq(DevTable* eq, const E* qprimes, int nq, int q_stride, int m, const E* seeds) {
    const size_t n = (size_t)1 << m;
    const int nhi = m / 2, nlo = m - nhi;
    const size_t shi = (size_t)1 << nhi, slo = (size_t)1 << nlo;
    // stage coordinates + seeds (+ the constant one for the lo tables)
    const size_t ncoord = (size_t)nq * q_stride;
    std::vector<E> stage(ncoord + 2 * (size_t)nq);
    for (size_t i = 0; i < ncoord; i++) stage[i] = qprimes[i];
    for (int j = 0; j < nq; j++) {
        stage[ncoord + j] = seeds[j];
        stage[ncoord + nq + j] = hfr::ONE;
    }
    CHK(stage_coords(stage.data(), stage.size()));

    DevTable thi, tlo;
    CHK(table_alloc(&thi, shi * nq));
    CHK(table_alloc(&tlo, slo * nq));
    EqSmallArgs s;
    s.q = cx().d_q;
    s.q_stride = q_stride;
    s.out = thi.planes();
    s.seeds = cx().d_q + ncoord;
    s.nbits = nhi;
    s.q_off = 0;
    s.tab_stride = shi;
    hipLaunchKernelGGL(k_eq_small, dim3(nq), dim3(1024), 0, cx().stream, s);
    s.out = tlo.planes();
    s.seeds = cx().d_q + ncoord + nq;
    s.nbits = nlo;
    s.q_off = nhi;
    s.tab_stride = slo;
    hipLaunchKernelGGL(k_eq_small, dim3(nq), dim3(1024), 0, cx().stream, s);
    EqExpandArgs x;
    x.out = eq->planes();
    x.thi = thi.cplanes();
    x.tlo = tlo.cplanes();
    x.hi_stride = shi;
    x.lo_stride = slo;
    x.nclaims = nq;
    x.nlo = nlo;
    x.n = n;
    hipLaunchKernelGGL(k_eq_expand, dim3(grid_for(n, cx().max_grid)), dim3(GKR_BLOCK), 0, cx().stream, x);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(cx().stream));
    table_release(&thi);
    table_release(&tlo);
    return 0;
}
