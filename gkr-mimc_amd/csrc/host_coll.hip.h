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
    decltype(&ncclAllGather) p_allgather = nullptr;
    decltype(&ncclCommDestroy) p_destroy = nullptr;
    decltype(&ncclGetErrorString) p_errstr = nullptr;
    // per-lane: LaneColl (RCCL communicator, or the host shared-memory transport used by processes of one
    // node without RCCL, e.cx(). several ranks time-sharing one GPU in the tests)
    std::vector<Ctx*> lanes;               // the lanes that carry a communicator (lane 0 = default lane)
    size_t next_lane = 0;
};
Coll gc;
const size_t kShmSlotWords = 8192;

// The shard a call works on.  Sessions shard over the installed communicator; the host-buffer entry points (Fold,
// Evaluate, FoldedEqTable, EvalBatch, sumcheck.Prove on host tables, the one-shot verifier) always run un-sharded
// on this process's GPU.  They say so with a LocalOnly scope on their own thread -- the process-wide communicator
// state is never rewritten, so a sharded proof in flight on another lane (another thread) is not disturbed.
struct ShardView {
    int world, rank, gamma;
};
thread_local int g_local_only = 0;
struct LocalOnly {
    LocalOnly() { g_local_only++; }
    ~LocalOnly() { g_local_only--; }
};
inline ShardView shard_view() {
    if (g_local_only) return ShardView{1, 0, 0};
    return ShardView{gc.world, gc.rank, gc.gamma};
}

// How long a rank waits for its peers (barrier of the shared-memory transport, completion of an RCCL all-reduce)
// before it gives up with an error: GKRHIP_COLL_TIMEOUT_S, default 300 s.  Ranks reach the first exchange of a proof
// at different times (seconds apart when one of them was still assigning), but never minutes.
inline double coll_timeout_ms() {
    static const double t = [] {
        const char* e = getenv("GKRHIP_COLL_TIMEOUT_S");
        const double s = e ? atof(e) : 300.0;
        return (s > 0 ? s : 300.0) * 1e3;
    }();
    return t;
}

// Barrier over the ranks of the lane's shared-memory segment.  Fails (instead of waiting for ever) when a peer has
// raised the segment's abort word -- every error return of a sharded call and gkrhip_comm_destroy raise it -- or when
// the peers do not arrive within the deadline; the rank that times out raises the abort word itself.
int shm_barrier() {
    ShmHdr* h = cx().lc.shm;
    const unsigned world = (unsigned)shard_view().world;
    const unsigned gen = h->gen.load(std::memory_order_acquire);
    if (h->abort.load(std::memory_order_acquire)) return fail("sharded prover: a peer rank failed or left (abort word set)");
    if (h->arrive.fetch_add(1, std::memory_order_acq_rel) == world - 1) {
        h->arrive.store(0, std::memory_order_relaxed);
        h->gen.fetch_add(1, std::memory_order_release);
        return 0;
    }
    Waiter w;
    unsigned long spins = 0;
    double t0 = 0;
    while (h->gen.load(std::memory_order_acquire) == gen) {
        w.step();
        if ((++spins & 0xffff) != 0) continue;
        if (h->abort.load(std::memory_order_acquire)) {
            // a peer that leaves after the barrier completed raises the word too: the generation decides
            if (h->gen.load(std::memory_order_acquire) != gen) break;
            return fail("sharded prover: a peer rank failed or left (abort word set)");
        }
        if (t0 == 0) t0 = now_ms();
        else if (now_ms() - t0 > coll_timeout_ms()) {
            h->abort.store(1, std::memory_order_release);
            return fail("sharded prover: timed out after %.0f s waiting for the other ranks", coll_timeout_ms() * 1e-3);
        }
    }
    return 0;
}
// raise the abort word of the current lane's segment (error paths of sharded calls)
inline void shm_abort() {
    if (cx().lc.shm) cx().lc.shm->abort.store(1, std::memory_order_release);
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
    gc.p_allgather = (decltype(gc.p_allgather))dlsym(gc.dl, "ncclAllGather");
    gc.p_destroy = (decltype(gc.p_destroy))dlsym(gc.dl, "ncclCommDestroy");
    gc.p_errstr = (decltype(gc.p_errstr))dlsym(gc.dl, "ncclGetErrorString");
    if (!gc.p_get_id || !gc.p_init || !gc.p_allreduce || !gc.p_allgather || !gc.p_destroy || !gc.p_errstr)
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

// the stream the collective (and what follows it) runs on: the lane's own (a reserved-CU stream for the collective was
// measured in round 2 and removed: no gain)
inline hipStream_t coll_stream() { return cx().stream; }

// in-place sum over ranks of n u64 lanes in device memory, ordered after everything queued on the lane's stream
int coll_allreduce(unsigned long long* d, int n) {
    if (cx().lc.comm) {
        NCCLCHK(gc.p_allreduce(d, d, (size_t)n, ncclUint64, ncclSum, cx().lc.comm, coll_stream()));
        return 0;
    }
    if (cx().lc.shm) {
        if ((size_t)n > kShmSlotWords) return fail("shm all-reduce of %d words exceeds the slot", n);
        if (!cx().lc.h_tmp) HIPCHK(hipHostMalloc(&cx().lc.h_tmp, sizeof(unsigned long long) * kShmSlotWords, hipHostMallocDefault));
        HIPCHK(hipMemcpyAsync(cx().lc.h_tmp, d, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost, cx().stream));
        HIPCHK(hipStreamSynchronize(cx().stream));
        memcpy(cx().lc.shm_slots + (size_t)shard_view().rank * kShmSlotWords, cx().lc.h_tmp, sizeof(unsigned long long) * n);
        CHK(shm_barrier());
        for (int i = 0; i < n; i++) {
            unsigned long long s = 0;
            for (int r = 0; r < shard_view().world; r++) s += cx().lc.shm_slots[(size_t)r * kShmSlotWords + i];
            cx().lc.h_tmp[i] = s;
        }
        CHK(shm_barrier());
        HIPCHK(hipMemcpyAsync(d, cx().lc.h_tmp, sizeof(unsigned long long) * n, hipMemcpyHostToDevice, cx().stream));
        HIPCHK(hipStreamSynchronize(cx().stream));
        return 0;
    }
    return 0;
}
// The all-reduced words (device memory, lc.d_buf) reach the host like the un-sharded round sums: a one-block copy kernel
// into the host-mapped buffer, then the sequence flag the host polls.  (A copy plus a stream memory operation executed by
// the command processor instead of the kernel was measured in round 2: no faster, removed.)
int coll_publish(int nwords, unsigned int seq) {
    hipLaunchKernelGGL(k_publish_words, dim3(1), dim3(128), 0, cx().stream, cx().lc.d_buf, cx().d_round, nwords, cx().d_flag, seq);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- the ticker: every lane's exchange through ONE communicator, ONE stream, ONE issuing thread ----------------------
// Several proofs in flight per rank (lanes) each need a tiny all-reduce per sumcheck round, at times that differ from
// lane to lane and from rank to rank.  One communicator per lane (round 1-2) makes each lane's collectives ordered, but
// the collectives of DIFFERENT lanes are then issued in rank-dependent order from different host threads, and that is
// only deadlock-free while every lane's stream sits on a hardware queue of its own (two collective kernels waiting for
// their peers on one queue can block each other across ranks) -- an assumption about the runtime, not a property of the
// code.  The ticker removes the assumption: a single thread per rank issues back-to-back "ticks" on a single
// communicator and stream; a tick is ONE ncclAllReduce (ncclUint64, ncclSum) over the concatenation of every lane's slot,
//     slot k = [ count | payload (kTickPayload words) ],
// where a lane that has words to exchange contributes (1, words) and every other lane contributes zeros.  After the
// tick every rank reads the same sums: count == world means lane k's exchange is complete on all ranks (the payload is
// the sum); 0 < count < world means some ranks were not there yet -- the ranks that were simply contribute again in the
// next tick.  Every rank takes the same decision from the same numbers, the order of collectives on the one communicator
// is trivially identical everywhere, and lanes progress independently.  An all-gather is the all-reduce of a vector each
// rank fills at its own offset.  A word of the header carries the votes to stop (gkrhip_comm_destroy): the tickers leave
// together after the tick in which every rank voted.
// The send/receive buffers are host-mapped (the all-reduce kernel reads and writes them over the fabric; 13 KB per
// tick), so a lane's sums go kernel -> host (as un-sharded) -> tick -> host with no extra device copy; only the ticker
// thread touches them, lanes hand their words over through per-lane staging areas.
const int kTickPayload = 192;                 // words: 72/18/81 round sums, or world x (arity + 1) x 4 gathered words
const int kTickStride = kTickPayload + 8;     // count + padding + payload
const int kTickHeader = 8;                    // word 0: votes to stop
const int kTickMaxLanes = 8;
struct TickSlot {
    std::atomic<int> state{0};                // 0 free, 1 posted by the lane, 2 result ready
    int nwords = 0;
    unsigned long long words[kTickPayload];
};
struct Ticker {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int nlanes = 0, world = 1;
    unsigned long long *h_send = nullptr, *d_send = nullptr, *h_recv = nullptr, *d_recv = nullptr;
    unsigned int *h_done = nullptr, *d_done = nullptr;
    // transport of a tick: ncclAllReduce on the host-mapped buffers (default) | ncclAllReduce on device staging buffers with
    // copies around it (dev_buf: insurance should a RCCL build refuse host-mapped user buffers) | a host all-reduce through
    // a POSIX shared-memory segment (shm: the SAME ticker logic with several ranks on one GPU, which RCCL cannot do -- tests)
    bool dev_buf = false;
    unsigned long long *d_stage_send = nullptr, *d_stage_recv = nullptr;
    ShmHdr* shm = nullptr;
    unsigned long long* shm_slots = nullptr;
    size_t shm_bytes = 0;
    std::string shm_name;
    int rank = 0;
    TickSlot slot[kTickMaxLanes];
    bool in_flight[kTickMaxLanes] = {false};
    std::thread th;
    std::atomic<bool> stop_vote{false}, failed{false}, running{false};
    std::atomic<unsigned long long> ticks{0}, idle_ticks{0};
    std::string error;
    std::mutex err_mu;
};
Ticker* g_ticker = nullptr;

void ticker_fail(Ticker* t, const std::string& why) {
    std::lock_guard<std::mutex> lk(t->err_mu);
    if (!t->failed.load()) t->error = why;
    t->failed.store(true, std::memory_order_release);
}

// barrier of the ticker's own shared-memory segment (the lanes' shm_barrier works on the current lane's segment)
bool ticker_shm_barrier(Ticker* t) {
    ShmHdr* h = t->shm;
    const unsigned gen = h->gen.load(std::memory_order_acquire);
    if (h->abort.load(std::memory_order_acquire)) return false;
    if (h->arrive.fetch_add(1, std::memory_order_acq_rel) == (unsigned)t->world - 1) {
        h->arrive.store(0, std::memory_order_relaxed);
        h->gen.fetch_add(1, std::memory_order_release);
        return true;
    }
    const double t0 = now_ms();
    unsigned long spins = 0;
    while (h->gen.load(std::memory_order_acquire) == gen) {
        __builtin_ia32_pause();
        if ((++spins & 0xfff) != 0) continue;
        if (h->abort.load(std::memory_order_acquire)) return h->gen.load(std::memory_order_acquire) != gen;
        if (now_ms() - t0 > coll_timeout_ms()) {
            h->abort.store(1, std::memory_order_release);
            return false;
        }
        struct timespec ts = {0, 2000};
        nanosleep(&ts, nullptr);       // several ranks' tickers share the cores of one box in the tests
    }
    return true;
}

void ticker_main(Ticker* t, int device) {
    if (hipSetDevice(device) != hipSuccess) {
        ticker_fail(t, "ticker: hipSetDevice failed");
        return;
    }
    const size_t total = kTickHeader + (size_t)t->nlanes * kTickStride;
    unsigned int tick_id = 0;
    int idle_run = 0;
    while (!t->failed.load(std::memory_order_acquire)) {
        // a tick goes out at once when a lane of this rank has words (or this rank wants to stop); an idle rank still
        // joins -- its peers may have words -- after a short wait that grows while nothing happens anywhere
        bool any = t->stop_vote.load(std::memory_order_acquire);
        const double t_wait0 = now_ms();
        const double max_wait_ms = idle_run < 64 ? 0.05 : (idle_run < 1024 ? 0.25 : 1.0);
        for (;;) {
            for (int k = 0; k < t->nlanes; k++) any = any || t->slot[k].state.load(std::memory_order_acquire) == 1;
            if (any || now_ms() - t_wait0 > max_wait_ms) break;
            __builtin_ia32_pause();
        }
        // assemble: nobody else touches h_send / h_recv, and no collective is in flight here
        t->h_send[0] = t->stop_vote.load(std::memory_order_acquire) ? 1ull : 0ull;
        for (int k = 0; k < t->nlanes; k++) {
            unsigned long long* s = t->h_send + kTickHeader + (size_t)k * kTickStride;
            if (t->slot[k].state.load(std::memory_order_acquire) == 1) {
                if (!t->in_flight[k]) {
                    s[0] = 1;
                    memcpy(s + 8, t->slot[k].words, sizeof(unsigned long long) * t->slot[k].nwords);
                    t->in_flight[k] = true;
                }
            } else if (s[0] != 0) {
                memset(s, 0, sizeof(unsigned long long) * kTickStride);
            }
        }
        __sync_synchronize();
        ++tick_id;
        if (t->shm) {
            // host all-reduce: slot write, barrier, sum, barrier
            memcpy(t->shm_slots + (size_t)t->rank * total, t->h_send, sizeof(unsigned long long) * total);
            if (!ticker_shm_barrier(t)) {
                ticker_fail(t, "ticker: a peer rank failed or left (shared-memory tick)");
                break;
            }
            for (size_t i = 0; i < total; i++) {
                unsigned long long sum = 0;
                for (int r = 0; r < t->world; r++) sum += t->shm_slots[(size_t)r * total + i];
                t->h_recv[i] = sum;
            }
            if (!ticker_shm_barrier(t)) {
                ticker_fail(t, "ticker: a peer rank failed or left (shared-memory tick)");
                break;
            }
        } else {
        unsigned long long* sb = t->dev_buf ? t->d_stage_send : t->d_send;
        unsigned long long* rb = t->dev_buf ? t->d_stage_recv : t->d_recv;
        if (t->dev_buf && hipMemcpyAsync(sb, t->h_send, sizeof(unsigned long long) * total, hipMemcpyHostToDevice, t->stream) != hipSuccess) {
            ticker_fail(t, "ticker: staging copy failed");
            break;
        }
        ncclResult_t r = gc.p_allreduce(sb, rb, total, ncclUint64, ncclSum, t->comm, t->stream);
        if (r != ncclSuccess) {
            ticker_fail(t, std::string("ticker: ncclAllReduce failed: ") + (gc.p_errstr ? gc.p_errstr(r) : "?"));
            break;
        }
        // the reduced words reach the host buffer (a copy kernel in staging mode), then the flag: ordered behind the all-reduce
        hipLaunchKernelGGL(k_publish_words, dim3(1), dim3(128), 0, t->stream, (const unsigned long long*)rb, t->d_recv,
                           t->dev_buf ? (int)total : 0, t->d_done, tick_id);
        {
            const double t0 = now_ms();
            unsigned long spins = 0;
            while (*(volatile unsigned int*)t->h_done != tick_id) {
                __builtin_ia32_pause();
                if ((++spins & 0x3fff) == 0) {
                    hipError_t e = hipStreamQuery(t->stream);
                    if (e != hipSuccess && e != hipErrorNotReady) {
                        ticker_fail(t, std::string("ticker: stream error: ") + hipGetErrorString(e));
                        break;
                    }
                    if (now_ms() - t0 > coll_timeout_ms()) {
                        ticker_fail(t, "ticker: a tick did not complete within the collective time-out (a peer rank is gone?)");
                        break;
                    }
                }
            }
            if (t->failed.load()) break;
            __sync_synchronize();
        }
        }
        t->ticks.fetch_add(1, std::memory_order_relaxed);
        bool progressed = false;
        for (int k = 0; k < t->nlanes; k++) {
            const unsigned long long* rv = t->h_recv + kTickHeader + (size_t)k * kTickStride;
            if (rv[0] == (unsigned long long)t->world && t->in_flight[k]) {
                memcpy(t->slot[k].words, rv + 8, sizeof(unsigned long long) * t->slot[k].nwords);
                unsigned long long* s = t->h_send + kTickHeader + (size_t)k * kTickStride;
                memset(s, 0, sizeof(unsigned long long) * kTickStride);
                t->in_flight[k] = false;
                t->slot[k].state.store(2, std::memory_order_release);
                progressed = true;
            } else if (rv[0] > (unsigned long long)t->world) {
                ticker_fail(t, "ticker: a lane's count exceeds the world size (ranks disagree about the lanes)");
            }
            progressed = progressed || rv[0] != 0;
        }
        idle_run = progressed ? 0 : idle_run + 1;
        if (!progressed) t->idle_ticks.fetch_add(1, std::memory_order_relaxed);
        if (t->h_recv[0] == (unsigned long long)t->world) break;      // every rank voted to stop
    }
    t->running.store(false, std::memory_order_release);
}

// in-place sum over the ranks of n host words of the current lane, through the ticker
int tick_allreduce(unsigned long long* words, int n) {
    Ticker* t = g_ticker;
    const int k = cx().lc.tick_lane;
    if (!t || k < 0 || k >= t->nlanes) return fail("tick exchange: the lane has no ticker slot");
    if (n > kTickPayload) return fail("tick exchange of %d words exceeds the slot (%d)", n, kTickPayload);
    TickSlot& s = t->slot[k];
    if (t->failed.load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lk(t->err_mu);
        return fail("sharded prover: %s", t->error.c_str());
    }
    if (s.state.load(std::memory_order_acquire) != 0) return fail("tick exchange: the lane's slot is still in use by an abandoned exchange");
    memcpy(s.words, words, sizeof(unsigned long long) * n);
    s.nwords = n;
    s.state.store(1, std::memory_order_release);
    Waiter w;
    const double t0 = now_ms();
    unsigned long spins = 0;
    while (s.state.load(std::memory_order_acquire) != 2) {
        w.step();
        if ((++spins & 0xfff) == 0) {
            if (t->failed.load(std::memory_order_acquire)) {
                std::lock_guard<std::mutex> lk(t->err_mu);
                return fail("sharded prover: %s", t->error.c_str());
            }
            if (!t->running.load(std::memory_order_acquire)) return fail("sharded prover: the ticker has stopped (communicator destroyed)");
            if (now_ms() - t0 > coll_timeout_ms()) {
                // The slot stays posted and the ticker would go on contributing these words: a peer that arrives later (or
                // starts its next proof on this lane) would complete ITS exchange against the payload of this abandoned one.
                // A lane's time-out therefore stops the ticker: every lane of this rank fails, and the peers' ticks run
                // into their own time-out -- all ranks error out together instead of one of them proving on stale sums.
                ticker_fail(t, "a lane timed out waiting for the other ranks (tick exchange): the ticker is stopped");
                return fail("sharded prover: timed out after %.0f s waiting for the other ranks (tick exchange)", coll_timeout_ms() * 1e-3);
            }
        }
    }
    memcpy(words, s.words, sizeof(unsigned long long) * n);
    s.state.store(0, std::memory_order_release);
    return 0;
}

// The same sum over ranks for words that are already on the host (the round kernel's host-mapped hand-off):
// slot write, barrier, sum, barrier.  No device round trip at all.
int shm_allreduce_host(unsigned long long* words, int n) {
    if ((size_t)n > kShmSlotWords) return fail("shm all-reduce of %d words exceeds the slot", n);
    const ShardView v = shard_view();
    memcpy(cx().lc.shm_slots + (size_t)v.rank * kShmSlotWords, words, sizeof(unsigned long long) * n);
    CHK(shm_barrier());
    for (int i = 0; i < n; i++) {
        unsigned long long s = 0;
        for (int r = 0; r < v.world; r++) s += cx().lc.shm_slots[(size_t)r * kShmSlotWords + i];
        words[i] = s;
    }
    CHK(shm_barrier());
    return 0;
}
// all-gather of `cnt` field elements per rank (host values): rank g's elements land in out[g*cnt ..].
// the per-round exchange of this lane runs on the host (shared memory or the ticker): round_collect can carry a vote
inline bool host_exchange() { return cx().lc.tick_lane >= 0 || cx().lc.shm != nullptr; }
// Sharded host tail (one gather of the exported tables instead of h + 1 exchanged rounds): through the ticker a gather of cnt
// elements per rank is ceil(cnt / (kTickPayload / (4 world))) ticks -- 11 for the 64 elements of a cipher layer at h = 4 on 8
// ranks, against the 5 ticks of the rounds it replaces -- so on that transport the tail is taken only where the gather is
// the shorter one (2 ranks: 3 ticks against 5).  The other transports gather in one exchange.
inline bool sharded_tail_pays(int elements_per_rank, int h) {
    if (cx().lc.tick_lane < 0) return true;
    const int per = kTickPayload / (4 * shard_view().world);
    return per >= 1 && (elements_per_rank + per - 1) / per <= h + 1;
}

int coll_allgather(const E* mine, int cnt, std::vector<E>& out) {
    const ShardView v = shard_view();
    out.assign((size_t)v.world * cnt, hfr::ZERO);
    if (v.world == 1 && !cx().force_collective) {
        for (int i = 0; i < cnt; i++) out[i] = mine[i];
        return 0;
    }
    const size_t words = (size_t)v.world * cnt * 4;
    if (cx().lc.tick_lane >= 0) {
        // through the ticker: the all-reduce of a vector every rank fills at its own offset, in chunks of as many elements per
        // rank as fit a slot
        const int per = kTickPayload / (4 * v.world);
        if (per < 1) return fail("tick all-gather: %d ranks do not fit a slot", v.world);
        for (int c0 = 0; c0 < cnt; c0 += per) {
            const int cn = std::min(per, cnt - c0);
            unsigned long long buf[kTickPayload];
            const size_t w = (size_t)v.world * cn * 4;
            memset(buf, 0, sizeof(unsigned long long) * w);
            memcpy(buf + (size_t)v.rank * cn * 4, mine + c0, (size_t)cn * 32);
            CHK(tick_allreduce(buf, (int)w));
            for (int r = 0; r < v.world; r++) memcpy(&out[(size_t)r * cnt + c0], buf + (size_t)r * cn * 4, (size_t)cn * 32);
        }
        return 0;
    }
    if (cx().lc.shm && !cx().lc.comm) {
        // host transport: every rank writes its elements into its slot and reads the others' -- no device round trip
        // (in chunks of a slot: the sharded host tail at h = 10 gathers 4 096 elements per rank, two slots -- found by the oracle
        // test ADVICE r5 asked for; until round 6 this was an error)
        const int per = (int)(kShmSlotWords / 4);
        for (int c0 = 0; c0 < cnt; c0 += per) {
            const int cn = std::min(per, cnt - c0);
            memcpy(cx().lc.shm_slots + (size_t)v.rank * kShmSlotWords, mine + c0, (size_t)cn * 32);
            CHK(shm_barrier());
            for (int r = 0; r < v.world; r++) memcpy(&out[(size_t)r * cnt + c0], cx().lc.shm_slots + (size_t)r * kShmSlotWords, (size_t)cn * 32);
            CHK(shm_barrier());
        }
        return 0;
    }
    if (!cx().lc.comm) {     // the collective code path forced at world = 1 without a communicator
        for (int i = 0; i < cnt; i++) out[i] = mine[i];
        return 0;
    }
    // RCCL: ncclAllGather of cnt elements per rank (staged through the lane's exchange buffer: send block after the
    // receive area)
    const size_t mine_words = (size_t)cnt * 4;
    CHK(coll_buffers(std::max<size_t>(words + mine_words, 256)));
    memcpy(cx().lc.h_buf, mine, mine_words * 8);
    HIPCHK(hipMemcpyAsync(cx().lc.d_buf + words, cx().lc.h_buf, mine_words * 8, hipMemcpyHostToDevice, cx().stream));
    NCCLCHK(gc.p_allgather(cx().lc.d_buf + words, cx().lc.d_buf, mine_words, ncclUint64, cx().lc.comm, coll_stream()));
    HIPCHK(hipMemcpyAsync(cx().lc.h_buf, cx().lc.d_buf, words * 8, hipMemcpyDeviceToHost, coll_stream()));
    HIPCHK(hipStreamSynchronize(coll_stream()));
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
int partial_evals(const GateDesc& g, const DevTable* eq, const DevTable* const* x, size_t mid, const E& ark, E* evals,
                  int nev, bool collective) {
    int nblocks = 0;
    const bool tick = collective && cx().lc.tick_lane >= 0;
    const bool direct = !collective || tick;      // un-sharded (or exchanged on the host through the ticker): the kernel hands the sums to the host itself
    const bool timed = 2 * mid >= cx().prof.min_n;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timed) {
        e0 = prof_event();
        e1 = prof_event();
        HIPCHK(hipEventRecord(e0, cx().stream));
    }
#define GKR_PE(P, A) CHK((launch_partial_eval_t<P, A, P + 2>(eq, x, mid, ark, g.mask, &nblocks, direct)))
    switch (g.power * 10 + g.n_in) {
        case 11: GKR_PE(1, 1); break;
        case 12: GKR_PE(1, 2); break;
        case 13: GKR_PE(1, 3); break;
        case 14: GKR_PE(1, 4); break;
        case 71: GKR_PE(7, 1); break;
        case 72: GKR_PE(7, 2); break;
        case 73: GKR_PE(7, 3); break;
        case 74: GKR_PE(7, 4); break;
        default: return fail("unsupported gate shape (power %d, %d inputs)", g.power, g.n_in);
    }
#undef GKR_PE
    HIPCHK(hipGetLastError());
    if (timed) {
        HIPCHK(hipEventRecord(e1, cx().stream));
        cx().prof.peval_ev.emplace_back(e0, e1);
        cx().prof.peval_launches++;
        cx().prof.peval_modmuls += (g.power == 7 ? 45.0 : 3.0) * (double)mid;
    }
    const int nwords = nev * GKR_ACC_WORDS;
    if (direct) {
        CHK(wait_flag(cx().seq));
        const unsigned long long* sums = cx().h_round;
        unsigned long long summed[GKR_MAX_EVALS * GKR_ACC_WORDS];
        if (tick) {
            memcpy(summed, cx().h_round, sizeof(unsigned long long) * nwords);
            CHK(tick_allreduce(summed, nwords));
            sums = summed;
        }
        for (int t = 0; t < nev; t++) evals[t] = limbs9_to_fr(sums + (size_t)t * GKR_ACC_WORDS);
        return 0;
    }
    hipLaunchKernelGGL(k_reduce_partials, dim3(nwords), dim3(GKR_BLOCK), 0, cx().stream, cx().d_partials, cx().d_sums, nblocks, nwords);
    HIPCHK(hipGetLastError());
    if (collective) CHK(coll_allreduce(cx().d_sums, nwords));
    hipStream_t after = collective ? coll_stream() : cx().stream;
    HIPCHK(hipMemcpyAsync(cx().h_sums, cx().d_sums, sizeof(unsigned long long) * nwords, hipMemcpyDeviceToHost, after));
    HIPCHK(hipStreamSynchronize(after));
    for (int t = 0; t < nev; t++) evals[t] = limbs9_to_fr(cx().h_sums + (size_t)t * GKR_ACC_WORDS);
    return 0;
}

// Coordinates (and seeds) of a layer -> cx().d_q.  Staged through one of two pinned buffers and copied on the lane's
// stream WITHOUT waiting: the kernels that read d_q are queued behind the copy, and the host waits for one of them
// (a round's flag, a stream synchronisation) before it stages the next coordinates -- two buffers make that a certainty.
int stage_coords(const E* coords, size_t n) {
    if (n > cx().d_q_cap) {
        HIPCHK(hipStreamSynchronize(cx().stream));
        if (cx().d_q) HIPCHK(hipFree(cx().d_q));
        for (auto& h : cx().h_q)
            if (h) HIPCHK(hipHostFree(h));
        cx().d_q_cap = std::max<size_t>(n, 256);
        HIPCHK(hipMalloc(&cx().d_q, sizeof(Fr) * cx().d_q_cap));
        for (auto& h : cx().h_q) HIPCHK(hipHostMalloc(&h, sizeof(Fr) * cx().d_q_cap, hipHostMallocDefault));
    }
    if (n == 0) return 0;
    Fr* stage = cx().h_q[cx().h_q_next ^= 1];
    for (size_t i = 0; i < n; i++) stage[i] = to_dev(coords[i]);
    HIPCHK(hipMemcpyAsync(cx().d_q, stage, sizeof(Fr) * n, hipMemcpyHostToDevice, cx().stream));
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

    ScopedTable thi, tlo;
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
