// host_ctx.hip.h -- lanes (per-proof contexts), the device table arena, boundary copies, profiling events and
// kernel launch wrappers of the host drivers.  Included by gkrhip.hip inside its anonymous namespace.
#pragma once

struct DevTable {
    uint4* base = nullptr;
    size_t cap = 0;  // elements per plane
    Planes planes() const { return Planes{base, base + cap}; }
    CPlanes cplanes() const { return CPlanes{base, base + cap}; }
};

static inline double now_ms() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

// GKRHIP_TRACE: where the host's time goes between the categories of Profile, by named lap (thread-local: one proof traced at a time)
struct Laps {
    static const int N = 40;
    const char* name[N] = {nullptr};
    double acc[N] = {0};
    double last = 0;
    bool on = false;
    void start() {
        on = getenv("GKRHIP_TRACE") != nullptr;
        for (int i = 0; i < N; i++) acc[i] = 0, name[i] = nullptr;
        last = now_ms();
    }
    void lap(const char* what) {
        if (!on) return;
        const double t = now_ms();
        for (int i = 0; i < N; i++) {
            if (name[i] == what || name[i] == nullptr) {
                name[i] = what;
                acc[i] += t - last;
                break;
            }
        }
        last = t;
    }
    void dump() const {
        if (!on) return;
        for (int i = 0; i < N && name[i]; i++) fprintf(stderr, "  lap %-38s %8.3f ms\n", name[i], acc[i]);
    }
};
thread_local Laps g_laps;
#define LAP(what) g_laps.lap(what)

// process-wide counts of the serial-latency paths (gkrhip_profile_latency): rounds whose kernel was queued ahead of its
// challenge, round-0 launches on look-ahead products, rounds of the cooperative kernel.  Not per lane: the lanes of
// one-shot calls go back to the pool (and are cleared) before anybody can ask.
std::atomic<uint64_t> g_cnt_prelaunched{0}, g_cnt_lookahead{0}, g_cnt_coop{0}, g_cnt_spec{0}, g_cnt_retries{0};
// the prover's own check of every sumcheck it produces (host_sumcheck.hip.h, sumcheck_closes): sumchecks checked, sumchecks
// that did not close and were run again, and gkrhip_set_option("layer_check", 0 | 1) / ("verify_after_prove", 0 | 1)
std::atomic<uint64_t> g_cnt_layer_checks{0}, g_cnt_layer_check_failures{0}, g_cnt_ahead{0};
std::atomic<int> g_layer_check{1}, g_verify_after_prove{0};
// what gkrhip_init did about the runtime's hardware queues: the count it put into GPU_MAX_HW_QUEUES (0: it left the variable alone), or the
// count it found there.  Whether the runtime honoured it depends on who made the process's first HIP call (gkrhip_profile_counter).
std::atomic<int> g_hwq_set_by_library{0}, g_hwq_from_env{0};
struct Profile {
    double host_hash_ms = 0, host_wait_ms = 0, host_launch_ms = 0, host_other_ms = 0;
    uint64_t rounds = 0;
    double wait_lg[40] = {0};                  // GKRHIP_TRACE_ROUNDS: host wait per cipher round, by log2(pairs)
    uint64_t cnt_lg[40] = {0};
    double setup_ms = 0, tail_ms = 0;          // per-layer set-up (coordinates, pyramids, tables); host-tail rounds
    size_t min_n = (size_t)1 << 62;
    uint64_t fold_launches = 0, peval_launches = 0;
    double fold_bytes = 0, peval_modmuls = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> fold_ev, peval_ev;
    std::vector<hipEvent_t> pool;
};

inline void prof_clear(Profile& p) {              // events go back to the lane's event pool
    for (auto& e : p.fold_ev) {
        p.pool.push_back(e.first);
        p.pool.push_back(e.second);
    }
    for (auto& e : p.peval_ev) {
        p.pool.push_back(e.first);
        p.pool.push_back(e.second);
    }
    p.fold_ev.clear();
    p.peval_ev.clear();
    p.fold_launches = p.peval_launches = 0;
    p.fold_bytes = p.peval_modmuls = 0;
    p.host_hash_ms = p.host_wait_ms = p.host_launch_ms = p.host_other_ms = 0;
    p.rounds = 0;
    for (int i = 0; i < 40; i++) p.wait_lg[i] = 0, p.cnt_lg[i] = 0;
    p.setup_ms = p.tail_ms = 0;
}

// per-lane state of the collective (one communicator / shared-memory segment per lane: the lanes of a rank
// issue their collectives independently, lane k pairing with lane k of the other ranks)
struct ShmHdr {
    std::atomic<unsigned> arrive, gen;
    std::atomic<unsigned> abort;             // set by a rank that fails (or leaves): peers stop waiting and fail too
    std::atomic<unsigned long long> magic;   // kShmMagic ^ creation time (s): written last by rank 0; peers refuse anything else
};
const unsigned long long kShmMagic = 0x676b726869700000ull;   // "gkrhip"
struct LaneColl {
    ncclComm_t comm = nullptr;
    ShmHdr* shm = nullptr;
    unsigned long long* shm_slots = nullptr;
    size_t shm_bytes = 0;
    unsigned long long* d_buf = nullptr;   // device staging: lanes / gathered elements
    unsigned long long* h_buf = nullptr;   // pinned mirror
    unsigned long long* h_tmp = nullptr;
    size_t buf_words = 0;
    unsigned int* h_cflag = nullptr;       // host-mapped completion word of the RCCL path (written by a stream memory operation)
    unsigned int* d_cflag = nullptr;
    int tick_lane = -1;                    // >= 0: this lane exchanges through the process's ticker (host_coll.hip.h), slot tick_lane
};

struct Ctx {
    bool ready = false;
    int device = -1;
    unsigned ordinal = 0;                      // the lane's number in creation order: its stream was the ordinal-th the library created (the runtime deals hardware queues to streams in turn)
    hipStream_t stream = nullptr;
    unsigned long long* d_partials = nullptr;  // per-block limb-split partial sums (reference-shaped evaluator)
    bool racc_dirty = false;                   // a call that uses d_racc / d_counter is under way or failed half-way
    unsigned long long* d_racc = nullptr;      // GKR_CR_WORDS-word accumulator of the fused round kernels (zero between launches)
    unsigned long long* d_sums = nullptr;      // reduced sums (device)
    unsigned long long* h_sums = nullptr;      // pinned
    uint4* d_small = nullptr;                  // gather buffer (AoS)
    uint4* h_small = nullptr;                  // pinned
    Fr* d_q = nullptr;                         // qPrime coordinates + seeds staging
    size_t d_q_cap = 0;
    Fr* h_q[2] = {nullptr, nullptr};           // pinned staging of d_q (alternating)
    int h_q_next = 0;
    unsigned long long* h_tail = nullptr;      // host-mapped: the tables of the round after which the host takes over (GKRHIP_HOST_TAIL)
    unsigned long long* d_tail = nullptr;
    unsigned int* h_bad = nullptr;             // host-mapped: set by k_aos_to_planes when an uploaded element is >= q
    unsigned int* d_bad = nullptr;
    int max_grid = 2048;
    int fold_grid = 1 << 20;                   // workgroups cap of the fold: one element per lane up to 2^28 outputs
    bool fold_split = true;                    // one single-table launch per table instead of a fused launch
    int n_cu = 256;
    // fused cipher round (cipher_round.hip.h)
    unsigned long long* h_round = nullptr;     // host-mapped: GKR_CR_WORDS sums + 16 tail words
    unsigned long long* d_round = nullptr;     // device view of h_round
    unsigned int* h_flag = nullptr;            // host-mapped completion flag
    unsigned int* d_flag = nullptr;
    unsigned int* d_counter = nullptr;         // block arrival counter
    unsigned int seq = 0;
    int g_max = 16;                            // log2(max threads of a round kernel) ...
    bool g_max_auto = true;                    // ... chosen per layer from the proofs in flight (round_threads_log2_max) unless GKRHIP_GMAX / "g_max" set it
    bool force_generic = false;
    bool claim_trick = true;                   // option claim_trick = 0: always compute all eight monomial sums
    int lat_mode = 1;                          // GKRHIP_LAT: 0 never, 1 rounds with one pair per lane, 2 always
    int wide_mode = 1;                         // GKRHIP_WIDE: deferred-reduction kernel for the rounds with several pairs per lane
    int solo_boost = 1;                        // option solo_boost: twice the threads for the big rounds of a proof that is alone on the GPU
    int pyr_split = 12;                        // option pyr_split: the per-lane eq pyramid above 2^n entries in two launches (0: one launch)
    int wt_late_lj = 3;                        // ... and from 2^3 pairs per lane on, the lane weight is applied after the loop
    bool force_collective = false;             // GKRHIP_FORCE_COLLECTIVE: take the collective path even at world == 1
    int host_tail = 5;                         // GKRHIP_HOST_TAIL: the rounds with at most 2^h pairs run on the host (0: never); measured: -5 % single-proof latency, +1.5 % throughput
    int host_tail_solo = 4;                    // the same for a proof that is alone on the GPU, whose small rounds are fast (cooperative kernel, pre-launched): bN = 20 107.6 ms against 110-112 with 5 and 111.5 with 3; GKRHIP_HOST_TAIL sets both
    int host_tail_sharded = 4;                 // GKRHIP_HOST_TAIL_SHARDED: sharded local rounds: the ranks gather the tables of the round with 2^(h+1) pairs and finish on the host (0: every local round exchanged)
    // ---- serial-latency measures of a proof that is alone on the GPU (round 3) -----------------------------------
    // pre-launched rounds: round k+1's kernel is queued before the host hashes round k and polls the challenge slot
    int prelaunch = 1;                         // GKRHIP_PRELAUNCH: 0 never, 1 when the proof is alone on the GPU, 2 always
    int prelaunch_lg = 30;                     // ... for rounds of at most 2^prelaunch_lg pairs (every round since the waiting workgroups poll the host only rarely; 16 before: same-box 279.8 -> 277.9 ms at bN = 24)
    unsigned long long* h_chal = nullptr;      // host-mapped challenge slots (kChalSlots x GKR_CHAL_WORDS words): slot 0 the pre-launched round kernels', 1 and 2 the speculative rounds' (alternating)
    unsigned long long* d_chal = nullptr;
    unsigned long long* d_chal_dev = nullptr;  // device-memory mailbox: workgroup 0 of a pre-launched kernel forwards the slot to the others
    // round 0 split: the q-independent products of the NEXT layer's round 0 are computed on `aux` while this layer's
    // small rounds leave the GPU idle (k_cipher_pre); the next layer's round 0 then only applies the weights
    int pre_mode = 1;                          // GKRHIP_PRE: 0 never, 1 when the proof is alone on the GPU, 2 always
    int coop = 1;                              // GKRHIP_COOP: cooperative small-round kernel (eight lanes per pair): 0 never, 1 alone on the GPU, 2 always
    int coop_lg = 14;                          // ... for rounds of at most 2^coop_lg pairs
    int coop_wgs = 512;                        // ... on at most this many workgroups
    // speculative small rounds (cipher_spec.hip.h): round k runs for the eight candidate values 0..7 of r_{k-1} while the host
    // still hashes round k-1; the host interpolates at the true challenge
    int spec = 1;                              // GKRHIP_SPEC: 0 never, 1 when the proof is alone on the GPU, 2 always (un-sharded rounds only)
    int spec_max_m = 23;                       // spec == 1 takes layers of at most 2^n entries
    int spec_lg = 13;                          // GKRHIP_SPEC_LG: ... for rounds of at most 2^spec_lg pairs (eight lanes per pair: 2^16 lanes = one wave per SIMD)
    unsigned long long* h_spec = nullptr;      // host-mapped: two result buffers of GKR_SPEC_BUF_WORDS words (rounds alternate)
    unsigned long long* d_spec = nullptr;
    unsigned long long* d_spec_racc = nullptr; // GKR_SPEC_CAND accumulator sets, zero between launches
    // what the host last did with the challenge slots (quoted by the error message when a waiting kernel gave up)
    unsigned int dbg_pub_seq[3] = {0, 0, 0};
    double dbg_pub_ms[3] = {0, 0, 0};
    unsigned int dbg_defer_seq = 0;
    double dbg_defer_ms = 0;
    E spec_pts[8];                             // Montgomery forms of the candidate points 0..7
    E spec_invden[8];                          // 1 / prod_{j != i} (i - j): Lagrange denominators on the points 0..7
    int pre_start_lg = 20;                     // the look-ahead kernel is queued when the layer's rounds reach 2^n pairs (16 before round 0 ran ahead of its point: its products are now wanted at the START of the host tail; bN = 24 alone 269.3 -> 262.1 ms, bN = 22 137.5 -> 134.9)
    hipStream_t aux = nullptr;                 // stream of the look-ahead kernel (normal priority: see pre_prepare)
    hipEvent_t pre_done = nullptr;
    hipEvent_t chk_fence = nullptr;            // arena_check: recorded in front of everything queued AHEAD for the next layer (ahead_launch)
    DevTable pre_t[6];                         // u^4, d^4, u^3, u^2 d, u d^2, d^3 (P entries each); arena tables, released by pre_release()
    const uint4* pre_K = nullptr;              // what pre_t was computed from (valid when pre_K != nullptr)
    const uint4* pre_S = nullptr;
    E pre_ark;
    int pre_m = 0;
    // round 0 ahead of its point (cipher_round.hip.h, ahead_publish): queued by the layer before, at the start of its host tail
    int ahead_mode = 2;                        // GKRHIP_AHEAD: 0 never, 1 when the proof is alone on the GPU, 2 always (un-sharded rounds only; the default: with lanes too the host tail is time the lane's stream has nothing to do -- bN = 20 x 24 lanes +1.3 %, bN = 24 x 5 +1.0 %, GMiMC bN = 22 x 12 +1.2 %)
    unsigned long long* h_ahead = nullptr;     // host-mapped: 7 * 2^t canonical class sums, then the flag word
    unsigned long long* d_ahead = nullptr;
    unsigned long long* d_ahead_racc = nullptr;   // GKR_RACC_SLOTS stripes of GKR_AHEAD_STRIPE words, zero between launches
    unsigned int* d_ahead_counter = nullptr;
    DevTable ahead_pyrU, ahead_pyrU2, ahead_pyrTh;   // arena tables, released by pre_release()
    bool ahead_in_flight = false;              // the class-sum kernel was queued and nobody has waited for it yet (its flag, or a stream
                                               // synchronisation): it may still be reading ahead_pyr* -- which then must not go back to the arena
    const uint4* ahead_K = nullptr;            // what the class sums in flight were computed from (valid when ahead_K != nullptr)
    const uint4* ahead_S = nullptr;
    E ahead_ark;
    int ahead_m = 0, ahead_t = 0;
    unsigned int ahead_seq = 0;
    std::vector<E> ahead_q;                    // the coordinates it used: q[0 .. m-1-t]
    const DevTable* nxt_K = nullptr;           // the layer gkr.Prove proves next, when it is a single-point cipher layer (as req_K, but not consumed by launch_pre)
    const DevTable* nxt_S = nullptr;
    const DevTable* req_K = nullptr;           // look-ahead request of gkr.Prove for the layer it will prove next
    const DevTable* req_S = nullptr;
    E req_ark;
    int req_m = 0;
    hfr::Lagrange* lag = nullptr;
    Profile prof;
    LaneColl lc;
    std::mutex mu;                             // serialises the calls that use this lane
};

// A Ctx is a "lane": one stream plus every buffer a proof in flight needs exclusively.  g0 is the default
// lane (host-buffer entry points, sharded sessions); each un-sharded session owns a lane of its own, so
// independent sessions can prove concurrently from different host threads (one proof's Fiat-Shamir hashing
// and small latency-bound rounds then overlap another proof's big rounds).
Ctx g0;
thread_local Ctx* g_cur = &g0;
// the lane (stream, hand-off buffers, counters, communicator) the calling thread currently works on
static inline Ctx& cx() { return *g_cur; }
struct UseLane {
    Ctx* prev;
    explicit UseLane(Ctx* l) : prev(g_cur) { g_cur = l; }
    ~UseLane() { g_cur = prev; }
};
// ---- how host threads wait (for the round kernel's flag, for the other ranks) -------------------------------
// Spinning is the lowest-latency wait and is what un-sharded runs use: a handful of lanes on a many-core host.
// A sharded run has (ranks on this node) x (lanes) waiting threads; when they outnumber the CPUs the container may
// use (affinity mask and cgroup quota), spinning starves the threads that hash and burns the quota, so the wait
// spins only briefly and then sleeps in short steps.  gkrhip_set_option("wait_spin_us", n) overrides (-1 = always spin).
inline int usable_cpus() {
    cpu_set_t set;
    int n = sched_getaffinity(0, sizeof set, &set) == 0 ? CPU_COUNT(&set) : 1;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {        // cgroup v2: "<quota|max> <period>"
        char q[64];
        long period = 0;
        if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const long quota = atol(q);
            if (quota > 0) n = std::min<long>(n, std::max<long>(1, (quota + period - 1) / period));
        }
        fclose(f);
    }
    return std::max(n, 1);
}
std::atomic<int> g_wait_override{-2};           // option wait_spin_us (-2: not set; -1: always spin; n: spin n us, then sleep)
std::atomic<int> g_wait_ranks{1};               // ranks assumed to share this host (set with the communicator)
std::atomic<int> g_proofs_in_flight{0};          // gkr.Prove calls currently running in this process (any lane)
std::atomic<int> g_group_passengers{0};          // ... of which this many share their host thread with another proof of their group (host_group.hip.h)
// What the plans of a layer read: the proofs of a group must all see the same number (they take the same decisions, launch for
// launch), so inside a group it is the number the group's driver read before it resumed them.
thread_local int t_group_in_flight = -1;
inline int proofs_in_flight_now() {
    return t_group_in_flight >= 0 ? t_group_in_flight : g_proofs_in_flight.load(std::memory_order_relaxed);
}
inline int wait_spin_limit_us() {
    static const int cpus = usable_cpus();
    const int ov = g_wait_override.load(std::memory_order_relaxed);
    if (ov > -2) return ov;
    // one waiting host thread per proof in flight (per group of proofs) and rank
    const int waiting = g_wait_ranks.load(std::memory_order_relaxed) *
                        std::max(1, g_proofs_in_flight.load(std::memory_order_relaxed) - g_group_passengers.load(std::memory_order_relaxed));
    return cpus >= waiting + 2 ? -1 : 25;
}
struct Waiter {                                 // one per wait: call step() in the polling loop
    unsigned long spins = 0;
    double t0 = 0;
    inline void step() {
        __builtin_ia32_pause();
        if ((++spins & 127) != 0) return;
        const int lim = wait_spin_limit_us();
        if (lim < 0) return;
        if (t0 == 0) { t0 = now_ms(); return; }
        if ((now_ms() - t0) * 1e3 > lim) {
            struct timespec ts = {0, 20000};
            nanosleep(&ts, nullptr);
        }
    }
};
// log2 of the threads a round kernel may use.  Measured (profiles/r04_gmax_by_lanes.txt): 2^16 (one workgroup per CU) with a few
// proofs in flight (bN = 24, five: 84.7 M hashes/s against 81.9 with 2^15), 2^15 with many (bN = 20, 24 in flight: 57.8 against
// 54.4; GMiMC bN = 22, 12 in flight: 101.9 against 97.3): the more lanes, the more the other lanes' kernels fill the CUs.
// The proofs of a group (host_group.hip.h) share their launches: 2^13 threads each (bN = 20, 72 in flight in groups of 3: 82.4 M hashes/s
// against 71.7 / 79.2 with 2^12 / 2^14 and 68-71 with 2^15; groups of 4 and 6 likewise: profiles/r06_proof_groups.txt).
thread_local int t_group_size = 0;
inline int round_threads_log2_max(int m) {        // m: log2 of the layer's table
    if (!cx().g_max_auto) return cx().g_max;
    if (t_group_size >= 2) return std::max(13, std::min(15, m - 7));      // (2^13 is for the proofs groups are for: 2^20 entries; 2^24 in groups of 2 with 2^13 threads ran at half the lanes' rate)
    return proofs_in_flight_now() >= 10 ? 15 : 16;
}
struct ProofInFlight {
    ProofInFlight() { g_proofs_in_flight.fetch_add(1, std::memory_order_relaxed); }
    ~ProofInFlight() { g_proofs_in_flight.fetch_sub(1, std::memory_order_relaxed); }
};
std::mutex g_lanes_mu;
std::vector<Ctx*> g_lanes;                     // every lane, for profile aggregation
struct Pool {
    std::mutex mu;
    // cache of table buffers by size class.  (A flat list searched linearly until round 5: after a job of 56 small proofs had left
    // 5 000 buffers of ITS size behind, every allocation of another size walked them all under the arena's lock -- ten allocations
    // per layer and lane: GMiMC bN = 22 with 12 lanes 103.8 M hashes/s behind such a job against 113.9 before it.)
    std::unordered_map<size_t, std::vector<uint4*>> free_list;
} g_pool;
// arena_check (table_release): 0 off, 1 count releases with the lane still busy, 2 also poison what is released
std::atomic<int> g_arena_check{0};
std::atomic<unsigned long long> g_cnt_busy_releases{0};
std::mutex g_busy_mu;
std::vector<std::pair<std::string, int>> g_busy_lines;
thread_local std::string g_err;
// Every failure gets a code of its own (<= -16) and its message is kept under that code in a process-wide ring, so that a
// caller whose thread has changed between the failing call and the question (a goroutine that migrated to another OS
// thread between the cgo call and must()) still gets ITS message: gkrhip_last_error_r(code, ...).  gkrhip_last_error()
// keeps answering from the calling thread's last failure.
struct ErrRing {
    static const int N = 256;
    std::mutex mu;
    int code[N] = {0};
    std::string msg[N];
    unsigned long long next = 0;
} g_errs;

int fail(const char* fmt, ...) {
    char buf[768];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    std::lock_guard<std::mutex> lk(g_errs.mu);
    const unsigned long long n = g_errs.next++;
    const int code = -(int)(16 + (n % 0x3ffffff0ull));
    g_errs.code[n % ErrRing::N] = code;
    g_errs.msg[n % ErrRing::N] = buf;
    return code;
}
// the message recorded under `code` (false: it has left the ring, or the code is not a failure code of this library)
bool error_lookup(int code, std::string* out) {
    std::lock_guard<std::mutex> lk(g_errs.mu);
    for (int i = 0; i < ErrRing::N; i++)
        if (g_errs.code[i] == code && code != 0) {
            *out = g_errs.msg[i];
            return true;
        }
    return false;
}

#define HIPCHK(x)                                                                                 \
    do {                                                                                          \
        hipError_t _e = (x);                                                                      \
        if (_e != hipSuccess) return fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)
#define CHK(x)                  \
    do {                        \
        int _r = (x);           \
        if (_r != 0) return _r; \
    } while (0)

const int kPartialBlocks = 1024;  // max blocks of the partial-evaluation kernel
const int kHostTailMax = 10;      // the host can take over from 2^10 pairs on (GKRHIP_HOST_TAIL <= 10)
const size_t kTailWords = (size_t)GKR_MAX_ARITY * 4 * 2 * (2 << kHostTailMax);   // up to four tables of 2P entries, P <= 2^(kHostTailMax+1), 4 u64 each
const int kChalSlots = 3;
const int kRaccWords = 128;       // shared accumulator / host hand-off buffer: 72 (fused rounds) or up to 81 (nine evaluations) + 16 tail words
int lane_alloc();
void table_release_fwd(DevTable* t);

int ctx_init(int dev) {
    if (cx().ready) {
        if (dev >= 0 && dev != cx().device) return fail("gkrhip already initialised on device %d", cx().device);
        return 0;
    }
    // Streams share the runtime's hardware queues (4 by default), and kernels of different streams in one hardware queue run
    // one after the other -- with many small proofs in flight that, not the GPU, is the limit.  Same-box interleaved A/B of
    // round 4, five runs each, runtime default against 16 queues (profiles/r04_hwq_ab_summary.json):
    //     bN = 20, 24 lanes      46.6 -> 53.8 M hashes/s (+15 %)     the proof alone 102.7 -> 102.1 ms
    //     GMiMC bN = 22, 12 lanes 87.4 -> 94.2 M/s       (+7.7 %)    the proof alone 119.9 -> 124.2 ms (+3.6 %)
    //     bN = 24, 5 lanes       82.2 -> 81.4 M hashes/s (-0.9 %, inside the spread of either set)   alone 278.4 -> 279.2 ms
    // (earlier sweeps: 8 queues give a third of the gain, 24 less than 16).  So the library asks for 16 unless told otherwise:
    // GKRHIP_HW_QUEUES=n sets another count, GKRHIP_HW_QUEUES=0 leaves the runtime's default, and an explicit
    // GPU_MAX_HW_QUEUES is left alone.  The runtime reads the variable when it initialises, so this only takes effect if
    // the library makes the process's first HIP call.
    if (!getenv("GPU_MAX_HW_QUEUES")) {
        const char* hq = getenv("GKRHIP_HW_QUEUES");
        const int nq = hq ? atoi(hq) : 16;
        if (nq > 0) {
            setenv("GPU_MAX_HW_QUEUES", std::to_string(std::min(nq, 64)).c_str(), 0);
            g_hwq_set_by_library.store(std::min(nq, 64));      // (a process-wide side effect: INTEGRATION.md)
        }
    } else {
        g_hwq_from_env.store(atoi(getenv("GPU_MAX_HW_QUEUES")));
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail("no HIP device available (%s): libgkrhip has no CPU fallback", hipGetErrorString(e));
    if (dev < 0) dev = 0;
    if (dev >= n) return fail("device ordinal %d out of range (%d devices)", dev, n);
    HIPCHK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail("device %d is %s; libgkrhip is built for gfx950 (MI355X) only", dev, prop.gcnArchName);
    cx().n_cu = prop.multiProcessorCount;
    cx().max_grid = cx().n_cu * 32;   // streaming kernels: 8192 workgroups measured best for the fold (profiles/)
    if (const char* e = getenv("GKRHIP_GMAX")) {
        cx().g_max = std::max(0, std::min(20, atoi(e)));
        cx().g_max_auto = false;
    }
    if (const char* e = getenv("GKRHIP_GENERIC")) cx().force_generic = atoi(e) != 0;
    if (const char* e = getenv("GKRHIP_LAT")) cx().lat_mode = atoi(e);
    if (const char* e = getenv("GKRHIP_WIDE")) cx().wide_mode = atoi(e);
    if (const char* e = getenv("GKRHIP_FORCE_COLLECTIVE")) cx().force_collective = atoi(e) != 0;
    if (const char* e = getenv("GKRHIP_HOST_TAIL")) cx().host_tail = cx().host_tail_solo = std::max(0, std::min(kHostTailMax, atoi(e)));   // an explicit setting holds for both
    if (const char* e = getenv("GKRHIP_HOST_TAIL_SHARDED")) cx().host_tail_sharded = std::max(0, std::min(kHostTailMax, atoi(e)));
    if (const char* e = getenv("GKRHIP_PRELAUNCH")) cx().prelaunch = atoi(e);
    if (const char* e = getenv("GKRHIP_PRE")) cx().pre_mode = atoi(e);
    if (const char* e = getenv("GKRHIP_SPEC")) cx().spec = atoi(e);
    if (const char* e = getenv("GKRHIP_AHEAD")) cx().ahead_mode = atoi(e);
    if (const char* e = getenv("GKRHIP_SPEC_LG")) cx().spec_lg = std::max(5, std::min(16, atoi(e)));
    if (const char* e = getenv("GKRHIP_COOP")) cx().coop = atoi(e);
    if (const char* e = getenv("GKRHIP_COOP_LG")) cx().coop_lg = std::max(0, std::min(20, atoi(e)));
    cx().lag = new hfr::Lagrange();
    cx().device = dev;
    CHK(lane_alloc());
    {
        std::lock_guard<std::mutex> lk(g_lanes_mu);
        g_lanes.push_back(&cx());
    }
    cx().ready = true;
    return 0;
}

// stream + buffers of the current lane
int lane_alloc() {
    HIPCHK(hipStreamCreateWithFlags(&cx().stream, hipStreamNonBlocking));
    const size_t nwords = (size_t)GKR_MAX_EVALS * GKR_ACC_WORDS;
    HIPCHK(hipMalloc(&cx().d_partials, sizeof(unsigned long long) * nwords * kPartialBlocks));
    HIPCHK(hipMalloc(&cx().d_sums, sizeof(unsigned long long) * nwords));
    static_assert(kRaccWords == GKR_RACC_STRIDE, "one accumulator copy per stride");
    HIPCHK(hipMalloc(&cx().d_racc, sizeof(unsigned long long) * kRaccWords * GKR_RACC_SLOTS));
    // (stream-ordered zeroing: the lane's stream is non-blocking, so a null-stream hipMemset -- asynchronous to the host for device
    // memory -- is NOT ordered before the lane's first kernels; with 16 hardware queues the first proof of a new lane lost that
    // race about once in ten runs of the solo soak: a wrong first transcript, caught by the native verifier)
    HIPCHK(hipMemsetAsync(cx().d_racc, 0, sizeof(unsigned long long) * kRaccWords * GKR_RACC_SLOTS, cx().stream));
    HIPCHK(hipHostMalloc(&cx().h_sums, sizeof(unsigned long long) * nwords, hipHostMallocDefault));
    HIPCHK(hipMalloc(&cx().d_small, sizeof(uint4) * 2 * 8));
    HIPCHK(hipHostMalloc(&cx().h_small, sizeof(uint4) * 2 * 8, hipHostMallocDefault));
    HIPCHK(hipHostMalloc(&cx().h_round, sizeof(unsigned long long) * kRaccWords, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer((void**)&cx().d_round, cx().h_round, 0));
    HIPCHK(hipHostMalloc(&cx().h_flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer((void**)&cx().d_flag, cx().h_flag, 0));
    *cx().h_flag = 0;
    cx().seq = 0;
    HIPCHK(hipMalloc(&cx().d_counter, 64));
    HIPCHK(hipMemsetAsync(cx().d_counter, 0, 64, cx().stream));
    HIPCHK(hipHostMalloc(&cx().h_tail, kTailWords * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer((void**)&cx().d_tail, cx().h_tail, 0));
    HIPCHK(hipHostMalloc(&cx().h_bad, 64, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer((void**)&cx().d_bad, cx().h_bad, 0));
    *cx().h_bad = 0;
    HIPCHK(hipHostMalloc(&cx().h_chal, sizeof(unsigned long long) * GKR_CHAL_WORDS * kChalSlots, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer((void**)&cx().d_chal, cx().h_chal, 0));
    memset(cx().h_chal, 0, sizeof(unsigned long long) * GKR_CHAL_WORDS * kChalSlots);
    HIPCHK(hipMalloc(&cx().d_chal_dev, sizeof(unsigned long long) * GKR_CHAL_WORDS * kChalSlots));
    HIPCHK(hipMemsetAsync(cx().d_chal_dev, 0, sizeof(unsigned long long) * GKR_CHAL_WORDS * kChalSlots, cx().stream));
    return 0;
}
// the look-ahead tables go back to the arena (end of a proof, lane teardown)
void pre_release() {
    if (cx().aux) (void)hipStreamSynchronize(cx().aux);
    // class sums nobody waited for (a layer that found them unusable marks them spent WITHOUT waiting; an error return between the
    // launch and the wait): their kernel may still be running and reads the pyramids released below
    if (cx().ahead_in_flight || cx().ahead_K) (void)hipStreamSynchronize(cx().stream);
    cx().ahead_in_flight = false;
    for (auto& t : cx().pre_t) table_release_fwd(&t);
    for (DevTable* t : {&cx().ahead_pyrU, &cx().ahead_pyrU2, &cx().ahead_pyrTh}) table_release_fwd(t);
    cx().pre_K = cx().pre_S = nullptr;
    cx().ahead_K = cx().ahead_S = nullptr;
    cx().req_K = cx().req_S = nullptr;
    cx().nxt_K = cx().nxt_S = nullptr;
}
void lane_free() {
    (void)hipStreamSynchronize(cx().stream);
    pre_release();
    if (cx().aux) (void)hipStreamDestroy(cx().aux);
    if (cx().pre_done) (void)hipEventDestroy(cx().pre_done);
    if (cx().chk_fence) (void)hipEventDestroy(cx().chk_fence);
    cx().aux = nullptr;
    cx().pre_done = nullptr;
    cx().chk_fence = nullptr;
    if (cx().h_ahead) (void)hipHostFree(cx().h_ahead);
    if (cx().d_ahead_racc) (void)hipFree(cx().d_ahead_racc);
    if (cx().d_ahead_counter) (void)hipFree(cx().d_ahead_counter);
    cx().h_ahead = cx().d_ahead = cx().d_ahead_racc = nullptr;
    cx().d_ahead_counter = nullptr;
    if (cx().h_chal) (void)hipHostFree(cx().h_chal);
    cx().h_chal = cx().d_chal = nullptr;
    if (cx().d_chal_dev) (void)hipFree(cx().d_chal_dev);
    cx().d_chal_dev = nullptr;
    (void)hipFree(cx().d_partials);
    (void)hipFree(cx().d_racc);
    (void)hipFree(cx().d_sums);
    (void)hipHostFree(cx().h_sums);
    (void)hipFree(cx().d_small);
    (void)hipHostFree(cx().h_small);
    (void)hipHostFree(cx().h_round);
    (void)hipHostFree(cx().h_flag);
    (void)hipFree(cx().d_counter);
    if (cx().d_q) (void)hipFree(cx().d_q);
    cx().d_q = nullptr;
    for (auto& h : cx().h_q) {
        if (h) (void)hipHostFree(h);
        h = nullptr;
    }
    cx().d_q_cap = 0;
    if (cx().h_tail) (void)hipHostFree(cx().h_tail);
    cx().h_tail = cx().d_tail = nullptr;
    if (cx().h_spec) (void)hipHostFree(cx().h_spec);
    cx().h_spec = cx().d_spec = nullptr;
    if (cx().d_spec_racc) (void)hipFree(cx().d_spec_racc);
    cx().d_spec_racc = nullptr;
    if (cx().h_bad) (void)hipHostFree(cx().h_bad);
    cx().h_bad = cx().d_bad = nullptr;
    // the exchange buffers of a communicator lane
    if (cx().lc.d_buf) (void)hipFree(cx().lc.d_buf);
    if (cx().lc.h_buf) (void)hipHostFree(cx().lc.h_buf);
    if (cx().lc.h_tmp) (void)hipHostFree(cx().lc.h_tmp);
    if (cx().lc.h_cflag) (void)hipHostFree(cx().lc.h_cflag);
    cx().lc.d_buf = cx().lc.h_buf = cx().lc.h_tmp = nullptr;
    cx().lc.h_cflag = cx().lc.d_cflag = nullptr;
    cx().lc.buf_words = 0;
    (void)hipStreamDestroy(cx().stream);
    cx().stream = nullptr;
}
// Lanes that lost their session wait here for the next one (the hint-shaped one-shot calls create a session per
// call: stream, pinned hand-off buffers and counters are reused instead of re-created every time).
std::mutex g_lane_pool_mu;
std::vector<Ctx*> g_lane_pool;
const size_t kLanePoolMax = 16;
void lane_configure(Ctx* l) {
    l->device = g0.device;
    l->n_cu = g0.n_cu;
    l->max_grid = g0.max_grid;
    l->fold_grid = g0.fold_grid;
    l->fold_split = g0.fold_split;
    l->g_max = g0.g_max;
    l->g_max_auto = g0.g_max_auto;
    l->force_generic = g0.force_generic;
    l->lat_mode = g0.lat_mode;
    l->wide_mode = g0.wide_mode;
    l->wt_late_lj = g0.wt_late_lj;
    l->solo_boost = g0.solo_boost;
    l->claim_trick = g0.claim_trick;
    l->force_collective = g0.force_collective;
    l->host_tail = g0.host_tail;
    l->host_tail_solo = g0.host_tail_solo;
    l->host_tail_sharded = g0.host_tail_sharded;
    l->pyr_split = g0.pyr_split;
    l->prelaunch = g0.prelaunch;
    l->prelaunch_lg = g0.prelaunch_lg;
    l->pre_mode = g0.pre_mode;
    l->pre_start_lg = g0.pre_start_lg;
    l->spec = g0.spec;
    l->spec_lg = g0.spec_lg;
    l->ahead_mode = g0.ahead_mode;
    l->spec_max_m = g0.spec_max_m;
    l->coop = g0.coop;
    l->coop_lg = g0.coop_lg;
    l->coop_wgs = g0.coop_wgs;
    l->lag = g0.lag;
    l->prof.min_n = g0.prof.min_n;
}
// a new lane configured like the default one
Ctx* lane_create() {
    {
        std::lock_guard<std::mutex> lk(g_lane_pool_mu);
        if (!g_lane_pool.empty()) {
            Ctx* l = g_lane_pool.back();
            g_lane_pool.pop_back();
            lane_configure(l);
            std::lock_guard<std::mutex> ll(g_lanes_mu);
            g_lanes.push_back(l);
            return l;
        }
    }
    Ctx* l = new Ctx();
    static std::atomic<unsigned> created{1};   // (0: the default lane)
    l->ordinal = created.fetch_add(1, std::memory_order_relaxed);
    lane_configure(l);
    UseLane u(l);
    if (lane_alloc() != 0) {
        delete l;
        return nullptr;
    }
    l->ready = true;
    std::lock_guard<std::mutex> lk(g_lanes_mu);
    g_lanes.push_back(l);
    return l;
}
void lane_destroy(Ctx* l, bool pool = true) {
    {
        std::lock_guard<std::mutex> lk(g_lanes_mu);
        g_lanes.erase(std::remove(g_lanes.begin(), g_lanes.end(), l), g_lanes.end());
    }
    UseLane u(l);
    if (pool && !l->lc.comm && !l->lc.shm && l->lc.tick_lane < 0) {
        (void)hipStreamSynchronize(l->stream);
        prof_clear(l->prof);
        std::lock_guard<std::mutex> lk(g_lane_pool_mu);
        if (g_lane_pool.size() < kLanePoolMax) {
            g_lane_pool.push_back(l);
            return;
        }
    }
    lane_free();
    delete l;
}
void lane_pool_drain() {
    std::lock_guard<std::mutex> lk(g_lane_pool_mu);
    for (Ctx* l : g_lane_pool) {
        UseLane u(l);
        lane_free();
        delete l;
    }
    g_lane_pool.clear();
}

int ensure_ctx() {
    if (!g0.ready) {
        UseLane u(&g0);
        CHK(ctx_init(-1));
    }
    HIPCHK(hipSetDevice(g0.device));
    return 0;
}

// ---- device table arena (replaces poly/pool.go:69-126; no 2^24 cap) ---------------------------------
int table_alloc(DevTable* t, size_t cap) {
    if (cap == 0) cap = 1;
    std::lock_guard<std::mutex> lk(g_pool.mu);
    {
        auto it = g_pool.free_list.find(cap);
        if (it != g_pool.free_list.end() && !it->second.empty()) {
            t->base = it->second.back();
            t->cap = cap;
            it->second.pop_back();
            return 0;
        }
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, sizeof(uint4) * 2 * cap);
    if (e != hipSuccess) {
        // drop the cache and retry once
        (void)hipGetLastError();       // the failed attempt must not surface later as a stale "out of memory"
        for (auto& f : g_pool.free_list)
            for (uint4* b : f.second) (void)hipFree(b);
        g_pool.free_list.clear();
        e = hipMalloc(&p, sizeof(uint4) * 2 * cap);
        if (e != hipSuccess) return fail("hipMalloc of a %zu-element table failed: %s", cap, hipGetErrorString(e));
    }
    t->base = (uint4*)p;
    t->cap = cap;
    return 0;
}
// gkrhip_set_option("arena_check", 1) (tests): a buffer handed back goes to the NEXT caller of its size class, which may be
// another lane with another stream, so nothing of the releasing lane may still be using it.  With the option on, every release
// asks the lane's streams whether they are idle; a release with work still queued is counted (`arena_busy_releases`) and its
// call site named once on stderr.  An idle stream proves the release safe; a busy one is a site to look at.
void table_release(DevTable* t, int line = __builtin_LINE(), const char* file = __builtin_FILE()) {
    if (t->base) {
        if (g_arena_check.load(std::memory_order_relaxed)) {
            // (the look-ahead stream only ever holds k_cipher_pre: it reads the NEXT layer's assignment tables and writes pre_t,
            // which pre_release hands back behind a synchronisation of that stream.  Round 0 of the next layer queued ahead of
            // its point -- ahead_launch -- legitimately runs past the end of THIS layer on the lane's own stream: it touches the
            // next layer's tables and the lane's ahead_* tables only, so what has to be idle is everything queued in front of it.)
            // (un-sharded round loops wait for each kernel through the flag its LAST workgroup raises once every workgroup has
            // arrived with its stores drained, not through the stream: the runtime may retire that kernel microseconds after the
            // host has seen the flag.  A stream that drains within 2 ms was in that tail; one that does not -- a kernel still
            // polling for a challenge, a launch nobody waited for -- is the finding.)
            auto idle = [&]() {
                if (!cx().stream || hipStreamQuery(cx().stream) != hipErrorNotReady) return true;
                return cx().ahead_K && cx().chk_fence && hipEventQuery(cx().chk_fence) == hipSuccess;
            };
            bool busy = !idle();
            for (const double t0 = now_ms(); busy && now_ms() - t0 < 2.0;) busy = !idle();
            (void)hipGetLastError();
            // arena_check = 2: the buffer is also filled with 0xff behind everything queued on the lane's stream before the next
            // owner can have it -- whoever still reads it afterwards, or relies on what a recycled buffer used to hold, computes
            // with values above q and its proof differs from the oracle's (arena_check = 2 over the whole GPU suite)
            if (g_arena_check.load(std::memory_order_relaxed) >= 2 && cx().stream) {
                (void)hipMemsetAsync(t->base, 0xff, sizeof(uint4) * 2 * t->cap, cx().stream);
                (void)hipStreamSynchronize(cx().stream);
            }
            if (busy) {
                g_cnt_busy_releases.fetch_add(1, std::memory_order_relaxed);
                std::lock_guard<std::mutex> lk(g_busy_mu);
                const char* base = strrchr(file, '/');
                const std::pair<std::string, int> site(base ? base + 1 : file, line);
                if (std::find(g_busy_lines.begin(), g_busy_lines.end(), site) == g_busy_lines.end()) {
                    g_busy_lines.push_back(site);
                    fprintf(stderr, "gkrhip arena_check: a table is released at %s:%d while its lane's stream still has work queued\n", site.first.c_str(), line);
                }
            }
        }
        std::lock_guard<std::mutex> lk(g_pool.mu);
        g_pool.free_list[t->cap].push_back(t->base);
    }
    t->base = nullptr;
    t->cap = 0;
}
void table_release_fwd(DevTable* t) { table_release(t, -1, "pre_release"); }
void table_free(DevTable* t) {
    if (t->base) (void)hipFree(t->base);
    t->base = nullptr;
    t->cap = 0;
}

inline int grid_for(size_t n, int cap_blocks) {
    size_t b = (n + GKR_BLOCK - 1) / GKR_BLOCK;
    if (b < 1) b = 1;
    return (int)std::min<size_t>(b, (size_t)cap_blocks);
}

inline Fr to_dev(const E& e) {
    Fr r;
    memcpy(r.v, e.l, 32);
    return r;
}

// A DevTable that goes back to the arena when it leaves scope, so that an early error return strands nothing.  A
// table that is still held at that point belongs to a call that failed half-way: the lane's stream is drained first
// (kernels may still be using the buffer, and the arena hands it to the next caller).
struct ScopedTable : DevTable {
    ScopedTable() = default;
    ScopedTable(const ScopedTable&) = delete;
    ScopedTable& operator=(const ScopedTable&) = delete;
    ~ScopedTable() {
        if (base) {
            (void)hipStreamSynchronize(cx().stream);
            table_release(this);
        }
    }
};

// ---- boundary copies -----------------------------------------------------------------------------
// The device image of a host AoS table (32 bytes per element) has the size of a table, so the staging buffers come
// from the table arena and go back to it: the hint-shaped one-shot entry points, which move the same sizes every
// call, pay no hipMalloc/hipFree per call.
// host AoS -> device planes.  Staged through a device AoS buffer and transposed by k_aos_to_planes, which also
// checks the fr.Element invariant the lazy-reduction bounds of the round kernels rely on (every element < q).
// RegularIO (thread-local scope): the host images of this thread's uploads / downloads are REGULAR-form elements (the
// big.Int words of the hint interface) instead of Montgomery fr.Elements; the conversion rides on the transposition.
thread_local bool g_regular_io = false;
// A pre-launched or speculative kernel that does not see its challenge within a second gives up; the round loop then fails with
// g_chal_timeout set, and its caller runs the layer's rounds once more in safe mode (nothing queued ahead of its challenge):
// the layer's inputs are untouched by the rounds, so the retry produces the same transcript.
thread_local bool g_chal_timeout = false;
thread_local bool g_safe_mode = false;
thread_local bool g_corrupt_collect = false;      // fault injection (test_corrupt_sum): armed by the round loop for the hand-off it is about to collect
struct RegularIO {
    bool prev;
    explicit RegularIO(bool on = true) : prev(g_regular_io) { g_regular_io = on; }
    ~RegularIO() { g_regular_io = prev; }
};
int upload_table(DevTable* t, const uint64_t* host_aos, size_t n) {
    ScopedTable st;
    CHK(table_alloc(&st, n));
    uint4* stage = st.base;
    HIPCHK(hipMemcpyAsync(stage, host_aos, 32 * n, hipMemcpyHostToDevice, cx().stream));
    if (g_regular_io)
        hipLaunchKernelGGL(k_aos_to_planes<true>, dim3(grid_for(n, cx().max_grid)), dim3(GKR_BLOCK), 0, cx().stream, stage, t->planes(), n,
                           cx().d_bad, to_dev(hfr::R2));
    else
        hipLaunchKernelGGL(k_aos_to_planes<false>, dim3(grid_for(n, cx().max_grid)), dim3(GKR_BLOCK), 0, cx().stream, stage, t->planes(), n,
                           cx().d_bad, to_dev(hfr::ZERO));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(cx().stream));
    if (*(volatile unsigned int*)cx().h_bad) {
        *cx().h_bad = 0;
        return fail(g_regular_io ? "input table holds a value that is not below q"
                                 : "input table holds an element that is not a canonical fr.Element (limbs >= q)");
    }
    table_release(&st);
    return 0;
}
int download_table(const DevTable* t, uint64_t* host_aos, size_t n) {
    ScopedTable st;
    CHK(table_alloc(&st, n));
    uint4* stage = st.base;
    if (g_regular_io) {
        const E one = {{1, 0, 0, 0}};
        hipLaunchKernelGGL(k_planes_to_aos<true>, dim3(grid_for(n, cx().max_grid)), dim3(GKR_BLOCK), 0, cx().stream, t->cplanes(), stage, n,
                           to_dev(one));
    } else {
        hipLaunchKernelGGL(k_planes_to_aos<false>, dim3(grid_for(n, cx().max_grid)), dim3(GKR_BLOCK), 0, cx().stream, t->cplanes(), stage, n,
                           to_dev(hfr::ZERO));
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(host_aos, stage, 32 * n, hipMemcpyDeviceToHost, cx().stream));
    HIPCHK(hipStreamSynchronize(cx().stream));
    table_release(&st);
    return 0;
}

// ---- profiling helpers ---------------------------------------------------------------------------
hipEvent_t prof_event() {
    if (!cx().prof.pool.empty()) {
        hipEvent_t e = cx().prof.pool.back();
        cx().prof.pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

// ---- kernel launch wrappers ------------------------------------------------------------------------
int launch_fold(const DevTable* const* src, const DevTable* const* dst, int ntab, size_t mid, const E& r) {
    FoldArgs a;
    memset(&a, 0, sizeof a);
    for (int t = 0; t < ntab; t++) {
        a.src[t] = src[t]->cplanes();
        a.dst[t] = dst[t]->planes();
    }
    a.ntab = ntab;
    a.mid = mid;
    a.r = to_dev(r);
    const bool timed = 2 * mid >= cx().prof.min_n;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timed) {
        e0 = prof_event();
        e1 = prof_event();
        HIPCHK(hipEventRecord(e0, cx().stream));
    }
    // one single-table launch per table, one element per lane: measured (interleaved A/B in one process,
    // profiles/r01_fold_variants.txt) 6.4-6.6 TB/s on 2^24/2^25-element tables, against 5.7-6.1 TB/s for a
    // fused three-table launch and 4.6-5.8 TB/s for grid-stride loops over 8192 workgroups
    const dim3 grid(grid_for(mid, cx().fold_grid)), block(GKR_BLOCK);
    if (!cx().fold_split || mid < ((size_t)1 << 19)) {      // small tables are launch-bound: one launch for all of them
        switch (ntab) {
            case 1: hipLaunchKernelGGL(k_fold<1>, grid, block, 0, cx().stream, a); break;
            case 2: hipLaunchKernelGGL(k_fold<2>, grid, block, 0, cx().stream, a); break;
            case 3: hipLaunchKernelGGL(k_fold<3>, grid, block, 0, cx().stream, a); break;
            case 4: hipLaunchKernelGGL(k_fold<4>, grid, block, 0, cx().stream, a); break;
            case 5: hipLaunchKernelGGL(k_fold<5>, grid, block, 0, cx().stream, a); break;
            default: return fail("fold of %d tables not supported", ntab);
        }
    } else
    for (int t = 0; t < ntab; t++) {
        FoldArgs one;
        memset(&one, 0, sizeof one);
        one.src[0] = a.src[t];
        one.dst[0] = a.dst[t];
        one.ntab = 1;
        one.mid = mid;
        one.r = a.r;
        hipLaunchKernelGGL(k_fold<1>, grid, block, 0, cx().stream, one);
    }
    HIPCHK(hipGetLastError());
    if (timed) {
        HIPCHK(hipEventRecord(e1, cx().stream));
        cx().prof.fold_ev.emplace_back(e0, e1);
        cx().prof.fold_launches++;
        cx().prof.fold_bytes += 96.0 * ntab * (double)mid;
    }
    return 0;
}

template <int POWER, int ARITY, int NEV>
int launch_partial_eval_t(const DevTable* eq, const DevTable* const* x, size_t mid, const E& ark, unsigned mask, int* nblocks,
                          bool direct) {
    PartialEvalArgs a;
    memset(&a, 0, sizeof a);
    a.eq = eq->cplanes();
    for (int k = 0; k < ARITY; k++) a.x[k] = x[k]->cplanes();
    a.mid = mid;
    a.ark = to_dev(ark);
    a.mask = mask;
    a.partials = cx().d_partials;
    if (direct) {            // sums straight to the host (host-mapped buffer + flag), no reduction kernel, no copy
        a.racc = cx().d_racc;
        a.counter = cx().d_counter;
        a.host_out = cx().d_round;
        a.host_flag = cx().d_flag;
        a.seq = ++cx().seq;
    }
    const int grid = grid_for(mid, kPartialBlocks);
    hipLaunchKernelGGL((k_partial_eval<POWER, ARITY, NEV>), dim3(grid), dim3(GKR_BLOCK), 0, cx().stream, a);
    *nblocks = grid;
    return 0;
}

// wait for the sequence number a kernel of this lane publishes with its hand-off (host-mapped flag)
// `f` defaults to the lane's round flag; deadline_ms > 0 bounds the wait (collective paths: a peer that died or a
// collective that cannot make progress must surface as an error, never as a hang)
int wait_flag(unsigned int seq, volatile unsigned int* f = nullptr, double deadline_ms = 0, hipStream_t watch = nullptr) {
    if (!f) f = cx().h_flag;
    if (!watch) watch = cx().stream;      // the stream whose last operation raises the flag
    unsigned long spins = 0;
    Waiter w;
    double t0 = 0;
    while (*f != seq) {
        w.step();
        if ((++spins & 0xfffff) == 0) {              // every now and then: make sure the GPU is alive
            hipError_t e = hipStreamQuery(watch);
            if (e != hipSuccess && e != hipErrorNotReady) return fail("round kernel failed: %s", hipGetErrorString(e));
            if (e == hipSuccess && *f != seq) {
                const unsigned long long* dg = cx().h_round + 104;      // wait_challenge's note, if it abandoned the launch
                const unsigned long long* ds = cx().h_spec ? cx().h_spec : dg;      // the speculative launches' notes (two buffers)
                // (code 2 in the low half of a note: a workgroup's wait for its challenge ran out -- recoverable, see g_chal_timeout)
                g_chal_timeout = (dg[0] & 0xffffffffull) == 2 || (cx().h_spec && ((ds[580] & 0xffffffffull) == 2 || (ds[640 + 580] & 0xffffffffull) == 2));
                const unsigned long long note[12] = {dg[0], dg[1], dg[2], dg[3], cx().h_spec ? ds[580] : 0ull, cx().h_spec ? ds[581] : 0ull,
                                                     cx().h_spec ? ds[582] : 0ull, cx().h_spec ? ds[583] : 0ull, cx().h_spec ? ds[640 + 580] : 0ull,
                                                     cx().h_spec ? ds[640 + 581] : 0ull, cx().h_spec ? ds[640 + 582] : 0ull, cx().h_spec ? ds[640 + 583] : 0ull};
                cx().h_round[104] = 0;                                           // the notes are quoted once
                if (cx().h_spec) cx().h_spec[580] = cx().h_spec[640 + 580] = 0;
                const double now = now_ms();
                return fail("round kernel finished without publishing its result (flag %u, expected %u; challenge wait: code %llx after %llu ticks, "
                            "word %llx, seq %llu; speculative: %llx %llu %llx %llu | %llx %llu %llx %llu; host: last deferred launch seq %u %.1f ms ago, "
                            "published seq %u / %u / %u to slots 0 / 1 / 2 %.1f / %.1f / %.1f ms ago, slot tags now %llx %llx %llx)",
                            *f, seq, note[0], note[1], note[2], note[3], note[4], note[5], note[6], note[7], note[8], note[9], note[10], note[11],
                            cx().dbg_defer_seq, now - cx().dbg_defer_ms, cx().dbg_pub_seq[0], cx().dbg_pub_seq[1], cx().dbg_pub_seq[2],
                            now - cx().dbg_pub_ms[0], now - cx().dbg_pub_ms[1], now - cx().dbg_pub_ms[2], cx().h_chal[0] >> 32,
                            cx().h_chal[16] >> 32, cx().h_chal[32] >> 32);
            }
            if (deadline_ms > 0) {
                if (t0 == 0) t0 = now_ms();
                else if (now_ms() - t0 > deadline_ms) return fail("timed out after %.0f s waiting for the per-round exchange", deadline_ms * 1e-3);
            }
        }
    }
    __sync_synchronize();
    return 0;
}
