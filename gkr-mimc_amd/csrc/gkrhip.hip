// gkrhip.hip -- C ABI (include/gkrhip.h) and host drivers of the MI355X GKR/sumcheck prover.
//
// Host side mirrors the reference's orchestration (sumcheck/prover.go:46-144, gkr/prover.go:21-91)
// with every table-sized step replaced by a HIP kernel from kernels.hip.h.  Nothing here includes,
// links or calls anything under oracle/.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/gkrhip.h"
#include "fr_host.h"
#include "kernels.hip.h"
#include "cipher_round.hip.h"
#include "linear_round.hip.h"
#include "cipher_coop.hip.h"
#include "cipher_spec.hip.h"
#include "ntt.hip.h"
#include "fp_host.h"
#include "g1.hip.h"
// the heavy template kernels are instantiated in units of their own (kern_unit.hip, one per group of kernel_groups.h)
#define GKR_INST extern
#define GKR_INST_EXTERN
#include "kernel_groups.h"
#undef GKR_INST

using hfr::E;

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
namespace {
#include "host_ctx.hip.h"
#include "host_gates.hip.h"
#include "host_coll.hip.h"
#include "host_group.hip.h"
#include "host_sumcheck.hip.h"
#include "host_circuit.hip.h"
#include "host_ntt.hip.h"
#include "host_msm.hip.h"
}  // namespace

// ------------------------------------------------------------------------------------------------
// resident session
// ------------------------------------------------------------------------------------------------
struct gkrhip_session {
    int bN = 0;
    size_t n = 0;
    Circuit c;
    std::vector<DevTable> a;     // assignment; identity layers alias their input (no storage)
    std::vector<int> alias;      // alias[l] = layer whose table layer l shares, or l
    bool have_inputs = false, assigned = false;
    unsigned long long inputs_loaded = 0;
    Ctx* lane = nullptr;         // &g0 for sharded sessions, otherwise a lane of its own
    // Where the lane goes when the session is destroyed.  The one-shot calls (a session per call: the hint) hand theirs back to
    // the pool -- the next call finds it there instead of paying ~3 ms.  A session the caller created keeps its lane to the end
    // and then FREES it: lanes that survived a job of many proofs and were handed to a later job of a dozen ran that job 10 %
    // slower than lanes created for it (GMiMC bN = 22 x 12 behind bN = 20 x 56: 102 against 115 M hashes/s, same process, same
    // box; fresh buffers and a fresh stream for the pooled lanes do not help, taking the OLDEST pooled lanes helps once:
    // profiles/r05_order_probe.txt -- the runtime's assignment of streams to hardware queues is the suspect, not proven).
    bool pool_lane = false;
};

namespace {

int session_alloc(gkrhip_session* s) {
    const size_t L = s->c.size();
    s->a.assign(L, DevTable());
    s->alias.resize(L);
    for (size_t l = 0; l < L; l++) {
        s->alias[l] = (int)l;
        if (s->c[l].gate == GKRHIP_GATE_IDENTITY) {
            // a copy layer holds exactly its input's values (circuit/gates/copy.go:15-17); Prove never
            // mutates assignment tables here, so the copy is an alias.
            s->alias[l] = s->alias[s->c[l].in[0]];
            continue;
        }
        CHK(table_alloc(&s->a[l], s->n));
    }
    return 0;
}
const DevTable* session_table(const gkrhip_session* s, int l) { return &s->a[s->alias[l]]; }

int session_assign(gkrhip_session* s) {  // circuit/assignment.go:12-32
    if (!s->have_inputs) return fail("session has no inputs");
    for (size_t l = 0; l < s->c.size(); l++) {
        const Layer& lay = s->c[l];
        if (lay.gate < 0 || s->alias[l] != (int)l) continue;
        const DevTable* in[GKR_MAX_ARITY];
        for (size_t k = 0; k < lay.in.size(); k++) in[k] = session_table(s, lay.in[k]);
        CHK(gate_eval_dev(lay.gate, lay.ark, in, (int)lay.in.size(), &s->a[l], s->n));
    }
    HIPCHK(hipStreamSynchronize(cx().stream));
    s->assigned = true;
    return 0;
}

// The upload of the input tables and Circuit.Assign in one pipeline (what GkrProverHint.Call runs before Prove,
// prover/gadget/hints.go:219-220).  Every layer is element-wise, so the tables are cut into slices: while slice i's layers
// run, slice i + 1 crosses PCIe on the lane's second stream (and its pageable source is staged by this thread) -- the ~45 ms
// of upload at bN = 24 hide behind the ~46 ms of the 91 layers instead of preceding them.
int session_load_assign_sliced(gkrhip_session* s, const uint64_t* const* host, int n_in) {
    const size_t n = s->n;
    const int S = n >= ((size_t)1 << 22) ? 8 : n >= ((size_t)1 << 20) ? 4 : 1;
    if (S == 1 || n_in > GKR_MAX_ARITY * 2) {
        for (int k = 0; k < n_in; k++) CHK(upload_table(&s->a[k], host[k], n));
        s->have_inputs = true;
        return session_assign(s);
    }
    const size_t cnt = n / S;
    if (!cx().aux) {
        HIPCHK(hipStreamCreateWithFlags(&cx().aux, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&cx().pre_done, hipEventDisableTiming));
    }
    struct Events {
        std::vector<hipEvent_t> ev;
        ~Events() {
            for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        }
    } up, tr;
    up.ev.resize(S, nullptr);
    tr.ev.resize(2, nullptr);
    for (auto& e : up.ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : tr.ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    std::vector<ScopedTable> stage((size_t)2 * n_in);       // AoS staging of one slice per input, two sets
    for (auto& t : stage) CHK(table_alloc(&t, cnt));
    struct Drain {                                           // an error return must not leave copies in flight into the staging tables
        ~Drain() { (void)hipStreamSynchronize(cx().aux); }
    } drain;
    const bool regular = g_regular_io;
    for (int sl = 0; sl < S; sl++) {
        const size_t off = (size_t)sl * cnt;
        const int set = sl & 1;
        if (sl >= 2) HIPCHK(hipStreamWaitEvent(cx().aux, tr.ev[set], 0));      // the transposition of slice sl - 2 has read this staging set
        for (int k = 0; k < n_in; k++)
            HIPCHK(hipMemcpyAsync(stage[(size_t)set * n_in + k].base, host[k] + 4 * off, 32 * cnt, hipMemcpyHostToDevice, cx().aux));
        HIPCHK(hipEventRecord(up.ev[sl], cx().aux));
        HIPCHK(hipStreamWaitEvent(cx().stream, up.ev[sl], 0));
        for (int k = 0; k < n_in; k++) {
            const Planes dst{s->a[k].base + off, s->a[k].base + s->a[k].cap + off};
            if (regular)
                hipLaunchKernelGGL(k_aos_to_planes<true>, dim3(grid_for(cnt, cx().max_grid)), dim3(GKR_BLOCK), 0, cx().stream,
                                   stage[(size_t)set * n_in + k].base, dst, cnt, cx().d_bad, to_dev(hfr::R2));
            else
                hipLaunchKernelGGL(k_aos_to_planes<false>, dim3(grid_for(cnt, cx().max_grid)), dim3(GKR_BLOCK), 0, cx().stream,
                                   stage[(size_t)set * n_in + k].base, dst, cnt, cx().d_bad, to_dev(hfr::ZERO));
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(tr.ev[set], cx().stream));
        for (size_t l = 0; l < s->c.size(); l++) {           // circuit/assignment.go:12-32 on the slice
            const Layer& lay = s->c[l];
            if (lay.gate < 0 || s->alias[l] != (int)l) continue;
            const DevTable* in[GKR_MAX_ARITY];
            for (size_t k = 0; k < lay.in.size(); k++) in[k] = session_table(s, lay.in[k]);
            CHK(gate_eval_dev(lay.gate, lay.ark, in, (int)lay.in.size(), &s->a[l], cnt, off));
        }
    }
    HIPCHK(hipStreamSynchronize(cx().stream));
    if (*(volatile unsigned int*)cx().h_bad) {
        *cx().h_bad = 0;
        return fail(regular ? "input table holds a value that is not below q"
                            : "input table holds an element that is not a canonical fr.Element (limbs >= q)");
    }
    for (auto& t : stage) table_release(&t);
    s->have_inputs = true;
    s->inputs_loaded = ~0ull;
    s->assigned = true;
    return 0;
}

int session_prove(gkrhip_session* s, const E* qprime, E* flat) {  // gkr/prover.go:21-91
    if (!s->assigned) return fail("session is not assigned");
    ProofInFlight in_flight;
    const Circuit& c = s->c;
    const int L = (int)c.size(), bN = s->bN;
    std::vector<std::vector<E>> claims(L), qps(L), sc(L);
    std::vector<char> has_claims(L, 0);
    for (int l = 0; l < L; l++) {
        const size_t slots = std::max<size_t>(c[l].out.size(), 1);
        claims[l].assign(slots, hfr::ZERO);
        qps[l].assign(slots * std::max(bN, 1), hfr::ZERO);
    }
    for (int k = 0; k < bN; k++) qps[L - 1][k] = qprime[k];
    struct PreScope {            // the look-ahead tables are valid for this call only (the session's inputs may change)
        ~PreScope() { pre_release(); }
    } pre_scope;

    g_laps.start();
    for (int layer = L - 1; layer >= 0; layer--) {
        const Layer& lay = c[layer];
        if (lay.gate < 0) break;
        LAP("session: between layers");
        const int arity = (int)lay.in.size();
        const DevTable* X[GKR_MAX_ARITY];
        for (int k = 0; k < arity; k++) X[k] = session_table(s, lay.in[k]);
        // Look-ahead: the layer proven next.  If it is a single-point cipher layer, the products of its round 0 that
        // do not depend on its evaluation point are computed while this layer's small rounds leave the GPU idle.
        cx().req_K = cx().req_S = nullptr;
        cx().nxt_K = cx().nxt_S = nullptr;
        if (layer >= 1 && c[layer - 1].gate >= 0 && c[layer - 1].out.size() == 1 && c[layer - 1].in.size() == 2) {
            GateDesc gn;
            if (gate_get(c[layer - 1].gate, &gn) && gate_is_cipher2(gn)) {
                cx().req_K = session_table(s, c[layer - 1].in[0]);
                cx().req_S = session_table(s, c[layer - 1].in[1]);
                cx().req_ark = c[layer - 1].ark;
                cx().req_m = bN - shard_view().gamma;
                cx().nxt_K = cx().req_K;
                cx().nxt_S = cx().req_S;
            }
        }
        const int nev = gate_degree(lay.gate) + 2;
        sc[layer].assign((size_t)std::max(bN, 1) * nev, hfr::ZERO);
        std::vector<E> next_q(std::max(bN, 1));
        E fin[GKR_MAX_ARITY + 1];
        const int nq = layer == L - 1 ? 1 : (int)lay.out.size();
        const int ncl = has_claims[layer] ? (int)lay.out.size() : 0;
        LAP("session: layer prologue");
        CHK(sumcheck_prove_dev(lay.gate, lay.ark, arity, bN, X, qps[layer].data(), nq, claims[layer].data(), ncl,
                               sc[layer].data(), next_q.data(), fin, /*trust_claims=*/true));
        LAP("session: sumcheck returned");
        for (int i = 1; i <= arity; i++) {  // updateWithSumcheck, prover.go:66-90
            const int inp = lay.in[i - 1];
            const std::vector<int>& o = c[inp].out;
            const auto it = std::lower_bound(o.begin(), o.end(), layer);
            if (it == o.end() || *it != layer)
                return fail("circuit misformatted, In and Out are inconsistent between layers %d and %d", layer, inp);
            const size_t w = (size_t)(it - o.begin());
            has_claims[inp] = 1;
            claims[inp][w] = fin[i];
            for (int k = 0; k < bN; k++) qps[inp][w * bN + k] = next_q[k];
        }
    }
    // GkrProofToVec order (hints.go:236-271)
    size_t cur = 0;
    for (int l = 0; l < L; l++)
        if (c[l].gate >= 0) {
            const size_t cnt = (size_t)bN * (gate_degree(c[l].gate) + 2);
            memcpy(flat + cur, sc[l].data(), cnt * sizeof(E));
            cur += cnt;
        }
    for (int l = 0; l < L; l++) {
        memcpy(flat + cur, claims[l].data(), c[l].out.size() * sizeof(E));
        cur += c[l].out.size();
    }
    for (int l = 0; l < L; l++) {
        const size_t slots = l == L - 1 ? 1 : c[l].out.size();
        memcpy(flat + cur, qps[l].data(), slots * bN * sizeof(E));
        cur += slots * bN;
    }
    if (cur != proof_len(c, bN)) return fail("internal: flat proof length mismatch");
    if (getenv("GKRHIP_TRACE")) {       // where one proof's host time went (cumulative since the last profile reset)
        const Profile& p = cx().prof;
        fprintf(stderr, "rounds trace: hash %.2f wait %.2f launch %.2f other %.2f setup %.2f host-tail arithmetic %.2f ms\n", p.host_hash_ms,
                p.host_wait_ms, p.host_launch_ms, p.host_other_ms, p.setup_ms, p.tail_ms);
        g_laps.dump();
        for (int lg = 39; lg >= 0; lg--)
            if (p.cnt_lg[lg]) fprintf(stderr, "  2^%-2d pairs: %5llu rounds, wait %.1f us each\n", lg, (unsigned long long)p.cnt_lg[lg], 1e3 * p.wait_lg[lg] / p.cnt_lg[lg]);
    }
    return 0;
}

}  // namespace

// Visit every lane with its mutex held.  Lock order: default lane (g0.mu) -> lane list -> the lane.  A lane in
// the middle of a proof keeps its mutex for the whole proof, so the visit waits for proofs in flight; holding the
// lane list keeps a concurrent session destroy from freeing a lane under the visitor.
namespace {
// A host-buffer entry point (Fold, Evaluate, FoldedEqTable, EvalBatch, sumcheck.Prove on host tables, the one-shot
// verifiers, the wire-format helpers) borrows a lane from the pool for the duration of the call: calls from different
// host threads (goroutines locked to their OS threads by cgo) run concurrently, each on a stream of its own, instead
// of queueing on the default lane's mutex.
struct LaneLease {
    Ctx* l = nullptr;
    std::unique_lock<std::mutex> lk;
    Ctx* prev = nullptr;
    int acquire() {
        {
            std::lock_guard<std::mutex> g(g0.mu);
            CHK(ensure_ctx());
            l = lane_create();
        }
        if (!l) return fail("cannot create a lane for the call: %s", g_err.c_str());
        lk = std::unique_lock<std::mutex>(l->mu);
        prev = g_cur;
        g_cur = l;
        return 0;
    }
    ~LaneLease() {
        if (!l) return;
        g_cur = prev;
        lk.unlock();
        lane_destroy(l);
    }
};
#define LEASE_LANE()  \
    LaneLease lease;  \
    CHK(lease.acquire())

template <class F>
int for_each_lane(F&& fn) {
    std::lock_guard<std::mutex> lk0(g0.mu);
    CHK(ensure_ctx());
    std::lock_guard<std::mutex> ll(g_lanes_mu);
    for (Ctx* l : g_lanes) {
        std::unique_lock<std::mutex> lk;
        if (l != &g0) lk = std::unique_lock<std::mutex>(l->mu);
        UseLane u(l);
        CHK(fn(l));
    }
    return 0;
}
}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int gkrhip_init(int device_ordinal) {
#if defined(__x86_64__) && defined(__BMI2__) && defined(__ADX__)
    if (!__builtin_cpu_supports("bmi2") || !__builtin_cpu_supports("adx"))
        return fail("this build of libgkrhip.so needs a host CPU with BMI2 and ADX (every x86-64 server CPU since 2015)");
#endif
    std::lock_guard<std::mutex> lk(g0.mu);
    UseLane u(&g0);
    return ctx_init(device_ordinal);
}

void gkrhip_shutdown(void) {
    std::lock_guard<std::mutex> lk(g0.mu);
    if (!g0.ready) return;
    UseLane u(&g0);
    (void)hipSetDevice(cx().device);
    lane_free();
    lane_pool_drain();
    ntt_domains_free();
    {
        std::lock_guard<std::mutex> pl(g_pool.mu);
        for (auto& f : g_pool.free_list)
            for (uint4* b : f.second) (void)hipFree(b);
        g_pool.free_list.clear();
    }
    {
        std::lock_guard<std::mutex> ll(g_lanes_mu);
        g_lanes.erase(std::remove(g_lanes.begin(), g_lanes.end(), &g0), g_lanes.end());
    }
    delete cx().lag;
    cx().lag = nullptr;
    cx().ready = false;
    cx().device = -1;
}

int gkrhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* gkrhip_last_error(void) { return g_err.c_str(); }
// Thread-independent form: the message of the failure that returned `code` (every failing call returns a code of its own),
// copied into buf (NUL-terminated, truncated to cap); falls back to the calling thread's last message for a code the ring
// no longer holds.  Returns the full length of the message.
size_t gkrhip_last_error_r(int code, char* buf, size_t cap) {
    std::string m;
    if (!error_lookup(code, &m)) m = g_err;
    if (buf && cap) {
        const size_t n = std::min(m.size(), cap - 1);
        memcpy(buf, m.data(), n);
        buf[n] = 0;
    }
    return m.size();
}
const char* gkrhip_version(void) { return "gkrhip 0.3 (gfx950)"; }
// (gkrhip_build_id lives in build_id.cpp, a unit of its own)

int gkrhip_set_option(const char* key, long value) {
    static const char* keys[] = {"fold_grid", "fold_split", "g_max", "lat_mode", "wide_mode", "wt_late_lj", "claim_trick", "host_tail",
                                 "prelaunch", "prelaunch_lg", "lookahead", "coop", "spec", "spec_lg", "ahead", "solo_boost", "pyr_split",
                                 "coop_wgs", "coop_lg", "pre_start_lg"};
    // fault injection of the tests (host_sumcheck.hip.h): process-wide, fires once, -1 disarms
    if (!strcmp(key, "test_fail_after_prelaunch")) {
        g_test_fail_round.store((int)value);
        return 0;
    }
    if (!strcmp(key, "test_drop_challenge")) {
        g_test_drop_round.store((int)value);
        return 0;
    }
    if (!strcmp(key, "test_corrupt_sum")) {
        g_test_corrupt_left.store(1);
        g_test_corrupt_skip.store(0);
        g_test_corrupt_round.store((int)value);
        return 0;
    }
    if (!strcmp(key, "test_corrupt_skip")) {
        g_test_corrupt_skip.store((int)std::max(0L, value));
        return 0;
    }
    if (!strcmp(key, "test_corrupt_times")) {
        g_test_corrupt_left.store((int)std::max(1L, value));
        return 0;
    }
    if (!strcmp(key, "test_corrupt_tail")) {
        g_test_corrupt_tail.store((int)value);
        return 0;
    }
    // the prover's own checks (host_sumcheck.hip.h: sumcheck_closes).  layer_check (default 1): every sumcheck is held against
    // the verifier's identities before it is returned, and run again in safe mode if it does not close.  verify_after_prove
    // (default 0): the one-shot calls run gkr.Verify on their proof before returning it, as the reference's hint does in debug
    // builds (prover/gadget/hints.go:224-228) -- three more passes over the input and output tables.
    if (!strcmp(key, "layer_check")) {
        g_layer_check.store(value != 0);
        return 0;
    }
    if (!strcmp(key, "verify_after_prove")) {
        g_verify_after_prove.store(value != 0);
        return 0;
    }
    if (!strcmp(key, "arena_check")) {          // host_ctx.hip.h: table_release
        g_arena_check.store((int)value);
        if (value) g_cnt_busy_releases.store(0);
        return 0;
    }
    if (!strcmp(key, "wait_spin_us")) {         // host_ctx.hip.h: how host threads wait (-2: by the CPUs available; -1: always spin; n: spin n us, then sleep)
        g_wait_override.store((int)std::max(-2L, value));
        return 0;
    }
    if (!strcmp(key, "group_wait_us")) {        // ... and how long the first caller of a group waits for company
        g_group_wait_us.store((int)std::max(0L, std::min(100000L, value)));
        return 0;
    }
    if (!strcmp(key, "group_size")) {           // single calls that meet form proof groups of this many (0 | 1: never; gkrhip_mimc_session_prove)
        g_group_size.store((int)std::max(0L, std::min((long)GKR_GROUP_MAX, value)));
        return 0;
    }
    if (!strcmp(key, "msm_sort_levels")) {      // 0: by size, 1 | 2: forced (host_msm.hip.h); takes effect at the next MSM of a handle
        g_msm_sort_levels.store((int)value);
        return 0;
    }
    bool known = false;
    for (const char* k : keys) known = known || !strcmp(key, k);
    if (!known) return fail("unknown option %s", key);
    return for_each_lane([&](Ctx* l) {
        if (!strcmp(key, "fold_grid")) l->fold_grid = (int)std::max(64L, value);
        else if (!strcmp(key, "fold_split")) l->fold_split = value != 0;
        else if (!strcmp(key, "g_max")) {
            l->g_max = (int)std::max(8L, std::min(20L, value));
            l->g_max_auto = false;
        }
        else if (!strcmp(key, "lat_mode")) l->lat_mode = (int)value;
        else if (!strcmp(key, "wide_mode")) l->wide_mode = (int)value;
        else if (!strcmp(key, "wt_late_lj")) l->wt_late_lj = (int)value;
        else if (!strcmp(key, "claim_trick")) l->claim_trick = value != 0;
        else if (!strcmp(key, "host_tail")) l->host_tail = l->host_tail_solo = (int)std::max(0L, std::min((long)kHostTailMax, value));
        else if (!strcmp(key, "ahead")) l->ahead_mode = (int)value;
        else if (!strcmp(key, "prelaunch")) l->prelaunch = (int)value;
        else if (!strcmp(key, "prelaunch_lg")) l->prelaunch_lg = (int)std::max(0L, std::min(30L, value));
        else if (!strcmp(key, "lookahead")) l->pre_mode = (int)value;
        else if (!strcmp(key, "coop")) l->coop = (int)value;
        else if (!strcmp(key, "spec")) l->spec = (int)value;
        else if (!strcmp(key, "spec_lg")) l->spec_lg = (int)std::max(5L, std::min(16L, value));
        else if (!strcmp(key, "solo_boost")) l->solo_boost = (int)value;
        else if (!strcmp(key, "pyr_split")) l->pyr_split = (int)std::max(0L, std::min(20L, value));
        else if (!strcmp(key, "coop_wgs")) l->coop_wgs = (int)std::max(1L, std::min(4096L, value));      // (the tests: several iterations per workgroup)
        else if (!strcmp(key, "coop_lg")) l->coop_lg = (int)std::max(0L, std::min(20L, value));
        else if (!strcmp(key, "pre_start_lg")) l->pre_start_lg = (int)std::max(8L, std::min(30L, value));
        return 0;
    });
}

int gkrhip_mem_info(size_t* free_bytes, size_t* total_bytes) {
    std::lock_guard<std::mutex> lk(g0.mu);
    CHK(ensure_ctx());
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return 0;
}

// Lanes (stream, hand-off buffers, accumulators: ~3 ms to create) are leased per call from a pool that grows on demand: the first
// burst of concurrent calls -- the goroutines of ComputeGroth16Proof -- would pay for its lanes (12 ms for four, measured).  A host
// that knows its concurrency creates them ahead.
int gkrhip_reserve_lanes(int n) {
    if (n < 0 || n > (int)kLanePoolMax) return fail("gkrhip_reserve_lanes: %d lanes (0..%d, the size of the pool)", n, (int)kLanePoolMax);
    std::vector<Ctx*> got;
    int rc = 0;
    {
        std::lock_guard<std::mutex> g(g0.mu);
        CHK(ensure_ctx());
        for (int i = 0; i < n; i++) {
            Ctx* l = lane_create();       // pooled lanes first: the pool ends up holding at least n
            if (!l) {
                rc = fail("cannot create lane %d of %d: %s", i, n, g_err.c_str());
                break;
            }
            got.push_back(l);
        }
    }
    for (Ctx* l : got) lane_destroy(l);
    return rc;
}

// Page-locked host memory for the vectors a caller hands over on every proof (scalars of the MSMs, the a, b, c of computeH):
// an upload from pageable memory is staged by the runtime (measured 38 GB/s), one from these buffers is a plain DMA.
int gkrhip_host_alloc(void** out, size_t bytes) {
    if (!out) return fail("gkrhip_host_alloc: null argument");
    *out = nullptr;
    std::lock_guard<std::mutex> lk(g0.mu);
    CHK(ensure_ctx());
    hipError_t e = hipHostMalloc(out, std::max<size_t>(bytes, 1), hipHostMallocDefault);
    if (e != hipSuccess) return fail("hipHostMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    return 0;
}
void gkrhip_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int gkrhip_device_synchronize(void) {
    return for_each_lane([&](Ctx* l) {
        HIPCHK(hipStreamSynchronize(l->stream));
        return 0;
    });
}

int gkrhip_fold(uint64_t* table, size_t n, const uint64_t r[4]) {
    LEASE_LANE();
    if (n < 2 || (n & (n - 1))) return fail("Fold: table length %zu is not a power of two >= 2", n);
    ScopedTable t, o;
    CHK(table_alloc(&t, n));
    CHK(table_alloc(&o, n / 2));
    CHK(upload_table(&t, table, n));
    E re;
    memcpy(re.l, r, 32);
    const DevTable* src[1] = {&t};
    const DevTable* dst[1] = {&o};
    CHK(launch_fold(src, dst, 1, n / 2, re));
    CHK(download_table(&o, table, n / 2));
    table_release(&t);
    table_release(&o);
    return 0;
}

int gkrhip_evaluate(uint64_t out[4], const uint64_t* table, size_t n, const uint64_t* coords, int ncoords) {
    LEASE_LANE();
    if (n < 1 || (n & (n - 1))) return fail("Evaluate: table length %zu is not a power of two", n);
    if (((size_t)1 << ncoords) != n) return fail("Evaluate: table has %zu elements but %d coordinates were given", n, ncoords);
    LocalOnly lo;
    ScopedTable t;
    CHK(table_alloc(&t, n));
    CHK(upload_table(&t, table, n));
    E res;
    CHK(evaluate_dev(&t, ncoords, (const E*)coords, &res));
    memcpy(out, res.l, 32);
    table_release(&t);
    return 0;
}

int gkrhip_eq_table(uint64_t* out, const uint64_t* q, int bN, const uint64_t* mult_or_null) {
    LEASE_LANE();
    if (bN < 0 || bN > 30) return fail("eq table: bN %d out of range", bN);
    ScopedTable t;
    const size_t n = (size_t)1 << bN;
    CHK(table_alloc(&t, n));
    E seed = hfr::ONE;
    if (mult_or_null) memcpy(seed.l, mult_or_null, 32);
    CHK(build_eq(&t, (const E*)q, 1, bN, bN, &seed));
    CHK(download_table(&t, out, n));
    table_release(&t);
    return 0;
}

// poly.ChunkOfEqTable (poly/eq.go:61-89): chunk `chunk_id` of the table, written at its place in `table` (which has
// 2^bN elements).  The prefix weight over the chunk-id bits is scalar host work, the chunk itself one device pass.
int gkrhip_chunk_of_eq_table(uint64_t* table, size_t chunk_id, size_t chunk_size, const uint64_t* q, int bN,
                             const uint64_t* mult_or_null) {
    LEASE_LANE();
    if (bN < 0 || bN > 30) return fail("eq table: bN %d out of range", bN);
    const size_t n = (size_t)1 << bN;
    if (chunk_size < 1 || (chunk_size & (chunk_size - 1)) || chunk_size > n) return fail("ChunkOfEqTable: chunk size %zu", chunk_size);
    const size_t n_chunks = n / chunk_size;
    if (chunk_id >= n_chunks) return fail("ChunkOfEqTable: chunk %zu of %zu", chunk_id, n_chunks);
    int log_chunks = 0;
    while (((size_t)1 << log_chunks) < n_chunks) log_chunks++;
    const E* qe = (const E*)q;
    E r = hfr::ONE;
    if (mult_or_null) memcpy(r.l, mult_or_null, 32);
    for (int k = 0; k < log_chunks; k++) {           // eq.go:74-82: bit k of the chunk id <-> qPrime[logNChunks-k-1]
        const E& rho = qe[log_chunks - k - 1];
        r = hfr::mul(r, ((chunk_id >> k) & 1) ? rho : hfr::sub(hfr::ONE, rho));
    }
    const int m = bN - log_chunks;
    ScopedTable t;
    CHK(table_alloc(&t, chunk_size));
    CHK(build_eq(&t, qe + log_chunks, 1, m, m, &r));
    CHK(download_table(&t, table + 4 * chunk_id * chunk_size, chunk_size));
    table_release(&t);
    return 0;
}

int gkrhip_gate_eval_batch(int gate, const uint64_t* ark_or_null, uint64_t* res, const uint64_t* const* xs, int arity,
                           size_t n) {
    LEASE_LANE();
    if (arity < 1 || arity > GKR_MAX_ARITY) return fail("arity %d not supported (1..%d)", arity, GKR_MAX_ARITY);
    if (n < 1) return fail("EvalBatch: empty tables");
    ScopedTable in[GKR_MAX_ARITY], out;
    const DevTable* inp[GKR_MAX_ARITY];
    for (int k = 0; k < arity; k++) {
        CHK(table_alloc(&in[k], n));
        CHK(upload_table(&in[k], xs[k], n));
        inp[k] = &in[k];
    }
    CHK(table_alloc(&out, n));
    E ark = hfr::ZERO;
    if (ark_or_null) memcpy(ark.l, ark_or_null, 32);
    CHK(gate_eval_dev(gate, ark, inp, arity, &out, n));
    CHK(download_table(&out, res, n));
    for (int k = 0; k < arity; k++) table_release(&in[k]);
    table_release(&out);
    return 0;
}

int gkrhip_sumcheck_prove(int gate, const uint64_t* ark_or_null, int arity, int bN, const uint64_t* const* X,
                          const uint64_t* qprimes, int nq, const uint64_t* claims, int nclaims, uint64_t* proof,
                          uint64_t* challenges, uint64_t* final_claims) {
    LEASE_LANE();
    if (bN < 0 || bN > 30) return fail("bN %d out of range", bN);
    if (arity < 1 || arity > GKR_MAX_ARITY) return fail("arity %d not supported (1..%d)", arity, GKR_MAX_ARITY);
    const size_t n = (size_t)1 << bN;
    LocalOnly lo;
    ScopedTable tabs[GKR_MAX_ARITY];
    const DevTable* X_[GKR_MAX_ARITY];
    for (int k = 0; k < arity; k++) {
        CHK(table_alloc(&tabs[k], n));
        CHK(upload_table(&tabs[k], X[k], n));
        X_[k] = &tabs[k];
    }
    E ark = hfr::ZERO;
    if (ark_or_null) memcpy(ark.l, ark_or_null, 32);
    const int rc = sumcheck_prove_dev(gate, ark, arity, bN, X_, (const E*)qprimes, nq, (const E*)claims, nclaims,
                                      (E*)proof, (E*)challenges, (E*)final_claims);
    for (int k = 0; k < arity; k++) table_release(&tabs[k]);
    return rc;
}

// sumcheck.Verify (sumcheck/verifier.go:28-65): scalar work only (Fiat-Shamir hashing, evaluations of the round
// polynomials), so it runs on the host and needs no GPU.  Returns 0 = accepted, 1 + i = round i's check
// P_i(0) + P_i(1) == expected failed (gkrhip_last_error has the reference's message).
int gkrhip_sumcheck_verify(const uint64_t* claims, int nclaims, const uint64_t* proof, int bN, int ncoeffs, uint64_t* challenges,
                           uint64_t final_claim[4], uint64_t recomb_chal[4]) {
    if (nclaims < 1) return fail("sumcheck.Verify: no claim (the reference indexes claims[0] of an empty slice and panics)");
    if (bN < 0 || ncoeffs < 1 || ncoeffs > 64) return fail("sumcheck.Verify: bad proof shape (%d rounds of %d coefficients)", bN, ncoeffs);
    const E* cl = (const E*)claims;
    const E* pr = (const E*)proof;
    // recombineMultiClaims (:58-65): with >= 1 claims the challenge is always drawn, also for a single claim
    const E recomb = hfr::mimc_hash(cl, (size_t)nclaims);
    E expected = hfr::eval_univariate(cl, nclaims, recomb);
    for (int i = 0; i < bN; i++) {
        const E* p = pr + (size_t)i * ncoeffs;
        const E actual = hfr::add(hfr::eval_univariate(p, ncoeffs, hfr::ZERO), hfr::eval_univariate(p, ncoeffs, hfr::ONE));
        if (actual != expected) {
            (void)fail("at round %d verifier eval at 0 + 1 = %s || expected = %s", i, hfr::to_decimal(actual).c_str(),
                       hfr::to_decimal(expected).c_str());
            return 1 + i;
        }
        const E r = hfr::mimc_hash(p, (size_t)ncoeffs);
        if (challenges) memcpy(challenges + 4 * (size_t)i, r.l, 32);
        expected = hfr::eval_univariate(p, ncoeffs, r);
    }
    if (final_claim) memcpy(final_claim, expected.l, 32);
    if (recomb_chal) memcpy(recomb_chal, recomb.l, 32);
    return 0;
}

size_t gkrhip_mimc_proof_len(int bN) { return (size_t)822 * bN + 183 + (size_t)184 * bN; }

static int session_create_for(gkrhip_session** out, const Circuit& circ, int bN);

int gkrhip_mimc_session_create(gkrhip_session** out, int bN) {
    std::lock_guard<std::mutex> lk(g0.mu);
    CHK(ensure_ctx());
    return session_create_for(out, mimc_circuit(), bN);
}

int gkrhip_session_create(gkrhip_session** out, const gkrhip_layer* layers, int n_layers, int bN) {
    std::lock_guard<std::mutex> lk(g0.mu);
    CHK(ensure_ctx());
    Circuit c;
    CHK(circuit_from_layers(layers, n_layers, &c));
    return session_create_for(out, c, bN);
}

int gkrhip_gmimc_t2_circuit(gkrhip_layer* layers_out, int capacity) {
    const std::vector<gkrhip_layer> v = gmimc_t2_layers();
    if (layers_out) {
        if (capacity < (int)v.size()) return fail("gmimc_t2_circuit: capacity %d < %zu layers", capacity, v.size());
        memcpy(layers_out, v.data(), v.size() * sizeof(gkrhip_layer));
    }
    return (int)v.size();
}

int gkrhip_gmimc_circuit(int t, gkrhip_layer* layers_out, int capacity, int* input_map_out) {
    std::vector<gkrhip_layer> v;
    std::vector<int> map;
    if (const int rc = gmimc_layers(t, &v, &map)) return rc;
    if (input_map_out)
        for (size_t k = 0; k < map.size(); k++) input_map_out[k] = map[k];
    if (layers_out) {
        if (capacity < (int)v.size()) return fail("gmimc_circuit: capacity %d < %zu layers", capacity, v.size());
        memcpy(layers_out, v.data(), v.size() * sizeof(gkrhip_layer));
    }
    return (int)v.size();
}

int gkrhip_gmimc_hash_circuit(int t, int nblocks, gkrhip_layer* layers_out, int capacity, int* input_map_out) {
    std::vector<gkrhip_layer> v;
    std::vector<int> map;
    if (const int rc = gmimc_hash_layers(t, nblocks, &v, &map)) return rc;
    if (input_map_out)
        for (size_t k = 0; k < map.size(); k++) input_map_out[k] = map[k];
    if (layers_out) {
        if (capacity < (int)v.size()) return fail("gmimc_hash_circuit: capacity %d < %zu layers", capacity, v.size());
        memcpy(layers_out, v.data(), v.size() * sizeof(gkrhip_layer));
    }
    return (int)v.size();
}

int gkrhip_gate_register(const gkrhip_gate_desc* desc, int* gate_id) { return gate_register(desc, gate_id); }
int gkrhip_gate_lookup(int gate_id, gkrhip_gate_desc* desc_out) {
    GateDesc g;
    if (!gate_get(gate_id, &g)) return fail("unknown gate id %d", gate_id);
    if (desc_out) {
        memset(desc_out, 0, sizeof *desc_out);
        strncpy(desc_out->id, g.id.c_str(), sizeof desc_out->id - 1);
        desc_out->n_in = g.n_in;
        desc_out->sum_mask = g.mask;
        desc_out->power = g.power;
    }
    return 0;
}

size_t gkrhip_session_proof_len(const gkrhip_session* s) { return s ? proof_len(s->c, s->bN) : 0; }
int gkrhip_session_num_inputs(const gkrhip_session* s) {
    int n = 0;
    while (s && n < (int)s->c.size() && s->c[n].gate < 0) n++;
    return n;
}

static int session_create_for(gkrhip_session** out, const Circuit& circ, int bN) {
    if (bN < 0 || bN > 32) return fail("bN %d out of range", bN);
    if (bN < gc.gamma) return fail("bN %d is smaller than log2(world) = %d", bN, gc.gamma);
    if (bN - gc.gamma > 28) return fail("a shard of 2^%d entries does not fit one GPU", bN - gc.gamma);
    gkrhip_session* s = new gkrhip_session();
    s->bN = bN;                                   // global number of variables
    s->n = (size_t)1 << (bN - gc.gamma);          // entries of this rank's shard
    s->c = circ;
    // with a communicator installed, session i runs on communicator lane i mod nlanes (lane k pairs with lane k
    // of the peers: create the sessions in the same order on every rank); otherwise it gets a lane of its own
    if (!gc.lanes.empty()) s->lane = gc.lanes[gc.next_lane++ % gc.lanes.size()];
    else s->lane = g0.force_collective ? &g0 : lane_create();
    if (!s->lane) {
        delete s;
        return fail("cannot create a lane for the session: %s", g_err.c_str());
    }
    const int rc = session_alloc(s);
    if (rc != 0) {
        for (auto& t : s->a) table_free(&t);
        if (s->lane != &g0 && gc.lanes.empty()) lane_destroy(s->lane);
        delete s;
        return rc;
    }
    *out = s;
    return 0;
}

// session entry points run on the session's lane: they take that lane's mutex only, so sessions with lanes
// of their own proceed concurrently
#define SESSION_ENTER(s)                                  \
    if (!(s) || !(s)->lane) return fail("null session");   \
    HIPCHK(hipSetDevice(g0.device));                      \
    std::lock_guard<std::mutex> lk((s)->lane->mu);        \
    UseLane ul((s)->lane)

static int session_load_assign(gkrhip_session* s, const uint64_t* const* host, int n_in) {
    SESSION_ENTER(s);
    return session_load_assign_sliced(s, host, n_in);
}

int gkrhip_mimc_session_load_inputs(gkrhip_session* s, const uint64_t* in0, const uint64_t* in1) {
    SESSION_ENTER(s);
    if (s->c.size() < 2 || s->c[0].gate >= 0 || s->c[1].gate >= 0 || (s->c.size() > 2 && s->c[2].gate < 0))
        return fail("load_inputs: the circuit does not have exactly two input layers");
    CHK(upload_table(&s->a[0], in0, s->n));
    CHK(upload_table(&s->a[1], in1, s->n));
    s->have_inputs = true;
    s->assigned = false;
    return 0;
}

int gkrhip_session_load_input(gkrhip_session* s, int input_index, const uint64_t* table) {
    SESSION_ENTER(s);
    if (input_index < 0 || input_index >= (int)s->c.size() || s->c[input_index].gate >= 0)
        return fail("layer %d is not an input layer", input_index);
    CHK(upload_table(&s->a[input_index], table, s->n));
    s->inputs_loaded |= 1ull << (input_index & 63);
    int n_in = 0;
    while (n_in < (int)s->c.size() && s->c[n_in].gate < 0) n_in++;
    s->have_inputs = s->inputs_loaded == ((n_in >= 64) ? ~0ull : ((1ull << n_in) - 1));
    s->assigned = false;
    return 0;
}

int gkrhip_mimc_session_synth_inputs(gkrhip_session* s, uint64_t index_stride, uint64_t index_offset) {
    SESSION_ENTER(s);
    int n_in = 0;
    while (n_in < (int)s->c.size() && s->c[n_in].gate < 0) n_in++;
    for (int l = 0; l < n_in; l++) {
        hipLaunchKernelGGL(k_random_fr_array, dim3(grid_for(s->n, cx().max_grid)), dim3(GKR_BLOCK), 0, cx().stream,
                           s->a[l].planes(), s->n, (unsigned long long)index_stride, (unsigned long long)index_offset);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(cx().stream));
    s->have_inputs = true;
    s->assigned = false;
    return 0;
}

int gkrhip_mimc_session_assign(gkrhip_session* s) {
    SESSION_ENTER(s);
    return session_assign(s);
}

static int session_prove_on_its_lane(gkrhip_session* s, const uint64_t* qprime, uint64_t* flat) {
    SESSION_ENTER(s);
    const int rc = session_prove(s, (const E*)qprime, (E*)flat);
    if (rc != 0) shm_abort();     // a sharded proof that fails on this rank must not leave the peers waiting
    return rc;
}
int gkrhip_mimc_session_prove_group(int n, gkrhip_session* const* ss, const uint64_t* const* qprimes, uint64_t* const* flats, int* rcs);

// ---- single calls that meet form groups ---------------------------------------------------------------------------------------
// The reference proves independent statements from a goroutine each: many host threads inside gkrhip_mimc_session_prove at once.
// When enough of them prove SMALL statements (2^18..2^21 entries: where proof groups pay) the calls that arrive
// together are proven as a group by the first of them -- the others wait for their result -- so that the host gets the groups'
// throughput through the reference's own call shape.  A caller waits for company at most group_wait_us (option; default 10 ms); a call that finds none
// runs as it always did.  Option "group_size" (default 3; 0 or 1: never).  Same transcripts either way.
namespace {
std::atomic<int> g_small_callers{0};               // threads inside gkrhip_mimc_session_prove with a small un-sharded session
// Where groups pay: the job must be bound by the GPU's dispatch, not by the hosts' hashing -- a group's proofs take turns on ONE host
// thread.  Fourteen callers with statements of 2^7..2^22 entries lost a quarter of their throughput to grouping (7 125 -> 5 064 proofs
// in 45 s, tools/stress.py; GMiMC lanes 9 890 -> 7 446): few callers and tiny statements are host-bound, every caller needs its
// own core.  So: from 24 callers on, statements of 2^18..2^21 entries.
const int kCoalesceFromCallers = 24, kCoalesceMinBn = 18, kCoalesceMaxBn = 21;
struct Forming {
    int n = 0, want = 0, refs = 0;
    gkrhip_session* s[GKR_GROUP_MAX];
    const uint64_t* q[GKR_GROUP_MAX];
    uint64_t* flat[GKR_GROUP_MAX];
    int rc[GKR_GROUP_MAX];
    bool closed = false, done = false;
    std::condition_variable cv_full, cv_done;
};
std::mutex g_forming_mu;
std::unordered_map<unsigned long long, Forming*> g_forming;      // by shape: the group that is waiting for company
}  // namespace

int gkrhip_mimc_session_prove(gkrhip_session* s, const uint64_t* qprime, uint64_t* flat) {
    if (!s || !s->lane) return fail("null session");
    const int want = g_group_size.load(std::memory_order_relaxed);
    // (a lane with a serial-latency path forced on -- the tests do that -- proves as it is told to: those paths queue kernels that poll
    // for the host, which a group never does)
    const Ctx* ln = s->lane;
    const bool forced = ln->prelaunch >= 2 || ln->pre_mode >= 2 || ln->spec >= 2 || ln->coop >= 2;
    const bool sharded = ln == &g0 || ln->lc.comm || ln->lc.shm || ln->lc.tick_lane >= 0;
    const bool small = want >= 2 && !sharded && s->bN >= kCoalesceMinBn && s->bN <= kCoalesceMaxBn && !forced && !t_group && !g_regular_io &&
                       !g_safe_mode && !g_local_only;
    if (!small) return session_prove_on_its_lane(s, qprime, flat);
    struct Count {
        Count() { g_small_callers.fetch_add(1, std::memory_order_relaxed); }
        ~Count() { g_small_callers.fetch_sub(1, std::memory_order_relaxed); }
    } count;
    if (g_small_callers.load(std::memory_order_relaxed) < kCoalesceFromCallers) return session_prove_on_its_lane(s, qprime, flat);
    // groups form among statements of one shape: the size and the circuit's gates and wiring, layer by layer (FNV-1a)
    unsigned long long key = 1469598103934665603ull ^ (unsigned long long)s->bN;
    for (const Layer& l : s->c) {
        key = (key ^ (unsigned long long)(unsigned)l.gate) * 1099511628211ull;
        for (int in : l.in) key = (key ^ (unsigned long long)(unsigned)in) * 1099511628211ull;
        key = (key ^ 0xffull) * 1099511628211ull;
    }
    std::unique_lock<std::mutex> lk(g_forming_mu);
    auto it = g_forming.find(key);
    if (it != g_forming.end()) {
        Forming* f = it->second;
        bool twice = false;
        for (int i = 0; i < f->n; i++) twice = twice || f->s[i] == s;
        if (!twice) {                     // join: the first caller proves, this one waits for its result
            const int me = f->n++;
            f->s[me] = s;
            f->q[me] = qprime;
            f->flat[me] = flat;
            f->refs++;
            if (f->n >= f->want) {
                f->closed = true;
                g_forming.erase(it);
                f->cv_full.notify_all();
            }
            f->cv_done.wait(lk, [&] { return f->done; });
            const int rc = f->rc[me];
            if (--f->refs == 0) delete f;
            lk.unlock();
            if (rc != 0) {                // the message of this proof's failure, on this thread too
                std::string m;
                if (error_lookup(rc, &m)) g_err = m;
            }
            return rc;
        }
        lk.unlock();                      // the same session from two threads: they take turns on its lane, as ever
        return session_prove_on_its_lane(s, qprime, flat);
    }
    Forming* f = new Forming();
    f->want = std::min(want, GKR_GROUP_MAX);
    f->n = 1;
    f->refs = 1;
    f->s[0] = s;
    f->q[0] = qprime;
    f->flat[0] = flat;
    g_forming[key] = f;
    f->cv_full.wait_for(lk, std::chrono::microseconds(g_group_wait_us.load(std::memory_order_relaxed)), [&] { return f->closed; });
    if (!f->closed) {
        f->closed = true;
        g_forming.erase(key);
    }
    const int n = f->n;
    lk.unlock();
    int rc0;
    if (n == 1) {
        rc0 = session_prove_on_its_lane(s, qprime, flat);
        f->rc[0] = rc0;
    } else {
        for (int i = 0; i < n; i++) f->rc[i] = 0;
        const int rc = gkrhip_mimc_session_prove_group(n, f->s, f->q, f->flat, f->rc);
        if (rc != 0) {                     // the group call itself was refused (no proof has a code of its own): every caller gets that code
            bool any = false;
            for (int i = 0; i < n; i++) any = any || f->rc[i] != 0;
            if (!any)
                for (int i = 0; i < n; i++) f->rc[i] = rc;
        }
        rc0 = f->rc[0];
        g_cnt_coalesced.fetch_add((unsigned long long)n, std::memory_order_relaxed);
    }
    lk.lock();
    f->done = true;
    f->cv_done.notify_all();
    if (--f->refs == 0) delete f;
    lk.unlock();
    return rc0;
}

// gkr.Prove for n sessions of the same shape from ONE host thread, in lock-step (host_group.hip.h): every proof is the proof
// gkrhip_mimc_session_prove returns for its session and point -- same transcript, bit for bit -- but the round kernels of the n
// proofs go to the GPU as one launch.  For many small proofs in flight (BASELINE config 2: bN = 20): with groups of 3 a third of the launches,
// each with three times the work, where the dispatch of tiny kernels is the bound.  Sessions with lanes of their own (un-sharded),
// all different; rcs[i] (may be NULL) receives proof i's code; returns 0 or the first failing proof's code.
int gkrhip_mimc_session_prove_group(int n, gkrhip_session* const* ss, const uint64_t* const* qprimes, uint64_t* const* flats, int* rcs) {
    if (n < 1 || n > GKR_GROUP_MAX) return fail("prove_group: %d proofs (1..%d)", n, GKR_GROUP_MAX);
    if (!ss || !qprimes || !flats) return fail("prove_group: null argument");
    for (int i = 0; i < n; i++) {
        if (!ss[i] || !ss[i]->lane) return fail("prove_group: null session (%d)", i);
        if (ss[i]->lane == &g0 || ss[i]->lane->lc.comm || ss[i]->lane->lc.shm || ss[i]->lane->lc.tick_lane >= 0 || shard_view().world > 1)
            return fail("prove_group: session %d is sharded (its rounds are exchanged with its peers, in step with them)", i);
        if (!flats[i] || (ss[i]->bN > 0 && !qprimes[i])) return fail("prove_group: null buffer (%d)", i);
        for (int j = 0; j < i; j++)
            if (ss[j] == ss[i]) return fail("prove_group: session %d is given twice", i);
    }
    HIPCHK(hipSetDevice(g0.device));
    // every lane's mutex, in address order (two groups over the same sessions cannot deadlock)
    std::vector<Ctx*> lanes(n);
    for (int i = 0; i < n; i++) lanes[i] = ss[i]->lane;
    std::vector<Ctx*> order(lanes);
    std::sort(order.begin(), order.end());
    std::vector<std::unique_lock<std::mutex>> locks;
    for (Ctx* l : order) locks.emplace_back(l->mu);
    // The group queues on the streams of ONE of its lanes -- the one whose hardware queue the fewest groups under way use: the
    // runtime deals its hardware queues to streams in creation order, and groups that all took their first lane's stream would
    // share a few queues (sessions created in a row, groups of 4: queues 0, 4, 8, 12 of 16 -- measured 62 M hashes/s at bN = 20 x 60
    // in flight against 75-80 M for groups of 3 and 6, whose first lanes happen to cover all the queues; profiles/r06_proof_groups.txt).
    // What a lane has in its own stream is waited for first.
    struct StreamSwap {
        std::vector<Ctx*>& lanes;
        std::vector<std::pair<hipStream_t, hipStream_t>> own;
        bool swapped = false;
        int lead = 0, slot = -1;
        explicit StreamSwap(std::vector<Ctx*>& l) : lanes(l) {}
        ~StreamSwap() {
            if (slot >= 0) {
                std::lock_guard<std::mutex> g(g_group_queue_mu);
                g_group_queue_use[slot]--;
            }
            if (!swapped) return;
            (void)hipStreamSynchronize(own[(size_t)lead].first);
            if (own[(size_t)lead].second) (void)hipStreamSynchronize(own[(size_t)lead].second);
            for (size_t i = 0; i < lanes.size(); i++) {
                lanes[i]->stream = own[i].first;
                lanes[i]->aux = own[i].second;
            }
        }
    } sw(lanes);
    for (int i = 0; i < n; i++) {
        HIPCHK(hipStreamSynchronize(lanes[i]->stream));
        if (lanes[i]->aux) HIPCHK(hipStreamSynchronize(lanes[i]->aux));
    }
    int lead = 0;
    {
        const int nq = std::max(1, std::min(64, hw_queue_count()));
        std::lock_guard<std::mutex> g(g_group_queue_mu);
        for (int i = 1; i < n; i++)
            if (g_group_queue_use[lanes[i]->ordinal % nq] < g_group_queue_use[lanes[lead]->ordinal % nq]) lead = i;
        sw.slot = (int)(lanes[lead]->ordinal % nq);
        g_group_queue_use[sw.slot]++;
        sw.lead = lead;
    }
    for (int i = 0; i < n; i++) sw.own.emplace_back(lanes[i]->stream, lanes[i]->aux);
    for (int i = 0; i < n; i++) {
        lanes[i]->stream = sw.own[(size_t)lead].first;
        lanes[i]->aux = sw.own[(size_t)lead].second;
    }
    sw.swapped = true;
    Group g;
    g.proofs.resize((size_t)n);
    for (int i = 0; i < n; i++) {
        gkrhip_session* s = ss[i];
        const E* q = (const E*)qprimes[i];
        E* flat = (E*)flats[i];
        g.proofs[(size_t)i].body = [s, q, flat]() {
            UseLane ul(s->lane);
            return session_prove(s, q, flat);
        };
    }
    CHK(group_run(g));
    int first = 0;
    for (int i = 0; i < n; i++) {
        if (rcs) rcs[i] = g.proofs[(size_t)i].rc;
        if (!first) first = g.proofs[(size_t)i].rc;
    }
    return first;
}

int gkrhip_mimc_session_outputs(gkrhip_session* s, uint64_t* outputs) {
    SESSION_ENTER(s);
    if (!s->assigned) return fail("session is not assigned");
    return download_table(session_table(s, (int)s->c.size() - 1), outputs, s->n);
}

int gkrhip_mimc_session_evaluate_layer(gkrhip_session* s, int layer, const uint64_t* coords, uint64_t out[4]) {
    SESSION_ENTER(s);
    if (layer < 0 || layer >= (int)s->c.size()) return fail("layer %d out of range", layer);
    if (!s->assigned && layer >= 2) return fail("session is not assigned");
    E res;
    CHK(evaluate_dev(session_table(s, layer), s->bN, (const E*)coords, &res));
    memcpy(out, res.l, 32);
    return 0;
}

void gkrhip_mimc_session_destroy(gkrhip_session* s) {
    if (!s) return;
    if (g0.ready) (void)hipSetDevice(g0.device);
    if (s->lane) {
        {
            std::lock_guard<std::mutex> lk(s->lane->mu);
            UseLane ul(s->lane);
            (void)hipStreamSynchronize(cx().stream);
            // back to the arena, not to the driver: the next session of the same size (one-shot calls from the
            // hint, one per proof) reuses the buffers instead of paying ~1 s of hipMalloc/hipFree for 50 GB;
            // table_alloc drops the cache when an allocation fails
            for (auto& t : s->a) table_release(&t);
        }
        bool owned = s->lane != &g0;
        for (Ctx* l : gc.lanes) owned = owned && l != s->lane;   // communicator lanes outlive their sessions
        if (owned) lane_destroy(s->lane, s->pool_lane);
    }
    delete s;
}

// regular: in0, in1, qprime, flat and the outputs are REGULAR-form values (the hint interface's big.Int words,
// prover/gadget/hints.go:202-205,224-231) instead of Montgomery fr.Elements; the bulk conversions ride on the boundary
// transposition of the tables, the few proof elements are converted on the host.
static int prove_mimc_oneshot(int bN, const uint64_t* in0, const uint64_t* in1, const uint64_t* qprime, uint64_t* flat,
                              uint64_t* outputs_or_null, bool regular) {
    RegularIO rio(regular);
    std::vector<E> qp_m;
    if (regular && bN > 0) {
        qp_m.resize(bN);
        for (int i = 0; i < bN; i++) {
            E v;
            memcpy(v.l, qprime + 4 * i, 32);
            if (!hfr::is_canonical(v)) return fail("qPrime holds a value that is not below q");
            qp_m[i] = hfr::mul(v, hfr::R2);
        }
        qprime = (const uint64_t*)qp_m.data();
    }
    const bool trace = getenv("GKRHIP_TRACE") != nullptr;     // stage times on stderr
    const double t_0 = now_ms();
    gkrhip_session* s = nullptr;
    CHK(gkrhip_mimc_session_create(&s, bN));
    s->pool_lane = true;
    const double t_c = now_ms();
    int rc;
    {
        const uint64_t* ins[2] = {in0, in1};
        rc = session_load_assign(s, ins, 2);
    }
    const double t_l = now_ms();
    if (trace && rc == 0) (void)hipStreamSynchronize(s->lane->stream);
    const double t_a = now_ms();
    // The output table is final once the assignment is: its download (transposition + 2^bN x 32 bytes over PCIe into
    // pageable memory) runs on a lane of its own from a second host thread while this thread proves.
    int rc_out = 0;
    std::string err_out;
    std::thread dl;
    Ctx* dl_lane = nullptr;
    if (rc == 0 && outputs_or_null && s->lane != &g0) {
        std::lock_guard<std::mutex> lk(g0.mu);
        dl_lane = lane_create();
    }
    if (dl_lane) {
        dl = std::thread([&]() {
            std::lock_guard<std::mutex> lk(dl_lane->mu);
            UseLane u(dl_lane);
            RegularIO rio2(regular);
            rc_out = hipSetDevice(g0.device) == hipSuccess ? download_table(session_table(s, (int)s->c.size() - 1), outputs_or_null, s->n)
                                                           : fail("hipSetDevice failed");
            if (rc_out) err_out = g_err;
        });
    }
    const double t_p0 = now_ms();
    if (rc == 0) {
        // the regular-form scope is for the BOUNDARY images only (inputs above, outputs and the flat proof below): inside
        // the proof every host value is a Montgomery element -- a sharded proof uploads gathered elements (small_table)
        RegularIO inside(false);
        rc = gkrhip_mimc_session_prove(s, qprime, flat);
        // hints.go:224-228 (`if debug`): the hint verifies its own proof -- the claims on the input and output tables included,
        // which the per-layer checks inside Prove cannot see
        if (rc == 0 && g_verify_after_prove.load(std::memory_order_relaxed)) {
            rc = gkrhip_mimc_session_verify(s, qprime, flat);
            if (rc > 0) rc = fail("GKR proof was wrong - Bug in proof generation - gkr.Verify rejected it (code %d)", rc);
        }
    }
    const double t_p = now_ms();
    if (dl.joinable()) {
        dl.join();
        lane_destroy(dl_lane);
        if (rc == 0 && rc_out) {
            g_err = err_out;
            rc = rc_out;
        }
    } else if (rc == 0 && outputs_or_null) {
        rc = gkrhip_mimc_session_outputs(s, outputs_or_null);
    }
    if (rc == 0 && regular) {
        const E one = {{1, 0, 0, 0}};
        const size_t len = proof_len(s->c, bN);
        E* f = (E*)flat;
        for (size_t i = 0; i < len; i++) f[i] = hfr::mul(f[i], one);
    }
    const double t_j = now_ms();
    gkrhip_mimc_session_destroy(s);
    if (trace)
        fprintf(stderr, "oneshot bN=%d: create %.1f load %.1f assign %.1f spawn-download %.1f prove %.1f join+convert %.1f destroy %.1f ms\n", bN,
                t_c - t_0, t_l - t_c, t_a - t_l, t_p0 - t_a, t_p - t_p0, t_j - t_p, now_ms() - t_j);
    return rc;
}
int gkrhip_gkr_prove_mimc(int bN, const uint64_t* in0, const uint64_t* in1, const uint64_t* qprime, uint64_t* flat,
                          uint64_t* outputs_or_null) {
    return prove_mimc_oneshot(bN, in0, in1, qprime, flat, outputs_or_null, false);
}
int gkrhip_gkr_prove_mimc_regular(int bN, const uint64_t* in0, const uint64_t* in1, const uint64_t* qprime, uint64_t* flat,
                                  uint64_t* outputs_or_null) {
    return prove_mimc_oneshot(bN, in0, in1, qprime, flat, outputs_or_null, true);
}

// Circuit.Assign + gkr.Prove for any circuit of library gates on host tables, in one call (the generic form of
// gkrhip_gkr_prove_mimc).
int gkrhip_gkr_prove(const gkrhip_layer* layers, int n_layers, int bN, const uint64_t* const* inputs, int n_inputs,
                     const uint64_t* qprime, uint64_t* flat, uint64_t* outputs_or_null) {
    gkrhip_session* s = nullptr;
    CHK(gkrhip_session_create(&s, layers, n_layers, bN));
    s->pool_lane = true;
    int rc = 0;
    if (gkrhip_session_num_inputs(s) != n_inputs) rc = fail("gkr.Prove: the circuit has %d input layers, %d tables were given", gkrhip_session_num_inputs(s), n_inputs);
    if (rc == 0) rc = session_load_assign(s, inputs, n_inputs);
    if (rc == 0) rc = gkrhip_mimc_session_prove(s, qprime, flat);
    if (rc == 0 && g_verify_after_prove.load(std::memory_order_relaxed)) {      // see prove_mimc_oneshot
        rc = gkrhip_mimc_session_verify(s, qprime, flat);
        if (rc > 0) rc = fail("GKR proof was wrong - Bug in proof generation - gkr.Verify rejected it (code %d)", rc);
    }
    if (rc == 0 && outputs_or_null) rc = gkrhip_mimc_session_outputs(s, outputs_or_null);
    const std::string err = g_err;
    gkrhip_mimc_session_destroy(s);
    if (rc != 0) g_err = err;
    return rc;
}

// ---- wire-format helpers (prover/gadget/hints.go) ---------------------------------------------------------
static int convert_inplace(uint64_t* data, size_t n, const E& factor) {
    LEASE_LANE();
    if (n == 0) return 0;
    ScopedTable st;                       // an AoS image of n elements has the size of a table: from the arena
    CHK(table_alloc(&st, n));
    uint4* d = st.base;
    HIPCHK(hipMemcpyAsync(d, data, 32 * n, hipMemcpyHostToDevice, cx().stream));
    hipLaunchKernelGGL(k_convert_aos, dim3(grid_for(n, cx().max_grid)), dim3(GKR_BLOCK), 0, cx().stream, d, n, to_dev(factor));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(data, d, 32 * n, hipMemcpyDeviceToHost, cx().stream));
    HIPCHK(hipStreamSynchronize(cx().stream));
    table_release(&st);
    return 0;
}
int gkrhip_to_regular(uint64_t* data, size_t n) {
    const E one = {{1, 0, 0, 0}};
    return convert_inplace(data, n, one);
}
int gkrhip_from_regular(uint64_t* data, size_t n) { return convert_inplace(data, n, hfr::R2); }

int gkrhip_mimc_permutation_batch(uint64_t* out, const uint64_t* x, const uint64_t* key, size_t n) {
    LEASE_LANE();
    if (n == 0) return 0;
    ScopedTable tx, tk, to;
    CHK(table_alloc(&tx, n));
    CHK(table_alloc(&tk, n));
    CHK(table_alloc(&to, n));
    CHK(upload_table(&tx, x, n));
    CHK(upload_table(&tk, key, n));
    hipLaunchKernelGGL(k_mimc_permutation, dim3(grid_for(n, cx().max_grid)), dim3(GKR_BLOCK), 0, cx().stream, tx.cplanes(),
                       tk.cplanes(), to.planes(), n);
    HIPCHK(hipGetLastError());
    CHK(download_table(&to, out, n));
    table_release(&tx);
    table_release(&tk);
    table_release(&to);
    return 0;
}

// ---- gkr.Verify (gkr/verifier.go:15-132, sumcheck/verifier.go:28-65) on a flat proof.  The O(N) parts --
// MultiLin.Evaluate of the output table and of the two input tables -- run on the device through `eval`;
// the rest is scalar work on <= 822*bN + 183 + 184*bN elements.  Returns 0 = accepted, > 0 = rejected.
static int verify_flat(const Circuit& c, int bN, const E* flat, const E* qprime,
                       const std::function<int(int, const E*, E*)>& eval) {
    const int L = (int)c.size();
    std::vector<const E*> sc(L, nullptr), claims(L), qps(L);
    size_t cur = 0;
    for (int l = 0; l < L; l++)
        if (c[l].gate >= 0) {
            sc[l] = flat + cur;
            cur += (size_t)bN * (gate_degree(c[l].gate) + 2);
        }
    for (int l = 0; l < L; l++) {
        claims[l] = flat + cur;
        cur += c[l].out.size();
    }
    for (int l = 0; l < L; l++) {
        qps[l] = flat + cur;
        cur += (l == L - 1 ? 1 : c[l].out.size()) * bN;
    }
    if (memcmp(qprime, qps[L - 1], (size_t)bN * sizeof(E)) != 0) return 1;   // verifier.go:25-30
    E top;
    CHK(eval(L - 1, qprime, &top));                                        // verifier.go:36
    std::vector<E> next_q(std::max(bN, 1));
    for (int layer = L - 1; layer >= 0; layer--) {
        if (c[layer].gate < 0) break;
        const E* cl = layer == L - 1 ? &top : claims[layer];
        const int ncl = layer == L - 1 ? 1 : (int)c[layer].out.size();
        const int nc = gate_degree(c[layer].gate) + 2;
        // sumcheck.Verify
        const E recomb = hfr::mimc_hash(cl, (size_t)ncl);
        E expected = hfr::eval_univariate(cl, ncl, recomb);
        for (int i = 0; i < bN; i++) {
            const E* p = sc[layer] + (size_t)i * nc;
            const E s01 = hfr::add(hfr::eval_univariate(p, nc, hfr::ZERO), hfr::eval_univariate(p, nc, hfr::ONE));
            if (s01 != expected) return 10 + layer;
            next_q[i] = hfr::mimc_hash(p, (size_t)nc);
            expected = hfr::eval_univariate(p, nc, next_q[i]);
        }
        // testSumcheck (verifier.go:61-117)
        E sub[GKR_MAX_ARITY];
        for (size_t k = 0; k < c[layer].in.size(); k++) {
            const int inp = c[layer].in[k];
            const std::vector<int>& o = c[inp].out;
            const size_t r_at = (size_t)(std::lower_bound(o.begin(), o.end(), layer) - o.begin());
            if (memcmp(qps[inp] + r_at * bN, next_q.data(), (size_t)bN * sizeof(E)) != 0) return 1000 + layer;
            sub[k] = claims[inp][r_at];
        }
        GateDesc gd;
        if (!gate_get(c[layer].gate, &gd)) return fail("verify: unknown gate %d", c[layer].gate);
        const E gate_val = gate_eval_host(gd, c[layer].gate == GKRHIP_GATE_IDENTITY ? hfr::ZERO : c[layer].ark, sub);
        std::vector<E> eqs(ncl);
        for (int i = 0; i < ncl; i++) eqs[i] = hfr::eval_eq(qps[layer] + (size_t)i * bN, next_q.data(), bN);
        const E eq_eval = hfr::eval_univariate(eqs.data(), ncl, recomb);
        if (hfr::mul(gate_val, eq_eval) != expected) return 2000 + layer;
    }
    for (int l = 0; l < L && c[l].gate < 0; l++) {   // testInitialRound (verifier.go:120-132)
        E actual;
        CHK(eval(l, qps[l], &actual));
        if (actual != claims[l][0]) return 3000 + l;
    }
    return 0;
}

int gkrhip_gkr_verify_mimc(int bN, const uint64_t* flat, const uint64_t* in0, const uint64_t* in1, const uint64_t* outputs,
                           const uint64_t* qprime) {
    LEASE_LANE();
    LocalOnly lo;
    if (bN < 0 || bN > 28) return fail("bN %d out of range", bN);
    const size_t n = (size_t)1 << bN;
    const Circuit c = mimc_circuit();
    ScopedTable t[3];
    const uint64_t* host[3] = {in0, in1, outputs};
    for (int i = 0; i < 3; i++) {
        CHK(table_alloc(&t[i], n));
        CHK(upload_table(&t[i], host[i], n));
    }
    auto eval = [&](int layer, const E* pt, E* out) -> int {
        const DevTable* tab = layer == 0 ? &t[0] : layer == 1 ? &t[1] : &t[2];
        return evaluate_dev(tab, bN, pt, out);
    };
    const int rc = verify_flat(c, bN, (const E*)flat, (const E*)qprime, eval);
    for (int i = 0; i < 3; i++) table_release(&t[i]);
    if (rc > 0) fail("gkr.Verify rejected the proof (code %d)", rc);
    return rc;
}

int gkrhip_gkr_verify(const gkrhip_layer* layers, int n_layers, int bN, const uint64_t* flat, const uint64_t* const* inputs,
                      int n_inputs, const uint64_t* outputs, const uint64_t* qprime) {
    LEASE_LANE();
    LocalOnly lo;
    if (bN < 0 || bN > 28) return fail("bN %d out of range", bN);
    Circuit c;
    CHK(circuit_from_layers(layers, n_layers, &c));
    int n_in = 0;
    while (n_in < (int)c.size() && c[n_in].gate < 0) n_in++;
    if (n_inputs != n_in) return fail("gkr.Verify: the circuit has %d input layers, %d tables were given", n_in, n_inputs);
    const size_t n = (size_t)1 << bN;
    std::vector<ScopedTable> t(n_in + 1);
    for (int i = 0; i <= n_in; i++) {
        CHK(table_alloc(&t[i], n));
        CHK(upload_table(&t[i], i < n_in ? inputs[i] : outputs, n));
    }
    auto eval = [&](int layer, const E* pt, E* out) -> int {
        return evaluate_dev(layer < n_in ? &t[layer] : &t[n_in], bN, pt, out);
    };
    const int rc = verify_flat(c, bN, (const E*)flat, (const E*)qprime, eval);
    if (rc > 0) fail("gkr.Verify rejected the proof (code %d)", rc);
    return rc;
}

int gkrhip_mimc_session_verify(gkrhip_session* s, const uint64_t* qprime, const uint64_t* flat) {
    SESSION_ENTER(s);
    if (!s->assigned) return fail("session is not assigned");
    auto eval = [&](int layer, const E* pt, E* out) -> int { return evaluate_dev(session_table(s, layer), s->bN, pt, out); };
    const int rc = verify_flat(s->c, s->bN, (const E*)flat, (const E*)qprime, eval);
    if (rc > 0) fail("gkr.Verify rejected the proof (code %d)", rc);
    return rc;
}

// computeH (prover/gadget/prove.go:308-359): a, b, c hold n Montgomery elements each (the solved R1CS vectors), zero-padded
// here to the domain (cardinality = a power of two >= n; 0 = the next power of two, as fft.NewDomain chooses it); h receives
// `cardinality` REGULAR-form values in the order the reference returns them (coefficients of H, bit-reversed positions).
int gkrhip_compute_h(uint64_t* h, const uint64_t* a, const uint64_t* b, const uint64_t* c, size_t n, size_t cardinality) {
    LEASE_LANE();
    if (n < 1) return fail("computeH: empty vectors");
    size_t card = cardinality;
    if (card == 0) {
        card = 1;
        while (card < n) card <<= 1;
    }
    if ((card & (card - 1)) || card < n) return fail("computeH: domain cardinality %zu is not a power of two >= %zu", card, n);
    int logn = 0;
    while (((size_t)1 << logn) < card) logn++;
    if (logn == 0) {      // a domain of one point: generator 1, coset shift u = -1 (order 2): h = (a*b - c) * (-2)^-1, regular form
        E ea, eb, ec;
        memcpy(ea.l, a, 32);
        memcpy(eb.l, b, 32);
        memcpy(ec.l, c, 32);
        if (!hfr::is_canonical(ea) || !hfr::is_canonical(eb) || !hfr::is_canonical(ec)) return fail("computeH: input is not a canonical fr.Element");
        const E r = to_plain(hfr::mul(hfr::sub(hfr::mul(ea, eb), ec), hfr::pow_q_minus_2(hfr::sub(hfr::ZERO, hfr::from_u64(2)))));
        memcpy(h, r.l, 32);
        return 0;
    }
    ScopedTable t[3];
    DevTable* tp[3];
    const uint64_t* src[3] = {a, b, c};
    for (int i = 0; i < 3; i++) {
        CHK(table_alloc(&t[i], card));
        CHK(upload_table(&t[i], src[i], n));
        if (card > n) {
            hipLaunchKernelGGL(k_ntt_zero, dim3(grid_for(card - n, cx().max_grid)), dim3(GKR_BLOCK), 0, cx().stream, t[i].planes(), n, card);
            HIPCHK(hipGetLastError());
        }
        tp[i] = &t[i];
    }
    CHK(compute_h_dev(tp, logn, nullptr));
    CHK(download_table(&t[0], h, card));
    for (int i = 0; i < 3; i++) table_release(&t[i]);
    return 0;
}

// computeH on device-resident synthetic vectors (a[i] = Montgomery(i + 1), b[i] = Montgomery(2 i + 3), c = a * b pointwise
// truncated to n = 2^logn - 1 entries... the values do not matter for the timing): `iters` runs after `warmup`;
// *avg_ms = HIP-event time of one computeH on the device (transforms and pointwise step; no PCIe), *passes = passes over HBM
// per computeH, *bytes = HBM bytes those passes move (32 B read + 32 B written per element and array named by a pass).
int gkrhip_bench_compute_h(int logn, int warmup, int iters, double* avg_ms, int* passes, double* bytes) {
    std::lock_guard<std::mutex> lk(g0.mu);
    CHK(ensure_ctx());
    if (logn < 1 || logn > 27 || iters < 1) return fail("bench_compute_h: bad arguments");
    const size_t n = (size_t)1 << logn;
    ScopedTable t[3];
    DevTable* tp[3];
    for (int i = 0; i < 3; i++) {
        CHK(table_alloc(&t[i], n));
        tp[i] = &t[i];
    }
    auto fill = [&]() {
        for (int i = 0; i < 3; i++)
            hipLaunchKernelGGL(k_random_fr_array, dim3(grid_for(n, cx().max_grid)), dim3(GKR_BLOCK), 0, cx().stream, t[i].planes(), n,
                               (unsigned long long)(i + 1), (unsigned long long)(7 * i + 1));
        return hipGetLastError();
    };
    int np = 0;
    double by = 0;
    for (int i = 0; i < warmup; i++) {
        HIPCHK(fill());
        CHK(compute_h_dev(tp, logn, &np, &by));
    }
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    double tot = 0;
    for (int i = 0; i < iters; i++) {
        HIPCHK(fill());
        HIPCHK(hipEventRecord(e0, cx().stream));
        CHK(compute_h_dev(tp, logn, &np, &by));
        HIPCHK(hipEventRecord(e1, cx().stream));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        tot += ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_ms = tot / iters;
    if (passes) *passes = np;
    if (bytes) *bytes = by;
    for (int i = 0; i < 3; i++) table_release(&t[i]);
    return 0;
}

// ---- multi-scalar multiplications (gnark-crypto's MultiExp at prover/gadget/prove.go:76,91,189,202,221 on G1, :277 on G2) -----
// One implementation over the group (F: the device field policy of g1.hip.h, HF: its host twin, B: the handle type).
}  // extern "C"
namespace {
template <class B>
int abi_bases_create(B** out, const uint64_t* points, size_t n, int w16) {
    if (!out || (n && !points)) return fail("bases_create: null argument");
    LEASE_LANE();
    return bases_upload(out, points, n, w16);
}
template <class F, class B>
int abi_bases_generate(B** out, const uint64_t* base, const uint64_t* scalars, size_t n, int flags) {
    if (!out || !base || (n && !scalars)) return fail("bases_generate: null argument");
    LEASE_LANE();
    B* b = nullptr;
    CHK(bases_alloc(&b, n, F::W16));
    const int rc = batch_mul_dev<F>(b->d_points, base, scalars, n, flags);
    if (rc) {
        bases_free(b);
        return rc;
    }
    *out = b;
    return 0;
}
int abi_bases_read(const MsmBases* b, uint64_t* out, size_t first, size_t count) {
    if (!b || !out) return fail("bases_read: null argument");
    if (first > b->n || count > b->n - first) return fail("bases_read: [%zu, %zu) outside %zu points", first, first + count, b->n);
    LEASE_LANE();
    if (count) {
        HIPCHK(hipMemcpyAsync(out, b->d_points + (size_t)2 * b->w16 * first, count * 32 * b->w16, hipMemcpyDeviceToHost, cx().stream));
        HIPCHK(hipStreamSynchronize(cx().stream));
    }
    return 0;
}
int abi_set_window(MsmBases* b, int c) {
    if (!b) return fail("msm_set_window: null handle");
    if (c != 0 && (c < 2 || c > 16)) return fail("msm: window size %d outside 2..16 (0 = automatic)", c);
    std::lock_guard<std::mutex> lk(b->mu);
    b->c_forced = c;
    return 0;
}
// fixed-base tables of a handle: c = 0 chosen from the number of points, 8..22 forced, -1 drops the tables (the handle's MSMs go
// back to the per-window sort)
template <class F>
int abi_precompute(MsmBases* b, int c) {
    if (!b) return fail("msm_precompute: null handle");
    LEASE_LANE();
    std::lock_guard<std::mutex> lk(b->mu);
    if (c < 0) {
        HIPCHK(hipStreamSynchronize(cx().stream));
        b->fb.release();
        return 0;
    }
    return msm_fb_prepare<F>(b, c);
}
template <class F, class HF>
int abi_msm(uint64_t* out_affine, MsmBases* b, const uint64_t* scalars, size_t n, int flags) {
    if (!out_affine || !b || (n && !scalars)) return fail("msm: null argument");
    LEASE_LANE();
    return msm_run<F, HF>(b, scalars, n, flags, out_affine);
}
// MSM of 2^logn synthetic device-resident bases [k_i] G and scalars (both pseudo-random below q), timed with HIP events
template <class F, class HF, class B>
int abi_bench_msm(const uint64_t* gen_image, int logn, int c_or_0, int warmup, int iters, double* avg_ms, double phase_ms[5], int* c_used,
                  double* host_tail_ms, uint64_t* result_or_null, bool fixed_base = false, double* precompute_ms = nullptr) {
    if (logn < 0 || logn > 26 || iters < 1 || !avg_ms) return fail("bench_msm: bad arguments");
    LEASE_LANE();
    const size_t n = (size_t)1 << logn;
    B* b = nullptr;
    CHK(bases_alloc(&b, n, F::W16));
    struct Guard {
        B* b;
        uint4* s = nullptr;
        MsmTimes tm;
        ~Guard() {
            for (hipEvent_t e : tm.ev)
                if (e) (void)hipEventDestroy(e);
            if (s) (void)hipFree(s);
            bases_free(b);
        }
    } g{b};
    b->c_forced = fixed_base ? 0 : c_or_0;
    if (!fixed_base) CHK(msm_work_prepare(&b->w, n, c_or_0, F::W16));
    HIPCHK(hipMalloc((void**)&g.s, n * 32));
    hipLaunchKernelGGL(k_msm_synth_scalars, dim3(grid_for(n, 4096)), dim3(GKR_BLOCK), 0, cx().stream, g.s, n, 0x1234567u);
    {
        MsmArgs a;
        memset(&a, 0, sizeof a);
        a.scalars = g.s;
        a.n = n;
        AffT<F> gen;
        memcpy(&gen, gen_image, sizeof gen);
        hipLaunchKernelGGL(k_ec_batch_scalar_mul<F>, dim3((unsigned)((n + GKR_BLOCK - 1) / GKR_BLOCK)), dim3(GKR_BLOCK), 0, cx().stream, a, gen, b->d_points);
    }
    hipLaunchKernelGGL(k_msm_synth_scalars, dim3(grid_for(n, 4096)), dim3(GKR_BLOCK), 0, cx().stream, g.s, n, 0x7654321u);
    HIPCHK(hipGetLastError());
    for (hipEvent_t& e : g.tm.ev) HIPCHK(hipEventCreate(&e));
    if (fixed_base) {      // the tables: once per key, outside the MSM's time (reported beside it)
        HIPCHK(hipStreamSynchronize(cx().stream));
        const double t0 = now_ms();
        CHK(msm_fb_prepare<F>(b, c_or_0));
        if (precompute_ms) *precompute_ms = now_ms() - t0;
    }
    MsmWork* wk = fixed_base ? &b->fb.w : &b->w;
    auto run = [&](MsmTimes* tm) { return fixed_base ? msm_fb_dev<F>(b, g.s, n, 0, tm) : msm_dev<F>(b, g.s, n, 0, tm); };
    for (int i = 0; i < warmup; i++) CHK(run(nullptr));
    HIPCHK(hipStreamSynchronize(cx().stream));
    double tot = 0, ph[5] = {0, 0, 0, 0, 0}, tail = 0;
    g.tm.on = true;
    hfp::AffH<HF> r{HF::zero(), HF::zero()};
    for (int i = 0; i < iters; i++) {
        CHK(run(&g.tm));
        HIPCHK(hipEventSynchronize(g.tm.ev[5]));
        const double t0 = now_ms();
        r = msm_host_tail<HF>(wk);
        tail += now_ms() - t0;
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, g.tm.ev[0], g.tm.ev[5]));
        tot += ms;
        for (int k = 0; k < 5; k++) {
            HIPCHK(hipEventElapsedTime(&ms, g.tm.ev[k], g.tm.ev[k + 1]));
            ph[k] += ms;
        }
    }
    *avg_ms = tot / iters;
    if (phase_ms)
        for (int k = 0; k < 5; k++) phase_ms[k] = ph[k] / iters;
    if (c_used) *c_used = wk->c;
    if (host_tail_ms) *host_tail_ms = tail / iters;
    if (result_or_null) memcpy(result_or_null, &r, sizeof r);
    return 0;
}
// the generators gnark-crypto uses (bn254.Generators): g1Gen = (1, 2); g2Gen below (regular form, converted at first use)
const uint64_t* g1_generator() {
    static const hfp::E two = hfp::add(hfp::ONE, hfp::ONE);
    static const hfp::AffH<hfp::HFp> g{hfp::ONE, two};
    return (const uint64_t*)&g;
}
const uint64_t* g2_generator() {
    static const hfp::AffH<hfp::HFp2> g = [] {
        // X = 10857046999023057135944570762232829481370756359578518086990519993285655852781
        //   + 11559732032986387107991004021392285783925812861821192530917403151452391805634 u,
        // Y = 8495653923123431417604973247489272438418190587263600148770280649306958101930
        //   + 4082367875863433681332203403145435568316851327593401208105741076214120093531 u   (little-endian 64-bit words)
        const hfp::E x0 = {{0x46debd5cd992f6edull, 0x674322d4f75edaddull, 0x426a00665e5c4479ull, 0x1800deef121f1e76ull}};
        const hfp::E x1 = {{0x97e485b7aef312c2ull, 0xf1aa493335a9e712ull, 0x7260bfb731fb5d25ull, 0x198e9393920d483aull}};
        const hfp::E y0 = {{0x4ce6cc0166fa7daaull, 0xe3d1e7690c43d37bull, 0x4aab71808dcb408full, 0x12c85ea5db8c6debull}};
        const hfp::E y1 = {{0x55acdadcd122975bull, 0xbc4b313370b38ef3ull, 0xec9e99ad690c3395ull, 0x090689d0585ff075ull}};
        auto mont = [](const hfp::E& v) { return hfp::mul(v, hfp::R2); };
        return hfp::AffH<hfp::HFp2>{hfp::E2{mont(x0), mont(x1)}, hfp::E2{mont(y0), mont(y1)}};
    }();
    return (const uint64_t*)&g;
}
}  // namespace
extern "C" {
int gkrhip_g1_bases_create(gkrhip_g1_bases** out, const uint64_t* points, size_t n) { return abi_bases_create(out, points, n, 2); }
int gkrhip_g2_bases_create(gkrhip_g2_bases** out, const uint64_t* points, size_t n) { return abi_bases_create(out, points, n, 4); }
int gkrhip_g1_bases_generate(gkrhip_g1_bases** out, const uint64_t base[8], const uint64_t* scalars, size_t n, int flags) {
    return abi_bases_generate<FpF>(out, base, scalars, n, flags);
}
int gkrhip_g2_bases_generate(gkrhip_g2_bases** out, const uint64_t base[16], const uint64_t* scalars, size_t n, int flags) {
    return abi_bases_generate<Fp2F>(out, base, scalars, n, flags);
}
size_t gkrhip_g1_bases_len(const gkrhip_g1_bases* b) { return b ? b->n : 0; }
size_t gkrhip_g2_bases_len(const gkrhip_g2_bases* b) { return b ? b->n : 0; }
int gkrhip_g1_bases_read(const gkrhip_g1_bases* b, uint64_t* out, size_t first, size_t count) { return abi_bases_read(b, out, first, count); }
int gkrhip_g2_bases_read(const gkrhip_g2_bases* b, uint64_t* out, size_t first, size_t count) { return abi_bases_read(b, out, first, count); }
void gkrhip_g1_bases_destroy(gkrhip_g1_bases* b) { bases_free(b); }
void gkrhip_g2_bases_destroy(gkrhip_g2_bases* b) { bases_free(b); }
int gkrhip_msm_g1_precompute(gkrhip_g1_bases* b, int c) { return abi_precompute<FpF>(b, c); }
int gkrhip_msm_g2_precompute(gkrhip_g2_bases* b, int c) { return abi_precompute<Fp2F>(b, c); }
int gkrhip_msm_g1_set_window(gkrhip_g1_bases* b, int c) { return abi_set_window(b, c); }
int gkrhip_msm_g2_set_window(gkrhip_g2_bases* b, int c) { return abi_set_window(b, c); }
int gkrhip_msm_g1(uint64_t out_affine[8], gkrhip_g1_bases* b, const uint64_t* scalars, size_t n, int flags) {
    return abi_msm<FpF, hfp::HFp>(out_affine, b, scalars, n, flags);
}
int gkrhip_msm_g2(uint64_t out_affine[16], gkrhip_g2_bases* b, const uint64_t* scalars, size_t n, int flags) {
    return abi_msm<Fp2F, hfp::HFp2>(out_affine, b, scalars, n, flags);
}
int gkrhip_msm_g1_g2(uint64_t out_g1[8], uint64_t out_g2[16], gkrhip_g1_bases* b1, gkrhip_g2_bases* b2, const uint64_t* scalars, size_t n, int flags) {
    if (!out_g1 || !out_g2 || !b1 || !b2 || (n && !scalars)) return fail("msm: null argument");
    LEASE_LANE();
    MsmBases* g1[1] = {b1};
    MsmBases* g2[1] = {b2};
    return msm_run_shared(g1, 1, g2, 1, scalars, n, flags, out_g1, out_g2);
}
int gkrhip_msm_shared(uint64_t* out_g1, uint64_t* out_g2, gkrhip_g1_bases* const* g1, size_t k1, gkrhip_g2_bases* const* g2, size_t k2,
                      const uint64_t* scalars, size_t n, int flags) {
    if ((k1 && (!out_g1 || !g1)) || (k2 && (!out_g2 || !g2)) || (n && !scalars)) return fail("msm: null argument");
    if (k1 + k2 > 64) return fail("msm: %zu handles in one shared call (at most 64)", k1 + k2);
    LEASE_LANE();
    std::vector<MsmBases*> a(k1), b(k2);
    for (size_t i = 0; i < k1; i++) a[i] = g1[i];
    for (size_t i = 0; i < k2; i++) b[i] = g2[i];
    return msm_run_shared(a.data(), k1, b.data(), k2, scalars, n, flags, out_g1, out_g2);
}
int gkrhip_msm_g1_once(uint64_t out_affine[8], const uint64_t* points, const uint64_t* scalars, size_t n, int flags) {
    gkrhip_g1_bases* b = nullptr;
    CHK(gkrhip_g1_bases_create(&b, points, n));
    const int rc = gkrhip_msm_g1(out_affine, b, scalars, n, flags);
    gkrhip_g1_bases_destroy(b);
    return rc;
}
int gkrhip_msm_g2_once(uint64_t out_affine[16], const uint64_t* points, const uint64_t* scalars, size_t n, int flags) {
    gkrhip_g2_bases* b = nullptr;
    CHK(gkrhip_g2_bases_create(&b, points, n));
    const int rc = gkrhip_msm_g2(out_affine, b, scalars, n, flags);
    gkrhip_g2_bases_destroy(b);
    return rc;
}
int gkrhip_g1_batch_scalar_mul(uint64_t* out, const uint64_t base[8], const uint64_t* scalars, size_t n, int flags) {
    if (!base || (n && (!out || !scalars))) return fail("g1_batch_scalar_mul: null argument");
    gkrhip_g1_bases* b = nullptr;
    CHK(gkrhip_g1_bases_generate(&b, base, scalars, n, flags));
    const int rc = gkrhip_g1_bases_read(b, out, 0, n);
    gkrhip_g1_bases_destroy(b);
    return rc;
}
int gkrhip_g2_batch_scalar_mul(uint64_t* out, const uint64_t base[16], const uint64_t* scalars, size_t n, int flags) {
    if (!base || (n && (!out || !scalars))) return fail("g2_batch_scalar_mul: null argument");
    gkrhip_g2_bases* b = nullptr;
    CHK(gkrhip_g2_bases_generate(&b, base, scalars, n, flags));
    const int rc = gkrhip_g2_bases_read(b, out, 0, n);
    gkrhip_g2_bases_destroy(b);
    return rc;
}
// computeH (prove.go:308-359) followed by krs2.MultiExp(pk.G1.Z, h) (prove.go:221) with h never leaving the device: the
// transforms leave H in a device table (limb planes, regular form, the reference's bit-reversed order) and the MSM reads its
// scalars from those planes.  h_or_null (optional): the `cardinality` values of H, for the caller that also wants them.
int gkrhip_compute_h_msm_g1(uint64_t out_affine[8], gkrhip_g1_bases* bases_z, const uint64_t* a, const uint64_t* b, const uint64_t* c,
                            size_t n, size_t cardinality, uint64_t* h_or_null) {
    if (!out_affine || !bases_z || !a || !b || !c) return fail("compute_h_msm_g1: null argument");
    LEASE_LANE();
    if (n < 1) return fail("computeH: empty vectors");
    size_t card = cardinality;
    if (card == 0) {
        card = 1;
        while (card < n) card <<= 1;
    }
    if ((card & (card - 1)) || card < n || card < 2) return fail("computeH: domain cardinality %zu is not a power of two >= max(%zu, 2)", card, n);
    if (card > bases_z->n) return fail("compute_h_msm_g1: %zu values of H for %zu bases", card, bases_z->n);
    int logn = 0;
    while (((size_t)1 << logn) < card) logn++;
    ScopedTable t[3];
    DevTable* tp[3];
    const uint64_t* src[3] = {a, b, c};
    for (int i = 0; i < 3; i++) {
        CHK(table_alloc(&t[i], card));
        CHK(upload_table(&t[i], src[i], n));
        if (card > n) {
            hipLaunchKernelGGL(k_ntt_zero, dim3(grid_for(card - n, cx().max_grid)), dim3(GKR_BLOCK), 0, cx().stream, t[i].planes(), n, card);
            HIPCHK(hipGetLastError());
        }
        tp[i] = &t[i];
    }
    CHK(compute_h_dev(tp, logn, nullptr));
    {
        std::lock_guard<std::mutex> lk(bases_z->mu);
        const CPlanes hp = t[0].cplanes();
        MsmWork* w = bases_z->fb.tables ? &bases_z->fb.w : &bases_z->w;
        if (bases_z->fb.tables) {      // pk.G1.Z with fixed-base tables (gkrhip_msm_g1_precompute): H's limb planes straight into the digit kernel
            CHK(msm_fb_dev<FpF>(bases_z, hp.lo, card, 0, nullptr, hp.hi));
        } else {
            CHK(msm_work_prepare(&bases_z->w, std::max<size_t>(bases_z->n, 1), bases_z->c_forced, FpF::W16));
            CHK(msm_dev<FpF>(bases_z, hp.lo, card, 0, nullptr, hp.hi));
        }
        HIPCHK(hipStreamSynchronize(cx().stream));
        CHK(msm_check_error(w, "a value of H"));
        const hfp::Aff r = msm_host_tail<hfp::HFp>(w);
        memcpy(out_affine, &r, sizeof r);
    }
    if (h_or_null) CHK(download_table(&t[0], h_or_null, card));
    for (int i = 0; i < 3; i++) table_release(&t[i]);
    return 0;
}
int gkrhip_bench_msm_g1(int logn, int c_or_0, int warmup, int iters, double* avg_ms, double phase_ms[5], int* c_used,
                        double* host_tail_ms, uint64_t result_or_null[8]) {
    return abi_bench_msm<FpF, hfp::HFp, gkrhip_g1_bases>(g1_generator(), logn, c_or_0, warmup, iters, avg_ms, phase_ms, c_used, host_tail_ms,
                                                         result_or_null);
}
int gkrhip_bench_msm_g2(int logn, int c_or_0, int warmup, int iters, double* avg_ms, double phase_ms[5], int* c_used,
                        double* host_tail_ms, uint64_t result_or_null[16]) {
    return abi_bench_msm<Fp2F, hfp::HFp2, gkrhip_g2_bases>(g2_generator(), logn, c_or_0, warmup, iters, avg_ms, phase_ms, c_used, host_tail_ms,
                                                           result_or_null);
}
int gkrhip_bench_msm_g1_fixed_base(int logn, int c_or_0, int warmup, int iters, double* avg_ms, double phase_ms[5], int* c_used,
                                   double* host_tail_ms, double* precompute_ms, uint64_t result_or_null[8]) {
    return abi_bench_msm<FpF, hfp::HFp, gkrhip_g1_bases>(g1_generator(), logn, c_or_0, warmup, iters, avg_ms, phase_ms, c_used, host_tail_ms,
                                                         result_or_null, /*fixed_base=*/true, precompute_ms);
}
int gkrhip_bench_msm_g2_fixed_base(int logn, int c_or_0, int warmup, int iters, double* avg_ms, double phase_ms[5], int* c_used,
                                   double* host_tail_ms, double* precompute_ms, uint64_t result_or_null[16]) {
    return abi_bench_msm<Fp2F, hfp::HFp2, gkrhip_g2_bases>(g2_generator(), logn, c_or_0, warmup, iters, avg_ms, phase_ms, c_used, host_tail_ms,
                                                           result_or_null, /*fixed_base=*/true, precompute_ms);
}
int gkrhip_g2_generator(uint64_t out[16]) {
    if (!out) return fail("g2_generator: null argument");
    memcpy(out, g2_generator(), 128);
    return 0;
}

int gkrhip_bench_fold(size_t n, int ntab, int warmup, int iters, double* avg_ms, double* isolated_ms_or_null) {
    std::lock_guard<std::mutex> lk(g0.mu);
    CHK(ensure_ctx());
    if (n < 2 || (n & (n - 1)) || ntab < 1 || ntab > GKR_MAX_ARITY + 1) return fail("bench_fold: bad arguments");
    std::vector<DevTable> src(ntab), dst(ntab);
    const DevTable* sp[GKR_MAX_ARITY + 1];
    const DevTable* dp[GKR_MAX_ARITY + 1];
    for (int t = 0; t < ntab; t++) {
        CHK(table_alloc(&src[t], n));
        CHK(table_alloc(&dst[t], n / 2));
        hipLaunchKernelGGL(k_iota, dim3(grid_for(n, cx().max_grid)), dim3(GKR_BLOCK), 0, cx().stream, src[t].planes(), n);
        HIPCHK(hipGetLastError());
        sp[t] = &src[t];
        dp[t] = &dst[t];
    }
    const E r = hfr::from_u64(5);
    const size_t saved_min = cx().prof.min_n;
    cx().prof.min_n = (size_t)1 << 62;  // keep these launches out of the profile accounting
    for (int i = 0; i < warmup; i++) CHK(launch_fold(sp, dp, ntab, n / 2, r));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, cx().stream));
    for (int i = 0; i < iters; i++) CHK(launch_fold(sp, dp, ntab, n / 2, r));
    HIPCHK(hipEventRecord(e1, cx().stream));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = (double)ms / iters;
    if (isolated_ms_or_null) {
        // every launch on an idle GPU with its own event pair: the per-kernel duration a profiler reports
        double tot = 0;
        for (int i = 0; i < iters; i++) {
            HIPCHK(hipStreamSynchronize(cx().stream));
            HIPCHK(hipEventRecord(e0, cx().stream));
            CHK(launch_fold(sp, dp, ntab, n / 2, r));
            HIPCHK(hipEventRecord(e1, cx().stream));
            HIPCHK(hipEventSynchronize(e1));
            HIPCHK(hipEventElapsedTime(&ms, e0, e1));
            tot += ms;
        }
        *isolated_ms_or_null = tot / iters;
    }
    cx().prof.min_n = saved_min;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    for (int t = 0; t < ntab; t++) {
        table_free(&src[t]);
        table_free(&dst[t]);
    }
    return 0;
}

// sumcheck.Prove micro-benchmarks on device-resident tables, shaped like the reference's
// (sumcheck/prover_test.go:96-125, instances of sumcheck/testing.go:11-57: L = R = [0, 1, 2, ...]):
//   kind 0  BenchmarkWithCipherGate: CipherGate(ark = 145646), one point q = RandomFrArray(bn)
//   kind 1  BenchmarkMultiIdentity : IdentityGate on [L, R], `ninstance` points q_i[j] = i*j + i with claims
//                                    Evaluation(q_i) = L(q_i)
// The instance is built outside the timer (as the reference does with StopTimer); *avg_ms = wall-clock per Prove,
// Fiat-Shamir hashing included.  final_claim0 receives finalClaims[0] of the last run (a value to cross-check).
int gkrhip_bench_sumcheck(int kind, int bn, int ninstance, int warmup, int iters, double* avg_ms, uint64_t final_claim0[4]) {
    std::lock_guard<std::mutex> lk(g0.mu);
    CHK(ensure_ctx());
    if (bn < 1 || bn > 28 || iters < 1 || (kind != 0 && kind != 1)) return fail("bench_sumcheck: bad arguments");
    if (kind == 0) ninstance = 1;
    if (ninstance < 1 || ninstance > 1024) return fail("bench_sumcheck: 1..1024 instances");
    LocalOnly lo;
    const size_t n = (size_t)1 << bn;
    ScopedTable L, R;
    CHK(table_alloc(&L, n));
    CHK(table_alloc(&R, n));
    for (DevTable* t : {(DevTable*)&L, (DevTable*)&R}) {
        hipLaunchKernelGGL(k_iota, dim3(grid_for(n, cx().max_grid)), dim3(GKR_BLOCK), 0, cx().stream, t->planes(), n);
        HIPCHK(hipGetLastError());
    }
    const DevTable* X[2] = {&L, &R};
    const int gate = kind == 0 ? GKRHIP_GATE_CIPHER : GKRHIP_GATE_IDENTITY;
    const E ark = kind == 0 ? hfr::from_u64(145646) : hfr::ZERO;
    const int nev = gate_degree(gate) + 2;
    std::vector<E> qs((size_t)ninstance * bn), claims(ninstance, hfr::ZERO);
    if (kind == 0) {
        for (int j = 0; j < bn; j++) qs[j] = hfr::from_u64(((unsigned long long)j * j) ^ 0xf45c9df123fULL);   // common.RandomFrArray
    } else {
        for (int i = 0; i < ninstance; i++) {
            for (int j = 0; j < bn; j++) qs[(size_t)i * bn + j] = hfr::from_u64((unsigned long long)i * j + i);
            CHK(evaluate_dev(&L, bn, &qs[(size_t)i * bn], &claims[i]));       // Evaluation(identity, q_i) = L(q_i)
        }
    }
    std::vector<E> proof((size_t)bn * nev), chal(bn);
    E fin[GKR_MAX_ARITY + 1];
    if (kind == 0) {
        // the claim of the instance (testing.go:24) = P_0(0) + P_0(1) of an honest first round; a single claim never
        // enters the transcript (it only seeds the unused recombination challenge, prover.go:121-128)
        CHK(sumcheck_prove_dev(gate, ark, 2, bn, X, qs.data(), 1, claims.data(), 1, proof.data(), chal.data(), fin));
        E c01 = hfr::add(proof[0], proof[0]);
        for (int j = 1; j < nev; j++) c01 = hfr::add(c01, proof[j]);
        claims[0] = c01;
    }
    for (int i = 0; i < warmup; i++)
        CHK(sumcheck_prove_dev(gate, ark, 2, bn, X, qs.data(), ninstance, claims.data(), ninstance, proof.data(), chal.data(), fin));
    HIPCHK(hipStreamSynchronize(cx().stream));
    const double t0 = now_ms();
    for (int i = 0; i < iters; i++)
        CHK(sumcheck_prove_dev(gate, ark, 2, bn, X, qs.data(), ninstance, claims.data(), ninstance, proof.data(), chal.data(), fin));
    HIPCHK(hipStreamSynchronize(cx().stream));
    *avg_ms = (now_ms() - t0) / iters;
    if (final_claim0) memcpy(final_claim0, fin[0].l, 32);
    return 0;
}

// BenchmarkPartialEvalWithCipher's shape (sumcheck/prover_test.go:127-147): InitializeCipherGateInstance(bn) -- L = R =
// [0, 1, 2, ...], CipherGate(Ark = 145646), q = RandomFrArray(bn) --, the Eq table built once (makeEqTable), then `iters`
// times dispatchPartialEvals of the un-folded instance: the nine evaluations t = 0..8 of round 0 over 2^(bn-1) pairs by the
// reference-shaped evaluator (k_partial_eval<7, 2, 9> on the materialised Eq table), each call handing its sums to the
// host as consumeAccumulate would receive them.  *us_per_call = wall clock per dispatch; evals0 (may be NULL) receives the
// first evaluation of the last call (a value to cross-check: it is the instance's claim minus evals[1]).
int gkrhip_bench_partial_eval(int bn, int warmup, int iters, double* us_per_call, uint64_t evals0[4]) {
    std::lock_guard<std::mutex> lk(g0.mu);
    CHK(ensure_ctx());
    if (bn < 1 || bn > 28 || iters < 1) return fail("bench_partial_eval: bad arguments");
    LocalOnly lo;
    const size_t n = (size_t)1 << bn;
    ScopedTable L, R, eq;
    CHK(table_alloc(&L, n));
    CHK(table_alloc(&R, n));
    CHK(table_alloc(&eq, n));
    for (DevTable* t : {(DevTable*)&L, (DevTable*)&R}) {
        hipLaunchKernelGGL(k_iota, dim3(grid_for(n, cx().max_grid)), dim3(GKR_BLOCK), 0, cx().stream, t->planes(), n);
        HIPCHK(hipGetLastError());
    }
    std::vector<E> q(bn);
    for (int j = 0; j < bn; j++) q[j] = hfr::from_u64(((unsigned long long)j * j) ^ 0xf45c9df123fULL);   // common.RandomFrArray
    const E one = hfr::ONE;
    CHK(build_eq(&eq, q.data(), 1, bn, bn, &one));
    GateDesc g;
    CHK(gate_resolve(GKRHIP_GATE_CIPHER, 2, &g));
    const E ark = hfr::from_u64(145646);
    const DevTable* X[2] = {&L, &R};
    E evals[GKR_MAX_EVALS];
    if (cx().racc_dirty) {
        HIPCHK(hipMemsetAsync(cx().d_racc, 0, sizeof(unsigned long long) * kRaccWords * GKR_RACC_SLOTS, cx().stream));
        HIPCHK(hipMemsetAsync(cx().d_counter, 0, sizeof(unsigned int), cx().stream));
    }
    cx().racc_dirty = true;
    const size_t saved_min = cx().prof.min_n;
    cx().prof.min_n = (size_t)1 << 62;
    for (int i = 0; i < warmup; i++) CHK(partial_evals(g, &eq, X, n / 2, ark, evals, 9, false));
    const double t0 = now_ms();
    for (int i = 0; i < iters; i++) CHK(partial_evals(g, &eq, X, n / 2, ark, evals, 9, false));
    *us_per_call = (now_ms() - t0) * 1e3 / iters;
    cx().prof.min_n = saved_min;
    HIPCHK(hipStreamSynchronize(cx().stream));
    cx().racc_dirty = false;
    if (evals0) memcpy(evals0, evals[0].l, 32);
    return 0;
}

int gkrhip_profile_reset(size_t min_n) {
    g_cnt_prelaunched = 0;
    g_cnt_lookahead = 0;
    g_cnt_coop = 0;
    g_cnt_spec = 0;
    g_cnt_retries = 0;
    g_cnt_layer_checks = 0;
    g_cnt_ahead = 0;
    g_cnt_layer_check_failures = 0;
    g_cnt_group_launches = 0;
    g_cnt_group_combined = 0;
    g_cnt_coalesced = 0;
    return for_each_lane([&](Ctx* l) {
        HIPCHK(hipStreamSynchronize(l->stream));
        prof_clear(l->prof);
        l->prof.min_n = min_n == 0 ? ((size_t)1 << 62) : min_n;
        return 0;
    });
}

int gkrhip_profile_get(uint64_t* fold_launches, double* fold_ms, double* fold_bytes, uint64_t* peval_launches,
                       double* peval_ms, double* peval_modmuls) {
    double fm = 0, pm = 0, fb = 0, pmm = 0;
    uint64_t fl = 0, pl = 0;
    CHK(for_each_lane([&](Ctx* l) {
        HIPCHK(hipStreamSynchronize(l->stream));
        for (auto& p : l->prof.fold_ev) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, p.first, p.second));
            fm += ms;
        }
        for (auto& p : l->prof.peval_ev) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, p.first, p.second));
            pm += ms;
        }
        fl += l->prof.fold_launches;
        pl += l->prof.peval_launches;
        fb += l->prof.fold_bytes;
        pmm += l->prof.peval_modmuls;
        return 0;
    }));
    if (fold_launches) *fold_launches = fl;
    if (fold_ms) *fold_ms = fm;
    if (fold_bytes) *fold_bytes = fb;
    if (peval_launches) *peval_launches = pl;
    if (peval_ms) *peval_ms = pm;
    if (peval_modmuls) *peval_modmuls = pmm;
    return 0;
}

int gkrhip_comm_unique_id(uint8_t out[128]) {
    std::lock_guard<std::mutex> lk(g0.mu);
    CHK(coll_load());
    ncclUniqueId id;
    NCCLCHK(gc.p_get_id(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(out, &id, 128);
    return 0;
}

// lane k of the communicator set: lane 0 is the default lane, further lanes are created on demand
static Ctx* comm_lane(int k) {
    while ((int)gc.lanes.size() <= k) {
        Ctx* l = gc.lanes.empty() ? &g0 : lane_create();
        if (!l) return nullptr;
        gc.lanes.push_back(l);
    }
    return gc.lanes[k];
}

static int shm_attach(int world, int rank, const char* name) {   // on the current lane
    const size_t bytes = 4096 + sizeof(unsigned long long) * kShmSlotWords * world;
    int fd = -1;
    const double t_start = now_ms();
    if (rank == 0) {
        (void)shm_unlink(name);                        // a leftover of a crashed run; names should be unique per run anyway
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) return fail("shm_open/ftruncate(%s) failed", name);
    } else {
        for (;;) {   // wait for rank 0 to create and size the segment
            fd = shm_open(name, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes) break;
            if (fd >= 0) close(fd);
            fd = -1;
            if (now_ms() - t_start > coll_timeout_ms()) return fail("shm segment %s did not appear", name);
            usleep(1000);
        }
    }
    ino_t ino = 0;
    {
        struct stat st;
        if (fstat(fd, &st) == 0) ino = st.st_ino;
    }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return fail("mmap of %s failed", name);
    ShmHdr* h = (ShmHdr*)p;
    // rank 0 stamps the (zero-filled) segment with its creation time, last of all; peers accept only a segment that
    // was stamped within the time-out window, so the leftover of an earlier run under the same name is refused
    // instead of being waited on
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    if (rank == 0) {
        h->magic.store(kShmMagic ^ (unsigned long long)ts.tv_sec, std::memory_order_release);
    } else {
        unsigned polls = 0;
        for (;;) {
            const unsigned long long m = h->magic.load(std::memory_order_acquire);
            clock_gettime(CLOCK_REALTIME, &ts);
            const long long age = (long long)ts.tv_sec - (long long)(m ^ kShmMagic);
            if (m != 0 && age >= -5 && age * 1e3 <= coll_timeout_ms() + 5e3) break;
            if (now_ms() - t_start > coll_timeout_ms()) {
                munmap(p, bytes);
                return fail("shm segment %s is stale or was never initialised by rank 0", name);
            }
            usleep(1000);
            // This mapping may be the LEFTOVER of an earlier run that this rank opened before rank 0 unlinked it and
            // created the fresh segment: look at the name again every now and then and move to the new file
            if ((++polls & 127) == 0) {
                const int fd2 = shm_open(name, O_RDWR, 0600);
                struct stat st;
                if (fd2 >= 0 && fstat(fd2, &st) == 0 && st.st_ino != ino && (size_t)st.st_size >= bytes) {
                    void* p2 = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd2, 0);
                    if (p2 != MAP_FAILED) {
                        munmap(p, bytes);
                        p = p2;
                        h = (ShmHdr*)p;
                        ino = st.st_ino;
                    }
                }
                if (fd2 >= 0) close(fd2);
            }
        }
    }
    cx().lc.shm = h;
    cx().lc.shm_slots = (unsigned long long*)((char*)p + 4096);
    cx().lc.shm_bytes = bytes;
    CHK(coll_buffers(4096));
    return 0;
}

static int comm_common(int world, int rank, int nlanes) {
    if (world < 1 || (world & (world - 1)) || rank < 0 || rank >= world)
        return fail("comm_init: world %d must be a power of two and 0 <= rank %d < world", world, rank);
    if (nlanes < 1 || nlanes > 8) return fail("comm_init: 1..8 lanes");
    if (!gc.lanes.empty()) return fail("communicator already initialised");
    return 0;
}
static void comm_set(int world, int rank) {
    int gamma = 0;
    while ((1 << gamma) < world) gamma++;
    gc.world = world;
    gc.rank = rank;
    gc.gamma = gamma;
    // every rank (taken to be on this node) runs one waiting host thread per lane
    g_wait_ranks.store(world);
}

int gkrhip_comm_init_lanes(int world, int rank, int nlanes, const uint8_t* ids /* nlanes x 128 */) {
    std::lock_guard<std::mutex> lk(g0.mu);
    CHK(ensure_ctx());
    CHK(comm_common(world, rank, nlanes));
    CHK(coll_load());
    for (int k = 0; k < nlanes; k++) {
        Ctx* l = comm_lane(k);
        if (!l) return fail("cannot create lane %d: %s", k, g_err.c_str());
        UseLane u(l);
        ncclUniqueId id;
        memcpy(&id, ids + (size_t)128 * k, 128);
        NCCLCHK(gc.p_init(&cx().lc.comm, world, id, rank));
        CHK(coll_buffers(4096));
    }
    comm_set(world, rank);
    return 0;
}

// The ticker (host_coll.hip.h): ONE exchange channel for the process, `nlanes` lanes that exchange through it.
// id_bytes != nullptr: an RCCL communicator (mode 0: all-reduce on the host-mapped buffers; mode 1: on device staging
// buffers); shm_name != nullptr: a host all-reduce through a POSIX shared-memory segment (several ranks on one GPU: tests).
// everything a (possibly half-built) ticker owns; the lanes that were pointed at it are released too
static void ticker_release(Ticker* t) {
    if (!t) return;
    for (Ctx* l : gc.lanes) {
        if (!l) continue;
        l->lc.tick_lane = -1;
        if (l != &g0) lane_destroy(l, /*pool=*/false);
    }
    gc.lanes.clear();
    gc.next_lane = 0;
    if (t->comm) (void)gc.p_destroy(t->comm);
    if (t->shm) {
        t->shm->abort.store(1, std::memory_order_release);      // a peer already waiting in the segment fails instead of hanging
        munmap((void*)t->shm, t->shm_bytes);
    }
    if (t->h_send) (void)hipHostFree(t->h_send);
    if (t->h_recv) (void)hipHostFree(t->h_recv);
    if (t->h_done) (void)hipHostFree(t->h_done);
    if (t->d_stage_send) (void)hipFree(t->d_stage_send);
    if (t->d_stage_recv) (void)hipFree(t->d_stage_recv);
    if (t->stream) (void)hipStreamDestroy(t->stream);
    delete t;
}
static int comm_init_tick_impl(int world, int rank, int nlanes, const uint8_t* id_bytes, int mode, const char* shm_name) {
    std::lock_guard<std::mutex> lk(g0.mu);
    CHK(ensure_ctx());
    CHK(comm_common(world, rank, nlanes));
    if (nlanes > kTickMaxLanes) return fail("comm_init_tick: at most %d lanes", kTickMaxLanes);
    if (g_ticker) return fail("a ticker is already running");
    Ticker* t = new Ticker();
    t->nlanes = nlanes;
    t->world = world;
    t->rank = rank;
    t->dev_buf = mode == 1;
    // ONE way out of every failure below: whatever exists by then (communicator, mapping, buffers, stream, the lanes'
    // slot numbers) is released, so that the peers see this rank leave instead of waiting for its first tick
    auto bail = [&](int rc) {
        ticker_release(t);
        return rc;
    };
    const size_t total = kTickHeader + (size_t)nlanes * kTickStride;
    if (id_bytes) {
        const int lrc = coll_load();
        if (lrc) return bail(lrc);
        ncclUniqueId id;
        memcpy(&id, id_bytes, 128);
        ncclResult_t r = gc.p_init(&t->comm, world, id, rank);
        if (r != ncclSuccess) {
            t->comm = nullptr;
            return bail(fail("ncclCommInitRank failed: %s", gc.p_errstr ? gc.p_errstr(r) : "?"));
        }
    } else {
        // rank 0 creates the segment (header + one tick buffer per rank), the others map it once it has its size
        const size_t bytes = 4096 + sizeof(unsigned long long) * total * world;
        int fd = -1;
        const double t_start = now_ms();
        if (rank == 0) {
            (void)shm_unlink(shm_name);
            fd = shm_open(shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
            if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) {
                if (fd >= 0) close(fd);
                return bail(fail("shm_open/ftruncate(%s) failed", shm_name));
            }
        } else {
            for (;;) {
                fd = shm_open(shm_name, O_RDWR, 0600);
                struct stat st;
                if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes) break;
                if (fd >= 0) close(fd);
                if (now_ms() - t_start > coll_timeout_ms()) return bail(fail("shm segment %s did not appear", shm_name));
                usleep(1000);
            }
        }
        ino_t ino = 0;
        {
            struct stat st;
            if (fstat(fd, &st) == 0) ino = st.st_ino;
        }
        void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) return bail(fail("mmap of %s failed", shm_name));
        t->shm = (ShmHdr*)p;
        t->shm_slots = (unsigned long long*)((char*)p + 4096);
        t->shm_bytes = bytes;
        t->shm_name = shm_name;
        struct timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        if (rank == 0) {
            t->shm->magic.store(kShmMagic ^ (unsigned long long)ts.tv_sec, std::memory_order_release);
        } else {
            // as shm_attach: only a segment rank 0 stamped within the time-out window is accepted, and the name is looked at
            // again every now and then -- this mapping may be the leftover of a crashed run that was opened before rank 0
            // unlinked it and created the fresh one
            unsigned polls = 0;
            for (;;) {
                const unsigned long long m = t->shm->magic.load(std::memory_order_acquire);
                clock_gettime(CLOCK_REALTIME, &ts);
                const long long age = (long long)ts.tv_sec - (long long)(m ^ kShmMagic);
                if (m != 0 && age >= -5 && age * 1e3 <= coll_timeout_ms() + 5e3) break;
                if (now_ms() - t_start > coll_timeout_ms()) return bail(fail("shm segment %s is stale or was never initialised by rank 0", shm_name));
                usleep(1000);
                if ((++polls & 127) == 0) {
                    const int fd2 = shm_open(shm_name, O_RDWR, 0600);
                    struct stat st;
                    if (fd2 >= 0 && fstat(fd2, &st) == 0 && st.st_ino != ino && (size_t)st.st_size >= bytes) {
                        void* p2 = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd2, 0);
                        if (p2 != MAP_FAILED) {
                            munmap((void*)t->shm, bytes);
                            t->shm = (ShmHdr*)p2;
                            t->shm_slots = (unsigned long long*)((char*)p2 + 4096);
                            ino = st.st_ino;
                        }
                    }
                    if (fd2 >= 0) close(fd2);
                }
            }
        }
    }
    hipError_t e = hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc(&t->h_send, sizeof(unsigned long long) * total, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostMalloc(&t->h_recv, sizeof(unsigned long long) * total, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostMalloc(&t->h_done, 64, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&t->d_send, t->h_send, 0);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&t->d_recv, t->h_recv, 0);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&t->d_done, t->h_done, 0);
    if (e == hipSuccess && t->dev_buf) e = hipMalloc(&t->d_stage_send, sizeof(unsigned long long) * total);
    if (e == hipSuccess && t->dev_buf) e = hipMalloc(&t->d_stage_recv, sizeof(unsigned long long) * total);
    if (e != hipSuccess) return bail(fail("comm_init_tick: %s", hipGetErrorString(e)));
    memset(t->h_send, 0, sizeof(unsigned long long) * total);
    memset(t->h_recv, 0, sizeof(unsigned long long) * total);
    *t->h_done = 0;
    for (int k = 0; k < nlanes; k++) {
        Ctx* l = comm_lane(k);
        if (!l) return bail(fail("cannot create lane %d: %s", k, g_err.c_str()));
        l->lc.tick_lane = k;
    }
    comm_set(world, rank);
    t->running.store(true);
    g_ticker = t;
    t->th = std::thread(ticker_main, t, g0.device);
    return 0;
}
int gkrhip_comm_init_tick(int world, int rank, int nlanes, const uint8_t id_bytes[128]) {
    if (!id_bytes) return fail("comm_init_tick: no unique id");
    static const int mode = [] {
        const char* e = getenv("GKRHIP_TICK_DEVICE_BUF");
        return e ? atoi(e) : 0;
    }();
    return comm_init_tick_impl(world, rank, nlanes, id_bytes, mode, nullptr);
}
int gkrhip_comm_init_tick_shm(int world, int rank, int nlanes, const char* name) {
    if (!name) return fail("comm_init_tick_shm: no segment name");
    return comm_init_tick_impl(world, rank, nlanes, nullptr, 0, name);
}
// ticks issued / ticks in which no lane of any rank had words (measurement)
int gkrhip_comm_tick_stats(uint64_t* ticks, uint64_t* idle_ticks) {
    if (ticks) *ticks = g_ticker ? g_ticker->ticks.load() : 0;
    if (idle_ticks) *idle_ticks = g_ticker ? g_ticker->idle_ticks.load() : 0;
    return 0;
}

int gkrhip_comm_init(int world, int rank, const uint8_t id_bytes[128]) {
    if (world == 1 && !id_bytes) {   // explicit single-GPU mode without a communicator
        std::lock_guard<std::mutex> lk(g0.mu);
        CHK(ensure_ctx());
        comm_set(1, 0);
        return 0;
    }
    return gkrhip_comm_init_lanes(world, rank, 1, id_bytes);
}

int gkrhip_comm_init_shm_lanes(int world, int rank, int nlanes, const char* name) {
    std::lock_guard<std::mutex> lk(g0.mu);
    CHK(ensure_ctx());
    CHK(comm_common(world, rank, nlanes));
    for (int k = 0; k < nlanes; k++) {
        Ctx* l = comm_lane(k);
        if (!l) return fail("cannot create lane %d: %s", k, g_err.c_str());
        UseLane u(l);
        char nm[256];
        snprintf(nm, sizeof nm, "%s_%d", name, k);
        CHK(shm_attach(world, rank, nm));
    }
    comm_set(world, rank);
    for (int k = 0; k < nlanes; k++) {
        UseLane u(gc.lanes[k]);
        CHK(shm_barrier());   // everybody mapped (the segments are zero-filled by ftruncate)
    }
    if (rank == 0) {     // every rank holds its mapping: the names can go (nothing is left behind in /dev/shm)
        for (int k = 0; k < nlanes; k++) {
            char nm[256];
            snprintf(nm, sizeof nm, "%s_%d", name, k);
            (void)shm_unlink(nm);
        }
    }
    return 0;
}

int gkrhip_comm_init_shm(int world, int rank, const char* name) { return gkrhip_comm_init_shm_lanes(world, rank, 1, name); }

int gkrhip_comm_destroy(void) {
    std::lock_guard<std::mutex> lk(g0.mu);
    if (g_ticker) {
        // the tickers of all ranks leave together: after the tick in which every rank has voted to stop
        Ticker* t = g_ticker;
        t->stop_vote.store(true, std::memory_order_release);
        if (t->th.joinable()) t->th.join();
        (void)hipStreamSynchronize(t->stream);
        if (t->comm) (void)gc.p_destroy(t->comm);
        if (t->shm) {
            t->shm->abort.store(1, std::memory_order_release);
            munmap((void*)t->shm, t->shm_bytes);
            if (t->rank == 0) (void)shm_unlink(t->shm_name.c_str());     // every ticker has left (they stop together)
        }
        if (t->d_stage_send) (void)hipFree(t->d_stage_send);
        if (t->d_stage_recv) (void)hipFree(t->d_stage_recv);
        (void)hipStreamDestroy(t->stream);
        (void)hipHostFree(t->h_send);
        (void)hipHostFree(t->h_recv);
        (void)hipHostFree(t->h_done);
        g_ticker = nullptr;
        delete t;
    }
    for (Ctx* l : gc.lanes) {
        {
            std::unique_lock<std::mutex> ll;
            if (l != &g0) ll = std::unique_lock<std::mutex>(l->mu);   // g0.mu is already held
            UseLane u(l);
            (void)hipStreamSynchronize(cx().stream);
            if (cx().lc.comm) {
                (void)gc.p_destroy(cx().lc.comm);
                cx().lc.comm = nullptr;
            }
            cx().lc.tick_lane = -1;
            if (cx().lc.shm) {
                cx().lc.shm->abort.store(1, std::memory_order_release);   // a peer still waiting here fails instead of hanging
                munmap((void*)cx().lc.shm, cx().lc.shm_bytes);
                cx().lc.shm = nullptr;
                cx().lc.shm_slots = nullptr;
            }
        }
        if (l != &g0) lane_destroy(l, /*pool=*/false);
    }
    gc.lanes.clear();
    gc.next_lane = 0;
    gc.world = 1;
    gc.rank = 0;
    gc.gamma = 0;
    return 0;
}

int gkrhip_comm_info(int* world, int* rank) {
    if (world) *world = gc.world;
    if (rank) *rank = gc.rank;
    return 0;
}

/* host-only scalar helpers of the sharded protocol (no GPU needed; used by the CPU multi-process tests) */
int gkrhip_host_shard_seed(uint64_t out[4], const uint64_t* q_tail, int gamma, int rank) {
    const E r = shard_seed((const E*)q_tail, gamma, rank);
    memcpy(out, r.l, 32);
    return 0;
}
int gkrhip_host_limbsplit_reduce(uint64_t out[4], const uint64_t* lanes, int nlanes) {
    if (nlanes != 8 && nlanes != 9) return fail("limbsplit_reduce: 8 or 9 lanes");
    unsigned long long w[9] = {0};
    for (int i = 0; i < nlanes; i++) w[i] = lanes[i];
    const E r = limbs9_to_fr(w);
    memcpy(out, r.l, 32);
    return 0;
}
int gkrhip_host_mimc_hash(uint64_t out[4], const uint64_t* in, size_t n) {
    const E r = hfr::mimc_hash((const E*)in, n);
    memcpy(out, r.l, 32);
    return 0;
}
/* coefficients (9) of the cipher round polynomial from the 8 monomial sums M_j, the running constant c and
 * the round's coordinate q_k:  c * ((1-q_k) + (2 q_k - 1) t) * sum_j C(7,j) M_j t^j */
int gkrhip_host_cipher_round_coeffs(uint64_t out[36], const uint64_t* M, const uint64_t c_[4], const uint64_t qk_[4]) {
    static const hfr::u64 binom7[8] = {1, 7, 21, 35, 35, 21, 7, 1};
    E c, qk, sp[8], co[9];
    memcpy(c.l, c_, 32);
    memcpy(qk.l, qk_, 32);
    for (int j = 0; j < 8; j++) {
        E mj;
        memcpy(mj.l, M + 4 * j, 32);
        sp[j] = hfr::mul(mj, hfr::from_u64(binom7[j]));
    }
    const E a0 = hfr::mul(c, hfr::sub(hfr::ONE, qk));
    const E a1 = hfr::mul(c, hfr::sub(hfr::add(qk, qk), hfr::ONE));
    co[0] = hfr::mul(a0, sp[0]);
    for (int j = 1; j < 8; j++) co[j] = hfr::add(hfr::mul(a0, sp[j]), hfr::mul(a1, sp[j - 1]));
    co[8] = hfr::mul(a1, sp[7]);
    memcpy(out, co, sizeof co);
    return 0;
}

/* The prover's own check of a finished sumcheck (sumcheck_closes, host_sumcheck.hip.h), on host data alone. */
int gkrhip_host_sumcheck_closes(int gate, const uint64_t* ark_or_null, int arity, int bN, const uint64_t* qprimes, int nq,
                                const uint64_t* claims, int nclaims, int claims_are_sums, const uint64_t* proof,
                                const uint64_t* challenges, const uint64_t* final_claims, int* verdict) {
    if (!verdict || !proof || !final_claims || (bN > 0 && (!challenges || !qprimes))) return fail("host_sumcheck_closes: null argument");
    if (bN < 0 || bN > 40 || nq < 1) return fail("host_sumcheck_closes: bad shape");
    GateDesc g;
    CHK(gate_resolve(gate, arity, &g));
    E ark = hfr::ZERO;
    if (ark_or_null && gate != GKRHIP_GATE_IDENTITY) memcpy(ark.l, ark_or_null, 32);
    const E rho = (nclaims > 1 || nq > 1) && nclaims >= 1 ? hfr::mimc_hash((const E*)claims, (size_t)nclaims) : hfr::ZERO;
    *verdict = sumcheck_closes(g, ark, bN, (const E*)qprimes, nclaims >= 1 ? nq : 1, (const E*)claims, nclaims, claims_are_sums != 0, rho,
                               (const E*)proof, (const E*)challenges, (const E*)final_claims);
    return 0;
}
/* Round 0 ahead of its point, the host's part (CipherLoop::coefficients): M_j = sum_y eq(q_low, y) S_j(y), j = 1..7, from the
 * 7 * 2^t class sums S (S_j(y) at (j - 1) * 2^t + y) and the t coordinates the layer before drew last. */
int gkrhip_host_ahead_contract(uint64_t out[28], const uint64_t* class_sums, const uint64_t* q_low, int t) {
    if (!out || !class_sums || t < 0 || t > 12 || (t > 0 && !q_low)) return fail("host_ahead_contract: bad argument");
    E M[7];
    ahead_contract((const E*)class_sums, (const E*)q_low, t, M);
    memcpy(out, M, sizeof M);
    return 0;
}

// Host-only self-test of the proof groups' driver (host_group.hip.h; no GPU): n proofs on stacks of their own ask for `steps` launches
// each through launch_batch; a recorder stands in for hipLaunchKernel.  Proof 1 (if any) asks for another grid at step
// `diverge_at` (it must get a launch of its own there and be back in the common one afterwards), the last proof returns after
// `leave_after` steps (it must leave the group without holding the others up).  The recorder checks every combined launch:
// its arguments are exactly what the proofs that are in it asked for, each proof's steps arrive in order, nobody is launched twice.
// counts[0] = launches asked for, counts[1] = launches made, counts[2] = the most proofs in one launch; *verdict 0 when all of it held.
namespace {
struct SelfArgs {
    int proof, step;
    unsigned long long tag;
};
GKR_KERNEL void k_group_selftest(Batch<SelfArgs>) {}
struct SelfRecorder {
    int n = 0, bad = 0, diverge_at = -1;
    std::vector<int> next_step;        // by proof
    unsigned long long made = 0, most = 0;
} g_self;
hipError_t self_launch(const void* fn, dim3 grid, dim3, void** params, size_t, hipStream_t) {
    const SelfArgs* a = (const SelfArgs*)params[0];
    if (fn != reinterpret_cast<const void*>(&k_group_selftest) || grid.z < 1 || grid.z > (unsigned)g_self.n) g_self.bad++;
    for (unsigned z = 0; z < grid.z; z++) {
        const SelfArgs& x = a[z];
        if (x.proof < 0 || x.proof >= g_self.n || x.tag != 0x9e3779b97f4a7c15ull * (unsigned long long)(x.proof * 1000 + x.step + 1) ||
            x.step != g_self.next_step[(size_t)x.proof] || grid.x != ((x.proof == 1 && x.step == g_self.diverge_at) ? 7u : 3u)) {
            g_self.bad++;
            continue;
        }
        g_self.next_step[(size_t)x.proof]++;
        for (unsigned y = 0; y < z; y++)
            if (a[y].proof == x.proof) g_self.bad++;
    }
    g_self.made++;
    g_self.most = std::max<unsigned long long>(g_self.most, grid.z);
    return hipSuccess;
}
std::mutex g_self_mu;
}  // namespace
int gkrhip_host_group_selftest(int n, int steps, int diverge_at, int leave_after, uint64_t counts[3], int* verdict) {
    if (n < 1 || n > GKR_GROUP_MAX || steps < 1 || !counts || !verdict) return fail("group_selftest: bad argument");
    std::lock_guard<std::mutex> lk(g_self_mu);
    g_self = SelfRecorder();
    g_self.n = n;
    g_self.diverge_at = diverge_at;
    g_self.next_step.assign((size_t)n, 0);
    Group g;
    g.launch = self_launch;
    g.proofs.resize((size_t)n);
    std::vector<int> done_steps((size_t)n, 0);
    for (int i = 0; i < n; i++) {
        g.proofs[(size_t)i].body = [i, n, steps, diverge_at, leave_after, &done_steps]() {
            const int mine = (i == n - 1 && n > 1 && leave_after >= 0) ? std::min(steps, leave_after) : steps;
            for (int k = 0; k < mine; k++) {
                SelfArgs a = {i, k, 0x9e3779b97f4a7c15ull * (unsigned long long)(i * 1000 + k + 1)};
                const unsigned gx = (i == 1 && k == diverge_at) ? 7u : 3u;
                if (launch_batch(k_group_selftest, dim3(gx), dim3(64), 0, nullptr, a) != hipSuccess) return -1;
                done_steps[(size_t)i] = k + 1;
            }
            return 100 + i;
        };
    }
    CHK(group_run(g));
    unsigned long long wanted = 0;
    int bad = g_self.bad;
    for (int i = 0; i < n; i++) {
        const int mine = (i == n - 1 && n > 1 && leave_after >= 0) ? std::min(steps, leave_after) : steps;
        wanted += (unsigned long long)mine;
        if (g.proofs[(size_t)i].rc != 100 + i || done_steps[(size_t)i] != mine || g_self.next_step[(size_t)i] != mine) bad++;
    }
    if (g.launches != wanted || g.combined != g_self.made) bad++;
    counts[0] = wanted;
    counts[1] = g_self.made;
    counts[2] = g_self.most;
    *verdict = bad;
    return 0;
}

int gkrhip_profile_latency(uint64_t* prelaunched_rounds, uint64_t* lookahead_round0, uint64_t* coop_rounds) {
    if (prelaunched_rounds) *prelaunched_rounds = g_cnt_prelaunched.load();
    if (lookahead_round0) *lookahead_round0 = g_cnt_lookahead.load();
    if (coop_rounds) *coop_rounds = g_cnt_coop.load();
    return 0;
}

int gkrhip_profile_counter(const char* name, uint64_t* value) {
    if (!name || !value) return fail("gkrhip_profile_counter: null argument");
    const std::string n(name);
    if (n == "prelaunched_rounds") *value = g_cnt_prelaunched.load();
    else if (n == "lookahead_round0") *value = g_cnt_lookahead.load();
    else if (n == "coop_rounds") *value = g_cnt_coop.load();
    else if (n == "spec_rounds") *value = g_cnt_spec.load();
    else if (n == "chal_retries") *value = g_cnt_retries.load();
    else if (n == "ahead_round0") *value = g_cnt_ahead.load();
    else if (n == "hw_queues_set_by_library") *value = (uint64_t)g_hwq_set_by_library.load();
    else if (n == "hw_queues_from_environment") *value = (uint64_t)g_hwq_from_env.load();
    else if (n == "layer_checks") *value = g_cnt_layer_checks.load();
    else if (n == "layer_check_failures") *value = g_cnt_layer_check_failures.load();
    else if (n == "arena_busy_releases") *value = g_cnt_busy_releases.load();
    else if (n == "coalesced_proofs") *value = g_cnt_coalesced.load();              // single calls that were proven in a group formed from single calls
    else if (n == "group_launches_wanted") *value = g_cnt_group_launches.load();      // launches the proofs of the groups asked for ...
    else if (n == "group_launches_made") *value = g_cnt_group_combined.load();       // ... and the combined launches that served them
    else return fail("gkrhip_profile_counter: unknown counter '%s'", name);
    return 0;
}

int gkrhip_profile_host(uint64_t* rounds, double* hash_ms, double* wait_ms, double* launch_ms, double* other_ms) {
    uint64_t r = 0;
    double h = 0, w = 0, l_ = 0, o = 0;
    CHK(for_each_lane([&](Ctx* l) {
        r += l->prof.rounds;
        h += l->prof.host_hash_ms;
        w += l->prof.host_wait_ms;
        l_ += l->prof.host_launch_ms;
        o += l->prof.host_other_ms;
        return 0;
    }));
    if (rounds) *rounds = r;
    if (hash_ms) *hash_ms = h;
    if (wait_ms) *wait_ms = w;
    if (launch_ms) *launch_ms = l_;
    if (other_ms) *other_ms = o;
    return 0;
}

}  // extern "C"
