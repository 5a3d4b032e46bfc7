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
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gkrhip.h"
#include "fr_host.h"
#include "kernels.hip.h"
#include "cipher_round.hip.h"

using hfr::E;

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
namespace {

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

struct Profile {
    double host_hash_ms = 0, host_wait_ms = 0, host_launch_ms = 0, host_other_ms = 0;
    uint64_t rounds = 0;
    size_t min_n = (size_t)1 << 62;
    uint64_t fold_launches = 0, peval_launches = 0;
    double fold_bytes = 0, peval_modmuls = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> fold_ev, peval_ev;
    std::vector<hipEvent_t> pool;
};

// per-lane state of the collective (one communicator / shared-memory segment per lane: the lanes of a rank
// issue their collectives independently, lane k pairing with lane k of the other ranks)
struct ShmHdr {
    std::atomic<unsigned> arrive, gen;
};
struct LaneColl {
    ncclComm_t comm = nullptr;
    ShmHdr* shm = nullptr;
    unsigned long long* shm_slots = nullptr;
    size_t shm_bytes = 0;
    unsigned long long* d_buf = nullptr;   // device staging: lanes / gathered elements
    unsigned long long* h_buf = nullptr;   // pinned mirror
    unsigned long long* h_tmp = nullptr;
    size_t buf_words = 0;
};

struct Ctx {
    bool ready = false;
    int device = -1;
    hipStream_t stream = nullptr;
    unsigned long long* d_partials = nullptr;  // per-block limb-split partial sums
    unsigned long long* d_sums = nullptr;      // reduced sums (device)
    unsigned long long* h_sums = nullptr;      // pinned
    uint4* d_small = nullptr;                  // gather buffer (AoS)
    uint4* h_small = nullptr;                  // pinned
    Fr* d_q = nullptr;                         // qPrime coordinates + seeds staging
    size_t d_q_cap = 0;
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
    int g_max = 16;                            // log2(max threads of the round kernel): 16 measured best with 4 proofs in flight (17 for one proof alone)
    bool force_generic = false;
    bool claim_trick = true;                   // GKRHIP_CLAIM_TRICK=0: always compute all eight monomial sums
    int lat_mode = 1;                          // GKRHIP_LAT: 0 never, 1 rounds with one pair per lane, 2 always
    int wide_mode = 1;                         // GKRHIP_WIDE: deferred-reduction kernel for the rounds with several pairs per lane
    int wt_late_lj = 3;                        // ... and from 2^3 pairs per lane on, the lane weight is applied after the loop
    bool force_collective = false;             // GKRHIP_FORCE_COLLECTIVE: take the collective path even at world == 1
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
#define g (*g_cur)
struct UseLane {
    Ctx* prev;
    explicit UseLane(Ctx* l) : prev(g_cur) { g_cur = l; }
    ~UseLane() { g_cur = prev; }
};
std::mutex g_lanes_mu;
std::vector<Ctx*> g_lanes;                     // every lane, for profile aggregation
struct Pool {
    std::mutex mu;
    std::vector<std::pair<size_t, uint4*>> free_list;  // (cap, base) cache of table buffers
} g_pool;
thread_local std::string g_err;

int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
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
int lane_alloc();

int ctx_init(int dev) {
    if (g.ready) {
        if (dev >= 0 && dev != g.device) return fail("gkrhip already initialised on device %d", g.device);
        return 0;
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
    g.n_cu = prop.multiProcessorCount;
    g.max_grid = g.n_cu * 32;   // streaming kernels: 8192 workgroups measured best for the fold (profiles/)
    if (const char* e = getenv("GKRHIP_GMAX")) g.g_max = std::max(8, std::min(20, atoi(e)));
    if (const char* e = getenv("GKRHIP_GENERIC")) g.force_generic = atoi(e) != 0;
    if (const char* e = getenv("GKRHIP_LAT")) g.lat_mode = atoi(e);
    if (const char* e = getenv("GKRHIP_WIDE")) g.wide_mode = atoi(e);
    if (const char* e = getenv("GKRHIP_WT_LATE_LJ")) g.wt_late_lj = atoi(e);
    if (const char* e = getenv("GKRHIP_CLAIM_TRICK")) g.claim_trick = atoi(e) != 0;
    if (const char* e = getenv("GKRHIP_FOLD_GRID")) g.fold_grid = std::max(64, atoi(e));
    if (const char* e = getenv("GKRHIP_FORCE_COLLECTIVE")) g.force_collective = atoi(e) != 0;
    g.lag = new hfr::Lagrange();
    g.device = dev;
    CHK(lane_alloc());
    {
        std::lock_guard<std::mutex> lk(g_lanes_mu);
        g_lanes.push_back(&g);
    }
    g.ready = true;
    return 0;
}

// stream + buffers of the current lane
int lane_alloc() {
    HIPCHK(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    const size_t nwords = (size_t)GKR_MAX_EVALS * GKR_ACC_WORDS;
    HIPCHK(hipMalloc(&g.d_partials, sizeof(unsigned long long) * nwords * kPartialBlocks));
    HIPCHK(hipMalloc(&g.d_sums, sizeof(unsigned long long) * nwords));
    HIPCHK(hipHostMalloc(&g.h_sums, sizeof(unsigned long long) * nwords, hipHostMallocDefault));
    HIPCHK(hipMalloc(&g.d_small, sizeof(uint4) * 2 * 8));
    HIPCHK(hipHostMalloc(&g.h_small, sizeof(uint4) * 2 * 8, hipHostMallocDefault));
    HIPCHK(hipHostMalloc(&g.h_round, sizeof(unsigned long long) * (GKR_CR_WORDS + 16), hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer((void**)&g.d_round, g.h_round, 0));
    HIPCHK(hipHostMalloc(&g.h_flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer((void**)&g.d_flag, g.h_flag, 0));
    *g.h_flag = 0;
    g.seq = 0;
    HIPCHK(hipMalloc(&g.d_counter, 64));
    HIPCHK(hipMemset(g.d_counter, 0, 64));
    return 0;
}
void lane_free() {
    (void)hipStreamSynchronize(g.stream);
    (void)hipFree(g.d_partials);
    (void)hipFree(g.d_sums);
    (void)hipHostFree(g.h_sums);
    (void)hipFree(g.d_small);
    (void)hipHostFree(g.h_small);
    (void)hipHostFree(g.h_round);
    (void)hipHostFree(g.h_flag);
    (void)hipFree(g.d_counter);
    if (g.d_q) (void)hipFree(g.d_q);
    g.d_q = nullptr;
    g.d_q_cap = 0;
    (void)hipStreamDestroy(g.stream);
    g.stream = nullptr;
}
// a new lane configured like the default one
Ctx* lane_create() {
    Ctx* l = new Ctx();
    l->device = g0.device;
    l->n_cu = g0.n_cu;
    l->max_grid = g0.max_grid;
    l->fold_grid = g0.fold_grid;
    l->fold_split = g0.fold_split;
    l->g_max = g0.g_max;
    l->force_generic = g0.force_generic;
    l->lat_mode = g0.lat_mode;
    l->wide_mode = g0.wide_mode;
    l->wt_late_lj = g0.wt_late_lj;
    l->claim_trick = g0.claim_trick;
    l->force_collective = g0.force_collective;
    l->lag = g0.lag;
    l->prof.min_n = g0.prof.min_n;
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
void lane_destroy(Ctx* l) {
    {
        std::lock_guard<std::mutex> lk(g_lanes_mu);
        g_lanes.erase(std::remove(g_lanes.begin(), g_lanes.end(), l), g_lanes.end());
    }
    UseLane u(l);
    lane_free();
    delete l;
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
    for (size_t i = 0; i < g_pool.free_list.size(); i++) {
        if (g_pool.free_list[i].first == cap) {
            t->base = g_pool.free_list[i].second;
            t->cap = cap;
            g_pool.free_list.erase(g_pool.free_list.begin() + i);
            return 0;
        }
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, sizeof(uint4) * 2 * cap);
    if (e != hipSuccess) {
        // drop the cache and retry once
        for (auto& f : g_pool.free_list) (void)hipFree(f.second);
        g_pool.free_list.clear();
        e = hipMalloc(&p, sizeof(uint4) * 2 * cap);
        if (e != hipSuccess) return fail("hipMalloc of a %zu-element table failed: %s", cap, hipGetErrorString(e));
    }
    t->base = (uint4*)p;
    t->cap = cap;
    return 0;
}
void table_release(DevTable* t) {
    if (t->base) {
        std::lock_guard<std::mutex> lk(g_pool.mu);
        g_pool.free_list.emplace_back(t->cap, t->base);
    }
    t->base = nullptr;
    t->cap = 0;
}
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

// ---- boundary copies -----------------------------------------------------------------------------
// host AoS -> device planes.  Staged through a device AoS buffer and transposed by k_aos_to_planes.
int upload_table(DevTable* t, const uint64_t* host_aos, size_t n) {
    uint4* stage = nullptr;
    HIPCHK(hipMalloc(&stage, 32 * n));
    HIPCHK(hipMemcpyAsync(stage, host_aos, 32 * n, hipMemcpyHostToDevice, g.stream));
    hipLaunchKernelGGL(k_aos_to_planes, dim3(grid_for(n, g.max_grid)), dim3(GKR_BLOCK), 0, g.stream, stage, t->planes(), n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(g.stream));
    HIPCHK(hipFree(stage));
    return 0;
}
int download_table(const DevTable* t, uint64_t* host_aos, size_t n) {
    uint4* stage = nullptr;
    HIPCHK(hipMalloc(&stage, 32 * n));
    hipLaunchKernelGGL(k_planes_to_aos, dim3(grid_for(n, g.max_grid)), dim3(GKR_BLOCK), 0, g.stream, t->cplanes(), stage, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(host_aos, stage, 32 * n, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    HIPCHK(hipFree(stage));
    return 0;
}

// ---- profiling helpers ---------------------------------------------------------------------------
hipEvent_t prof_event() {
    if (!g.prof.pool.empty()) {
        hipEvent_t e = g.prof.pool.back();
        g.prof.pool.pop_back();
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
    const bool timed = 2 * mid >= g.prof.min_n;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timed) {
        e0 = prof_event();
        e1 = prof_event();
        HIPCHK(hipEventRecord(e0, g.stream));
    }
    // one single-table launch per table, one element per lane: measured (interleaved A/B in one process,
    // profiles/r01_fold_variants.txt) 6.4-6.6 TB/s on 2^24/2^25-element tables, against 5.7-6.1 TB/s for a
    // fused three-table launch and 4.6-5.8 TB/s for grid-stride loops over 8192 workgroups
    const dim3 grid(grid_for(mid, g.fold_grid)), block(GKR_BLOCK);
    if (!g.fold_split) {
        switch (ntab) {
            case 1: hipLaunchKernelGGL(k_fold<1>, grid, block, 0, g.stream, a); break;
            case 2: hipLaunchKernelGGL(k_fold<2>, grid, block, 0, g.stream, a); break;
            case 3: hipLaunchKernelGGL(k_fold<3>, grid, block, 0, g.stream, a); break;
            case 4: hipLaunchKernelGGL(k_fold<4>, grid, block, 0, g.stream, a); break;
            case 5: hipLaunchKernelGGL(k_fold<5>, grid, block, 0, g.stream, a); break;
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
        hipLaunchKernelGGL(k_fold<1>, grid, block, 0, g.stream, one);
    }
    HIPCHK(hipGetLastError());
    if (timed) {
        HIPCHK(hipEventRecord(e1, g.stream));
        g.prof.fold_ev.emplace_back(e0, e1);
        g.prof.fold_launches++;
        g.prof.fold_bytes += 96.0 * ntab * (double)mid;
    }
    return 0;
}

template <int GATE, int ARITY, int NEV>
int launch_partial_eval_t(const DevTable* eq, const DevTable* const* x, size_t mid, const E& ark, int* nblocks) {
    PartialEvalArgs a;
    memset(&a, 0, sizeof a);
    a.eq = eq->cplanes();
    for (int k = 0; k < ARITY; k++) a.x[k] = x[k]->cplanes();
    a.mid = mid;
    a.ark = to_dev(ark);
    a.partials = g.d_partials;
    const int grid = grid_for(mid, kPartialBlocks);
    hipLaunchKernelGGL((k_partial_eval<GATE, ARITY, NEV>), dim3(grid), dim3(GKR_BLOCK), 0, g.stream, a);
    *nblocks = grid;
    return 0;
}

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
    // node without RCCL, e.g. several ranks time-sharing one GPU in the tests)
    std::vector<Ctx*> lanes;               // the lanes that carry a communicator (lane 0 = default lane)
    size_t next_lane = 0;
};
Coll gc;
const size_t kShmSlotWords = 8192;

void shm_barrier() {
    const unsigned gen = g.lc.shm->gen.load(std::memory_order_acquire);
    if (g.lc.shm->arrive.fetch_add(1, std::memory_order_acq_rel) == (unsigned)gc.world - 1) {
        g.lc.shm->arrive.store(0, std::memory_order_relaxed);
        g.lc.shm->gen.fetch_add(1, std::memory_order_release);
    } else {
        unsigned spins = 0;
        while (g.lc.shm->gen.load(std::memory_order_acquire) == gen) {
            __builtin_ia32_pause();
            if (++spins > 2000) sched_yield();   // ranks may outnumber the cores the cgroup allows
        }
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
    if (words <= g.lc.buf_words) return 0;
    if (g.lc.d_buf) (void)hipFree(g.lc.d_buf);
    if (g.lc.h_buf) (void)hipHostFree(g.lc.h_buf);
    HIPCHK(hipMalloc(&g.lc.d_buf, sizeof(unsigned long long) * words));
    HIPCHK(hipHostMalloc(&g.lc.h_buf, sizeof(unsigned long long) * words, hipHostMallocDefault));
    g.lc.buf_words = words;
    return 0;
}
#define NCCLCHK(x)                                                                             \
    do {                                                                                       \
        ncclResult_t _r = (x);                                                                 \
        if (_r != ncclSuccess) return fail("%s failed: %s", #x, gc.p_errstr ? gc.p_errstr(_r) : "?"); \
    } while (0)

// in-place sum over ranks of n u64 lanes in device memory, on the library's stream
int coll_allreduce(unsigned long long* d, int n) {
    if (g.lc.comm) {
        NCCLCHK(gc.p_allreduce(d, d, (size_t)n, ncclUint64, ncclSum, g.lc.comm, g.stream));
        return 0;
    }
    if (g.lc.shm) {
        if ((size_t)n > kShmSlotWords) return fail("shm all-reduce of %d words exceeds the slot", n);
        if (!g.lc.h_tmp) HIPCHK(hipHostMalloc(&g.lc.h_tmp, sizeof(unsigned long long) * kShmSlotWords, hipHostMallocDefault));
        HIPCHK(hipMemcpyAsync(g.lc.h_tmp, d, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
        memcpy(g.lc.shm_slots + (size_t)gc.rank * kShmSlotWords, g.lc.h_tmp, sizeof(unsigned long long) * n);
        shm_barrier();
        for (int i = 0; i < n; i++) {
            unsigned long long s = 0;
            for (int r = 0; r < gc.world; r++) s += g.lc.shm_slots[(size_t)r * kShmSlotWords + i];
            g.lc.h_tmp[i] = s;
        }
        shm_barrier();
        HIPCHK(hipMemcpyAsync(d, g.lc.h_tmp, sizeof(unsigned long long) * n, hipMemcpyHostToDevice, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
        return 0;
    }
    return 0;
}
// all-gather of `cnt` field elements per rank (host values): rank g's elements land in out[g*cnt ..].
// Implemented as an all-reduce of a zero-padded buffer (one contributor per slot: the sum is exact).
int coll_allgather(const E* mine, int cnt, std::vector<E>& out) {
    out.assign((size_t)gc.world * cnt, hfr::ZERO);
    if (gc.world == 1 && !g.force_collective) {
        for (int i = 0; i < cnt; i++) out[i] = mine[i];
        return 0;
    }
    const size_t words = (size_t)gc.world * cnt * 4;
    CHK(coll_buffers(std::max<size_t>(words, 256)));
    memset(g.lc.h_buf, 0, words * 8);
    memcpy(g.lc.h_buf + (size_t)gc.rank * cnt * 4, mine, (size_t)cnt * 32);
    HIPCHK(hipMemcpyAsync(g.lc.d_buf, g.lc.h_buf, words * 8, hipMemcpyHostToDevice, g.stream));
    CHK(coll_allreduce(g.lc.d_buf, (int)words));
    HIPCHK(hipMemcpyAsync(g.lc.h_buf, g.lc.d_buf, words * 8, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    memcpy(out.data(), g.lc.h_buf, words * 8);
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
    ga.out = g.d_small;
    hipLaunchKernelGGL(k_gather0, dim3(1), dim3(64), 0, g.stream, ga);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(g.h_small, g.d_small, 32 * ntab, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    memcpy(out, g.h_small, 32 * ntab);
    return 0;
}

// evals[t] (t < nev) for the current round.  Launches the partial evaluation, the block reduction,
// (all-reduces the limb-split sums across ranks,) copies them to the host and reduces them mod q.
int partial_evals(int gate, int arity, const DevTable* eq, const DevTable* const* x, size_t mid, const E& ark, E* evals,
                  int nev, bool collective) {
    int nblocks = 0;
    const bool timed = 2 * mid >= g.prof.min_n;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timed) {
        e0 = prof_event();
        e1 = prof_event();
        HIPCHK(hipEventRecord(e0, g.stream));
    }
    if (gate == GKRHIP_GATE_CIPHER && arity == 2) {
        CHK((launch_partial_eval_t<GKR_GATE_CIPHER, 2, 9>(eq, x, mid, ark, &nblocks)));
    } else if (gate == GKRHIP_GATE_IDENTITY && arity == 1) {
        CHK((launch_partial_eval_t<GKR_GATE_IDENTITY, 1, 3>(eq, x, mid, ark, &nblocks)));
    } else if (gate == GKRHIP_GATE_IDENTITY && arity == 2) {
        CHK((launch_partial_eval_t<GKR_GATE_IDENTITY, 2, 3>(eq, x, mid, ark, &nblocks)));
    } else if (gate == GKRHIP_GATE_ADD && arity == 2) {
        CHK((launch_partial_eval_t<GKR_GATE_ADD, 2, 3>(eq, x, mid, ark, &nblocks)));
    } else {
        return fail("unsupported gate/arity combination (gate %d, arity %d)", gate, arity);
    }
    HIPCHK(hipGetLastError());
    if (timed) {
        HIPCHK(hipEventRecord(e1, g.stream));
        g.prof.peval_ev.emplace_back(e0, e1);
        g.prof.peval_launches++;
        g.prof.peval_modmuls += (gate == GKRHIP_GATE_CIPHER ? 45.0 : 3.0) * (double)mid;
    }
    const int nwords = nev * GKR_ACC_WORDS;
    hipLaunchKernelGGL(k_reduce_partials, dim3(nwords), dim3(GKR_BLOCK), 0, g.stream, g.d_partials, g.d_sums, nblocks, nwords);
    HIPCHK(hipGetLastError());
    if (collective) CHK(coll_allreduce(g.d_sums, nwords));
    HIPCHK(hipMemcpyAsync(g.h_sums, g.d_sums, sizeof(unsigned long long) * nwords, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    for (int t = 0; t < nev; t++) evals[t] = limbs9_to_fr(g.h_sums + (size_t)t * GKR_ACC_WORDS);
    return 0;
}

int stage_coords(const E* coords, size_t n) {
    if (n > g.d_q_cap) {
        if (g.d_q) HIPCHK(hipFree(g.d_q));
        g.d_q_cap = std::max<size_t>(n, 256);
        HIPCHK(hipMalloc(&g.d_q, sizeof(Fr) * g.d_q_cap));
    }
    if (n == 0) return 0;
    std::vector<Fr> stage(n);
    for (size_t i = 0; i < n; i++) stage[i] = to_dev(coords[i]);
    HIPCHK(hipMemcpyAsync(g.d_q, stage.data(), sizeof(Fr) * n, hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));  // `stage` is pageable host memory
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
    s.q = g.d_q;
    s.q_stride = q_stride;
    s.out = thi.planes();
    s.seeds = g.d_q + ncoord;
    s.nbits = nhi;
    s.q_off = 0;
    s.tab_stride = shi;
    hipLaunchKernelGGL(k_eq_small, dim3(nq), dim3(1024), 0, g.stream, s);
    s.out = tlo.planes();
    s.seeds = g.d_q + ncoord + nq;
    s.nbits = nlo;
    s.q_off = nhi;
    s.tab_stride = slo;
    hipLaunchKernelGGL(k_eq_small, dim3(nq), dim3(1024), 0, g.stream, s);
    EqExpandArgs x;
    x.out = eq->planes();
    x.thi = thi.cplanes();
    x.tlo = tlo.cplanes();
    x.hi_stride = shi;
    x.lo_stride = slo;
    x.nclaims = nq;
    x.nlo = nlo;
    x.n = n;
    hipLaunchKernelGGL(k_eq_expand, dim3(grid_for(n, g.max_grid)), dim3(GKR_BLOCK), 0, g.stream, x);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(g.stream));
    table_release(&thi);
    table_release(&tlo);
    return 0;
}

// ---- single-point cipher sumcheck: one fused launch per round (cipher_round.hip.h) -------------------
int wait_flag(unsigned int seq) {
    volatile unsigned int* f = g.h_flag;
    unsigned long spins = 0;
    while (*f != seq) {
        __builtin_ia32_pause();
        if ((++spins & 0xfffff) == 0) {              // every ~millisecond: make sure the GPU is alive
            hipError_t e = hipStreamQuery(g.stream);
            if (e != hipSuccess && e != hipErrorNotReady) return fail("round kernel failed: %s", hipGetErrorString(e));
            if (e == hipSuccess && *f != seq) return fail("round kernel finished without publishing its result");
        }
    }
    __sync_synchronize();
    return 0;
}

template <bool FOLD, bool HAS_WJ>
void launch_cipher_round(const CipherRoundArgs& a, int grid, bool lat) {
    if (lat) hipLaunchKernelGGL((k_cipher_round_lat<FOLD, HAS_WJ>), dim3(grid), dim3(GKR_BLOCK), 0, g.stream, a);
    else hipLaunchKernelGGL((k_cipher_round<FOLD, HAS_WJ>), dim3(grid), dim3(GKR_BLOCK), 0, g.stream, a);
}

// The rounds of a single-point cipher sumcheck over tables K, S of 2^m entries (m >= 1) and coordinates
// q[0:m].  `seed` multiplies every eq weight (the shard weight; 1 on one GPU); with `collective` the
// monomial sums are all-reduced across ranks before the host reads them.  On return: c has absorbed
// eq(q_k, r_k) of every round, proof/chal hold m rounds, tail = the two remaining entries of each table
// (K_lo, K_hi, S_lo, S_hi) and r_last the last challenge (the caller applies the final fold).
int cipher_rounds(const E& ark, int m, const DevTable* K, const DevTable* S, const E* q, const E& seed, bool collective,
                  E& c, E* proof, E* chal, E tail[4], E& r_last, E* claim /* running claim, or nullptr */,
                  bool* claim_known) {
    const size_t n = (size_t)1 << m;
    const int gT = std::min(g.g_max, m - 1);           // threads of round 0 = 2^gT
    const int mU = m - 1 - gT;                         // log2(iterations of round 0)
    CHK(stage_coords(q, (size_t)m));
    DevTable pyrT, pyrU, ks, ss;
    CHK(table_alloc(&pyrT, (size_t)2 << gT));
    CHK(table_alloc(&pyrU, (size_t)2 << std::max(mU, 0)));
    CHK(table_alloc(&ks, std::max<size_t>(n / 2, 1)));
    CHK(table_alloc(&ss, std::max<size_t>(n / 2, 1)));
    PyramidArgs pa;
    pa.out = pyrT.planes();
    pa.q = g.d_q;
    pa.nc = m;
    pa.max_level = gT;
    pa.seed = to_dev(seed);
    hipLaunchKernelGGL(k_eq_suffix_pyramid, dim3(grid_for((size_t)1 << gT, 1 << 20)), dim3(GKR_BLOCK), 0, g.stream, pa);
    if (mU > 0) {
        pa.out = pyrU.planes();
        pa.nc = m - gT;                                // q[0 .. m-gT-1]; level L = eq(q[nc-L .. nc-1], .)
        pa.max_level = mU;
        pa.seed = to_dev(hfr::ONE);
        hipLaunchKernelGGL(k_eq_suffix_pyramid, dim3(grid_for((size_t)1 << mU, 1 << 20)), dim3(GKR_BLOCK), 0, g.stream, pa);
    }
    HIPCHK(hipGetLastError());
    if (collective) CHK(coll_buffers(256));

    static const hfr::u64 binom7[8] = {1, 7, 21, 35, 35, 21, 7, 1};
    E r_prev = hfr::ZERO;
    for (int k = 0; k < m; k++) {
        const size_t P = n >> (k + 1);
        const int gk = std::min(g.g_max, m - 1 - k);
        const int lj = m - 1 - k - gk;                 // log2(iterations)
        CipherRoundArgs a;
        memset(&a, 0, sizeof a);
        const bool fold = k > 0;
        a.k_src = (k <= 1 ? K : &ks)->cplanes();
        a.s_src = (k <= 1 ? S : &ss)->cplanes();
        a.k_dst = ks.planes();
        a.s_dst = ss.planes();
        const size_t offT = ((size_t)1 << gk) - 1;
        a.wt = CPlanes{pyrT.base + offT, pyrT.base + pyrT.cap + offT};
        if (lj > 0) {
            const size_t offU = ((size_t)1 << lj) - 1;
            a.wj = CPlanes{pyrU.base + offU, pyrU.base + pyrU.cap + offU};
        }
        a.P = P;
        a.lg_threads = (unsigned)gk;
        a.r = to_dev(r_prev);
        a.ark = to_dev(ark);
        a.partials = g.d_partials;
        a.counter = g.d_counter;
        a.host_out = collective ? g.lc.d_buf : g.d_round;      // sharded: sums stay on the device for the all-reduce
        a.host_flag = g.d_flag;
        a.seq = ++g.seq;
        const bool derive_m0 = claim && *claim_known;
        a.need_m0 = derive_m0 ? 0u : 1u;
        const int grid = (int)std::max<size_t>(((size_t)1 << gk) / GKR_BLOCK, 1);
        const bool timed = 2 * P >= g.prof.min_n;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (timed) {
            e0 = prof_event();
            e1 = prof_event();
            HIPCHK(hipEventRecord(e0, g.stream));
        }
        const double t_l0 = now_ms();
        // interleaved-pair (latency) variant for the rounds with one pair per lane; GKRHIP_LAT=0 never, 2 always
        const bool lat = g.lat_mode == 2 || (g.lat_mode == 1 && lj == 0);
        const bool wide = g.wide_mode && lj > 0 && derive_m0 && !lat;
        const bool late = wide && lj >= g.wt_late_lj;  // the lane weight multiplies the sums after the loop: 8 products per lane
        if (wide) {
            if (fold) {
                if (late) hipLaunchKernelGGL((k_cipher_round_wide<true, true>), dim3(grid), dim3(GKR_BLOCK), 0, g.stream, a);
                else hipLaunchKernelGGL((k_cipher_round_wide<true, false>), dim3(grid), dim3(GKR_BLOCK), 0, g.stream, a);
            } else {
                if (late) hipLaunchKernelGGL((k_cipher_round_wide<false, true>), dim3(grid), dim3(GKR_BLOCK), 0, g.stream, a);
                else hipLaunchKernelGGL((k_cipher_round_wide<false, false>), dim3(grid), dim3(GKR_BLOCK), 0, g.stream, a);
            }
        } else if (fold) {
            if (lj > 0) launch_cipher_round<true, true>(a, grid, lat);
            else launch_cipher_round<true, false>(a, grid, lat);
        } else {
            if (lj > 0) launch_cipher_round<false, true>(a, grid, lat);
            else launch_cipher_round<false, false>(a, grid, lat);
        }
        HIPCHK(hipGetLastError());
        if (timed) {
            HIPCHK(hipEventRecord(e1, g.stream));
            g.prof.peval_ev.emplace_back(e0, e1);
            g.prof.peval_launches++;
            g.prof.peval_modmuls += ((derive_m0 ? 21.0 : 23.0) + (lj > 0 && !late ? 1.0 : 0.0) + (fold ? 4.0 : 0.0)) * (double)P;
        }
        const double t_l1 = now_ms();
        const unsigned long long* words = g.h_round;
        if (collective) {
            // sums: all-reduce over ranks (exact integer sum of limb-split lanes); the tail words are
            // rank-local and are copied as they are
            CHK(coll_allreduce(g.lc.d_buf, GKR_CR_WORDS));
            HIPCHK(hipMemcpyAsync(g.lc.h_buf, g.lc.d_buf, sizeof(unsigned long long) * (GKR_CR_WORDS + 16),
                                  hipMemcpyDeviceToHost, g.stream));
            HIPCHK(hipStreamSynchronize(g.stream));
            words = g.lc.h_buf;
        } else {
            CHK(wait_flag(a.seq));
        }
        const double t_w = now_ms();
        // S_k(t) = sum_j C(7,j) M_j t^j ;  P_k(t) = c_k * ((1-q_k) + (2 q_k - 1) t) * S_k(t)
        // csp[j] = c_k * C(7,j) * M_j.  With a known claim, P_k(0) + P_k(1) = claim_k gives
        // c_k*M_0 = claim_k - q_k * sum_{j>=1} csp[j] (the verifier's round check, sumcheck/verifier.go:41-47).
        E csp[8];
        for (int j = derive_m0 ? 1 : 0; j < 8; j++)
            csp[j] = hfr::mul(c, hfr::mul(limbs9_to_fr(words + (size_t)j * GKR_ACC_WORDS), hfr::from_u64(binom7[j])));
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
        const double t_h1 = now_ms();
        chal[k] = r;
        c = hfr::mul(c, hfr::eval_eq(&q[k], &r, 1));
        r_prev = r;
        if (claim) {   // next round's claim = P_k(r_k)
            *claim = hfr::eval_univariate(co, 9, r);
            *claim_known = true;
        }
        if (k == m - 1) memcpy(tail, words + GKR_CR_WORDS, 4 * sizeof(E));  // written by the P == 1 launch
        g.prof.host_launch_ms += t_l1 - t_l0;
        g.prof.host_wait_ms += t_w - t_l1;
        g.prof.host_other_ms += t_h0 - t_w;
        g.prof.host_hash_ms += t_h1 - t_h0;
        g.prof.rounds++;
    }
    r_last = r_prev;
    HIPCHK(hipStreamSynchronize(g.stream));
    table_release(&pyrT);
    table_release(&pyrU);
    table_release(&ks);
    table_release(&ss);
    return 0;
}

inline E fold2(const E& lo, const E& hi, const E& r) { return hfr::add(lo, hfr::mul(hfr::sub(hi, lo), r)); }

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
    const int gamma = gc.gamma, m1 = bN - gamma;
    if (m1 < 0) return fail("bN %d is smaller than log2(world) %d", bN, gamma);
    E c = hfr::ONE, tail[4], r_last, kv, sv;
    // running claim: known from the start when the caller vouches for it, otherwise from round 1 on
    E claim = trusted_claim ? *trusted_claim : hfr::ZERO;
    bool claim_known = trusted_claim != nullptr;
    E* claim_p = track_claim ? &claim : nullptr;
    if (m1 >= 1) {
        const E seed = gamma ? shard_seed(q + m1, gamma, gc.rank) : hfr::ONE;
        CHK(cipher_rounds(ark, m1, K, S, q, seed, gamma > 0 || g.force_collective, c, proof, challenges, tail, r_last,
                          claim_p, &claim_known));
        kv = fold2(tail[0], tail[1], r_last);
        sv = fold2(tail[2], tail[3], r_last);
    } else {
        const DevTable* t[2] = {K, S};
        E v[2];
        CHK(gather0(t, 2, v));
        kv = v[0];
        sv = v[1];
    }
    if (gamma > 0) {
        const E mine[2] = {kv, sv};
        std::vector<E> all;
        CHK(coll_allgather(mine, 2, all));
        std::vector<E> k2(gc.world), s2(gc.world);
        for (int r = 0; r < gc.world; r++) {
            k2[r] = all[2 * r];
            s2[r] = all[2 * r + 1];
        }
        DevTable K2, S2;
        CHK(small_table(&K2, k2));
        CHK(small_table(&S2, s2));
        CHK(cipher_rounds(ark, gamma, &K2, &S2, q + m1, hfr::ONE, false, c, proof + (size_t)9 * m1, challenges + m1, tail,
                          r_last, claim_p, &claim_known));
        kv = fold2(tail[0], tail[1], r_last);
        sv = fold2(tail[2], tail[3], r_last);
        table_release(&K2);
        table_release(&S2);
    }
    final_claims[0] = c;
    final_claims[1] = kv;
    final_claims[2] = sv;
    return 0;
}

int gate_degree(int gate) { return gate == GKRHIP_GATE_CIPHER ? 7 : 1; }   // cipher.go:68-70; copy.go:30-32; add: linear

// The reference-shaped rounds (sumcheck/prover.go:70-76) over an Eq table and `arity` tables of 2^m entries:
// partial evaluation at t = 0..deg+1, interpolation, Fiat-Shamir, fold.  eq is folded in place, X is
// read-only (round 0 folds into scratch).  On return `last` = [Eq[0], X_1[0], ...] of this rank.
int generic_rounds(int gate, const E& ark, int arity, int m, DevTable* eq, const DevTable* const* X, bool collective,
                   E* proof, E* chal, E* last) {
    const size_t n = (size_t)1 << m;
    const int nev = gate_degree(gate) + 2;
    DevTable scratch[GKR_MAX_ARITY];
    for (int k = 0; k < arity; k++) CHK(table_alloc(&scratch[k], std::max<size_t>(n / 2, 1)));
    const DevTable* cur[GKR_MAX_ARITY + 1];
    for (int k = 0; k < arity; k++) cur[k] = X[k];
    for (int k = 0; k < m; k++) {
        const size_t mid = n >> (k + 1);
        E evals[GKR_MAX_EVALS];
        CHK(partial_evals(gate, arity, eq, cur, mid, ark, evals, nev, collective));
        E* coeffs = proof + (size_t)k * nev;
        g.lag->interpolate(coeffs, evals, nev);
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
    for (int k = 0; k < arity; k++) table_release(&scratch[k]);
    return 0;
}

// sumcheck.Prove on device-resident tables (sumcheck/prover.go:46-90).  bN = GLOBAL number of variables;
// X = this rank's shards (read-only).  proof: bN*(deg+2), challenges: bN, final: arity+1.
// trust_claims: the caller guarantees that `claims` are the true sums (gkr.Prove: every claim is a previous
// sumcheck's output).  The single-point cipher path then derives one monomial sum per round from the running
// claim instead of computing it.  Entry points that take claims from outside never set it: for them the
// output must be the reference's whatever the claims are (they only feed Fiat-Shamir there).
int sumcheck_prove_dev(int gate, const E& ark, int arity, int bN, const DevTable* const* X, const E* qprimes, int nq,
                       const E* claims, int nclaims, E* proof, E* challenges, E* final_claims, bool trust_claims = false) {
    if (arity < 1 || arity > 2) return fail("arity %d not supported (1..2)", arity);
    if (nq < 1) return fail("need at least one evaluation point");
    if (nclaims != nq && nq > 1)  // sumcheck/prover.go:113-115
        return fail("provided a multi-instance %d but the number of claims does not match %d", nq, nclaims);
    const int gamma = gc.gamma, m1 = bN - gamma;
    if (m1 < 0) return fail("bN %d is smaller than log2(world) %d", bN, gamma);
    const int nev = gate_degree(gate) + 2;

    // ---- makeEqTable (prover.go:102-144)
    std::vector<E> seeds(nq, hfr::ONE);
    int nq_used = 1;
    if (nclaims >= 1) {
        const E rho = hfr::mimc_hash(claims, (size_t)nclaims);  // computed even when unused, as the reference
        E mlt = rho;
        for (int j = 1; j < nq; j++) {
            seeds[j] = mlt;
            mlt = hfr::mul(mlt, rho);
        }
        nq_used = nq;
    }
    if (gate == GKRHIP_GATE_CIPHER && arity == 2 && nq_used == 1 && bN >= 1 && !g.force_generic)
        return sumcheck_cipher_fast(ark, bN, X[0], X[1], qprimes, proof, challenges, final_claims,
                                    (trust_claims && nclaims == 1) ? &claims[0] : nullptr, trust_claims && g.claim_trick);

    // phase 1: this rank's shard; Eq_local = sum_j seed_j * eq(q_j tail, rank) * eq(q_j[0:m1], .)
    if (gamma > 0)
        for (int j = 0; j < nq_used; j++) seeds[j] = hfr::mul(seeds[j], shard_seed(qprimes + (size_t)j * bN + m1, gamma, gc.rank));
    DevTable eq;
    CHK(table_alloc(&eq, (size_t)1 << m1));
    CHK(build_eq(&eq, qprimes, nq_used, bN, m1, seeds.data()));
    E last[GKR_MAX_ARITY + 1];
    CHK(generic_rounds(gate, ark, arity, m1, &eq, X, gamma > 0 || g.force_collective, proof, challenges, last));
    table_release(&eq);
    if (gamma > 0) {
        // phase 2: one entry per table per rank -> tables over the gamma shard bits, same rounds on every rank
        std::vector<E> all;
        CHK(coll_allgather(last, arity + 1, all));
        std::vector<std::vector<E>> cols(arity + 1, std::vector<E>(gc.world));
        for (int r = 0; r < gc.world; r++)
            for (int t = 0; t <= arity; t++) cols[t][r] = all[(size_t)r * (arity + 1) + t];
        DevTable eq2, x2[GKR_MAX_ARITY];
        const DevTable* X2[GKR_MAX_ARITY];
        CHK(small_table(&eq2, cols[0]));
        for (int t = 0; t < arity; t++) {
            CHK(small_table(&x2[t], cols[1 + t]));
            X2[t] = &x2[t];
        }
        CHK(generic_rounds(gate, ark, arity, gamma, &eq2, X2, false, proof + (size_t)nev * m1, challenges + m1, last));
        table_release(&eq2);
        for (int t = 0; t < arity; t++) table_release(&x2[t]);
    }
    for (int t = 0; t <= arity; t++) final_claims[t] = last[t];
    return 0;
}

template <int GATE, int ARITY>
void launch_gate_eval(const AssignArgs& a) {
    hipLaunchKernelGGL((k_gate_eval_batch<GATE, ARITY>), dim3(grid_for(a.n, g.max_grid)), dim3(GKR_BLOCK), 0, g.stream, a);
}
int gate_eval_dev(int gate, const E& ark, const DevTable* const* in, int arity, const DevTable* out, size_t n) {
    AssignArgs a;
    memset(&a, 0, sizeof a);
    for (int k = 0; k < arity; k++) a.in[k] = in[k]->cplanes();
    a.out = out->planes();
    a.arity = arity;
    a.n = n;
    a.ark = to_dev(ark);
    if (gate == GKRHIP_GATE_CIPHER && arity == 2) launch_gate_eval<GKR_GATE_CIPHER, 2>(a);
    else if (gate == GKRHIP_GATE_IDENTITY && arity == 1) launch_gate_eval<GKR_GATE_IDENTITY, 1>(a);
    else if (gate == GKRHIP_GATE_IDENTITY && arity == 2) launch_gate_eval<GKR_GATE_IDENTITY, 2>(a);
    else if (gate == GKRHIP_GATE_ADD && arity == 2) launch_gate_eval<GKR_GATE_ADD, 2>(a);
    else return fail("unsupported gate/arity combination (gate %d, arity %d)", gate, arity);
    HIPCHK(hipGetLastError());
    return 0;
}

// MultiLin.Evaluate (poly/multilin.go:59-66) of a table of 2^nc entries: fold chain into scratch.  The
// table is this rank's shard of 2^(nc-gamma) entries; the shard values are all-gathered and the last gamma
// coordinates are applied to the gathered (<= world-entry) table.
int evaluate_dev(const DevTable* t, int nc, const E* coords, E* out) {
    const int gamma = gc.gamma, m1 = nc - gamma;
    if (m1 < 0) return fail("Evaluate: %d coordinates for a table sharded over 2^%d ranks", nc, gamma);
    const size_t n = (size_t)1 << m1;
    DevTable s;
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

// ---- circuit description (circuit/circuit.go:11-44) -------------------------------------------------
struct Layer {
    std::vector<int> in, out;
    int gate = -1;  // -1: input layer
    E ark = hfr::ZERO;
};
typedef std::vector<Layer> Circuit;

Circuit mimc_circuit() {  // examples/mimc.go:10-37
    Circuit c(94);
    c[2].in = {0};
    c[2].gate = GKRHIP_GATE_IDENTITY;
    for (int i = 0; i < 91; i++) {
        c[i + 3].in = {2, i == 0 ? 1 : i + 2};
        c[i + 3].gate = GKRHIP_GATE_CIPHER;
        c[i + 3].ark = hfr::ARKS[i];
    }
    for (size_t l = 0; l < c.size(); l++)  // BuildCircuit
        for (int p : c[l].in) c[p].out.push_back((int)l);
    return c;
}

// circuit from a flat description (circuit/circuit.go:11-44: In given, Out computed by BuildCircuit)
int circuit_from_layers(const gkrhip_layer* layers, int n, Circuit* out) {
    if (n < 2 || n > 4096) return fail("circuit: %d layers", n);
    Circuit c(n);
    bool seen_gate = false;
    for (int l = 0; l < n; l++) {
        const gkrhip_layer& d = layers[l];
        if (d.gate < 0) {
            if (seen_gate) return fail("circuit: input layer %d after a gate layer", l);
            if (d.n_in != 0) return fail("circuit: input layer %d has inputs", l);
            continue;
        }
        seen_gate = true;
        if (d.gate != GKRHIP_GATE_IDENTITY && d.gate != GKRHIP_GATE_CIPHER && d.gate != GKRHIP_GATE_ADD)
            return fail("circuit: layer %d has unknown gate %d", l, d.gate);
        const int want = d.gate == GKRHIP_GATE_IDENTITY ? 1 : 2;
        if (d.n_in != want) return fail("circuit: layer %d: gate %d takes %d inputs, got %d", l, d.gate, want, d.n_in);
        c[l].gate = d.gate;
        memcpy(c[l].ark.l, d.ark, 32);
        if (!hfr::is_canonical(c[l].ark)) return fail("circuit: layer %d: Ark is not a canonical element", l);
        for (int k = 0; k < d.n_in; k++) {
            if (d.in[k] < 0 || d.in[k] >= l) return fail("circuit: layer %d reads layer %d (must be an earlier layer)", l, d.in[k]);
            c[l].in.push_back(d.in[k]);
        }
    }
    if (c[0].gate >= 0) return fail("circuit: no input layer");
    if (c[n - 1].gate < 0) return fail("circuit: the last layer must be a gate layer (the output)");
    for (int l = 0; l < n; l++)
        for (int p : c[l].in) c[p].out.push_back(l);
    for (int l = 0; l < n; l++) {
        if (c[l].gate < 0 && c[l].out.size() > 1)   // circuit/circuit.go:36-41
            return fail("Layer %d is an input layer but has %zu outputs", l, c[l].out.size());
        if (l < n - 1 && c[l].out.empty()) return fail("circuit: layer %d feeds nothing (only the last layer may)", l);
    }
    *out = c;
    return 0;
}

// Build-defined circuit of one GMiMC (t = 2) compression, out = GMimcT2.UpdateInplace([s0,s1],[b0,b1])[0]
// (hash/gmimc.go:52-65): inputs 0..3 = s0, s1, b0, b1; per round one add layer x' = y + b1 + Ark_i and one
// cipher layer y' = (b0 + x + Ark_i)^7; explicit copy layers for the multi-use inputs; feed-forward by two add
// layers with Ark = 0; layers that do not reach the output are pruned.
std::vector<gkrhip_layer> gmimc_t2_layers() {
    struct Tmp {
        int gate, n_in, in[2];
        E ark;
    };
    std::vector<Tmp> L;
    auto add = [&](int gate, int a, int b, const E& ark) {
        Tmp t;
        t.gate = gate;
        t.n_in = gate < 0 ? 0 : (gate == GKRHIP_GATE_IDENTITY ? 1 : 2);
        t.in[0] = a;
        t.in[1] = b;
        t.ark = ark;
        L.push_back(t);
        return (int)L.size() - 1;
    };
    for (int i = 0; i < 4; i++) add(-1, 0, 0, hfr::ZERO);
    const int cs0 = add(GKRHIP_GATE_IDENTITY, 0, 0, hfr::ZERO);
    const int cb0 = add(GKRHIP_GATE_IDENTITY, 2, 0, hfr::ZERO);
    const int cb1 = add(GKRHIP_GATE_IDENTITY, 3, 0, hfr::ZERO);
    int x = cs0, y = 1;
    for (int i = 0; i < hfr::MIMC_ROUNDS; i++) {
        const int nx = add(GKRHIP_GATE_ADD, y, cb1, hfr::ARKS[i]);
        const int ny = add(GKRHIP_GATE_CIPHER, cb0, x, hfr::ARKS[i]);
        x = nx;
        y = ny;
    }
    const int t1 = add(GKRHIP_GATE_ADD, x, cs0, hfr::ZERO);
    add(GKRHIP_GATE_ADD, t1, cb0, hfr::ZERO);
    std::vector<char> need(L.size(), 0);
    need.back() = 1;
    for (int l = (int)L.size() - 1; l >= 0; l--)
        if (need[l])
            for (int k = 0; k < L[l].n_in; k++) need[L[l].in[k]] = 1;
    for (int i = 0; i < 4; i++) need[i] = 1;
    std::vector<int> ren(L.size(), -1);
    std::vector<gkrhip_layer> out;
    for (size_t l = 0; l < L.size(); l++) {
        if (!need[l]) continue;
        ren[l] = (int)out.size();
        gkrhip_layer d;
        memset(&d, 0, sizeof d);
        d.gate = L[l].gate;
        d.n_in = L[l].n_in;
        for (int k = 0; k < d.n_in; k++) d.in[k] = ren[L[l].in[k]];
        memcpy(d.ark, L[l].ark.l, 32);
        out.push_back(d);
    }
    return out;
}

size_t proof_len(const Circuit& c, int bN) {  // hints.go:76-116
    size_t sc = 0, cl = 0, qp = 0;
    for (const Layer& l : c) {
        if (l.gate >= 0) sc += (size_t)bN * (gate_degree(l.gate) + 2);
        cl += l.out.size();
        qp += (size_t)bN * l.out.size();
    }
    return sc + cl + qp + bN;
}

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
    HIPCHK(hipStreamSynchronize(g.stream));
    s->assigned = true;
    return 0;
}

int session_prove(gkrhip_session* s, const E* qprime, E* flat) {  // gkr/prover.go:21-91
    if (!s->assigned) return fail("session is not assigned");
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

    for (int layer = L - 1; layer >= 0; layer--) {
        const Layer& lay = c[layer];
        if (lay.gate < 0) break;
        const int arity = (int)lay.in.size();
        const DevTable* X[GKR_MAX_ARITY];
        for (int k = 0; k < arity; k++) X[k] = session_table(s, lay.in[k]);
        const int nev = gate_degree(lay.gate) + 2;
        sc[layer].assign((size_t)std::max(bN, 1) * nev, hfr::ZERO);
        std::vector<E> next_q(std::max(bN, 1));
        E fin[GKR_MAX_ARITY + 1];
        const int nq = layer == L - 1 ? 1 : (int)lay.out.size();
        const int ncl = has_claims[layer] ? (int)lay.out.size() : 0;
        CHK(sumcheck_prove_dev(lay.gate, lay.ark, arity, bN, X, qps[layer].data(), nq, claims[layer].data(), ncl,
                               sc[layer].data(), next_q.data(), fin, /*trust_claims=*/true));
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
    return 0;
}

}  // namespace

// Host-buffer entry points (Fold, Evaluate, FoldedEqTable, EvalBatch, sumcheck.Prove on host tables) always
// run un-sharded on this process's GPU, whatever communicator is installed; only sessions shard.
struct LocalOnly {
    int world, rank, gamma;
    LocalOnly() : world(gc.world), rank(gc.rank), gamma(gc.gamma) {
        gc.world = 1;
        gc.rank = 0;
        gc.gamma = 0;
    }
    ~LocalOnly() {
        gc.world = world;
        gc.rank = rank;
        gc.gamma = gamma;
    }
};

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int gkrhip_init(int device_ordinal) {
    std::lock_guard<std::mutex> lk(g0.mu);
    UseLane u(&g0);
    return ctx_init(device_ordinal);
}

void gkrhip_shutdown(void) {
    std::lock_guard<std::mutex> lk(g0.mu);
    if (!g0.ready) return;
    UseLane u(&g0);
    (void)hipSetDevice(g.device);
    lane_free();
    {
        std::lock_guard<std::mutex> pl(g_pool.mu);
        for (auto& f : g_pool.free_list) (void)hipFree(f.second);
        g_pool.free_list.clear();
    }
    {
        std::lock_guard<std::mutex> ll(g_lanes_mu);
        g_lanes.erase(std::remove(g_lanes.begin(), g_lanes.end(), &g0), g_lanes.end());
    }
    delete g.lag;
    g.lag = nullptr;
    g.ready = false;
    g.device = -1;
}

int gkrhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* gkrhip_last_error(void) { return g_err.c_str(); }
const char* gkrhip_version(void) { return "gkrhip 0.1 (gfx950)"; }

int gkrhip_set_option(const char* key, long value) {
    std::lock_guard<std::mutex> lk(g0.mu);
    CHK(ensure_ctx());
    std::vector<Ctx*> lanes;
    {
        std::lock_guard<std::mutex> ll(g_lanes_mu);
        lanes = g_lanes;
    }
    for (Ctx* l : lanes) {
        if (!strcmp(key, "fold_grid")) l->fold_grid = (int)std::max(64L, value);
        else if (!strcmp(key, "fold_split")) l->fold_split = value != 0;
        else if (!strcmp(key, "g_max")) l->g_max = (int)std::max(8L, std::min(20L, value));
        else if (!strcmp(key, "lat_mode")) l->lat_mode = (int)value;
        else if (!strcmp(key, "wide_mode")) l->wide_mode = (int)value;
        else if (!strcmp(key, "wt_late_lj")) l->wt_late_lj = (int)value;
        else if (!strcmp(key, "claim_trick")) l->claim_trick = value != 0;
        else return fail("unknown option %s", key);
    }
    return 0;
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

int gkrhip_device_synchronize(void) {
    std::lock_guard<std::mutex> lk(g0.mu);
    CHK(ensure_ctx());
    std::vector<Ctx*> lanes;
    {
        std::lock_guard<std::mutex> ll(g_lanes_mu);
        lanes = g_lanes;
    }
    for (Ctx* l : lanes) HIPCHK(hipStreamSynchronize(l->stream));
    return 0;
}

int gkrhip_fold(uint64_t* table, size_t n, const uint64_t r[4]) {
    std::lock_guard<std::mutex> lk(g0.mu);
    CHK(ensure_ctx());
    if (n < 2 || (n & (n - 1))) return fail("Fold: table length %zu is not a power of two >= 2", n);
    DevTable t, o;
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
    std::lock_guard<std::mutex> lk(g0.mu);
    CHK(ensure_ctx());
    if (n < 1 || (n & (n - 1))) return fail("Evaluate: table length %zu is not a power of two", n);
    if (((size_t)1 << ncoords) != n) return fail("Evaluate: table has %zu elements but %d coordinates were given", n, ncoords);
    LocalOnly lo;
    DevTable t;
    CHK(table_alloc(&t, n));
    CHK(upload_table(&t, table, n));
    E res;
    CHK(evaluate_dev(&t, ncoords, (const E*)coords, &res));
    memcpy(out, res.l, 32);
    table_release(&t);
    return 0;
}

int gkrhip_eq_table(uint64_t* out, const uint64_t* q, int bN, const uint64_t* mult_or_null) {
    std::lock_guard<std::mutex> lk(g0.mu);
    CHK(ensure_ctx());
    if (bN < 0 || bN > 30) return fail("eq table: bN %d out of range", bN);
    DevTable t;
    const size_t n = (size_t)1 << bN;
    CHK(table_alloc(&t, n));
    E seed = hfr::ONE;
    if (mult_or_null) memcpy(seed.l, mult_or_null, 32);
    CHK(build_eq(&t, (const E*)q, 1, bN, bN, &seed));
    CHK(download_table(&t, out, n));
    table_release(&t);
    return 0;
}

int gkrhip_gate_eval_batch(int gate, const uint64_t* ark_or_null, uint64_t* res, const uint64_t* const* xs, int arity,
                           size_t n) {
    std::lock_guard<std::mutex> lk(g0.mu);
    CHK(ensure_ctx());
    if (arity < 1 || arity > 2) return fail("arity %d not supported (1..2)", arity);
    DevTable in[GKR_MAX_ARITY], out;
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
    std::lock_guard<std::mutex> lk(g0.mu);
    CHK(ensure_ctx());
    if (bN < 0 || bN > 30) return fail("bN %d out of range", bN);
    if (arity < 1 || arity > 2) return fail("arity %d not supported (1..2)", arity);
    const size_t n = (size_t)1 << bN;
    LocalOnly lo;
    DevTable tabs[GKR_MAX_ARITY];
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
        hipLaunchKernelGGL(k_random_fr_array, dim3(grid_for(s->n, g.max_grid)), dim3(GKR_BLOCK), 0, g.stream,
                           s->a[l].planes(), s->n, (unsigned long long)index_stride, (unsigned long long)index_offset);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(g.stream));
    s->have_inputs = true;
    s->assigned = false;
    return 0;
}

int gkrhip_mimc_session_assign(gkrhip_session* s) {
    SESSION_ENTER(s);
    return session_assign(s);
}

int gkrhip_mimc_session_prove(gkrhip_session* s, const uint64_t* qprime, uint64_t* flat) {
    SESSION_ENTER(s);
    return session_prove(s, (const E*)qprime, (E*)flat);
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
            (void)hipStreamSynchronize(g.stream);
            // back to the arena, not to the driver: the next session of the same size (one-shot calls from the
            // hint, one per proof) reuses the buffers instead of paying ~1 s of hipMalloc/hipFree for 50 GB;
            // table_alloc drops the cache when an allocation fails
            for (auto& t : s->a) table_release(&t);
        }
        bool owned = s->lane != &g0;
        for (Ctx* l : gc.lanes) owned = owned && l != s->lane;   // communicator lanes outlive their sessions
        if (owned) lane_destroy(s->lane);
    }
    delete s;
}

int gkrhip_gkr_prove_mimc(int bN, const uint64_t* in0, const uint64_t* in1, const uint64_t* qprime, uint64_t* flat,
                          uint64_t* outputs_or_null) {
    gkrhip_session* s = nullptr;
    CHK(gkrhip_mimc_session_create(&s, bN));
    int rc = gkrhip_mimc_session_load_inputs(s, in0, in1);
    if (rc == 0) rc = gkrhip_mimc_session_assign(s);
    if (rc == 0) rc = gkrhip_mimc_session_prove(s, qprime, flat);
    if (rc == 0 && outputs_or_null) rc = gkrhip_mimc_session_outputs(s, outputs_or_null);
    gkrhip_mimc_session_destroy(s);
    return rc;
}

// ---- wire-format helpers (prover/gadget/hints.go) ---------------------------------------------------------
static int convert_inplace(uint64_t* data, size_t n, const E& factor) {
    std::lock_guard<std::mutex> lk(g0.mu);
    CHK(ensure_ctx());
    if (n == 0) return 0;
    uint4* d = nullptr;
    HIPCHK(hipMalloc(&d, 32 * n));
    HIPCHK(hipMemcpyAsync(d, data, 32 * n, hipMemcpyHostToDevice, g.stream));
    hipLaunchKernelGGL(k_convert_aos, dim3(grid_for(n, g.max_grid)), dim3(GKR_BLOCK), 0, g.stream, d, n, to_dev(factor));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(data, d, 32 * n, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    HIPCHK(hipFree(d));
    return 0;
}
int gkrhip_to_regular(uint64_t* data, size_t n) {
    const E one = {{1, 0, 0, 0}};
    return convert_inplace(data, n, one);
}
int gkrhip_from_regular(uint64_t* data, size_t n) { return convert_inplace(data, n, hfr::R2); }

int gkrhip_mimc_permutation_batch(uint64_t* out, const uint64_t* x, const uint64_t* key, size_t n) {
    std::lock_guard<std::mutex> lk(g0.mu);
    CHK(ensure_ctx());
    if (n == 0) return 0;
    DevTable tx, tk, to;
    CHK(table_alloc(&tx, n));
    CHK(table_alloc(&tk, n));
    CHK(table_alloc(&to, n));
    CHK(upload_table(&tx, x, n));
    CHK(upload_table(&tk, key, n));
    hipLaunchKernelGGL(k_mimc_permutation, dim3(grid_for(n, g.max_grid)), dim3(GKR_BLOCK), 0, g.stream, tx.cplanes(),
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
        E gate_val;
        if (c[layer].gate == GKRHIP_GATE_CIPHER) gate_val = hfr::pow7(hfr::add(hfr::add(sub[1], c[layer].ark), sub[0]));
        else if (c[layer].gate == GKRHIP_GATE_ADD) gate_val = hfr::add(hfr::add(sub[0], sub[1]), c[layer].ark);
        else gate_val = sub[0];
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
    std::lock_guard<std::mutex> lk(g0.mu);
    CHK(ensure_ctx());
    LocalOnly lo;
    if (bN < 0 || bN > 28) return fail("bN %d out of range", bN);
    const size_t n = (size_t)1 << bN;
    const Circuit c = mimc_circuit();
    DevTable t[3];
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

int gkrhip_mimc_session_verify(gkrhip_session* s, const uint64_t* qprime, const uint64_t* flat) {
    SESSION_ENTER(s);
    if (!s->assigned) return fail("session is not assigned");
    auto eval = [&](int layer, const E* pt, E* out) -> int { return evaluate_dev(session_table(s, layer), s->bN, pt, out); };
    const int rc = verify_flat(s->c, s->bN, (const E*)flat, (const E*)qprime, eval);
    if (rc > 0) fail("gkr.Verify rejected the proof (code %d)", rc);
    return rc;
}

int gkrhip_bench_fold(size_t n, int ntab, int warmup, int iters, double* avg_ms) {
    std::lock_guard<std::mutex> lk(g0.mu);
    CHK(ensure_ctx());
    if (n < 2 || (n & (n - 1)) || ntab < 1 || ntab > GKR_MAX_ARITY + 1) return fail("bench_fold: bad arguments");
    std::vector<DevTable> src(ntab), dst(ntab);
    const DevTable* sp[GKR_MAX_ARITY + 1];
    const DevTable* dp[GKR_MAX_ARITY + 1];
    for (int t = 0; t < ntab; t++) {
        CHK(table_alloc(&src[t], n));
        CHK(table_alloc(&dst[t], n / 2));
        hipLaunchKernelGGL(k_iota, dim3(grid_for(n, g.max_grid)), dim3(GKR_BLOCK), 0, g.stream, src[t].planes(), n);
        HIPCHK(hipGetLastError());
        sp[t] = &src[t];
        dp[t] = &dst[t];
    }
    const E r = hfr::from_u64(5);
    const size_t saved_min = g.prof.min_n;
    g.prof.min_n = (size_t)1 << 62;  // keep these launches out of the profile accounting
    for (int i = 0; i < warmup; i++) CHK(launch_fold(sp, dp, ntab, n / 2, r));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, g.stream));
    for (int i = 0; i < iters; i++) CHK(launch_fold(sp, dp, ntab, n / 2, r));
    HIPCHK(hipEventRecord(e1, g.stream));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = (double)ms / iters;
    g.prof.min_n = saved_min;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    for (int t = 0; t < ntab; t++) {
        table_free(&src[t]);
        table_free(&dst[t]);
    }
    return 0;
}

int gkrhip_profile_reset(size_t min_n) {
    {
        std::lock_guard<std::mutex> lk(g0.mu);
        CHK(ensure_ctx());
    }
    std::vector<Ctx*> lanes;
    {
        std::lock_guard<std::mutex> ll(g_lanes_mu);
        lanes = g_lanes;
    }
    for (Ctx* l : lanes) {
        std::lock_guard<std::mutex> lk(l->mu);
        UseLane u(l);
        HIPCHK(hipStreamSynchronize(g.stream));
        for (auto& p : g.prof.fold_ev) {
            g.prof.pool.push_back(p.first);
            g.prof.pool.push_back(p.second);
        }
        for (auto& p : g.prof.peval_ev) {
            g.prof.pool.push_back(p.first);
            g.prof.pool.push_back(p.second);
        }
        g.prof.fold_ev.clear();
        g.prof.peval_ev.clear();
        g.prof.fold_launches = g.prof.peval_launches = 0;
        g.prof.fold_bytes = g.prof.peval_modmuls = 0;
        g.prof.host_hash_ms = g.prof.host_wait_ms = g.prof.host_launch_ms = g.prof.host_other_ms = 0;
        g.prof.rounds = 0;
        g.prof.min_n = min_n == 0 ? ((size_t)1 << 62) : min_n;
    }
    return 0;
}

int gkrhip_profile_get(uint64_t* fold_launches, double* fold_ms, double* fold_bytes, uint64_t* peval_launches,
                       double* peval_ms, double* peval_modmuls) {
    {
        std::lock_guard<std::mutex> lk(g0.mu);
        CHK(ensure_ctx());
    }
    double fm = 0, pm = 0, fb = 0, pmm = 0;
    uint64_t fl = 0, pl = 0;
    std::vector<Ctx*> lanes;
    {
        std::lock_guard<std::mutex> ll(g_lanes_mu);
        lanes = g_lanes;
    }
    for (Ctx* l : lanes) {
        std::lock_guard<std::mutex> lk(l->mu);
        UseLane u(l);
        HIPCHK(hipStreamSynchronize(g.stream));
        for (auto& p : g.prof.fold_ev) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, p.first, p.second));
            fm += ms;
        }
        for (auto& p : g.prof.peval_ev) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, p.first, p.second));
            pm += ms;
        }
        fl += g.prof.fold_launches;
        pl += g.prof.peval_launches;
        fb += g.prof.fold_bytes;
        pmm += g.prof.peval_modmuls;
    }
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
    if (rank == 0) {
        fd = shm_open(name, O_CREAT | O_RDWR | O_TRUNC, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) return fail("shm_open/ftruncate(%s) failed", name);
    } else {
        for (int tries = 0; tries < 20000; tries++) {   // wait for rank 0 to create and size the segment
            fd = shm_open(name, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes) break;
            if (fd >= 0) close(fd);
            fd = -1;
            usleep(1000);
        }
        if (fd < 0) return fail("shm segment %s did not appear", name);
    }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return fail("mmap of %s failed", name);
    g.lc.shm = (ShmHdr*)p;
    g.lc.shm_slots = (unsigned long long*)((char*)p + 4096);
    g.lc.shm_bytes = bytes;
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
        NCCLCHK(gc.p_init(&g.lc.comm, world, id, rank));
        CHK(coll_buffers(4096));
    }
    comm_set(world, rank);
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
        shm_barrier();   // everybody mapped (the segments are zero-filled by ftruncate)
    }
    return 0;
}

int gkrhip_comm_init_shm(int world, int rank, const char* name) { return gkrhip_comm_init_shm_lanes(world, rank, 1, name); }

int gkrhip_comm_destroy(void) {
    std::lock_guard<std::mutex> lk(g0.mu);
    for (Ctx* l : gc.lanes) {
        {
            std::unique_lock<std::mutex> ll;
            if (l != &g0) ll = std::unique_lock<std::mutex>(l->mu);   // g0.mu is already held
            UseLane u(l);
            (void)hipStreamSynchronize(g.stream);
            if (g.lc.comm) {
                (void)gc.p_destroy(g.lc.comm);
                g.lc.comm = nullptr;
            }
            if (g.lc.shm) {
                munmap((void*)g.lc.shm, g.lc.shm_bytes);
                g.lc.shm = nullptr;
                g.lc.shm_slots = nullptr;
            }
        }
        if (l != &g0) lane_destroy(l);
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

int gkrhip_profile_host(uint64_t* rounds, double* hash_ms, double* wait_ms, double* launch_ms, double* other_ms) {
    uint64_t r = 0;
    double h = 0, w = 0, l_ = 0, o = 0;
    std::vector<Ctx*> lanes;
    {
        std::lock_guard<std::mutex> ll(g_lanes_mu);
        lanes = g_lanes;
    }
    for (Ctx* l : lanes) {
        std::lock_guard<std::mutex> lk(l->mu);
        r += l->prof.rounds;
        h += l->prof.host_hash_ms;
        w += l->prof.host_wait_ms;
        l_ += l->prof.host_launch_ms;
        o += l->prof.host_other_ms;
    }
    if (rounds) *rounds = r;
    if (hash_ms) *hash_ms = h;
    if (wait_ms) *wait_ms = w;
    if (launch_ms) *launch_ms = l_;
    if (other_ms) *other_ms = o;
    return 0;
}

}  // extern "C"
