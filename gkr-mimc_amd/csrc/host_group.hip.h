// host_group.hip.h -- proof groups: up to GKR_GROUP_MAX proofs of the same shape proven in lock-step by ONE host thread, their
// round kernels launched together (cipher_round.hip.h: Batch, blockIdx.z selects the proof).  Included by gkrhip.hip inside its
// anonymous namespace, after host_ctx.hip.h.
//
// Why: a proof of 2^20 hashes is ~1 700 dependent launches, most of them of a few microseconds; with 56 such proofs in flight
// the GPU's dispatch of tiny kernels from many queues bounds the job, not its arithmetic (profiles/r04_bn20_plateau.txt; round
// 6's A/Bs of the kernels changed nothing: profiles/r06_variants_ab.txt).  Three proofs in one launch are a third of the
// launches, each with three times the work: bN = 20 64 -> 82 M hashes/s (profiles/r06_proof_groups.txt).
//
// How: every proof of the group runs the unchanged prover (session_prove) on a stack of its own (ucontext), on its own lane's
// buffers but on the GROUP's stream.  launch_batch() -- the launch site of the batched kernels -- does not launch: it records
// what the proof wants launched and switches to the driver.  When every proof of the group has arrived at a launch, the driver
// puts the launches that agree (same kernel, grid, stream) into one and resumes the proofs in turn.  Everything else a proof
// queues (the odd memset or copy, kernels that are not batched) goes to the shared stream at once, in the proof's own order, and
// the combined launch is queued behind all of it -- stream order is what the prover relies on, and it holds.  (The coordinates
// of a layer's pyramids travel in the launch's arguments -- PyramidArgs3::qv -- so a layer queues nothing but batched kernels.)  A proof only ever waits (flag words, stream synchronisation) for work that has been queued: a proof resumes only
// after its launch is in the stream.  Proofs that part ways (a challenge retry, a layer run again in safe mode, an error) get
// launches of their own until they meet again; a proof that returns leaves the group.
#pragma once
#include <ucontext.h>
#include <sys/mman.h>

struct PendingLaunch {
    const void* fn = nullptr;
    dim3 grid, block;
    size_t shmem = 0;
    hipStream_t stream = nullptr;
    const void* arg = nullptr;      // the proof's argument struct (on the proof's stack, which is parked)
    size_t arg_size = 0;
    unsigned cap = 1;               // proofs one launch of this kernel takes (Batch<A>::N)
    hipError_t rc = hipSuccess;
};
struct GroupTls {                   // the calling thread's per-proof state
    Ctx* cur = nullptr;
    bool regular_io = false, chal_timeout = false, safe_mode = false, corrupt_collect = false;
    int local_only = 0;
    void save() {
        cur = g_cur;
        regular_io = g_regular_io;
        chal_timeout = g_chal_timeout;
        safe_mode = g_safe_mode;
        corrupt_collect = g_corrupt_collect;
        local_only = g_local_only;
    }
    void load() const {
        g_cur = cur;
        g_regular_io = regular_io;
        g_chal_timeout = chal_timeout;
        g_safe_mode = safe_mode;
        g_corrupt_collect = corrupt_collect;
        g_local_only = local_only;
    }
};
struct Group;
struct GroupProof {
    Group* g = nullptr;
    ucontext_t uc;
    void* stack = nullptr;
    size_t stack_bytes = 0;
    GroupTls tls;
    PendingLaunch pend;
    bool at_launch = false, done = false;
    int rc = 0;
    std::function<int()> body;
};
typedef hipError_t (*GroupLaunchFn)(const void* fn, dim3 grid, dim3 block, void** params, size_t shmem, hipStream_t st);
struct Group {
    GroupLaunchFn launch = nullptr;      // nullptr: hipLaunchKernel (the host-only self-test records instead)
    ucontext_t driver;
    GroupTls driver_tls;
    std::vector<GroupProof> proofs;
    int cur = -1;
    unsigned long long combined = 0, launches = 0;      // combined launches, and what they stood for
};
thread_local Group* t_group = nullptr;
std::mutex g_group_queue_mu;
int g_group_queue_use[64] = {0};            // groups under way by the hardware queue (stream ordinal mod the queue count) they launch on
inline int hw_queue_count() {
    const int a = g_hwq_set_by_library.load(), b = g_hwq_from_env.load();
    return a > 0 ? a : (b > 0 ? b : 4);       // (4: the runtime's default)
}
std::atomic<unsigned long long> g_cnt_group_launches{0}, g_cnt_group_combined{0};
std::atomic<int> g_group_size{3};                     // option group_size: single calls that meet form groups of this many (gkrhip_mimc_session_prove)
// option group_wait_us: how long the first caller of a group waits for company.  A caller that went alone once comes back out of step
// with everybody else and goes alone again unless the wait is long enough for another one to come by: bN = 20, 72 callers: 100 us /
// 300 us / 1 ms / 3 ms / 10 ms / 30 ms -> 71.5 / 74.4-76.4 / 76.0 / 78.5-78.8 / 81.7 / 78.5 M hashes/s (explicit groups: 82;
// profiles/r06_proof_groups.txt).  10 ms is 1 % of such a group's 0.9 s and only ever spent while 24 callers are inside the call.
std::atomic<int> g_group_wait_us{10000};
std::atomic<unsigned long long> g_cnt_coalesced{0};   // proofs that were proven in a group formed from single calls

// the launch site of a batched kernel
template <class A>
inline hipError_t launch_batch(void (*kern)(Batch<A>), dim3 grid, dim3 block, size_t shmem, hipStream_t st, const A& a) {
    static_assert(sizeof(Batch<A>) <= 4096, "a batch is kernel-argument memory");
    Group* g = t_group;
    if (!g) {
        Batch<A> b;                 // grid.z == 1: only inst[0] is read
        b.inst[0] = a;
        hipLaunchKernelGGL(kern, grid, block, shmem, st, b);
        return hipGetLastError();
    }
    GroupProof& me = g->proofs[(size_t)g->cur];
    me.pend.fn = reinterpret_cast<const void*>(kern);
    me.pend.grid = grid;
    me.pend.block = block;
    me.pend.shmem = shmem;
    me.pend.stream = st;
    me.pend.arg = &a;
    me.pend.arg_size = sizeof(A);
    me.pend.cap = (unsigned)Batch<A>::N;
    me.pend.rc = hipSuccess;
    me.at_launch = true;
    me.tls.save();
    swapcontext(&me.uc, &g->driver);      // back when the launch is in the stream
    return me.pend.rc;
}

inline void group_trampoline(unsigned lo, unsigned hi) {
    GroupProof* p = (GroupProof*)(((uintptr_t)hi << 32) | (uintptr_t)lo);
    p->rc = p->body();
    p->done = true;
    p->tls.save();
    // (returns to uc_link: the driver)
}

// queue what the parked proofs asked for: launches that agree go out as one
inline void group_fire(Group& g) {
    alignas(16) unsigned char args[4096];
    const size_t n = g.proofs.size();
    for (size_t i = 0; i < n; i++) {
        GroupProof& p = g.proofs[i];
        if (!p.at_launch) continue;
        unsigned cnt = 0;
        size_t member[GKR_GROUP_MAX];
        for (size_t j = i; j < n && cnt < p.pend.cap; j++) {
            GroupProof& o = g.proofs[j];
            const PendingLaunch &x = p.pend, &y = o.pend;
            if (!o.at_launch || x.fn != y.fn || x.stream != y.stream || x.shmem != y.shmem || x.arg_size != y.arg_size || x.grid.x != y.grid.x ||
                x.grid.y != y.grid.y || x.grid.z != y.grid.z || x.block.x != y.block.x || x.block.y != y.block.y || x.block.z != y.block.z)
                continue;
            memcpy(args + (size_t)cnt * x.arg_size, y.arg, x.arg_size);
            member[cnt++] = j;
        }
        void* params[1] = {args};
        const hipError_t rc = (g.launch ? g.launch : (GroupLaunchFn)hipLaunchKernel)(p.pend.fn, dim3(p.pend.grid.x, p.pend.grid.y, cnt), p.pend.block, params,
                                                                                     p.pend.shmem, p.pend.stream);
        if (rc != hipSuccess) (void)hipGetLastError();      // handed to the proofs below
        for (unsigned c = 0; c < cnt; c++) {
            g.proofs[member[c]].pend.rc = rc;
            g.proofs[member[c]].at_launch = false;
        }
        g.combined++;
        g.launches += cnt;
    }
}

// run the bodies to completion on the calling thread; proofs[i].rc holds each body's return value
inline int group_run(Group& g) {
    if (t_group) return fail("a proof group cannot start inside another");
    const size_t kStack = (size_t)2 << 20;
    int rc = 0;
    for (GroupProof& p : g.proofs) {
        p.g = &g;
        p.stack_bytes = kStack;
        p.stack = mmap(nullptr, kStack + 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
        if (p.stack == MAP_FAILED) {
            p.stack = nullptr;
            rc = fail("proof group: cannot map a stack");
            break;
        }
        (void)mprotect(p.stack, 4096, PROT_NONE);      // the page under the stack
        getcontext(&p.uc);
        p.uc.uc_stack.ss_sp = (char*)p.stack + 4096;
        p.uc.uc_stack.ss_size = kStack;
        p.uc.uc_link = &g.driver;
        const uintptr_t ptr = (uintptr_t)&p;
        makecontext(&p.uc, (void (*)())group_trampoline, 2, (unsigned)(ptr & 0xffffffffu), (unsigned)(ptr >> 32));
        p.tls.save();                                  // starts from the caller's state (the body picks its lane)
    }
    if (rc == 0) {
        g.driver_tls.save();
        t_group = &g;
        const int passengers = (int)g.proofs.size() - 1;
        g_group_passengers.fetch_add(passengers, std::memory_order_relaxed);
        t_group_size = (int)g.proofs.size();
        for (;;) {
            // what the proofs' plans read until they are parked again: one number for all of them
            t_group_in_flight = std::max(g_proofs_in_flight.load(std::memory_order_relaxed), (int)g.proofs.size());
            bool any = false;
            for (size_t i = 0; i < g.proofs.size(); i++) {
                GroupProof& p = g.proofs[i];
                if (p.done) continue;
                any = true;
                g.cur = (int)i;
                p.tls.load();
                swapcontext(&g.driver, &p.uc);         // until its next launch, or its end
                g.driver_tls.load();
            }
            if (!any) break;
            group_fire(g);
        }
        g_group_passengers.fetch_sub(passengers, std::memory_order_relaxed);
        t_group_in_flight = -1;
        t_group_size = 0;
        t_group = nullptr;
        g.cur = -1;
        g_cnt_group_launches.fetch_add(g.launches, std::memory_order_relaxed);
        g_cnt_group_combined.fetch_add(g.combined, std::memory_order_relaxed);
    }
    for (GroupProof& p : g.proofs)
        if (p.stack) munmap(p.stack, p.stack_bytes + 4096);
    return rc;
}

// a batched kernel's launch site (inside a function that returns the library's int codes)
#define GKR_LAUNCH_BATCH(kern, grid, block, shmem, stream, a)                                               \
    do {                                                                                                    \
        const hipError_t e_ = launch_batch(kern, grid, block, shmem, stream, a);                            \
        if (e_ != hipSuccess) return fail("launch of %s: %s", #kern, hipGetErrorString(e_));                 \
    } while (0)
