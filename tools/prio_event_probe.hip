// prio_event_probe.hip -- does a kernel on a LOWEST-priority stream always leave complete output behind its event?
//
// Round 4 found 4 % wrong proofs (57 of 1 440 at bN = 18) with the look-ahead kernel (k_cipher_pre, cipher_round.hip.h) on a
// lowest-priority non-blocking stream and twelve lanes forced to use it, 0 of 1 440 at normal priority -- although the consumer
// is queued behind hipStreamWaitEvent(stream, pre_done).  This probe is that shape WITHOUT the library, to tell a runtime /
// hardware behaviour from a library bug:
//   per lane: one host thread, a main stream (non-blocking, normal priority), an aux stream (non-blocking, priority under test),
//   one event, six output tables of P x 32 bytes (two planes of 16-byte words, as DevTable), two source tables;
//   per iteration:  producer on aux -- reads the sources, a chain of integer products per element, six NON-TEMPORAL 16-byte
//                   store pairs per element, `lds_kb` of unused dynamic LDS (one workgroup per CU), grid min(P / 256, 4096) --,
//                   hipEventRecord(ev, aux);
//                   `rounds` small kernels on main, each polling a host-mapped word the host publishes ~35 us later (the
//                   pre-launched rounds) and acknowledging through host memory, plus one VALU-heavy kernel (a big round);
//                   hipStreamSynchronize(main); hipStreamWaitEvent(main, ev); consumer on main: non-temporal loads of all
//                   six tables, every word compared with the value the producer must have written for THIS iteration;
//                   mismatches are counted, the first one is recorded (table, index, got, want, stale-by-one-iteration?).
// Modes (argv): lanes seconds prio(low|normal|high) hwq lds_kb logP rounds event(1 re-recorded|0 fresh|2 alternating) plain_stores(0|1)
//               producer_release_fence(0|1) query_before_wait(0|1)
//   hipcc --offload-arch=gfx950 -O3 tools/prio_event_probe.hip -o /tmp/prio_event_probe
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__device__ __forceinline__ unsigned mix(unsigned x, unsigned it, unsigned t, int work) {
    unsigned v = x * 2654435761u + it * 40503u + t * 977u + 1u;
    for (int i = 0; i < work; i++) v = v * 1664525u + 1013904223u + (v >> 13);      // a dependent chain: the kernel is VALU-bound like k_cipher_pre
    return v;
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 word_of(unsigned v, unsigned half) { u32x4 r = {v, v ^ 0x9e3779b9u, v + half, ~v}; return r; }

struct ProdArgs {
    const uint4* src;      // 2 tables x 2 planes x 2P (read like K and S)
    uint4* out[6];         // plane lo at out[t], plane hi at out[t] + P
    size_t P;
    unsigned it;
    int work;
    int plain;             // 1: plain stores instead of non-temporal ones
    int fence;             // 1: every workgroup ends with an agent-scope release fence (buffer_wbl2 sc1) of its own
};
__global__ void __launch_bounds__(256, 2) k_producer(ProdArgs a) {
    extern __shared__ unsigned char unused_lds[];
    for (size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x; x < a.P; x += (size_t)gridDim.x * blockDim.x) {
        const uint4 s0 = a.src[x], s1 = a.src[x + a.P];
        const unsigned seed = (unsigned)x ^ (s0.x & 0u) ^ (s1.y & 0u);      // the loads are real, the value does not depend on them
#pragma unroll
        for (int t = 0; t < 6; t++) {
            const unsigned v = mix(seed, a.it, (unsigned)t, a.work);
            if (a.plain) {
                *(u32x4*)(a.out[t] + x) = word_of(v, 0u);
                *(u32x4*)(a.out[t] + a.P + x) = word_of(v, 1u);
            } else {
                __builtin_nontemporal_store(word_of(v, 0u), (u32x4*)(a.out[t] + x));
                __builtin_nontemporal_store(word_of(v, 1u), (u32x4*)(a.out[t] + a.P + x));
            }
        }
    }
    if (a.fence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}

struct ConsArgs {
    const uint4* in[6];
    size_t P;
    unsigned it;
    int work;
    unsigned long long* bad;       // device: [0] count, [1] first bad (table << 56 | plane << 48 | index), [2] got.x, [3] want.x, [4] stale?
    unsigned int* counter;         // arrival counter
    unsigned long long* host_out;  // host-mapped: 5 words + flag word [7] = iteration
};
__global__ void __launch_bounds__(256, 2) k_consumer(ConsArgs a) {
    unsigned long long mine = 0;
    for (size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x; x < a.P; x += (size_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int t = 0; t < 6; t++) {
            const unsigned v = mix((unsigned)x, a.it, (unsigned)t, a.work);
            for (unsigned h = 0; h < 2; h++) {
                const u32x4 got = __builtin_nontemporal_load((const u32x4*)(a.in[t] + h * a.P + x));
                const u32x4 want = word_of(v, h);
                if (got.x != want.x || got.y != want.y || got.z != want.z || got.w != want.w) {
                    mine++;
                    if (atomicAdd(&a.bad[0], 1ull) == 0) {
                        a.bad[1] = ((unsigned long long)t << 56) | ((unsigned long long)h << 48) | (unsigned long long)x;
                        a.bad[2] = got.x;
                        a.bad[3] = want.x;
                        a.bad[4] = got.x == mix((unsigned)x, a.it - 1, (unsigned)t, a.work) ? 1 : 0;      // the previous iteration's value?
                    }
                }
            }
        }
    }
    __shared__ unsigned last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(a.counter, 1u) == gridDim.x - 1;
    __syncthreads();
    if (last && threadIdx.x == 0) {
        __threadfence();
        for (int i = 0; i < 5; i++) a.host_out[i] = __hip_atomic_load(a.bad + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = 0; i < 5; i++) a.bad[i] = 0;
        *a.counter = 0;
        __threadfence_system();
        __hip_atomic_store(a.host_out + 7, (unsigned long long)a.it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    (void)mine;
}

// a small round: polls a host-mapped word, a little arithmetic, acknowledges through host memory
__global__ void k_small(const unsigned long long* slot, unsigned long long* ack, unsigned long long want) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != want) {
            if (wall_clock64() - t0 > 200000000ull) break;      // 2 s
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    float v = threadIdx.x;
    for (int i = 0; i < 3000; i++) v = v * 1.0001f + 0.5f;
    if (v == 12345.678f) ack[1] = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(ack, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// a big round: VALU-bound, one workgroup per CU and more
__global__ void __launch_bounds__(256, 2) k_big(unsigned* sink, int work) {
    unsigned v = blockIdx.x * 256u + threadIdx.x;
    for (int i = 0; i < work; i++) v = v * 1664525u + 1013904223u + (v >> 11);
    if (v == 0x12345u) sink[0] = v;
}

int main(int argc, char** argv) {
    const int lanes = argc > 1 ? atoi(argv[1]) : 12;
    const double seconds = argc > 2 ? atof(argv[2]) : 20;
    const std::string prio = argc > 3 ? argv[3] : "low";
    const int hwq = argc > 4 ? atoi(argv[4]) : 16;
    const int lds_kb = argc > 5 ? atoi(argv[5]) : 100;
    const int logP = argc > 6 ? atoi(argv[6]) : 17;
    const int rounds = argc > 7 ? atoi(argv[7]) : 6;
    const int reuse_event = argc > 8 ? atoi(argv[8]) : 1;      // 1: one event re-recorded; 0: a fresh event per iteration; 2: two events alternating
    const int plain = argc > 9 ? atoi(argv[9]) : 0;
    const int fence = argc > 10 ? atoi(argv[10]) : 0;
    const int query = argc > 11 ? atoi(argv[11]) : 0;          // 1: hipEventQuery before the wait; wrong iterations are split by its answer
    if (hwq > 0) setenv("GPU_MAX_HW_QUEUES", std::to_string(hwq).c_str(), 1);
    CK(hipSetDevice(0));
    int lo_p = 0, hi_p = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));      // lo_p = numerically greatest = lowest priority
    const int aux_prio = prio == "low" ? lo_p : prio == "high" ? hi_p : 0;
    const size_t P = (size_t)1 << logP;
    const int work = 48;
    CK(hipFuncSetAttribute((const void*)k_producer, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024));
    printf("prio_event_probe: %d lanes, %.0f s, aux priority %s (%d of [%d..%d]), GPU_MAX_HW_QUEUES=%d, %d KB dynamic LDS, P = 2^%d, %d small rounds, %s event\n",
           lanes, seconds, prio.c_str(), aux_prio, lo_p, hi_p, hwq, lds_kb, logP, rounds, reuse_event == 1 ? "one re-recorded" : reuse_event == 2 ? "two alternating" : "a fresh");
    printf("  stores %s, producer release fence %d, hipEventQuery before the wait %d\n", plain ? "plain" : "non-temporal", fence, query);
    std::atomic<unsigned long long> iters{0}, bad_iters{0}, bad_words{0}, stale{0}, q_ready{0}, q_notready{0}, bad_ready{0}, bad_notready{0};
    std::vector<std::thread> th;
    for (int l = 0; l < lanes; l++)
        th.emplace_back([&, l]() {
            CK(hipSetDevice(0));
            hipStream_t mainS, auxS;
            CK(hipStreamCreateWithFlags(&mainS, hipStreamNonBlocking));
            CK(hipStreamCreateWithPriority(&auxS, hipStreamNonBlocking, aux_prio));
            hipEvent_t ev, ev2;
            CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            CK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
            uint4 *src, *out[6];
            CK(hipMalloc(&src, sizeof(uint4) * 4 * P));
            CK(hipMemsetAsync(src, 1, sizeof(uint4) * 4 * P, mainS));
            for (auto& o : out) {
                CK(hipMalloc(&o, sizeof(uint4) * 2 * P));
                CK(hipMemsetAsync(o, 0, sizeof(uint4) * 2 * P, mainS));
            }
            unsigned long long *bad, *h_out, *d_out, *h_slot, *d_slot, *h_ack, *d_ack;
            unsigned int* counter;
            unsigned* sink;
            CK(hipMalloc(&bad, 64));
            CK(hipMalloc(&counter, 4));
            CK(hipMalloc(&sink, 4));
            CK(hipMemsetAsync(bad, 0, 64, mainS));
            CK(hipMemsetAsync(counter, 0, 4, mainS));
            CK(hipHostMalloc(&h_out, 64, hipHostMallocMapped | hipHostMallocCoherent));
            CK(hipHostMalloc(&h_slot, 64, hipHostMallocMapped | hipHostMallocCoherent));
            CK(hipHostMalloc(&h_ack, 64, hipHostMallocMapped | hipHostMallocCoherent));
            memset(h_out, 0, 64);
            memset(h_slot, 0, 64);
            memset(h_ack, 0, 64);
            CK(hipHostGetDevicePointer((void**)&d_out, h_out, 0));
            CK(hipHostGetDevicePointer((void**)&d_slot, h_slot, 0));
            CK(hipHostGetDevicePointer((void**)&d_ack, h_ack, 0));
            CK(hipStreamSynchronize(mainS));
            const double T0 = now_us();
            unsigned long long seq = 0;
            const int grid = (int)std::min<size_t>(P / 256, 4096);
            for (unsigned it = 1; now_us() - T0 < seconds * 1e6; it++) {
                ProdArgs pa;
                pa.src = src;
                for (int t = 0; t < 6; t++) pa.out[t] = out[t];
                pa.P = P;
                pa.it = it;
                pa.work = work;
                pa.plain = plain;
                pa.fence = fence;
                if (reuse_event == 2) std::swap(ev, ev2);
                hipLaunchKernelGGL(k_producer, dim3(grid), dim3(256), (size_t)lds_kb * 1024, auxS, pa);
                if (!reuse_event) {
                    CK(hipEventDestroy(ev));
                    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                }
                CK(hipEventRecord(ev, auxS));
                hipLaunchKernelGGL(k_big, dim3(512), dim3(256), 0, mainS, sink, 20000);
                for (int r = 0; r < rounds; r++) {
                    seq++;
                    hipLaunchKernelGGL(k_small, dim3(4), dim3(256), 0, mainS, d_slot, d_ack, seq);
                    const double h0 = now_us();
                    while (now_us() - h0 < 35.0) {}
                    *(volatile unsigned long long*)h_slot = seq;
                    __sync_synchronize();
                    const double w0 = now_us();
                    while (*(volatile unsigned long long*)h_ack != seq)
                        if (now_us() - w0 > 5e6) { printf("lane %d: a small round never acknowledged\n", l); exit(2); }
                }
                CK(hipStreamSynchronize(mainS));
                int was_ready = -1;
                if (query) {
                    const hipError_t qe = hipEventQuery(ev);
                    was_ready = qe == hipSuccess;
                    (was_ready ? q_ready : q_notready)++;
                }
                CK(hipStreamWaitEvent(mainS, ev, 0));
                ConsArgs ca;
                for (int t = 0; t < 6; t++) ca.in[t] = out[t];
                ca.P = P;
                ca.it = it;
                ca.work = work;
                ca.bad = bad;
                ca.counter = counter;
                ca.host_out = d_out;
                hipLaunchKernelGGL(k_consumer, dim3(256), dim3(256), 0, mainS, ca);
                const double w0 = now_us();
                while (*(volatile unsigned long long*)(h_out + 7) != it)
                    if (now_us() - w0 > 20e6) { printf("lane %d: the consumer of iteration %u never finished\n", l, it); exit(2); }
                iters++;
                if (h_out[0]) {
                    bad_iters++;
                    bad_words += h_out[0];
                    stale += h_out[4];
                    if (was_ready == 1) bad_ready++;
                    if (was_ready == 0) bad_notready++;
                    if (bad_iters.load() <= 10)
                        printf("lane %d iteration %u: %llu wrong words; first: table %llu plane %llu index %llu got %08llx want %08llx%s\n", l, it, h_out[0],
                               h_out[1] >> 56, (h_out[1] >> 48) & 0xff, h_out[1] & 0xffffffffffffull, h_out[2], h_out[3],
                               h_out[4] ? " (= the PREVIOUS iteration's value: the consumer ran before the producer had written it)" : "");
                }
            }
            CK(hipStreamSynchronize(mainS));
            CK(hipStreamSynchronize(auxS));
        });
    for (auto& t : th) t.join();
    printf("RESULT prio=%s hwq=%d lanes=%d lds_kb=%d logP=%d: %llu iterations, %llu with wrong words (%llu words, %llu first-mismatches stale by one iteration)\n",
           prio.c_str(), hwq, lanes, lds_kb, logP, iters.load(), bad_iters.load(), bad_words.load(), stale.load());
    if (query) printf("  hipEventQuery said ready %llu times (%llu of them wrong), not ready %llu times (%llu of them wrong)\n", q_ready.load(), bad_ready.load(),
                      q_notready.load(), bad_notready.load());
    return 0;
}
