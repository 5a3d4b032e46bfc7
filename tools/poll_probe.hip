// poll_probe.hip -- how long does a kernel that polls a host-mapped word take to see the host's write, with many such kernels
// (streams, host threads) at once?  The shape of the pre-launched round kernels' challenge hand-over (cipher_round.hip.h,
// wait_challenge), reduced to its bones: per lane one stream, one host thread and a chain of kernels; kernel i polls slot until
// it holds i (system-scope atomic loads from `pollers` workgroups, s_sleep between), then writes ack = i to host memory; the host
// thread waits ~35 us (the hash), writes i, and times write -> ack.  Prints the distribution and the worst case per lane.
//   hipcc --offload-arch=gfx950 -O3 tools/poll_probe.hip -o /tmp/poll_probe && /tmp/poll_probe <lanes> <rounds> <pollers> <wgs>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void k_wait(const unsigned long long* slot, unsigned long long* ack, unsigned long long want, int pollers, unsigned long long* gave_up) {
    __shared__ int ok;
    if (threadIdx.x == 0) ok = 1;
    __syncthreads();
    if (threadIdx.x < 16 && (int)blockIdx.x < pollers) {          // sixteen lanes, sixteen tagged words: one 128-byte read per poll
        const unsigned long long t0 = wall_clock64();
        while ((__hip_atomic_load(slot + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >> 32) != want) {
            if (wall_clock64() - t0 > 300000000ull) { ok = 0; break; }      // 3 s
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    // some arithmetic so that the launch occupies its slots for a few microseconds like a small round kernel
    float v = threadIdx.x;
    for (int i = 0; i < 2000; i++) v = v * 1.0001f + 0.5f;
    if (v == 12345.678f) ack[1] = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (!ok) __hip_atomic_fetch_add(gave_up, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(ack, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int main(int argc, char** argv) {
    const int lanes = argc > 1 ? atoi(argv[1]) : 14, rounds = argc > 2 ? atoi(argv[2]) : 20000, pollers = argc > 3 ? atoi(argv[3]) : 1,
              wgs = argc > 4 ? atoi(argv[4]) : 64;
    std::vector<unsigned long long*> slot(lanes), ack(lanes), dslot(lanes), dack(lanes);
    unsigned long long *gave, *dgave;
    CK(hipHostMalloc(&gave, 8, hipHostMallocMapped | hipHostMallocCoherent));
    *gave = 0;
    CK(hipHostGetDevicePointer((void**)&dgave, gave, 0));
    std::vector<hipStream_t> st(lanes);
    for (int l = 0; l < lanes; l++) {
        CK(hipHostMalloc(&slot[l], 128, hipHostMallocMapped | hipHostMallocCoherent));
        CK(hipHostMalloc(&ack[l], 128, hipHostMallocMapped | hipHostMallocCoherent));
        for (int w = 0; w < 16; w++) slot[l][w] = 0;
        ack[l][0] = 0;
        CK(hipHostGetDevicePointer((void**)&dslot[l], slot[l], 0));
        CK(hipHostGetDevicePointer((void**)&dack[l], ack[l], 0));
        CK(hipStreamCreateWithFlags(&st[l], hipStreamNonBlocking));
    }
    std::vector<std::vector<double>> rtt(lanes);
    std::vector<std::thread> th;
    for (int l = 0; l < lanes; l++)
        th.emplace_back([&, l]() {
            CK(hipSetDevice(0));
            rtt[l].reserve(rounds);
            hipLaunchKernelGGL(k_wait, dim3(wgs), dim3(256), 0, st[l], dslot[l], dack[l], 1ull, pollers, dgave);       // pre-launched
            for (unsigned long long i = 1; i <= (unsigned long long)rounds; i++) {
                if (i < (unsigned long long)rounds) hipLaunchKernelGGL(k_wait, dim3(wgs), dim3(256), 0, st[l], dslot[l], dack[l], i + 1, pollers, dgave);
                const double th0 = now_us();
                while (now_us() - th0 < 35.0) {}                      // the hash
                const double t0 = now_us();
                for (int w = 0; w < 16; w++) ((volatile unsigned long long*)slot[l])[w] = (i << 32) | (unsigned)(w * 2654435761u + i);
                __sync_synchronize();
                while (*(volatile unsigned long long*)ack[l] != i) {
                    if (now_us() - t0 > 5e6) { printf("lane %d round %llu: no ack after 5 s (ack %llu)\n", l, i, ack[l][0]); exit(2); }
                }
                rtt[l].push_back(now_us() - t0);
            }
        });
    const double T0 = now_us();
    for (auto& t : th) t.join();
    const double T1 = now_us();
    std::vector<double> all;
    double worst = 0;
    for (int l = 0; l < lanes; l++) {
        for (double v : rtt[l]) all.push_back(v);
        worst = std::max(worst, *std::max_element(rtt[l].begin(), rtt[l].end()));
    }
    std::sort(all.begin(), all.end());
    printf("lanes %d rounds %d pollers/launch %d workgroups/launch %d: write->ack median %.1f us, 99%% %.1f, 99.99%% %.1f, worst %.1f us; kernels that gave up after 3 s: %llu; %.2f s\n",
           lanes, rounds, pollers, wgs, all[all.size() / 2], all[(size_t)(all.size() * 0.99)], all[(size_t)(all.size() * 0.9999)], worst, *gave, (T1 - T0) * 1e-6);
    return 0;
}
