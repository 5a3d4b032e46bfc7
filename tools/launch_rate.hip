// How many kernel launches per second one process gets out of the HIP runtime, from 1 .. 24 host threads with a stream each
// (the shape of many small proofs in flight: ~2 000 small launches per bN = 20 proof and lane).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_rate tools/launch_rate.hip -pthread && /tmp/launch_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <thread>
#include <vector>

__global__ void k_tiny(unsigned int* p) {
    if (threadIdx.x == 0 && p) atomicAdd(p, 1u);
}

int main() {
    unsigned int* d = nullptr;
    hipMalloc(&d, 4);
    hipMemset(d, 0, 4);
    const int per_thread = 20000;
    for (int T : {1, 2, 4, 8, 16, 24}) {
        std::vector<hipStream_t> st(T);
        for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        for (auto& s : st) {      // warm
            hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, d);
            hipStreamSynchronize(s);
        }
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                for (int i = 0; i < per_thread; i++) {
                    hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st[t], d);
                    if ((i & 7) == 7) hipStreamSynchronize(st[t]);      // a proof waits for its round before it queues the next
                }
                hipStreamSynchronize(st[t]);
            });
        for (auto& x : th) x.join();
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%2d host threads / streams: %.0f launches/s in all (%.1f us per launch and thread)\n", T, T * per_thread / s, 1e6 * s / per_thread);
        for (auto& x : st) hipStreamDestroy(x);
    }
    return 0;
}
