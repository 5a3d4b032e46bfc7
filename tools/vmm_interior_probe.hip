// vmm_interior_probe -- stand-alone (no library code): do ROCm's copy / fill / launch paths cope with pointers INSIDE a
// virtual-memory mapping, and with virtual addresses that are unmapped and mapped again?  tools/gpu_efence.c hands out such
// pointers when it fences a request whose size is not a whole number of granules, and the GPU suite then showed wrong sums,
// "Memobj map does not have ptr", hipErrorUnknown and host heap corruption that it shows in no other mode
// (profiles/r05_efence.txt).  This probe repeats what the shim does, per iteration:
//   reserve (size rounded to the granule + one granule), create, map, set access; p = start + (mapped - size rounded to 256)
//   host -> p (hipMemcpyAsync from pageable memory), a kernel adds 1 to every word, p -> host, compare
//   hipMemsetAsync(p, 0x5a), p -> host, compare
//   synchronise, unmap, release, free the address range
// with `threads` host threads at once, each on a stream of its own.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/vmm_probe tools/vmm_interior_probe.hip && /tmp/vmm_probe <threads> <iterations> <interior 0|1> <keep 0|1|2|3|4>
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

struct Args {
    unsigned int* p;
    size_t n;
};
__global__ void k_inc(Args a) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) a.p[i] += 1u;
}

static std::atomic<long> n_api{0}, n_copy{0}, n_fill{0}, n_ok{0};
static size_t gran = 4096;

static int g_keep = 0;      // 0: unmap, release, free the address range (it comes back); 1: unmap and release, the range stays reserved; 2: nothing is given back;
                            // 3: unmap and release, the range stays reserved AND the next request of that size maps new memory into it;
                            // 4: everything is freed, but every reservation ASKS for an address never used before (a cursor that only grows)
static std::atomic<unsigned long long> g_cursor{0x200000000000ull};      // 32 TiB
static std::atomic<long> n_hint_ignored{0};
static void worker(int tid, int iters, bool interior) {
    (void)hipSetDevice(0);
    hipStream_t st;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
        n_api++;
        return;
    }
    const size_t sizes[] = {648, 2872, 8192, 96000, 118800, 131072, 348000, 663552, 1283200};
    unsigned int seed = 12345u + 977u * (unsigned)tid;
    std::vector<unsigned int> src, back;
    std::vector<std::pair<size_t, void*>> mine;      // keep 3: this thread's reserved, unmapped ranges
    for (int it = 0; it < iters; it++) {
        seed = seed * 1664525u + 1013904223u;
        size_t n = sizes[(seed >> 8) % (sizeof sizes / sizeof sizes[0])];
        if (!interior) n = (n + gran - 1) / gran * gran;
        const size_t n256 = (n + 255) & ~(size_t)255, mapped = (n256 + gran - 1) / gran * gran, reserved = mapped + gran;
        hipMemAllocationProp prop;
        memset(&prop, 0, sizeof prop);
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        void* va = nullptr;
        hipMemGenericAllocationHandle_t h;
        for (size_t i = 0; g_keep == 3 && i < mine.size(); i++)
            if (mine[i].first == reserved) {
                va = mine[i].second;
                mine.erase(mine.begin() + i);
                break;
            }
        void* hint = nullptr;
        if (g_keep == 4) hint = (void*)g_cursor.fetch_add((reserved + (2u << 20) - 1) & ~(unsigned long long)((2u << 20) - 1));
        if (!va && hipMemAddressReserve(&va, reserved, gran, hint, 0) != hipSuccess) {
            n_api++;
            continue;
        }
        if (hint && va != hint) n_hint_ignored++;
        if (hipMemCreate(&h, mapped, &prop, 0) != hipSuccess) {
            n_api++;
            (void)hipMemAddressFree(va, reserved);
            continue;
        }
        hipMemAccessDesc acc;
        memset(&acc, 0, sizeof acc);
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = 0;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        if (hipMemMap(va, mapped, 0, h, 0) != hipSuccess || hipMemSetAccess(va, mapped, &acc, 1) != hipSuccess) {
            n_api++;
            (void)hipMemRelease(h);
            (void)hipMemAddressFree(va, reserved);
            continue;
        }
        unsigned int* p = (unsigned int*)((char*)va + (mapped - n256));
        const size_t words = n / 4;
        src.assign(words, 0);
        back.assign(words, 0);
        for (size_t i = 0; i < words; i++) src[i] = (unsigned)(i * 2654435761u) ^ seed;
        bool bad_api = false, bad_copy = false, bad_fill = false;
        bad_api |= hipMemcpyAsync(p, src.data(), words * 4, hipMemcpyHostToDevice, st) != hipSuccess;
        Args a{p, words};
        hipLaunchKernelGGL(k_inc, dim3(64), dim3(256), 0, st, a);
        bad_api |= hipGetLastError() != hipSuccess;
        bad_api |= hipMemcpyAsync(back.data(), p, words * 4, hipMemcpyDeviceToHost, st) != hipSuccess;
        bad_api |= hipStreamSynchronize(st) != hipSuccess;
        for (size_t i = 0; i < words && !bad_copy; i++) bad_copy = back[i] != src[i] + 1u;
        bad_api |= hipMemsetAsync(p, 0x5a, words * 4, st) != hipSuccess;
        bad_api |= hipMemcpyAsync(back.data(), p, words * 4, hipMemcpyDeviceToHost, st) != hipSuccess;
        bad_api |= hipStreamSynchronize(st) != hipSuccess;
        for (size_t i = 0; i < words && !bad_fill; i++) bad_fill = back[i] != 0x5a5a5a5au;
        (void)hipGetLastError();
        if (bad_api) n_api++;
        if (bad_copy) n_copy++;
        if (bad_fill) n_fill++;
        if (!bad_api && !bad_copy && !bad_fill) n_ok++;
        (void)hipDeviceSynchronize();
        if (g_keep < 2) {
            (void)hipMemUnmap(va, mapped);
            (void)hipMemRelease(h);
        }
        if (g_keep < 1 || g_keep == 4) (void)hipMemAddressFree(va, reserved);
        if (g_keep == 3) mine.emplace_back(reserved, va);
    }
    (void)hipStreamDestroy(st);
}

int main(int argc, char** argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 1, iters = argc > 2 ? atoi(argv[2]) : 1000;
    const bool interior = argc > 3 ? atoi(argv[3]) != 0 : true;
    g_keep = argc > 4 ? atoi(argv[4]) : 0;
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || !gran) gran = 4096;
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back(worker, t, iters, interior);
    for (auto& t : th) t.join();
    printf("vmm probe: keep %d, %d thread(s) x %d iterations, %s pointers, granule %zu: ok %ld, api errors %ld, wrong after copy+kernel %ld, wrong after fill %ld, address hints ignored %ld\n",
           g_keep, threads, iters, interior ? "INTERIOR" : "start-of-mapping", gran, n_ok.load(), n_api.load(), n_copy.load(), n_fill.load(), n_hint_ignored.load());
    return 0;
}
