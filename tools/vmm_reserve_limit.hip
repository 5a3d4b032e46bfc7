// How much virtual address space hipMemAddressReserve hands out before it fails (ROCm 7.2.0, gfx950): one call of `chunk` GiB
// after the other, nothing mapped.  And: does ONE huge reservation work?  hipcc -O2 -o /tmp/vmm_limit tools/vmm_reserve_limit.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
int main(int argc, char** argv) {
    const size_t chunk = (size_t)(argc > 1 ? atoi(argv[1]) : 64) << 30;
    const int maxn = argc > 2 ? atoi(argv[2]) : 4096;
    (void)hipSetDevice(0);
    size_t total = 0;
    int n = 0;
    for (; n < maxn; n++) {
        void* va = nullptr;
        hipError_t e = hipMemAddressReserve(&va, chunk, 2u << 20, nullptr, 0);
        if (e != hipSuccess) {
            printf("reservation %d of %zu GiB failed: %s\n", n + 1, chunk >> 30, hipGetErrorString(e));
            break;
        }
        total += chunk;
    }
    printf("reserved %zu GiB in %d ranges of %zu GiB\n", total >> 30, n, chunk >> 30);
    return 0;
}
