// rocprim_sort_probe.hip -- how fast is rocPRIM's radix sort of (bucket, entry) pairs on this GPU?  The question behind a
// fixed-base MSM with one bucket space of 2^(c-1) buckets for all windows (c = 20..22): its sort needs 21-bit keys and 28-bit
// point indices, which the library's own LDS-staged counting sort (16-bit digit planes, 32-bit entries) does not hold.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/rocprim_sort_probe.hip -o /tmp/rocprim_sort_probe && /tmp/rocprim_sort_probe
#include <hip/hip_runtime.h>
#include <cstring>
#include <string.h>
#include <cstdio>
#include <rocprim/rocprim.hpp>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_fill(unsigned int* k, unsigned int* v, size_t n, unsigned int mask) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long x = (i + 1) * 0x9e3779b97f4a7c15ull;
        x ^= x >> 29; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 32;
        k[i] = (unsigned int)x & mask;
        v[i] = (unsigned int)i;
    }
}
int main() {
    const struct { int lg, w, bits; } cases[] = {{20, 13, 20}, {22, 13, 20}, {22, 12, 22}, {24, 13, 20}, {24, 12, 22}, {24, 16, 16}};
    for (auto cs : cases) {
        const size_t n = ((size_t)1 << cs.lg) * cs.w;
        unsigned int *ki, *ko, *vi, *vo;
        CK(hipMalloc(&ki, n * 4)); CK(hipMalloc(&ko, n * 4)); CK(hipMalloc(&vi, n * 4)); CK(hipMalloc(&vo, n * 4));
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, ki, vi, n, (1u << (cs.bits - 1)) - 1u);
        size_t bytes = 0;
        CK(rocprim::radix_sort_pairs(nullptr, bytes, ki, ko, vi, vo, n, 0, cs.bits, (hipStream_t)0));
        void* tmp;
        CK(hipMalloc(&tmp, bytes));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(rocprim::radix_sort_pairs(tmp, bytes, ki, ko, vi, vo, n, 0, cs.bits, (hipStream_t)0));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 5; r++) CK(rocprim::radix_sort_pairs(tmp, bytes, ki, ko, vi, vo, n, 0, cs.bits, (hipStream_t)0));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("2^%d points x %d windows = %zu pairs, %d key bits: %.3f ms per sort (%.1f G pairs/s), %zu MB of temporary storage\n", cs.lg, cs.w, n,
               cs.bits, ms / 5, n / (ms / 5) * 1e-6, bytes >> 20);
        CK(hipFree(ki)); CK(hipFree(ko)); CK(hipFree(vi)); CK(hipFree(vo)); CK(hipFree(tmp));
    }
    return 0;
}
