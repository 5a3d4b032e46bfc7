// Does hipMemUnmap + hipMemRelease give the physical memory back while the address range stays reserved (ROCm 7.2.0, gfx950)?
// 4 GiB per iteration: reserve, create, map, set access, touch, unmap, release; `keep` = 1: the range stays reserved, 0: freed too.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
int main(int argc, char** argv) {
    const int keep = argc > 1 ? atoi(argv[1]) : 1, iters = argc > 2 ? atoi(argv[2]) : 200;
    const size_t sz = (size_t)4 << 30;
    (void)hipSetDevice(0);
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    hipMemAccessDesc acc;
    memset(&acc, 0, sizeof acc);
    acc.location.type = hipMemLocationTypeDevice;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int i = 0; i < iters; i++) {
        void* va = nullptr;
        hipMemGenericAllocationHandle_t h;
        hipError_t e = hipMemAddressReserve(&va, sz, 2u << 20, nullptr, 0);
        if (e == hipSuccess) e = hipMemCreate(&h, sz, &prop, 0);
        if (e != hipSuccess) {
            printf("iteration %d: %s\n", i, hipGetErrorString(e));
            return 1;
        }
        e = hipMemMap(va, sz, 0, h, 0);
        if (e == hipSuccess) e = hipMemSetAccess(va, sz, &acc, 1);
        if (e == hipSuccess) e = hipMemset(va, 1, sz);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            printf("iteration %d (map / touch): %s\n", i, hipGetErrorString(e));
            return 1;
        }
        (void)hipMemUnmap(va, sz);
        (void)hipMemRelease(h);
        if (!keep) (void)hipMemAddressFree(va, sz);
        if (i % 25 == 0) {
            size_t fr = 0, tot = 0;
            (void)hipMemGetInfo(&fr, &tot);
            printf("iteration %d: free %zu GiB of %zu\n", i, fr >> 30, tot >> 30);
        }
    }
    printf("keep %d: %d iterations of 4 GiB done\n", keep, iters);
    return 0;
}
