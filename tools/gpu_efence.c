/* gpu_efence -- LD_PRELOAD shim that turns reads and writes past the END of a device buffer into a deterministic
 * "Memory access fault by GPU" (an electric fence for hipMalloc).  Diagnostic only; not part of the product.
 *
 * Why: round 5's one-in-twelve abort of the default bench line was a kernel reading 64 KiB behind a 64 KiB table
 * (k_ntt_twiddles' padding entry, profiles/r05_anomalies.md (c)).  Behind a hipMalloc'ed buffer there is nearly always more
 * mapped memory, so such a read is silent; it faults only when the buffer happens to end a mapped range.  Under this shim EVERY
 * buffer ends a mapped range: hipMalloc(n) reserves virtual addresses for the rounded-up size PLUS one unmapped granule, maps
 * physical memory over the first part only, and returns a pointer placed so that the buffer's last byte (rounded up to 256 B,
 * hipMalloc's alignment) is the last mapped byte.  hipFree waits for the device, unmaps and releases.
 *
 *   gcc -O2 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o /tmp/gpu_efence.so tools/gpu_efence.c -ldl -lpthread
 *   LD_PRELOAD="$LD_PRELOAD:/tmp/gpu_efence.so" python -m pytest tests -m gpu ...
 * Every request is fenced; one whose size (rounded to 256 B) is not a whole number of granules (4 KiB on gfx950) gets a pointer
 * INSIDE its mapping.  GPU_EFENCE_ALIGNED=1 fences only the granule-sized requests (pointer = start of the mapping).
 * Address ranges are NEVER given back (hipFree unmaps and releases the memory, the reservation stays): on ROCm 7.2.0 a range that
 * is freed and handed out again by the next hipMemAddressReserve reads and writes the wrong memory (tools/vmm_interior_probe.hip:
 * 103 of 600 iterations right when ranges are recycled, 600 of 600 when they are kept) -- the first version of this shim freed
 * them and produced wrong sums, "Memobj map does not have ptr" and host heap corruption in the program under test.
 * The price: while a range stays reserved the runtime does not give the unmapped, released memory back either
 * (tools/vmm_release_probe.hip: 4 GiB less free memory per iteration), so a process under the shim can only allocate 288 GB in
 * TOTAL -- enough for the GPU suite (4 441 allocations), not for bench.py's default line.  GPU_EFENCE_FREE_RANGES=1 frees the
 * ranges again (memory comes back, recycled ranges may misbehave): what the bench-sized run of profiles/r05_efence.txt used.
 * GPU_EFENCE_UNDER=1: the fence goes IN FRONT of the buffer instead (one unmapped granule, then the mapping, the pointer at its
 * start): accesses before the first byte fault; accesses behind the last byte are then only caught from the next granule on.
 * GPU_EFENCE_MIN (bytes, default 0): smaller requests go to the real hipMalloc.  GPU_EFENCE_LOG=1: one line per call on stderr.
 * GPU_EFENCE_POISON=<byte, e.g. 0xff>: every buffer, fenced or not, is filled with that byte before it is handed out, so a
 * kernel that reads memory nobody wrote (and got away with it because fresh device memory is zero) computes with garbage
 * and its test fails.  GPU_EFENCE_MIN=-1 with a poison byte: poisoning only, no fences. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

typedef hipError_t (*fn_malloc)(void**, size_t);
typedef hipError_t (*fn_free)(void*);

static void* rt(const char* name) {
    static void* h = 0;
    void* f = dlsym(RTLD_NEXT, name);
    if (!f) {
        if (!h) h = dlopen("libamdhip64.so", RTLD_LAZY | RTLD_NOLOAD);
        if (!h) h = dlopen("libamdhip64.so.7", RTLD_LAZY | RTLD_NOLOAD);
        if (!h) h = dlopen("libamdhip64.so", RTLD_LAZY);
        if (h) f = dlsym(h, name);
    }
    if (!f) {
        fprintf(stderr, "gpu_efence: %s not found in the HIP runtime\n", name);
        _exit(97);
    }
    return f;
}
#define RT(name) ((__typeof__(&name))rt(#name))

struct rec {
    void* user;      /* what hipMalloc returned */
    void* va;        /* start of the reservation */
    size_t mapped, reserved;
    hipMemGenericAllocationHandle_t h;
};
static struct rec* recs = 0;
static size_t nrecs = 0, caprecs = 0;
static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_mutex_t vmm = PTHREAD_MUTEX_INITIALIZER;      /* the virtual-memory calls, one at a time (GPU_EFENCE_PARALLEL=1: not) */
static int serial = 1, interior = 0, poison = -1, free_ranges = 0, under = 0;
static hipError_t fill(void* p, size_t n) {
    if (poison < 0) return hipSuccess;
    hipError_t e = RT(hipMemset)(p, poison, n);
    if (e == hipSuccess) e = RT(hipDeviceSynchronize)();
    return e;
}
static size_t gran = 0, min_bytes = 0;
static int logging = 0, n_fenced = 0, n_plain = 0;

static void setup(void) {
    if (gran) return;
    const char* e = getenv("GPU_EFENCE_MIN");
    if (e) min_bytes = strtoull(e, 0, 0);      /* -1: nothing is fenced */
    logging = getenv("GPU_EFENCE_LOG") != 0;
    serial = getenv("GPU_EFENCE_PARALLEL") == 0;
    interior = getenv("GPU_EFENCE_ALIGNED") == 0;
    free_ranges = getenv("GPU_EFENCE_FREE_RANGES") != 0;
    under = getenv("GPU_EFENCE_UNDER") != 0;
    if (getenv("GPU_EFENCE_POISON")) poison = (int)(strtoul(getenv("GPU_EFENCE_POISON"), 0, 0) & 255);
    int dev = 0;
    (void)RT(hipGetDevice)(&dev);
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t g = 0;
    if (RT(hipMemGetAllocationGranularity)(&g, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || g == 0) g = 2u << 20;
    gran = g;
    fprintf(stderr, "gpu_efence: active, granule %zu bytes, fencing %s requests >= %zu bytes, poison %d%s\n", gran, interior ? "all" : "granule-sized", min_bytes, poison, under ? ", fence IN FRONT of the buffers" : "");
}

hipError_t hipMalloc(void** out, size_t n) {
    pthread_mutex_lock(&mu);
    setup();
    pthread_mutex_unlock(&mu);
    if (n == 0 || n < min_bytes || (!interior && ((n + 255) & ~(size_t)255) % gran != 0)) {
        __sync_fetch_and_add(&n_plain, 1);
        hipError_t e0 = ((fn_malloc)rt("hipMalloc"))(out, n);
        if (e0 == hipSuccess && n) (void)fill(*out, n);
        return e0;
    }
    int dev = 0;
    (void)RT(hipGetDevice)(&dev);
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    const size_t n256 = (n + 255) & ~(size_t)255;
    const size_t mapped = (n256 + gran - 1) / gran * gran, reserved = mapped + gran;
    void* va = 0;
    hipMemGenericAllocationHandle_t h;
    if (serial) pthread_mutex_lock(&vmm);
    hipError_t e = RT(hipMemAddressReserve)(&va, reserved, gran, 0, 0);
    if (e != hipSuccess) {
        if (serial) pthread_mutex_unlock(&vmm);
        return hipErrorOutOfMemory;
    }
    e = RT(hipMemCreate)(&h, mapped, &prop, 0);
    if (e != hipSuccess) {
        (void)RT(hipMemAddressFree)(va, reserved);
        (void)RT(hipGetLastError)();
        if (serial) pthread_mutex_unlock(&vmm);
        return hipErrorOutOfMemory;
    }
    char* const at = (char*)va + (under ? gran : 0);      /* where the mapping starts inside the reservation */
    e = RT(hipMemMap)(at, mapped, 0, h, 0);
    if (e == hipSuccess) {
        hipMemAccessDesc acc;
        memset(&acc, 0, sizeof acc);
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = dev;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = RT(hipMemSetAccess)(at, mapped, &acc, 1);
        if (e != hipSuccess) (void)RT(hipMemUnmap)(at, mapped);
    }
    if (e != hipSuccess) {
        (void)RT(hipMemRelease)(h);
        (void)RT(hipMemAddressFree)(va, reserved);
        (void)RT(hipGetLastError)();
        if (serial) pthread_mutex_unlock(&vmm);
        return hipErrorOutOfMemory;
    }
    if (serial) pthread_mutex_unlock(&vmm);
    void* user = under ? (void*)at : (void*)((char*)va + (mapped - n256));
    pthread_mutex_lock(&mu);
    if (nrecs == caprecs) {
        caprecs = caprecs ? 2 * caprecs : 1024;
        recs = (struct rec*)realloc(recs, caprecs * sizeof *recs);
    }
    recs[nrecs].user = user;
    recs[nrecs].va = va;
    recs[nrecs].mapped = mapped;
    recs[nrecs].reserved = reserved;
    recs[nrecs].h = h;
    nrecs++;
    n_fenced++;
    pthread_mutex_unlock(&mu);
    if (logging) fprintf(stderr, "gpu_efence: malloc %zu -> %p (mapped %p + %zu)\n", n, user, va, mapped);
    (void)fill(user, n);
    *out = user;
    return hipSuccess;
}

hipError_t hipFree(void* p) {
    if (!p) return hipSuccess;
    struct rec r;
    int found = 0;
    pthread_mutex_lock(&mu);
    for (size_t i = nrecs; i-- > 0;)
        if (recs[i].user == p) {
            r = recs[i];
            recs[i] = recs[--nrecs];
            found = 1;
            break;
        }
    pthread_mutex_unlock(&mu);
    if (!found) return ((fn_free)rt("hipFree"))(p);
    if (logging) fprintf(stderr, "gpu_efence: free %p ...\n", p);
    (void)RT(hipDeviceSynchronize)();      /* hipFree's implicit wait for every stream */
    if (serial) pthread_mutex_lock(&vmm);
    hipError_t e = RT(hipMemUnmap)((char*)r.va + (under ? gran : 0), r.mapped);
    (void)RT(hipMemRelease)(r.h);      /* the address range stays reserved: see the header */
    if (free_ranges) (void)RT(hipMemAddressFree)(r.va, r.reserved);
    if (serial) pthread_mutex_unlock(&vmm);
    if (logging) fprintf(stderr, "gpu_efence: ... freed %p\n", p);
    return e;
}

__attribute__((destructor)) static void report(void) {
    if (gran) fprintf(stderr, "gpu_efence: %d fenced allocations, %d passed through, %zu still live at exit\n", n_fenced, n_plain, nrecs);
}
