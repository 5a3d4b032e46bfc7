/* LD_PRELOAD shim: one line per hipMalloc / hipFree / hipHostMalloc / hipHostFree (time, thread, pointer, size) appended
 * to $ALLOC_TRACE_FILE with write(2), so that the address of a "Memory access fault by GPU" can be matched with the
 * allocation it once belonged to.  Diagnostic only (profiles/r05_anomalies.md (c)); not part of the product.
 *   gcc -O2 -shared -fPIC -o /tmp/alloc_trace.so tools/alloc_trace.c -ldl
 *   LD_PRELOAD="$LD_PRELOAD:/tmp/alloc_trace.so" ALLOC_TRACE_FILE=trace.txt python bench.py */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

static int fd = -2;
static void out(const char* op, void* p, size_t n, int rc) {
    if (fd == -2) {
        const char* f = getenv("ALLOC_TRACE_FILE");
        fd = f ? open(f, O_WRONLY | O_CREAT | O_APPEND, 0644) : -1;
    }
    if (fd < 0) return;
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    char b[160];
    int k = snprintf(b, sizeof b, "%ld.%06ld %d %ld %s %p %zu rc=%d\n", (long)ts.tv_sec, ts.tv_nsec / 1000, (int)getpid(),
                     (long)syscall(SYS_gettid), op, p, n, rc);
    if (k > 0) (void)!write(fd, b, (size_t)k);
}
typedef int (*fn_malloc)(void**, size_t);
typedef int (*fn_free)(void*);
typedef int (*fn_hostmalloc)(void**, size_t, unsigned int);
/* libamdhip64 comes in through a dlopen of the library under test (local scope): RTLD_NEXT does not see it */
static void* lookup(const char* name) {
    void* f = dlsym(RTLD_NEXT, name);
    if (!f) {
        void* h = dlopen("libamdhip64.so", RTLD_LAZY | RTLD_NOLOAD);
        if (!h) h = dlopen("libamdhip64.so.7", RTLD_LAZY | RTLD_NOLOAD);
        if (!h) h = dlopen("libamdhip64.so", RTLD_LAZY);
        if (h) f = dlsym(h, name);
    }
    if (!f) {
        static const char msg[] = "alloc_trace: cannot find the HIP runtime\n";
        (void)!write(2, msg, sizeof msg - 1);
        _exit(97);
    }
    return f;
}
#define NEXT(name, type) static type real = 0; if (!real) real = (type)lookup(name)

int hipMalloc(void** p, size_t n) {
    NEXT("hipMalloc", fn_malloc);
    int rc = real(p, n);
    out("malloc", rc == 0 ? *p : 0, n, rc);
    return rc;
}
int hipFree(void* p) {
    NEXT("hipFree", fn_free);
    out("free>", p, 0, 0);
    int rc = real(p);
    out("free<", p, 0, rc);
    return rc;
}
int hipHostMalloc(void** p, size_t n, unsigned int flags) {
    NEXT("hipHostMalloc", fn_hostmalloc);
    int rc = real(p, n, flags);
    out("hostmalloc", rc == 0 ? *p : 0, n, rc);
    return rc;
}
int hipHostFree(void* p) {
    NEXT("hipHostFree", fn_free);
    out("hostfree>", p, 0, 0);
    int rc = real(p);
    out("hostfree<", p, 0, rc);
    return rc;
}
