// The thread-independent error query of the C ABI (include/gkrhip.h): a failure on thread A is read back, by its code,
// on thread B -- what a Go caller needs when the goroutine has moved to another OS thread between the cgo call and
// must() -- while gkrhip_last_error() on thread B (no failure of its own) is empty.  Needs no GPU: the failing calls
// are argument checks of host-only entry points.
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <thread>

#include "../../include/gkrhip.h"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) {
        fprintf(stderr, "dlopen: %s\n", dlerror());
        return 2;
    }
    auto verify = (decltype(&gkrhip_sumcheck_verify))dlsym(h, "gkrhip_sumcheck_verify");
    auto last = (decltype(&gkrhip_last_error))dlsym(h, "gkrhip_last_error");
    auto last_r = (decltype(&gkrhip_last_error_r))dlsym(h, "gkrhip_last_error_r");
    if (!verify || !last || !last_r) return 2;
    int code_a = 0, code_b = 0;
    uint64_t z[4] = {0, 0, 0, 0};
    std::thread a([&] {
        code_a = verify(nullptr, 0, z, 1, 3, nullptr, nullptr, nullptr);             // no claim
        code_b = verify(z, 1, z, 1, 0, nullptr, nullptr, nullptr);                   // bad proof shape
    });
    a.join();
    if (code_a > -16 || code_b > -16 || code_a == code_b) {
        printf("FAIL: failure codes %d %d\n", code_a, code_b);
        return 1;
    }
    int rc = 0;
    std::thread b([&] {
        char buf[256];
        if (strlen(last()) != 0) rc = 1;                                               // thread B has not failed
        const size_t n = last_r(code_a, buf, sizeof buf);
        if (n == 0 || !strstr(buf, "no claim")) rc = 1;
        last_r(code_b, buf, sizeof buf);
        if (!strstr(buf, "bad proof shape")) rc = 1;
        char tiny[8];
        if (last_r(code_a, tiny, sizeof tiny) != n || strlen(tiny) != 7) rc = 1;       // truncated, NUL-terminated, full length returned
    });
    b.join();
    printf(rc == 0 ? "ABI-ERRORS-OK\n" : "FAIL: messages\n");
    return rc;
}
