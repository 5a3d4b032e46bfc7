#include <chrono>
#include <cstdio>
#include "../gkr-mimc_amd/csrc/fr_host.h"
int main() {
    hfr::E in[9];
    for (int i = 0; i < 9; i++) in[i] = hfr::from_u64(1234567 + i * 7919);
    hfr::E acc = hfr::ZERO;
    const int N = 20000;
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < N; k++) { in[0] = acc; acc = hfr::mimc_hash(in, 9); }
    auto t1 = std::chrono::steady_clock::now();
    printf("%.2f us per 9-element hash  (%016llx)\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / N, (unsigned long long)acc.l[0]);
}
