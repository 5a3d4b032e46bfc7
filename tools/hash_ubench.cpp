// Host-side micro-benchmarks for the Fiat-Shamir hash (fr_host.h): the one strictly serial host cost of a proof that is alone
// on the GPU (92 x bN hashes of 9 elements).  Probes the core (clock, mulx latency / throughput), then times single dependent
// chains and pairs of independent chains of the Montgomery product variants, then the whole 9-element hash.
//   clang++ -O3 -std=c++17 -mbmi2 -madx tools/hash_ubench.cpp -o /tmp/hash_ubench && /tmp/hash_ubench
#include <chrono>
#include <cstdio>
#include <cstring>
#include <immintrin.h>
#include "../gkr-mimc_amd/csrc/fr_host.h"
using namespace hfr;
static double now_ns() { return std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static double g_ghz = 0;

static void probe_core() {
    const long N = 400000000;
    u64 a = 1, b = 3;
    double t0 = now_ns();
    for (long i = 0; i < N; i += 8) {
        asm volatile("add %1, %0\n\tadd %1, %0\n\tadd %1, %0\n\tadd %1, %0\n\tadd %1, %0\n\tadd %1, %0\n\tadd %1, %0\n\tadd %1, %0" : "+r"(a) : "r"(b));
    }
    double t1 = now_ns();
    g_ghz = N / (t1 - t0);
    printf("core clock (dependent add chain): %.2f GHz   (%llx)\n", g_ghz, (unsigned long long)a);
    // mulx latency: lo feeds the next multiplicand
    {
        const long M = 100000000;
        u64 x = 0x9e3779b97f4a7c15ull, hi;
        t0 = now_ns();
        for (long i = 0; i < M; i += 4) {
            asm volatile("mulx %0, %0, %1\n\tmulx %0, %0, %1\n\tmulx %0, %0, %1\n\tmulx %0, %0, %1" : "+r"(x), "=r"(hi) : "d"(0x2545f4914f6cdd1dull));
        }
        t1 = now_ns();
        printf("mulx latency (lo -> operand): %.2f cycles\n", (t1 - t0) / M * g_ghz);
        t0 = now_ns();
        for (long i = 0; i < M; i += 4) {
            asm volatile("mulx %0, %1, %0\n\tmulx %0, %1, %0\n\tmulx %0, %1, %0\n\tmulx %0, %1, %0" : "+r"(x), "=r"(hi) : "d"(0x2545f4914f6cdd1dull));
        }
        t1 = now_ns();
        printf("mulx latency (hi -> operand): %.2f cycles   (%llx)\n", (t1 - t0) / M * g_ghz, (unsigned long long)(x + hi));
    }
    // mulx throughput: 6 independent products per iteration
    {
        const long M = 60000000;
        u64 l0, l1, l2, l3, l4, l5, h, src = 0x9e3779b97f4a7c15ull;
        t0 = now_ns();
        for (long i = 0; i < M; i++) {
            asm volatile("mulx %7, %0, %6\n\tmulx %7, %1, %6\n\tmulx %7, %2, %6\n\tmulx %7, %3, %6\n\tmulx %7, %4, %6\n\tmulx %7, %5, %6"
                         : "=&r"(l0), "=&r"(l1), "=&r"(l2), "=&r"(l3), "=&r"(l4), "=&r"(l5), "=&r"(h)
                         : "r"(src), "d"(0x2545f4914f6cdd1dull));
        }
        t1 = now_ns();
        printf("mulx throughput: %.2f per cycle   (%llx)\n", 6.0 * M / ((t1 - t0) * g_ghz), (unsigned long long)(l0 + l1 + l2 + l3 + l4 + l5 + h));
    }
}

// ---- product variants --------------------------------------------------------------------------------------------
// (a) the product's own: fr_host.h mul_lazy (coarsely integrated operand scanning, compiler-scheduled)
// (b) one 256-bit reduction step: T = x*y (8 limbs), m = T_lo * QINV256 mod 2^256, r = (T + m*q) >> 256
static const u64 QINV256[4] = {0xc2e1f593efffffffULL, 0x6586864b4c6911b3ULL, 0xe39a982899062391ULL, 0x73f82f1d0d8341b2ULL};

static inline void mul4x4(const u64 x[4], const u64 y[4], u64 t[8]) {
    u128 a;
    u64 c;
    // row 0
    a = (u128)x[0] * y[0]; t[0] = (u64)a; c = (u64)(a >> 64);
    a = (u128)x[1] * y[0] + c; t[1] = (u64)a; c = (u64)(a >> 64);
    a = (u128)x[2] * y[0] + c; t[2] = (u64)a; c = (u64)(a >> 64);
    a = (u128)x[3] * y[0] + c; t[3] = (u64)a; t[4] = (u64)(a >> 64);
    for (int i = 1; i < 4; i++) {
        a = (u128)x[0] * y[i] + t[i]; t[i] = (u64)a; c = (u64)(a >> 64);
        a = (u128)x[1] * y[i] + t[i + 1] + c; t[i + 1] = (u64)a; c = (u64)(a >> 64);
        a = (u128)x[2] * y[i] + t[i + 2] + c; t[i + 2] = (u64)a; c = (u64)(a >> 64);
        a = (u128)x[3] * y[i] + t[i + 3] + c; t[i + 3] = (u64)a; t[i + 4] = (u64)(a >> 64);
    }
}
// low 4 limbs of x*y
static inline void mullo4(const u64 x[4], const u64 y[4], u64 m[4]) {
    u128 a;
    u64 c;
    a = (u128)x[0] * y[0]; m[0] = (u64)a; c = (u64)(a >> 64);
    a = (u128)x[1] * y[0] + c; m[1] = (u64)a; c = (u64)(a >> 64);
    a = (u128)x[2] * y[0] + c; m[2] = (u64)a; c = (u64)(a >> 64);
    m[3] = x[3] * y[0] + c;
    a = (u128)x[0] * y[1] + m[1]; m[1] = (u64)a; c = (u64)(a >> 64);
    a = (u128)x[1] * y[1] + m[2] + c; m[2] = (u64)a; c = (u64)(a >> 64);
    m[3] += x[2] * y[1] + c;
    a = (u128)x[0] * y[2] + m[2]; m[2] = (u64)a; c = (u64)(a >> 64);
    m[3] += x[1] * y[2] + c;
    m[3] += x[0] * y[3];
}
static inline E mul_sos256(const E& x, const E& y) {
    u64 t[8], m[4], u[8];
    mul4x4(x.l, y.l, t);
    mullo4(t, QINV256, m);
    mul4x4(m, Q, u);
    // low halves cancel: t_lo + u_lo = 0 mod 2^256, carry = (t_lo != 0)
    const u64 cy = (t[0] | t[1] | t[2] | t[3]) != 0;
    E r;
    u128 a = (u128)t[4] + u[4] + cy; r.l[0] = (u64)a;
    a = (u128)t[5] + u[5] + (u64)(a >> 64); r.l[1] = (u64)a;
    a = (u128)t[6] + u[6] + (u64)(a >> 64); r.l[2] = (u64)a;
    r.l[3] = t[7] + u[7] + (u64)(a >> 64);
    return r;
}
// (c) two 128-bit reduction steps on the 8-limb product
static const u64 QINV128[2] = {0xc2e1f593efffffffULL, 0x6586864b4c6911b3ULL};
// t[0..5] += m*q with m = t[0..1] * QINV128 mod 2^128 (t[0], t[1] become zero); returns the carry out of t[5]
static inline u64 redc128_step(u64* t) {
    u128 a = (u128)t[0] * QINV128[0];
    const u64 m0 = (u64)a;
    const u64 m1 = (u64)(a >> 64) + t[0] * QINV128[1] + t[1] * QINV128[0];
    u64 u[6], c;
    a = (u128)m0 * Q[0]; u[0] = (u64)a; c = (u64)(a >> 64);
    a = (u128)m0 * Q[1] + c; u[1] = (u64)a; c = (u64)(a >> 64);
    a = (u128)m0 * Q[2] + c; u[2] = (u64)a; c = (u64)(a >> 64);
    a = (u128)m0 * Q[3] + c; u[3] = (u64)a; u[4] = (u64)(a >> 64);
    a = (u128)m1 * Q[0] + u[1]; u[1] = (u64)a; c = (u64)(a >> 64);
    a = (u128)m1 * Q[1] + u[2] + c; u[2] = (u64)a; c = (u64)(a >> 64);
    a = (u128)m1 * Q[2] + u[3] + c; u[3] = (u64)a; c = (u64)(a >> 64);
    a = (u128)m1 * Q[3] + u[4] + c; u[4] = (u64)a; u[5] = (u64)(a >> 64);
    const u64 cy = (t[0] | t[1]) != 0;          // the two low limbs cancel exactly
    a = (u128)t[2] + u[2] + cy; t[2] = (u64)a;
    a = (u128)t[3] + u[3] + (u64)(a >> 64); t[3] = (u64)a;
    a = (u128)t[4] + u[4] + (u64)(a >> 64); t[4] = (u64)a;
    a = (u128)t[5] + u[5] + (u64)(a >> 64); t[5] = (u64)a;
    return (u64)(a >> 64);
}
static inline E mul_sos128(const E& x, const E& y) {
    u64 t[8];
    mul4x4(x.l, y.l, t);
    u64 c = redc128_step(t);
    u128 a = (u128)t[6] + c; t[6] = (u64)a;
    t[7] += (u64)(a >> 64);
    redc128_step(t + 2);                          // t < 2^512 throughout: no carry out of t[7]
    E r = {{t[4], t[5], t[6], t[7]}};
    return r;
}

// (d) no-carry CIOS in assembly: mulx with the adcx / adox carry chains (generated, see tools/gen_adx.py)
static inline E mul_adx(const E& x, const E& y) {
    u64 t0, t1, t2, t3, A, s;
    asm(
        "movq 0(%[x]), %%rdx\n\t"
        "mulx %[y0], %[t0], %[t1]\n\t"
        "mulx %[y1], %%rax, %[t2]\n\t"
        "addq %%rax, %[t1]\n\t"
        "mulx %[y2], %%rax, %[t3]\n\t"
        "adcq %%rax, %[t2]\n\t"
        "mulx %[y3], %%rax, %[A]\n\t"
        "adcq %%rax, %[t3]\n\t"
        "adcq $0, %[A]\n\t"
        "movq %[t0], %%rdx\n\t"
        "imulq %[qinv], %%rdx\n\t"
        "xorl %%eax, %%eax\n\t"
        "mulx 0(%[q]), %%rax, %[s]\n\t"
        "adcx %[t0], %%rax\n\t"
        "movq %[s], %[t0]\n\t"
        "adcx %[t1], %[t0]\n\t"
        "mulx 8(%[q]), %%rax, %[t1]\n\t"
        "adox %%rax, %[t0]\n\t"
        "adcx %[t2], %[t1]\n\t"
        "mulx 16(%[q]), %%rax, %[t2]\n\t"
        "adox %%rax, %[t1]\n\t"
        "adcx %[t3], %[t2]\n\t"
        "mulx 24(%[q]), %%rax, %[t3]\n\t"
        "adox %%rax, %[t2]\n\t"
        "movl $0, %%eax\n\t"
        "adcx %%rax, %[t3]\n\t"
        "adox %[A], %[t3]\n\t"
        "movq 8(%[x]), %%rdx\n\t"
        "xorl %%eax, %%eax\n\t"
        "mulx %[y0], %%rax, %[s]\n\t"
        "adox %%rax, %[t0]\n\t"
        "adcx %[s], %[t1]\n\t"
        "mulx %[y1], %%rax, %[s]\n\t"
        "adox %%rax, %[t1]\n\t"
        "adcx %[s], %[t2]\n\t"
        "mulx %[y2], %%rax, %[s]\n\t"
        "adox %%rax, %[t2]\n\t"
        "adcx %[s], %[t3]\n\t"
        "mulx %[y3], %%rax, %[A]\n\t"
        "adox %%rax, %[t3]\n\t"
        "movl $0, %%eax\n\t"
        "adox %%rax, %[A]\n\t"
        "adcx %%rax, %[A]\n\t"
        "movq %[t0], %%rdx\n\t"
        "imulq %[qinv], %%rdx\n\t"
        "xorl %%eax, %%eax\n\t"
        "mulx 0(%[q]), %%rax, %[s]\n\t"
        "adcx %[t0], %%rax\n\t"
        "movq %[s], %[t0]\n\t"
        "adcx %[t1], %[t0]\n\t"
        "mulx 8(%[q]), %%rax, %[t1]\n\t"
        "adox %%rax, %[t0]\n\t"
        "adcx %[t2], %[t1]\n\t"
        "mulx 16(%[q]), %%rax, %[t2]\n\t"
        "adox %%rax, %[t1]\n\t"
        "adcx %[t3], %[t2]\n\t"
        "mulx 24(%[q]), %%rax, %[t3]\n\t"
        "adox %%rax, %[t2]\n\t"
        "movl $0, %%eax\n\t"
        "adcx %%rax, %[t3]\n\t"
        "adox %[A], %[t3]\n\t"
        "movq 16(%[x]), %%rdx\n\t"
        "xorl %%eax, %%eax\n\t"
        "mulx %[y0], %%rax, %[s]\n\t"
        "adox %%rax, %[t0]\n\t"
        "adcx %[s], %[t1]\n\t"
        "mulx %[y1], %%rax, %[s]\n\t"
        "adox %%rax, %[t1]\n\t"
        "adcx %[s], %[t2]\n\t"
        "mulx %[y2], %%rax, %[s]\n\t"
        "adox %%rax, %[t2]\n\t"
        "adcx %[s], %[t3]\n\t"
        "mulx %[y3], %%rax, %[A]\n\t"
        "adox %%rax, %[t3]\n\t"
        "movl $0, %%eax\n\t"
        "adox %%rax, %[A]\n\t"
        "adcx %%rax, %[A]\n\t"
        "movq %[t0], %%rdx\n\t"
        "imulq %[qinv], %%rdx\n\t"
        "xorl %%eax, %%eax\n\t"
        "mulx 0(%[q]), %%rax, %[s]\n\t"
        "adcx %[t0], %%rax\n\t"
        "movq %[s], %[t0]\n\t"
        "adcx %[t1], %[t0]\n\t"
        "mulx 8(%[q]), %%rax, %[t1]\n\t"
        "adox %%rax, %[t0]\n\t"
        "adcx %[t2], %[t1]\n\t"
        "mulx 16(%[q]), %%rax, %[t2]\n\t"
        "adox %%rax, %[t1]\n\t"
        "adcx %[t3], %[t2]\n\t"
        "mulx 24(%[q]), %%rax, %[t3]\n\t"
        "adox %%rax, %[t2]\n\t"
        "movl $0, %%eax\n\t"
        "adcx %%rax, %[t3]\n\t"
        "adox %[A], %[t3]\n\t"
        "movq 24(%[x]), %%rdx\n\t"
        "xorl %%eax, %%eax\n\t"
        "mulx %[y0], %%rax, %[s]\n\t"
        "adox %%rax, %[t0]\n\t"
        "adcx %[s], %[t1]\n\t"
        "mulx %[y1], %%rax, %[s]\n\t"
        "adox %%rax, %[t1]\n\t"
        "adcx %[s], %[t2]\n\t"
        "mulx %[y2], %%rax, %[s]\n\t"
        "adox %%rax, %[t2]\n\t"
        "adcx %[s], %[t3]\n\t"
        "mulx %[y3], %%rax, %[A]\n\t"
        "adox %%rax, %[t3]\n\t"
        "movl $0, %%eax\n\t"
        "adox %%rax, %[A]\n\t"
        "adcx %%rax, %[A]\n\t"
        "movq %[t0], %%rdx\n\t"
        "imulq %[qinv], %%rdx\n\t"
        "xorl %%eax, %%eax\n\t"
        "mulx 0(%[q]), %%rax, %[s]\n\t"
        "adcx %[t0], %%rax\n\t"
        "movq %[s], %[t0]\n\t"
        "adcx %[t1], %[t0]\n\t"
        "mulx 8(%[q]), %%rax, %[t1]\n\t"
        "adox %%rax, %[t0]\n\t"
        "adcx %[t2], %[t1]\n\t"
        "mulx 16(%[q]), %%rax, %[t2]\n\t"
        "adox %%rax, %[t1]\n\t"
        "adcx %[t3], %[t2]\n\t"
        "mulx 24(%[q]), %%rax, %[t3]\n\t"
        "adox %%rax, %[t2]\n\t"
        "movl $0, %%eax\n\t"
        "adcx %%rax, %[t3]\n\t"
        "adox %[A], %[t3]\n\t"
        : [t0] "=&r"(t0), [t1] "=&r"(t1), [t2] "=&r"(t2), [t3] "=&r"(t3), [A] "=&r"(A), [s] "=&r"(s)
        : [x] "r"(x.l), [y0] "r"(y.l[0]), [y1] "r"(y.l[1]), [y2] "r"(y.l[2]), [y3] "r"(y.l[3]), [q] "r"(Q), [qinv] "m"(QINV),
          "m"(*(const u64(*)[4])x.l)
        : "rax", "rdx", "cc");
    E r = {{t0, t1, t2, t3}};
    return r;
}


// (e) one 256-bit reduction step in assembly (generated, tools/gen_sos_adx.py): the four m_i come from T_lo * (-q^-1 mod 2^256),
// so the reduction rows do not wait for one another
static inline E mul_sos256_adx(const E& x, const E& y) {
    u64 t0, t1, t2, t3, t4, t5, t6, t7, s, m0, m1, m2, m3;
    asm(
        "movq 0(%[x]), %%rdx\n\t"
        "mulx 0(%[y]), %[t0], %[t1]\n\t"
        "mulx 8(%[y]), %%rax, %[t2]\n\t"
        "addq %%rax, %[t1]\n\t"
        "mulx 16(%[y]), %%rax, %[t3]\n\t"
        "adcq %%rax, %[t2]\n\t"
        "mulx 24(%[y]), %%rax, %[t4]\n\t"
        "adcq %%rax, %[t3]\n\t"
        "adcq $0, %[t4]\n\t"
        "movq 8(%[x]), %%rdx\n\t"
        "xorl %%eax, %%eax\n\t"
        "mulx 0(%[y]), %%rax, %[s]\n\t"
        "adox %%rax, %[t1]\n\t"
        "adcx %[s], %[t2]\n\t"
        "mulx 8(%[y]), %%rax, %[s]\n\t"
        "adox %%rax, %[t2]\n\t"
        "adcx %[s], %[t3]\n\t"
        "mulx 16(%[y]), %%rax, %[s]\n\t"
        "adox %%rax, %[t3]\n\t"
        "adcx %[s], %[t4]\n\t"
        "mulx 24(%[y]), %%rax, %[t5]\n\t"
        "adox %%rax, %[t4]\n\t"
        "movl $0, %%eax\n\t"
        "adox %%rax, %[t5]\n\t"
        "adcx %%rax, %[t5]\n\t"
        "movq 16(%[x]), %%rdx\n\t"
        "xorl %%eax, %%eax\n\t"
        "mulx 0(%[y]), %%rax, %[s]\n\t"
        "adox %%rax, %[t2]\n\t"
        "adcx %[s], %[t3]\n\t"
        "mulx 8(%[y]), %%rax, %[s]\n\t"
        "adox %%rax, %[t3]\n\t"
        "adcx %[s], %[t4]\n\t"
        "mulx 16(%[y]), %%rax, %[s]\n\t"
        "adox %%rax, %[t4]\n\t"
        "adcx %[s], %[t5]\n\t"
        "mulx 24(%[y]), %%rax, %[t6]\n\t"
        "adox %%rax, %[t5]\n\t"
        "movl $0, %%eax\n\t"
        "adox %%rax, %[t6]\n\t"
        "adcx %%rax, %[t6]\n\t"
        "movq 24(%[x]), %%rdx\n\t"
        "xorl %%eax, %%eax\n\t"
        "mulx 0(%[y]), %%rax, %[s]\n\t"
        "adox %%rax, %[t3]\n\t"
        "adcx %[s], %[t4]\n\t"
        "mulx 8(%[y]), %%rax, %[s]\n\t"
        "adox %%rax, %[t4]\n\t"
        "adcx %[s], %[t5]\n\t"
        "mulx 16(%[y]), %%rax, %[s]\n\t"
        "adox %%rax, %[t5]\n\t"
        "adcx %[s], %[t6]\n\t"
        "mulx 24(%[y]), %%rax, %[t7]\n\t"
        "adox %%rax, %[t6]\n\t"
        "movl $0, %%eax\n\t"
        "adox %%rax, %[t7]\n\t"
        "adcx %%rax, %[t7]\n\t"
        : [t0] "=&r"(t0), [t1] "=&r"(t1), [t2] "=&r"(t2), [t3] "=&r"(t3), [t4] "=&r"(t4), [t5] "=&r"(t5), [t6] "=&r"(t6), [t7] "=&r"(t7), [s] "=&r"(s)
        : [x] "r"(x.l), [y] "r"(y.l), "m"(*(const u64(*)[4])x.l), "m"(*(const u64(*)[4])y.l)
        : "rax", "rdx", "cc");
    asm(
        "movq %[t0], %%rdx\n\t"
        "movq %[t0], %[m3]\n\t"
        "imulq %[n3], %[m3]\n\t"
        "mulx %[n0], %[m0], %[m1]\n\t"
        "mulx %[n1], %%rax, %[m2]\n\t"
        "addq %%rax, %[m1]\n\t"
        "mulx %[n2], %%rax, %[s]\n\t"
        "adcq %%rax, %[m2]\n\t"
        "adcq %[s], %[m3]\n\t"
        "movq %[t1], %%rdx\n\t"
        "movq %[t1], %%rax\n\t"
        "imulq %[n2], %%rax\n\t"
        "addq %%rax, %[m3]\n\t"
        "xorl %%eax, %%eax\n\t"
        "mulx %[n0], %%rax, %[s]\n\t"
        "adox %%rax, %[m1]\n\t"
        "adcx %[s], %[m2]\n\t"
        "mulx %[n1], %%rax, %[s]\n\t"
        "adox %%rax, %[m2]\n\t"
        "adcx %[s], %[m3]\n\t"
        "movl $0, %%eax\n\t"
        "adox %%rax, %[m3]\n\t"
        "movq %[t2], %%rdx\n\t"
        "movq %[t2], %%rax\n\t"
        "imulq %[n1], %%rax\n\t"
        "addq %%rax, %[m3]\n\t"
        "mulx %[n0], %%rax, %[s]\n\t"
        "addq %%rax, %[m2]\n\t"
        "adcq %[s], %[m3]\n\t"
        "movq %[t3], %%rax\n\t"
        "imulq %[n0], %%rax\n\t"
        "addq %%rax, %[m3]\n\t"
        "movq %[m0], %%rdx\n\t"
        "xorl %%eax, %%eax\n\t"
        "mulx %[q0], %%rax, %[s]\n\t"
        "adox %%rax, %[t0]\n\t"
        "adcx %[s], %[t1]\n\t"
        "mulx %[q1], %%rax, %[s]\n\t"
        "adox %%rax, %[t1]\n\t"
        "adcx %[s], %[t2]\n\t"
        "mulx %[q2], %%rax, %[s]\n\t"
        "adox %%rax, %[t2]\n\t"
        "adcx %[s], %[t3]\n\t"
        "mulx %[q3], %%rax, %[s]\n\t"
        "adox %%rax, %[t3]\n\t"
        "adcx %[s], %[t4]\n\t"
        "movl $0, %%eax\n\t"
        "adox %%rax, %[t4]\n\t"
        "adcx %%rax, %[t5]\n\t"
        "adox %%rax, %[t5]\n\t"
        "adcx %%rax, %[t6]\n\t"
        "adox %%rax, %[t6]\n\t"
        "adcx %%rax, %[t7]\n\t"
        "adox %%rax, %[t7]\n\t"
        "movq %[m1], %%rdx\n\t"
        "xorl %%eax, %%eax\n\t"
        "mulx %[q0], %%rax, %[s]\n\t"
        "adox %%rax, %[t1]\n\t"
        "adcx %[s], %[t2]\n\t"
        "mulx %[q1], %%rax, %[s]\n\t"
        "adox %%rax, %[t2]\n\t"
        "adcx %[s], %[t3]\n\t"
        "mulx %[q2], %%rax, %[s]\n\t"
        "adox %%rax, %[t3]\n\t"
        "adcx %[s], %[t4]\n\t"
        "mulx %[q3], %%rax, %[s]\n\t"
        "adox %%rax, %[t4]\n\t"
        "adcx %[s], %[t5]\n\t"
        "movl $0, %%eax\n\t"
        "adox %%rax, %[t5]\n\t"
        "adcx %%rax, %[t6]\n\t"
        "adox %%rax, %[t6]\n\t"
        "adcx %%rax, %[t7]\n\t"
        "adox %%rax, %[t7]\n\t"
        "movq %[m2], %%rdx\n\t"
        "xorl %%eax, %%eax\n\t"
        "mulx %[q0], %%rax, %[s]\n\t"
        "adox %%rax, %[t2]\n\t"
        "adcx %[s], %[t3]\n\t"
        "mulx %[q1], %%rax, %[s]\n\t"
        "adox %%rax, %[t3]\n\t"
        "adcx %[s], %[t4]\n\t"
        "mulx %[q2], %%rax, %[s]\n\t"
        "adox %%rax, %[t4]\n\t"
        "adcx %[s], %[t5]\n\t"
        "mulx %[q3], %%rax, %[s]\n\t"
        "adox %%rax, %[t5]\n\t"
        "adcx %[s], %[t6]\n\t"
        "movl $0, %%eax\n\t"
        "adox %%rax, %[t6]\n\t"
        "adcx %%rax, %[t7]\n\t"
        "adox %%rax, %[t7]\n\t"
        "movq %[m3], %%rdx\n\t"
        "xorl %%eax, %%eax\n\t"
        "mulx %[q0], %%rax, %[s]\n\t"
        "adox %%rax, %[t3]\n\t"
        "adcx %[s], %[t4]\n\t"
        "mulx %[q1], %%rax, %[s]\n\t"
        "adox %%rax, %[t4]\n\t"
        "adcx %[s], %[t5]\n\t"
        "mulx %[q2], %%rax, %[s]\n\t"
        "adox %%rax, %[t5]\n\t"
        "adcx %[s], %[t6]\n\t"
        "mulx %[q3], %%rax, %[s]\n\t"
        "adox %%rax, %[t6]\n\t"
        "adcx %[s], %[t7]\n\t"
        "movl $0, %%eax\n\t"
        "adox %%rax, %[t7]\n\t"
        : [t0] "+&r"(t0), [t1] "+&r"(t1), [t2] "+&r"(t2), [t3] "+&r"(t3), [t4] "+&r"(t4), [t5] "+&r"(t5), [t6] "+&r"(t6), [t7] "+&r"(t7), [s] "=&r"(s),
          [m0] "=&r"(m0), [m1] "=&r"(m1), [m2] "=&r"(m2), [m3] "=&r"(m3)
        : [n0] "m"(QINV256[0]), [n1] "m"(QINV256[1]), [n2] "m"(QINV256[2]), [n3] "m"(QINV256[3]),
          [q0] "m"(Q[0]), [q1] "m"(Q[1]), [q2] "m"(Q[2]), [q3] "m"(Q[3])
        : "rax", "rdx", "cc");
    E r = {{t4, t5, t6, t7}};
    return r;
}


template <E (*F)(const E&, const E&)>
static double chain1(const char* name, const E& seed, const E& y, E* out) {
    const int N = 4000000;
    E x = seed;
    double t0 = now_ns();
    for (int i = 0; i < N; i++) x = F(x, y);
    double t1 = now_ns();
    *out = x;
    printf("  %-28s one chain : %6.2f ns = %5.1f cycles per product\n", name, (t1 - t0) / N, (t1 - t0) / N * g_ghz);
    return (t1 - t0) / N;
}
template <E (*F)(const E&, const E&)>
static void chain2(const char* name, const E& seed, const E& y) {
    const int N = 4000000;
    E x = seed, z = y;
    double t0 = now_ns();
    for (int i = 0; i < N; i++) { x = F(x, y); z = F(z, seed); }
    double t1 = now_ns();
    printf("  %-28s two chains: %6.2f ns = %5.1f cycles per PAIR      (%llx)\n", name, (t1 - t0) / N, (t1 - t0) / N * g_ghz, (unsigned long long)(x.l[0] ^ z.l[0]));
}
static inline E sqr_chain_cios(const E& x, const E&) { return mul_lazy(x, x); }
static inline E sqr_chain_sos256(const E& x, const E&) { return mul_sos256(x, x); }

// ---- hash variants -------------------------------------------------------------------------------------------------
template <E (*LONE)(const E&, const E&), E (*PAIR)(const E&, const E&)>
static inline E perm_variant(const E& x, const E& key) {
    E kc[MIMC_ROUNDS];
    for (int i = 0; i < MIMC_ROUNDS; i++) kc[i] = add(key, ARKS[i]);
    E res = x;
    for (int i = 0; i < MIMC_ROUNDS; i++) {
        const E s = add_raw(res, kc[i]);
        const E s2 = LONE(s, s);
        const E s3 = PAIR(s2, s);
        const E s4 = PAIR(s2, s2);
        res = LONE(s3, s4);
        cond_sub_q(res.l);
    }
    return res;
}
template <E (*LONE)(const E&, const E&), E (*PAIR)(const E&, const E&)>
static E hash_variant(const E* in, size_t n) {
    E state = ZERO;
    for (size_t k = 0; k < n; k++) {
        E ns = add(perm_variant<LONE, PAIR>(in[k], state), state);
        state = add(add(state, ns), in[k]);
    }
    return state;
}
template <E (*H)(const E*, size_t)>
static E time_hash(const char* name) {
    E in[9];
    for (int i = 0; i < 9; i++) in[i] = from_u64(1234567 + i * 7919);
    E acc = ZERO;
    const int N = 8000;
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
        acc = ZERO;
        double t0 = now_ns();
        for (int k = 0; k < N; k++) { in[0] = acc; acc = H(in, 9); }
        double t1 = now_ns();
        best = std::min(best, (t1 - t0) / N);
    }
    printf("  %-44s %6.2f us per 9-element hash = %5.1f cycles per MiMC round  (%016llx)\n", name, best / 1e3, best / (9 * 91) * g_ghz, (unsigned long long)acc.l[0]);
    return acc;
}
static E hash_product(const E* in, size_t n) { return mimc_hash(in, n); }

int main() {
    probe_core();
    E x = from_u64(0x123456789abcdefull), y = from_u64(0xfedcba987654321ull);
    // exactness of the variants against the product's own multiplication (canonical results compared)
    for (int i = 0; i < 100000; i++) {
        E a = mul(x, y);
        E b = mul_sos256(x, y); cond_sub_q(b.l);
        E c = mul_sos128(x, y); cond_sub_q(c.l);
        E d = mul_adx(x, y); E d2 = mul_lazy(x, y);
        E d3 = mul_sos256_adx(x, y);
        if (d3 != d2) { printf("MISMATCH sos256 adx at %d\n", i); return 1; }
        if (d != d2) { printf("MISMATCH adx at %d\n", i); return 1; }
        if (a != b || a != c) { printf("MISMATCH at %d: sos256 %d sos128 %d\n", i, a != b, a != c); return 1; }
        x = add(a, y); y = add(y, ONE);
        if (i % 3 == 0) { x = add_raw(x, x); if (x.l[3] >> 63) cond_sub_q(x.l); }   // lazy operands (< 2^255)
    }
    printf("variants agree with fr_host mul on 100000 products\n");
    E o;
    printf("Montgomery product, dependent chains:\n");
    chain1<mul_lazy>("cios (fr_host mul_lazy)", x, y, &o);
    chain2<mul_lazy>("cios (fr_host mul_lazy)", x, y);
    chain1<mul_sos256>("one 256-bit reduction step", x, y, &o);
    chain2<mul_sos256>("one 256-bit reduction step", x, y);
    chain1<mul_sos128>("two 128-bit reduction steps", x, y, &o);
    chain2<mul_sos128>("two 128-bit reduction steps", x, y);
    chain1<mul_adx>("cios, mulx/adcx/adox asm", x, y, &o);
    chain2<mul_adx>("cios, mulx/adcx/adox asm", x, y);
    chain1<mul_sos256_adx>("256-bit step, asm", x, y, &o);
    chain2<mul_sos256_adx>("256-bit step, asm", x, y);
    chain1<sqr_chain_cios>("cios, squaring chain", x, y, &o);
    chain1<sqr_chain_sos256>("256-bit step, squaring chain", x, y, &o);
    printf("9-element MiMC hash (the Fiat-Shamir step of every sumcheck round):\n");
    E h0 = time_hash<hash_product>("fr_host.h mimc_hash (the product's)");
    E h1 = time_hash<hash_variant<mul_lazy, mul_lazy>>("template, cios everywhere");
    E h2 = time_hash<hash_variant<mul_sos256, mul_lazy>>("256-bit step for the lone products");
    E h3 = time_hash<hash_variant<mul_sos256, mul_sos256>>("256-bit step everywhere");
    E h4 = time_hash<hash_variant<mul_sos128, mul_lazy>>("128-bit steps for the lone products");
    E h5 = time_hash<hash_variant<mul_sos128, mul_sos128>>("128-bit steps everywhere");
    E h6 = time_hash<hash_variant<mul_adx, mul_adx>>("asm cios everywhere");
    if (h0 != h6) { printf("HASH MISMATCH (asm)\n"); return 1; }
    E h7 = time_hash<hash_variant<mul_sos256_adx, mul_sos256_adx>>("asm 256-bit step everywhere");
    E h8 = time_hash<hash_variant<mul_sos256_adx, mul_adx>>("asm 256-bit step lone, asm cios pair");
    if (h0 != h7 || h0 != h8) { printf("HASH MISMATCH (asm sos)\n"); return 1; }
    if (h0 != h1 || h0 != h2 || h0 != h3 || h0 != h4 || h0 != h5) { printf("HASH MISMATCH\n"); return 1; }
    printf("all hash variants agree\n");
    return 0;
}
