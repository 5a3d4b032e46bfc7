// fr_host.h -- host-side BN254-Fr arithmetic (4 x 64-bit Montgomery limbs) for the serial parts of the
// prover that stay on a CPU core: Fiat-Shamir (hash.MimcHash, reference hash/mimc.go:11-49 via
// common/challenge.go:10-12), Lagrange interpolation of the round polynomial (poly/lagrange.go:96-111),
// the reduction of the limb-split device sums, claim routing and the tiny tail rounds.
// Product code: never includes or links anything under oracle/.
#pragma once
#include <stdio.h>
#include <string>
#include <stdint.h>
#include <string.h>

#include <vector>

namespace hfr {

typedef unsigned __int128 u128;
typedef uint64_t u64;

struct E {
    u64 l[4];
    bool operator==(const E& o) const { return memcmp(l, o.l, 32) == 0; }
    bool operator!=(const E& o) const { return !(*this == o); }
};

static const u64 Q[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const u64 QINV = 0xc2e1f593efffffffULL;
static const E ONE = {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}};
static const E R2 = {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}};
static const E ZERO = {{0, 0, 0, 0}};

static const E ARKS[100] = {
#include "arks_bn254.inc"
};
static const int MIMC_ROUNDS = 91;  // hash/mimc.go:8

static inline bool geq_q(const u64 t[4]) {
    for (int i = 3; i >= 0; i--) {
        if (t[i] != Q[i]) return t[i] > Q[i];
    }
    return true;
}
static inline void sub_q(u64 t[4]) {
    u64 b = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)t[i] - Q[i] - b;
        t[i] = (u64)d;
        b = (u64)(d >> 64) & 1;
    }
}

// t >= q ? t - q : t without a data-dependent branch (the comparison outcome is a coin flip: a mispredicted branch per
// product would sit on the Fiat-Shamir critical path)
static inline void cond_sub_q(u64 t[4]) {
    u64 d[4], b = 0;
    for (int i = 0; i < 4; i++) {
        u128 x = (u128)t[i] - Q[i] - b;
        d[i] = (u64)x;
        b = (u64)(x >> 64) & 1;
    }
    const u64 keep = 0 - b;            // borrow: t < q, keep t
    for (int i = 0; i < 4; i++) t[i] = (t[i] & keep) | (d[i] & ~keep);
}
// x + y without reduction (the caller knows the sum stays below 2^256)
static inline E add_raw(const E& x, const E& y) {
    E r;
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)x.l[i] + y.l[i];
        r.l[i] = (u64)c;
        c >>= 64;
    }
    return r;
}

static inline E mul(const E& x, const E& y) {
    u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#define HFR_ROW(yi)                                                            \
    {                                                                          \
        u128 a = (u128)x.l[0] * (yi) + t0;                                     \
        u64 lo0 = (u64)a;                                                      \
        a = (u128)x.l[1] * (yi) + t1 + (u64)(a >> 64);                         \
        u64 lo1 = (u64)a;                                                      \
        a = (u128)x.l[2] * (yi) + t2 + (u64)(a >> 64);                         \
        u64 lo2 = (u64)a;                                                      \
        a = (u128)x.l[3] * (yi) + t3 + (u64)(a >> 64);                         \
        u64 lo3 = (u64)a;                                                      \
        u64 hi = (u64)(a >> 64);                                               \
        u64 m = lo0 * QINV;                                                    \
        a = (u128)m * Q[0] + lo0;                                              \
        a = (u128)m * Q[1] + lo1 + (u64)(a >> 64);                             \
        t0 = (u64)a;                                                           \
        a = (u128)m * Q[2] + lo2 + (u64)(a >> 64);                             \
        t1 = (u64)a;                                                           \
        a = (u128)m * Q[3] + lo3 + (u64)(a >> 64);                             \
        t2 = (u64)a;                                                           \
        t3 = hi + (u64)(a >> 64); /* q < 2^254: no overflow (no-carry CIOS) */ \
    }
    HFR_ROW(y.l[0])
    HFR_ROW(y.l[1])
    HFR_ROW(y.l[2])
    HFR_ROW(y.l[3])
#undef HFR_ROW
    E r = {{t0, t1, t2, t3}};
    if (geq_q(r.l)) sub_q(r.l);
    return r;
}
static inline E sqr(const E& x) { return mul(x, x); }
static inline E add(const E& x, const E& y) {
    E r;
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)x.l[i] + y.l[i];
        r.l[i] = (u64)c;
        c >>= 64;
    }
    if (geq_q(r.l)) sub_q(r.l);
    return r;
}
static inline E sub(const E& x, const E& y) {
    E r;
    u64 b = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)x.l[i] - y.l[i] - b;
        r.l[i] = (u64)d;
        b = (u64)(d >> 64) & 1;
    }
    if (b) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)r.l[i] + Q[i];
            r.l[i] = (u64)c;
            c >>= 64;
        }
    }
    return r;
}
static inline E neg(const E& x) { return sub(ZERO, x); }
static inline E from_u64(u64 v) {
    E t = {{v, 0, 0, 0}};
    return mul(t, R2);
}
static inline E pow_q_minus_2(const E& a) {  // a^-1 (0 -> 0, as gnark-crypto's Inverse)
    u64 e[4] = {Q[0] - 2, Q[1], Q[2], Q[3]};
    E res = ONE, base = a;
    for (int i = 0; i < 256; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) res = mul(res, base);
        base = sqr(base);
    }
    return res;
}
static inline bool is_canonical(const E& a) { return !geq_q(a.l); }

static inline E pow7(const E& x) {  // hash/poseidon.go:129-135
    E t = sqr(x);
    t = mul(t, x);
    t = sqr(t);
    return mul(t, x);
}

// Lazy Montgomery product: no final conditional subtraction.  For x < 2^256 - q the running value stays
// below x + q, so it fits four limbs; the result is < x*y/2^256 + q.
static inline E mul_lazy(const E& x, const E& y) {
    u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#define HFR_ROW(yi)                                        \
    {                                                      \
        u128 a = (u128)x.l[0] * (yi) + t0;                 \
        u64 lo0 = (u64)a;                                  \
        a = (u128)x.l[1] * (yi) + t1 + (u64)(a >> 64);     \
        u64 lo1 = (u64)a;                                  \
        a = (u128)x.l[2] * (yi) + t2 + (u64)(a >> 64);     \
        u64 lo2 = (u64)a;                                  \
        a = (u128)x.l[3] * (yi) + t3 + (u64)(a >> 64);     \
        u64 lo3 = (u64)a;                                  \
        u64 hi = (u64)(a >> 64);                           \
        u64 m = lo0 * QINV;                                \
        a = (u128)m * Q[0] + lo0;                          \
        a = (u128)m * Q[1] + lo1 + (u64)(a >> 64);         \
        t0 = (u64)a;                                       \
        a = (u128)m * Q[2] + lo2 + (u64)(a >> 64);         \
        t1 = (u64)a;                                       \
        a = (u128)m * Q[3] + lo3 + (u64)(a >> 64);         \
        t2 = (u64)a;                                       \
        t3 = hi + (u64)(a >> 64);                          \
    }
    HFR_ROW(y.l[0])
    HFR_ROW(y.l[1])
    HFR_ROW(y.l[2])
    HFR_ROW(y.l[3])
#undef HFR_ROW
    E r = {{t0, t1, t2, t3}};
    return r;
}

// The same lazy product (same integer: (x*y + m*q) / 2^256 with the same m) as no-carry CIOS rows in assembly -- mulx with the
// adcx / adox carry chains (generated by tools/gen_adx.py) -- for the Fiat-Shamir chain on x86-64 hosts with BMI2 and ADX:
// 49 instead of ~52 cycles per dependent product in the chain, 33.1 instead of 33.8 us per 9-element hash on the EPYC 9575F
// host (tools/hash_ubench.cpp, profiles/r03_v4_host_hash_ubench_epyc9575f.txt).  Everything else keeps the portable source.
#if defined(__x86_64__) && defined(__BMI2__) && defined(__ADX__) && !defined(GKRHIP_NO_HOST_ASM)
#define HFR_HAVE_MUL_ADX 1
static inline E mul_lazy_adx(const E& x, const E& y) {
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

#else
static inline E mul_lazy_adx(const E& x, const E& y) { return mul_lazy(x, y); }
#endif

// hash/mimc.go:31-39
static inline E mimc_keyed_permutation(const E& x, const E& key) {
    // Fiat-Shamir is the serial floor of the prover (one hash of the round polynomial per round,
    // 9 x 91 x^7 in a dependent chain), so this is written for latency: the round keys key+Ark_i do not
    // depend on the state and are hoisted; x^7 = x^3 * x^4 has multiplicative depth 3 instead of the
    // reference's sq-mul-sq-mul depth 4 (same value); the inner products are lazy (s < 2q:
    // s2 < 1.76q, s3 < 1.67q, s4 < 1.59q, s7 < 1.51q before the single conditional subtraction), the round's
    // addition is not reduced and the one conditional subtraction is branch-free (its outcome is a coin flip): 36.6 ->
    // 34.0 us per 9-element hash on the EPYC 9575F host; a separated-operand-scanning product (512-bit product first,
    // then the four Montgomery steps) was measured too and is slower (40.6 us); one 256-bit reduction step (three wide
    // product trees, 42 limb products) ties (34.0 us).
    E kc[MIMC_ROUNDS];
    for (int i = 0; i < MIMC_ROUNDS; i++) kc[i] = add(key, ARKS[i]);
    E res = x;
    for (int i = 0; i < MIMC_ROUNDS; i++) {
        const E s = add_raw(res, kc[i]);           // res, kc < q: s < 2q, no reduction needed before the lazy products
        const E s2 = mul_lazy_adx(s, s);
        const E s3 = mul_lazy_adx(s2, s);
        const E s4 = mul_lazy_adx(s2, s2);
        res = mul_lazy_adx(s3, s4);            // < 1.51q
        cond_sub_q(res.l);                         // canonical again, branch-free
    }
    return res;
}
// hash/mimc.go:11-28,43-49 : state <- state + (Perm_state(x) + state) + x
static inline E mimc_hash(const E* in, size_t n) {
    E state = ZERO;
    for (size_t k = 0; k < n; k++) {
        E ns = add(mimc_keyed_permutation(in[k], state), state);
        state = add(add(state, ns), in[k]);
    }
    return state;
}

// fr.Element.String(): the decimal of the regular form
static inline std::string to_decimal(const E& a) {
    const E one = {{1, 0, 0, 0}};
    E v = mul(a, one);                       // out of Montgomery form
    std::string out;
    for (;;) {
        // v /= 10^18, remainder rem
        const u64 D = 1000000000000000000ull;
        u128 rem = 0;
        bool zero = true;
        for (int i = 3; i >= 0; i--) {
            const u128 cur = (rem << 64) | v.l[i];
            v.l[i] = (u64)(cur / D);
            rem = cur % D;
            zero = zero && v.l[i] == 0;
        }
        char buf[32];
        snprintf(buf, sizeof buf, zero ? "%llu" : "%018llu", (unsigned long long)rem);
        out = std::string(buf) + out;
        if (zero) break;
    }
    return out;
}

// poly/lagrange.go:31-39
static inline E eval_univariate(const E* c, int n, const E& x) {
    E res = c[n - 1];
    for (int i = n - 2; i >= 0; i--) res = add(mul(res, x), c[i]);
    return res;
}

// poly/eq.go:19-32
static inline E eval_eq(const E* q, const E* h, int n) {
    E res = ONE;
    for (int i = 0; i < n; i++) {
        E nxt = mul(q[i], h[i]);
        nxt = add(add(nxt, nxt), ONE);
        nxt = sub(nxt, add(q[i], h[i]));
        res = mul(res, nxt);
    }
    return res;
}

// Monomial-basis Lagrange matrices on {0..n-1} (poly/lagrange.go:42-92), built once per n.
struct Lagrange {
    std::vector<std::vector<E>> mats;  // mats[n][i*n + j] = coefficient j of L_i
    Lagrange() : mats(13) {
        for (int n = 1; n <= 12; n++) {
            mats[n].assign((size_t)n * n, ZERO);
            for (int l = 0; l < n; l++) {
                std::vector<E> acc(n, ZERO);
                acc[0] = ONE;
                int deg = 0;
                for (int i = 0; i < n; i++) {
                    if (i == l) continue;
                    E mi = neg(from_u64((u64)i));  // multiply acc by (X - i)
                    std::vector<E> up(n, ZERO);
                    for (int j = 0; j <= deg; j++) {
                        up[j] = add(up[j], mul(acc[j], mi));
                        up[j + 1] = add(up[j + 1], acc[j]);
                    }
                    acc = up;
                    deg++;
                }
                E norm = pow_q_minus_2(eval_univariate(acc.data(), n, from_u64((u64)l)));
                for (int j = 0; j < n; j++) mats[n][(size_t)l * n + j] = mul(acc[j], norm);
            }
        }
    }
    // poly/lagrange.go:96-111
    void interpolate(E* out, const E* values, int n) const {
        for (int j = 0; j < n; j++) out[j] = ZERO;
        const std::vector<E>& m = mats[n];
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) out[j] = add(out[j], mul(m[(size_t)i * n + j], values[i]));
    }
};

// Reduce a limb-split sum (8 u64 lanes, lane j = sum over terms of 32-bit limb j) to a canonical
// element.  Valid for up to 2^31 terms (each lane < 2^63).  Field addition is exact, so the order in
// which device lanes / blocks / ranks were summed is irrelevant (reference: consumeAccumulate sums
// chunk results in arrival order, sumcheck/prover.go:236-245).
static inline E reduce_limbsplit(const u64 lanes[8]) {
    // carry-propagate into 9 x 32-bit words + overflow
    u64 w[10];
    u128 c = 0;
    for (int j = 0; j < 8; j++) {
        c += lanes[j];
        w[j] = (u64)c & 0xffffffffULL;
        c >>= 32;
    }
    w[8] = (u64)c & 0xffffffffULL;
    w[9] = (u64)(c >> 32);
    E lo = {{w[0] | (w[1] << 32), w[2] | (w[3] << 32), w[4] | (w[5] << 32), w[6] | (w[7] << 32)}};
    u64 hi = w[8] | (w[9] << 32);  // value = lo + hi * 2^256, hi < 2^63
    // lo mod q : lo < 2^256 < 6q
    while (geq_q(lo.l)) sub_q(lo.l);
    // hi * 2^256 mod q = mont_mul(hi_as_plain_integer, R2) * ... : 2^256 = R, so hi*R mod q = mul(E{hi}, R2)
    E hv = {{hi, 0, 0, 0}};
    E hr = mul(hv, R2);  // = hi * R^2 / R = hi * R mod q
    return add(lo, hr);
}

}  // namespace hfr
