// host_ntt.hip.h -- computeH on device tables (prover/gadget/prove.go:308-359): pass schedule of the transforms of
// ntt.hip.h, the twiddle tables (cached per size), the domain constants of gnark-crypto's fft.NewDomain(n, 1, .).
// Included by gkrhip.hip inside its anonymous namespace.
#pragma once

// 2-adic root of unity of BN254 Fr that gnark-crypto's fft.NewDomain starts from (order 2^28 = 5^((q-1)/2^28)), Montgomery form
// is computed at first use from its decimal value's limbs below (regular form)
const hfr::u64 kRoot2_28[4] = {0x9bd61b6e725b19f0ull, 0x402d111e41112ed4ull, 0x00e0a7eb8ef62abcull, 0x2a3c09f0a58a7e85ull};
const int kMaxOrderRoot = 28;

inline E host_pow(E base, unsigned long long e) {
    E r = hfr::ONE;
    while (e) {
        if (e & 1) r = hfr::mul(r, base);
        base = hfr::mul(base, base);
        e >>= 1;
    }
    return r;
}
inline E to_plain(const E& m) {       // the regular-form integer of a Montgomery element, as a limb image
    const E one = {{1, 0, 0, 0}};
    return hfr::mul(m, one);
}

struct NttDomain {
    int logn = -1;
    E gen, finer, finer_inv, card_inv;      // Domain.Generator (order n), FinerGenerator (order 2n), its inverse, 1/n
    DevTable tw;                            // omega^i, i < n/2
};
std::mutex g_ntt_mu;
std::vector<NttDomain*> g_ntt_domains;      // one per size, kept for the life of the process (n/2 elements each)

// the domain of cardinality 2^logn with depth 1 (fft.NewDomain(m, 1, .): domain.go), its twiddles built on the current lane
int ntt_domain(int logn, NttDomain** out) {
    std::lock_guard<std::mutex> lk(g_ntt_mu);
    for (NttDomain* d : g_ntt_domains)
        if (d->logn == logn) {
            *out = d;
            return 0;
        }
    if (logn < 1 || logn + 1 > kMaxOrderRoot) return fail("computeH: a domain of 2^%d points is outside the 2-adic subgroup (2 .. 2^%d)", logn, kMaxOrderRoot - 1);
    NttDomain* d = new NttDomain();
    d->logn = logn;
    E root_plain;
    memcpy(root_plain.l, kRoot2_28, 32);
    const E root = hfr::mul(root_plain, hfr::R2);
    d->finer = host_pow(root, 1ull << (kMaxOrderRoot - (logn + 1)));
    d->gen = host_pow(root, 1ull << (kMaxOrderRoot - logn));
    d->finer_inv = hfr::pow_q_minus_2(d->finer);
    d->card_inv = hfr::pow_q_minus_2(hfr::from_u64(1ull << logn));
    const size_t n_half = (size_t)1 << (logn - 1);
    const int l0 = std::min(12, logn - 1);
    const size_t nlo = (size_t)1 << l0, nhi = n_half >> l0;
    std::vector<E> lo(nlo), hi(nhi);
    lo[0] = hfr::ONE;
    for (size_t i = 1; i < nlo; i++) lo[i] = hfr::mul(lo[i - 1], d->gen);
    const E step = host_pow(d->gen, nlo);
    hi[0] = hfr::ONE;
    for (size_t i = 1; i < nhi; i++) hi[i] = hfr::mul(hi[i - 1], step);
    ScopedTable tlo, thi;
    CHK(table_alloc(&tlo, nlo));
    CHK(table_alloc(&thi, nhi));
    CHK(upload_table(&tlo, (const uint64_t*)lo.data(), nlo));
    CHK(upload_table(&thi, (const uint64_t*)hi.data(), nhi));
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, sizeof(uint4) * 2 * n_half));
    d->tw.base = (uint4*)p;
    d->tw.cap = n_half;
    hipLaunchKernelGGL(k_ntt_twiddles, dim3(grid_for(n_half, cx().max_grid)), dim3(GKR_BLOCK), 0, cx().stream, d->tw.planes(), tlo.cplanes(),
                       thi.cplanes(), l0, n_half);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(cx().stream));
    table_release(&tlo);
    table_release(&thi);
    g_ntt_domains.push_back(d);
    *out = d;
    return 0;
}
void ntt_domains_free() {
    std::lock_guard<std::mutex> lk(g_ntt_mu);
    for (NttDomain* d : g_ntt_domains) {
        if (d->tw.base) (void)hipFree(d->tw.base);
        delete d;
    }
    g_ntt_domains.clear();
}

template <bool DIT, bool TRIPLE>
int ntt_launch(const NttPassArgs& a, int R, int narr) {
    const size_t groups = (size_t)1 << (a.logn - R);
    const dim3 grid((unsigned)((groups + GKR_BLOCK - 1) / GKR_BLOCK), TRIPLE ? 1 : narr), block(GKR_BLOCK);
    switch (R) {
        case 1: hipLaunchKernelGGL((k_ntt_pass<1, DIT, TRIPLE>), grid, block, 0, cx().stream, a); break;
        case 2: hipLaunchKernelGGL((k_ntt_pass<2, DIT, TRIPLE>), grid, block, 0, cx().stream, a); break;
        case 3: hipLaunchKernelGGL((k_ntt_pass<3, DIT, TRIPLE>), grid, block, 0, cx().stream, a); break;
        default: return fail("ntt: %d stages per pass", R);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// computeH on three device tables of n = 2^logn elements (already zero-padded).  On return t[0] holds the reference's
// result -- the coefficients of H in bit-reversed order, REGULAR form --, t[1] and t[2] are scratch.  passes_out (optional)
// receives the number of passes over HBM (each reads and writes the arrays it names once).
int compute_h_dev(DevTable* const* t, int logn, int* passes_out) {
    NttDomain* dom = nullptr;
    CHK(ntt_domain(logn, &dom));
    const int kR = 3;                                 // stages per pass (eight elements per lane)
    int passes = 0;
    NttPassArgs a;
    auto base_args = [&](int narr) {
        memset(&a, 0, sizeof a);
        for (int i = 0; i < narr; i++) a.d[i] = t[i]->planes();
        a.tw = dom->tw.cplanes();
        a.logn = logn;
    };
    // 1. FFTInverse(a | b | c, DIF, 0) without its 1/n (folded into the next load)                      (:326-328)
    for (int s0 = 0; s0 < logn; s0 += kR) {
        base_args(3);
        a.s0 = s0;
        a.inverse = 1;
        CHK((ntt_launch<false, false>(a, std::min(kR, logn - s0), 3)));
        passes++;
    }
    // 2. FFT(., DIT, 1): first load multiplies position p by u^rev(p) / n, last pass does the pointwise step        (:330-347)
    const E minus_two_inv = hfr::pow_q_minus_2(hfr::sub(hfr::ZERO, hfr::from_u64(2)));
    for (int s0 = 0; s0 < logn; s0 += kR) {
        const bool first = s0 == 0, last = s0 + kR >= logn;
        base_args(3);
        a.s0 = s0;
        if (first) {
            a.pre = 2;
            a.k0 = to_dev(dom->card_inv);
            a.k1 = to_dev(hfr::mul(dom->card_inv, dom->finer));
        }
        if (last) {
            a.k2 = to_dev(minus_two_inv);
            CHK((ntt_launch<true, true>(a, std::min(kR, logn - s0), 3)));
        } else {
            CHK((ntt_launch<true, false>(a, std::min(kR, logn - s0), 3)));
        }
        passes++;
    }
    // 3. FFTInverse(a, DIF, 1): last store multiplies position p by u^-rev(p) / n and leaves Montgomery form       (:350-356)
    for (int s0 = 0; s0 < logn; s0 += kR) {
        base_args(1);
        a.s0 = s0;
        a.inverse = 1;
        if (s0 + kR >= logn) {
            a.post = 3;
            a.k0 = to_dev(to_plain(dom->card_inv));
            a.k1 = to_dev(to_plain(hfr::mul(dom->card_inv, dom->finer_inv)));
        }
        CHK((ntt_launch<false, false>(a, std::min(kR, logn - s0), 1)));
        passes++;
    }
    if (passes_out) *passes_out = passes;
    return 0;
}
