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
    DevTable tw;                            // per-stage twiddle tables T_s[j] = omega^(j << s), n + logn entries (ntt.hip.h)
    DevTable coset_fwd, coset_inv;          // u^rev(p) / n (Montgomery form) and u^-rev(p) / n (REGULAR form), p < n
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
    const size_t n = (size_t)1 << logn;
    const size_t ntw = n + logn;
    HIPCHK(hipMalloc(&p, sizeof(uint4) * 2 * (ntw + 2 * n)));
    d->tw.base = (uint4*)p;
    d->tw.cap = ntw;
    d->coset_fwd.base = d->tw.base + 2 * ntw;
    d->coset_fwd.cap = n;
    d->coset_inv.base = d->coset_fwd.base + 2 * n;
    d->coset_inv.cap = n;
    // (-2)^-1: Z = X^n - 1 on the coset; the pointwise step's constant is folded into the inverse coset factors (prove.go:341-347)
    const E minus_two_inv = hfr::pow_q_minus_2(hfr::sub(hfr::ZERO, hfr::from_u64(2)));
    hipLaunchKernelGGL(k_ntt_twiddles, dim3(grid_for(ntw, cx().max_grid)), dim3(GKR_BLOCK), 0, cx().stream, d->tw.planes(), tlo.cplanes(),
                       thi.cplanes(), l0, logn);
    hipLaunchKernelGGL(k_ntt_coset_table, dim3(grid_for(n, cx().max_grid)), dim3(GKR_BLOCK), 0, cx().stream, d->coset_fwd.planes(),
                       d->tw.cplanes(), logn, 0, to_dev(d->card_inv), to_dev(hfr::mul(d->card_inv, d->finer)));
    hipLaunchKernelGGL(k_ntt_coset_table, dim3(grid_for(n, cx().max_grid)), dim3(GKR_BLOCK), 0, cx().stream, d->coset_inv.planes(),
                       d->tw.cplanes(), logn, 1, to_dev(to_plain(hfr::mul(d->card_inv, minus_two_inv))),
                       to_dev(to_plain(hfr::mul(hfr::mul(d->card_inv, minus_two_inv), d->finer_inv))));
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

// The passes of one transform of 2^logn points: the contiguous tile takes min(logn, 11) stages (the last of a DIF transform,
// the first of a DIT transform), the other stages go in strided tiles of at most seven stages each, as even as possible
// (24 -> 7 + 6 | 11).  Returned in DIF order (first stage first): {stages, lcols} per pass, the contiguous pass last.
struct NttPassPlan {
    int stages, lcols;
};
inline std::vector<NttPassPlan> ntt_plan(int logn) {
    std::vector<NttPassPlan> v;
    const int lc = std::min(logn, GKR_NTT_LTILE), rest = logn - lc;
    if (rest > 0) {
        const int np = (rest + 6) / 7;
        for (int i = 0; i < np; i++) {
            const int st = rest / np + (i < rest % np ? 1 : 0);
            v.push_back({st, GKR_NTT_LTILE - st});
        }
    }
    v.push_back({lc, 0});
    return v;
}

// computeH on three device tables of n = 2^logn elements (already zero-padded).  On return t[0] holds the reference's
// result -- the coefficients of H in bit-reversed order, REGULAR form --, t[1] and t[2] are scratch.  passes_out / bytes_out
// (optional): passes over HBM and the bytes they move (32 B read + 32 B written per element of every array a pass names).
int compute_h_dev(DevTable* const* t, int logn, int* passes_out, double* bytes_out = nullptr) {
    NttDomain* dom = nullptr;
    CHK(ntt_domain(logn, &dom));
    int passes = 0;
    double bytes = 0;
    const double arr_bytes = 32.0 * (double)((size_t)1 << logn);
    const std::vector<NttPassPlan> plan = ntt_plan(logn);
    NttPassArgs a;
    auto base_args = [&](int narr, bool inverse) {
        memset(&a, 0, sizeof a);
        for (int i = 0; i < 3; i++) a.d[i] = t[i]->planes();
        (void)narr;
        a.tw = dom->tw.cplanes();
        a.logn = logn;
        (void)inverse;
    };
    // one pass: local stages `stages` starting at transform stage s0, tile columns 2^lcols
    auto launch = [&](bool dit, int narr, int s0, const NttPassPlan& pp, int arrays_read) -> int {
        a.s0 = s0;
        a.lrows = pp.stages;
        a.lcols = pp.lcols;
        a.lgQ = pp.lcols == 0 ? 0 : (dit ? s0 : logn - s0 - pp.stages);
        const dim3 grid((unsigned)((size_t)1 << (logn - pp.stages - pp.lcols)), narr), block(GKR_NTT_WG);
        // computeH needs exactly these two: FFT(., DIT, .) forward and FFTInverse(., DIF, .)
        if (dit) hipLaunchKernelGGL((k_ntt_tile<true, false>), grid, block, 0, cx().stream, a);
        else hipLaunchKernelGGL((k_ntt_tile<false, true>), grid, block, 0, cx().stream, a);
        HIPCHK(hipGetLastError());
        passes++;
        bytes += (arrays_read + narr) * arr_bytes;
        return 0;
    };
    // FFTInverse(., DIF, coset): strided tiles over the large distances, then the contiguous tile
    auto dif_inverse = [&](int narr, bool pointwise, int post) -> int {
        int s0 = 0;
        for (size_t i = 0; i < plan.size(); i++) {
            base_args(narr, true);
            if (pointwise && i == 0) a.pre = 3;
            if (i + 1 == plan.size() && post) {
                a.post = post;
                a.coset = dom->coset_inv.cplanes();
            }
            CHK(launch(false, narr, s0, plan[i], pointwise && i == 0 ? 3 : narr));
            s0 += plan[i].stages;
        }
        return 0;
    };
    // 1. FFTInverse(a | b | c, DIF, 0) without its 1/n (folded into the next load)                      (:326-328)
    CHK(dif_inverse(3, false, 0));
    // 2. FFT(., DIT, 1): the first load multiplies position p by u^rev(p) / n                            (:330-332)
    {
        int s0 = 0;
        for (size_t i = plan.size(); i-- > 0;) {          // DIT: the contiguous tile first, then the strided ones, small distances first
            base_args(3, false);
            if (i + 1 == plan.size()) {
                a.pre = 2;
                a.coset = dom->coset_fwd.cplanes();
            }
            CHK(launch(true, 3, s0, plan[i], 3));
            s0 += plan[i].stages;
        }
    }
    // 3. a * b - c when the last transform loads (:334-347); FFTInverse(a, DIF, 1): the last store multiplies
    //    position p by (-2)^-1 u^-rev(p) / n and leaves Montgomery form                                    (:350-356)
    CHK(dif_inverse(1, true, 3));
    if (passes_out) *passes_out = passes;
    if (bytes_out) *bytes_out = bytes;
    return 0;
}
