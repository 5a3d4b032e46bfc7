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
// S stages in passes of at most three, as even as possible (13 -> 3 3 3 2 2)
inline std::vector<int> ntt_split(int S) {
    std::vector<int> v;
    if (S <= 0) return v;
    const int np = (S + 2) / 3;
    for (int i = 0; i < np; i++) v.push_back(S / np + (i < S % np ? 1 : 0));
    return v;
}

// computeH on three device tables of n = 2^logn elements (already zero-padded).  On return t[0] holds the reference's
// result -- the coefficients of H in bit-reversed order, REGULAR form --, t[1] and t[2] are scratch.  passes_out / bytes_out
// (optional): passes over HBM and the bytes they move (32 B read + 32 B written per element of every array a pass names).
int compute_h_dev(DevTable* const* t, int logn, int* passes_out, double* bytes_out = nullptr) {
    NttDomain* dom = nullptr;
    CHK(ntt_domain(logn, &dom));
    // transforms of more than 2^ltile points: the ltile stages that stay inside 2^ltile consecutive elements run in the
    // LDS-tiled kernel (one pass), the others in register passes of up to three stages
    const int L = logn > GKR_NTT_LTILE ? GKR_NTT_LTILE : 0;
    NttDomain* tile_dom = nullptr;
    if (L) CHK(ntt_domain(L, &tile_dom));
    int passes = 0;
    double bytes = 0;
    const double arr_bytes = 32.0 * (double)((size_t)1 << logn);
    E root_plain;
    memcpy(root_plain.l, kRoot2_28, 32);
    const E zeta = host_pow(hfr::mul(root_plain, hfr::R2), 1ull << (kMaxOrderRoot - 3));     // primitive 8th root of unity
    const E zeta_inv = hfr::pow_q_minus_2(zeta);
    NttPassArgs a;
    auto base_args = [&](int narr, bool inverse) {
        memset(&a, 0, sizeof a);
        for (int i = 0; i < narr; i++) a.d[i] = t[i]->planes();
        a.tw = dom->tw.cplanes();
        if (L) a.tw_tile = tile_dom->tw.cplanes();
        a.logn = logn;
        a.ltile = L;
        a.inverse = inverse ? 1 : 0;
        E z = hfr::ONE;
        for (int k = 0; k < 4; k++) {
            a.z[k] = to_dev(z);
            z = hfr::mul(z, inverse ? zeta_inv : zeta);
        }
    };
    auto tile = [&](bool dit, int narr) -> int {
        const dim3 grid((unsigned)((size_t)1 << (logn - L)), narr), block(GKR_BLOCK);
        if (dit) hipLaunchKernelGGL(k_ntt_tile<true>, grid, block, 0, cx().stream, a);
        else hipLaunchKernelGGL(k_ntt_tile<false>, grid, block, 0, cx().stream, a);
        HIPCHK(hipGetLastError());
        passes++;
        bytes += 2 * narr * arr_bytes;
        return 0;
    };
    const std::vector<int> split = ntt_split(logn - L);      // the register passes of one transform
    // FFTInverse(., DIF, coset): register passes over the large distances, then the tile; `post` rides on the last store
    auto dif_inverse = [&](int narr, int post, const E& k0, const E& k1) -> int {
        int s0 = 0;
        for (size_t i = 0; i < split.size(); i++) {
            base_args(narr, true);
            a.s0 = s0;
            if (!L && i + 1 == split.size()) {
                a.post = post;
                a.k0 = to_dev(k0);
                a.k1 = to_dev(k1);
            }
            CHK((ntt_launch<false, false>(a, split[i], narr)));
            passes++;
            bytes += 2 * narr * arr_bytes;
            s0 += split[i];
        }
        if (L) {
            base_args(narr, true);
            a.post = post;
            a.k0 = to_dev(k0);
            a.k1 = to_dev(k1);
            CHK(tile(false, narr));
        }
        return 0;
    };
    // 1. FFTInverse(a | b | c, DIF, 0) without its 1/n (folded into the next load)                      (:326-328)
    CHK(dif_inverse(3, 0, hfr::ZERO, hfr::ZERO));
    // 2. FFT(., DIT, 1): first load multiplies position p by u^rev(p) / n, last pass does the pointwise step        (:330-347)
    const E minus_two_inv = hfr::pow_q_minus_2(hfr::sub(hfr::ZERO, hfr::from_u64(2)));
    const E pre_k0 = dom->card_inv, pre_k1 = hfr::mul(dom->card_inv, dom->finer);
    if (L) {
        base_args(3, false);
        a.pre = 2;
        a.k0 = to_dev(pre_k0);
        a.k1 = to_dev(pre_k1);
        CHK(tile(true, 3));
    }
    {
        int s0 = L;
        for (size_t i = 0; i < split.size(); i++) {
            const bool first = !L && i == 0, last = i + 1 == split.size();
            base_args(3, false);
            a.s0 = s0;
            if (first) {
                a.pre = 2;
                a.k0 = to_dev(pre_k0);
                a.k1 = to_dev(pre_k1);
            }
            if (last) {
                a.k2 = to_dev(minus_two_inv);
                CHK((ntt_launch<true, true>(a, split[i], 3)));
                bytes += 4 * arr_bytes;              // reads three arrays, writes one
            } else {
                CHK((ntt_launch<true, false>(a, split[i], 3)));
                bytes += 6 * arr_bytes;
            }
            passes++;
            s0 += split[i];
        }
    }
    // 3. FFTInverse(a, DIF, 1): last store multiplies position p by u^-rev(p) / n and leaves Montgomery form       (:350-356)
    CHK(dif_inverse(1, 3, to_plain(dom->card_inv), to_plain(hfr::mul(dom->card_inv, dom->finer_inv))));
    if (passes_out) *passes_out = passes;
    if (bytes_out) *bytes_out = bytes;
    return 0;
}
