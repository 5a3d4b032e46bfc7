// host_msm.hip.h -- G1 multi-scalar multiplication on device-resident bases: window choice, work buffers, the kernel
// sequence of g1.hip.h and the scalar tail on the host (Horner over the window sums, conversion to affine).
// Reference: gnark-crypto's (*G1Jac).MultiExp as called at prover/gadget/prove.go:76,91,189,202,221 (the bases there are
// proving-key vectors, fixed across proofs: here they are uploaded once and stay in HBM, like the assignment of a session).
// Included by gkrhip.hip inside its anonymous namespace.
#pragma once

const size_t kMsmMaxPoints = (size_t)1 << 26;      // entries are 31-bit point indices + a sign bit; W * n stays below 2^32

// number of windows for c-bit signed digits of a scalar below 2^254 (the top window absorbs the last carry: g1.hip.h)
inline int msm_windows(int c) { return (255 + c - 1) / c; }
// Window size: W * n mixed additions (8 M + 2 S) against W * 2^(c-1) buckets that each cost two full additions (12 M + 2 S)
// plus their share of the chunk offsets in the reduction -- about 3.5 mixed additions per bucket.
inline int msm_pick_c(size_t n) {
    int best = 4;
    double best_cost = 1e300;
    for (int c = 4; c <= 16; c++) {
        const double cost = (double)msm_windows(c) * ((double)n + 3.5 * (double)((size_t)1 << (c - 1)));
        if (cost < best_cost) best_cost = cost, best = c;
    }
    return best;
}

struct MsmWork {
    size_t n_cap = 0;
    int c = 0, W = 0;
    unsigned int nb = 0;
    int chunk = 0;
    unsigned int big_cap = 0;
    unsigned int* counts = nullptr;      // count | offset | order (W * nb words each) | chunk histograms (W * nchunk * nb)
    unsigned int nchunk = 0, ntiles = 0;
    size_t chunk_len = 0;
    uint32_t bias[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned int* entries = nullptr;
    unsigned int* big = nullptr;
    uint4* scalars = nullptr;            // n_cap x 32 B staging of host scalars
    uint4* xyzz = nullptr;               // buckets | parts | wins planes
    uint4* h_wins = nullptr;             // pinned: 8 planes x W
    size_t nparts = 0;
    void release() {
        if (counts) (void)hipFree(counts);
        if (entries) (void)hipFree(entries);
        if (big) (void)hipFree(big);
        if (scalars) (void)hipFree(scalars);
        if (xyzz) (void)hipFree(xyzz);
        if (h_wins) (void)hipHostFree(h_wins);
        *this = MsmWork();
    }
};
}  // namespace (the handle type is part of the C ABI)
struct gkrhip_g1_bases {
    uint4* d_points = nullptr;     // n x 64 B
    size_t n = 0;
    std::mutex mu;                 // one MSM at a time per handle (they share the work buffers)
    int c_forced = 0;              // gkrhip_msm_g1_set_window
    MsmWork w;
};
namespace {

int msm_work_prepare(MsmWork* w, size_t n, int c_forced) {
    const int c = c_forced > 0 ? c_forced : msm_pick_c(n);
    if (c < 2 || c > 16) return fail("msm: window size %d outside 2..16", c);
    if (w->counts && w->c == c && w->n_cap >= n) return 0;
    w->release();
    w->c = c;
    w->W = msm_windows(c);
    w->nb = 1u << (c - 1);
    w->n_cap = n;
    // chunk of the window reduction: enough lanes to cover the GPU once (W * nb / chunk >= ~16 K) without making the
    // chunk offset's double-and-add (~2 c group operations) the bulk of a lane's work
    // (measured at c = 16: chunks of 32 / 16 / 8 / 4 buckets 0.73 / 0.54 / 0.52 / 0.71 ms for the whole reduction)
    int chunk = 8;
    while (chunk > 4 && (size_t)w->W * (w->nb / chunk) < 16384) chunk >>= 1;
    if ((unsigned)chunk > w->nb) chunk = (int)w->nb;
    w->chunk = chunk;
    w->nparts = (size_t)w->W * (w->nb / chunk);
    const size_t nbk = (size_t)w->W * w->nb;
    w->big_cap = (unsigned int)std::min<size_t>((size_t)w->W * n / 128 + 16, (size_t)1 << 25);
    // sorting workgroups: one per window and chunk of the scalars; a chunk is long enough to amortise the workgroup's
    // histogram traffic (2^(c-1) words in and out) and short enough that W * nchunk workgroups cover the CUs several times
    // (one workgroup per CU at c = 16: measured 2.72 / 1.96 / 1.96 ms of sorting at 2^22 points with 8 / 16..32 / 64 chunks,
    // 10.6 / 8.9 / 6.9 ms at 2^24 with 8 / 16 / 64)
    w->nchunk = (unsigned int)std::min<size_t>(128, std::max<size_t>(1, n / 131072));
    w->chunk_len = (n + w->nchunk - 1) / w->nchunk;
    memset(w->bias, 0, sizeof w->bias);
    for (int j = 0; j + 1 < w->W; j++) {
        const int bit = j * c + c - 1;
        w->bias[bit >> 5] |= 1u << (bit & 31);
    }
    w->ntiles = (unsigned int)((nbk + MSM_SCAN_TILE - 1) / MSM_SCAN_TILE);
    HIPCHK(hipMalloc((void**)&w->counts, ((3 + (size_t)w->nchunk) * nbk + w->ntiles) * sizeof(unsigned int)));
    HIPCHK(hipMalloc((void**)&w->entries, std::max<size_t>(1, (size_t)w->W * n) * sizeof(unsigned int)));
    HIPCHK(hipMalloc((void**)&w->big, ((size_t)w->big_cap + 2) * sizeof(unsigned int)));
    HIPCHK(hipMalloc((void**)&w->scalars, std::max<size_t>(1, n) * 32));
    HIPCHK(hipMalloc((void**)&w->xyzz, 8 * (nbk + w->nparts + (size_t)w->W) * sizeof(uint4)));
    HIPCHK(hipHostMalloc((void**)&w->h_wins, (8 * (size_t)w->W + 1) * sizeof(uint4)));      // + the error word
    return 0;
}

inline hfp::E fp_from_words(const uint4& a, const uint4& b) {
    hfp::E e = {{(hfp::u64)a.x | ((hfp::u64)a.y << 32), (hfp::u64)a.z | ((hfp::u64)a.w << 32), (hfp::u64)b.x | ((hfp::u64)b.y << 32),
                 (hfp::u64)b.z | ((hfp::u64)b.w << 32)}};
    return hfp::canon(e);
}

struct MsmTimes {      // HIP-event split of one MSM (bench only)
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool on = false;
};

// The device part: the W window sums of sum_{i<n} [s_i] P_i land in w->h_wins (the caller synchronises the stream).
// d_scalars: n x 32 B on the device.
int msm_dev(gkrhip_g1_bases* b, const uint4* d_scalars, size_t n, int flags, MsmTimes* tm) {
    MsmWork* w = &b->w;
    hipStream_t st = cx().stream;
    const size_t nbk = (size_t)w->W * w->nb;
    MsmArgs a;
    memset(&a, 0, sizeof a);
    a.scalars = d_scalars;
    a.points = b->d_points;
    a.n = n;
    a.c = w->c;
    a.W = w->W;
    a.nb = w->nb;
    a.scalars_mont = (flags & GKRHIP_MSM_SCALARS_MONT) ? 1 : 0;
    a.count = w->counts;
    a.offset = w->counts + nbk;
    a.order = w->counts + 2 * nbk;
    a.chist = w->counts + 3 * nbk;
    a.tile_sum = a.chist + (size_t)w->nchunk * nbk;
    a.ntiles = w->ntiles;
    a.nchunk = w->nchunk;
    a.chunk_len = w->chunk_len;
    memcpy(a.bias, w->bias, sizeof a.bias);
    a.entries = w->entries;
    a.big = w->big;
    // a bucket far above the mean (n / nb points per bucket and window for uniform digits) gets a workgroup of its own
    a.big_threshold = (unsigned int)std::max<size_t>(128, 4 * (n / w->nb));
    a.big_cap = w->big_cap;
    a.buckets = G1XPlanes{w->xyzz, nbk};
    a.parts = G1XPlanes{w->xyzz + 8 * nbk, w->nparts};
    a.wins = G1XPlanes{w->xyzz + 8 * (nbk + w->nparts), (size_t)w->W};
    a.chunk = w->chunk;
    if (tm && tm->on) HIPCHK(hipEventRecord(tm->ev[0], st));
    HIPCHK(hipMemsetAsync(a.big, 0, sizeof(unsigned int), st));
    HIPCHK(hipMemsetAsync(a.big + a.big_cap + 1, 0, sizeof(unsigned int), st));
    {
        static std::once_flag once;       // histograms above 64 KiB of dynamic LDS need the attribute (gfx950: 160 KiB per CU)
        std::call_once(once, [] {
            (void)hipFuncSetAttribute((const void*)k_msm_hist, hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4);
            (void)hipFuncSetAttribute((const void*)k_msm_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4);
        });
        // windows fastest: the W workgroups that read one chunk of the scalars run side by side (window by window instead:
        // -3 % at 2^22 points, +3 % at 2^20)
        const dim3 sgrid(w->W, w->nchunk), sblock(MSM_SORT_THREADS);
        const size_t lds = (size_t)w->nb * sizeof(unsigned int);
        hipLaunchKernelGGL(k_msm_hist, sgrid, sblock, lds, st, a);
        hipLaunchKernelGGL(k_msm_totals, dim3(w->ntiles), dim3(MSM_SCAN_THREADS), 0, st, a);
        hipLaunchKernelGGL(k_msm_scan, dim3(1), dim3(MSM_SCAN_THREADS), 0, st, a);
        hipLaunchKernelGGL(k_msm_offsets, dim3(w->ntiles), dim3(MSM_SCAN_THREADS), 0, st, a);
        hipLaunchKernelGGL(k_msm_scatter, sgrid, sblock, lds, st, a);
        hipLaunchKernelGGL(k_msm_order, dim3(w->W), sblock, 0, st, a);
    }
    HIPCHK(hipGetLastError());
    if (tm && tm->on) HIPCHK(hipEventRecord(tm->ev[1], st));
    hipLaunchKernelGGL(k_msm_accumulate, dim3((unsigned)((nbk + GKR_BLOCK - 1) / GKR_BLOCK)), dim3(GKR_BLOCK), 0, st, a);
    if (tm && tm->on) HIPCHK(hipEventRecord(tm->ev[2], st));
    hipLaunchKernelGGL(k_msm_accumulate_big, dim3(1024), dim3(GKR_BLOCK), 0, st, a);
    if (tm && tm->on) HIPCHK(hipEventRecord(tm->ev[3], st));
    hipLaunchKernelGGL(k_msm_reduce_chunks, dim3((unsigned)((w->nparts + GKR_BLOCK - 1) / GKR_BLOCK)), dim3(GKR_BLOCK), 0, st, a);
    hipLaunchKernelGGL(k_msm_reduce_windows, dim3(w->W), dim3(GKR_BLOCK), 0, st, a);
    HIPCHK(hipGetLastError());
    if (tm && tm->on) HIPCHK(hipEventRecord(tm->ev[4], st));
    HIPCHK(hipMemcpyAsync(w->h_wins, a.wins.base, 8 * (size_t)w->W * sizeof(uint4), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(w->h_wins + 8 * (size_t)w->W, a.big + a.big_cap + 1, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
    if (tm && tm->on) HIPCHK(hipEventRecord(tm->ev[5], st));
    return 0;
}

// sum_j 2^(c j) win_j by Horner's rule, then affine (infinity -> (0, 0), gnark-crypto's encoding)
hfp::Aff msm_host_tail(const MsmWork* w) {
    hfp::XYZZ acc = hfp::xyzz_inf();
    const size_t W = (size_t)w->W;
    for (int j = w->W - 1; j >= 0; j--) {
        for (int k = 0; k < w->c; k++) hfp::xyzz_dbl(acc);
        hfp::XYZZ p;
        p.x = fp_from_words(w->h_wins[0 * W + j], w->h_wins[1 * W + j]);
        p.y = fp_from_words(w->h_wins[2 * W + j], w->h_wins[3 * W + j]);
        p.zz = fp_from_words(w->h_wins[4 * W + j], w->h_wins[5 * W + j]);
        p.zzz = fp_from_words(w->h_wins[6 * W + j], w->h_wins[7 * W + j]);
        hfp::xyzz_add(acc, p);
    }
    return hfp::to_affine(acc);
}

int msm_check_points(const uint64_t* points, size_t n) {
    // every coordinate must be a canonical fp.Element: the lazy range of the kernels starts from values below p
    for (size_t i = 0; i < 2 * n; i++)
        if (hfp::geq_p(points + 4 * i)) return fail("msm: coordinate %zu of point %zu is not a canonical fp.Element", i & 1, i >> 1);
    return 0;
}

int msm_run(gkrhip_g1_bases* b, const uint64_t* scalars, size_t n, int flags, uint64_t out_affine[8]) {
    if (n > b->n) return fail("msm: %zu scalars for %zu bases", n, b->n);
    std::lock_guard<std::mutex> lk(b->mu);
    CHK(msm_work_prepare(&b->w, std::max<size_t>(b->n, 1), b->c_forced));
    if (n) HIPCHK(hipMemcpyAsync(b->w.scalars, scalars, n * 32, hipMemcpyHostToDevice, cx().stream));
    CHK(msm_dev(b, b->w.scalars, n, flags, nullptr));
    HIPCHK(hipStreamSynchronize(cx().stream));
    if (b->w.h_wins[8 * (size_t)b->w.W].x) return fail("msm: a scalar is not below 2^254 (not a reduced fr.Element)");
    const hfp::Aff r = msm_host_tail(&b->w);
    memcpy(out_affine, r.x.l, 32);
    memcpy(out_affine + 4, r.y.l, 32);
    return 0;
}

int g1_bases_alloc(gkrhip_g1_bases** out, size_t n) {
    if (n > kMsmMaxPoints) return fail("msm: %zu points (at most 2^26)", n);
    gkrhip_g1_bases* b = new gkrhip_g1_bases();
    b->n = n;
    hipError_t e = hipMalloc((void**)&b->d_points, std::max<size_t>(n, 1) * 64);
    if (e != hipSuccess) {
        delete b;
        return fail("hipMalloc of %zu G1 points failed: %s", n, hipGetErrorString(e));
    }
    *out = b;
    return 0;
}
void g1_bases_free(gkrhip_g1_bases* b) {
    if (!b) return;
    b->w.release();
    if (b->d_points) (void)hipFree(b->d_points);
    delete b;
}
// out_dev[i] = [s_i] base on the device (affine, canonical); scalars on the host
int g1_batch_mul_dev(uint4* out_dev, const uint64_t base[8], const uint64_t* scalars, size_t n, int flags) {
    if (hfp::geq_p(base) || hfp::geq_p(base + 4)) return fail("g1: the base point's coordinates are not canonical fp.Elements");
    if (!n) return 0;
    uint4* d_s = nullptr;
    HIPCHK(hipMalloc((void**)&d_s, n * 32));
    hipError_t e = hipMemcpyAsync(d_s, scalars, n * 32, hipMemcpyHostToDevice, cx().stream);
    if (e == hipSuccess) {
        MsmArgs a;
        memset(&a, 0, sizeof a);
        a.scalars = d_s;
        a.n = n;
        a.scalars_mont = (flags & GKRHIP_MSM_SCALARS_MONT) ? 1 : 0;
        G1Aff bp;
        memcpy(bp.x.v, base, 32);
        memcpy(bp.y.v, base + 4, 32);
        hipLaunchKernelGGL(k_g1_batch_scalar_mul, dim3((unsigned)((n + GKR_BLOCK - 1) / GKR_BLOCK)), dim3(GKR_BLOCK), 0, cx().stream, a, bp, out_dev);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(cx().stream);
    }
    (void)hipFree(d_s);
    if (e != hipSuccess) return fail("g1 batch scalar multiplication failed: %s", hipGetErrorString(e));
    return 0;
}
