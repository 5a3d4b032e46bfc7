// host_msm.hip.h -- G1 multi-scalar multiplication on device-resident bases: window choice, work buffers, the kernel
// sequence of g1.hip.h and the scalar tail on the host (Horner over the window sums, conversion to affine).
// Reference: gnark-crypto's (*G1Jac).MultiExp as called at prover/gadget/prove.go:76,91,189,202,221 (the bases there are
// proving-key vectors, fixed across proofs: here they are uploaded once and stay in HBM, like the assignment of a session).
// Included by gkrhip.hip inside its anonymous namespace.
#pragma once

const size_t kMsmMaxPoints = (size_t)1 << 26;      // entries are 31-bit point indices + a sign bit; W * n stays below 2^32

// number of windows for c-bit signed digits of a scalar below 2^254 (the top window absorbs the last carry: g1.hip.h)
inline int msm_windows(int c) { return (255 + c - 1) / c; }
// Window size, from the measured sweep (tools/msm_window_sweep.py, G1 and G2 alike): W * n mixed additions against
// W * 2^(c-1) buckets whose reduction is latency-bound (~0.4 ms whatever c: a chain of ~45 group operations per lane), more
// buckets meaning more lanes for the accumulation, and c = 15 and 16 leaving no short top window (255 = 17 * 15; a top window
// of a few bits puts n / 4 points into each of its buckets):
//     n < 2^11: c = 8     2^11 .. 2^13: c = log2(n) - 2     2^14 .. 2^18: c = 15     from 2^19: c = 16
// (G1, ms with c = 14 / 15 / 16: 2^18 1.51 / 1.16 / 1.20, 2^19 2.14 / 1.56 / 1.54, 2^20 3.42 / 2.49 / 2.30,
//  2^21 6.05 / 4.39 / 3.79, 2^22 11.1 / 8.1 / 6.9; 2^8: 0.33 with c = 8 against 0.81 with 6.)
inline int msm_pick_c(size_t n) {
    int logn = 0;
    while (((size_t)1 << (logn + 1)) <= n) logn++;
    if (logn >= 19) return 16;
    if (logn >= 14) return 15;
    return std::max(8, logn - 2);
}

struct MsmWork {
    size_t n_cap = 0;
    int c = 0, W = 0;
    unsigned int nb = 0;
    int chunk = 0;
    unsigned int big_cap = 0, seg = 4096;
    unsigned int* counts = nullptr;      // count | offset | order (W * nb words each) | chunk histograms (W * nchunk * nb)
    unsigned int nchunk = 0, ntiles = 0;
    size_t chunk_len = 0;
    uint32_t bias[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned int* entries = nullptr;
    unsigned short* digits = nullptr;    // W planes of 16-bit raw windows
    unsigned int* big = nullptr;
    uint4* scalars = nullptr;            // n_cap x 32 B staging of host scalars
    uint4* xyzz = nullptr;               // buckets | parts | wins planes
    uint4* h_wins = nullptr;             // pinned: 4 * w16 planes x W, then the error word
    size_t nparts = 0;
    int w16 = 2;                         // planes of a bucket: 4 * w16
    // two-level sort (lowbits > 0): coarse bins of 2^lowbits buckets, slices of slice_len entries
    int lowbits = 0;
    unsigned int slice_len = 0, slice_cap = 0, ctiles = 0;
    unsigned int* lvl2 = nullptr;        // coarse count | offset | first slice (W * nbc each) | chunk histograms | tile sums | slice list | slice histograms
    unsigned int* c_entries = nullptr;   // the coarse pass's entries
    bool ready = false;                  // every buffer this handle needs is allocated (set last by msm_work_prepare)
    bool has_sort = false;               // ... the sort's buffers among them (a handle that only ever SUMS on another handle's sort, in
                                         // a shared call, holds buckets, partial sums and the pinned window sums only: 3+ GiB less at 2^24)
    void release() {
        if (counts) (void)hipFree(counts);
        if (lvl2) (void)hipFree(lvl2);
        if (c_entries) (void)hipFree(c_entries);
        if (entries) (void)hipFree(entries);
        if (digits) (void)hipFree(digits);
        if (big) (void)hipFree(big);
        if (scalars) (void)hipFree(scalars);
        if (xyzz) (void)hipFree(xyzz);
        if (h_wins) (void)hipHostFree(h_wins);
        *this = MsmWork();
    }
};
// Fixed-base tables and work buffers of a handle (g1.hip.h: "Fixed-base MSM"; gkrhip_msm_g1_precompute)
struct MsmFixedBase {
    int c = 0, W = 0;                    // window bits; windows of a scalar
    unsigned int nb = 0;                 // buckets: 2^(c-1), ONE space for all windows
    uint4* tables = nullptr;             // [W][n] points: [2^(c j)] P_i, affine
    // the three-level counting sort (g1.hip.h: FbSortArgs): digit planes, the entries and residual keys of levels 1 and 2, bookkeeping
    unsigned int* raw = nullptr;                     // [W][n] bucket | sign << 31 (k_msm_fb_digits)
    unsigned int* vals[2] = {nullptr, nullptr};      // [W * n] entries (table index | sign << 31): level 1 and 3 write [0], level 2 writes [1]
    unsigned short* k16[2] = {nullptr, nullptr};
    unsigned int* lv = nullptr;
    unsigned char wb[32] = {0};          // bits of window j: the 255 bits dealt evenly over the W windows, the wider ones on top
    unsigned short wo[32] = {0};         // first bit of window j
    int bits1 = 0, bits2 = 0, bits3 = 0;
    unsigned int nchunk = 0, slice_len = 0, cap1 = 0, cap2 = 0;
    size_t chunk_len = 0;
    MsmWork w;                           // counts, big list, scalars, bucket planes, pinned window sum: geometry (c, W = 1, nb)
    void release() {
        for (void* p : {(void*)tables, (void*)raw, (void*)vals[0], (void*)vals[1], (void*)k16[0], (void*)k16[1], (void*)lv})
            if (p) (void)hipFree(p);
        w.release();
        *this = MsmFixedBase();
    }
};
struct MsmBases {
    uint4* d_points = nullptr;     // n x 64 B (G1) or 128 B (G2)
    size_t n = 0;
    int w16 = 2;                   // 16-byte words per coordinate: 2 (Fp: G1) | 4 (Fp2: G2)
    std::mutex mu;                 // one MSM at a time per handle (they share the work buffers)
    int c_forced = 0;              // gkrhip_msm_g1_set_window
    MsmWork w;
    MsmFixedBase fb;               // gkrhip_msm_g1_precompute: when fb.tables exists, the handle's own MSMs take the fixed-base path
};
}  // namespace (the handle types are part of the C ABI)
struct gkrhip_g1_bases : MsmBases {};
struct gkrhip_g2_bases : MsmBases {};
namespace {

// Levels of the counting sort (g1.hip.h): one pass up to 2^19 points, two from 2^20 (the single pass is bound by partial-line
// writes there); gkrhip_set_option("msm_sort_levels", 1 | 2) forces either for the tests and the A/B.
std::atomic<int> g_msm_sort_levels{0};
inline int msm_pick_lowbits(size_t n, int c) {
    const int forced = g_msm_sort_levels.load();
    if (forced == 1 || c < 3) return 0;
    if (forced != 2 && n < ((size_t)1 << 20)) return 0;
    int logn = 0;
    while (((size_t)1 << logn) < n) logn++;
    return std::max(1, std::min(std::min(7, 31 - logn), (c - 1) / 2));      // the entry keeps index, sign and low bits in 32 bits
}

int msm_work_prepare(MsmWork* w, size_t n, int c_forced, int w16, bool sum_only = false) {
    const int c = c_forced > 0 ? c_forced : msm_pick_c(n);
    if (c < 2 || c > 16) return fail("msm: window size %d outside 2..16", c);
    const int lowbits = msm_pick_lowbits(n, c);
    if (w->ready && w->c == c && w->n_cap >= n && w->w16 == w16 && w->lowbits == lowbits && (sum_only || w->has_sort)) return 0;
    w->release();
    // an allocation that fails half-way (several GiB per handle at 2^24 points) must not leave a half-prepared handle behind:
    // the next MSM would take the fast path above and launch its kernels on null pointers
    struct Guard {
        MsmWork* w;
        ~Guard() {
            if (w) w->release();
        }
    } guard{w};
    w->c = c;
    w->lowbits = lowbits;
    w->w16 = w16;
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
    if (nbk > ((size_t)1 << MSM_LIST_ID_BITS)) return fail("msm: %zu buckets do not fit the %d-bit ids of the big-bucket list", nbk, MSM_LIST_ID_BITS);
    // segments of big buckets: a bucket is big above max(128, 4 n / nb) points, so there are at most W * min(n / 128, nb / 4) of
    // them, and cutting them into segments of `seg` points adds at most W * n / seg entries
    w->seg = 4096;
    while ((size_t)w->seg * 2048 < n) w->seg <<= 1;           // at most 2048 segments per bucket (11 bits of the list entry)
    w->big_cap = (unsigned int)(std::min<size_t>((size_t)w->W * n / 128, (size_t)w->W * w->nb / 4) + (size_t)w->W * n / w->seg + 16);
    // sorting workgroups: one per window and chunk of the scalars; a chunk is long enough to amortise the workgroup's
    // histogram traffic (2^(c-1) words in and out) and short enough that W * nchunk workgroups cover the CUs several times
    // (one workgroup per CU at c = 16: measured 2.72 / 1.96 / 1.96 ms of sorting at 2^22 points with 8 / 16..32 / 64 chunks,
    // 10.6 / 8.9 / 6.9 ms at 2^24 with 8 / 16 / 64)
    w->nchunk = (unsigned int)std::min<size_t>(128, std::max<size_t>(1, n / 131072));
    w->chunk_len = ((n + w->nchunk - 1) / w->nchunk + 7) & ~(size_t)7;      // the sorting kernels read eight 16-bit digits per load
    memset(w->bias, 0, sizeof w->bias);
    for (int j = 0; j + 1 < w->W; j++) {
        const int bit = j * c + c - 1;
        w->bias[bit >> 5] |= 1u << (bit & 31);
    }
    w->ntiles = (unsigned int)((nbk + MSM_SCAN_TILE - 1) / MSM_SCAN_TILE);
    if (sum_only) {
        HIPCHK(hipMalloc((void**)&w->xyzz, (size_t)4 * w16 * (nbk + w->nparts + (size_t)w->W + w->big_cap) * sizeof(uint4)));
        HIPCHK(hipHostMalloc((void**)&w->h_wins, ((size_t)4 * w16 * w->W + 1) * sizeof(uint4)));
        w->ready = true;
        guard.w = nullptr;
        return 0;
    }
    if (lowbits) {
        const size_t nbc = (size_t)w->W * (w->nb >> lowbits);      // coarse bins
        w->slice_len = 8192;
        while ((size_t)w->slice_len * 2048 < n) w->slice_len <<= 1;           // at most 2048 slices per bin (11 bits of the list entry)
        w->slice_cap = (unsigned int)((size_t)w->W * n / w->slice_len + nbc + 16);
        w->ctiles = (unsigned int)((nbc + MSM_SCAN_TILE - 1) / MSM_SCAN_TILE);
        const size_t words = (3 + (size_t)w->nchunk) * nbc + w->ctiles + ((size_t)w->slice_cap + 2) + ((size_t)w->slice_cap << lowbits);
        HIPCHK(hipMalloc((void**)&w->lvl2, words * sizeof(unsigned int)));
        HIPCHK(hipMalloc((void**)&w->c_entries, std::max<size_t>(1, (size_t)w->W * n) * sizeof(unsigned int)));
        HIPCHK(hipMalloc((void**)&w->counts, (3 * nbk) * sizeof(unsigned int)));
    } else
        HIPCHK(hipMalloc((void**)&w->counts, ((3 + (size_t)w->nchunk) * nbk + w->ntiles) * sizeof(unsigned int)));
    HIPCHK(hipMalloc((void**)&w->entries, std::max<size_t>(1, (size_t)w->W * n) * sizeof(unsigned int)));
    HIPCHK(hipMalloc((void**)&w->digits, std::max<size_t>(8, (size_t)w->W * ((n + 7) & ~(size_t)7)) * sizeof(unsigned short)));
    HIPCHK(hipMalloc((void**)&w->big, ((size_t)w->big_cap + 2) * sizeof(unsigned int)));
    HIPCHK(hipMalloc((void**)&w->scalars, std::max<size_t>(1, n) * 32));
    HIPCHK(hipMalloc((void**)&w->xyzz, (size_t)4 * w16 * (nbk + w->nparts + (size_t)w->W + w->big_cap) * sizeof(uint4)));
    HIPCHK(hipHostMalloc((void**)&w->h_wins, ((size_t)4 * w16 * w->W + 1) * sizeof(uint4)));      // + the error word
    w->ready = w->has_sort = true;
    guard.w = nullptr;
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

// The device part in two steps.  msm_sort_dev: digits and the counting sort of the (window, bucket) pairs -- a function of the
// scalars and the window size alone, so two MSMs over the same scalars (bs1 and Bs of prove.go:189,277) share it; fills *out.
// msm_sum_dev<F>: bucket sums, window sums; the W window sums land in w->h_wins (the caller synchronises the stream).
// d_scalars: n x 32 B on the device.
int msm_sort_dev(MsmWork* w, const uint4* d_scalars, size_t n, int flags, MsmTimes* tm, const uint4* d_scalars_hi, MsmArgs* out) {
    hipStream_t st = cx().stream;
    const size_t nbk = (size_t)w->W * w->nb;
    MsmArgs a;
    memset(&a, 0, sizeof a);
    a.scalars = d_scalars;
    a.scalars_hi = d_scalars_hi;
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
    a.seg = w->seg;
    a.chunk = w->chunk;
    if (tm && tm->on) HIPCHK(hipEventRecord(tm->ev[0], st));
    a.err = a.big + a.big_cap + 1;
    a.digits = w->digits;
    a.dstride = (n + 7) & ~(size_t)7;
    HIPCHK(hipMemsetAsync(a.big, 0, sizeof(unsigned int), st));
    HIPCHK(hipMemsetAsync(a.err, 0, sizeof(unsigned int), st));
    if (n) hipLaunchKernelGGL(k_msm_digits, dim3((unsigned)((a.dstride + GKR_BLOCK - 1) / GKR_BLOCK)), dim3(GKR_BLOCK), 0, st, a);
    if (w->lowbits) {
        // coarse pass: the same five kernels over W * nbc bins; its "big bucket" list (threshold 0: every nonempty bin, cut
        // into segments of slice_len entries) is the slice list of the refine kernels
        const unsigned int nbc1 = w->nb >> w->lowbits;
        const size_t nbc = (size_t)w->W * nbc1;
        MsmArgs ca = a;
        ca.lowbits = w->lowbits;
        ca.count = w->lvl2;
        ca.offset = w->lvl2 + nbc;
        ca.bin_first = w->lvl2 + 2 * nbc;
        ca.chist = w->lvl2 + 3 * nbc;
        ca.tile_sum = ca.chist + (size_t)w->nchunk * nbc;
        ca.ntiles = w->ctiles;
        ca.big = ca.tile_sum + w->ctiles;
        ca.big_cap = w->slice_cap;
        ca.big_threshold = 0;
        ca.seg = w->slice_len;
        ca.entries = w->c_entries;
        a.c_entries = w->c_entries;
        a.c_count = ca.count;
        a.c_offset = ca.offset;
        a.c_first = ca.bin_first;
        a.slices = ca.big;
        a.slice_cap = w->slice_cap;
        a.slice_len = w->slice_len;
        a.rbits = w->lowbits;
        a.slice_hist = ca.big + w->slice_cap + 2;
        a.chist = nullptr;
        a.tile_sum = nullptr;
        HIPCHK(hipMemsetAsync(ca.big, 0, sizeof(unsigned int), st));
        const dim3 sgrid(w->W, w->nchunk), sblock(MSM_SORT_THREADS);
        const size_t lds = (size_t)nbc1 * sizeof(unsigned int);
        hipLaunchKernelGGL(k_msm_hist, sgrid, sblock, lds, st, ca);
        hipLaunchKernelGGL(k_msm_totals, dim3(w->ctiles), dim3(MSM_SCAN_THREADS), 0, st, ca);
        hipLaunchKernelGGL(k_msm_scan, dim3(1), dim3(MSM_SCAN_THREADS), 0, st, ca);
        hipLaunchKernelGGL(k_msm_offsets, dim3(w->ctiles), dim3(MSM_SCAN_THREADS), 0, st, ca);
        hipLaunchKernelGGL(k_msm_scatter_coarse, sgrid, sblock, 0, st, ca);
        hipLaunchKernelGGL(k_msm_refine_count, dim3(w->slice_cap), dim3(MSM_REFINE_THREADS), 0, st, a);
        hipLaunchKernelGGL(k_msm_refine_offsets, dim3((unsigned)((nbk + GKR_BLOCK - 1) / GKR_BLOCK)), dim3(GKR_BLOCK), 0, st, a);
        hipLaunchKernelGGL(k_msm_refine_scatter, dim3(w->slice_cap), dim3(MSM_REFINE_THREADS), 0, st, a);
        hipLaunchKernelGGL(k_msm_order, dim3(w->W), sblock, 0, st, a);
    } else {
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
    *out = a;
    return 0;
}
template <class F>
int msm_sum_dev(MsmWork* w, const uint4* d_points, MsmArgs a, MsmTimes* tm) {
    hipStream_t st = cx().stream;
    const size_t nbk = (size_t)w->W * w->nb;
    a.points = d_points;
    const size_t npl = (size_t)4 * F::W16;      // planes of an XYZZ point
    a.buckets = XPlanes{w->xyzz, nbk};
    a.parts = XPlanes{w->xyzz + npl * nbk, w->nparts};
    a.wins = XPlanes{w->xyzz + npl * (nbk + w->nparts), (size_t)w->W};
    a.bigparts = XPlanes{w->xyzz + npl * (nbk + w->nparts + (size_t)w->W), (size_t)w->big_cap};
    if (tm && tm->on) HIPCHK(hipEventRecord(tm->ev[1], st));
    hipLaunchKernelGGL(k_msm_accumulate<F>, dim3((unsigned)((nbk + GKR_BLOCK - 1) / GKR_BLOCK)), dim3(GKR_BLOCK), 0, st, a);
    if (tm && tm->on) HIPCHK(hipEventRecord(tm->ev[2], st));
    hipLaunchKernelGGL(k_msm_accumulate_big<F>, dim3(1024), dim3(GKR_BLOCK), 0, st, a);
    hipLaunchKernelGGL(k_msm_big_combine<F>, dim3(256), dim3(GKR_BLOCK), 0, st, a);
    if (tm && tm->on) HIPCHK(hipEventRecord(tm->ev[3], st));
    hipLaunchKernelGGL(k_msm_reduce_chunks<F>, dim3((unsigned)((w->nparts + GKR_BLOCK - 1) / GKR_BLOCK)), dim3(GKR_BLOCK), 0, st, a);
    hipLaunchKernelGGL(k_msm_reduce_windows<F>, dim3(w->W), dim3(GKR_BLOCK), 0, st, a);
    HIPCHK(hipGetLastError());
    if (tm && tm->on) HIPCHK(hipEventRecord(tm->ev[4], st));
    HIPCHK(hipMemcpyAsync(w->h_wins, a.wins.base, npl * w->W * sizeof(uint4), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(w->h_wins + npl * w->W, a.err, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
    if (tm && tm->on) HIPCHK(hipEventRecord(tm->ev[5], st));
    return 0;
}
template <class F>
int msm_dev(MsmBases* b, const uint4* d_scalars, size_t n, int flags, MsmTimes* tm, const uint4* d_scalars_hi = nullptr) {
    MsmArgs a;
    CHK(msm_sort_dev(&b->w, d_scalars, n, flags, tm, d_scalars_hi, &a));
    return msm_sum_dev<F>(&b->w, b->d_points, a, tm);
}

// coordinate j of plane group g (0: x, 1: y, 2: zz, 3: zzz) of the window sums
inline void msm_read_coord(const MsmWork* w, int g, int j, hfp::E* out) {
    const size_t W = (size_t)w->W, base = (size_t)g * w->w16;
    *out = fp_from_words(w->h_wins[base * W + j], w->h_wins[(base + 1) * W + j]);
}
inline void msm_read_coord(const MsmWork* w, int g, int j, hfp::E2* out) {
    const size_t W = (size_t)w->W, base = (size_t)g * w->w16;
    out->a0 = fp_from_words(w->h_wins[base * W + j], w->h_wins[(base + 1) * W + j]);
    out->a1 = fp_from_words(w->h_wins[(base + 2) * W + j], w->h_wins[(base + 3) * W + j]);
}
// sum_j 2^(c j) win_j by Horner's rule, then affine (infinity -> (0, 0), gnark-crypto's encoding)
template <class HF>
hfp::AffH<HF> msm_host_tail(const MsmWork* w) {
    hfp::XyzzH<HF> acc = hfp::xyzz_inf<HF>();
    for (int j = w->W - 1; j >= 0; j--) {
        for (int k = 0; k < w->c; k++) hfp::xyzz_dbl(acc);
        hfp::XyzzH<HF> p;
        msm_read_coord(w, 0, j, &p.x);
        msm_read_coord(w, 1, j, &p.y);
        msm_read_coord(w, 2, j, &p.zz);
        msm_read_coord(w, 3, j, &p.zzz);
        hfp::xyzz_add(acc, p);
    }
    return hfp::to_affine(acc);
}
// the device's error word: 0 | 1 (a scalar is not below 2^254) | 2 (a segment or slice list overflowed: excluded by their sizing)
inline int msm_check_error(const MsmWork* w, const char* what) {
    const unsigned int e = w->h_wins[(size_t)4 * w->w16 * w->W].x;
    if (e == 0) return 0;
    if (e == 1) return fail("msm: %s is not below 2^254 (not a reduced fr.Element)", what);
    return fail("msm: internal list overflow (error word %u)", e);
}

int msm_check_points(const uint64_t* points, size_t n, int w16) {
    // every coordinate must be a canonical fp.Element: the lazy range of the kernels starts from values below p
    const size_t per = (size_t)w16;         // fp.Elements per point: 2 (G1) | 4 (G2)
    for (size_t i = 0; i < per * n; i++)
        if (hfp::geq_p(points + 4 * i)) return fail("msm: element %zu of point %zu is not a canonical fp.Element", i % per, i / per);
    return 0;
}

// ---- fixed-base path ---------------------------------------------------------------------------------------------------
// window size of the one bucket space: 13 n additions at c = 20, 12 n at c = 22 (against 16 n), the reduction over 2^(c-1) buckets once
// (measured, profiles/r06_msm_fixed_base.txt, fixed-base c = 20 / c = 22 / per-window: 2^20 points 2.03 / 2.36 / 2.22 ms, 2^22 6.24 / 6.12 /
// 6.87, 2^23 11.7 / 11.3 / 13.3, 2^24 21.9 / 20.4 / 25.8, 2^25 42.8 / 38.6 / 50.2)
// G2 (w16 = 4: the reduction over 2^21 buckets of Fp2 points is 3.4 ms): c = 20 / c = 22 / per-window 2^20 points 6.00 / 7.40 / 6.66 ms,
// 2^21 9.76 / 10.75 / 11.14, 2^22 16.3 / 17.4 / 19.7, 2^23 29.75 / 29.81 / 36.3
inline int msm_fb_pick_c(size_t n, int w16) {
    const size_t from22 = (size_t)1 << (w16 == 4 ? 24 : 22);
    return n >= from22 ? 22 : n >= ((size_t)1 << 17) ? 20 : 16;
}
// tables and buffers; the tables are computed here (one lane per point: c doublings and an inversion per table entry)
template <class F>
int msm_fb_prepare(MsmBases* b, int c_or_0) {
    const size_t n = std::max<size_t>(b->n, 1);
    const int c = c_or_0 > 0 ? c_or_0 : msm_fb_pick_c(n, b->w16);
    if (c < 8 || c > MSM_LIST_ID_BITS + 1) return fail("msm: fixed-base window size %d outside 8..%d", c, MSM_LIST_ID_BITS + 1);
    MsmFixedBase& f = b->fb;
    if (f.tables && f.c == c) return 0;
    f.release();
    struct Guard {
        MsmFixedBase* f;
        ~Guard() {
            if (f) f->release();
        }
    } guard{&f};
    f.c = c;
    f.W = msm_windows(c);
    f.nb = 1u << (c - 1);
    if (f.W > 32) return fail("msm: %d windows (at most 32)", f.W);
    {
        const int base = 255 / f.W, extra = 255 - base * f.W;      // `extra` windows of base + 1 bits (<= c), on top
        int bit = 0;
        for (int j = 0; j < f.W; j++) {
            f.wb[j] = (unsigned char)(base + (j >= f.W - extra ? 1 : 0));
            f.wo[j] = (unsigned short)bit;
            bit += f.wb[j];
        }
    }
    const size_t V = (size_t)f.W * n;
    if (V >= ((size_t)1 << 31)) return fail("msm: %d windows of %zu points do not fit 31-bit table indices", f.W, n);
    HIPCHK(hipMalloc((void**)&f.tables, V * 32 * b->w16));
    HIPCHK(hipMalloc((void**)&f.raw, V * sizeof(unsigned int)));
    for (int k = 0; k < 2; k++) HIPCHK(hipMalloc((void**)&f.vals[k], V * sizeof(unsigned int)));
    {
        const int kb = c - 1;
        f.bits3 = std::min(7, kb);
        f.bits2 = std::min(7, kb - f.bits3);
        f.bits1 = kb - f.bits3 - f.bits2;                        // <= 7 (c <= 22)
        f.nchunk = (unsigned int)std::min<size_t>(512, std::max<size_t>(1, n / 32768));
        f.chunk_len = ((n + f.nchunk - 1) / f.nchunk + 7) & ~(size_t)7;
        f.slice_len = 8192;                                      // (the slice lists of the levels split their words by the level's bins: no cap on slices per bin)
        const size_t nb1 = (size_t)1 << f.bits1, nb2 = nb1 << f.bits2;
        f.cap1 = (unsigned int)(V / f.slice_len + nb1 + 16);
        f.cap2 = (unsigned int)(V / f.slice_len + nb2 + 16);
        const size_t words = (size_t)f.W * f.nchunk * nb1 + 3 * nb1 + (f.cap1 + 2) + ((size_t)f.cap1 << f.bits2) + 3 * nb2 + (f.cap2 + 2) +
                             ((size_t)f.cap2 << f.bits3);
        HIPCHK(hipMalloc((void**)&f.lv, words * sizeof(unsigned int)));
        for (int k = 0; k < 2; k++) HIPCHK(hipMalloc((void**)&f.k16[k], V * sizeof(unsigned short)));
    }
    // the sums' buffers: W = 1 window of nb buckets
    MsmWork& w = f.w;
    w.c = c;
    w.W = 1;
    w.nb = f.nb;
    w.w16 = b->w16;
    w.n_cap = n;
    // buckets per lane of the window reduction: 2^21 buckets are throughput-bound (a lane's offset doubling is shared by its
    // chunk: 47 group operations per 8 buckets, 93 per 32), 2^15 are latency-bound (8: the per-window path's measured best)
    // (measured: 2^21 buckets 1.45 -> 0.90 ms with 32; 2^19 buckets 0.52 -> 0.80: there the chain of 93 operations is the limit)
    w.chunk = (int)std::min<unsigned int>(w.nb >= (1u << 21) ? 32 : 8, w.nb);
    w.nparts = w.nb / w.chunk;
    w.seg = 4096;
    while ((size_t)w.seg * 2048 < V) w.seg <<= 1;
    w.big_cap = (unsigned int)(std::min<size_t>(V / 128, w.nb / 4) + V / w.seg + 16);
    memset(w.bias, 0, sizeof w.bias);
    for (int j = 0; j + 1 < f.W; j++) {
        const int bit = f.wo[j] + f.wb[j] - 1;
        w.bias[bit >> 5] |= 1u << (bit & 31);
    }
    HIPCHK(hipMalloc((void**)&w.counts, (size_t)3 * w.nb * sizeof(unsigned int)));
    HIPCHK(hipMalloc((void**)&w.big, ((size_t)w.big_cap + 2) * sizeof(unsigned int)));
    HIPCHK(hipMalloc((void**)&w.scalars, n * 32));
    HIPCHK(hipMalloc((void**)&w.xyzz, (size_t)4 * b->w16 * ((size_t)w.nb + w.nparts + 1 + w.big_cap) * sizeof(uint4)));
    HIPCHK(hipHostMalloc((void**)&w.h_wins, ((size_t)4 * b->w16 + 1) * sizeof(uint4)));
    w.ready = true;
    FbWindows fw;
    memcpy(fw.wb, f.wb, sizeof fw.wb);
    hipLaunchKernelGGL(k_msm_fb_precompute<F>, dim3((unsigned)((n + GKR_BLOCK - 1) / GKR_BLOCK)), dim3(GKR_BLOCK), 0, cx().stream,
                       (const uint4*)b->d_points, f.tables, b->n, fw, f.W);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(cx().stream));
    guard.f = nullptr;
    return 0;
}
// The device part in two steps, as the per-window path: msm_fb_sort_dev -- digits and the three-level sort, a function of the
// scalars and the window geometry alone, so handles with tables of one geometry share it (msm_run_shared) -- fills *out;
// msm_sum_dev<F> (above) then runs on the handle's own tables and bucket planes.
int msm_fb_sort_dev(MsmBases* b, const uint4* d_scalars, size_t n, int flags, MsmTimes* tm, const uint4* d_scalars_hi, MsmArgs* out) {
    MsmFixedBase& f = b->fb;
    MsmWork* w = &f.w;
    hipStream_t st = cx().stream;
    const size_t V = (size_t)f.W * n;
    MsmArgs a;
    memset(&a, 0, sizeof a);
    a.scalars = d_scalars;
    a.scalars_hi = d_scalars_hi;
    a.n = n;
    a.c = f.c;
    a.W = f.W;
    a.nb = f.nb;
    a.scalars_mont = (flags & GKRHIP_MSM_SCALARS_MONT) ? 1 : 0;
    a.count = w->counts;
    a.offset = w->counts + w->nb;
    a.order = w->counts + 2 * (size_t)w->nb;
    memcpy(a.bias, w->bias, sizeof a.bias);
    a.big = w->big;
    a.big_threshold = (unsigned int)std::max<size_t>(128, 4 * (V / w->nb));
    a.big_cap = w->big_cap;
    a.seg = w->seg;
    a.chunk = w->chunk;
    a.err = a.big + a.big_cap + 1;
    a.fb_raw = f.raw;
    memcpy(a.fb_wb, f.wb, sizeof a.fb_wb);
    memcpy(a.fb_wo, f.wo, sizeof a.fb_wo);
    a.dstride = std::max<size_t>(b->n, 1);      // the tables' window stride (n <= b->n scalars: keys are packed [W][n], entries index [W][b->n])
    if (tm && tm->on) HIPCHK(hipEventRecord(tm->ev[0], st));
    HIPCHK(hipMemsetAsync(a.big, 0, sizeof(unsigned int), st));
    HIPCHK(hipMemsetAsync(a.err, 0, sizeof(unsigned int), st));
    if (n) hipLaunchKernelGGL(k_msm_fb_digits, dim3((unsigned)((n + GKR_BLOCK - 1) / GKR_BLOCK)), dim3(GKR_BLOCK), 0, st, a);
    {
        // three levels of the LDS counting sort: (vals[0], k16[0]) <- level 1, (vals[1], k16[1]) <- level 2, vals[0] <- level 3
        const size_t nb1 = (size_t)1 << f.bits1, nb2 = nb1 << f.bits2;
        unsigned int* p = f.lv;
        unsigned int* chist = p;             p += (size_t)f.W * f.nchunk * nb1;
        unsigned int* cnt1 = p;              p += nb1;
        unsigned int* off1 = p;              p += nb1;
        unsigned int* first1 = p;            p += nb1;
        unsigned int* slices1 = p;           p += f.cap1 + 2;
        unsigned int* hist1 = p;             p += (size_t)f.cap1 << f.bits2;
        unsigned int* cnt2 = p;              p += nb2;
        unsigned int* off2 = p;              p += nb2;
        unsigned int* first2 = p;            p += nb2;
        unsigned int* slices2 = p;           p += f.cap2 + 2;
        unsigned int* hist2 = p;
        HIPCHK(hipMemsetAsync(slices1, 0, sizeof(unsigned int), st));
        HIPCHK(hipMemsetAsync(slices2, 0, sizeof(unsigned int), st));
        FbSortArgs s1;
        memset(&s1, 0, sizeof s1);
        s1.raw = f.raw;
        s1.n = n;
        s1.tstride = a.dstride;
        s1.W = f.W;
        s1.nchunk = f.nchunk;
        s1.chunk_len = f.chunk_len;
        s1.sh1 = f.bits2 + f.bits3;
        s1.nb1 = (unsigned int)nb1;
        s1.chist = chist;
        s1.e_out = f.vals[0];
        s1.k_out = f.k16[0];
        s1.out_count = cnt1;
        s1.out_offset = off1;
        s1.out_first = first1;
        s1.next_list = slices1;
        s1.next_cap = f.cap1;
        s1.next_seg = f.slice_len;
        s1.next_id_bits = std::max(f.bits1, 1);
        s1.err = a.err;
        const dim3 g1(f.W, f.nchunk);
        hipLaunchKernelGGL(k_fb_l1_hist, g1, dim3(MSM_SORT_THREADS), 0, st, s1);
        hipLaunchKernelGGL(k_fb_l1_columns, dim3(s1.nb1), dim3(MSM_SCAN_THREADS), 0, st, s1);
        hipLaunchKernelGGL(k_fb_l1_offsets, dim3(1), dim3(MSM_SCAN_THREADS), 0, st, s1);
        hipLaunchKernelGGL(k_fb_l1_scatter, g1, dim3(FB_L1_SCATTER_THREADS), 0, st, s1);
        FbSortArgs s2;
        memset(&s2, 0, sizeof s2);
        s2.e_in = f.vals[0];
        s2.k_in = f.k16[0];
        s2.e_out = f.vals[1];
        s2.k_out = f.k16[1];
        s2.sh = f.bits3;
        s2.bits = f.bits2;
        s2.nbins_in = (unsigned int)nb1;
        s2.in_count = cnt1;
        s2.in_offset = off1;
        s2.in_first = first1;
        s2.slices = slices1;
        s2.slice_cap = f.cap1;
        s2.slice_len = f.slice_len;
        s2.slice_hist = hist1;
        s2.out_count = cnt2;
        s2.out_offset = off2;
        s2.out_first = first2;
        s2.next_list = slices2;
        s2.next_cap = f.cap2;
        s2.next_seg = f.slice_len;
        s2.next_threshold = 0;
        s2.id_bits = std::max(f.bits1, 1);
        s2.next_id_bits = std::max(f.bits1 + f.bits2, 1);
        s2.err = a.err;
        hipLaunchKernelGGL(k_fb_lv_count, dim3(f.cap1), dim3(FB_LV_THREADS), 0, st, s2);
        hipLaunchKernelGGL(k_fb_lv_offsets_bin, dim3((unsigned)nb1), dim3(MSM_SCAN_THREADS), 0, st, s2);      // few bins, many slices each
        hipLaunchKernelGGL(k_fb_lv_scatter, dim3(f.cap1), dim3(FB_LV_THREADS), 0, st, s2);
        FbSortArgs s3;
        memset(&s3, 0, sizeof s3);
        s3.e_in = f.vals[1];
        s3.k_in = f.k16[1];
        s3.e_out = f.vals[0];
        s3.k_out = nullptr;
        s3.sh = 0;
        s3.bits = f.bits3;
        s3.nbins_in = (unsigned int)nb2;
        s3.in_count = cnt2;
        s3.in_offset = off2;
        s3.in_first = first2;
        s3.slices = slices2;
        s3.slice_cap = f.cap2;
        s3.slice_len = f.slice_len;
        s3.slice_hist = hist2;
        s3.out_count = a.count;
        s3.out_offset = a.offset;
        s3.out_first = nullptr;
        s3.next_list = a.big;
        s3.next_cap = a.big_cap;
        s3.next_seg = a.seg;
        s3.next_threshold = a.big_threshold;
        s3.id_bits = std::max(f.bits1 + f.bits2, 1);
        s3.next_id_bits = MSM_LIST_ID_BITS;
        s3.err = a.err;
        hipLaunchKernelGGL(k_fb_lv_count, dim3(f.cap2), dim3(FB_LV_THREADS), 0, st, s3);
        hipLaunchKernelGGL(k_fb_lv_offsets, dim3((unsigned)((w->nb + GKR_BLOCK - 1) / GKR_BLOCK)), dim3(GKR_BLOCK), 0, st, s3);
        hipLaunchKernelGGL(k_fb_lv_scatter, dim3(f.cap2), dim3(FB_LV_THREADS), 0, st, s3);
        a.entries = f.vals[0];
    }
    // from here on: ONE window of nb buckets over the table array
    a.W = 1;
    a.acc_nb = std::min<unsigned int>(w->nb, 32768);      // the ordering and the bucket sums: ranges of 2^15 buckets, a workgroup orders one
    a.acc_W = w->nb / a.acc_nb;
    hipLaunchKernelGGL(k_msm_order, dim3(a.acc_W), dim3(MSM_SORT_THREADS), 0, st, a);
    HIPCHK(hipGetLastError());
    *out = a;
    return 0;
}
template <class F>
int msm_fb_dev(MsmBases* b, const uint4* d_scalars, size_t n, int flags, MsmTimes* tm, const uint4* d_scalars_hi = nullptr) {
    MsmArgs a;
    CHK(msm_fb_sort_dev(b, d_scalars, n, flags, tm, d_scalars_hi, &a));
    return msm_sum_dev<F>(&b->fb.w, b->fb.tables, a, tm);
}

template <class F, class HF>
int msm_run(MsmBases* b, const uint64_t* scalars, size_t n, int flags, uint64_t* out_affine) {
    if (n > b->n) return fail("msm: %zu scalars for %zu bases", n, b->n);
    std::lock_guard<std::mutex> lk(b->mu);
    if (b->fb.tables) {                   // fixed-base: the tables hold [2^(c j)] P_i for i < b->n, window-major with stride b->n
        MsmWork* w = &b->fb.w;
        if (n) HIPCHK(hipMemcpyAsync(w->scalars, scalars, n * 32, hipMemcpyHostToDevice, cx().stream));
        CHK(msm_fb_dev<F>(b, w->scalars, n, flags, nullptr));
        HIPCHK(hipStreamSynchronize(cx().stream));
        CHK(msm_check_error(w, "a scalar"));
        const hfp::AffH<HF> r = msm_host_tail<HF>(w);
        memcpy(out_affine, &r, sizeof r);
        return 0;
    }
    CHK(msm_work_prepare(&b->w, std::max<size_t>(b->n, 1), b->c_forced, F::W16));
    if (n) HIPCHK(hipMemcpyAsync(b->w.scalars, scalars, n * 32, hipMemcpyHostToDevice, cx().stream));
    CHK(msm_dev<F>(b, b->w.scalars, n, flags, nullptr));
    HIPCHK(hipStreamSynchronize(cx().stream));
    CHK(msm_check_error(&b->w, "a scalar"));
    const hfp::AffH<HF> r = msm_host_tail<HF>(&b->w);
    memcpy(out_affine, &r, sizeof r);        // {X, Y} as consecutive fp.Elements: the G1Affine / G2Affine image
    return 0;
}

// Several MSMs over the SAME scalars -- bs1 = MultiExp(pk.G1.B, wireValuesB) and Bs = MultiExp(pk.G2.B, wireValuesB) of
// prove.go:189,277; with the proving key's vectors expanded by points at infinity where pk.InfinityA / pk.InfinityB filter them,
// also ar = MultiExp(pk.G1.A, .) of :202 over the unfiltered wireValues -- share one upload, one decoding and one sort: the
// sort depends on the scalars and the window size only.  Every handle holds the same number of points; the first handle's
// window size serves all.  out_g1: k1 x 8 words, out_g2: k2 x 16 words.
int msm_run_shared(MsmBases* const* g1, size_t k1, MsmBases* const* g2, size_t k2, const uint64_t* scalars, size_t n, int flags,
                   uint64_t* out_g1, uint64_t* out_g2) {
    std::vector<MsmBases*> all(g1, g1 + k1);
    all.insert(all.end(), g2, g2 + k2);
    if (all.empty()) return 0;
    for (MsmBases* b : all) {
        if (!b) return fail("msm: null bases handle");
        if (b->n != all[0]->n) return fail("msm: bases of %zu and %zu points in one shared call (one length required)", all[0]->n, b->n);
    }
    for (size_t i = 0; i < k1; i++)
        if (g1[i]->w16 != FpF::W16) return fail("msm: handle %zu of the G1 list is not a G1 handle", i);
    for (size_t i = 0; i < k2; i++)
        if (g2[i]->w16 != Fp2F::W16) return fail("msm: handle %zu of the G2 list is not a G2 handle", i);
    if (n > all[0]->n) return fail("msm: %zu scalars for %zu bases", n, all[0]->n);
    std::vector<MsmBases*> order(all);
    std::sort(order.begin(), order.end());           // one locking order for every caller
    if (std::adjacent_find(order.begin(), order.end()) != order.end()) return fail("msm: a bases handle appears twice in one shared call");
    std::vector<std::unique_lock<std::mutex>> locks;
    for (MsmBases* b : order) locks.emplace_back(b->mu);
    MsmBases* first = all[0];
    // every handle with fixed-base tables of one geometry: the tables' sort (digits + three levels) once, then every handle's sums on
    // its own tables -- ar, bs1 and Bs of the Groth16 back half at 12 additions per scalar
    bool all_fb = true;
    for (MsmBases* b : all) all_fb = all_fb && b->fb.tables && b->fb.c == first->fb.c;
    if (all_fb) {
        if (n) HIPCHK(hipMemcpyAsync(first->fb.w.scalars, scalars, n * 32, hipMemcpyHostToDevice, cx().stream));
        MsmArgs a;
        CHK(msm_fb_sort_dev(first, first->fb.w.scalars, n, flags, nullptr, nullptr, &a));
        for (size_t i = 0; i < k1; i++) CHK(msm_sum_dev<FpF>(&g1[i]->fb.w, g1[i]->fb.tables, a, nullptr));
        for (size_t i = 0; i < k2; i++) CHK(msm_sum_dev<Fp2F>(&g2[i]->fb.w, g2[i]->fb.tables, a, nullptr));
        HIPCHK(hipStreamSynchronize(cx().stream));
        CHK(msm_check_error(&all.back()->fb.w, "a scalar"));
        for (size_t i = 0; i < k1; i++) {
            const hfp::AffH<hfp::HFp> r = msm_host_tail<hfp::HFp>(&g1[i]->fb.w);
            memcpy(out_g1 + 8 * i, &r, sizeof r);
        }
        for (size_t i = 0; i < k2; i++) {
            const hfp::AffH<hfp::HFp2> r = msm_host_tail<hfp::HFp2>(&g2[i]->fb.w);
            memcpy(out_g2 + 16 * i, &r, sizeof r);
        }
        return 0;
    }
    CHK(msm_work_prepare(&first->w, std::max<size_t>(first->n, 1), first->c_forced, first->w16));
    // (the other handles sum on the first one's sort: buckets, partial sums and window sums only -- unless they already hold
    // the sort's buffers of a call of their own with this geometry)
    for (size_t i = 1; i < all.size(); i++) CHK(msm_work_prepare(&all[i]->w, std::max<size_t>(all[i]->n, 1), first->w.c, all[i]->w16, /*sum_only=*/true));
    if (n) HIPCHK(hipMemcpyAsync(first->w.scalars, scalars, n * 32, hipMemcpyHostToDevice, cx().stream));
    MsmArgs a;
    CHK(msm_sort_dev(&first->w, first->w.scalars, n, flags, nullptr, nullptr, &a));
    for (size_t i = 0; i < k1; i++) CHK(msm_sum_dev<FpF>(&g1[i]->w, g1[i]->d_points, a, nullptr));      // same sort, same window geometry,
    for (size_t i = 0; i < k2; i++) CHK(msm_sum_dev<Fp2F>(&g2[i]->w, g2[i]->d_points, a, nullptr));     // its own bucket planes
    HIPCHK(hipStreamSynchronize(cx().stream));
    CHK(msm_check_error(&all.back()->w, "a scalar"));      // the error word is read back with every sum
    for (size_t i = 0; i < k1; i++) {
        const hfp::AffH<hfp::HFp> r = msm_host_tail<hfp::HFp>(&g1[i]->w);
        memcpy(out_g1 + 8 * i, &r, sizeof r);
    }
    for (size_t i = 0; i < k2; i++) {
        const hfp::AffH<hfp::HFp2> r = msm_host_tail<hfp::HFp2>(&g2[i]->w);
        memcpy(out_g2 + 16 * i, &r, sizeof r);
    }
    return 0;
}

template <class B>
int bases_alloc(B** out, size_t n, int w16) {
    if (n > kMsmMaxPoints) return fail("msm: %zu points (at most 2^26)", n);
    B* b = new B();
    b->n = n;
    b->w16 = w16;
    hipError_t e = hipMalloc((void**)&b->d_points, std::max<size_t>(n, 1) * 32 * w16);
    if (e != hipSuccess) {
        delete b;
        return fail("hipMalloc of %zu points failed: %s", n, hipGetErrorString(e));
    }
    *out = b;
    return 0;
}
template <class B>
void bases_free(B* b) {
    if (!b) return;
    b->w.release();
    b->fb.release();
    if (b->d_points) (void)hipFree(b->d_points);
    delete b;
}
template <class B>
int bases_upload(B** out, const uint64_t* points, size_t n, int w16) {
    CHK(msm_check_points(points, n, w16));
    B* b = nullptr;
    CHK(bases_alloc(&b, n, w16));
    if (n) {
        hipError_t e = hipMemcpyAsync(b->d_points, points, n * 32 * w16, hipMemcpyHostToDevice, cx().stream);
        if (e == hipSuccess) e = hipStreamSynchronize(cx().stream);
        if (e != hipSuccess) {
            bases_free(b);
            return fail("upload of %zu points failed: %s", n, hipGetErrorString(e));
        }
    }
    *out = b;
    return 0;
}
// out_dev[i] = [s_i] base on the device (affine, canonical); scalars on the host
template <class F>
int batch_mul_dev(uint4* out_dev, const uint64_t* base, const uint64_t* scalars, size_t n, int flags) {
    for (int k = 0; k < F::W16; k++)
        if (hfp::geq_p(base + 4 * k)) return fail("the base point's coordinates are not canonical fp.Elements");
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
        AffT<F> bp;
        static_assert(sizeof(bp) == (size_t)32 * F::W16, "affine image");
        memcpy(&bp, base, sizeof bp);
        hipLaunchKernelGGL(k_ec_batch_scalar_mul<F>, dim3((unsigned)((n + GKR_BLOCK - 1) / GKR_BLOCK)), dim3(GKR_BLOCK), 0, cx().stream, a, bp, out_dev);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(cx().stream);
    }
    (void)hipFree(d_s);
    if (e != hipSuccess) return fail("batch scalar multiplication failed: %s", hipGetErrorString(e));
    return 0;
}
