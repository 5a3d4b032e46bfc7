// fb_sort.hip -- the sort of a fixed-base MSM (g1.hip.h: "Fixed-base MSM"): (bucket, entry) pairs by bucket, stable.
// rocPRIM's device radix sort: the keys are 21 bits wide and the entries index W * n table points, which the library's own
// LDS-staged counting sort (16-bit digit planes, 32-bit entries that carry the low bucket bits) does not hold; a plain library
// sort beside hand-written kernels, like a library GEMM would be (measured: profiles/r06_msm_fixed_base.txt).
// A translation unit of its own: rocPRIM's templates take ~13 s to compile and change with nothing else.
#include <hip/hip_runtime.h>
#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>

// bytes of temporary storage for `n` pairs; 0 on success
int gkrhip_fb_sort_bytes(size_t n, int key_bits, size_t* bytes) {
    unsigned int* p = nullptr;
    *bytes = 0;
    return (int)rocprim::radix_sort_pairs(nullptr, *bytes, p, p, p, p, n, 0u, (unsigned int)key_bits, (hipStream_t) nullptr);
}
int gkrhip_fb_sort(void* tmp, size_t bytes, const unsigned int* keys_in, unsigned int* keys_out, const unsigned int* vals_in,
                   unsigned int* vals_out, size_t n, int key_bits, hipStream_t st) {
    return (int)rocprim::radix_sort_pairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, (unsigned int)key_bits, st);
}
