// kernel_groups.h -- every instantiation of the heavy template kernels, by translation unit.
//
// gkrhip.hip includes this file with GKR_INST defined as `extern` (after the kernel headers, before the host code): the launch
// sites there then link against the instantiations instead of making their own, and its device pass compiles the small kernels
// only.  kern_<group>.hip defines GKR_GROUP_<GROUP> and GKR_INST as nothing: explicit instantiation definitions, device code
// and host stubs of that group.  A launch site that names an instantiation missing here still works (it is instantiated where
// it is launched, in gkrhip.hip) -- it only costs that unit compile time; the link is checked with --no-undefined.
//
// Groups are sized for build time (8 cores compile them side by side; one edit rebuilds the units that include the edited file).
#ifndef GKR_INST
#error "define GKR_INST (extern | nothing) before including kernel_groups.h"
#endif
#if defined(GKR_INST_EXTERN) || defined(GKR_GROUP_WIDE2)
GKR_INST template __global__ void k_cipher_round_wide<false, false, false, false>(Batch<CipherRoundArgs>);
GKR_INST template __global__ void k_cipher_round_wide<false, true, false, false>(Batch<CipherRoundArgs>);
GKR_INST template __global__ void k_cipher_round_wide<true, false, false, false>(Batch<CipherRoundArgs>);
GKR_INST template __global__ void k_cipher_round_wide<true, true, false, false>(Batch<CipherRoundArgs>);
#endif
#if defined(GKR_INST_EXTERN) || defined(GKR_GROUP_WIDEPRE)
GKR_INST template __global__ void k_cipher_round_wide<false, false, true, false>(Batch<CipherRoundArgs>);
GKR_INST template __global__ void k_cipher_round_wide<false, true, true, false>(Batch<CipherRoundArgs>);
GKR_INST template __global__ void k_cipher_round_wide<false, true, false, true>(Batch<CipherRoundArgs>);
GKR_INST template __global__ void k_cipher_round_wide<false, true, true, true>(Batch<CipherRoundArgs>);
#endif
#if defined(GKR_INST_EXTERN) || defined(GKR_GROUP_ROUND)
GKR_INST template __global__ void k_cipher_round<false, false>(Batch<CipherRoundArgs>);
GKR_INST template __global__ void k_cipher_round<false, true>(Batch<CipherRoundArgs>);
GKR_INST template __global__ void k_cipher_round<true, false>(Batch<CipherRoundArgs>);
GKR_INST template __global__ void k_cipher_round<true, true>(Batch<CipherRoundArgs>);
GKR_INST template __global__ void k_cipher_round_lat<false, false>(Batch<CipherRoundArgs>);
GKR_INST template __global__ void k_cipher_round_lat<false, true>(Batch<CipherRoundArgs>);
GKR_INST template __global__ void k_cipher_round_lat<true, false>(Batch<CipherRoundArgs>);
GKR_INST template __global__ void k_cipher_round_lat<true, true>(Batch<CipherRoundArgs>);
GKR_INST template __global__ void k_cipher_round_coop<false>(CipherRoundArgs);
GKR_INST template __global__ void k_cipher_round_coop<true>(CipherRoundArgs);
GKR_INST template __global__ void k_linear_round<false, false>(Batch<LinearRoundArgs>);
GKR_INST template __global__ void k_linear_round<false, true>(Batch<LinearRoundArgs>);
GKR_INST template __global__ void k_linear_round<true, false>(Batch<LinearRoundArgs>);
GKR_INST template __global__ void k_linear_round<true, true>(Batch<LinearRoundArgs>);
GKR_INST template __global__ void k_partial_eval<1, 1, 3>(PartialEvalArgs);
GKR_INST template __global__ void k_partial_eval<1, 2, 3>(PartialEvalArgs);
GKR_INST template __global__ void k_partial_eval<1, 3, 3>(PartialEvalArgs);
GKR_INST template __global__ void k_partial_eval<1, 4, 3>(PartialEvalArgs);
GKR_INST template __global__ void k_partial_eval<7, 1, 9>(PartialEvalArgs);
GKR_INST template __global__ void k_partial_eval<7, 2, 9>(PartialEvalArgs);
GKR_INST template __global__ void k_partial_eval<7, 3, 9>(PartialEvalArgs);
GKR_INST template __global__ void k_partial_eval<7, 4, 9>(PartialEvalArgs);
#endif
#if defined(GKR_INST_EXTERN) || defined(GKR_GROUP_MSM_G1)
GKR_INST template __global__ void k_msm_accumulate<FpF>(MsmArgs);
GKR_INST template __global__ void k_msm_accumulate_big<FpF>(MsmArgs);
GKR_INST template __global__ void k_msm_big_combine<FpF>(MsmArgs);
GKR_INST template __global__ void k_msm_reduce_chunks<FpF>(MsmArgs);
GKR_INST template __global__ void k_msm_reduce_windows<FpF>(MsmArgs);
GKR_INST template __global__ void k_ec_batch_scalar_mul<FpF>(MsmArgs, AffT<FpF>, uint4*);
GKR_INST template __global__ void k_msm_fb_precompute<FpF>(const uint4*, uint4*, size_t, FbWindows, int);
#endif
#if defined(GKR_INST_EXTERN) || defined(GKR_GROUP_MSM_G2A)
GKR_INST template __global__ void k_msm_accumulate<Fp2F>(MsmArgs);
GKR_INST template __global__ void k_msm_accumulate_big<Fp2F>(MsmArgs);
GKR_INST template __global__ void k_msm_big_combine<Fp2F>(MsmArgs);
#endif
#if defined(GKR_INST_EXTERN) || defined(GKR_GROUP_MSM_G2B)
GKR_INST template __global__ void k_msm_reduce_chunks<Fp2F>(MsmArgs);
#endif
#if defined(GKR_INST_EXTERN) || defined(GKR_GROUP_MSM_G2C)
GKR_INST template __global__ void k_msm_reduce_windows<Fp2F>(MsmArgs);
GKR_INST template __global__ void k_ec_batch_scalar_mul<Fp2F>(MsmArgs, AffT<Fp2F>, uint4*);
GKR_INST template __global__ void k_msm_fb_precompute<Fp2F>(const uint4*, uint4*, size_t, FbWindows, int);
#endif
#if defined(GKR_INST_EXTERN) || defined(GKR_GROUP_NTT)
GKR_INST template __global__ void k_ntt_tile<false, true>(NttPassArgs);
GKR_INST template __global__ void k_ntt_tile<true, false>(NttPassArgs);
#endif
