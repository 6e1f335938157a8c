// sd_common.h -- shared device helpers for libsegdistill_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/segdistill_hip.h"

namespace sd {

struct bf16_t {
    uint16_t bits;
};

// Round-to-nearest-even f32 -> bf16.  A plain cast through __bf16 lowers to
// v_cvt_pk_bf16_f32 on gfx950, which keeps NaNs NaN (MI355X_MICROARCH.md, Correctness boundaries).
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    __bf16 h = static_cast<__bf16>(f);
    return __builtin_bit_cast(uint16_t, h);
}

// returns the value (set == 0) or SD_OK / an SD_E_* code (set != 0); SD_E_UNSUPPORTED if the key is not ours
int cgd_tunable(const char *key, int set, int v);
int cgd_up_tunable(const char *key, int set, int v);
int sra_tunable(const char *key, int set, int v);
int token_gemm_tunable(const char *key, int set, int v);
int ce_tunable(const char *key, int set, int v);
int headfuse_tunable(const char *key, int set, int v);
int wgrad_tn_tunable(const char *key, int set, int v);
int tok_gemm_bf16_tunable(const char *key, int set, int v);
int align_stream_tunable(const char *key, int set, int v);
// the NCHW align projection's forward for Cs <= 128 with W resident in registers (align_stream.hip); SD_E_UNSUPPORTED: not its shape
int align_f32_fwd_stream(const float *X, const float *W, const float *bias, float *Y, int B, int Cs, int Ct, long P, hipStream_t st);
// token-major -> class-planes Linear, fp32 (token_gemm.hip); the dtype-dispatching C entry points live in align1x1.hip
size_t linear_nchw_workspace_bytes(int B, long P, int in_features, int out_features);
int pred_splits(int B, long P);
int linear_nchw_f32_fwd(const float *X, const float *W, const float *bias, float *Y, int B, long P, int in_features, int out_features, void *stream);
int linear_nchw_f32_bwd_data(const float *dY, const float *W, float *dX, int B, long P, int in_features, int out_features, void *stream);
int linear_nchw_f32_bwd_weight(const float *dY, const float *X, float *dW, float *dbias, int B, long P, int in_features, int out_features, void *workspace,
                               size_t workspace_bytes, void *stream);

// bf16 token-major weight gradient on transposed LDS reads (wgrad_tn.hip); used by the generic weight-gradient entry points of align1x1.hip
bool wgrad_tn_supported(long T, int M, int N, const void *dY, const void *X);
void wgrad_tn_plan(long T, int M, int N, int *nsplit, int *klen);
int wgrad_tn_launch(const void *dY, const void *X, float *slabs, long T, int M, int N, int nsplit, int klen, hipStream_t st);
long wgrad_slab_cap(long T, int M, int N, int es);      // most k-splits whose slabs stay within the set percentage of the operand bytes

}  // namespace sd
