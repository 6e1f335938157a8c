// window_attn.hip -- window multi-head self-attention of a Swin block, forward only (the frozen teacher of BASELINE config 4), gfx950.
//
// Reference: mmseg/models/backbones/swin_transformer.py:119-153 (WindowAttention.forward): per window of N = Wh*Ww tokens and per head
//   attn = softmax((q * scale) k^T + relative_position_bias[head] + mask[window % nW]) ; out = attn v
// with q, k, v sliced out of ONE Linear's output [windows, N, 3, heads, D].
//
// Shape of the problem: N = 49, D = 32 -- 49 x 49 x 32 products, thousands of (window, head) pairs, 290 MB of operands per stage-1 block.  The
// framework's fused attention wants a materialised [windows, heads, N, N] additive tensor (111 MB per stage-1 block, read every block), runs a
// finiteness scan over it per call and takes permuted copies of q / k / v; the unshifted blocks fell to the three-kernel math path.  Here:
//   * one wave per (window, head), lane i owns query row i (49 of 64 lanes; the idle lanes shadow row N-1 and do not store);
//   * k_j and v_j are WAVE-UNIFORM rows: they arrive as scalar loads (s_load_dwordx8/16 through the scalar cache) and enter the FMAs as SGPR
//     operands -- no LDS, no barrier, no cross-lane traffic at all;
//   * the 49 scores of a row stay in registers between the two passes (max / exp / sum in-lane);
//   * bias and mask are read TRANSPOSED ([head][j][i], [window][j][i]) so that a wave's read of column j is one 196-byte segment; both tables are
//     a few hundred KB and stay in L2;
//   * q / k / v are read where the Linear left them and the output is written token-major [windows, N, heads*D], the layout the projection reads --
//     no permute copy either side.
// fp32 storage and arithmetic (packed FMAs: even and odd d accumulate separately), exp through v_exp_f32.  bf16 storage: SD_E_DTYPE.
#include <math.h>

#include "sd_common.h"

namespace sd {
namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int N, int D, bool MASK>
__global__ __launch_bounds__(256) void window_attn_fwd(const float *__restrict__ qkv, const float *__restrict__ bias_t, const float *__restrict__ mask_t,
                                                        float *__restrict__ out, int njobs, int nW, int heads, float scale) {
    const int lane = threadIdx.x & 63;
    const int job = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
    if (job >= njobs) return;                                          // whole waves only
    const int win = job / heads, head = job - win * heads;
    const int C = heads * D;
    const int i = lane < N ? lane : N - 1;
    const float *base = qkv + (size_t)win * N * 3 * C + (size_t)head * D;   // wave-uniform
    const float *bt = bias_t + (size_t)head * N * N + i;
    const float *mt = MASK ? mask_t + (size_t)(win % nW) * N * N + i : nullptr;

    f32x2 q[D / 2];
    {
        const f32x2 *qi = reinterpret_cast<const f32x2 *>(base + (size_t)i * 3 * C);
#pragma unroll
        for (int d = 0; d < D / 2; ++d) q[d] = qi[d] * scale;
    }
    // pass 1: the row's N scores.  Even and odd d accumulate separately (v_pk_fma_f32 with the k pair as an SGPR-pair operand), folded once per key.
    float s[N];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const f32x2 *kj = reinterpret_cast<const f32x2 *>(base + (size_t)j * 3 * C + C);      // uniform address: scalar loads
        f32x2 a2 = {0.f, 0.f};
#pragma unroll
        for (int d = 0; d < D / 2; ++d) a2 = __builtin_elementwise_fma(q[d], kj[d], a2);
        float a = (a2.x + a2.y) + bt[j * N];
        if (MASK) a += mt[j * N];
        s[j] = a;
        m = fmaxf(m, a);
    }
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        s[j] = __expf(s[j] - m);
        l += s[j];
    }
    // pass 2: out_i = sum_j p_ij v_j
    f32x2 o[D / 2];
#pragma unroll
    for (int d = 0; d < D / 2; ++d) o[d] = f32x2{0.f, 0.f};
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const f32x2 *vj = reinterpret_cast<const f32x2 *>(base + (size_t)j * 3 * C + 2 * C);
        const f32x2 p = {s[j], s[j]};
#pragma unroll
        for (int d = 0; d < D / 2; ++d) o[d] = __builtin_elementwise_fma(p, vj[d], o[d]);
    }
    if (lane < N) {
        const float inv = 1.f / l;
        f32x2 *oi = reinterpret_cast<f32x2 *>(out + ((size_t)win * N + i) * C + (size_t)head * D);
#pragma unroll
        for (int d = 0; d < D / 2; ++d) oi[d] = o[d] * inv;
    }
}

}  // namespace
}  // namespace sd

extern "C" {

int sd_window_attn_supported(int tokens_per_window, int head_dim) { return (tokens_per_window == 49 && head_dim == 32) ? 1 : 0; }

int sd_window_attn_fwd(const void *qkv, const float *bias_t, const float *mask_t, void *out, int dtype, long windows, int mask_windows, int heads,
                       int tokens_per_window, int head_dim, float scale, void *stream) {
    if (!qkv || !bias_t || !out) return SD_E_NULL;
    if (dtype != SD_F32) return SD_E_DTYPE;                     // a bf16 network takes the framework's fused attention (bf16 MFMA)
    if (!sd_window_attn_supported(tokens_per_window, head_dim)) return SD_E_UNSUPPORTED;
    if (windows <= 0 || heads <= 0 || windows * heads > 0x7ffffff0L) return SD_E_SHAPE;
    if (mask_t && (mask_windows <= 0 || windows % mask_windows != 0)) return SD_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(out)) & 15) return SD_E_ALIGN;
    const int njobs = (int)(windows * heads);
    const dim3 grid((unsigned)((njobs + 3) / 4)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float *q = (const float *)qkv;
    float *o = (float *)out;
    if (mask_t) hipLaunchKernelGGL((sd::window_attn_fwd<49, 32, true>), grid, block, 0, st, q, bias_t, mask_t, o, njobs, mask_windows, heads, scale);
    else hipLaunchKernelGGL((sd::window_attn_fwd<49, 32, false>), grid, block, 0, st, q, bias_t, mask_t, o, njobs, 1, heads, scale);
    return (int)hipGetLastError();
}

}  // extern "C"
