// affine_act.hip -- eval-mode BatchNorm (+ residual add) (+ ReLU) of a FROZEN convolutional network as ONE in-place pass over an NCHW map:
//   y[b, c, p] = act( x[b, c, p] * scale[c] + shift[c] (+ r[b, c, p]) ),   scale = gamma / sqrt(var + eps), shift = beta - mean * scale
// reference: the teacher ResNetV1c / PSPHead / UPerHead of mmseg (backbones/resnet.py:18-100 BasicBlock / Bottleneck `relu(bn(conv(x)))`,
// `out += identity; relu(out)`; ConvModule conv -> norm -> act in psp_head.py:38-44,84-91, uper_head.py:30-75) with the teacher in eval mode
// (SURVEY Q1).  torch runs MIOpenBatchNormFwdInferSpatialEst, an add and a ReLU as three launches and three passes (config 1: 111 + 57 + 99 of
// the 967 launches of a step for the R101 teacher, most of them 5-7 us on maps that fit L2).  HBM-bound: one read (+ the residual), one write.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sd_common.h"

namespace sd {
namespace {

// one workgroup walks a run of one (b, c) plane; VEC: HW % 4 == 0 and 16-byte aligned bases
template <bool VEC, bool RES, bool RELU>
__global__ __launch_bounds__(256) void affine_act_nchw(const float *__restrict__ x, const float *__restrict__ res, float *__restrict__ y,
                                                        const float *__restrict__ scale, const float *__restrict__ shift, int C, long HW,
                                                        int chunks_per_plane) {
    const long plane = blockIdx.x / chunks_per_plane;
    const int chunk = blockIdx.x - (int)(plane * chunks_per_plane);
    const int c = (int)(plane % C);
    const float a = scale[c], b = shift[c];
    const long base = plane * HW;
    if (VEC) {
        const long n4 = HW >> 2;
        const long per = (n4 + chunks_per_plane - 1) / chunks_per_plane;
        const long lo = chunk * per, hi = lo + per < n4 ? lo + per : n4;
        const float4 *xs = reinterpret_cast<const float4 *>(x + base);
        const float4 *rs = RES ? reinterpret_cast<const float4 *>(res + base) : nullptr;
        float4 *ys = reinterpret_cast<float4 *>(y + base);
        for (long i = lo + threadIdx.x; i < hi; i += 256) {
            float4 v = xs[i];
            v.x = fmaf(v.x, a, b), v.y = fmaf(v.y, a, b), v.z = fmaf(v.z, a, b), v.w = fmaf(v.w, a, b);
            if (RES) {
                const float4 r = rs[i];
                v.x += r.x, v.y += r.y, v.z += r.z, v.w += r.w;
            }
            if (RELU) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
            ys[i] = v;
        }
    } else {
        const long per = (HW + chunks_per_plane - 1) / chunks_per_plane;
        const long lo = chunk * per, hi = lo + per < HW ? lo + per : HW;
        for (long i = lo + threadIdx.x; i < hi; i += 256) {
            float v = fmaf(x[base + i], a, b);
            if (RES) v += res[base + i];
            if (RELU) v = fmaxf(v, 0.f);
            y[base + i] = v;
        }
    }
}

}  // namespace
}  // namespace sd

extern "C" {

int sd_affine_act_nchw(const float *x, const float *residual, float *y, const float *scale, const float *shift, long planes, int C, long HW, int relu,
                       void *stream) {
    if (!x || !y || !scale || !shift) return SD_E_NULL;
    if (planes <= 0 || C <= 0 || HW <= 0 || planes % C != 0) return SD_E_SHAPE;
    // ~2048 elements per thread-iteration-free workgroup pass; at least one workgroup per plane
    int chunks = (int)((HW + 4 * 256 * 4 - 1) / (4 * 256 * 4));
    if (chunks < 1) chunks = 1;
    if (planes * chunks > 0x7fffffffL) return SD_E_SHAPE;
    const bool vec = HW % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 15) == 0;
    const dim3 g((unsigned)(planes * chunks)), blk(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define SD_AA(V, R, A) hipLaunchKernelGGL((sd::affine_act_nchw<V, R, A>), g, blk, 0, st, x, residual, y, scale, shift, C, HW, chunks)
    if (vec) {
        if (residual) { if (relu) SD_AA(true, true, true); else SD_AA(true, true, false); }
        else { if (relu) SD_AA(true, false, true); else SD_AA(true, false, false); }
    } else {
        if (residual) { if (relu) SD_AA(false, true, true); else SD_AA(false, true, false); }
        else { if (relu) SD_AA(false, false, true); else SD_AA(false, false, false); }
    }
#undef SD_AA
    return (int)hipGetLastError();
}

}  // extern "C"
