// resize.hip -- bilinear resize of NCHW maps, forward and backward (gfx950).
//
// Replaces F.interpolate(x, size, mode='bilinear', align_corners=...) as the reference calls it through mmseg/ops/wrappers.py::resize
// (:6-28) inside the networks: the pyramid-pooling branches of PSPHead (psp_head.py:52-58), the top-down path and the level fusion of
// UPerHead (uper_head.py:101-121), the `resize_concat` input transform (decode_head.py:130-139), and -- for the sizes the fused
// up-sample kernels do not take -- KLDLoss.resize (distillation/losses.py:25-33).  ATen's NCHW kernels run these at ~3 % of HBM speed on
// MI355X (profiles/r03_train_step_kernels_cfg4.txt: upsample_bilinear2d_out_frame 1.3 ms for a 268 MB map, its atomic backward 1.5 ms:
// 27 of the 99 ms of a config-4 step).
//   forward : one thread per 4 consecutive output columns of one plane (16-byte store); the four taps of an output are gathers that hit
//             L1 / L2 (an input row is reused by ~scale output rows).
//   backward: GATHER form, deterministic, no atomics, SEPARABLE (rows, then columns, through an fp32 workspace of planes x h x W): every
//             sum runs over the conservative candidate range of outputs and re-evaluates the forward's own index arithmetic, so forward and
//             backward agree on every tap by construction.
// Index arithmetic as ATen's area_pixel_compute_source_index: align_corners ? scale * dst : max(0, scale * (dst + 0.5) - 0.5) with
// scale = align_corners ? (in - 1) / (out - 1) : in / out in fp32.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cgd_device.h"

namespace sd {
namespace {

struct Tap {
    int i0, i1;
    float l0, l1;
};
__device__ __forceinline__ Tap tap_of(int dst, float scale, int n_in, bool align) {
    float src = align ? scale * dst : fmaxf(scale * (dst + 0.5f) - 0.5f, 0.f);
    int i0 = (int)src;
    if (i0 > n_in - 1) i0 = n_in - 1;
    Tap t;
    t.i0 = i0;
    t.i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    t.l1 = src - (float)i0;
    t.l0 = 1.f - t.l1;
    return t;
}

template <typename T>
__global__ __launch_bounds__(256) void resize_bilinear_fwd(const T *__restrict__ in, T *__restrict__ out, long planes, int h, int w, int H, int W,
                                                            float sy, float sx, int align) {
    const int wq = (W + 3) / 4;
    const long item = (long)blockIdx.x * 256 + threadIdx.x;        // (plane, Y, column quad)
    if (item >= planes * H * wq) return;
    const int q = (int)(item % wq);
    const long rest = item / wq;
    const int Y = (int)(rest % H);
    const long p = rest / H;
    const T *src = in + p * (long)h * w;
    const Tap ty = tap_of(Y, sy, h, align);
    const T *r0 = src + (long)ty.i0 * w, *r1 = src + (long)ty.i1 * w;
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int X = min(4 * q + e, W - 1);
        const Tap tx = tap_of(X, sx, w, align);
        const float a = VecIO<T>::load1(r0 + tx.i0), b = VecIO<T>::load1(r0 + tx.i1), c = VecIO<T>::load1(r1 + tx.i0), d = VecIO<T>::load1(r1 + tx.i1);
        o[e] = ty.l0 * (tx.l0 * a + tx.l1 * b) + ty.l1 * (tx.l0 * c + tx.l1 * d);
    }
    T *dst = out + (p * H + Y) * (long)W + 4 * q;
    if (4 * q + 3 < W && (W & 3) == 0) {
        if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
            uint2 v;
            v.x = (unsigned)f32_to_bf16(o[0]) | ((unsigned)f32_to_bf16(o[1]) << 16);
            v.y = (unsigned)f32_to_bf16(o[2]) | ((unsigned)f32_to_bf16(o[3]) << 16);
            *reinterpret_cast<uint2 *>(dst) = v;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * q + e < W) VecIO<T>::store1(dst + e, o[e]);
    }
}

// candidate output range [lo, hi] whose taps may touch input index i (conservative: widened by 1 on both sides, clamped; every candidate is
// re-tested with the forward's own arithmetic, so slack costs time, never correctness)
__device__ __forceinline__ void out_range(int i, float scale, int n_out, bool align, int &lo, int &hi) {
    const float inv = scale > 0.f ? 1.f / scale : 0.f;
    float a, b;
    if (align) { a = (i - 1) * inv; b = (i + 1) * inv; }
    else { a = (i - 1 + 0.5f) * inv - 0.5f; b = (i + 1 + 0.5f) * inv - 0.5f; }
    lo = max(0, (int)floorf(a) - 1);
    hi = min(n_out - 1, (int)ceilf(b) + 1);
    if (scale <= 0.f) { lo = 0; hi = n_out - 1; }
}

// Separable gather backward (the interpolation weights factor into wy * wx):
//   pass 1  tmp[p][y][X] = sum_Y wy(Y, y) dOut[p][Y][X]      one thread per (p, y, X): coalesced along X, ~2 scale rows each
//   pass 2  dIn[p][y][x] = sum_X wx(X, x) tmp[p][y][X]       one thread per input pixel, ~2 scale columns of ONE tmp row each
// instead of one thread per input pixel scanning (2 scale)^2 outputs with the weights re-derived per output (0.51 ms for [8,512,64,64] <- 128^2,
// 8 % of HBM speed).  tmp is fp32 whatever the storage type.
template <typename T>
__global__ __launch_bounds__(256) void resize_bwd_rows(const T *__restrict__ dout, float *__restrict__ tmp, long planes, int h, int H, int W, float sy,
                                                        int align) {
    // one thread per (plane, y, FOUR consecutive X): the row weight is shared by the four columns and dOut is read as 16-byte (fp32) / 8-byte
    // (bf16) vectors when rows are aligned (W % 4 == 0; the launcher checks the base pointer)
    const int wq = (W + 3) / 4;
    const long item = (long)blockIdx.x * 256 + threadIdx.x;
    if (item >= planes * h * wq) return;
    const int q = (int)(item % wq);
    const long rest = item / wq;
    const int y = (int)(rest % h);
    const long p = rest / h;
    int Y0, Y1;
    out_range(y, sy, H, align, Y0, Y1);
    const T *g = dout + p * (long)H * W + 4 * q;
    const bool vec = (W & 3) == 0;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int Y = Y0; Y <= Y1; ++Y) {
        const Tap ty = tap_of(Y, sy, h, align);
        const float wy = (ty.i0 == y ? ty.l0 : 0.f) + (ty.i1 == y ? ty.l1 : 0.f);
        if (wy == 0.f) continue;
        const T *r = g + (long)Y * W;
        float v0, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        if (vec) {
            if constexpr (sizeof(T) == 4) {
                const float4 v = *reinterpret_cast<const float4 *>(r);
                v0 = v.x, v1 = v.y, v2 = v.z, v3 = v.w;
            } else {
                const uint2 v = *reinterpret_cast<const uint2 *>(r);
                v0 = __uint_as_float(v.x << 16), v1 = __uint_as_float(v.x & 0xffff0000u), v2 = __uint_as_float(v.y << 16), v3 = __uint_as_float(v.y & 0xffff0000u);
            }
        } else {
            v0 = VecIO<T>::load1(r);
            if (4 * q + 1 < W) v1 = VecIO<T>::load1(r + 1);
            if (4 * q + 2 < W) v2 = VecIO<T>::load1(r + 2);
            if (4 * q + 3 < W) v3 = VecIO<T>::load1(r + 3);
        }
        a0 = fmaf(wy, v0, a0), a1 = fmaf(wy, v1, a1), a2 = fmaf(wy, v2, a2), a3 = fmaf(wy, v3, a3);
    }
    float *o = tmp + (p * h + y) * (long)W + 4 * q;
    if (vec) {
        *reinterpret_cast<float4 *>(o) = make_float4(a0, a1, a2, a3);
    } else {
        o[0] = a0;
        if (4 * q + 1 < W) o[1] = a1;
        if (4 * q + 2 < W) o[2] = a2;
        if (4 * q + 3 < W) o[3] = a3;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void resize_bwd_cols(const float *__restrict__ tmp, T *__restrict__ din, long planes, int h, int w, int W, float sx,
                                                        int align) {
    const long item = (long)blockIdx.x * 256 + threadIdx.x;        // (plane, y, x) of the INPUT map
    if (item >= planes * h * w) return;
    const int x = (int)(item % w);
    const long row = item / w;                                      // (plane, y)
    int X0, X1;
    out_range(x, sx, W, align, X0, X1);
    const float *r = tmp + row * W;
    float acc = 0.f;
    for (int X = X0; X <= X1; ++X) {
        const Tap tx = tap_of(X, sx, w, align);
        const float wx = (tx.i0 == x ? tx.l0 : 0.f) + (tx.i1 == x ? tx.l1 : 0.f);
        if (wx != 0.f) acc = fmaf(wx, r[X], acc);
    }
    VecIO<T>::store1(din + item, acc);
}

float scale_of(int n_in, int n_out, int align) {
    if (align) return n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f;
    return (float)n_in / (float)n_out;
}

}  // namespace
}  // namespace sd

extern "C" {

int sd_resize_bilinear_fwd(const void *in, void *out, int dtype, long planes, int h, int w, int H, int W, int align_corners, void *stream) {
    if (!in || !out) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (planes <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return SD_E_SHAPE;
    const long items = planes * H * ((W + 3) / 4);
    if ((items + 255) / 256 > 0x7fffffffL) return SD_E_SHAPE;
    if ((W & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & (dtype == SD_F32 ? 15 : 7))) return SD_E_ALIGN;
    const float sy = sd::scale_of(h, H, align_corners), sx = sd::scale_of(w, W, align_corners);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const unsigned grid = (unsigned)((items + 255) / 256);
    if (dtype == SD_F32)
        hipLaunchKernelGGL(sd::resize_bilinear_fwd<float>, dim3(grid), dim3(256), 0, st, (const float *)in, (float *)out, planes, h, w, H, W, sy, sx,
                           align_corners ? 1 : 0);
    else
        hipLaunchKernelGGL(sd::resize_bilinear_fwd<sd::bf16_t>, dim3(grid), dim3(256), 0, st, (const sd::bf16_t *)in, (sd::bf16_t *)out, planes, h, w,
                           H, W, sy, sx, align_corners ? 1 : 0);
    return (int)hipGetLastError();
}

size_t sd_resize_bilinear_bwd_workspace_bytes(long planes, int h, int W) {
    if (planes <= 0 || h <= 0 || W <= 0) return 0;
    return (size_t)planes * h * W * sizeof(float) + 16;
}

int sd_resize_bilinear_bwd(const void *dout, void *din, int dtype, long planes, int h, int w, int H, int W, int align_corners, void *workspace,
                           size_t workspace_bytes, void *stream) {
    if (!dout || !din || !workspace) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (planes <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return SD_E_SHAPE;
    if (workspace_bytes < sd_resize_bilinear_bwd_workspace_bytes(planes, h, W) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return SD_E_WORKSPACE;
    const long items1 = planes * h * ((W + 3) / 4), items2 = planes * h * w;
    if ((W & 3) == 0 && (reinterpret_cast<uintptr_t>(dout) & (dtype == SD_F32 ? 15 : 7))) return SD_E_ALIGN;
    if ((items1 + 255) / 256 > 0x7fffffffL) return SD_E_SHAPE;
    const float sy = sd::scale_of(h, H, align_corners), sx = sd::scale_of(w, W, align_corners);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *tmp = static_cast<float *>(workspace);
    const unsigned g1 = (unsigned)((items1 + 255) / 256), g2 = (unsigned)((items2 + 255) / 256);
    const int al = align_corners ? 1 : 0;
    if (dtype == SD_F32) {
        hipLaunchKernelGGL(sd::resize_bwd_rows<float>, dim3(g1), dim3(256), 0, st, (const float *)dout, tmp, planes, h, H, W, sy, al);
        hipLaunchKernelGGL(sd::resize_bwd_cols<float>, dim3(g2), dim3(256), 0, st, tmp, (float *)din, planes, h, w, W, sx, al);
    } else {
        hipLaunchKernelGGL(sd::resize_bwd_rows<sd::bf16_t>, dim3(g1), dim3(256), 0, st, (const sd::bf16_t *)dout, tmp, planes, h, H, W, sy, al);
        hipLaunchKernelGGL(sd::resize_bwd_cols<sd::bf16_t>, dim3(g2), dim3(256), 0, st, tmp, (sd::bf16_t *)din, planes, h, w, W, sx, al);
    }
    return (int)hipGetLastError();
}

}  // extern "C"
