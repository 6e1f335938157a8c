// patch_embed.hip -- the overlapping patch-embedding convolutions of the MiT encoders as (window gather) + (token GEMM).
//
//   reference: mix_transformer.py:185-215 (OverlapPatchEmbed: nn.Conv2d(k = 7, s = 4, p = 3) for stage 1, (3, 2, 1) for stages 2-4, then
//   flatten(2).transpose(1, 2) and LayerNorm).
// Why not MIOpen: (1) its filter-gradient kernels for these shapes accumulate with float atomics, so two runs of the SAME step differ in the
// last bits (tests/test_graph_gpu.py: 157 of 191 tensors), and the reference's only launch mode, `--deterministic`
// (tools/dist_train.sh:8), makes it fall back to `naive_conv_*` kernels: 145 ms of a 112 ms-per-step config-2 run
// (profiles/r06_det_step_kernels_before.txt) against 9.7 ms without the switch; (2) its output is an NCHW tensor that the encoder
// immediately re-reads as tokens.  Here the windows are gathered ONCE into a token-major matrix
//   col [B * Ho * Wo][Kp],   column (ky * k + kx) * Cin + ci  (the order of the filter's channels-last storage; Kp = K rounded up to 8, zero-filled)
// and the three products run on the token GEMMs the encoders already use (forward / input gradient: csrc/token_gemm.hip or the library;
// filter gradient: the backward's grouped launch of csrc/wgrad_tn.hip -- fixed-order reductions, no atomics), so the projection's output IS the
// token map and the whole step is run-to-run bit-identical without any switch.  The input gradient is the transposed gather (col2im): every
// input element sums the <= ceil(k/s)^2 windows that cover it, in a fixed order.
// Both kernels are pure data movement (HBM-bound): 16 bytes per lane, consecutive lanes on consecutive 16-byte chunks of a col row.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sd_common.h"

namespace sd {
namespace {

template <typename T> struct Chunk;                       // EPC elements = 16 bytes
template <> struct Chunk<float> { static constexpr int EPC = 4; typedef float4 vec; };
template <> struct Chunk<bf16_t> { static constexpr int EPC = 8; typedef uint4 vec; };

// VEC: the channel axis is the fastest one (sc == 1), Cin % EPC == 0 and every stride / the base are 16-byte multiples: a chunk is one aligned
// 16-byte load.  Otherwise (the NCHW image of stage 1: Cin = 3) EPC scalar loads.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void im2col_tokens(const T *__restrict__ x, T *__restrict__ col, long chunks, int cpr, int H, int W, int Cin, long sb,
                                                      long sy, long sx, long sc, int k, int s, int p, int Ho, int Wo, int K) {
    constexpr int EPC = Chunk<T>::EPC;
    typedef typename Chunk<T>::vec vec;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= chunks) return;
    const long t = idx / cpr;
    const int e0 = (int)(idx - t * cpr) * EPC;
    const int ox = (int)(t % Wo);
    const long t2 = t / Wo;
    const int oy = (int)(t2 % Ho);
    const long b = t2 / Ho;
    const int kc = k * Cin;
    vec out;
    if (VEC) {
        const int ky = e0 / kc, r = e0 - ky * kc, kx = r / Cin, ci = r - kx * Cin;
        const int iy = oy * s - p + ky, ix = ox * s - p + kx;
        const bool in = e0 < K && iy >= 0 && iy < H && ix >= 0 && ix < W;
        const T *src = x + b * sb + (long)(in ? iy : 0) * sy + (long)(in ? ix : 0) * sx + ci;
        out = *reinterpret_cast<const vec *>(src);
        if (!in) {
            if constexpr (sizeof(T) == 4) out = make_float4(0.f, 0.f, 0.f, 0.f);
            else out = make_uint4(0u, 0u, 0u, 0u);
        }
    } else {
        alignas(16) T v[EPC];
        // (ky, kx, ci) of the chunk's first column by division, the following ones by carry: runtime integer divisions cost ~40 instructions each
        int ky = e0 / kc, r = e0 - ky * kc, kx = r / Cin, ci = r - kx * Cin;
        const T *xb = x + b * sb;
        const int y0 = oy * s - p, x0 = ox * s - p;
#pragma unroll
        for (int i = 0; i < EPC; ++i) {
            const int iy = y0 + ky, ix = x0 + kx;
            const bool in = e0 + i < K && iy >= 0 && iy < H && ix >= 0 && ix < W;
            T val = xb[in ? (long)iy * sy + (long)ix * sx + (long)ci * sc : 0];
            if (!in) {
                if constexpr (sizeof(T) == 4) val = 0.f;
                else val.bits = 0;
            }
            v[i] = val;
            if (++ci == Cin) {
                ci = 0;
                if (++kx == k) kx = 0, ++ky;
            }
        }
        out = *reinterpret_cast<const vec *>(v);
    }
    *reinterpret_cast<vec *>(col + idx * EPC) = out;
}

// dx [B][H][W][Cin] (contiguous, channels fastest) = sum over the windows (oy, ky), (ox, kx) with oy * s - p + ky == iy, ox * s - p + kx == ix of
// dcol[(b, oy, ox)][(ky * k + kx) * Cin + ci]; fp32 accumulation in the fixed (ky, kx) order; one thread per 16-byte chunk of dx.
template <typename T>
__global__ __launch_bounds__(256) void col2im_tokens(const T *__restrict__ dcol, T *__restrict__ dx, long chunks, int H, int W, int Cin, int k, int s,
                                                      int p, int Ho, int Wo, int Kp) {
    constexpr int EPC = Chunk<T>::EPC;
    typedef typename Chunk<T>::vec vec;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= chunks) return;
    const int cpp = Cin / EPC;                            // chunks per pixel
    const long pix = idx / cpp;
    const int ci = (int)(idx - pix * cpp) * EPC;
    const int ix = (int)(pix % W);
    const long p2 = pix / W;
    const int iy = (int)(p2 % H);
    const long b = p2 / H;
    float acc[EPC];
#pragma unroll
    for (int i = 0; i < EPC; ++i) acc[i] = 0.f;
    for (int ky = 0; ky < k; ++ky) {
        const int ny = iy + p - ky;
        if (ny < 0 || ny % s != 0) continue;
        const int oy = ny / s;
        if (oy >= Ho) continue;
        for (int kx = 0; kx < k; ++kx) {
            const int nx = ix + p - kx;
            if (nx < 0 || nx % s != 0) continue;
            const int ox = nx / s;
            if (ox >= Wo) continue;
            const T *src = dcol + ((b * Ho + oy) * Wo + ox) * (long)Kp + (ky * k + kx) * Cin + ci;
            const vec v = *reinterpret_cast<const vec *>(src);
            if constexpr (sizeof(T) == 4) {
                acc[0] += v.x, acc[1] += v.y, acc[2] += v.z, acc[3] += v.w;
            } else {
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[2 * i] += __uint_as_float(w[i] << 16);
                    acc[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u);
                }
            }
        }
    }
    vec out;
    if constexpr (sizeof(T) == 4) {
        out = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = (uint32_t)f32_to_bf16(acc[2 * i]) | ((uint32_t)f32_to_bf16(acc[2 * i + 1]) << 16);
        out = make_uint4(w[0], w[1], w[2], w[3]);
    }
    *reinterpret_cast<vec *>(dx + idx * EPC) = out;
}

bool geometry_ok(int B, int H, int W, int Cin, int k, int s, int p, int Ho, int Wo, int Kp) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || k <= 0 || s <= 0 || p < 0 || Ho <= 0 || Wo <= 0) return false;
    if (Ho != (H + 2 * p - k) / s + 1 || Wo != (W + 2 * p - k) / s + 1) return false;
    const long K = (long)k * k * Cin;
    return Kp >= K && Kp % 8 == 0 && Kp < K + 8 && (long)B * Ho * Wo * (Kp / 4) / 256 < 0x7fffffffL;
}

}  // namespace
}  // namespace sd

extern "C" {

int sd_im2col_tokens(const void *x, void *col, int dtype, int B, int H, int W, int Cin, long sb, long sc, long sy, long sx, int k, int stride, int pad,
                     int Ho, int Wo, int Kp, void *stream) {
    if (!x || !col) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (!sd::geometry_ok(B, H, W, Cin, k, stride, pad, Ho, Wo, Kp)) return SD_E_SHAPE;
    if (reinterpret_cast<uintptr_t>(col) & 15) return SD_E_ALIGN;
    const int es = dtype == SD_F32 ? 4 : 2, epc = 16 / es;
    const int cpr = Kp / epc;
    const long chunks = (long)B * Ho * Wo * cpr;
    const bool vec = sc == 1 && Cin % epc == 0 && sb % epc == 0 && sy % epc == 0 && sx % epc == 0 && !(reinterpret_cast<uintptr_t>(x) & 15);
    const dim3 g((unsigned)((chunks + 255) / 256)), blk(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int K = k * k * Cin;
#define SD_I2C(T, V) hipLaunchKernelGGL((sd::im2col_tokens<T, V>), g, blk, 0, st, (const T *)x, (T *)col, chunks, cpr, H, W, Cin, sb, sy, sx, sc, k, stride, pad, Ho, Wo, K)
    if (dtype == SD_F32) { if (vec) SD_I2C(float, true); else SD_I2C(float, false); }
    else { if (vec) SD_I2C(sd::bf16_t, true); else SD_I2C(sd::bf16_t, false); }
#undef SD_I2C
    return (int)hipGetLastError();
}

int sd_col2im_tokens(const void *dcol, void *dx, int dtype, int B, int H, int W, int Cin, int k, int stride, int pad, int Ho, int Wo, int Kp, void *stream) {
    if (!dcol || !dx) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (!sd::geometry_ok(B, H, W, Cin, k, stride, pad, Ho, Wo, Kp)) return SD_E_SHAPE;
    const int es = dtype == SD_F32 ? 4 : 2, epc = 16 / es;
    if (Cin % epc != 0) return SD_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(dcol) | reinterpret_cast<uintptr_t>(dx)) & 15) return SD_E_ALIGN;
    const long chunks = (long)B * H * W * (Cin / epc);
    if ((chunks + 255) / 256 > 0x7fffffffL) return SD_E_SHAPE;
    const dim3 g((unsigned)((chunks + 255) / 256)), blk(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32)
        hipLaunchKernelGGL(sd::col2im_tokens<float>, g, blk, 0, st, (const float *)dcol, (float *)dx, chunks, H, W, Cin, k, stride, pad, Ho, Wo, Kp);
    else
        hipLaunchKernelGGL(sd::col2im_tokens<sd::bf16_t>, g, blk, 0, st, (const sd::bf16_t *)dcol, (sd::bf16_t *)dx, chunks, H, W, Cin, k, stride, pad, Ho,
                           Wo, Kp);
    return (int)hipGetLastError();
}

}  // extern "C"
