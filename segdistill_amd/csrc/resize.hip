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
// Integer factors 2 / 4 / 8 with align_corners = False (every use inside the BASELINE networks but the PPM's 3- and 6-bin branches) take the
// gap-wise kernels further down: forward 64-69 % of HBM with streaming stores, ONE backward kernel without the workspace at 46-79 %
// (profiles/r04_kernels_resize.txt; the generic pair: 23-36 % and 18-28 %).
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

// ---- integer up-sampling factors F in {2, 4, 8}, align_corners = False (round 4) -------------------------------------------
// Every use of this file inside the BASELINE networks is one of these (UPerHead's top-down path and level fusion: x2, x4, x8; the PPM
// branches with 1 / 2 bins): the output rows F j - F/2 .. F j + F/2 - 1 lie between tap rows j-1 and j ("gap j", j = 0..h, the first and
// last being half gaps whose taps clamp), with the tap weight (q + 1/2) / F -- exact in fp32 for a power of two, so these kernels and the
// generic ones above (ATen's index arithmetic) agree on every weight.  The generic forward spends 16 scalar gathers and four float -> int
// source-index evaluations per 16-byte store; the generic backward goes through an fp32 workspace of planes x h x W.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
// streaming (non-temporal) stores: the up-sampled map is 4 - 64x the input and is not read again by this kernel
template <typename T, int F>
__device__ __forceinline__ void store_row(T *dst, const float (&o)[F]) {
    if constexpr (sizeof(T) == 4) {
        if constexpr (F == 2) {
            __builtin_nontemporal_store(f32x2_t{o[0], o[1]}, reinterpret_cast<f32x2_t *>(dst));
        } else {
#pragma unroll
            for (int e = 0; e < F; e += 4) __builtin_nontemporal_store(f32x4_t{o[e], o[e + 1], o[e + 2], o[e + 3]}, reinterpret_cast<f32x4_t *>(dst + e));
        }
    } else {
        unsigned v[F / 2];
#pragma unroll
        for (int e = 0; e < F / 2; ++e) v[e] = (unsigned)f32_to_bf16(o[2 * e]) | ((unsigned)f32_to_bf16(o[2 * e + 1]) << 16);
        if constexpr (F == 2) __builtin_nontemporal_store(v[0], reinterpret_cast<unsigned *>(dst));
        else if constexpr (F == 4) __builtin_nontemporal_store(u32x2_t{v[0], v[1]}, reinterpret_cast<u32x2_t *>(dst));
        else __builtin_nontemporal_store(u32x4_t{v[0], v[1], v[2], v[3]}, reinterpret_cast<u32x4_t *>(dst));
    }
}

template <typename T> __device__ __forceinline__ float2 ld2(const T *p);
// (plain loads: neighbouring windows overlap and the second reader should hit L1 -- non-temporal loads measured 10-90 % slower here)
template <> __device__ __forceinline__ float2 ld2<float>(const float *p) { return *reinterpret_cast<const float2 *>(p); }
template <> __device__ __forceinline__ float2 ld2<bf16_t>(const bf16_t *p) {
    const unsigned v = *reinterpret_cast<const unsigned *>(p);
    return make_float2(__uint_as_float(v << 16), __uint_as_float(v & 0xffff0000u));
}
template <typename T> __device__ __forceinline__ float4 ld4(const T *p);
template <> __device__ __forceinline__ float4 ld4<float>(const float *p) { return *reinterpret_cast<const float4 *>(p); }
template <> __device__ __forceinline__ float4 ld4<bf16_t>(const bf16_t *p) {
    const uint2 v = *reinterpret_cast<const uint2 *>(p);
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
}

// forward: one thread per (plane, gap j, FOUR consecutive output columns): lane l of a row writes bytes 16 l .. 16 l + 15 of each of the gap's F
// output rows, so every store instruction covers whole contiguous lines -- which is what lets the stores be non-temporal (with 32 bytes per
// lane in two instructions the same kernel ran 2x SLOWER with streaming stores, and 1.5x slower than this form without them).  The four
// columns lie in one or two x-gaps: taps gx0 - 1 .. gx0 - 1 + NT - 1 of the two tap rows (NT = 4 / 3 / 2 for F = 2 / 4 / 8).
template <typename T, int F>
__global__ __launch_bounds__(256) void resize_up_fwd(const T *__restrict__ in, T *__restrict__ out, int h, int w) {
    constexpr int LF = F == 2 ? 1 : F == 4 ? 2 : 3, NT = F == 2 ? 4 : F == 4 ? 3 : 2;
    const int wq = F * w / 4;
    const int item = (int)blockIdx.y * 256 + (int)threadIdx.x;      // within the plane: 32-bit index arithmetic
    if (item >= (h + 1) * wq) return;
    const int j = item / wq, c = item - j * wq;
    const long p = blockIdx.x;
    const T *src = in + p * (long)h * w;
    const T *r0 = src + (long)max(j - 1, 0) * w, *r1 = src + (long)min(j, h - 1) * w;
    const int gx0 = (4 * c + F / 2) >> LF;                           // x-gap of the first column; taps gx0 - 1, gx0, ...
    float t0[NT], t1[NT];
#pragma unroll
    for (int e = 0; e < NT; ++e) {
        const int x = min(max(gx0 - 1 + e, 0), w - 1);
        t0[e] = VecIO<T>::load1(r0 + x);
        t1[e] = VecIO<T>::load1(r1 + x);
    }
    float top[4], bot[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        constexpr int kRel[3][4] = {{0, 1, 1, 2}, {0, 0, 1, 1}, {0, 0, 0, 0}};      // x-gap of column e relative to gx0, per F
        const int rel = kRel[LF - 1][e];
        const int q = (4 * c + e + F / 2) & (F - 1);
        const float l1 = (gx0 + rel) > 0 ? (q + 0.5f) * (1.f / F) : 0.f, l0 = 1.f - l1;      // gap 0: the source index clamps to 0 (weights 1, 0)
        top[e] = l0 * t0[rel] + l1 * t0[rel + 1];
        bot[e] = l0 * t1[rel] + l1 * t1[rel + 1];
    }
    const int H = F * h, W = F * w;
    T *dst = out + p * (long)H * W + 4 * c;
#pragma unroll
    for (int q = 0; q < F; ++q) {
        const int Y = F * j - F / 2 + q;
        if (Y < 0 || Y >= H) continue;
        const float l1 = j > 0 ? (q + 0.5f) / F : 0.f, l0 = 1.f - l1;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = l0 * top[e] + l1 * bot[e];
        store_row<T, 4>(dst + (long)Y * W, o);
    }
}

// weight of output index F k - F/2 + t (t = 0 .. 2F-1: gap k, then gap k + 1) on tap k of n taps; the edge taps also collect the clamped halves
template <int F>
__device__ __forceinline__ float up_weight(int t, int k, int n) {
    if (t < F) {
        const float l = (t + 0.5f) / F;
        return k == 0 ? 1.f : l;                         // gap 0: both taps are tap 0
    }
    const float l = (t - F + 0.5f) / F;
    return k == n - 1 ? 1.f : 1.f - l;                   // gap n: both taps are tap n - 1
}

// sum over x of one output-gradient row onto KT taps starting at k0: the window F k0 - F/2 .. F (k0 + KT) + F/2 - 1 comes straight from global
// memory in the widest loads its alignment allows (neighbouring threads overlap by F values: L1 hits); out-of-row halves read as 0
template <typename T, int F, int KT>
__device__ __forceinline__ void fold_x(const T *__restrict__ row, int k0, int w, float (&acc)[KT]) {
    constexpr int LW = F * KT + F;
    const int W = F * w, X0 = F * k0 - F / 2;
    float v[LW];
    if constexpr (F == 2) {                              // KT == 2: X0 = 4 m - 1
        v[0] = X0 >= 0 ? VecIO<T>::load1(row + X0) : 0.f;
        const float4 c = ld4<T>(row + X0 + 1);
        v[1] = c.x, v[2] = c.y, v[3] = c.z, v[4] = c.w;
        v[5] = X0 + 5 < W ? VecIO<T>::load1(row + X0 + 5) : 0.f;
    } else if constexpr (F == 4) {                       // X0 = 4 k - 2
        const float2 a = X0 >= 0 ? ld2<T>(row + X0) : make_float2(0.f, 0.f);
        const float4 b = ld4<T>(row + X0 + 2);
        const float2 c = X0 + 6 < W ? ld2<T>(row + X0 + 6) : make_float2(0.f, 0.f);
        v[0] = a.x, v[1] = a.y, v[2] = b.x, v[3] = b.y, v[4] = b.z, v[5] = b.w, v[6] = c.x, v[7] = c.y;
    } else {                                             // F == 8: X0 = 8 k - 4
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 a = X0 >= 0 ? ld4<T>(row + X0) : z, b = ld4<T>(row + X0 + 4), c = ld4<T>(row + X0 + 8), d = X0 + 12 < W ? ld4<T>(row + X0 + 12) : z;
        v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
        v[8] = c.x, v[9] = c.y, v[10] = c.z, v[11] = c.w, v[12] = d.x, v[13] = d.y, v[14] = d.z, v[15] = d.w;
    }
#pragma unroll
    for (int c = 0; c < KT; ++c) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 2 * F; ++t) s = fmaf(up_weight<F>(t, k0 + c, w), v[F * c + t], s);
        acc[c] = s;
    }
}

// backward, ONE kernel, no workspace: a workgroup owns R input rows of one plane (the whole plane when its x-folded rows fit 48 KB of LDS).  Pass 1
// folds every output-gradient row that touches them along x into tmp[row][w] in LDS (fold_x: global -> registers, consecutive lanes write
// consecutive words); pass 2 folds 2F of those rows along y into each input row, written contiguously.  Adjacent bands re-read F rows.
template <typename T, int F, int KT>
__global__ __launch_bounds__(256) void resize_up_bwd(const T *__restrict__ dout, T *__restrict__ din, int h, int w, int R, int bands) {
    extern __shared__ __attribute__((aligned(16))) float up_lds[];
    const int W = F * w, H = F * h, wk = w / KT;
    const long p = blockIdx.x / bands;
    const int j0 = (int)(blockIdx.x % bands) * R, j1 = min(j0 + R, h);
    const int Ylo = max(0, F * j0 - F / 2), Yhi = min(H, F * (j1 - 1) + 3 * F / 2 + (j1 == h ? F : 0));
    const int nrows = Yhi - Ylo;
    const T *g = dout + (p * H + Ylo) * (long)W;
    for (int it = threadIdx.x; it < nrows * wk; it += 256) {
        const int r = it / wk, m = it - r * wk;
        float acc[KT];
        fold_x<T, F, KT>(g + (long)r * W, KT * m, w, acc);
#pragma unroll
        for (int c = 0; c < KT; ++c) up_lds[(size_t)r * w + KT * m + c] = acc[c];
    }
    __syncthreads();
    for (int it = threadIdx.x; it < (j1 - j0) * w; it += 256) {
        const int jj = it / w, k = it - jj * w, j = j0 + jj;
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 2 * F; ++t) {
            const int Y = F * j - F / 2 + t;
            if (Y >= 0 && Y < H) acc = fmaf(up_weight<F>(t, j, h), up_lds[(size_t)(Y - Ylo) * w + k], acc);
        }
        VecIO<T>::store1(din + (p * h + j) * (long)w + k, acc);
    }
}

// integer factor of the fast path (0: not covered), and rows per workgroup of its backward for ~40 KB of LDS
int up_factor(int h, int w, int H, int W, int align, int dtype, const void *a, const void *b) {
    if (align || h <= 0 || w <= 0 || H % h || W % w || H / h != W / w) return 0;
    const int F = H / h;
    if (F != 2 && F != 4 && F != 8) return 0;
    const size_t es = dtype == SD_F32 ? 4 : 2;
    if ((W * es) % 16 || ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15)) return 0;
    return F;
}
int up_band_rows(int h, int w, int F) {
    const long budget = 48 * 1024 / ((long)w * 4);       // x-folded rows that fit
    if ((long)F * h <= budget) return h;                   // the whole plane: nothing is read twice
    const long R = (budget - F - F / 2) / F;
    return R < 1 ? 0 : (int)(R > h ? h : R);
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
    if (const int F = sd::up_factor(h, w, H, W, align_corners, dtype, in, out)) {
        if (planes <= 0x7fffffffL && ((long)(h + 1) * (W / 4) + 255) / 256 <= 65535) {
            const dim3 b(256);
#define SD_UPF(T, FF) hipLaunchKernelGGL((sd::resize_up_fwd<T, FF>), dim3((unsigned)planes, (unsigned)(((h + 1) * (W / 4) + 255) / 256)), b, 0, st, (const T *)in, (T *)out, h, w)
            if (dtype == SD_F32) { if (F == 2) SD_UPF(float, 2); else if (F == 4) SD_UPF(float, 4); else SD_UPF(float, 8); }
            else { if (F == 2) SD_UPF(sd::bf16_t, 2); else if (F == 4) SD_UPF(sd::bf16_t, 4); else SD_UPF(sd::bf16_t, 8); }
#undef SD_UPF
            return (int)hipGetLastError();
        }
    }
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
    if (const int F = sd::up_factor(h, w, H, W, align_corners, dtype, dout, din)) {
        const int R = sd::up_band_rows(h, w, F);
        const long bands = R ? (h + R - 1) / R : 0;
        if (R && planes * bands <= 0x7fffffffL) {
            const size_t lds = (size_t)((R + 1) * F < H ? (R + 1) * F : H) * (size_t)w * sizeof(float);      // <= 48 KB (up_band_rows)
            const dim3 g((unsigned)(planes * bands)), b(256);
#define SD_UPB(T, FF, KK) hipLaunchKernelGGL((sd::resize_up_bwd<T, FF, KK>), g, b, lds, st, (const T *)dout, (T *)din, h, w, R, (int)bands)
            if (dtype == SD_F32) { if (F == 2) SD_UPB(float, 2, 2); else if (F == 4) SD_UPB(float, 4, 1); else SD_UPB(float, 8, 1); }
            else { if (F == 2) SD_UPB(sd::bf16_t, 2, 2); else if (F == 4) SD_UPB(sd::bf16_t, 4, 1); else SD_UPB(sd::bf16_t, 8, 1); }
#undef SD_UPB
            return (int)hipGetLastError();
        }
    }
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
