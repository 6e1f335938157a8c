// headfuse.hip -- SegFormer all-MLP head: "up-sample three coarse branch maps to 1/4 resolution and add
// them to the fine one" as ONE pass, on token-major ([B, h*w, E] = NHWC) tensors, gfx950.
//
// Context (reference mmseg/models/decode_heads/segformer_head.py:75-98): the head resizes the 1/8, 1/16 and
// 1/32 branch maps to the 1/4 grid, concatenates and applies the 1x1 fuse conv.  segdistill_amd runs the fuse
// conv per branch at native resolution (exact: a 1x1 conv commutes with bilinear interpolation), which leaves
//     y = z1 + up2(z2) + up4(z3) + up8(z4) + bias
// With ATen that is 3 upsample kernels + 4 adds, each a full read+write of a [B,E,128,128] tensor (403 MB for
// the E=768 teachers): ~2.3 ms/step forward and 1.6 ms backward in the round-1 profile.  Here: one read of z1
// and one write of y (the coarse maps are L2-resident), and one gather kernel per branch for the backward
// (transposed interpolation, no atomics).  HBM-bound byte work.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "cgd_device.h"

namespace sd {

namespace {

template <typename T> struct HV;  // 16-byte channel vector
template <> struct HV<float> {
    static constexpr int N = 4;
    typedef float raw_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ void load(const float *p, float (&o)[4]) {
        raw_t v = *reinterpret_cast<const raw_t *>(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    static __device__ __forceinline__ void store(float *p, const float (&o)[4]) {
        raw_t v = {o[0], o[1], o[2], o[3]};
        *reinterpret_cast<raw_t *>(p) = v;
    }
};
template <> struct HV<bf16_t> {
    static constexpr int N = 8;
    typedef unsigned int raw_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ void load(const bf16_t *p, float (&o)[8]) {
        raw_t v = *reinterpret_cast<const raw_t *>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[2 * i] = __uint_as_float(v[i] << 16);
            o[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store(bf16_t *p, const float (&o)[8]) {
        raw_t v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (unsigned)f32_to_bf16(o[2 * i]) | ((unsigned)f32_to_bf16(o[2 * i + 1]) << 16);
        *reinterpret_cast<raw_t *>(p) = v;
    }
};

// bilinear source coordinates of output index o for an integer factor F (align_corners=False, ATen semantics)
// XCD-aware work order for a 1-D grid: workgroups are dealt round-robin to the 8 XCDs (one L2 each) by linear id; giving XCD x a
// CONTIGUOUS range of work items keeps the coarse maps an image band gathers from (and neighbouring output rows) in one L2.
__device__ __forceinline__ size_t xcd_item(unsigned lin, unsigned total) {
    const unsigned q8 = total / 8, r8 = total % 8, xcd = lin % 8;
    return (size_t)xcd * q8 + (xcd < r8 ? xcd : r8) + lin / 8;
}

__device__ __forceinline__ void src_of(int o, int F, int n_in, int &i0, int &i1, float &lam) {
    float s = (o + 0.5f) / F - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i0 = i0 < n_in - 1 ? i0 : n_in - 1;
    i1 = i0 + 1 < n_in ? i0 + 1 : n_in - 1;
    lam = s - i0;
}

template <typename T>
__device__ __forceinline__ void add_branch(float (&acc)[HV<T>::N], const T *__restrict__ z, int b, int Y, int X, int H, int W, int F, int E,
                                           int c) {
    constexpr int N = HV<T>::N;
    const int h = H / F, w = W / F;
    int y0, y1, x0, x1;
    float ly, lx;
    src_of(Y, F, h, y0, y1, ly);
    src_of(X, F, w, x0, x1, lx);
    const T *base = z + (size_t)b * h * w * E + c;
    float v00[N], v01[N], v10[N], v11[N];
    HV<T>::load(base + ((size_t)y0 * w + x0) * E, v00);
    HV<T>::load(base + ((size_t)y0 * w + x1) * E, v01);
    HV<T>::load(base + ((size_t)y1 * w + x0) * E, v10);
    HV<T>::load(base + ((size_t)y1 * w + x1) * E, v11);
    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] += w00 * v00[i] + w01 * v01[i] + w10 * v10[i] + w11 * v11[i];
}

// y[b,Y,X,:] = z1[b,Y,X,:] + up(z2) + up(z3) + up(z4) (+ bias).  grid: ceil(B*H*W*(E/N) / 256)
// Inference epilogue (scale != nullptr): y = max(0, y * scale[c] + shift[c]) -- the eval-mode BatchNorm (folded to an affine
// map per channel) and the ReLU that follow the sum in linear_fuse (segformer_head.py:66-71), for the frozen teacher.
template <typename T>
__global__ __launch_bounds__(256) void upsum_fwd(const T *__restrict__ z1, const T *__restrict__ z2, const T *__restrict__ z3,
                                                  const T *__restrict__ z4, const float *__restrict__ bias, const float *__restrict__ scale,
                                                  const float *__restrict__ shift, int relu, T *__restrict__ y, int B, int H, int W, int E,
                                                  int f2, int f3, int f4) {
    constexpr int N = HV<T>::N;
    const int ev = E / N;
    const size_t t = xcd_item(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (t >= (size_t)B * H * W * ev) return;
    const int c = (int)(t % ev) * N;
    const size_t pix = t / ev;
    const int X = (int)(pix % W);
    const int Y = (int)((pix / W) % H);
    const int b = (int)(pix / ((size_t)W * H));
    float acc[N];
    HV<T>::load(z1 + pix * E + c, acc);
    if (bias) {
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i] += bias[c + i];
    }
    add_branch<T>(acc, z2, b, Y, X, H, W, f2, E, c);
    add_branch<T>(acc, z3, b, Y, X, H, W, f3, E, c);
    add_branch<T>(acc, z4, b, Y, X, H, W, f4, E, c);
    if (scale) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            acc[i] = fmaf(acc[i], scale[c + i], shift[c + i]);
            if (relu) acc[i] = fmaxf(acc[i], 0.f);
        }
    }
    HV<T>::store(y + pix * E + c, acc);
}

// Strip form for the SegFormer geometry (branch factors 2, 4, 8; W % 4 == 0): a thread produces FOUR consecutive output
// pixels of a row for one channel vector.  The generic kernel above issues 12 gathers per output vector and is bound by
// L2 -> CU bandwidth (12x the HBM bytes); the four outputs of a strip share their coarse taps -- 4 / 3 / 2 columns x 2
// rows for the x2 / x4 / x8 branches -- so a strip needs 18 gathers instead of 48.  Border handling: loading the CLAMPED
// column / row indices with the un-clamped interpolation weights equals ATen's edge-clamped bilinear exactly (the clamped
// taps coincide, so their weights add up).
template <typename T, int F, int NC>
__device__ __forceinline__ void add_branch_strip(float (&acc)[4][HV<T>::N], const T *__restrict__ z, int b, int Y, int X0, int H, int W,
                                                 int E, int c) {
    constexpr int N = HV<T>::N;
    const int h = H / F, w = W / F;
    const float sy = (Y + 0.5f) / F - 0.5f;
    const int y0u = (int)floorf(sy);
    const float ly = sy - (float)y0u;
    const int y0 = min(max(y0u, 0), h - 1), y1 = min(max(y0u + 1, 0), h - 1);
    const int base = (int)floorf((X0 + 0.5f) / F - 0.5f);          // un-clamped left tap of the strip's first pixel
    const T *img = z + (size_t)b * h * w * E + c;                  // offsets inside an image are 32-bit (the launcher checks H * W * E < 2^31)
    const T *r0 = img + y0 * w * E, *r1 = img + y1 * w * E;
    float col[NC][N];                                               // vertically interpolated coarse columns base .. base+NC-1
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int x = min(max(base + j, 0), w - 1);
        float a[N], d[N];
        HV<T>::load(r0 + x * E, a);
        HV<T>::load(r1 + x * E, d);
#pragma unroll
        for (int i = 0; i < N; ++i) col[j][i] = fmaf(ly, d[i] - a[i], a[i]);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float sx = (X0 + p + 0.5f) / F - 0.5f;
        const float lx = sx - floorf(sx);
        // offset of this pixel's left tap from `base` is static for X0 % 4 == 0:  F=2: 0,1,1,2   F=4: 0,0,1,1   F=8: 0,0,0,0
        constexpr int kOff2[4] = {0, 1, 1, 2}, kOff4[4] = {0, 0, 1, 1};
        const int o = F == 2 ? kOff2[p] : (F == 4 ? kOff4[p] : 0);
#pragma unroll
        for (int i = 0; i < N; ++i) acc[p][i] += fmaf(lx, col[o + 1][i] - col[o][i], col[o][i]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void upsum_fwd_strip(const T *__restrict__ z1, const T *__restrict__ z2, const T *__restrict__ z3,
                                                        const T *__restrict__ z4, const float *__restrict__ bias,
                                                        const float *__restrict__ scale, const float *__restrict__ shift, int relu,
                                                        T *__restrict__ y, int B, int H, int W, int E) {
    constexpr int N = HV<T>::N;
    const int ev = E / N, sw = W / 4;
    // grid (ceil(sw * ev / 256), B * H): the row index and everything derived from it is wave-uniform (scalar registers), a thread divides
    // once by `ev`, and offsets inside an image are 32-bit.  (The 1-D form decomposed a 64-bit thread index with five runtime divisions:
    // 754 vector instructions per thread, about as much vector-ALU time as the kernel's HBM time.)
    const unsigned item = (unsigned)xcd_item(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const unsigned row = item / gridDim.x;                          // b * H + Y
    const int t = (int)(item - row * gridDim.x) * 256 + threadIdx.x;
    if (t >= sw * ev) return;
    const int sx = t / ev;
    const int c = (t - sx * ev) * N;
    const int X0 = sx * 4;
    const int b = (int)(row / (unsigned)H), Y = (int)(row - (unsigned)b * H);
    const size_t pix = (size_t)row * W + X0;
    float acc[4][N];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        HV<T>::load(z1 + pix * E + c + p * E, acc[p]);
        if (bias) {
#pragma unroll
            for (int i = 0; i < N; ++i) acc[p][i] += bias[c + i];
        }
    }
    add_branch_strip<T, 2, 4>(acc, z2, b, Y, X0, H, W, E, c);
    add_branch_strip<T, 4, 3>(acc, z3, b, Y, X0, H, W, E, c);
    add_branch_strip<T, 8, 2>(acc, z4, b, Y, X0, H, W, E, c);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (scale) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                acc[p][i] = fmaf(acc[p][i], scale[c + i], shift[c + i]);
                if (relu) acc[p][i] = fmaxf(acc[p][i], 0.f);
            }
        }
        HV<T>::store(y + pix * E + c + p * E, acc[p]);
    }
}

// ---- backward of all three coarse branches in two separable passes (the SegFormer geometry F = 2, 4, 8) ----------------------------------
// The per-branch gather below re-reads dy once per branch with a serial loop over the (2F)^2 outputs of a tap (256 iterations at F = 8 on
// 512 workgroups: 108 + 67 + 56 us at config 2).  The interpolation weights are separable, so:
//   rows:  r_F[b, Y, kx, :] = sum_X wx_F(X, kx) dy[b, Y, X, :]   for F = 2, 4, 8 from ONE read of the dy row (staged in LDS, 64 channels per
//          workgroup); fp32 partials in the workspace (dy bytes x 7/8);
//   cols:  dz_F[b, ky, kx, :] = sum_Y wy_F(Y, ky) r_F[b, Y, kx, :]  -- at most 2F coalesced vector reads per output.
__device__ __forceinline__ float tap_weight(int o, int F, int n_in, int k) {       // weight of output index o on input tap k (0 if unused)
    int i0, i1;
    float lam;
    src_of(o, F, n_in, i0, i1, lam);
    return (i0 == k ? 1.f - lam : 0.f) + (i1 == k ? lam : 0.f);
}

template <typename T>
__global__ __launch_bounds__(256) void upsum_bwd_rows(const T *__restrict__ dy, float *__restrict__ r2, float *__restrict__ r4, float *__restrict__ r8,
                                                       int H, int W, int E) {
    constexpr int N = HV<T>::N, CP = 68;                      // 64 channels per workgroup, row pitch 68 floats
    extern __shared__ __attribute__((aligned(16))) float row_tile[];   // [W][CP]
    const size_t row = blockIdx.x;                            // b * H + Y
    const int c0 = blockIdx.y * 64;
    const T *src = dy + row * W * E + c0;
    for (int idx = threadIdx.x; idx < W * (64 / N); idx += 256) {
        const int X = idx / (64 / N), v = (idx % (64 / N)) * N;
        float t[N];
        HV<T>::load(src + (size_t)X * E + v, t);
#pragma unroll
        for (int i = 0; i < N; ++i) row_tile[X * CP + v + i] = t[i];
    }
    __syncthreads();
    auto reduce = [&](int F, float *__restrict__ r) {
        const int w = W / F;
        for (int idx = threadIdx.x; idx < w * 16; idx += 256) {
            const int kx = idx >> 4, c4 = (idx & 15) * 4;
            const int Xa = max(0, F * kx - F / 2), Xb = min(W, F * kx + F + F / 2);
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int X = Xa; X < Xb; ++X) {
                const float wx = tap_weight(X, F, w, kx);
                const float4 v = *reinterpret_cast<const float4 *>(row_tile + X * CP + c4);
                a.x = fmaf(wx, v.x, a.x); a.y = fmaf(wx, v.y, a.y); a.z = fmaf(wx, v.z, a.z); a.w = fmaf(wx, v.w, a.w);
            }
            *reinterpret_cast<float4 *>(r + (row * w + kx) * E + c0 + c4) = a;
        }
    };
    reduce(2, r2);
    reduce(4, r4);
    reduce(8, r8);
}

template <typename T>
__global__ __launch_bounds__(256) void upsum_bwd_cols(const float *__restrict__ r2, const float *__restrict__ r4, const float *__restrict__ r8,
                                                       T *__restrict__ dz2, T *__restrict__ dz3, T *__restrict__ dz4, int B, int H, int W, int E) {
    constexpr int N = HV<T>::N;
    const int ev = E / N;
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n2 = (size_t)B * (H / 2) * (W / 2) * ev, n4 = (size_t)B * (H / 4) * (W / 4) * ev, n8 = (size_t)B * (H / 8) * (W / 8) * ev;
    int F;
    const float *r;
    T *dz;
    if (t < n2) { F = 2; r = r2; dz = dz2; }
    else if (t < n2 + n4) { t -= n2; F = 4; r = r4; dz = dz3; }
    else if (t < n2 + n4 + n8) { t -= n2 + n4; F = 8; r = r8; dz = dz4; }
    else return;
    const int h = H / F, w = W / F;
    const int c = (int)(t % ev) * N;
    const size_t tap = t / ev;
    const int kx = (int)(tap % w), ky = (int)((tap / w) % h), b = (int)(tap / ((size_t)w * h));
    float acc[N];
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = 0.f;
    const int Ya = max(0, F * ky - F / 2), Yb = min(H, F * ky + F + F / 2);
    for (int Y = Ya; Y < Yb; ++Y) {
        const float wy = tap_weight(Y, F, h, ky);
        const float *p = r + (((size_t)b * H + Y) * w + kx) * E + c;
#pragma unroll
        for (int i = 0; i < N; i += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(p + i);
            acc[i] = fmaf(wy, v.x, acc[i]); acc[i + 1] = fmaf(wy, v.y, acc[i + 1]); acc[i + 2] = fmaf(wy, v.z, acc[i + 2]); acc[i + 3] = fmaf(wy, v.w, acc[i + 3]);
        }
    }
    HV<T>::store(dz + tap * E + c, acc);
}

// ---- round 3: the same three gradients in ONE pass, no global partials (W = 128 maps: every SegFormer head at 512 x 512) ---------------------
// The two-pass form above writes the fp32 row partials (7/8 of the dy bytes) and reads them back: 412 MB of traffic for 178 MB of algorithmic
// bytes (58 + 31 us at config 2).  Here a workgroup owns a band of 16 output rows x all 128 columns x 32 channels and produces the COMPLETE
// taps of that band -- 8 / 4 / 2 tap rows of the x2 / x4 / x8 branches -- by walking the 24 output rows their supports touch (rows
// [16 m - 4, 16 m + 20): the halo is re-read, mostly from L2, by the neighbouring band): each dy row is staged in LDS once (double-buffered,
// requested one row ahead), reduced along x for all three factors from LDS with per-thread weight tables, and added into per-tap-row
// accumulators that live in registers (static indices: the row loop is unrolled, only the weights are data).  thread = (4-channel vector
// c4 = tid & 7, column group xg = tid >> 3): taps kx = xg and xg + 32 of the x2 branch, kx = xg of the x4 branch, kx = xg < 16 of the x8 branch.
__global__ __launch_bounds__(256) void upsum_bwd3_band(const float *__restrict__ dy, float *__restrict__ dz2, float *__restrict__ dz3,
                                                        float *__restrict__ dz4, int H, int E) {
    constexpr int W = 128, CP = 36, R = 16;
    __shared__ __attribute__((aligned(16))) float row_tile[2][W * CP];
    const int c4 = threadIdx.x & 7, xg = threadIdx.x >> 3;
    const int nband = H / R, ncs = E / 32;
    const int cs = blockIdx.x % ncs, band = (blockIdx.x / ncs) % nband, b = blockIdx.x / (ncs * nband);
    const int c0 = cs * 32 + c4 * 4, Y0 = band * R;
    // x weights of this thread's taps: output column X = F kx - F/2 + t, t < 2F (clamped; weight 0 outside the image)
    float wx2[2][4], wx4[8], wx8[16];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int X = 2 * (xg + 32 * k) - 1 + t;
            wx2[k][t] = (X >= 0 && X < W) ? tap_weight(X, 2, W / 2, xg + 32 * k) : 0.f;
        }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int X = 4 * xg - 2 + t;
        wx4[t] = (X >= 0 && X < W) ? tap_weight(X, 4, W / 4, xg) : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int X = 8 * xg - 4 + t;
        wx8[t] = (xg < 16 && X >= 0 && X < W) ? tap_weight(X, 8, W / 8, xg) : 0.f;
    }
    float4 a2[2][8], a4[4], a8[2];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int q = 0; q < 8; ++q) a2[k][q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < 4; ++q) a4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    a8[0] = a8[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float *img = dy + (size_t)b * H * W * E + cs * 32;
    // staging: thread -> pixels X = (tid >> 3) + 32 i, its own 4-channel vector
    float4 st[4];
    auto request = [&](int Y) {
        const float *src = img + (size_t)Y * W * E + c4 * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) st[i] = *reinterpret_cast<const float4 *>(src + (size_t)(xg + 32 * i) * E);
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4 *>(&row_tile[buf][(xg + 32 * i) * CP + c4 * 4]) = st[i];
    };
    auto axpy = [](float4 &a, float w, const float4 v) { a.x = fmaf(w, v.x, a.x); a.y = fmaf(w, v.y, a.y); a.z = fmaf(w, v.z, a.z); a.w = fmaf(w, v.w, a.w); };
    const int ylo = max(Y0 - 4, 0), yhi = min(Y0 + R + 4, H);      // rows [ylo, yhi)
    request(ylo);
    int buf = 0;
    for (int Y = ylo; Y < yhi; ++Y) {
        park(buf);
        if (Y + 1 < yhi) request(Y + 1);       // lands while this row is reduced
        __syncthreads();
        const float *row = row_tile[buf];
        // ---- x2 branch: tap rows 8 m .. 8 m + 7
        {
            int i0, i1;
            float lam;
            src_of(Y, 2, H / 2, i0, i1, lam);
            const int q0 = i0 - (Y0 >> 1), q1 = i1 - (Y0 >> 1);
            if ((q0 >= 0 && q0 < 8) || (q1 >= 0 && q1 < 8)) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    float4 hsum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int X = min(max(2 * (xg + 32 * k) - 1 + t, 0), W - 1);
                        axpy(hsum, wx2[k][t], *reinterpret_cast<const float4 *>(row + X * CP + c4 * 4));
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float wy = (q == q0 ? 1.f - lam : 0.f) + (q == q1 ? lam : 0.f);
                        if (wy != 0.f) axpy(a2[k][q], wy, hsum);
                    }
                }
            }
        }
        // ---- x4 branch: tap rows 4 m .. 4 m + 3
        {
            int i0, i1;
            float lam;
            src_of(Y, 4, H / 4, i0, i1, lam);
            const int q0 = i0 - (Y0 >> 2), q1 = i1 - (Y0 >> 2);
            if ((q0 >= 0 && q0 < 4) || (q1 >= 0 && q1 < 4)) {
                float4 hsum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int X = min(max(4 * xg - 2 + t, 0), W - 1);
                    axpy(hsum, wx4[t], *reinterpret_cast<const float4 *>(row + X * CP + c4 * 4));
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float wy = (q == q0 ? 1.f - lam : 0.f) + (q == q1 ? lam : 0.f);
                    if (wy != 0.f) axpy(a4[q], wy, hsum);
                }
            }
        }
        // ---- x8 branch: tap rows 2 m, 2 m + 1 (threads xg < 16: whole waves 0 and 1)
        if (xg < 16) {
            int i0, i1;
            float lam;
            src_of(Y, 8, H / 8, i0, i1, lam);
            const int q0 = i0 - (Y0 >> 3), q1 = i1 - (Y0 >> 3);
            if ((q0 >= 0 && q0 < 2) || (q1 >= 0 && q1 < 2)) {
                float4 hsum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int X = min(max(8 * xg - 4 + t, 0), W - 1);
                    axpy(hsum, wx8[t], *reinterpret_cast<const float4 *>(row + X * CP + c4 * 4));
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float wy = (q == q0 ? 1.f - lam : 0.f) + (q == q1 ? lam : 0.f);
                    if (wy != 0.f) axpy(a8[q], wy, hsum);
                }
            }
        }
        buf ^= 1;
    }
    // ---- store the band's taps
    const int h2 = H / 2, h4 = H / 4, h8 = H / 8;
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int q = 0; q < 8; ++q)
            *reinterpret_cast<float4 *>(dz2 + (((size_t)b * h2 + (Y0 >> 1) + q) * (W / 2) + xg + 32 * k) * E + c0) = a2[k][q];
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4 *>(dz3 + (((size_t)b * h4 + (Y0 >> 2) + q) * (W / 4) + xg) * E + c0) = a4[q];
    if (xg < 16) {
#pragma unroll
        for (int q = 0; q < 2; ++q) *reinterpret_cast<float4 *>(dz4 + (((size_t)b * h8 + (Y0 >> 3) + q) * (W / 8) + xg) * E + c0) = a8[q];
    }
}

// dz[b,ky,kx,:] = sum over the outputs that use tap (ky,kx) of weight * dy.  grid: ceil(B*h*w*(E/N) / 256)
template <typename T>
__global__ __launch_bounds__(256) void upsum_bwd(const T *__restrict__ dy, T *__restrict__ dz, int B, int h, int w, int E, int F) {
    constexpr int N = HV<T>::N;
    const int ev = E / N;
    const size_t t = xcd_item(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (t >= (size_t)B * h * w * ev) return;
    const int c = (int)(t % ev) * N;
    const size_t tap = t / ev;
    const int kx = (int)(tap % w);
    const int ky = (int)((tap / w) % h);
    const int b = (int)(tap / ((size_t)w * h));
    const int H = h * F, W = w * F;
    float acc[N];
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = 0.f;
    const int Ya = max(0, F * ky - F / 2), Yb = min(H, F * ky + F + F / 2);  // exactly the outputs that use tap row ky
    const int Xa = max(0, F * kx - F / 2), Xb = min(W, F * kx + F + F / 2);
    for (int Y = Ya; Y < Yb; ++Y) {
        int y0, y1;
        float ly;
        src_of(Y, F, h, y0, y1, ly);
        const float wy = (y0 == ky ? 1.f - ly : 0.f) + (y1 == ky ? ly : 0.f);
        if (wy == 0.f) continue;
        const T *row = dy + (((size_t)b * H + Y) * W) * E + c;
        for (int X = Xa; X < Xb; ++X) {
            int x0, x1;
            float lx;
            src_of(X, F, w, x0, x1, lx);
            const float wx = (x0 == kx ? 1.f - lx : 0.f) + (x1 == kx ? lx : 0.f);
            if (wx == 0.f) continue;
            float g[N];
            HV<T>::load(row + (size_t)X * E, g);
            const float ww = wy * wx;
#pragma unroll
            for (int i = 0; i < N; ++i) acc[i] = fmaf(ww, g[i], acc[i]);
        }
    }
    HV<T>::store(dz + tap * E + c, acc);
}

bool ok_factor(int f) { return f == 2 || f == 4 || f == 8; }

}  // namespace

int g_upsum_bwd_band = 1;     // tunable "upsum_bwd_band" (A/B, tests): 0 = the two-pass form with global row partials (round 2)

int headfuse_tunable(const char *key, int set, int v) {
    if (strcmp(key, "upsum_bwd_band")) return SD_E_UNSUPPORTED;
    if (!set) return g_upsum_bwd_band;
    if (v != 0 && v != 1) return SD_E_SHAPE;
    g_upsum_bwd_band = v;
    return SD_OK;
}

namespace {
}  // namespace
}  // namespace sd

extern "C" {

static int upsum_fwd_impl(const void *z1, const void *z2, const void *z3, const void *z4, const float *bias, const float *scale,
                          const float *shift, int relu, void *y, int dtype, int B, int H, int W, int E, int f2, int f3, int f4, void *stream) {
    if (!z1 || !z2 || !z3 || !z4 || !y) return SD_E_NULL;
    if ((scale == nullptr) != (shift == nullptr)) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (B <= 0 || H <= 0 || W <= 0 || E <= 0) return SD_E_SHAPE;
    if (!sd::ok_factor(f2) || !sd::ok_factor(f3) || !sd::ok_factor(f4) || H % f2 || W % f2 || H % f3 || W % f3 || H % f4 || W % f4)
        return SD_E_UNSUPPORTED;
    if (E % (dtype == SD_F32 ? 4 : 8)) return SD_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int N = dtype == SD_F32 ? 4 : 8;
    if (f2 == 2 && f3 == 4 && f4 == 8 && W % 4 == 0 && (size_t)H * W * E < 0x7fffffffull && (size_t)B * H <= 65535) {   // the SegFormer geometry: strip kernel
        const dim3 g((unsigned)(((size_t)(W / 4) * (E / N) + 255) / 256), (unsigned)(B * H));
        if (dtype == SD_F32)
            hipLaunchKernelGGL((sd::upsum_fwd_strip<float>), g, dim3(256), 0, st, (const float *)z1, (const float *)z2, (const float *)z3,
                               (const float *)z4, bias, scale, shift, relu, (float *)y, B, H, W, E);
        else
            hipLaunchKernelGGL((sd::upsum_fwd_strip<sd::bf16_t>), g, dim3(256), 0, st, (const sd::bf16_t *)z1, (const sd::bf16_t *)z2,
                               (const sd::bf16_t *)z3, (const sd::bf16_t *)z4, bias, scale, shift, relu, (sd::bf16_t *)y, B, H, W, E);
        return (int)hipGetLastError();
    }
    const size_t total = (size_t)B * H * W * (E / N);
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (dtype == SD_F32)
        hipLaunchKernelGGL((sd::upsum_fwd<float>), dim3(grid), dim3(256), 0, st, (const float *)z1, (const float *)z2, (const float *)z3,
                           (const float *)z4, bias, scale, shift, relu, (float *)y, B, H, W, E, f2, f3, f4);
    else
        hipLaunchKernelGGL((sd::upsum_fwd<sd::bf16_t>), dim3(grid), dim3(256), 0, st, (const sd::bf16_t *)z1, (const sd::bf16_t *)z2,
                           (const sd::bf16_t *)z3, (const sd::bf16_t *)z4, bias, scale, shift, relu, (sd::bf16_t *)y, B, H, W, E, f2, f3, f4);
    return (int)hipGetLastError();
}

int sd_upsum_fwd(const void *z1, const void *z2, const void *z3, const void *z4, const float *bias, void *y, int dtype, int B, int H, int W,
                 int E, int f2, int f3, int f4, void *stream) {
    return upsum_fwd_impl(z1, z2, z3, z4, bias, nullptr, nullptr, 0, y, dtype, B, H, W, E, f2, f3, f4, stream);
}

int sd_upsum_affine_fwd(const void *z1, const void *z2, const void *z3, const void *z4, const float *bias, const float *scale,
                        const float *shift, int relu, void *y, int dtype, int B, int H, int W, int E, int f2, int f3, int f4, void *stream) {
    if (!scale || !shift) return SD_E_NULL;
    return upsum_fwd_impl(z1, z2, z3, z4, bias, scale, shift, relu, y, dtype, B, H, W, E, f2, f3, f4, stream);
}

size_t sd_upsum_bwd3_workspace_bytes(int B, int H, int W, int E) {
    if (B <= 0 || H <= 0 || W <= 0 || E <= 0) return 0;
    return (size_t)B * H * (W / 2 + W / 4 + W / 8) * E * sizeof(float) + 16;
}

int sd_upsum_bwd3(const void *dy, void *dz2, void *dz3, void *dz4, int dtype, int B, int H, int W, int E, void *workspace, size_t workspace_bytes,
                  void *stream) {
    if (!dy || !dz2 || !dz3 || !dz4 || !workspace) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (B <= 0 || H <= 0 || W <= 0 || E <= 0) return SD_E_SHAPE;
    if (H % 8 || W % 8 || E % 64 || W > 512) return SD_E_UNSUPPORTED;
    if (workspace_bytes < sd_upsum_bwd3_workspace_bytes(B, H, W, E) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return SD_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *r2 = static_cast<float *>(workspace);
    float *r4 = r2 + (size_t)B * H * (W / 2) * E;
    float *r8 = r4 + (size_t)B * H * (W / 4) * E;
    const size_t lds = (size_t)W * 68 * sizeof(float);
    const size_t taps = (size_t)B * ((size_t)(H / 2) * (W / 2) + (size_t)(H / 4) * (W / 4) + (size_t)(H / 8) * (W / 8));
    if (dtype == SD_F32 && W == 128 && H % 16 == 0 && E % 32 == 0 && sd::g_upsum_bwd_band) {
        // one pass, no global partials (round 3)
        hipLaunchKernelGGL(sd::upsum_bwd3_band, dim3((unsigned)((size_t)B * (H / 16) * (E / 32))), dim3(256), 0, st, (const float *)dy, (float *)dz2,
                           (float *)dz3, (float *)dz4, H, E);
        return (int)hipGetLastError();
    }
    if (dtype == SD_F32) {
        static bool raised = false;
        if (lds > 64 * 1024 && !raised) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sd::upsum_bwd_rows<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            raised = true;
        }
        hipLaunchKernelGGL((sd::upsum_bwd_rows<float>), dim3((unsigned)(B * H), E / 64), dim3(256), lds, st, (const float *)dy, r2, r4, r8, H, W, E);
        const size_t total = taps * (E / 4);
        hipLaunchKernelGGL((sd::upsum_bwd_cols<float>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, r2, r4, r8, (float *)dz2, (float *)dz3,
                           (float *)dz4, B, H, W, E);
    } else {
        static bool raised = false;
        if (lds > 64 * 1024 && !raised) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sd::upsum_bwd_rows<sd::bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               160 * 1024);
            if (e != hipSuccess) return (int)e;
            raised = true;
        }
        hipLaunchKernelGGL((sd::upsum_bwd_rows<sd::bf16_t>), dim3((unsigned)(B * H), E / 64), dim3(256), lds, st, (const sd::bf16_t *)dy, r2, r4, r8, H, W, E);
        const size_t total = taps * (E / 8);
        hipLaunchKernelGGL((sd::upsum_bwd_cols<sd::bf16_t>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, r2, r4, r8, (sd::bf16_t *)dz2,
                           (sd::bf16_t *)dz3, (sd::bf16_t *)dz4, B, H, W, E);
    }
    return (int)hipGetLastError();
}

int sd_upsum_bwd(const void *dy, void *dz, int dtype, int B, int h, int w, int E, int F, void *stream) {
    if (!dy || !dz) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (B <= 0 || h <= 0 || w <= 0 || E <= 0) return SD_E_SHAPE;
    if (!sd::ok_factor(F) || E % (dtype == SD_F32 ? 4 : 8)) return SD_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int N = dtype == SD_F32 ? 4 : 8;
    const size_t total = (size_t)B * h * w * (E / N);
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (dtype == SD_F32)
        hipLaunchKernelGGL((sd::upsum_bwd<float>), dim3(grid), dim3(256), 0, st, (const float *)dy, (float *)dz, B, h, w, E, F);
    else
        hipLaunchKernelGGL((sd::upsum_bwd<sd::bf16_t>), dim3(grid), dim3(256), 0, st, (const sd::bf16_t *)dy, (sd::bf16_t *)dz, B, h, w, E, F);
    return (int)hipGetLastError();
}

}  // extern "C"
