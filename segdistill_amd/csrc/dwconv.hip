// dwconv.hip -- depth-wise 3x3 convolution (stride 1, zero pad 1) on TOKEN-MAJOR activations
// [B, H*W, C] (= NHWC), forward / backward-data / backward-weight(+bias), gfx950.
//
// Why it exists: the Mix-FFN of every MiT block is Linear -> DWConv3x3 -> GELU -> Linear
// (reference mmseg/models/backbones/mix_transformer.py:20-55, DWConv :376-387).  The reference
// transposes the tokens to NCHW, calls a grouped cuDNN conv and transposes back.  On ROCm that
// grouped conv lands on generic MIOpen/CK kernels: the round-1 rocprof of the KD step shows
// 15.1 ms/step (22 %) in the depth-wise WEIGHT gradient alone and 8.2 ms in the forward
// (profiles/r01_train_step_kernels_baseline.txt), for an op whose roofline is one read and one
// write of the hidden tensor.  Depth-wise conv is HBM-bound byte work, not a GEMM: in the
// token-major layout the channel axis is contiguous, so a lane owns one 16-byte channel vector
// and consecutive lanes consecutive channels (fully coalesced), no transposes at all.
//
//   fwd      y[b,p,c]  = bias[c] + sum_k w[k][c] * x[b, p+off(k), c]          k = 3*ky+kx
//   bwd-data dx        = same kernel on dy with the taps mirrored (k -> 8-k), no bias
//   bwd-wgt  dw[k][c]  = sum_{b,p} dy[b,p,c] * x[b,p+off(k),c] ;  db[c] = sum dy
//            (per-workgroup partials in a workspace, then a deterministic second pass; no float atomics)
// Weights are passed, and weight gradients returned, in nn.Conv2d's OWN layout [C][9] fp32 (= [C,1,3,3] contiguous, k = 3*ky+kx): a
// lane's N channels are 9N consecutive floats, and the binding needs no transpose copy in either direction (it used to launch one per
// block and pass -- 16 tiny kernels per step of Segformer-B0).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <type_traits>

#include "cgd_device.h"

namespace sd {

namespace {

constexpr int kStrip = 4;  // output pixels per thread along the row

template <typename T> struct CV;  // channel vector of 16 bytes
template <> struct CV<float> {
    static constexpr int N = 4;
    typedef float raw_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ void unpack(const raw_t &v, float (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
    static __device__ __forceinline__ void load(const float *p, float (&o)[4]) { unpack(*reinterpret_cast<const raw_t *>(p), o); }
    static __device__ __forceinline__ void store(float *p, const float (&o)[4]) {
        raw_t v = {o[0], o[1], o[2], o[3]};
        *reinterpret_cast<raw_t *>(p) = v;
    }
};
template <> struct CV<bf16_t> {
    static constexpr int N = 8;
    typedef unsigned int raw_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ void unpack(const raw_t &v, float (&o)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[2 * i] = __uint_as_float(v[i] << 16);
            o[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void load(const bf16_t *p, float (&o)[8]) { unpack(*reinterpret_cast<const raw_t *>(p), o); }
    static __device__ __forceinline__ void store(bf16_t *p, const float (&o)[8]) {
        raw_t v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (unsigned)f32_to_bf16(o[2 * i]) | ((unsigned)f32_to_bf16(o[2 * i + 1]) << 16);
        *reinterpret_cast<raw_t *>(p) = v;
    }
};

// grid: (ceil(strips_per_row * (C/N) / 256), B*H).  A thread produces kStrip consecutive pixels of one row
// for one channel vector; per input row it loads the kStrip+2 columns once (1.5 loads per tap-row-pixel).
// exact (erf) GELU, the nn.GELU() default used by the Mix-FFN (mix_transformer.py:20, act_layer=nn.GELU)
//
// erff() of the device library is two polynomial branches (|x| < 1, else an exp form) and the pre-activations straddle |x| = 1 in nearly
// every wave, so both run: ~50 VALU slots per element against 9 FMAs of convolution -- the kernel was VALU-bound at twice its HBM floor.
// One branch-free form instead (Abramowitz-Stegun 7.1.26 shape, refit to degree 6):
//     erfc(a) = t P(t) exp(-a^2),  t = 1 / (1 + 0.39 a),  a = |v| / sqrt(2)          max |error| 1.1e-8 in exact arithmetic on [0, inf)
//     1 + erf(v / sqrt 2) = erfc(a) for v < 0,  2 - erfc(a) otherwise
// ~16 VALU + rcp + exp2.  The negative tail is computed without the 1 + erf cancellation, so it is MORE accurate there than the library
// form; over v in [-12, 12] the f32 result is within 3.9e-7 (< 1 ulp of the value) of the f64 GELU, the library form within 4.5e-7
// (tests/test_dwconv_gpu.py::test_gelu_error_bound holds the kernel to 6e-7).
__device__ __forceinline__ float gelu_erf(float v) {
    const float a = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.39f, a, 1.f));
    float p = -0.22753699123859406f;
    p = fmaf(p, t, 0.8866638541221619f);
    p = fmaf(p, t, -0.6353994011878967f);
    p = fmaf(p, t, 0.6495586633682251f);
    p = fmaf(p, t, 0.09138870239257812f);
    p = fmaf(p, t, 0.2353251725435257f);
    const float q = p * t * __builtin_amdgcn_exp2f(a * a * -1.4426950408889634f);   // erfc(a); exp2 underflows to 0 for |v| > ~14.4
    return 0.5f * v * (v < 0.f ? q : 2.f - q);
}

template <typename T, bool FLIP, bool BIAS, bool GELU = false>
__global__ __launch_bounds__(256) void dw3x3_fwd(const T *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                  T *__restrict__ y, int H, int W, int C, T *__restrict__ y_pre = nullptr) {
    constexpr int N = CV<T>::N;
    const int cv = C / N;
    // XCD-aware work order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2) in linear-id order, and a
    // row needs its two neighbour rows -- so each XCD is given a CONTIGUOUS band of (row, x-block) items instead of every
    // 8th one, and the 3x re-read of the input rows is served by that XCD's L2 rather than by the fabric.
    const unsigned total = gridDim.x * gridDim.y, q8 = total / 8, r8 = total % 8;
    const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x, xcd = lin % 8;
    const unsigned item = xcd * q8 + (xcd < r8 ? xcd : r8) + lin / 8;      // a bijection of [0, total): XCD x owns q8 (+1) consecutive items
    const int t = (item % gridDim.x) * 256 + threadIdx.x;
    const int spr = (W + kStrip - 1) / kStrip;
    if (t >= spr * cv) return;
    const int c = (t % cv) * N;
    // The 9 taps of this lane's N channels: 9N consecutive floats of nn.Conv2d's [C][9] layout (16-byte aligned: N % 4 == 0), held in
    // registers -- 9N/4 vector loads instead of the 9N scalar gathers (one per tap and channel) this kernel used to issue.  Every index
    // below is a compile-time constant after unrolling.  (Staging them tap-major through LDS for the whole workgroup cost more VALU in
    // index arithmetic than the gathers: 1217 instructions per wave against ~700, on a kernel that is VALU-bound -- SQ counters in
    // profiles/r02_pmc_dw3x3.txt.)
    float wreg[9 * N];
#pragma unroll
    for (int i = 0; i < 9 * N; i += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(w + (size_t)c * 9 + i);
        wreg[i] = v.x; wreg[i + 1] = v.y; wreg[i + 2] = v.z; wreg[i + 3] = v.w;
    }
    const int x0 = (t / cv) * kStrip;
    const int row = item / gridDim.x;  // b*H + yy
    const int yy = row % H;
    const size_t img = (size_t)(row - yy) * W;  // pixel index of (b, 0, 0)
    // `inner`: all kStrip + 2 input columns and all kStrip outputs lie inside the row -- no bounds test, no zero fill.  Only the first
    // and last strip of a row take the checked path (the halo tests and their zero fills were a quarter of this VALU-bound kernel).
    // In-row offsets are 32-bit (the launcher rejects W * C >= 2^31); only the row origin is a 64-bit product.
    const bool inner = x0 >= 1 && x0 + kStrip + 1 <= W;

    float acc[kStrip][N];
#pragma unroll
    for (int p = 0; p < kStrip; ++p)
#pragma unroll
        for (int i = 0; i < N; ++i) acc[p][i] = 0.f;
    if constexpr (BIAS) {
        float bv[N];
#pragma unroll
        for (int i = 0; i < N; i += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(bias + c + i);
            bv[i] = v.x; bv[i + 1] = v.y; bv[i + 2] = v.z; bv[i + 3] = v.w;
        }
#pragma unroll
        for (int p = 0; p < kStrip; ++p)
#pragma unroll
            for (int i = 0; i < N; ++i) acc[p][i] = bv[i];
    }
    // All 3 x (kStrip + 2) input vectors are requested before the first one is used, from clamped (always valid) addresses and with no
    // branch in between: a wave has 18 16-byte loads in flight instead of three dependent rounds of 6 (each round waited for the previous
    // one: ~7 us per wave, 2.9 TB/s on an HBM-resident map with the VALU 14 % busy).  Out-of-image rows are skipped below (uniform per
    // workgroup), out-of-row columns zeroed on the checked path.
    typename CV<T>::raw_t raw[3][kStrip + 2];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = min(max(yy + ky - 1, 0), H - 1);
        const T *xr = x + ((img + (size_t)iy * W) * C + c);
#pragma unroll
        for (int j = 0; j < kStrip + 2; ++j) {
            raw[ky][j] = *reinterpret_cast<const typename CV<T>::raw_t *>(xr + min(max(x0 + j - 1, 0), W - 1) * C);
        }
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = yy + ky - 1;
        if (iy < 0 || iy >= H) continue;
        float col[kStrip + 2][N];
#pragma unroll
        for (int j = 0; j < kStrip + 2; ++j) CV<T>::unpack(raw[ky][j], col[j]);
        if (!inner) {
#pragma unroll
            for (int j = 0; j < kStrip + 2; ++j) {
                const int ix = x0 + j - 1;
                if (ix < 0 || ix >= W) {
#pragma unroll
                    for (int i = 0; i < N; ++i) col[j][i] = 0.f;
                }
            }
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            float wv[N];
            const int k = FLIP ? 8 - (3 * ky + kx) : 3 * ky + kx;
#pragma unroll
            for (int i = 0; i < N; ++i) wv[i] = wreg[9 * i + k];
#pragma unroll
            for (int p = 0; p < kStrip; ++p)
#pragma unroll
                for (int i = 0; i < N; ++i) acc[p][i] = fmaf(wv[i], col[p + kx][i], acc[p][i]);
        }
    }
    const size_t orow = (img + (size_t)yy * W) * C + c;
    auto put = [&](T *dst) {
        if (inner) {
#pragma unroll
            for (int p = 0; p < kStrip; ++p) CV<T>::store(dst + orow + (x0 + p) * C, acc[p]);
        } else {
#pragma unroll
            for (int p = 0; p < kStrip; ++p)
                if (x0 + p < W) CV<T>::store(dst + orow + (x0 + p) * C, acc[p]);
        }
    };
    if constexpr (GELU) {  // epilogue GELU; the frozen teacher never needs the pre-activation, training keeps it in y_pre for the backward
        if (y_pre) put(y_pre);
#pragma unroll
        for (int p = 0; p < kStrip; ++p)
#pragma unroll
            for (int i = 0; i < N; ++i) acc[p][i] = gelu_erf(acc[p][i]);
    }
    put(y);
}

// Weight / bias gradient partials.  A "segment" is up to SEG consecutive pixels of one image row.
// grid: (ceil(C/N / LX), ceil(B*H*nseg / RY)); block = LX x RY threads, LX lanes along the channel vectors and
// RY segments; each thread walks its segment with a sliding 3x3 window (loads for column ix+2 are issued one
// iteration ahead of their use: 4 vector loads + 9 vector FMAs per pixel) and the RY threads of a channel
// vector are combined through LDS.  part layout: [gridDim.y][10][C]  (k = 0..8 taps, k = 9 bias).
// (gx x gy workgroups, this one the lin-th of them: the whole grid in the single launch, a slice of it -- starting at a multiple of 8, so that
// lin % 8 is still the XCD -- in the grouped launch below)
template <typename T>
__device__ __forceinline__ void dw3x3_wgrad_body(const T *__restrict__ x, const T *__restrict__ dy, float *__restrict__ part, int nsegs, int nseg, int SEG,
                                                 int H, int W, int C, int LX, int RY, unsigned gx, unsigned gy, unsigned lin) {
    constexpr int N = CV<T>::N;
    extern __shared__ float red[];  // [RY][10][LX*N]
    const int lx = threadIdx.x % LX, ry = threadIdx.x / LX;
    // XCD-aware order (see dw3x3_fwd): consecutive segment groups -- neighbouring image rows -- stay on one XCD's L2
    const unsigned total = gx * gy, q8 = total / 8, r8 = total % 8;
    const unsigned xcd = lin % 8;
    const unsigned item = xcd * q8 + (xcd < r8 ? xcd : r8) + lin / 8;
    const unsigned bx = item % gx, by = item / gx;
    const int c = (bx * LX + lx) * N;
    const int sid = by * RY + ry;
    const bool live = (c < C) && (sid < nsegs);
    float acc[10][N];
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int i = 0; i < N; ++i) acc[k][i] = 0.f;
    if (live) {
        const int row = sid / nseg;
        const int xa = (sid - row * nseg) * SEG, xb = min(W, xa + SEG);
        const int yy = row % H;
        const long img = (long)(row - yy) * W;
        const bool up = yy > 0, dn = yy + 1 < H;
        const T *r1 = x + ((img + (long)yy * W) * C + c);
        const T *r0 = r1 - (long)W * C;   // only dereferenced when `up`
        const T *r2 = r1 + (long)W * C;   // only dereferenced when `dn`
        const T *g = dy + ((img + (long)yy * W) * C + c);
        float win[3][3][N], nxt[3][N], gv[N], gn[N];
        auto load_col = [&](int ix, float (&d)[3][N]) {
            const bool in = ix >= 0 && ix < W;
            if (in && up) CV<T>::load(r0 + (long)ix * C, d[0]);
            else {
#pragma unroll
                for (int i = 0; i < N; ++i) d[0][i] = 0.f;
            }
            if (in) CV<T>::load(r1 + (long)ix * C, d[1]);
            else {
#pragma unroll
                for (int i = 0; i < N; ++i) d[1][i] = 0.f;
            }
            if (in && dn) CV<T>::load(r2 + (long)ix * C, d[2]);
            else {
#pragma unroll
                for (int i = 0; i < N; ++i) d[2][i] = 0.f;
            }
        };
        float c0[3][N], c1[3][N];
        load_col(xa - 1, c0);
        load_col(xa, c1);
        load_col(xa + 1, nxt);
        CV<T>::load(g + (long)xa * C, gn);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int i = 0; i < N; ++i) { win[r][1][i] = c0[r][i]; win[r][2][i] = c1[r][i]; }
        for (int ix = xa; ix < xb; ++ix) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int i = 0; i < N; ++i) { win[r][0][i] = win[r][1][i]; win[r][1][i] = win[r][2][i]; win[r][2][i] = nxt[r][i]; }
#pragma unroll
            for (int i = 0; i < N; ++i) gv[i] = gn[i];
            load_col(ix + 2, nxt);                                  // consumed next iteration
            if (ix + 1 < xb) CV<T>::load(g + (long)(ix + 1) * C, gn);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int i = 0; i < N; ++i) acc[3 * r + j][i] = fmaf(gv[i], win[r][j][i], acc[3 * r + j][i]);
#pragma unroll
            for (int i = 0; i < N; ++i) acc[9][i] += gv[i];
        }
    }
    const int LC = LX * N;
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int i = 0; i < N; ++i) red[(ry * 10 + k) * LC + lx * N + i] = acc[k][i];
    __syncthreads();
    for (int e = threadIdx.x; e < 10 * LC; e += 256) {  // element e = k*LC + col, summed over the RY slices
        float s = 0.f;
        for (int r = 0; r < RY; ++r) s += red[r * 10 * LC + e];
        const int k = e / LC, col = e - k * LC;
        const int cc = bx * LC + col;
        if (cc < C) part[(size_t)by * 10 * C + (k < 9 ? cc * 9 + k : 9 * C + cc)] = s;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dw3x3_wgrad_partials(const T *__restrict__ x, const T *__restrict__ dy, float *__restrict__ part,
                                                             int nsegs, int nseg, int SEG, int H, int W, int C, int LX, int RY) {
    dw3x3_wgrad_body<T>(x, dy, part, nsegs, nseg, SEG, H, W, C, LX, RY, gridDim.x, gridDim.y, blockIdx.y * gridDim.x + blockIdx.x);
}

// The filter gradients of ALL depth-wise convolutions of a backward in one launch (round 5; deferred to the backward's end like the Linear weight
// gradients -- csrc/wgrad_tn.hip, segdistill_amd/deferred.py): eight launches of 128 ... 512 workgroups per Segformer-B0 step become one.
constexpr int kDwMultiMax = 24;
struct DwMultiTable {
    const void *x[kDwMultiMax], *dy[kDwMultiMax];
    float *part[kDwMultiMax];
    int nsegs[kDwMultiMax], nseg[kDwMultiMax], H[kDwMultiMax], W[kDwMultiMax], C[kDwMultiMax], LX[kDwMultiMax], RY[kDwMultiMax], gx[kDwMultiMax],
        gy[kDwMultiMax];
    int blk_begin[kDwMultiMax + 1];     // multiples of 8
    int njobs;
};
template <typename T>
__global__ __launch_bounds__(256) void dw3x3_wgrad_partials_multi(const DwMultiTable t) {
    int lo = 0, hi = t.njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int)blockIdx.x >= t.blk_begin[mid]) lo = mid;
        else hi = mid - 1;
    }
    const int j = lo;
    const unsigned lin = blockIdx.x - (unsigned)t.blk_begin[j];
    if (lin >= (unsigned)(t.gx[j] * t.gy[j])) return;       // padding of the slice (uniform over the workgroup: no barrier is skipped by a part of it)
    dw3x3_wgrad_body<T>((const T *)t.x[j], (const T *)t.dy[j], t.part[j], t.nsegs[j], t.nseg[j], 32, t.H[j], t.W[j], t.C[j], t.LX[j], t.RY[j],
                        (unsigned)t.gx[j], (unsigned)t.gy[j], lin);
}

// out[o] = sum_p part[p][o], o < 10*C (9*C weight-gradient entries in conv layout, then C bias sums).  grid: ceil(10*C / 64);
// block 256 = 4 partial-groups x 64 outputs.
__global__ __launch_bounds__(256) void dw3x3_wgrad_reduce(const float *__restrict__ part, float *__restrict__ dw, float *__restrict__ db,
                                                           int nparts, int C) {
    __shared__ float red[4][64];
    const int o = blockIdx.x * 64 + (threadIdx.x & 63);
    const int grp = threadIdx.x >> 6;
    float s = 0.f;
    if (o < 10 * C)
        for (int p = grp; p < nparts; p += 4) s += part[(size_t)p * 10 * C + o];
    red[grp][threadIdx.x & 63] = s;
    __syncthreads();
    if (grp == 0 && o < 10 * C) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (o < 9 * C) dw[o] = v;
        else if (db) db[o - 9 * C] = v;
    }
}

constexpr int kSeg = 32;  // pixels per wgrad segment

struct WgGeo {
    int LX, RY, gx, gy, nseg, nsegs;
    size_t lds;
};
template <typename T> WgGeo wgrad_geo(int B, int H, int W, int C) {
    WgGeo q;
    const int cv = C / CV<T>::N;
    q.LX = cv < 64 ? (cv < 1 ? 1 : cv) : 64;
    while (256 % q.LX) --q.LX;                // LX must divide the block
    q.RY = 256 / q.LX;
    q.gx = (cv + q.LX - 1) / q.LX;
    q.nseg = (W + kSeg - 1) / kSeg;
    q.nsegs = B * H * q.nseg;
    q.gy = (q.nsegs + q.RY - 1) / q.RY;
    q.lds = (size_t)q.RY * 10 * q.LX * CV<T>::N * sizeof(float);
    return q;
}

int check_dw(const void *a, const void *b, int dtype, int B, int H, int W, int C) {
    if (!a || !b) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (long)B * H > 0x7fffffffL || (long)(W + 2) * C > 0x7fffffffL) return SD_E_SHAPE;   // in-row offsets are int
    if (C % (dtype == SD_F32 ? 4 : 8)) return SD_E_UNSUPPORTED;  // 16-byte channel vectors
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) return SD_E_ALIGN;
    return SD_OK;
}

template <typename T>
int fwd_launch(const void *x, const float *w, const float *bias, void *y, int B, int H, int W, int C, bool flip, hipStream_t st,
               bool gelu = false, void *y_pre = nullptr) {
    const int cv = C / CV<T>::N;
    const int spr = (W + kStrip - 1) / kStrip;
    dim3 grid((spr * cv + 255) / 256, B * H);
    auto go = [&](auto kern, const float *b, void *pre) -> int {
        hipLaunchKernelGGL(kern, grid, dim3(256), 0, st, (const T *)x, w, b, (T *)y, H, W, C, (T *)pre);
        return (int)hipGetLastError();
    };
    if (gelu && bias) return go(dw3x3_fwd<T, false, true, true>, bias, y_pre);
    if (gelu) return go(dw3x3_fwd<T, false, false, true>, nullptr, y_pre);
    if (flip) return go(dw3x3_fwd<T, true, false>, nullptr, nullptr);
    if (bias) return go(dw3x3_fwd<T, false, true>, bias, nullptr);
    return go(dw3x3_fwd<T, false, false>, nullptr, nullptr);
}

template <typename T>
int wgrad_launch(const void *x, const void *dy, float *dw, float *db, void *ws, size_t ws_bytes, int B, int H, int W, int C, hipStream_t st) {
    const WgGeo q = wgrad_geo<T>(B, H, W, C);
    if (ws_bytes < (size_t)q.gy * 10 * C * sizeof(float) || (reinterpret_cast<uintptr_t>(ws) & 15)) return SD_E_WORKSPACE;
    if (q.gy > 65535) return SD_E_SHAPE;
    float *part = static_cast<float *>(ws);
    if (q.lds > 64 * 1024) {  // bf16: 80 KB of the CU's 160 KB LDS
        static bool raised = false;
        if (!raised) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&dw3x3_wgrad_partials<T>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            if (e != hipSuccess) return (int)e;
            raised = true;
        }
    }
    hipLaunchKernelGGL((dw3x3_wgrad_partials<T>), dim3(q.gx, q.gy), dim3(256), q.lds, st, (const T *)x, (const T *)dy, part, q.nsegs,
                       q.nseg, kSeg, H, W, C, q.LX, q.RY);
    // dw == NULL: the partials [gy][10*C] stay in the workspace for a deferred combine (sd_multi_slab_reduce; gy = sd_dwconv3x3_wgrad_slabs)
    if (dw) hipLaunchKernelGGL(dw3x3_wgrad_reduce, dim3((10 * C + 63) / 64), dim3(256), 0, st, part, dw, db, q.gy, C);
    return (int)hipGetLastError();
}

template <typename T>
int wgrad_multi_launch(const sd_dw_wgrad_job *jobs, int njobs, hipStream_t st) {
    static_assert(kSeg == 32, "the grouped kernel passes SEG = 32");
    for (int base = 0; base < njobs; base += kDwMultiMax) {
        const int n = njobs - base < kDwMultiMax ? njobs - base : kDwMultiMax;
        DwMultiTable t;
        memset(&t, 0, sizeof(t));
        long blk = 0;
        size_t lds = 0;
        for (int i = 0; i < n; ++i) {
            const sd_dw_wgrad_job &q = jobs[base + i];
            int rc = check_dw(q.x, q.dy, std::is_same<T, float>::value ? SD_F32 : SD_BF16, q.B, q.H, q.W, q.C);
            if (rc) return rc;
            if (!q.partials) return SD_E_NULL;
            const WgGeo g = wgrad_geo<T>(q.B, q.H, q.W, q.C);
            if (q.partials_bytes < (size_t)g.gy * 10 * q.C * sizeof(float) || (reinterpret_cast<uintptr_t>(q.partials) & 15)) return SD_E_WORKSPACE;
            t.x[i] = q.x, t.dy[i] = q.dy, t.part[i] = q.partials;
            t.nsegs[i] = g.nsegs, t.nseg[i] = g.nseg, t.H[i] = q.H, t.W[i] = q.W, t.C[i] = q.C, t.LX[i] = g.LX, t.RY[i] = g.RY, t.gx[i] = g.gx, t.gy[i] = g.gy;
            t.blk_begin[i] = (int)blk;
            blk += ((long)g.gx * g.gy + 7) / 8 * 8;
            if (blk > 0x7fffffffL) return SD_E_SHAPE;
            if (g.lds > lds) lds = g.lds;
        }
        t.blk_begin[n] = (int)blk;
        t.njobs = n;
        if (lds > 64 * 1024) {
            static bool raised = false;
            if (!raised) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&dw3x3_wgrad_partials_multi<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   96 * 1024);
                if (e != hipSuccess) return (int)e;
                raised = true;
            }
        }
        hipLaunchKernelGGL((dw3x3_wgrad_partials_multi<T>), dim3((unsigned)blk), dim3(256), lds, st, t);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return SD_OK;
}

}  // namespace
}  // namespace sd

extern "C" {

int sd_dwconv3x3_wgrad_multi(const sd_dw_wgrad_job *jobs, int njobs, int dtype, void *stream) {
    if (!jobs) return SD_E_NULL;
    if (njobs <= 0 || njobs > 4096) return SD_E_SHAPE;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return dtype == SD_F32 ? sd::wgrad_multi_launch<float>(jobs, njobs, st) : sd::wgrad_multi_launch<sd::bf16_t>(jobs, njobs, st);
}

int sd_dwconv3x3_wgrad_slabs(int dtype, int B, int H, int W, int C) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    return dtype == SD_F32 ? sd::wgrad_geo<float>(B, H, W, C).gy : sd::wgrad_geo<sd::bf16_t>(B, H, W, C).gy;
}

size_t sd_dwconv3x3_workspace_bytes(int dtype, int B, int H, int W, int C) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    const int gy = dtype == SD_F32 ? sd::wgrad_geo<float>(B, H, W, C).gy : sd::wgrad_geo<sd::bf16_t>(B, H, W, C).gy;
    return (size_t)gy * 10 * C * sizeof(float) + 16;
}

int sd_dwconv3x3_fwd(const void *x, const float *w, const float *bias, void *y, int dtype, int B, int H, int W, int C,
                     void *stream) {
    int rc = sd::check_dw(x, y, dtype, B, H, W, C);
    if (rc) return rc;
    if (!w) return SD_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32) return sd::fwd_launch<float>(x, w, bias, y, B, H, W, C, false, st);
    return sd::fwd_launch<sd::bf16_t>(x, w, bias, y, B, H, W, C, false, st);
}

int sd_dwconv3x3_gelu_fwd(const void *x, const float *w, const float *bias, void *y, int dtype, int B, int H, int W, int C,
                          void *stream) {
    int rc = sd::check_dw(x, y, dtype, B, H, W, C);
    if (rc) return rc;
    if (!w) return SD_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32) return sd::fwd_launch<float>(x, w, bias, y, B, H, W, C, false, st, true);
    return sd::fwd_launch<sd::bf16_t>(x, w, bias, y, B, H, W, C, false, st, true);
}

int sd_dwconv3x3_gelu_fwd_train(const void *x, const float *w, const float *bias, void *y_pre, void *y, int dtype, int B, int H, int W,
                                int C, void *stream) {
    int rc = sd::check_dw(x, y, dtype, B, H, W, C);
    if (rc) return rc;
    if (!w || !y_pre) return SD_E_NULL;
    if (reinterpret_cast<uintptr_t>(y_pre) & 15) return SD_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32) return sd::fwd_launch<float>(x, w, bias, y, B, H, W, C, false, st, true, y_pre);
    return sd::fwd_launch<sd::bf16_t>(x, w, bias, y, B, H, W, C, false, st, true, y_pre);
}

int sd_dwconv3x3_bwd_data(const void *dy, const float *w, void *dx, int dtype, int B, int H, int W, int C, void *stream) {
    int rc = sd::check_dw(dy, dx, dtype, B, H, W, C);
    if (rc) return rc;
    if (!w) return SD_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32) return sd::fwd_launch<float>(dy, w, nullptr, dx, B, H, W, C, true, st);
    return sd::fwd_launch<sd::bf16_t>(dy, w, nullptr, dx, B, H, W, C, true, st);
}

int sd_dwconv3x3_bwd_weight(const void *x, const void *dy, float *dw, float *dbias, int dtype, int B, int H, int W, int C,
                            void *workspace, size_t workspace_bytes, void *stream) {
    int rc = sd::check_dw(x, dy, dtype, B, H, W, C);
    if (rc) return rc;
    if (!workspace) return SD_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32) return sd::wgrad_launch<float>(x, dy, dw, dbias, workspace, workspace_bytes, B, H, W, C, st);
    return sd::wgrad_launch<sd::bf16_t>(x, dy, dw, dbias, workspace, workspace_bytes, B, H, W, C, st);
}

}  // extern "C"
