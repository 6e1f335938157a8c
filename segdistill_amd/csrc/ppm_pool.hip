// ppm_pool.hip -- the adaptive average pools of a Pyramid Pooling Module, ALL pool scales in one pass over the map, forward and backward, gfx950.
//
// reference mmseg/models/decode_heads/psp_head.py:10-58 (PPM: one nn.AdaptiveAvgPool2d(s) per pool scale, s = 1, 2, 3, 6, each followed by
// a 1x1 ConvModule and a bilinear resize back), used by PSPHead :61-101 and UPerHead (uper_head.py:76-126).  On PyTorch-ROCm each scale is
// its own pass over the [B, C, h, w] map at ~0.3 TB/s (`adaptive_average_pool`: 225 us for 67 MB at config 4), and each backward is
// `atomic_adaptive_average_gradinput` -- float atomics, ~0.9 ms per scale, the only non-deterministic kernel of an otherwise deterministic
// step (profiles/r03_train_step_kernels_cfg4.txt: 2.78 ms of a 78 ms step) -- followed by the adds of the four branch gradients.
// Here: one workgroup per (image, channel) plane.
//   forward : the plane is staged in LDS once (coalesced 16-byte loads); separable sums -- rows first (y, bin j), then columns -- in a fixed
//             order; every scale's [s x s] map is written from that one read.
//   backward: the plane's ~50 upstream values (pre-divided by their bin areas) go to LDS, are spread along y into a [h][sum s] table, and
//             every pixel GATHERS the <= 2 bins per scale and axis that contain it; dx is written once, 16 bytes per lane.  No atomics:
//             deterministic, and the four branch gradients never exist as separate maps.
// Bins as ATen's: start = floor(i h / s), end = ceil((i + 1) h / s) (they overlap by a row when s does not divide h).
// HBM-bound: forward reads the map once (+ a few hundred bytes per plane), backward writes it once.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cgd_device.h"

namespace sd {
namespace {

constexpr int kPpmMaxScales = 4, kPpmMaxBins = 8, kPpmMaxQ = 24;      // sum of the scales of one call <= kPpmMaxQ
constexpr int kPpmMaxPlane = 12288;                                    // floats of one plane in LDS (48 KB: three workgroups per CU)

struct PpmArgs {
    void *out[kPpmMaxScales];          // forward: pooled maps [BC][s][s]; backward: upstream gradients of those maps
    int s[kPpmMaxScales];
    int qoff[kPpmMaxScales];           // prefix sums of s
    int n, Q;                          // number of scales, sum of s
};

__device__ __forceinline__ int bin_start(int i, int n, int s) { return (i * n) / s; }
__device__ __forceinline__ int bin_end(int i, int n, int s) { return ((i + 1) * n + s - 1) / s; }

template <typename T>
__global__ __launch_bounds__(256) void ppm_pool_fwd(const T *__restrict__ x, const PpmArgs a, int h, int w) {
    extern __shared__ __attribute__((aligned(16))) float ppm_lds[];    // plane [h][w + 1] | rowsum [h][Q]
    const int pitch = w + 1;
    float *plane = ppm_lds, *rows = ppm_lds + (size_t)h * pitch;
    const size_t bc = blockIdx.x;
    const T *px = x + bc * (size_t)h * w;
    const int t = threadIdx.x, hw = h * w;
    constexpr int N = VecIO<T>::N;
    if (w % N == 0 && (reinterpret_cast<uintptr_t>(px) & 15) == 0) {
        // a 16-byte vector never crosses a row (w % N == 0): ONE division per vector -- the first version divided per element and was bound by
        // that index arithmetic (49 us for a 67 MB map: ~3000 vector instructions per wave against 16 loads)
        for (int e = t * N; e < hw; e += 256 * N) {
            float v[N];
            VecIO<T>::load(px + e, v);
            const int y = e / w, x0 = e - y * w;
            float *dst = plane + y * pitch + x0;
#pragma unroll
            for (int i = 0; i < N; ++i) dst[i] = v[i];
        }
    } else {
        for (int e = t; e < hw; e += 256) plane[(e / w) * pitch + e % w] = VecIO<T>::load1(px + e);
    }
    __syncthreads();
    // rows: (y, scale k, bin j) -> sum of plane[y][sx .. ex)
    const int Q = a.Q;
    for (int task = t; task < h * Q; task += 256) {
        const int y = task / Q, q = task - y * Q;
        int k = 0;
        while (k + 1 < a.n && q >= a.qoff[k + 1]) ++k;
        const int s = a.s[k], j = q - a.qoff[k];
        const int sx = bin_start(j, w, s), ex = bin_end(j, w, s);
        float acc = 0.f;
        for (int xx = sx; xx < ex; ++xx) acc += plane[y * pitch + xx];
        rows[y * Q + q] = acc;
    }
    __syncthreads();
    // columns: (scale k, i, j) -> sum over y in [sy .. ey) of rows[y][q] / area
    int total = 0;
    for (int k = 0; k < a.n; ++k) total += a.s[k] * a.s[k];
    for (int task = t; task < total; task += 256) {
        int k = 0, rem = task;
        while (rem >= a.s[k] * a.s[k]) { rem -= a.s[k] * a.s[k]; ++k; }
        const int s = a.s[k], i = rem / s, j = rem - i * s;
        const int sy = bin_start(i, h, s), ey = bin_end(i, h, s), sx = bin_start(j, w, s), ex = bin_end(j, w, s);
        float acc = 0.f;
        for (int y = sy; y < ey; ++y) acc += rows[y * Q + a.qoff[k] + j];
        VecIO<T>::store1(static_cast<T *>(a.out[k]) + bc * (size_t)(s * s) + rem, acc / (float)((ey - sy) * (ex - sx)));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void ppm_pool_bwd(const PpmArgs a, T *__restrict__ dx, int h, int w) {
    extern __shared__ __attribute__((aligned(16))) float ppm_lds[];    // g [sum s^2] | ycol [h][Q] | xlo, xn [n][w]
    const size_t bc = blockIdx.x;
    const int t = threadIdx.x, Q = a.Q;
    int total = 0;
    for (int k = 0; k < a.n; ++k) total += a.s[k] * a.s[k];
    float *g = ppm_lds, *ycol = g + ((total + 3) & ~3);
    int *xlo = reinterpret_cast<int *>(ycol + (size_t)h * Q), *xn = xlo + a.n * w;
    for (int task = t; task < total; task += 256) {
        int k = 0, rem = task;
        while (rem >= a.s[k] * a.s[k]) { rem -= a.s[k] * a.s[k]; ++k; }
        const int s = a.s[k], i = rem / s, j = rem - i * s;
        const int area = (bin_end(i, h, s) - bin_start(i, h, s)) * (bin_end(j, w, s) - bin_start(j, w, s));
        g[task] = VecIO<T>::load1(static_cast<const T *>(a.out[k]) + bc * (size_t)(s * s) + rem) / (float)area;
    }
    // per (scale, x): the first bin that contains column x and how many do (1 or 2)
    for (int task = t; task < a.n * w; task += 256) {
        const int k = task / w, xx = task - k * w, s = a.s[k];
        int lo = (xx * s) / w;                                       // a bin that contains xx; the previous one may as well
        if (lo > 0 && bin_end(lo - 1, w, s) > xx) --lo;
        int cnt = 1;
        if (lo + 1 < s && bin_start(lo + 1, w, s) <= xx) cnt = 2;
        xlo[task] = lo;
        xn[task] = cnt;
    }
    __syncthreads();
    // ycol[y][q = (k, j)] = sum over the bins i of scale k that contain row y of g_k[i][j]   (ascending i: fixed order)
    int goff[kPpmMaxScales];
    {
        int o = 0;
        for (int k = 0; k < a.n; ++k) { goff[k] = o; o += a.s[k] * a.s[k]; }
    }
    for (int task = t; task < h * Q; task += 256) {
        const int y = task / Q, q = task - y * Q;
        int k = 0;
        while (k + 1 < a.n && q >= a.qoff[k + 1]) ++k;
        const int s = a.s[k], j = q - a.qoff[k];
        int lo = (y * s) / h;
        if (lo > 0 && bin_end(lo - 1, h, s) > y) --lo;
        float acc = g[goff[k] + lo * s + j];
        if (lo + 1 < s && bin_start(lo + 1, h, s) <= y) acc += g[goff[k] + (lo + 1) * s + j];
        ycol[task] = acc;
    }
    __syncthreads();
    T *pd = dx + bc * (size_t)h * w;
    const int hw = h * w;
    constexpr int N = VecIO<T>::N;
    auto at = [&](int idx) {
        const int y = idx / w, xx = idx - y * w;
        float acc = 0.f;
        for (int k = 0; k < a.n; ++k) {
            const int lo = xlo[k * w + xx];
            const float *r = ycol + y * Q + a.qoff[k] + lo;
            acc += r[0];
            if (xn[k * w + xx] == 2) acc += r[1];
        }
        return acc;
    };
    if (w % N == 0 && (reinterpret_cast<uintptr_t>(pd) & 15) == 0) {
        for (int e = t * N; e < hw; e += 256 * N) {             // one division per 16-byte vector (it lies inside one row)
            const int y = e / w, x0 = e - y * w;
            float v[N];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                float acc = 0.f;
                for (int k = 0; k < a.n; ++k) {
                    const int lo = xlo[k * w + x0 + i];
                    const float *r = ycol + y * Q + a.qoff[k] + lo;
                    acc += r[0];
                    if (xn[k * w + x0 + i] == 2) acc += r[1];
                }
                v[i] = acc;
            }
            VecIO<T>::template store<false>(pd + e, v);
        }
    } else {
        for (int e = t; e < hw; e += 256) VecIO<T>::store1(pd + e, at(e));
    }
}

int fill_args(PpmArgs &a, void *const *maps, const int *scales, int nscales) {
    if (nscales <= 0 || nscales > kPpmMaxScales) return SD_E_UNSUPPORTED;
    a.n = nscales;
    a.Q = 0;
    for (int k = 0; k < nscales; ++k) {
        if (!maps[k]) return SD_E_NULL;
        if (scales[k] <= 0 || scales[k] > kPpmMaxBins) return SD_E_UNSUPPORTED;
        a.out[k] = maps[k];
        a.s[k] = scales[k];
        a.qoff[k] = a.Q;
        a.Q += scales[k];
    }
    return a.Q <= kPpmMaxQ ? SD_OK : SD_E_UNSUPPORTED;
}

}  // namespace
}  // namespace sd

extern "C" {

int sd_ppm_pool_supported(int h, int w, const int *scales, int nscales) {
    if (h <= 0 || w <= 0 || !scales || nscales <= 0 || nscales > sd::kPpmMaxScales) return 0;
    int Q = 0;
    for (int k = 0; k < nscales; ++k) {
        if (scales[k] <= 0 || scales[k] > sd::kPpmMaxBins || scales[k] > h || scales[k] > w) return 0;
        Q += scales[k];
    }
    return ((long)h * (w + 1) + (long)h * Q <= sd::kPpmMaxPlane && Q <= sd::kPpmMaxQ) ? 1 : 0;
}

int sd_ppm_pool_fwd(const void *x, int dtype, long planes, int h, int w, const int *scales, int nscales, void *const *pooled, void *stream) {
    if (!x || !scales || !pooled) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (planes <= 0 || planes > 0x7fffffffL) return SD_E_SHAPE;
    if (!sd_ppm_pool_supported(h, w, scales, nscales)) return SD_E_UNSUPPORTED;
    sd::PpmArgs a;
    int rc = sd::fill_args(a, pooled, scales, nscales);
    if (rc) return rc;
    const size_t lds = ((size_t)h * (w + 1) + (size_t)h * a.Q) * sizeof(float);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32) hipLaunchKernelGGL((sd::ppm_pool_fwd<float>), dim3((unsigned)planes), dim3(256), lds, st, (const float *)x, a, h, w);
    else hipLaunchKernelGGL((sd::ppm_pool_fwd<sd::bf16_t>), dim3((unsigned)planes), dim3(256), lds, st, (const sd::bf16_t *)x, a, h, w);
    return (int)hipGetLastError();
}

int sd_ppm_pool_bwd(void *const *d_pooled, int dtype, long planes, int h, int w, const int *scales, int nscales, void *dx, void *stream) {
    if (!d_pooled || !scales || !dx) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (planes <= 0 || planes > 0x7fffffffL) return SD_E_SHAPE;
    if (!sd_ppm_pool_supported(h, w, scales, nscales)) return SD_E_UNSUPPORTED;
    sd::PpmArgs a;
    int rc = sd::fill_args(a, d_pooled, scales, nscales);
    if (rc) return rc;
    int total = 0;
    for (int k = 0; k < nscales; ++k) total += scales[k] * scales[k];
    const size_t lds = ((size_t)((total + 3) & ~3) + (size_t)h * a.Q) * sizeof(float) + 2 * (size_t)nscales * w * sizeof(int);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32) hipLaunchKernelGGL((sd::ppm_pool_bwd<float>), dim3((unsigned)planes), dim3(256), lds, st, a, (float *)dx, h, w);
    else hipLaunchKernelGGL((sd::ppm_pool_bwd<sd::bf16_t>), dim3((unsigned)planes), dim3(256), lds, st, a, (sd::bf16_t *)dx, h, w);
    return (int)hipGetLastError();
}

}  // extern "C"
