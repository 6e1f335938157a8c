// batchnorm.hip -- training-mode BatchNorm over token-major activations [rows, C] with the ReLU and the channel
// dropout that follow it fused in, forward and backward, gfx950.
//
// The SegFormer head's `linear_fuse` is conv -> (Sync)BatchNorm -> ReLU, followed by Dropout2d and the classifier
// (reference segformer_head.py:66-71,93-96; decode_head.py:210-215).  At config 2 the normalised map is
// [8,256,128,128] = 134 MB, held as tokens [131072, 256] (a channels-last view).  Every piece is HBM-bound byte work:
//   statistics     read x                      (134 MB)
//   apply          read x, write y             (268 MB)   y = relu((x - mean) * invstd * w + b) * drop[b, c]
//   backward sums  read x, dy                  (268 MB)   g = dy * drop * [z > 0];  sum g, sum g * (x - mean)
//   backward dx    read x, dy, write dx        (402 MB)   dx = (g - sum_g / N - (x - mean) * invstd^2 * sum_gx / N) * invstd * w
// ~0.2 ms at 5.5 TB/s.  The library route costs ~1.1 ms: ATen's channels-last batch-norm kernels (the ones
// torch.nn.SyncBatchNorm is made of) run 328 us (statistics) and 446 us (backward sums) on this shape, plus separate
// ReLU and dropout passes each way.  The pieces are separate entry points because with more than one rank a
// collective sits between statistics and apply (all-gather of mean / invstd / count) and between the backward sums
// and dx (all-reduce): the binding -- or the segmented hipGraph capture -- issues it there.
//
// Mapping: lane j of a row group owns channels 4j..4j+3 (16-byte accesses, a row of C channels is contiguous), the
// 256 / CVp row groups of a workgroup walk rows with a grid stride; per-channel sums are combined through LDS and
// written as per-workgroup partials [nblk][2][C]; a second tiny kernel combines them in fp64 (deterministic, no
// float atomics).  Statistics are accumulated around a PIVOT (row 0 of the tensor) so that E[(x-p)^2] - E[x-p]^2 does
// not cancel when |mean| >> std.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sd_common.h"

namespace sd {

namespace {

constexpr int kBnThreads = 256;

template <typename T> struct BV;
template <> struct BV<float> {
    static __device__ __forceinline__ float4 load(const float *p) { return *reinterpret_cast<const float4 *>(p); }
    static __device__ __forceinline__ void store(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
};
template <> struct BV<bf16_t> {
    static __device__ __forceinline__ float4 load(const bf16_t *p) {
        const uint2 v = *reinterpret_cast<const uint2 *>(p);
        return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                           __uint_as_float(v.y & 0xffff0000u));
    }
    static __device__ __forceinline__ void store(bf16_t *p, float4 v) {
        uint2 o;
        o.x = (unsigned)f32_to_bf16(v.x) | ((unsigned)f32_to_bf16(v.y) << 16);
        o.y = (unsigned)f32_to_bf16(v.z) | ((unsigned)f32_to_bf16(v.w) << 16);
        *reinterpret_cast<uint2 *>(p) = o;
    }
};

__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// combine the row groups of a workgroup: a, b are this thread's sums for channels 4j..4j+3 -> part[blk][2][C]
__device__ __forceinline__ void block_partials(float4 a, float4 b, float *red, float *part, int j, int rg, int rpb, int cv, int C) {
    if (j < cv) {
        float *pa = red + ((size_t)rg * 2 + 0) * C + 4 * j, *pb = red + ((size_t)rg * 2 + 1) * C + 4 * j;
        pa[0] = a.x; pa[1] = a.y; pa[2] = a.z; pa[3] = a.w;
        pb[0] = b.x; pb[1] = b.y; pb[2] = b.z; pb[3] = b.w;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * C; e += kBnThreads) {
        float s = 0.f;
        for (int r = 0; r < rpb; ++r) s += red[(size_t)r * 2 * C + e];
        part[(size_t)blockIdx.x * 2 * C + e] = s;
    }
}

// ---- statistics --------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBnThreads) void bn_stats_partials(const T *__restrict__ x, float *__restrict__ part, long rows, int C,
                                                                 int cvp) {
    extern __shared__ float red[];   // [rpb][2][C]
    const int j = threadIdx.x & (cvp - 1), rg = threadIdx.x / cvp, rpb = kBnThreads / cvp, cv = C / 4;
    const bool lane = j < cv;
    const float4 p = lane ? BV<T>::load(x + 4 * j) : f4(0.f);
    float4 s = f4(0.f), q = f4(0.f);
    const long stride = (long)gridDim.x * rpb;
    long row = (long)blockIdx.x * rpb + rg;
    if (lane) {
        for (; row + 3 * stride < rows; row += 4 * stride) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = BV<T>::load(x + (row + u * stride) * C + 4 * j);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float a = v[u].x - p.x, b = v[u].y - p.y, c = v[u].z - p.z, d = v[u].w - p.w;
                s.x += a; s.y += b; s.z += c; s.w += d;
                q.x = fmaf(a, a, q.x); q.y = fmaf(b, b, q.y); q.z = fmaf(c, c, q.z); q.w = fmaf(d, d, q.w);
            }
        }
        for (; row < rows; row += stride) {
            const float4 v = BV<T>::load(x + row * C + 4 * j);
            const float a = v.x - p.x, b = v.y - p.y, c = v.z - p.z, d = v.w - p.w;
            s.x += a; s.y += b; s.z += c; s.w += d;
            q.x = fmaf(a, a, q.x); q.y = fmaf(b, b, q.y); q.z = fmaf(c, c, q.z); q.w = fmaf(d, d, q.w);
        }
    }
    block_partials(s, q, red, part, j, rg, rpb, cv, C);
}

// one workgroup per 16 channels; 16 thread groups stride over the partials; fp64 combination.
// mode 0: statistics  -> out1 = mean, out2 = invstd (+ running statistics when given)
// mode 1: plain sums  -> out1 = sum of entry 0, out2 = sum of entry 1
template <typename T>
__global__ __launch_bounds__(256) void bn_finalize(const float *__restrict__ part, int nblk, int C, int mode, const T *__restrict__ x,
                                                    long rows, float eps, float momentum, float *__restrict__ out1,
                                                    float *__restrict__ out2, float *__restrict__ running_mean,
                                                    float *__restrict__ running_var) {
    __shared__ double red[2][16][17];
    const int o = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + o;
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int p = grp; p < nblk; p += 16) {
            a += (double)part[(size_t)p * 2 * C + c];
            b += (double)part[(size_t)p * 2 * C + C + c];
        }
    red[0][grp][o] = a;
    red[1][grp][o] = b;
    __syncthreads();
    if (grp != 0 || c >= C) return;
    a = 0.0; b = 0.0;
#pragma unroll
    for (int g2 = 0; g2 < 16; ++g2) { a += red[0][g2][o]; b += red[1][g2][o]; }
    if (mode == 1) {
        out1[c] = (float)a;
        out2[c] = (float)b;
        return;
    }
    double pivot;
    if constexpr (sizeof(T) == 2) pivot = (double)__uint_as_float((unsigned)reinterpret_cast<const uint16_t *>(x)[c] << 16);
    else pivot = (double)reinterpret_cast<const float *>(x)[c];
    const double n = (double)rows, m1 = a / n;
    double var = b / n - m1 * m1;
    if (var < 0.0) var = 0.0;
    const double mean = pivot + m1;
    out1[c] = (float)mean;
    out2[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
    if (running_var) running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * (rows > 1 ? var * n / (n - 1.0) : var));
}

// ---- apply / backward --------------------------------------------------------------------------------------------------
struct ChanConst {   // per-lane constants of channels 4j..4j+3
    float4 mean, k, sh;
};
__device__ __forceinline__ ChanConst chan_const(const float *mean, const float *invstd, const float *w, const float *b, int j) {
    ChanConst c;
    c.mean = ld4(mean + 4 * j);
    const float4 is = ld4(invstd + 4 * j);
    const float4 wv = w ? ld4(w + 4 * j) : f4(1.f), bv = b ? ld4(b + 4 * j) : f4(0.f);
    c.k = make_float4(is.x * wv.x, is.y * wv.y, is.z * wv.z, is.w * wv.w);
    c.sh = make_float4(bv.x - c.mean.x * c.k.x, bv.y - c.mean.y * c.k.y, bv.z - c.mean.z * c.k.z, bv.w - c.mean.w * c.k.w);
    return c;
}

template <typename T, bool RELU>
__global__ __launch_bounds__(kBnThreads) void bn_act_fwd(const T *__restrict__ x, const float *__restrict__ mean,
                                                          const float *__restrict__ invstd, const float *__restrict__ w,
                                                          const float *__restrict__ b, const float *__restrict__ drop, long rpi,
                                                          T *__restrict__ y, long rows, int C, int cvp) {
    const int j = threadIdx.x & (cvp - 1), rg = threadIdx.x / cvp, rpb = kBnThreads / cvp;
    if (j >= C / 4) return;
    const ChanConst cc = chan_const(mean, invstd, w, b, j);
    const long stride = (long)gridDim.x * rpb;
    for (long row = (long)blockIdx.x * rpb + rg; row < rows; row += stride) {
        const float4 v = BV<T>::load(x + row * C + 4 * j);
        float4 z = make_float4(fmaf(v.x, cc.k.x, cc.sh.x), fmaf(v.y, cc.k.y, cc.sh.y), fmaf(v.z, cc.k.z, cc.sh.z), fmaf(v.w, cc.k.w, cc.sh.w));
        if (RELU) z = make_float4(fmaxf(z.x, 0.f), fmaxf(z.y, 0.f), fmaxf(z.z, 0.f), fmaxf(z.w, 0.f));
        if (drop) {
            const float4 d = ld4(drop + (row / rpi) * C + 4 * j);
            z.x *= d.x; z.y *= d.y; z.z *= d.z; z.w *= d.w;
        }
        BV<T>::store(y + row * C + 4 * j, z);
    }
}

// g = dy * drop * [z > 0] for one 4-channel vector
template <bool RELU>
__device__ __forceinline__ float4 masked_grad(float4 v, float4 dv, const ChanConst &cc, const float *drop, long img, int C, int j) {
    float4 g = dv;
    if (drop) {
        const float4 d = ld4(drop + img * C + 4 * j);
        g.x *= d.x; g.y *= d.y; g.z *= d.z; g.w *= d.w;
    }
    if (RELU) {
        if (!(fmaf(v.x, cc.k.x, cc.sh.x) > 0.f)) g.x = 0.f;
        if (!(fmaf(v.y, cc.k.y, cc.sh.y) > 0.f)) g.y = 0.f;
        if (!(fmaf(v.z, cc.k.z, cc.sh.z) > 0.f)) g.z = 0.f;
        if (!(fmaf(v.w, cc.k.w, cc.sh.w) > 0.f)) g.w = 0.f;
    }
    return g;
}

template <typename T, bool RELU>
__global__ __launch_bounds__(kBnThreads) void bn_bwd_partials(const T *__restrict__ x, const T *__restrict__ dy,
                                                               const float *__restrict__ mean, const float *__restrict__ invstd,
                                                               const float *__restrict__ w, const float *__restrict__ b,
                                                               const float *__restrict__ drop, long rpi, float *__restrict__ part, long rows,
                                                               int C, int cvp) {
    extern __shared__ float red[];
    const int j = threadIdx.x & (cvp - 1), rg = threadIdx.x / cvp, rpb = kBnThreads / cvp, cv = C / 4;
    float4 a1 = f4(0.f), a2 = f4(0.f);
    if (j < cv) {
        const ChanConst cc = chan_const(mean, invstd, w, b, j);
        const long stride = (long)gridDim.x * rpb;
        long row = (long)blockIdx.x * rpb + rg;
        // four rows (eight 16-byte loads) requested before any is folded -- two were: 62 us against a 42 us floor on the 134 MB map of
        // config 2; the pairs are folded exactly as two iterations of the two-row loop below would fold them (bit-identical sums)
        for (; row + 3 * stride < rows; row += 4 * stride) {
            float4 v[4], d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = BV<T>::load(x + (row + u * stride) * C + 4 * j);
                d[u] = BV<T>::load(dy + (row + u * stride) * C + 4 * j);
            }
#pragma unroll
            for (int u = 0; u < 4; u += 2) {
                const float4 g0 = masked_grad<RELU>(v[u], d[u], cc, drop, (row + u * stride) / rpi, C, j);
                const float4 g1 = masked_grad<RELU>(v[u + 1], d[u + 1], cc, drop, (row + (u + 1) * stride) / rpi, C, j);
                a1.x += g0.x + g1.x; a1.y += g0.y + g1.y; a1.z += g0.z + g1.z; a1.w += g0.w + g1.w;
                a2.x = fmaf(g0.x, v[u].x - cc.mean.x, a2.x); a2.y = fmaf(g0.y, v[u].y - cc.mean.y, a2.y);
                a2.z = fmaf(g0.z, v[u].z - cc.mean.z, a2.z); a2.w = fmaf(g0.w, v[u].w - cc.mean.w, a2.w);
                a2.x = fmaf(g1.x, v[u + 1].x - cc.mean.x, a2.x); a2.y = fmaf(g1.y, v[u + 1].y - cc.mean.y, a2.y);
                a2.z = fmaf(g1.z, v[u + 1].z - cc.mean.z, a2.z); a2.w = fmaf(g1.w, v[u + 1].w - cc.mean.w, a2.w);
            }
        }
        for (; row + stride < rows; row += 2 * stride) {
            const float4 v0 = BV<T>::load(x + row * C + 4 * j), d0 = BV<T>::load(dy + row * C + 4 * j);
            const float4 v1 = BV<T>::load(x + (row + stride) * C + 4 * j), d1 = BV<T>::load(dy + (row + stride) * C + 4 * j);
            const float4 g0 = masked_grad<RELU>(v0, d0, cc, drop, row / rpi, C, j);
            const float4 g1 = masked_grad<RELU>(v1, d1, cc, drop, (row + stride) / rpi, C, j);
            a1.x += g0.x + g1.x; a1.y += g0.y + g1.y; a1.z += g0.z + g1.z; a1.w += g0.w + g1.w;
            a2.x = fmaf(g0.x, v0.x - cc.mean.x, a2.x); a2.y = fmaf(g0.y, v0.y - cc.mean.y, a2.y);
            a2.z = fmaf(g0.z, v0.z - cc.mean.z, a2.z); a2.w = fmaf(g0.w, v0.w - cc.mean.w, a2.w);
            a2.x = fmaf(g1.x, v1.x - cc.mean.x, a2.x); a2.y = fmaf(g1.y, v1.y - cc.mean.y, a2.y);
            a2.z = fmaf(g1.z, v1.z - cc.mean.z, a2.z); a2.w = fmaf(g1.w, v1.w - cc.mean.w, a2.w);
        }
        for (; row < rows; row += stride) {
            const float4 v0 = BV<T>::load(x + row * C + 4 * j), d0 = BV<T>::load(dy + row * C + 4 * j);
            const float4 g0 = masked_grad<RELU>(v0, d0, cc, drop, row / rpi, C, j);
            a1.x += g0.x; a1.y += g0.y; a1.z += g0.z; a1.w += g0.w;
            a2.x = fmaf(g0.x, v0.x - cc.mean.x, a2.x); a2.y = fmaf(g0.y, v0.y - cc.mean.y, a2.y);
            a2.z = fmaf(g0.z, v0.z - cc.mean.z, a2.z); a2.w = fmaf(g0.w, v0.w - cc.mean.w, a2.w);
        }
    }
    block_partials(a1, a2, red, part, j, rg, rpb, cv, C);
}

template <typename T, bool RELU>
__global__ __launch_bounds__(kBnThreads) void bn_bwd_elemt(const T *__restrict__ x, const T *__restrict__ dy, const float *__restrict__ mean,
                                                            const float *__restrict__ invstd, const float *__restrict__ w,
                                                            const float *__restrict__ b, const float *__restrict__ drop, long rpi,
                                                            const float *__restrict__ sum_dy, const float *__restrict__ sum_dy_xmu,
                                                            float inv_count, const float *__restrict__ inv_count_dev, T *__restrict__ dx,
                                                            long rows, int C, int cvp) {
    const int j = threadIdx.x & (cvp - 1), rg = threadIdx.x / cvp, rpb = kBnThreads / cvp;
    if (j >= C / 4) return;
    if (inv_count_dev) inv_count = *inv_count_dev;
    const ChanConst cc = chan_const(mean, invstd, w, b, j);
    const float4 is = ld4(invstd + 4 * j), sd = ld4(sum_dy + 4 * j), sx = ld4(sum_dy_xmu + 4 * j);
    const float4 c1 = make_float4(sd.x * inv_count, sd.y * inv_count, sd.z * inv_count, sd.w * inv_count);
    const float4 c2 = make_float4(is.x * is.x * sx.x * inv_count, is.y * is.y * sx.y * inv_count, is.z * is.z * sx.z * inv_count,
                                  is.w * is.w * sx.w * inv_count);
    const long stride = (long)gridDim.x * rpb;
    for (long row = (long)blockIdx.x * rpb + rg; row < rows; row += stride) {
        const float4 v = BV<T>::load(x + row * C + 4 * j), dv = BV<T>::load(dy + row * C + 4 * j);
        const float4 g = masked_grad<RELU>(v, dv, cc, drop, row / rpi, C, j);
        float4 o;
        o.x = (g.x - c1.x - (v.x - cc.mean.x) * c2.x) * cc.k.x;
        o.y = (g.y - c1.y - (v.y - cc.mean.y) * c2.y) * cc.k.y;
        o.z = (g.z - c1.z - (v.z - cc.mean.z) * c2.z) * cc.k.z;
        o.w = (g.w - c1.w - (v.w - cc.mean.w) * c2.w) * cc.k.w;
        BV<T>::store(dx + row * C + 4 * j, o);
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
int bn_cvp(int C) {
    int p = 1;
    while (p < C / 4) p <<= 1;
    return p;
}
int bn_reduce_blocks(long rows, int cvp) {   // workgroups of the two reducing kernels (= partial slabs)
    const long rpb = kBnThreads / cvp;
    long n = (rows + rpb * 32 - 1) / (rpb * 32);
    if (n > 512) n = 512;
    return (int)(n < 1 ? 1 : n);
}
int bn_stream_blocks(long rows, int cvp) {   // workgroups of the two element-wise kernels
    const long rpb = kBnThreads / cvp;
    long n = (rows + rpb * 8 - 1) / (rpb * 8);
    if (n > 4096) n = 4096;
    return (int)(n < 1 ? 1 : n);
}
int bn_check(const void *a, const void *b, int dtype, long rows, int C) {
    if (!a || !b) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (rows <= 0 || C <= 0) return SD_E_SHAPE;
    if (C % 4 || C > 1024) return SD_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) return SD_E_ALIGN;
    return SD_OK;
}
int bn_check_drop(const float *drop, long rpi, long rows) {
    if (drop && (rpi <= 0 || rows % rpi)) return SD_E_SHAPE;
    return SD_OK;
}

}  // namespace
}  // namespace sd

extern "C" {

int sd_bn_supported(int C) { return (C > 0 && C % 4 == 0 && C <= 1024) ? 1 : 0; }

size_t sd_bn_workspace_bytes(long rows, int C) {
    if (rows <= 0 || !sd_bn_supported(C)) return 0;
    return (size_t)sd::bn_reduce_blocks(rows, sd::bn_cvp(C)) * 2 * C * sizeof(float) + 16;
}

int sd_bn_stats(const void *x, int dtype, long rows, int C, float eps, float *mean, float *invstd, float *running_mean,
                float *running_var, float momentum, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = sd::bn_check(x, mean, dtype, rows, C);
    if (rc) return rc;
    if (!invstd || !workspace) return SD_E_NULL;
    const int cvp = sd::bn_cvp(C), nblk = sd::bn_reduce_blocks(rows, cvp);
    if (workspace_bytes < (size_t)nblk * 2 * C * sizeof(float) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return SD_E_WORKSPACE;
    float *part = static_cast<float *>(workspace);
    const size_t lds = (size_t)(sd::kBnThreads / cvp) * 2 * C * sizeof(float);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32) {
        hipLaunchKernelGGL(sd::bn_stats_partials<float>, dim3(nblk), dim3(sd::kBnThreads), lds, st, (const float *)x, part, rows, C, cvp);
        hipLaunchKernelGGL(sd::bn_finalize<float>, dim3((C + 15) / 16), dim3(256), 0, st, part, nblk, C, 0, (const float *)x, rows, eps,
                           momentum, mean, invstd, running_mean, running_var);
    } else {
        hipLaunchKernelGGL(sd::bn_stats_partials<sd::bf16_t>, dim3(nblk), dim3(sd::kBnThreads), lds, st, (const sd::bf16_t *)x, part, rows, C,
                           cvp);
        hipLaunchKernelGGL(sd::bn_finalize<sd::bf16_t>, dim3((C + 15) / 16), dim3(256), 0, st, part, nblk, C, 0, (const sd::bf16_t *)x, rows,
                           eps, momentum, mean, invstd, running_mean, running_var);
    }
    return (int)hipGetLastError();
}

int sd_bn_act_fwd(const void *x, const float *mean, const float *invstd, const float *weight, const float *bias, const float *drop_scale,
                  long rows_per_image, int relu, void *y, int dtype, long rows, int C, void *stream) {
    int rc = sd::bn_check(x, y, dtype, rows, C);
    if (rc) return rc;
    if (!mean || !invstd) return SD_E_NULL;
    rc = sd::bn_check_drop(drop_scale, rows_per_image, rows);
    if (rc) return rc;
    const int cvp = sd::bn_cvp(C), nblk = sd::bn_stream_blocks(rows, cvp);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define SD_BN_FWD(TT, RR)                                                                                                            \
    hipLaunchKernelGGL((sd::bn_act_fwd<TT, RR>), dim3(nblk), dim3(sd::kBnThreads), 0, st, (const TT *)x, mean, invstd, weight, bias, \
                       drop_scale, rows_per_image, (TT *)y, rows, C, cvp)
    if (dtype == SD_F32) { if (relu) SD_BN_FWD(float, true); else SD_BN_FWD(float, false); }
    else { if (relu) SD_BN_FWD(sd::bf16_t, true); else SD_BN_FWD(sd::bf16_t, false); }
#undef SD_BN_FWD
    return (int)hipGetLastError();
}

int sd_bn_act_bwd_reduce(const void *x, const void *dy, const float *mean, const float *invstd, const float *weight, const float *bias,
                         const float *drop_scale, long rows_per_image, int relu, float *sum_dy, float *sum_dy_xmu, int dtype, long rows,
                         int C, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = sd::bn_check(x, dy, dtype, rows, C);
    if (rc) return rc;
    if (!mean || !invstd || !sum_dy || !sum_dy_xmu || !workspace) return SD_E_NULL;
    rc = sd::bn_check_drop(drop_scale, rows_per_image, rows);
    if (rc) return rc;
    const int cvp = sd::bn_cvp(C), nblk = sd::bn_reduce_blocks(rows, cvp);
    if (workspace_bytes < (size_t)nblk * 2 * C * sizeof(float) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return SD_E_WORKSPACE;
    float *part = static_cast<float *>(workspace);
    const size_t lds = (size_t)(sd::kBnThreads / cvp) * 2 * C * sizeof(float);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define SD_BN_RED(TT, RR)                                                                                                                  \
    hipLaunchKernelGGL((sd::bn_bwd_partials<TT, RR>), dim3(nblk), dim3(sd::kBnThreads), lds, st, (const TT *)x, (const TT *)dy, mean, invstd, \
                       weight, bias, drop_scale, rows_per_image, part, rows, C, cvp)
    if (dtype == SD_F32) { if (relu) SD_BN_RED(float, true); else SD_BN_RED(float, false); }
    else { if (relu) SD_BN_RED(sd::bf16_t, true); else SD_BN_RED(sd::bf16_t, false); }
#undef SD_BN_RED
    hipLaunchKernelGGL(sd::bn_finalize<float>, dim3((C + 15) / 16), dim3(256), 0, st, part, nblk, C, 1, (const float *)nullptr, rows, 0.f, 0.f,
                       sum_dy, sum_dy_xmu, (float *)nullptr, (float *)nullptr);
    return (int)hipGetLastError();
}

int sd_bn_act_bwd_elemt(const void *x, const void *dy, const float *mean, const float *invstd, const float *weight, const float *bias,
                        const float *drop_scale, long rows_per_image, int relu, const float *sum_dy, const float *sum_dy_xmu,
                        float inv_count, const float *inv_count_dev, void *dx, int dtype, long rows, int C, void *stream) {
    int rc = sd::bn_check(x, dy, dtype, rows, C);
    if (rc) return rc;
    if (!mean || !invstd || !sum_dy || !sum_dy_xmu || !dx) return SD_E_NULL;
    if (reinterpret_cast<uintptr_t>(dx) & 15) return SD_E_ALIGN;
    rc = sd::bn_check_drop(drop_scale, rows_per_image, rows);
    if (rc) return rc;
    const int cvp = sd::bn_cvp(C), nblk = sd::bn_stream_blocks(rows, cvp);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define SD_BN_DX(TT, RR)                                                                                                                \
    hipLaunchKernelGGL((sd::bn_bwd_elemt<TT, RR>), dim3(nblk), dim3(sd::kBnThreads), 0, st, (const TT *)x, (const TT *)dy, mean, invstd,  \
                       weight, bias, drop_scale, rows_per_image, sum_dy, sum_dy_xmu, inv_count, inv_count_dev, (TT *)dx, rows, C, cvp)
    if (dtype == SD_F32) { if (relu) SD_BN_DX(float, true); else SD_BN_DX(float, false); }
    else { if (relu) SD_BN_DX(sd::bf16_t, true); else SD_BN_DX(sd::bf16_t, false); }
#undef SD_BN_DX
    return (int)hipGetLastError();
}

}  // extern "C"
