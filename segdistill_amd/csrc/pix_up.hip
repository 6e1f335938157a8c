// pix_up.hip -- the pixel-wise criterion (PDLoss) with the bilinear up-sampling FUSED in, gfx950.
//
// reference: KLDLoss.forward with transform_config loss_type 'pixel' (losses.py:47-49: softmax over the C classes at every pixel) behind
// KLDLoss.resize (losses.py:25-33,101-102: both logits to the label size, bilinear, align_corners=False) -- the PDLoss preset of
// losses.py:115-128 (exp_tab5/*_PD.py).  Rounds 1-4 materialised both [B,C,H,W] maps (2 x 1.26 GB at the config-2 taps) with resize.hip and ran
// pix_kl.hip on them (0.44 + 0.65 ms + two resizes + the resize backward).  Here, as in cgd_up.hip, the up-sampled values exist only in registers:
//   forward : a thread owns tap column kx (F output columns) of one y-gap (F output rows between tap rows j-1 and j): its F*F pixels keep one online
//             softmax state each (one exponential per element: pix_kl.hip's online update) while the thread walks the CLASS axis once, interpolating
//             both taps on the fly; out go the per-pixel base-2 log-partitions [2][B*H*W] (8 bytes per pixel: what the backward needs) and one
//             partial KL sum per workgroup.
//   backward: cgd_up.hip's transposed-interpolation gather (vertically in registers, horizontally through an LDS row; no atomics, deterministic),
//             with the row constants replaced by the PIXEL constants: a workgroup owns a band of tap rows and a block of CB classes, loads the
//             log-partitions of a gap's pixels once and reuses them for its CB classes (the maps are read C / CB times from L2, not C times).
// exp / VALU-bound like cgd_up; HBM traffic: the taps (2 * B*C*h*w*e each way) + the log-partition maps.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cgd_device.h"
#include "up_device.h"

namespace sd {
namespace {

// classes per backward workgroup (their gather state lives in registers: 3 F floats each)
template <int F> struct PuCB { static constexpr int CB = F == 2 ? 8 : (F == 4 ? 6 : 2); };

// Online update with ONE exp per element: e = 2^{-|x - m| c2} (pix_kl.hip).
__device__ __forceinline__ void pu_online1(float x, float &m, float &z, float c2) {
    const float d = x - m;
    const float e = ex2(-fabsf(d) * c2);
    z = d > 0.f ? fmaf(z, e, 1.f) : z + e;
    m = fmaxf(m, x);
}
__device__ __forceinline__ void pu_online2(float x, float diff, float &m, float &z, float &a, float c2) {
    const float d = x - m;
    const float e = ex2(-fabsf(d) * c2);
    if (d > 0.f) { z = fmaf(z, e, 1.f); a = fmaf(a, e, diff); }
    else { z += e; a = fmaf(e, diff, a); }
    m = fmaxf(m, x);
}

// grid = (h + 1 gaps, B, F / QR row slices); blockDim = tap width rounded up to 64.  Gap j: output rows Y = F*j - F/2 + q (rows outside the image
// skipped), this workgroup's q in [QR z, QR z + QR); the thread's columns X = F*kx + rx.  lse2: [2][B*H*W].
template <int F> struct PuGeo { static constexpr int QR = 2; };     // rows of a gap per workgroup: 2 F pixel states per thread (F = 4: 16 states were 174 registers, two waves per SIMD, 422 us)

template <typename T, int F, int NT>
__global__ __launch_bounds__(NT) void pix_up_fwd(const T *__restrict__ s, const T *__restrict__ t, float *__restrict__ lse2, double *__restrict__ wg_sum, int C, int h, int w,
                           float c2, float inv_tau) {
    const int j = blockIdx.x, b = blockIdx.y;
    const int kx = threadIdx.x;
    const bool active = kx < w;
    const int kxc = min(kx, w - 1);
    const int H = F * h, W = F * w;
    const long BHW = (long)gridDim.y * H * W;
    const int ra = max(j - 1, 0), rb = min(j, h - 1);         // the gap's two tap rows (equal in the half gaps at the top and the bottom)
    constexpr int QR = PuGeo<F>::QR;
    const int qz = (int)blockIdx.z * QR;
    const int q_lo = j == 0 ? F / 2 : 0, q_hi = j == h ? F / 2 : F;
    float ms[QR][F], zs[QR][F], mt[QR][F], zt[QR][F], a[QR][F];
#pragma unroll
    for (int q = 0; q < QR; ++q)
#pragma unroll
        for (int rx = 0; rx < F; ++rx) { ms[q][rx] = mt[q][rx] = kNegBig; zs[q][rx] = zt[q][rx] = a[q][rx] = 0.f; }
    const size_t plane = (size_t)h * w;
    const T *ps = s + (size_t)b * C * plane, *pt = t + (size_t)b * C * plane;
#pragma unroll 2
    for (int c = 0; c < C; ++c) {
        float sa[F], sb[F], ta[F], tb[F];
        hrow<T, F>(ps + c * plane + (size_t)ra * w, kxc, w, sa);
        hrow<T, F>(ps + c * plane + (size_t)rb * w, kxc, w, sb);
        hrow<T, F>(pt + c * plane + (size_t)ra * w, kxc, w, ta);
        hrow<T, F>(pt + c * plane + (size_t)rb * w, kxc, w, tb);
#pragma unroll
        for (int rx = 0; rx < F; ++rx) { sb[rx] -= sa[rx]; tb[rx] -= ta[rx]; }
#pragma unroll
        for (int q = 0; q < QR; ++q) {
            const float lam = ((float)(qz + q) + 0.5f) / F;
#pragma unroll
            for (int rx = 0; rx < F; ++rx) {
                const float S = fmaf(lam, sb[rx], sa[rx]), Tv = fmaf(lam, tb[rx], ta[rx]);
                pu_online1(S, ms[q][rx], zs[q][rx], c2);
                pu_online2(Tv, Tv - S, mt[q][rx], zt[q][rx], a[q][rx], c2);
            }
        }
    }
    float kl = 0.f;
    if (active) {
        const float ln2 = 0.69314718055994531f;
#pragma unroll
        for (int q = 0; q < QR; ++q) {
            if (qz + q < q_lo || qz + q >= q_hi) continue;
            const int Y = F * j - F / 2 + qz + q;
            float l2s[F], l2t[F];
#pragma unroll
            for (int rx = 0; rx < F; ++rx) {
                l2s[rx] = fmaf(ms[q][rx], c2, __builtin_amdgcn_logf(zs[q][rx]));
                l2t[rx] = fmaf(mt[q][rx], c2, __builtin_amdgcn_logf(zt[q][rx]));
                kl += a[q][rx] * inv_tau / zt[q][rx] + (l2s[rx] - l2t[rx]) * ln2;
            }
            float *o = lse2 + ((size_t)b * H + Y) * W + (size_t)F * kx;
#pragma unroll
            for (int rx = 0; rx < F; ++rx) { o[rx] = l2s[rx]; o[BHW + rx] = l2t[rx]; }
        }
    }
    __shared__ double acc[16];
    double v = (double)kl;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) acc[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) tot += acc[i];
        wg_sum[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = tot;
    }
}

__global__ __launch_bounds__(256) void pix_up_loss(const double *__restrict__ wg_sum, float *__restrict__ loss, int n, float loss_scale) {
    __shared__ double acc[4];
    double v = 0;
    for (int i = threadIdx.x; i < n; i += 256) v += wg_sum[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) acc[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (float)(((acc[0] + acc[1]) + acc[2] + acc[3]) * (double)loss_scale);
}

// grid = (nband, ceil(C / CB), B); the workgroup produces tap-gradient rows [y0, y1) of CB class planes.  It walks gaps j = y0 .. y1: gap j contributes
// to tap rows j-1 (weight 1 - lambda) and j (weight lambda); a finished tap row is transposed along x through LDS, class by class.
template <typename T, int F, int CB, int NT>
__global__ __launch_bounds__(NT) void pix_up_bwd(const T *__restrict__ s, const T *__restrict__ t, const float *__restrict__ lse2, const float *__restrict__ upstream,
                           T *__restrict__ ds, int C, int h, int w, int R, float c2, float coef) {
    extern __shared__ float pu_rowbuf[];                     // 2 buffers of F * blockDim.x floats
    const int k = blockIdx.x, c0 = blockIdx.y * CB, b = blockIdx.z;
    const int kx = threadIdx.x;
    const bool active = kx < w;
    const int kxc = min(kx, w - 1);
    const int y0 = k * R, y1 = min(h, y0 + R);
    const int H = F * h, W = F * w;
    const long BHW = (long)gridDim.z * H * W;
    const int bufstride = F * blockDim.x;
    const float kk = upstream ? coef * upstream[0] : coef;
    const size_t plane = (size_t)h * w;
    const T *ps = s + ((size_t)b * C + c0) * plane, *pt = t + ((size_t)b * C + c0) * plane;
    T *pd = ds + ((size_t)b * C + c0) * plane;
    const int ncl = min(CB, C - c0);                          // classes of this block that exist (workgroup-uniform)

    float sp[CB][F], tp[CB][F], accA[CB][F];
#pragma unroll
    for (int cc = 0; cc < CB; ++cc) {
        const int cl = min(cc, ncl - 1);                      // surplus slots repeat the last class (never stored)
        const int r = max(y0 - 1, 0);
        hrow<T, F>(ps + cl * plane + (size_t)r * w, kxc, w, sp[cc]);
        hrow<T, F>(pt + cl * plane + (size_t)r * w, kxc, w, tp[cc]);
#pragma unroll
        for (int rx = 0; rx < F; ++rx) accA[cc][rx] = 0.f;
    }
    int parity = 0;
    for (int j = y0; j <= y1; ++j) {
        const int r = min(j, h - 1);
        const bool top = (j == 0), bot = (j == h);
        // the gap's pixel constants, once for all CB classes (clamped addresses for the rows / columns outside the image: their weight is zero)
        float ls[F][F], lt[F][F];
#pragma unroll
        for (int q = 0; q < F; ++q) {
            const int Y = min(max(F * j - F / 2 + q, 0), H - 1);
            const float *o = lse2 + ((size_t)b * H + Y) * W + (size_t)F * kxc;
#pragma unroll
            for (int rx = 0; rx < F; ++rx) { ls[q][rx] = o[rx]; lt[q][rx] = o[BHW + rx]; }
        }
#pragma unroll
        for (int cc = 0; cc < CB; ++cc) {
            const int cl = min(cc, ncl - 1);
            float sc[F], tc[F], accB[F];
            hrow<T, F>(ps + cl * plane + (size_t)r * w, kxc, w, sc);
            hrow<T, F>(pt + cl * plane + (size_t)r * w, kxc, w, tc);
            float dsv[F], dtv[F];
#pragma unroll
            for (int rx = 0; rx < F; ++rx) { dsv[rx] = sc[rx] - sp[cc][rx]; dtv[rx] = tc[rx] - tp[cc][rx]; accB[rx] = 0.f; }
#pragma unroll
            for (int q = 0; q < F; ++q) {
                if ((top && q < F / 2) || (bot && q >= F / 2)) continue;      // rows outside the image
                const float lam = (q + 0.5f) / F;
                const float wa = top ? 0.f : (bot ? 1.f : 1.f - lam);
                const float wb = top ? 1.f : (bot ? 0.f : lam);
#pragma unroll
                for (int rx = 0; rx < F; ++rx) {
                    const float S = fmaf(lam, dsv[rx], sp[cc][rx]);
                    const float Tv = fmaf(lam, dtv[rx], tp[cc][rx]);
                    const float D = kk * (ex2(fmaf(S, c2, -ls[q][rx])) - ex2(fmaf(Tv, c2, -lt[q][rx])));
                    accA[cc][rx] = fmaf(wa, D, accA[cc][rx]);
                    accB[rx] = fmaf(wb, D, accB[rx]);
                }
            }
            if (j > y0) {                                     // tap row j-1 has received both of its gaps: transpose along x through LDS
                float *buf = pu_rowbuf + parity * bufstride;
#pragma unroll
                for (int rx = 0; rx < F; ++rx) buf[F * kx + rx] = accA[cc][rx];
                __syncthreads();
                if (active && cc < ncl) {
                    float sum = 0.f;
#pragma unroll
                    for (int q = 0; q < F; ++q) {
                        const float lam = (q + 0.5f) / F;
                        const int xl = F * kx - F / 2 + q;    // x-gap kx   : this column is the right tap
                        const int xr = F * kx + F / 2 + q;    // x-gap kx+1 : this column is the left tap
                        if (xl >= 0) sum = fmaf(kx == 0 ? 1.f : lam, buf[xl], sum);
                        if (xr < W) sum = fmaf(kx == w - 1 ? 1.f : 1.f - lam, buf[xr], sum);
                    }
                    VecIO<T>::store1(pd + cc * plane + (size_t)(j - 1) * w + kx, sum);
                }
                parity ^= 1;
            }
#pragma unroll
            for (int rx = 0; rx < F; ++rx) { accA[cc][rx] = accB[rx]; sp[cc][rx] = sc[rx]; tp[cc][rx] = tc[rx]; }
        }
    }
}

int pu_factor(int h, int w, int H, int W) {
    if (h <= 0 || w <= 0 || H % h || W % w) return 0;
    const int f = H / h;
    if (W / w != f || (f != 2 && f != 4 && f != 8)) return 0;
    if (w > 1024 || (long)f * ((w + 63) / 64 * 64) > 8192) return 0;          // one workgroup spans the tap width; 2 LDS rows <= 64 KB
    return f;
}

int pu_check(const void *s, const void *t, int dtype, int B, int C, int h, int w, int H, int W) {
    if (!s || !t) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (B <= 0 || C <= 0 || B > 65535 || (C + 1) / 2 > 65535) return SD_E_SHAPE;
    if (!pu_factor(h, w, H, W)) return SD_E_UNSUPPORTED;
    const size_t es = dtype == SD_F32 ? 4 : 2;
    if ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(t)) & (es - 1)) return SD_E_ALIGN;
    return SD_OK;
}

template <typename T, int F>
void pu_launch_fwd(const void *s, const void *t, float *lse2, double *sums, int B, int C, int h, int w, float c2, float inv_tau, hipStream_t st) {
    const int threads = (w + 63) / 64 * 64;                  // tap widths up to 256 (every BASELINE config) get the register budget of a 256-thread workgroup
    const dim3 grid(h + 1, B, F / PuGeo<F>::QR);
    if (threads <= 256) hipLaunchKernelGGL((pix_up_fwd<T, F, 256>), grid, dim3(threads), 0, st, (const T *)s, (const T *)t, lse2, sums, C, h, w, c2, inv_tau);
    else hipLaunchKernelGGL((pix_up_fwd<T, F, 1024>), grid, dim3(threads), 0, st, (const T *)s, (const T *)t, lse2, sums, C, h, w, c2, inv_tau);
}
template <typename T, int F>
void pu_launch_bwd(const void *s, const void *t, const float *lse2, const float *up, void *ds, int B, int C, int h, int w, float c2, float coef,
                   hipStream_t st) {
    const int R = h < 8 ? h : 8, nband = (h + R - 1) / R, threads = (w + 63) / 64 * 64;
    constexpr int CB = PuCB<F>::CB;
    const dim3 grid(nband, (C + CB - 1) / CB, B);
    const size_t lds = 2ull * F * threads * sizeof(float);
    if (threads <= 256) hipLaunchKernelGGL((pix_up_bwd<T, F, CB, 256>), grid, dim3(threads), lds, st, (const T *)s, (const T *)t, lse2, up, (T *)ds, C, h, w, R, c2, coef);
    else hipLaunchKernelGGL((pix_up_bwd<T, F, CB, 1024>), grid, dim3(threads), lds, st, (const T *)s, (const T *)t, lse2, up, (T *)ds, C, h, w, R, c2, coef);
}

}  // namespace
}  // namespace sd

extern "C" {

int sd_pix_kl_up_supported(int h, int w, int H, int W) { return sd::pu_factor(h, w, H, W) ? 1 : 0; }

size_t sd_pix_kl_up_workspace_bytes(int B, int h) { return B > 0 && h > 0 ? (size_t)B * (h + 1) * 4 * sizeof(double) + 16 : 0; }   /* one partial per (gap, image, row slice <= 4) */

int sd_pix_kl_up_fwd(const void *s, const void *t, int dtype, int B, int C, int h, int w, int H, int W, float inv_tau, float loss_scale,
                     float *pix_lse2, float *loss, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = sd::pu_check(s, t, dtype, B, C, h, w, H, W);
    if (rc) return rc;
    if (!pix_lse2 || !loss || !workspace) return SD_E_NULL;
    if (workspace_bytes < sd_pix_kl_up_workspace_bytes(B, h) || (reinterpret_cast<uintptr_t>(workspace) & 7)) return SD_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int F = sd::pu_factor(h, w, H, W);
    const float c2 = inv_tau * 1.44269504088896340736f;
    double *sums = static_cast<double *>(workspace);
#define SD_PU_FWD(TT)                                                                                        \
    do {                                                                                                     \
        if (F == 2) sd::pu_launch_fwd<TT, 2>(s, t, pix_lse2, sums, B, C, h, w, c2, inv_tau, st);             \
        else if (F == 4) sd::pu_launch_fwd<TT, 4>(s, t, pix_lse2, sums, B, C, h, w, c2, inv_tau, st);        \
        else sd::pu_launch_fwd<TT, 8>(s, t, pix_lse2, sums, B, C, h, w, c2, inv_tau, st);                    \
    } while (0)
    if (dtype == SD_F32) SD_PU_FWD(float);
    else SD_PU_FWD(sd::bf16_t);
#undef SD_PU_FWD
    hipLaunchKernelGGL(sd::pix_up_loss, dim3(1), dim3(256), 0, st, sums, loss, B * (h + 1) * (F / 2), loss_scale);
    return (int)hipGetLastError();
}

int sd_pix_kl_up_bwd(const void *s, const void *t, int dtype, int B, int C, int h, int w, int H, int W, float inv_tau, float coef,
                     const float *pix_lse2, const float *upstream, void *ds, void *stream) {
    int rc = sd::pu_check(s, t, dtype, B, C, h, w, H, W);
    if (rc) return rc;
    if (!pix_lse2 || !ds) return SD_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int F = sd::pu_factor(h, w, H, W);
    const float c2 = inv_tau * 1.44269504088896340736f;
#define SD_PU_BWD(TT)                                                                                        \
    do {                                                                                                     \
        if (F == 2) sd::pu_launch_bwd<TT, 2>(s, t, pix_lse2, upstream, ds, B, C, h, w, c2, coef, st);        \
        else if (F == 4) sd::pu_launch_bwd<TT, 4>(s, t, pix_lse2, upstream, ds, B, C, h, w, c2, coef, st);   \
        else sd::pu_launch_bwd<TT, 8>(s, t, pix_lse2, upstream, ds, B, C, h, w, c2, coef, st);               \
    } while (0)
    if (dtype == SD_F32) SD_PU_BWD(float);
    else SD_PU_BWD(sd::bf16_t);
#undef SD_PU_BWD
    return (int)hipGetLastError();
}

}  // extern "C"
