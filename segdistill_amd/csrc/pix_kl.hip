// pix_kl.hip -- pixel-wise distillation (PDLoss and the KL term of ATLoss / IFVDLoss), gfx950.
//
// Rows are (b, pixel): softmax over the C channels of one pixel (stride H*W in NCHW), loss
// normalised by B*H*W (reference losses.py:47-49,108-111 with loss_type='pixel').  Each lane owns
// N adjacent pixels (one 16-byte load per channel -> fully coalesced across the wave) and walks
// the channel axis once with an online softmax per pixel; no cross-lane traffic is needed for the
// row statistics, only for the final sum of row KLs.  HBM-bound: 8 B/element fwd, 12 B/element bwd.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cgd_device.h"

namespace sd {

namespace {

constexpr int kCU = 4;  // channels in flight per lane

// Online update with ONE exp per element: e = 2^{-|x - m| c2}.
__device__ __forceinline__ void online1(float x, float &m, float &z, float c2) {
    const float d = x - m;
    const float e = ex2(-fabsf(d) * c2);
    z = d > 0.f ? fmaf(z, e, 1.f) : z + e;
    m = fmaxf(m, x);
}
__device__ __forceinline__ void online2(float x, float diff, float &m, float &z, float &a, float c2) {
    const float d = x - m;
    const float e = ex2(-fabsf(d) * c2);
    if (d > 0.f) { z = fmaf(z, e, 1.f); a = fmaf(a, e, diff); }
    else { z += e; a = fmaf(e, diff, a); }
    m = fmaxf(m, x);
}

// grid = (ceil(HW / (kThreads*N)), B).  Writes lse2 planes [2][B*HW] and one partial KL sum per workgroup.
// AT = true adds the attention-transfer term of ATLoss (reference losses.py:191-192: MSE between the channel-MEAN maps of
// student and teacher) to the same pass: the per-pixel channel sums ride along the online softmax, the difference of the
// means is stored as a third plane for the backward, and at_ratio * dm^2 joins the pixel's KL in the workgroup sum.
template <typename T, bool VECTOR, bool AT>
__global__ __launch_bounds__(kThreads) void pix_fwd(const T *__restrict__ S, const T *__restrict__ Tt, float *__restrict__ lse2,
                                                     double *__restrict__ wg_sum, int C, int HW, long BHW, float c2, float inv_tau,
                                                     float at_ratio) {
    constexpr int N = VECTOR ? VecIO<T>::N : 1;
    const int b = blockIdx.y;
    const int p0 = (blockIdx.x * kThreads + threadIdx.x) * N;
    float kl = 0.f;
    if (p0 < HW) {
        const T *ps = S + (size_t)b * C * HW + p0, *pt = Tt + (size_t)b * C * HW + p0;
        float ms[N], zs[N], mt[N], zt[N], a[N], sd_[N];
#pragma unroll
        for (int i = 0; i < N; ++i) { ms[i] = mt[i] = kNegBig; zs[i] = zt[i] = a[i] = sd_[i] = 0.f; }
        int c = 0;
        for (; c + kCU <= C; c += kCU) {
            float s[kCU][N], t[kCU][N];
#pragma unroll
            for (int u = 0; u < kCU; ++u) {
                if constexpr (VECTOR) {
                    VecIO<T>::load(ps + (size_t)(c + u) * HW, s[u]);
                    VecIO<T>::load(pt + (size_t)(c + u) * HW, t[u]);
                } else {
                    s[u][0] = VecIO<T>::load1(ps + (size_t)(c + u) * HW);
                    t[u][0] = VecIO<T>::load1(pt + (size_t)(c + u) * HW);
                }
            }
#pragma unroll
            for (int u = 0; u < kCU; ++u)
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    online1(s[u][i], ms[i], zs[i], c2);
                    online2(t[u][i], t[u][i] - s[u][i], mt[i], zt[i], a[i], c2);
                    if constexpr (AT) sd_[i] += s[u][i] - t[u][i];
                }
        }
        for (; c < C; ++c) {
            float s[N], t[N];
            if constexpr (VECTOR) {
                VecIO<T>::load(ps + (size_t)c * HW, s);
                VecIO<T>::load(pt + (size_t)c * HW, t);
            } else {
                s[0] = VecIO<T>::load1(ps + (size_t)c * HW);
                t[0] = VecIO<T>::load1(pt + (size_t)c * HW);
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                online1(s[i], ms[i], zs[i], c2);
                online2(t[i], t[i] - s[i], mt[i], zt[i], a[i], c2);
                if constexpr (AT) sd_[i] += s[i] - t[i];
            }
        }
        const float ln2 = 0.69314718055994531f;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float l2s = fmaf(ms[i], c2, __builtin_amdgcn_logf(zs[i])), l2t = fmaf(mt[i], c2, __builtin_amdgcn_logf(zt[i]));
            lse2[(size_t)b * HW + p0 + i] = l2s;
            lse2[BHW + (size_t)b * HW + p0 + i] = l2t;
            kl += a[i] * inv_tau / zt[i] + (l2s - l2t) * ln2;
            if constexpr (AT) {
                const float dm = sd_[i] / (float)C;
                lse2[2 * BHW + (size_t)b * HW + p0 + i] = dm;
                kl = fmaf(at_ratio * dm, dm, kl);
            }
        }
    }
    __shared__ double acc[kThreads / 64];
    double v = (double)kl;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) acc[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0;
        for (int i = 0; i < kThreads / 64; ++i) tot += acc[i];
        wg_sum[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = tot;
    }
}

__global__ __launch_bounds__(kThreads) void pix_loss(const double *__restrict__ wg_sum, float *__restrict__ loss, int n, float loss_scale) {
    __shared__ double acc[kThreads / 64];
    double v = 0;
    for (int i = threadIdx.x; i < n; i += kThreads) v += wg_sum[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) acc[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0;
        for (int i = 0; i < kThreads / 64; ++i) tot += acc[i];
        loss[0] = (float)(tot * (double)loss_scale);
    }
}

template <typename T, bool VECTOR, bool AT>
__global__ __launch_bounds__(kThreads) void pix_bwd(const T *__restrict__ S, const T *__restrict__ Tt, const float *__restrict__ lse2,
                                                     const float *__restrict__ upstream, T *__restrict__ dS, int C, int HW, long BHW,
                                                     int cpb, float c2, float coef, float at_coef) {
    constexpr int N = VECTOR ? VecIO<T>::N : 1;
    const int b = blockIdx.y;
    const int p0 = (blockIdx.x * kThreads + threadIdx.x) * N;
    if (p0 >= HW) return;
    const float kk = upstream ? coef * upstream[0] : coef;
    float ls[N], lt[N], ad[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        ls[i] = lse2[(size_t)b * HW + p0 + i];
        lt[i] = lse2[BHW + (size_t)b * HW + p0 + i];
        ad[i] = 0.f;
        if constexpr (AT) ad[i] = (upstream ? at_coef * upstream[0] : at_coef) * lse2[2 * BHW + (size_t)b * HW + p0 + i];   // same for every channel
    }
    const int c_lo = blockIdx.z * cpb, c_hi = min(C, c_lo + cpb);
    const size_t off = (size_t)b * C * HW + p0;
    for (int c = c_lo; c < c_hi; ++c) {
        float s[N], t[N], d[N];
        if constexpr (VECTOR) {
            VecIO<T>::load(S + off + (size_t)c * HW, s);
            VecIO<T>::load(Tt + off + (size_t)c * HW, t);
        } else {
            s[0] = VecIO<T>::load1(S + off + (size_t)c * HW);
            t[0] = VecIO<T>::load1(Tt + off + (size_t)c * HW);
        }
#pragma unroll
        for (int i = 0; i < N; ++i) d[i] = fmaf(kk, ex2(fmaf(s[i], c2, -ls[i])) - ex2(fmaf(t[i], c2, -lt[i])), ad[i]);
        if constexpr (VECTOR) VecIO<T>::template store<true>(dS + off + (size_t)c * HW, d);
        else VecIO<T>::store1(dS + off + (size_t)c * HW, d[0]);
    }
}

template <typename T> bool vec_ok(const void *a, const void *b, const void *c, long HW) {
    auto al = [](const void *p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return HW % VecIO<T>::N == 0 && al(a) && al(b) && al(c);
}

int check_pix(const void *S, const void *Tt, int dtype, int B, int C, int H, int W) {
    if (!S || !Tt) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || B > 65535) return SD_E_SHAPE;
    if ((long)H * W > 0x3fffffffL) return SD_E_SHAPE;
    const size_t es = dtype == SD_F32 ? 4 : 2;
    if ((reinterpret_cast<uintptr_t>(S) | reinterpret_cast<uintptr_t>(Tt)) & (es - 1)) return SD_E_ALIGN;
    return SD_OK;
}

template <typename T, bool AT>
int pix_fwd_impl(const void *S, const void *Tt, int B, int C, int H, int W, float inv_tau, float loss_scale, float at_ratio, float *lse2,
                 float *loss, void *ws, size_t ws_bytes, hipStream_t st) {
    const long HW = (long)H * W;
    const bool vec = vec_ok<T>(S, Tt, nullptr, HW);
    const int N = vec ? VecIO<T>::N : 1;
    const int gx = (int)((HW + (long)kThreads * N - 1) / ((long)kThreads * N));
    const size_t need = (size_t)gx * B * sizeof(double);
    if (ws_bytes < need || (reinterpret_cast<uintptr_t>(ws) & 15)) return SD_E_WORKSPACE;
    const float c2 = inv_tau * 1.44269504088896340736f;
    double *sums = static_cast<double *>(ws);
    if (vec) hipLaunchKernelGGL((pix_fwd<T, true, AT>), dim3(gx, B), dim3(kThreads), 0, st, (const T *)S, (const T *)Tt, lse2, sums, C, (int)HW,
                                (long)B * HW, c2, inv_tau, at_ratio);
    else hipLaunchKernelGGL((pix_fwd<T, false, AT>), dim3(gx, B), dim3(kThreads), 0, st, (const T *)S, (const T *)Tt, lse2, sums, C, (int)HW,
                            (long)B * HW, c2, inv_tau, at_ratio);
    hipLaunchKernelGGL(pix_loss, dim3(1), dim3(kThreads), 0, st, sums, loss, gx * B, loss_scale);
    return (int)hipGetLastError();
}

template <typename T, bool AT>
int pix_bwd_impl(const void *S, const void *Tt, int B, int C, int H, int W, float inv_tau, float coef, float at_coef, const float *lse2,
                 const float *upstream, void *dS, hipStream_t st) {
    const long HW = (long)H * W;
    const bool vec = vec_ok<T>(S, Tt, dS, HW);
    const int N = vec ? VecIO<T>::N : 1;
    const int gx = (int)((HW + (long)kThreads * N - 1) / ((long)kThreads * N));
    // split the channel axis over grid.z so that small images still fill the chip
    int gz = 1;
    while ((long)gx * B * gz < 4096 && gz * 2 <= C && gz < 64) gz *= 2;
    const int cpb = (C + gz - 1) / gz;
    gz = (C + cpb - 1) / cpb;
    const float c2 = inv_tau * 1.44269504088896340736f;
    if (vec) hipLaunchKernelGGL((pix_bwd<T, true, AT>), dim3(gx, B, gz), dim3(kThreads), 0, st, (const T *)S, (const T *)Tt, lse2, upstream,
                                (T *)dS, C, (int)HW, (long)B * HW, cpb, c2, coef, at_coef);
    else hipLaunchKernelGGL((pix_bwd<T, false, AT>), dim3(gx, B, gz), dim3(kThreads), 0, st, (const T *)S, (const T *)Tt, lse2, upstream,
                            (T *)dS, C, (int)HW, (long)B * HW, cpb, c2, coef, at_coef);
    return (int)hipGetLastError();
}

}  // namespace
}  // namespace sd

extern "C" {

size_t sd_pix_kl_workspace_bytes(int B, int C, int H, int W) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    const long HW = (long)H * W;
    const long gx = (HW + sd::kThreads - 1) / sd::kThreads;  // scalar build upper bound
    return (size_t)gx * B * sizeof(double) + 16;
}

int sd_pix_kl_fwd(const void *S, const void *T, int dtype, int B, int C, int H, int W, float inv_tau, float loss_scale, float *pix_lse2,
                  float *loss, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = sd::check_pix(S, T, dtype, B, C, H, W);
    if (rc) return rc;
    if (!pix_lse2 || !loss || !workspace) return SD_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32)
        return sd::pix_fwd_impl<float, false>(S, T, B, C, H, W, inv_tau, loss_scale, 0.f, pix_lse2, loss, workspace, workspace_bytes, st);
    return sd::pix_fwd_impl<sd::bf16_t, false>(S, T, B, C, H, W, inv_tau, loss_scale, 0.f, pix_lse2, loss, workspace, workspace_bytes, st);
}

int sd_pix_kl_bwd(const void *S, const void *T, int dtype, int B, int C, int H, int W, float inv_tau, float coef, const float *pix_lse2,
                  const float *upstream, void *dS, void *stream) {
    int rc = sd::check_pix(S, T, dtype, B, C, H, W);
    if (rc) return rc;
    if (!pix_lse2 || !dS) return SD_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32) return sd::pix_bwd_impl<float, false>(S, T, B, C, H, W, inv_tau, coef, 0.f, pix_lse2, upstream, dS, st);
    return sd::pix_bwd_impl<sd::bf16_t, false>(S, T, B, C, H, W, inv_tau, coef, 0.f, pix_lse2, upstream, dS, st);
}

/* ATLoss (reference losses.py:175-197) in one pass each way:
 *   loss = mean_{b,p} (mean_c S - mean_c T)^2  +  1/(B*H*W) * sum_pixels KL(softmax_C(T) || softmax_C(S))
 * planes: [3][B*H*W] fp32 (base-2 lse of S, of T, and the channel-mean difference). */
int sd_at_kl_fwd(const void *S, const void *T, int dtype, int B, int C, int H, int W, float *planes, float *loss, void *workspace,
                 size_t workspace_bytes, void *stream) {
    int rc = sd::check_pix(S, T, dtype, B, C, H, W);
    if (rc) return rc;
    if (!planes || !loss || !workspace) return SD_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float inv_rows = 1.f / ((float)B * (float)H * (float)W);
    if (dtype == SD_F32)
        return sd::pix_fwd_impl<float, true>(S, T, B, C, H, W, 1.f, inv_rows, 1.f, planes, loss, workspace, workspace_bytes, st);
    return sd::pix_fwd_impl<sd::bf16_t, true>(S, T, B, C, H, W, 1.f, inv_rows, 1.f, planes, loss, workspace, workspace_bytes, st);
}

int sd_at_kl_bwd(const void *S, const void *T, int dtype, int B, int C, int H, int W, const float *planes, const float *upstream, void *dS,
                 void *stream) {
    int rc = sd::check_pix(S, T, dtype, B, C, H, W);
    if (rc) return rc;
    if (!planes || !dS) return SD_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float inv_rows = 1.f / ((float)B * (float)H * (float)W);
    const float at_coef = 2.f * inv_rows / (float)C;          // d/dS[b,c,p] of mean_{b,p} dm^2, dm = (sum_c S - sum_c T) / C
    if (dtype == SD_F32) return sd::pix_bwd_impl<float, true>(S, T, B, C, H, W, 1.f, inv_rows, at_coef, planes, upstream, dS, st);
    return sd::pix_bwd_impl<sd::bf16_t, true>(S, T, B, C, H, W, 1.f, inv_rows, at_coef, planes, upstream, dS, st);
}

}  // extern "C"
