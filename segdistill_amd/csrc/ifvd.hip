// ifvd.hip -- the intra-class feature-variation term of IFVDLoss (reference losses.py:199-238), gfx950.
//
//   centre_k = mean of the features of the pixels labelled k (per image)        (:226-230, a 150-pass full-tensor mask loop)
//   sim_p    = cos(feature_p, centre_{label(p)})                                 (:231-233)
//   loss     = 10 * mean_p (sim^S_p - sim^T_p)^2                                 (:235)
// and its gradient with respect to the student feature, which the reference's autograd also takes THROUGH the class centres.
//
// The label map is shared by every channel and by both networks, so the host sorts the pixels of an image by class ONCE
// (order[b][HW], offsets[b][K+1]); after that a class mean is a contiguous run of gathers -- deterministic (no float
// atomics), one wave per (image, channel, class) run:
//   ifvd_seg_sum   out[b,k,c] = sum_{p in class k} w_p * X[b,c,p]   (/ (n_k + 1e-6) in mean mode; w optional)
//   ifvd_cos       per pixel: dot / norms against its class centre (one pass over C, coalesced over pixels); in student mode
//                  also the squared difference to the teacher's similarity and the three per-pixel gradient coefficients
//   ifvd_bwd       dS[b,c,p] = g * ( alpha_p * mu_k[c] - gamma_p * S[b,c,p] + (A_k[c] - mu_k[c] * B_k) / (n_k + 1e-6) )
// with alpha = w/(|a||mu|), gamma = w*sim/|a|^2, beta = w*sim/|mu|^2, w = 20 (sim^S - sim^T)/(B*HW),
// A_k[c] = seg_sum(alpha * S), B_k = seg_sum(beta).  Pixels without a class (label outside [0,K)) compare a feature with itself:
// similarity 1, gradient 0.  HBM-bound byte work at tap resolution (78 MB per tensor at config-2 sizes).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cgd_device.h"

namespace sd {
namespace {

constexpr float kCosEps = 1e-8f;   // F.cosine_similarity's eps: each norm is clamped from below

// grid (C, B), 256 threads = 4 waves; wave w sums the class runs k = w, w+4, ...
template <typename T>
__global__ __launch_bounds__(256) void ifvd_seg_sum(const T *__restrict__ X, const float *__restrict__ wgt, const int *__restrict__ order,
                                                     const int *__restrict__ offsets, float *__restrict__ out, int C, int HW, int K,
                                                     int mean_mode) {
    const int c = blockIdx.x, b = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const T *plane = X + ((size_t)b * C + c) * HW;
    const int *ord = order + (size_t)b * HW, *off = offsets + (size_t)b * (K + 1);
    const float *wp = wgt ? wgt + (size_t)b * HW : nullptr;
    for (int k = wave; k < K; k += 4) {
        const int lo = off[k], hi = off[k + 1];
        float acc = 0.f;
        for (int i = lo + lane; i < hi; i += 64) {
            const int p = ord[i];
            const float v = VecIO<T>::load1(plane + p);
            acc = wp ? fmaf(wp[p], v, acc) : acc + v;
        }
        acc = wave_sum(acc);
        if (lane == 0) out[((size_t)b * K + k) * C + c] = mean_mode ? acc / ((float)(hi - lo) + 1e-6f) : acc;
    }
}

// grid (ceil(HW/256), B).  sim_ref == nullptr: teacher mode (only `sim` is written).
template <typename T>
__global__ __launch_bounds__(256) void ifvd_cos(const T *__restrict__ X, const int *__restrict__ cls, const float *__restrict__ mean,
                                                 const float *__restrict__ sim_ref, float *__restrict__ sim, float *__restrict__ coefs,
                                                 double *__restrict__ wg_sum, int C, int HW, int K, long BHW, float w_scale) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    float d2 = 0.f;
    if (p < HW) {
        const int k = cls[(size_t)b * HW + p];
        const bool valid = k >= 0 && k < K;
        const T *px = X + (size_t)b * C * HW + p;
        const float *mu = mean + ((size_t)b * K + (valid ? k : 0)) * C;
        float dot = 0.f, na2 = 0.f, nb2 = 0.f;
        for (int c = 0; c < C; ++c) {
            const float a = VecIO<T>::load1(px + (size_t)c * HW);
            const float m = valid ? mu[c] : a;
            dot = fmaf(a, m, dot);
            na2 = fmaf(a, a, na2);
            nb2 = fmaf(m, m, nb2);
        }
        const float na = fmaxf(sqrtf(na2), kCosEps), nb = fmaxf(sqrtf(nb2), kCosEps);
        const float s = dot / (na * nb);
        sim[(size_t)b * HW + p] = s;
        if (sim_ref) {
            const float d = s - sim_ref[(size_t)b * HW + p];
            d2 = d * d;
            const float w = valid ? w_scale * d : 0.f;
            coefs[(size_t)b * HW + p] = w / (na * nb);                 // alpha
            coefs[BHW + (size_t)b * HW + p] = w * s / (nb * nb);       // beta
            coefs[2 * BHW + (size_t)b * HW + p] = w * s / (na * na);   // gamma
        }
    }
    if (!sim_ref) return;
    __shared__ double acc[4];
    double v = (double)d2;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) acc[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) wg_sum[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

__global__ __launch_bounds__(256) void ifvd_loss(const double *__restrict__ wg_sum, float *__restrict__ loss, int n, float scale) {
    __shared__ double acc[4];
    double v = 0;
    for (int i = threadIdx.x; i < n; i += 256) v += wg_sum[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) acc[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (float)((acc[0] + acc[1] + acc[2] + acc[3]) * (double)scale);
}

// grid (ceil(HW/256), B)
template <typename T>
__global__ __launch_bounds__(256) void ifvd_bwd(const T *__restrict__ X, const int *__restrict__ cls, const float *__restrict__ mean,
                                                 const float *__restrict__ coefs, const float *__restrict__ A, const float *__restrict__ Bk,
                                                 const int *__restrict__ offsets, const float *__restrict__ upstream, T *__restrict__ dS, int C,
                                                 int HW, int K, long BHW) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int k = cls[(size_t)b * HW + p];
    const bool valid = k >= 0 && k < K;
    const float g = upstream ? upstream[0] : 1.f;
    const T *px = X + (size_t)b * C * HW + p;
    T *pd = dS + (size_t)b * C * HW + p;
    if (!valid) {
        for (int c = 0; c < C; ++c) VecIO<T>::store1(pd + (size_t)c * HW, 0.f);
        return;
    }
    const float alpha = g * coefs[(size_t)b * HW + p], gamma = g * coefs[2 * BHW + (size_t)b * HW + p];
    const int *off = offsets + (size_t)b * (K + 1);
    const float invn = g / ((float)(off[k + 1] - off[k]) + 1e-6f);
    const float bk = Bk[(size_t)b * K + k];
    const float *mu = mean + ((size_t)b * K + k) * C, *ak = A + ((size_t)b * K + k) * C;
    for (int c = 0; c < C; ++c) {
        const float a = VecIO<T>::load1(px + (size_t)c * HW);
        const float m = mu[c];
        VecIO<T>::store1(pd + (size_t)c * HW, fmaf(alpha, m, fmaf(-gamma, a, (ak[c] - m * bk) * invn)));
    }
}

int check_ifvd(const void *X, int dtype, int B, int C, int HW, int K) {
    if (!X) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (B <= 0 || C <= 0 || HW <= 0 || K <= 0 || B > 65535 || C > 65535) return SD_E_SHAPE;
    return SD_OK;
}

}  // namespace
}  // namespace sd

extern "C" {

size_t sd_ifvd_workspace_bytes(int B, int HW) {
    if (B <= 0 || HW <= 0) return 0;
    return (size_t)((HW + 255) / 256) * B * sizeof(double) + 16;
}

int sd_ifvd_seg_sum(const void *X, int dtype, const float *wgt, const int *order, const int *offsets, float *out, int B, int C, int HW, int K,
                    int mean_mode, void *stream) {
    int rc = sd::check_ifvd(X, dtype, B, C, HW, K);
    if (rc) return rc;
    if (!order || !offsets || !out) return SD_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32)
        hipLaunchKernelGGL((sd::ifvd_seg_sum<float>), dim3(C, B), dim3(256), 0, st, (const float *)X, wgt, order, offsets, out, C, HW, K,
                           mean_mode);
    else
        hipLaunchKernelGGL((sd::ifvd_seg_sum<sd::bf16_t>), dim3(C, B), dim3(256), 0, st, (const sd::bf16_t *)X, wgt, order, offsets, out, C, HW, K,
                           mean_mode);
    return (int)hipGetLastError();
}

int sd_ifvd_cos(const void *X, int dtype, const int *cls, const float *mean, const float *sim_ref, float *sim, float *coefs, float *loss,
                void *workspace, size_t workspace_bytes, int B, int C, int HW, int K, void *stream) {
    int rc = sd::check_ifvd(X, dtype, B, C, HW, K);
    if (rc) return rc;
    if (!cls || !mean || !sim) return SD_E_NULL;
    if (sim_ref && (!coefs || !loss || !workspace)) return SD_E_NULL;
    if (sim_ref && (workspace_bytes < sd_ifvd_workspace_bytes(B, HW) - 16 || (reinterpret_cast<uintptr_t>(workspace) & 7))) return SD_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int gx = (HW + 255) / 256;
    const long BHW = (long)B * HW;
    const float w_scale = 20.f / (float)BHW;            // d/dsim of 10 * mean (sim_s - sim_t)^2
    double *sums = static_cast<double *>(workspace);
    if (dtype == SD_F32)
        hipLaunchKernelGGL((sd::ifvd_cos<float>), dim3(gx, B), dim3(256), 0, st, (const float *)X, cls, mean, sim_ref, sim, coefs, sums, C, HW, K,
                           BHW, w_scale);
    else
        hipLaunchKernelGGL((sd::ifvd_cos<sd::bf16_t>), dim3(gx, B), dim3(256), 0, st, (const sd::bf16_t *)X, cls, mean, sim_ref, sim, coefs, sums, C,
                           HW, K, BHW, w_scale);
    if (sim_ref) hipLaunchKernelGGL(sd::ifvd_loss, dim3(1), dim3(256), 0, st, sums, loss, gx * B, 10.f / (float)BHW);
    return (int)hipGetLastError();
}

int sd_ifvd_bwd(const void *X, int dtype, const int *cls, const float *mean, const float *coefs, const float *A, const float *Bk,
                const int *offsets, const float *upstream, void *dS, int B, int C, int HW, int K, void *stream) {
    int rc = sd::check_ifvd(X, dtype, B, C, HW, K);
    if (rc) return rc;
    if (!cls || !mean || !coefs || !A || !Bk || !offsets || !dS) return SD_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int gx = (HW + 255) / 256;
    if (dtype == SD_F32)
        hipLaunchKernelGGL((sd::ifvd_bwd<float>), dim3(gx, B), dim3(256), 0, st, (const float *)X, cls, mean, coefs, A, Bk, offsets, upstream,
                           (float *)dS, C, HW, K, (long)B * HW);
    else
        hipLaunchKernelGGL((sd::ifvd_bwd<sd::bf16_t>), dim3(gx, B), dim3(256), 0, st, (const sd::bf16_t *)X, cls, mean, coefs, A, Bk, offsets,
                           upstream, (sd::bf16_t *)dS, C, HW, K, (long)B * HW);
    return (int)hipGetLastError();
}

}  // extern "C"
