// cgd_kl.hip -- Channel-Group-Distillation KL criterion for gfx950 (MI355X), regime R1
// (operands already at softmax resolution).
//
// What the reference computes with ~14 ATen passes (losses.py:105-112: pad/view, div tau,
// log_softmax, softmax, KLDivLoss(sum) and their autograd) is done here in one streaming
// read for the forward and one read+write pass for the backward.
//
// A softmax row is g planes of H*W pixels (2.1 M elements at g=8, 512x512): far beyond LDS,
// and there are only B*ceil(C/g) rows (152 at the headline config), so parallelism comes
// from INSIDE rows.  Each 256-thread workgroup owns one chunk of one channel plane and
// keeps an online (max, partition-sum) pair for S and for T plus the cross term
//     a = sum_i 2^{(t_i - m_t) c2} (t_i - s_i),   c2 = log2(e)/tau,
// per lane, rescaling once per 16 elements; lanes are combined with wave64 shuffles, waves
// through LDS, and the per-chunk partials by a tiny second kernel in fp64:
//     KL_r = (a/tau)/Z_t - lse_t + lse_s.
// HBM-bound by design: 8 B/element forward, 12 B/element backward (fp32).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "cgd_device.h"

namespace sd {

namespace {

// ---- forward, streaming pass ----------------------------------------------------------------
// grid.x = B*C*nchunk; workgroup k of plane slot (b,c') covers elements
// [k*chunk, min((k+1)*chunk, HW)) of channel perm[c'] and writes partial[(b*C+c')*nchunk+k].
template <typename T, bool VECTOR>
__global__ __launch_bounds__(kThreads) void cgd_fwd_partials(const T *__restrict__ S, const T *__restrict__ Tt,
                                                              const int32_t *__restrict__ perm, RowPart *__restrict__ part,
                                                              int C, int HW, int nchunk, int iters, float c2) {
    constexpr int N = VECTOR ? VecIO<T>::N : 1;
    const int wg = blockIdx.x;
    const int k = wg % nchunk;
    const int slot = wg / nchunk;  // b*C + c'
    const int b = slot / C, cs = slot - b * C;
    const int ch = perm ? perm[cs] : cs;
    const size_t base = ((size_t)b * C + ch) * (size_t)HW;
    const T *ps = S + base, *pt = Tt + base;
    const int chunk = kThreads * N * kUnroll * iters;
    const int lo = k * chunk;
    const int hi = min(lo + chunk, HW);

    RowPart st = {kNegBig, 0.f, kNegBig, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        const int e0 = lo + (it * kUnroll * kThreads + threadIdx.x) * N;
        if (e0 + (kUnroll - 1) * kThreads * N + N <= hi) {  // all kUnroll vectors in range
            float s[kUnroll * N], t[kUnroll * N];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int e = e0 + u * kThreads * N;
                if constexpr (VECTOR) {
                    VecIO<T>::load(ps + e, *reinterpret_cast<float(*)[N]>(&s[u * N]));
                    VecIO<T>::load(pt + e, *reinterpret_cast<float(*)[N]>(&t[u * N]));
                } else {
                    s[u] = VecIO<T>::load1(ps + e);
                    t[u] = VecIO<T>::load1(pt + e);
                }
            }
            fold<kUnroll * N>(st, s, t, c2);
        } else {
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int e = e0 + u * kThreads * N;
                if (e < hi) {  // HW % N == 0 in the vector build, so a vector is all-in or all-out
                    float s[N], t[N];
                    if constexpr (VECTOR) {
                        VecIO<T>::load(ps + e, s);
                        VecIO<T>::load(pt + e, t);
                    } else {
                        s[0] = VecIO<T>::load1(ps + e);
                        t[0] = VecIO<T>::load1(pt + e);
                    }
                    fold<N>(st, s, t, c2);
                }
            }
        }
    }
    const RowPart w = block_combine(st, c2);
    if (threadIdx.x == 0) part[wg] = w;
}

// ---- forward, per-row finalisation (fp64) ---------------------------------------------------
// One wave per row; row (b,j) owns partials [(b*C + j*g)*nchunk, (b*C + min(C,(j+1)*g))*nchunk).
__global__ __launch_bounds__(kThreads) void cgd_fwd_rows(const RowPart *__restrict__ part, float *__restrict__ row_lse2,
                                                          float *__restrict__ row_kl, int rows, int C, int g, int G,
                                                          int nchunk, float c2, float inv_tau, long batch_stride, long slot_stride,
                                                          long chunk_stride, const int32_t *__restrict__ chan_of_slot) {
    const int row = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (row >= rows) return;  // whole wave exits together
    const int lane = threadIdx.x & 63;
    const int b = row / G, j = row - b * G;
    const int c_lo = j * g, c_hi = min(C, c_lo + g);
    // partial (slot s, chunk k) sits at b*batch_stride + chan(s)*slot_stride + k*chunk_stride; the slot-major layout of the NCHW kernels
    // makes that p[i]
    const RowPart *pb = part + (size_t)b * batch_stride;
    const bool dense = chan_of_slot == nullptr && slot_stride == nchunk && chunk_stride == 1;
    auto at = [&](int i) -> const RowPart & {
        if (dense) return pb[(size_t)c_lo * nchunk + i];
        const int s = c_lo + i / nchunk, k = i - (i / nchunk) * nchunk;
        return pb[(size_t)(chan_of_slot ? chan_of_slot[s] : s) * slot_stride + (size_t)k * chunk_stride];
    };
    const int n = (c_hi - c_lo) * nchunk;
    float ms = kNegBig, mt = kNegBig;
    for (int i = lane; i < n; i += 64) { const RowPart &q = at(i); ms = fmaxf(ms, q.ms); mt = fmaxf(mt, q.mt); }
    ms = wave_max(ms); mt = wave_max(mt);
    double zs = 0, zt = 0, a = 0;
    for (int i = lane; i < n; i += 64) {
        const RowPart q = at(i);
        const double fs = exp2((double)(q.ms - ms) * (double)c2), ft = exp2((double)(q.mt - mt) * (double)c2);
        zs += (double)q.zs * fs;
        zt += (double)q.zt * ft;
        a += (double)q.a * ft;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        zs += __shfl_xor(zs, o, 64);
        zt += __shfl_xor(zt, o, 64);
        a += __shfl_xor(a, o, 64);
    }
    if (lane == 0) {
        const double l2s = (double)ms * c2 + log2(zs), l2t = (double)mt * c2 + log2(zt);
        const double ln2 = 0.69314718055994530942;
        row_lse2[2 * row] = (float)l2s;
        row_lse2[2 * row + 1] = (float)l2t;
        row_kl[row] = (float)(a * (double)inv_tau / zt + (l2s - l2t) * ln2);
    }
}

__global__ __launch_bounds__(kThreads) void cgd_fwd_loss(const float *__restrict__ row_kl, float *__restrict__ loss, int rows,
                                                          float loss_scale) {
    __shared__ double acc[kThreads / 64];
    double v = 0;
    for (int i = threadIdx.x; i < rows; i += kThreads) v += (double)row_kl[i];
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

// ---- backward -------------------------------------------------------------------------------
// dS_i = k (2^{s_i c2 - lse2_s} - 2^{t_i c2 - lse2_t}),  k = coef * upstream.
template <typename T, bool VECTOR, bool NT, int U>
__global__ __launch_bounds__(kThreads) void cgd_bwd(const T *__restrict__ S, const T *__restrict__ Tt, const int32_t *__restrict__ perm,
                                                     const float *__restrict__ row_lse2, const float *__restrict__ upstream,
                                                     T *__restrict__ dS, int C, int HW, int g, int G, int nchunk, int iters,
                                                     float c2, float coef) {
    constexpr int N = VECTOR ? VecIO<T>::N : 1;
    const int wg = blockIdx.x;
    const int k = wg % nchunk;
    const int slot = wg / nchunk;
    const int b = slot / C, cs = slot - b * C;
    const int ch = perm ? perm[cs] : cs;
    const int row = b * G + cs / g;
    const float ls = row_lse2[2 * row], lt = row_lse2[2 * row + 1];
    const float kk = upstream ? coef * upstream[0] : coef;
    const size_t base = ((size_t)b * C + ch) * (size_t)HW;
    const T *ps = S + base, *pt = Tt + base;
    T *pd = dS + base;
    const int chunk = kThreads * N * U * iters;
    const int lo = k * chunk;
    const int hi = min(lo + chunk, HW);
    for (int it = 0; it < iters; ++it) {
        const int e0 = lo + (it * U * kThreads + threadIdx.x) * N;
        float s[U][N], t[U][N];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * kThreads * N;
            if (e < hi) {
                if constexpr (VECTOR) {
                    VecIO<T>::load(ps + e, s[u]);
                    VecIO<T>::load(pt + e, t[u]);
                } else {
                    s[u][0] = VecIO<T>::load1(ps + e);
                    t[u][0] = VecIO<T>::load1(pt + e);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * kThreads * N;
            if (e < hi) {
                float d[N];
#pragma unroll
                for (int i = 0; i < N; ++i) d[i] = kk * (ex2(fmaf(s[u][i], c2, -ls)) - ex2(fmaf(t[u][i], c2, -lt)));
                if constexpr (VECTOR) VecIO<T>::template store<NT>(pd + e, d);
                else VecIO<T>::store1(pd + e, d[0]);
            }
        }
    }
}

int g_fwd_iters = 4;   // tunable "cgd_fwd_chunk_iters": 4096 float4 per operand per workgroup
int g_bwd_iters = 1;   // tunable "cgd_bwd_chunk_iters"
int g_bwd_unroll = 4; // tunable "cgd_bwd_unroll": 16-byte loads per operand in flight per lane (2 | 4 | 8)
int g_bwd_nt = 1;      // tunable "cgd_bwd_nt_store": dS is consumed by a later kernel, never re-read here

struct Geo {
    int N, iters, chunk, nchunk, G, rows;
    bool vec;
};

template <typename T>
Geo geometry(const void *S, const void *Tt, const void *dS, int C, int H, int W, int g, int B, int want_iters, int unroll = kUnroll) {
    Geo q;
    const long HW = (long)H * W;
    const int VN = VecIO<T>::N;
    auto al = [](const void *p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    q.vec = (HW % VN == 0) && al(S) && al(Tt) && al(dS);
    q.N = q.vec ? VN : 1;
    int iters = want_iters;
    const long per_iter = (long)kThreads * q.N * unroll;
    const long need = (HW + per_iter - 1) / per_iter;
    if (iters > need) iters = (int)need;
    if (iters < 1) iters = 1;
    q.iters = iters;
    q.chunk = (int)(per_iter * iters);
    q.nchunk = (int)((HW + q.chunk - 1) / q.chunk);
    q.G = (C + g - 1) / g;
    q.rows = B * q.G;
    return q;
}

int check_common(const void *S, const void *Tt, int dtype, int B, int C, int H, int W, int g) {
    if (!S || !Tt) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || g <= 0) return SD_E_SHAPE;
    if ((long)H * W > 0x3fffffffL) return SD_E_SHAPE;               // in-plane offsets are int32
    const size_t es = dtype == SD_F32 ? 4 : 2;
    if ((reinterpret_cast<uintptr_t>(S) | reinterpret_cast<uintptr_t>(Tt)) & (es - 1)) return SD_E_ALIGN;
    return SD_OK;
}

template <typename T>
int fwd_impl(const void *S, const void *Tt, int B, int C, int H, int W, int g, float inv_tau, float loss_scale,
             const int32_t *perm, float *row_lse2, float *row_kl, float *loss, void *ws, size_t ws_bytes, hipStream_t st) {
    const Geo q = geometry<T>(S, Tt, nullptr, C, H, W, g, B, g_fwd_iters);
    const long nwg = (long)B * C * q.nchunk;
    if (nwg > 0x7fffffffL) return SD_E_SHAPE;
    if (ws_bytes < (size_t)nwg * sizeof(RowPart) || (reinterpret_cast<uintptr_t>(ws) & 15)) return SD_E_WORKSPACE;
    const float c2 = inv_tau * 1.44269504088896340736f;
    RowPart *part = static_cast<RowPart *>(ws);
    const int HW = H * W;
    if (q.vec)
        hipLaunchKernelGGL((cgd_fwd_partials<T, true>), dim3((unsigned)nwg), dim3(kThreads), 0, st, (const T *)S, (const T *)Tt, perm, part,
                           C, HW, q.nchunk, q.iters, c2);
    else
        hipLaunchKernelGGL((cgd_fwd_partials<T, false>), dim3((unsigned)nwg), dim3(kThreads), 0, st, (const T *)S, (const T *)Tt, perm,
                           part, C, HW, q.nchunk, q.iters, c2);
    launch_row_finalize(part, row_lse2, row_kl, loss, B, C, g, q.nchunk, c2, inv_tau, loss_scale, st);
    return (int)hipGetLastError();
}

template <typename T>
int bwd_impl(const void *S, const void *Tt, int B, int C, int H, int W, int g, float inv_tau, float coef, const int32_t *perm,
             const float *row_lse2, const float *upstream, void *dS, hipStream_t st) {
    const int U = g_bwd_unroll;
    const Geo q = geometry<T>(S, Tt, dS, C, H, W, g, B, g_bwd_iters, U);
    const long nwg = (long)B * C * q.nchunk;
    if (nwg > 0x7fffffffL) return SD_E_SHAPE;
    const float c2 = inv_tau * 1.44269504088896340736f;
    const int HW = H * W;
#define SD_BWD(VEC, NTS, UU)                                                                                                          \
    hipLaunchKernelGGL((cgd_bwd<T, VEC, NTS, UU>), dim3((unsigned)nwg), dim3(kThreads), 0, st, (const T *)S, (const T *)Tt, perm, row_lse2, \
                       upstream, (T *)dS, C, HW, g, q.G, q.nchunk, q.iters, c2, coef)
    if (q.vec && g_bwd_nt) {
        if (U == 2) SD_BWD(true, true, 2);
        else if (U == 8) SD_BWD(true, true, 8);
        else SD_BWD(true, true, 4);
    } else if (q.vec) {
        if (U == 2) SD_BWD(true, false, 2);
        else if (U == 8) SD_BWD(true, false, 8);
        else SD_BWD(true, false, 4);
    } else {
        if (U == 2) SD_BWD(false, false, 2);
        else if (U == 8) SD_BWD(false, false, 8);
        else SD_BWD(false, false, 4);
    }
#undef SD_BWD
    return (int)hipGetLastError();
}

}  // namespace

void launch_row_finalize_strided(const RowPart *part, float *row_lse2, float *row_kl, float *loss, int B, int C, int g, int nchunk,
                                 long batch_stride, long slot_stride, long chunk_stride, const int32_t *chan_of_slot, float c2, float inv_tau,
                                 float loss_scale, hipStream_t st) {
    const int G = (C + g - 1) / g, rows = B * G;
    const int rows_per_wg = kThreads / 64;
    hipLaunchKernelGGL(cgd_fwd_rows, dim3((rows + rows_per_wg - 1) / rows_per_wg), dim3(kThreads), 0, st, part, row_lse2, row_kl, rows,
                       C, g, G, nchunk, c2, inv_tau, batch_stride, slot_stride, chunk_stride, chan_of_slot);
    hipLaunchKernelGGL(cgd_fwd_loss, dim3(1), dim3(kThreads), 0, st, row_kl, loss, rows, loss_scale);
}

void launch_row_finalize(const RowPart *part, float *row_lse2, float *row_kl, float *loss, int B, int C, int g, int nchunk,
                         float c2, float inv_tau, float loss_scale, hipStream_t st) {
    launch_row_finalize_strided(part, row_lse2, row_kl, loss, B, C, g, nchunk, (long)C * nchunk, nchunk, 1, nullptr, c2, inv_tau, loss_scale, st);
}

int cgd_tunable(const char *key, int set, int v) {
    int *p = nullptr;
    int lo = 1, hi = 4096;
    if (!strcmp(key, "cgd_fwd_chunk_iters")) p = &g_fwd_iters;
    else if (!strcmp(key, "cgd_bwd_chunk_iters")) p = &g_bwd_iters;
    else if (!strcmp(key, "cgd_bwd_nt_store")) { p = &g_bwd_nt; lo = 0; hi = 1; }
    else if (!strcmp(key, "cgd_bwd_unroll")) {
        if (set && v != 2 && v != 4 && v != 8) return SD_E_SHAPE;
        p = &g_bwd_unroll; lo = 2; hi = 8;
    }
    if (!p) return SD_E_UNSUPPORTED;
    if (!set) return *p;
    if (v < lo || v > hi) return SD_E_SHAPE;
    *p = v;
    return SD_OK;
}

}  // namespace sd

extern "C" {

size_t sd_cgd_kl_workspace_bytes(int B, int C, int H, int W, int g) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || g <= 0) return 0;
    // Upper bound over both dtypes and the scalar fallback: the smallest chunk is one
    // iteration of the scalar build.
    const long HW = (long)H * W;
    const long min_chunk = (long)sd::kThreads * sd::kUnroll;  // N = 1, iters = 1
    long iters = sd::g_fwd_iters;
    const long need = (HW + min_chunk - 1) / min_chunk;
    if (iters > need) iters = need;
    const long chunk = min_chunk * iters;
    const long nchunk = (HW + chunk - 1) / chunk;
    return (size_t)B * C * nchunk * sizeof(sd::RowPart) + 16;
}

int sd_cgd_kl_fwd(const void *S, const void *T, int dtype, int B, int C, int H, int W, int g, float inv_tau, float loss_scale,
                  const int32_t *perm, float *row_lse2, float *row_kl, float *loss, void *workspace, size_t workspace_bytes,
                  void *stream) {
    int rc = sd::check_common(S, T, dtype, B, C, H, W, g);
    if (rc) return rc;
    if (!row_lse2 || !row_kl || !loss || !workspace) return SD_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32)
        return sd::fwd_impl<float>(S, T, B, C, H, W, g, inv_tau, loss_scale, perm, row_lse2, row_kl, loss, workspace, workspace_bytes, st);
    return sd::fwd_impl<sd::bf16_t>(S, T, B, C, H, W, g, inv_tau, loss_scale, perm, row_lse2, row_kl, loss, workspace, workspace_bytes,
                                    st);
}

int sd_cgd_kl_bwd(const void *S, const void *T, int dtype, int B, int C, int H, int W, int g, float inv_tau, float coef,
                  const int32_t *perm, const float *row_lse2, const float *upstream, void *dS, void *stream) {
    int rc = sd::check_common(S, T, dtype, B, C, H, W, g);
    if (rc) return rc;
    if (!row_lse2 || !dS) return SD_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32) return sd::bwd_impl<float>(S, T, B, C, H, W, g, inv_tau, coef, perm, row_lse2, upstream, dS, st);
    return sd::bwd_impl<sd::bf16_t>(S, T, B, C, H, W, g, inv_tau, coef, perm, row_lse2, upstream, dS, st);
}

}  // extern "C"
