// ifvd.hip -- the intra-class feature-variation term of IFVDLoss (reference losses.py:199-238), gfx950.
//
//   centre_k = mean of the features of the pixels labelled k (per image)        (:226-230, a 150-pass full-tensor mask loop)
//   sim_p    = cos(feature_p, centre_{label(p)})                                 (:231-233)
//   loss     = 10 * mean_p (sim^S_p - sim^T_p)^2                                 (:235)
// and its gradient with respect to the student feature, which the reference's autograd also takes THROUGH the class centres.
//
// A class sum over an image is a product with the one-hot label matrix:  sums[k][c] = sum_p [label_p == k] * X[c][p]  -- GEMM-shaped
// floating-point work with a reduction over all pixels, so it runs on the matrix pipe: the one-hot operand is EXACT in bf16, the features
// are split exactly into three bf16 terms (fp32-grade, as in token_gemm.hip), every product 1.0 * x is exact and the accumulation is the
// MFMA's fp32 in a fixed order -- deterministic, no sort of the pixels, no gathers, no float atomics, the features read once, coalesced.
// (History: round 2 sorted the pixels by class with a torch sort chain and walked each run with one dependent gather per lane: 650 us of
// device time forward at 8 x 150 x 128 x 128, ~45 launches.  Round 3 first moved the sort into a counting-sort kernel and gathered from an
// LDS image of the channel plane with a segmented scan: 220 us, vector-instruction bound -- ~75 instructions per 64 elements.)
//   ifvd_counts        n_k per image (integer LDS atomics)
//   ifvd_onehot_sums   per (image, block of 32 channels, slice of the pixels): for every step of 16 pixels the lanes build the one-hot rows of
//                      the class blocks PRESENT in the step (a wave-uniform test: label maps are spatially coherent), split their 8 feature
//                      values and issue 3 MFMAs per present class block; the four waves' accumulators are added in wave order through LDS
//   ifvd_finish        adds the slices in order (and divides by n_k + 1e-6) -> tables [B][C][K]: the per-pixel passes read one channel's K
//                      values per wave, whatever the labels of the 64 pixels are
//   ifvd_cos           per pixel, both networks: dot / norms against the class centre, the channels split over the four waves (64 pixels per
//                      workgroup), the squared difference and the per-pixel gradient coefficients alpha, beta, gamma
//   ifvd_bwd           dS[b,c,p] = g * ( alpha_p * mu_k[c] - gamma_p * S[b,c,p] + (A_k[c] - mu_k[c] * B_k) / (n_k + 1e-6) )
// with alpha = w/(|a||mu|), gamma = w*sim/|a|^2, beta = w*sim/|mu|^2, w = 20 (sim^S - sim^T)/(B*HW),
// A_k[c] = sum_{p in k} alpha_p S[c,p], B_k = sum_{p in k} beta_p (the same one-hot products, on alpha * S and on the beta plane).
// Pixels without a class (label outside [0,K)) compare a feature with itself: similarity 1, gradient 0.
// HBM-bound at tap resolution (78 MB per tensor at config-2 sizes): forward 4 N e, backward 3 N e algorithmic bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cgd_device.h"

namespace sd {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr float kCosEps = 1e-8f;   // F.cosine_similarity's eps: each norm is clamped from below
constexpr int kNKB = 5;            // class blocks of 32 per pass of ifvd_onehot_sums (160 classes; more classes: more passes, grid.y)
constexpr int kKGroup = 32 * kNKB;

// grid (B), 1024 threads
__global__ __launch_bounds__(1024) void ifvd_counts(const int *__restrict__ cls, int *__restrict__ counts, int HW, int K) {
    extern __shared__ int cbins[];
    const int b = blockIdx.x;
    const int *row = cls + (size_t)b * HW;
    for (int k = threadIdx.x; k < K; k += 1024) cbins[k] = 0;
    __syncthreads();
    auto count = [&](int c) {
        if (c >= 0 && c < K) atomicAdd(&cbins[c], 1);                 // integers: any order gives the same counts
    };
    if (HW % 4 == 0 && (reinterpret_cast<uintptr_t>(cls) & 15) == 0) {
        for (int p = threadIdx.x * 4; p < HW; p += 4096) {
            const int4 c = *reinterpret_cast<const int4 *>(row + p);
            count(c.x), count(c.y), count(c.z), count(c.w);
        }
    } else {
        for (int p = threadIdx.x; p < HW; p += 1024) count(row[p]);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += 1024) counts[(size_t)b * K + k] = cbins[k];
}

// grid (S slices, channel blocks * class groups, images * tensors), 256 threads.  part[z][s][Cp][Kp] (k contiguous).
// MFMA operands (v_mfma_f32_32x32x16_bf16): lane (r = lane & 31, g = lane >> 5) holds A[row r][k = 8 g .. 8 g + 7] and B[k = 8 g ..][column r];
// here row = class within its block, k = pixel within the 16-pixel step, column = channel within its block.
template <typename T, bool WEIGHTED>
__global__ __launch_bounds__(256) void ifvd_onehot_sums(const T *__restrict__ X0, const T *__restrict__ X1, const float *__restrict__ wgt,
                                                         const int *__restrict__ cls, float *__restrict__ part, int C, int HW, int K, int S, int Ls,
                                                         int KG, int nt) {
    constexpr bool kOnePlane = sizeof(T) == 2 && !WEIGHTED;           // bf16 features: already one exact bf16 term
    __shared__ float tile[kNKB][32][33];
    const int s = blockIdx.x, cb = blockIdx.y / KG, kg = blockIdx.y - cb * KG, z = blockIdx.z;
    const int b = z / nt;
    const T *X = (z - b * nt) ? X1 : X0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, g = lane >> 5;
    const int c = cb * 32 + r, kb0 = kg * kKGroup;
    const bool cvalid = c < C;
    const T *xrow = X + ((size_t)b * C + (cvalid ? c : C - 1)) * HW;
    const int *crow = cls + (size_t)b * HW;
    const float *wrow = WEIGHTED ? wgt + (size_t)b * HW : nullptr;
    const bool vec_x = HW % 8 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0;
    const bool vec_c = HW % 4 == 0 && (reinterpret_cast<uintptr_t>(cls) & 15) == 0 && (!WEIGHTED || (reinterpret_cast<uintptr_t>(wgt) & 15) == 0);
    f32x16 acc[kNKB];
#pragma unroll
    for (int j = 0; j < kNKB; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    const int p_end = min((s + 1) * Ls, HW);
    // U steps of 16 pixels per round: all their loads are requested before the first is consumed (a step alone would run at the memory
    // latency -- 32 bytes per lane in flight; measured 128 us for both networks at 8 x 150 x 128 x 128, whatever the labels)
    constexpr int U = 4;
    for (int q0 = s * Ls + 16 * wave; q0 < p_end; q0 += 64 * U) {
        float x[U][8], w[U][8];
        int ck[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pix = q0 + 64 * u + 8 * g;
            const bool live = q0 + 64 * u < p_end;
            const bool whole = live && pix + 8 <= HW;
            if (whole && vec_x) {
                if constexpr (sizeof(T) == 4) {
                    VecIO<T>::load(xrow + pix, reinterpret_cast<float(&)[4]>(x[u][0]));
                    VecIO<T>::load(xrow + pix + 4, reinterpret_cast<float(&)[4]>(x[u][4]));
                } else {
                    VecIO<T>::load(xrow + pix, x[u]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[u][e] = (live && pix + e < HW) ? VecIO<T>::load1(xrow + pix + e) : 0.f;
            }
            if (whole && vec_c) {
                const int4 c0 = *reinterpret_cast<const int4 *>(crow + pix), c1 = *reinterpret_cast<const int4 *>(crow + pix + 4);
                ck[u][0] = c0.x, ck[u][1] = c0.y, ck[u][2] = c0.z, ck[u][3] = c0.w, ck[u][4] = c1.x, ck[u][5] = c1.y, ck[u][6] = c1.z, ck[u][7] = c1.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) ck[u][e] = (live && pix + e < HW) ? crow[pix + e] : -1;
            }
            if (WEIGHTED) {
                if (whole && vec_c) {
                    const float4 w0 = *reinterpret_cast<const float4 *>(wrow + pix), w1 = *reinterpret_cast<const float4 *>(wrow + pix + 4);
                    w[u][0] = w0.x, w[u][1] = w0.y, w[u][2] = w0.z, w[u][3] = w0.w, w[u][4] = w1.x, w[u][5] = w1.y, w[u][6] = w1.z, w[u][7] = w1.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) w[u][e] = (live && pix + e < HW) ? wrow[pix + e] : 0.f;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // class blocks present among the 16 pixels of the step (the lanes of a half-wave hold the same 8 labels)
            int present = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int rel = ck[u][e] - kb0;
                const bool in = ck[u][e] >= 0 && ck[u][e] < K && rel >= 0 && rel < kKGroup;
                ck[u][e] = in ? rel : -1;                              // relative to the group; -1: matches no row
                present |= in ? 1 << (rel >> 5) : 0;
            }
            present = __builtin_amdgcn_readlane(present, 0) | __builtin_amdgcn_readlane(present, 32);
            if (!present) continue;
            bf16x8 xh, xm, xl;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                f32x2 v = {cvalid ? x[u][e] : 0.f, cvalid ? x[u][e + 1] : 0.f};
                if (WEIGHTED) v[0] *= w[u][e], v[1] *= w[u][e + 1];
                const bf16x2 hh = __builtin_convertvector(v, bf16x2);
                xh[e] = hh[0], xh[e + 1] = hh[1];
                if constexpr (!kOnePlane) {
                    const f32x2 r1 = v - __builtin_convertvector(hh, f32x2);
                    const bf16x2 mm = __builtin_convertvector(r1, bf16x2);
                    const f32x2 r2 = r1 - __builtin_convertvector(mm, f32x2);
                    const bf16x2 ll = __builtin_convertvector(r2, bf16x2);
                    xm[e] = mm[0], xm[e + 1] = mm[1], xl[e] = ll[0], xl[e + 1] = ll[1];
                }
            }
#pragma unroll
            for (int j = 0; j < kNKB; ++j) {
                if (!((present >> j) & 1)) continue;                   // wave-uniform
                const int row = 32 * j + r;
                bf16x8 a;
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = ck[u][e] == row ? (__bf16)1.0f : (__bf16)0.0f;
                if constexpr (!kOnePlane) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xl, acc[j], 0, 0, 0);  // small terms first
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xm, acc[j], 0, 0, 0);
                }
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xh, acc[j], 0, 0, 0);
            }
        }
    }
    // the four waves' accumulators, added in wave order (accumulator layout: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5))
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int j = 0; j < kNKB; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = (e & 3) + 8 * (e >> 2) + 4 * g;
                    if (w == 0) tile[j][row][r] = acc[j][e];
                    else tile[j][row][r] += acc[j][e];
                }
        }
        __syncthreads();
    }
    const int CB = gridDim.y / KG, Cp = CB * 32, Kp = KG * kKGroup;
    float *dst = part + (((size_t)z * S + s) * Cp + cb * 32) * Kp + kb0;
    for (int i = threadIdx.x; i < 32 * kKGroup; i += 256) {
        const int cl = i / kKGroup, kk = i - cl * kKGroup;
        dst[(size_t)cl * Kp + kk] = tile[kk >> 5][kk & 31][cl];
    }
}

// grid (C, Z), 256 threads over k: out[z][c][k] = sum_s part[z][s][c][k]  (/ (n_k + 1e-6) in mean mode; n from counts[z / nt][k])
__global__ __launch_bounds__(256) void ifvd_finish(const float *__restrict__ part, const int *__restrict__ counts, float *__restrict__ out0,
                                                    float *__restrict__ out1, int C, int K, int S, int Cp, int Kp, int nt, int mean_mode) {
    const int c = blockIdx.x, z = blockIdx.y, b = z / nt;
    float *out = (z - b * nt) ? out1 : out0;
    for (int k = threadIdx.x; k < K; k += 256) {
        float acc = 0.f;
        for (int s = 0; s < S; ++s) acc += part[(((size_t)z * S + s) * Cp + c) * Kp + k];
        if (mean_mode) acc /= (float)counts[(size_t)b * K + k] + 1e-6f;
        out[((size_t)b * C + c) * K + k] = acc;
    }
}

// grid (ceil(HW/64), B), 256 threads: lane = pixel, wave = a quarter of the channels.
template <typename T>
__global__ __launch_bounds__(256) void ifvd_cos(const T *__restrict__ S, const T *__restrict__ Tt, const int *__restrict__ cls,
                                                 const float *__restrict__ mean_s, const float *__restrict__ mean_t, float *__restrict__ coefs,
                                                 double *__restrict__ wg_sum, int C, int HW, int K, long BHW, float w_scale) {
    __shared__ float red[6][4][64];
    const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = blockIdx.x * 64 + lane;
    const bool inb = p < HW;
    const int k = inb ? cls[(size_t)b * HW + p] : -1;
    const bool valid = k >= 0 && k < K;
    const int Cq = (C + 3) / 4, c0 = wave * Cq, c1 = min(C, c0 + Cq);
    const T *ps = S + (size_t)b * C * HW + (inb ? p : 0), *pt = Tt + (size_t)b * C * HW + (inb ? p : 0);
    const float *ms = mean_s + (size_t)b * C * K + (valid ? k : 0), *mt = mean_t + (size_t)b * C * K + (valid ? k : 0);   // [B][C][K]
    float ds = 0.f, as = 0.f, bs = 0.f, dt = 0.f, at = 0.f, bt = 0.f;
    int c = c0;
    for (; c + 4 <= c1; c += 4) {
        float xs[4], xt[4], us[4], ut[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xs[u] = VecIO<T>::load1(ps + (size_t)(c + u) * HW);
            xt[u] = VecIO<T>::load1(pt + (size_t)(c + u) * HW);
            us[u] = ms[(size_t)(c + u) * K];
            ut[u] = mt[(size_t)(c + u) * K];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float m1 = valid ? us[u] : xs[u], m2 = valid ? ut[u] : xt[u];
            ds = fmaf(xs[u], m1, ds), as = fmaf(xs[u], xs[u], as), bs = fmaf(m1, m1, bs);
            dt = fmaf(xt[u], m2, dt), at = fmaf(xt[u], xt[u], at), bt = fmaf(m2, m2, bt);
        }
    }
    for (; c < c1; ++c) {
        const float x1 = VecIO<T>::load1(ps + (size_t)c * HW), x2 = VecIO<T>::load1(pt + (size_t)c * HW);
        const float m1 = valid ? ms[(size_t)c * K] : x1, m2 = valid ? mt[(size_t)c * K] : x2;
        ds = fmaf(x1, m1, ds), as = fmaf(x1, x1, as), bs = fmaf(m1, m1, bs);
        dt = fmaf(x2, m2, dt), at = fmaf(x2, x2, at), bt = fmaf(m2, m2, bt);
    }
    red[0][wave][lane] = ds, red[1][wave][lane] = as, red[2][wave][lane] = bs;
    red[3][wave][lane] = dt, red[4][wave][lane] = at, red[5][wave][lane] = bt;
    __syncthreads();
    if (wave != 0) return;
    float v[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) v[q] = (red[q][0][lane] + red[q][1][lane]) + (red[q][2][lane] + red[q][3][lane]);
    double d2 = 0.0;
    if (inb) {
        const float na = fmaxf(sqrtf(v[1]), kCosEps), nb = fmaxf(sqrtf(v[2]), kCosEps);
        const float ta = fmaxf(sqrtf(v[4]), kCosEps), tb = fmaxf(sqrtf(v[5]), kCosEps);
        const float s = v[0] / (na * nb), tsim = v[3] / (ta * tb);
        const float d = s - tsim;
        d2 = (double)(d * d);
        const float w = valid ? w_scale * d : 0.f;
        const float alpha = w / (na * nb), beta = w * s / (nb * nb), gamma = w * s / (na * na);
        const size_t at = (size_t)b * HW + p;
        coefs[at] = alpha;
        coefs[BHW + at] = beta;
        coefs[2 * BHW + at] = gamma;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d2 += __shfl_xor(d2, o, 64);
    if (lane == 0) wg_sum[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = d2;
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

// grid (ceil(HW/64), B), 256 threads: lane = pixel, wave = a quarter of the channels
template <typename T>
__global__ __launch_bounds__(256) void ifvd_bwd(const T *__restrict__ X, const int *__restrict__ cls, const float *__restrict__ mean,
                                                 const float *__restrict__ coefs, const float *__restrict__ A, const float *__restrict__ Bk,
                                                 const int *__restrict__ counts, const float *__restrict__ upstream, T *__restrict__ dS, int C,
                                                 int HW, int K, long BHW) {
    const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = blockIdx.x * 64 + lane;
    if (p >= HW) return;
    const int k = cls[(size_t)b * HW + p];
    const bool valid = k >= 0 && k < K;
    const int Cq = (C + 3) / 4, c0 = wave * Cq, c1 = min(C, c0 + Cq);
    const float g = upstream ? upstream[0] : 1.f;
    const T *px = X + (size_t)b * C * HW + p;
    T *pd = dS + (size_t)b * C * HW + p;
    if (!valid) {
        for (int c = c0; c < c1; ++c) VecIO<T>::store1(pd + (size_t)c * HW, 0.f);
        return;
    }
    const float alpha = g * coefs[(size_t)b * HW + p], gamma = g * coefs[2 * BHW + (size_t)b * HW + p];
    const float invn = g / ((float)counts[(size_t)b * K + k] + 1e-6f);
    const float bk = Bk[(size_t)b * K + k];
    const float *mu = mean + (size_t)b * C * K + k, *ak = A + (size_t)b * C * K + k;                 // [B][C][K]
    int c = c0;
    for (; c + 4 <= c1; c += 4) {
        float a[4], m[4], q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = VecIO<T>::load1(px + (size_t)(c + u) * HW), m[u] = mu[(size_t)(c + u) * K], q[u] = ak[(size_t)(c + u) * K];
#pragma unroll
        for (int u = 0; u < 4; ++u) VecIO<T>::store1(pd + (size_t)(c + u) * HW, fmaf(alpha, m[u], fmaf(-gamma, a[u], (q[u] - m[u] * bk) * invn)));
    }
    for (; c < c1; ++c) {
        const float a = VecIO<T>::load1(px + (size_t)c * HW), m = mu[(size_t)c * K];
        VecIO<T>::store1(pd + (size_t)c * HW, fmaf(alpha, m, fmaf(-gamma, a, (ak[(size_t)c * K] - m * bk) * invn)));
    }
}

int check_ifvd(const void *X, int dtype, int B, int C, int HW, int K) {
    if (!X) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (B <= 0 || C <= 0 || HW <= 0 || K <= 0 || B > 32767 || C > 65535 || K > 8192) return SD_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(X) & (dtype == SD_F32 ? 3 : 1)) != 0) return SD_E_ALIGN;
    return SD_OK;
}

struct SumPlan { int S, Ls, CB, KG; size_t floats; };
// slices of the pixels so that the launch has ~1024 workgroups (a slice: whole 64-pixel rounds of the four waves)
SumPlan sum_plan(int Z, int C, int HW, int K) {
    SumPlan p;
    p.CB = (C + 31) / 32, p.KG = (K + kKGroup - 1) / kKGroup;
    const long per_slice = (long)p.CB * p.KG * Z;
    int S = (int)((1024 + per_slice - 1) / per_slice);
    const int max_s = (HW + 63) / 64;
    S = S < 1 ? 1 : (S > max_s ? max_s : S);
    if (S > 64) S = 64;
    p.Ls = (((HW + S - 1) / S + 63) / 64) * 64;
    p.S = (HW + p.Ls - 1) / p.Ls;
    p.floats = (size_t)Z * p.S * (p.CB * 32) * (p.KG * kKGroup);
    return p;
}

// sums over the classes of X0 (and X1) -> out0 (out1), tables [B][C][K]; workspace = the slices' partial tables
template <typename T, bool WEIGHTED>
int class_sums(const T *X0, const T *X1, const float *wgt, const int *cls, const int *counts, float *out0, float *out1, void *workspace,
               size_t workspace_bytes, int B, int C, int HW, int K, int mean_mode, hipStream_t st) {
    const int nt = X1 ? 2 : 1, Z = B * nt;
    const SumPlan p = sum_plan(Z, C, HW, K);
    if (!workspace || workspace_bytes < p.floats * sizeof(float) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return SD_E_WORKSPACE;
    if ((long)p.CB * p.KG > 65535 || Z > 65535) return SD_E_SHAPE;
    float *part = static_cast<float *>(workspace);
    hipLaunchKernelGGL((ifvd_onehot_sums<T, WEIGHTED>), dim3(p.S, p.CB * p.KG, Z), dim3(256), 0, st, X0, X1, wgt, cls, part, C, HW, K, p.S, p.Ls, p.KG,
                       nt);
    hipLaunchKernelGGL(ifvd_finish, dim3(C, Z), dim3(256), 0, st, (const float *)part, counts, out0, out1, C, K, p.S, p.CB * 32, p.KG * kKGroup, nt,
                       mean_mode);
    return (int)hipGetLastError();
}

}  // namespace
}  // namespace sd

extern "C" {

size_t sd_ifvd_workspace_bytes(int B, int C, int HW, int K) {
    if (B <= 0 || C <= 0 || HW <= 0 || K <= 0) return 0;
    // the larger of: the slices' partial class tables of both networks (class_means), one double per 64-pixel block (cos)
    const size_t sums = sd::sum_plan(2 * B, C, HW, K).floats * sizeof(float);
    const size_t cosb = (size_t)((HW + 63) / 64) * B * sizeof(double);
    return (sums > cosb ? sums : cosb) + 16;
}

int sd_ifvd_counts(const int *cls, int B, int HW, int K, int *counts, void *stream) {
    if (!cls || !counts) return SD_E_NULL;
    if (B <= 0 || HW <= 0 || K <= 0 || K > 8192) return SD_E_SHAPE;
    hipLaunchKernelGGL(sd::ifvd_counts, dim3(B), dim3(1024), (size_t)K * sizeof(int), static_cast<hipStream_t>(stream), cls, counts, HW, K);
    return (int)hipGetLastError();
}

int sd_ifvd_class_means(const void *S, const void *T, int dtype, const int *cls, const int *counts, float *mean_s, float *mean_t, void *workspace,
                        size_t workspace_bytes, int B, int C, int HW, int K, void *stream) {
    int rc = sd::check_ifvd(S, dtype, B, C, HW, K);
    if (rc) return rc;
    if (!cls || !counts || !mean_s || (T && !mean_t)) return SD_E_NULL;
    if (T && (reinterpret_cast<uintptr_t>(T) & (dtype == SD_F32 ? 3 : 1))) return SD_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32)
        return sd::class_sums<float, false>((const float *)S, (const float *)T, nullptr, cls, counts, mean_s, mean_t, workspace, workspace_bytes, B, C, HW,
                                            K, 1, st);
    return sd::class_sums<sd::bf16_t, false>((const sd::bf16_t *)S, (const sd::bf16_t *)T, nullptr, cls, counts, mean_s, mean_t, workspace,
                                             workspace_bytes, B, C, HW, K, 1, st);
}

int sd_ifvd_cos(const void *S, const void *T, int dtype, const int *cls, const float *mean_s, const float *mean_t, float *coefs, float *loss,
                void *workspace, size_t workspace_bytes, int B, int C, int HW, int K, void *stream) {
    int rc = sd::check_ifvd(S, dtype, B, C, HW, K);
    if (rc) return rc;
    if (!T || !cls || !mean_s || !mean_t || !coefs || !loss || !workspace) return SD_E_NULL;
    if (reinterpret_cast<uintptr_t>(T) & (dtype == SD_F32 ? 3 : 1)) return SD_E_ALIGN;
    const int gx = (HW + 63) / 64;
    if (workspace_bytes < (size_t)gx * B * sizeof(double) || (reinterpret_cast<uintptr_t>(workspace) & 7)) return SD_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long BHW = (long)B * HW;
    const float w_scale = 20.f / (float)BHW;            // d/dsim of 10 * mean (sim_s - sim_t)^2
    double *sums = static_cast<double *>(workspace);
    if (dtype == SD_F32)
        hipLaunchKernelGGL((sd::ifvd_cos<float>), dim3(gx, B), dim3(256), 0, st, (const float *)S, (const float *)T, cls, mean_s, mean_t, coefs, sums, C, HW,
                           K, BHW, w_scale);
    else
        hipLaunchKernelGGL((sd::ifvd_cos<sd::bf16_t>), dim3(gx, B), dim3(256), 0, st, (const sd::bf16_t *)S, (const sd::bf16_t *)T, cls, mean_s, mean_t,
                           coefs, sums, C, HW, K, BHW, w_scale);
    hipLaunchKernelGGL(sd::ifvd_loss, dim3(1), dim3(256), 0, st, sums, loss, gx * B, 10.f / (float)BHW);
    return (int)hipGetLastError();
}

int sd_ifvd_coef_sums(const void *S, int dtype, const int *cls, const int *counts, const float *coefs, float *A, float *Bk, void *workspace,
                      size_t workspace_bytes, int B, int C, int HW, int K, void *stream) {
    int rc = sd::check_ifvd(S, dtype, B, C, HW, K);
    if (rc) return rc;
    if (!cls || !counts || !coefs || !A || !Bk) return SD_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float *alpha = coefs, *beta = coefs + (size_t)B * HW;
    // B_k first (its partial tables are consumed by its own finish launch before the second product overwrites the workspace: stream order)
    rc = sd::class_sums<float, false>(beta, nullptr, nullptr, cls, counts, Bk, nullptr, workspace, workspace_bytes, B, 1, HW, K, 0, st);
    if (rc) return rc;
    if (dtype == SD_F32)
        return sd::class_sums<float, true>((const float *)S, nullptr, alpha, cls, counts, A, nullptr, workspace, workspace_bytes, B, C, HW, K, 0, st);
    return sd::class_sums<sd::bf16_t, true>((const sd::bf16_t *)S, nullptr, alpha, cls, counts, A, nullptr, workspace, workspace_bytes, B, C, HW, K, 0, st);
}

int sd_ifvd_bwd(const void *X, int dtype, const int *cls, const float *mean, const float *coefs, const float *A, const float *Bk,
                const int *counts, const float *upstream, void *dS, int B, int C, int HW, int K, void *stream) {
    int rc = sd::check_ifvd(X, dtype, B, C, HW, K);
    if (rc) return rc;
    if (!cls || !mean || !coefs || !A || !Bk || !counts || !dS) return SD_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int gx = (HW + 63) / 64;
    if (dtype == SD_F32)
        hipLaunchKernelGGL((sd::ifvd_bwd<float>), dim3(gx, B), dim3(256), 0, st, (const float *)X, cls, mean, coefs, A, Bk, counts, upstream, (float *)dS,
                           C, HW, K, (long)B * HW);
    else
        hipLaunchKernelGGL((sd::ifvd_bwd<sd::bf16_t>), dim3(gx, B), dim3(256), 0, st, (const sd::bf16_t *)X, cls, mean, coefs, A, Bk, counts, upstream,
                           (sd::bf16_t *)dS, C, HW, K, (long)B * HW);
    return (int)hipGetLastError();
}

}  // extern "C"
