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
constexpr int kMaxSliceSteps = 256;  // steps of 16 pixels per slice of ifvd_onehot_sums (its LDS copy of the step masks)

// grid (B), 1024 threads.  Also the class blocks present in every step of 16 pixels, per class group: smask[b][step][kg] bit j = some pixel of
// the step has a class in [160 kg + 32 j, + 32) -- computed once here instead of by every (channel block, network) workgroup of the products.
__global__ __launch_bounds__(1024) void ifvd_counts(const int *__restrict__ cls, int *__restrict__ counts, int *__restrict__ smask, int HW, int K,
                                                     int KG) {
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
    const int nsteps = (HW + 15) / 16;
    for (int i = threadIdx.x; i < nsteps * KG; i += 1024) {
        const int step = i / KG, kg = i - step * KG;
        int m = 0;
        for (int e = 0; e < 16; ++e) {
            const int p = step * 16 + e;
            const int c = p < HW ? row[p] : -1;
            const int rel = c - kg * kKGroup;
            if (c >= 0 && c < K && rel >= 0 && rel < kKGroup) m |= 1 << (rel >> 5);
        }
        smask[((size_t)b * nsteps + step) * KG + kg] = m;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += 1024) counts[(size_t)b * K + k] = cbins[k];
}

// grid (S slices, channel blocks * class groups, images * networks), 256 threads.  part[z = b * nt + t][s][Cp][Kp] (k contiguous).
// MFMA operands (v_mfma_f32_32x32x16_bf16): lane (r = lane & 31, g = lane >> 5) holds A[row r][k = 8 g .. 8 g + 7] and B[k = 8 g ..][column r];
// here row = class within its block, k = pixel within the 16-pixel step, column = channel within its block.
//   !WEIGHTED (class means): network t = z % nt reads X0 or X1
//   WEIGHTED  (backward):   alpha * X0; channel index C is the beta plane itself (-> B_k)
// Vector-instruction bound: ~400 per step of 16 pixels x 32 channels (label tests, one-hot rows, the three-way split) against 15 MFMAs when all
// five class blocks are present.  (Tried, same box: two feature blocks per wave sharing the one-hot rows -- 320 instructions per block-step but
// 356 registers, one wave per SIMD: 178 vs 115 us for both networks' means.)
template <typename T, bool WEIGHTED, bool FAST>
__global__ __launch_bounds__(256) void ifvd_onehot_sums(const T *__restrict__ X0, const T *__restrict__ X1, const float *__restrict__ alpha,
                                                         const float *__restrict__ beta, const int *__restrict__ cls, const int *__restrict__ smask,
                                                         float *__restrict__ part, int C, int HW, int K, int S, int Ls, int KG, int Cp) {
    constexpr bool kOnePlane = sizeof(T) == 2 && !WEIGHTED;           // bf16 features: already one exact bf16 term
    __shared__ float tile[kNKB][32][33];
    constexpr int NQ = 1;                                             // feature blocks per wave
    const int nt = (!WEIGHTED && X1) ? 2 : 1;
    const int s = blockIdx.x, op = blockIdx.y / KG, kg = blockIdx.y - op * KG, b = blockIdx.z / nt, tnet = blockIdx.z - b * nt;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, g = lane >> 5;
    const int kb0 = kg * kKGroup, nsteps = (HW + 15) / 16;
    const int Call = WEIGHTED ? C + 1 : C;                            // WEIGHTED: the beta plane is channel C
    // the B operand(s) of this wave: q -> channel block
    int ch[NQ];
    bool live_q[NQ], is_beta[NQ];
    const T *xrow[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        ch[q] = (NQ * op + q) * 32 + r;
        live_q[q] = (NQ * op + q) * 32 < Call;                        // wave-uniform: the whole block exists
        is_beta[q] = WEIGHTED && ch[q] == C;
        const T *X = tnet ? X1 : X0;
        xrow[q] = X + ((size_t)b * C + min(ch[q], C - 1)) * HW;       // lanes beyond the last channel read the last one (never written out)
    }
    const int *crow = cls + (size_t)b * HW;
    const float *arow = WEIGHTED ? alpha + (size_t)b * HW : nullptr, *brow = WEIGHTED ? beta + (size_t)b * HW : nullptr;
    const bool vec_x = HW % 8 == 0 && (reinterpret_cast<uintptr_t>(X0) & 15) == 0 && (reinterpret_cast<uintptr_t>(X1) & 15) == 0;
    const bool vec_c = HW % 4 == 0 && (reinterpret_cast<uintptr_t>(cls) & 15) == 0 &&
                       (!WEIGHTED || ((reinterpret_cast<uintptr_t>(alpha) & 15) == 0 && (reinterpret_cast<uintptr_t>(beta) & 15) == 0));
    f32x16 acc[NQ][kNKB];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int j = 0; j < kNKB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[q][j][e] = 0.f;
    const int p_end = min((s + 1) * Ls, HW);
    // the slice's step masks, once (a mask read in front of each step's loads would put a second memory latency into every round)
    __shared__ int smask_l[kMaxSliceSteps];
    for (int i = threadIdx.x; i < (p_end - s * Ls + 15) / 16; i += 256) smask_l[i] = smask[((size_t)b * nsteps + (s * Ls >> 4) + i) * KG + kg];
    __syncthreads();
    // U steps of 16 pixels per round: all their loads are requested before the first is consumed (a step alone runs at the memory latency)
    constexpr int U = 4;
    for (int q0 = s * Ls + 16 * wave; q0 < p_end; q0 += 64 * U) {
        float x[U][NQ][8], w[U][8];
        int ck[U][8], present[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p0 = q0 + 64 * u, pix = p0 + 8 * g;
            // FAST (the launcher's check): 16-byte aligned rows, whole rounds in every slice -- no tail, no scalar path in the loop
            const bool live = FAST || p0 < p_end;
            present[u] = live ? __builtin_amdgcn_readfirstlane(smask_l[(p0 - s * Ls) >> 4]) : 0;
            const bool whole = FAST || (live && pix + 8 <= HW);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (!live_q[q] || !live) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[u][q][e] = 0.f;
                } else if (is_beta[q]) {
                    if (FAST) {
                        const float4 b0 = *reinterpret_cast<const float4 *>(brow + pix), b1 = *reinterpret_cast<const float4 *>(brow + pix + 4);
                        x[u][q][0] = b0.x, x[u][q][1] = b0.y, x[u][q][2] = b0.z, x[u][q][3] = b0.w;
                        x[u][q][4] = b1.x, x[u][q][5] = b1.y, x[u][q][6] = b1.z, x[u][q][7] = b1.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[u][q][e] = pix + e < HW ? brow[pix + e] : 0.f;
                    }
                } else if (FAST || (whole && vec_x)) {
                    if constexpr (sizeof(T) == 4) {
                        VecIO<T>::load(xrow[q] + pix, reinterpret_cast<float(&)[4]>(x[u][q][0]));
                        VecIO<T>::load(xrow[q] + pix + 4, reinterpret_cast<float(&)[4]>(x[u][q][4]));
                    } else {
                        VecIO<T>::load(xrow[q] + pix, x[u][q]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[u][q][e] = (live && pix + e < HW) ? VecIO<T>::load1(xrow[q] + pix + e) : 0.f;
                }
            }
            if (FAST || (whole && vec_c)) {
                const int4 c0 = *reinterpret_cast<const int4 *>(crow + pix), c1 = *reinterpret_cast<const int4 *>(crow + pix + 4);
                ck[u][0] = c0.x, ck[u][1] = c0.y, ck[u][2] = c0.z, ck[u][3] = c0.w, ck[u][4] = c1.x, ck[u][5] = c1.y, ck[u][6] = c1.z, ck[u][7] = c1.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) ck[u][e] = (live && pix + e < HW) ? crow[pix + e] : -1;
            }
            if (WEIGHTED) {
                if (FAST || (whole && vec_c)) {
                    const float4 w0 = *reinterpret_cast<const float4 *>(arow + pix), w1 = *reinterpret_cast<const float4 *>(arow + pix + 4);
                    w[u][0] = w0.x, w[u][1] = w0.y, w[u][2] = w0.z, w[u][3] = w0.w, w[u][4] = w1.x, w[u][5] = w1.y, w[u][6] = w1.z, w[u][7] = w1.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) w[u][e] = (live && pix + e < HW) ? arow[pix + e] : 0.f;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!present[u]) continue;                                 // wave-uniform (smask: no pixel of the step has a class of this group)
            // labels relative to the group and to this lane's row: a label matches row 32 j + r  <=>  label - r == 32 j; a label that is not a
            // class of this group (or not a class at all) is sent out of range so that it matches no j
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned rel = (unsigned)(ck[u][e] - kb0);
                ck[u][e] = (ck[u][e] < K && rel < (unsigned)kKGroup) ? (int)rel - r : -1;     // (negative labels: rel wraps to a huge value)
            }
            bf16x8 xh[NQ], xm[NQ], xl[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (!live_q[q]) continue;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    f32x2 v = {x[u][q][e], x[u][q][e + 1]};
                    if (WEIGHTED && !is_beta[q]) v[0] *= w[u][e], v[1] *= w[u][e + 1];
                    const bf16x2 hh = __builtin_convertvector(v, bf16x2);
                    xh[q][e] = hh[0], xh[q][e + 1] = hh[1];
                    if constexpr (!kOnePlane) {
                        const f32x2 r1 = v - __builtin_convertvector(hh, f32x2);
                        const bf16x2 mm = __builtin_convertvector(r1, bf16x2);
                        const f32x2 r2 = r1 - __builtin_convertvector(mm, f32x2);
                        const bf16x2 ll = __builtin_convertvector(r2, bf16x2);
                        xm[q][e] = mm[0], xm[q][e + 1] = mm[1], xl[q][e] = ll[0], xl[q][e + 1] = ll[1];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < kNKB; ++j) {
                if (!((present[u] >> j) & 1)) continue;                // wave-uniform
                bf16x8 a;
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = ck[u][e] == 32 * j ? (__bf16)1.0f : (__bf16)0.0f;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    if (!live_q[q]) continue;
                    if constexpr (!kOnePlane) {
                        acc[q][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xl[q], acc[q][j], 0, 0, 0);  // small terms first
                        acc[q][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xm[q], acc[q][j], 0, 0, 0);
                    }
                    acc[q][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xh[q], acc[q][j], 0, 0, 0);
                }
            }
        }
    }
    // per operand: the four waves' accumulators, added in wave order (accumulator layout: column = lane & 31, row = (e & 3) + 8 (e >> 2) +
    // 4 (lane >> 5)), then written out class-contiguous
    const int Kp = KG * kKGroup;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (!live_q[q]) continue;
        for (int w4 = 0; w4 < 4; ++w4) {
            if (wave == w4) {
#pragma unroll
                for (int j = 0; j < kNKB; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = (e & 3) + 8 * (e >> 2) + 4 * g;
                        if (w4 == 0) tile[j][row][r] = acc[q][j][e];
                        else tile[j][row][r] += acc[q][j][e];
                    }
            }
            __syncthreads();
        }
        const int z = blockIdx.z, cblock = NQ * op + q;
        float *dst = part + (((size_t)z * S + s) * Cp + cblock * 32) * Kp + kb0;
        for (int i = threadIdx.x; i < 32 * kKGroup; i += 256) {
            const int cl = i / kKGroup, kk = i - cl * kKGroup;
            dst[(size_t)cl * Kp + kk] = tile[kk >> 5][kk & 31][cl];
        }
        __syncthreads();
    }
}

// grid (C [+ 1 with outB], Z), 256 threads over k: out[z][c][k] = sum_s part[z][s][c][k]  (/ (n_k + 1e-6) in mean mode; n from counts[z / nt][k]);
// channel index C (backward only): the beta plane's sums -> outB[b][k]
__global__ __launch_bounds__(256) void ifvd_finish(const float *__restrict__ part, const int *__restrict__ counts, float *__restrict__ out0,
                                                    float *__restrict__ out1, float *__restrict__ outB, int C, int K, int S, int Cp, int Kp, int nt,
                                                    int mean_mode) {
    const int c = blockIdx.x, z = blockIdx.y, b = z / nt;
    float *out = (z - b * nt) ? out1 : out0;
    for (int k = threadIdx.x; k < K; k += 256) {
        float acc = 0.f;
        for (int s = 0; s < S; ++s) acc += part[(((size_t)z * S + s) * Cp + c) * Kp + k];
        if (mean_mode) acc /= (float)counts[(size_t)b * K + k] + 1e-6f;
        if (c == C) outB[(size_t)b * K + k] = acc;
        else out[((size_t)b * C + c) * K + k] = acc;
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

struct SumPlan { int S, Ls, OPS, KG, Cp; size_t floats; };
// weighted: the backward's product (channels + the beta plane).  Slices of the pixels so that the launch has ~1024 workgroups (a slice: whole 64-pixel rounds of the four waves).
SumPlan sum_plan(int B, int nt, int C, int HW, int K, bool weighted) {
    SumPlan p;
    const int CB = ((weighted ? C + 1 : C) + 31) / 32;
    p.OPS = CB;
    p.Cp = CB * 32;
    p.KG = (K + kKGroup - 1) / kKGroup;
    const long per_slice = (long)p.OPS * p.KG * B * nt;
    int S = (int)((1024 + per_slice - 1) / per_slice);
    const int max_s = (HW + 63) / 64;
    S = S < 1 ? 1 : (S > max_s ? max_s : S);
    if (S > 32) S = 32;
    const int min_s = (HW + 16 * kMaxSliceSteps - 1) / (16 * kMaxSliceSteps);
    if (S < min_s) S = min_s;
    p.Ls = (((HW + S - 1) / S + 255) / 256) * 256;          // whole rounds of four waves x four 16-pixel steps
    p.S = (HW + p.Ls - 1) / p.Ls;
    p.floats = (size_t)B * nt * p.S * p.Cp * (p.KG * kKGroup);
    return p;
}

// sums over the classes -> tables [B][C][K]; workspace = the slices' partial tables.  !WEIGHTED: X0 (and X1) -> out0 (out1), divided by the
// class sizes.  WEIGHTED: alpha * X0 -> out0 and the beta plane -> outB.
template <typename T, bool WEIGHTED>
int class_sums(const T *X0, const T *X1, const float *alpha, const float *beta, const int *cls, const int *smask, const int *counts, float *out0,
               float *out1, float *outB, void *workspace, size_t workspace_bytes, int B, int C, int HW, int K, hipStream_t st) {
    const int nt = (!WEIGHTED && X1) ? 2 : 1;
    const SumPlan p = sum_plan(B, nt, C, HW, K, WEIGHTED);
    if (!workspace || workspace_bytes < p.floats * sizeof(float) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return SD_E_WORKSPACE;
    if ((long)p.OPS * p.KG > 65535 || B * nt > 65535) return SD_E_SHAPE;
    float *part = static_cast<float *>(workspace);
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool fast = HW % 256 == 0 && al16(X0) && al16(X1) && al16(cls) && al16(alpha) && al16(beta);
    const dim3 grid(p.S, p.OPS * p.KG, B * nt);
    if (fast)
        hipLaunchKernelGGL((ifvd_onehot_sums<T, WEIGHTED, true>), grid, dim3(256), 0, st, X0, X1, alpha, beta, cls, smask, part, C, HW, K, p.S, p.Ls, p.KG,
                           p.Cp);
    else
        hipLaunchKernelGGL((ifvd_onehot_sums<T, WEIGHTED, false>), grid, dim3(256), 0, st, X0, X1, alpha, beta, cls, smask, part, C, HW, K, p.S, p.Ls, p.KG,
                           p.Cp);
    hipLaunchKernelGGL(ifvd_finish, dim3(C + (WEIGHTED ? 1 : 0), B * nt), dim3(256), 0, st, (const float *)part, counts, out0, out1, outB, C, K, p.S, p.Cp,
                       p.KG * kKGroup, nt, WEIGHTED ? 0 : 1);
    return (int)hipGetLastError();
}

}  // namespace
}  // namespace sd

extern "C" {

size_t sd_ifvd_workspace_bytes(int B, int C, int HW, int K) {
    if (B <= 0 || C <= 0 || HW <= 0 || K <= 0) return 0;
    // the largest of: the slices' partial class tables (class_means, coef_sums), one double per 64-pixel block (cos)
    size_t n = sd::sum_plan(B, 2, C, HW, K, false).floats * sizeof(float);
    const size_t n2 = sd::sum_plan(B, 1, C, HW, K, true).floats * sizeof(float), n3 = (size_t)((HW + 63) / 64) * B * sizeof(double);
    n = n2 > n ? n2 : n;
    return (n3 > n ? n3 : n) + 16;
}

size_t sd_ifvd_stepmask_ints(int B, int HW, int K) {
    if (B <= 0 || HW <= 0 || K <= 0) return 0;
    return (size_t)B * ((HW + 15) / 16) * ((K + sd::kKGroup - 1) / sd::kKGroup);
}

int sd_ifvd_counts(const int *cls, int B, int HW, int K, int *counts, int *stepmask, void *stream) {
    if (!cls || !counts || !stepmask) return SD_E_NULL;
    if (B <= 0 || HW <= 0 || K <= 0 || K > 8192) return SD_E_SHAPE;
    hipLaunchKernelGGL(sd::ifvd_counts, dim3(B), dim3(1024), (size_t)K * sizeof(int), static_cast<hipStream_t>(stream), cls, counts, stepmask, HW, K,
                       (K + sd::kKGroup - 1) / sd::kKGroup);
    return (int)hipGetLastError();
}

int sd_ifvd_class_means(const void *S, const void *T, int dtype, const int *cls, const int *stepmask, const int *counts, float *mean_s, float *mean_t,
                        void *workspace, size_t workspace_bytes, int B, int C, int HW, int K, void *stream) {
    int rc = sd::check_ifvd(S, dtype, B, C, HW, K);
    if (rc) return rc;
    if (!cls || !stepmask || !counts || !mean_s || (T && !mean_t)) return SD_E_NULL;
    if (T && (reinterpret_cast<uintptr_t>(T) & (dtype == SD_F32 ? 3 : 1))) return SD_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32)
        return sd::class_sums<float, false>((const float *)S, (const float *)T, nullptr, nullptr, cls, stepmask, counts, mean_s, mean_t, nullptr, workspace,
                                            workspace_bytes, B, C, HW, K, st);
    return sd::class_sums<sd::bf16_t, false>((const sd::bf16_t *)S, (const sd::bf16_t *)T, nullptr, nullptr, cls, stepmask, counts, mean_s, mean_t, nullptr,
                                             workspace, workspace_bytes, B, C, HW, K, st);
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

int sd_ifvd_coef_sums(const void *S, int dtype, const int *cls, const int *stepmask, const int *counts, const float *coefs, float *A, float *Bk,
                      void *workspace, size_t workspace_bytes, int B, int C, int HW, int K, void *stream) {
    int rc = sd::check_ifvd(S, dtype, B, C, HW, K);
    if (rc) return rc;
    if (!cls || !stepmask || !counts || !coefs || !A || !Bk) return SD_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float *alpha = coefs, *beta = coefs + (size_t)B * HW;
    if (dtype == SD_F32)
        return sd::class_sums<float, true>((const float *)S, nullptr, alpha, beta, cls, stepmask, counts, A, nullptr, Bk, workspace, workspace_bytes, B, C, HW,
                                           K, st);
    return sd::class_sums<sd::bf16_t, true>((const sd::bf16_t *)S, nullptr, alpha, beta, cls, stepmask, counts, A, nullptr, Bk, workspace, workspace_bytes,
                                            B, C, HW, K, st);
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
