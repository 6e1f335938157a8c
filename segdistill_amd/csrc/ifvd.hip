// ifvd.hip -- the intra-class feature-variation term of IFVDLoss (reference losses.py:199-238), gfx950.
//
//   centre_k = mean of the features of the pixels labelled k (per image)        (:226-230, a 150-pass full-tensor mask loop)
//   sim_p    = cos(feature_p, centre_{label(p)})                                 (:231-233)
//   loss     = 10 * mean_p (sim^S_p - sim^T_p)^2                                 (:235)
// and its gradient with respect to the student feature, which the reference's autograd also takes THROUGH the class centres.
//
// The label map is shared by every channel and by both networks, so the pixels of an image are grouped by class ONCE
// (order[b][HW], offsets[b][K+1], the inverse pos[b][HW]); after that a class sum is a contiguous run of the sorted index.
// Round 3: four launches forward, two backward (round 2: ~45 -- a torch sort chain, a class-sum and a cosine launch per network -- and
// 650 us of device time at 8 x 150 x 128 x 128, the class sums alone 2 x 169 us at one dependent gather in flight per lane):
//   ifvd_group        stable counting sort of the pixels by class, one workgroup per image: every wave counts a contiguous range of pixels
//                     per class (same-class lanes found with one ballot per bit of the class number), prefix over (class, wave), scatter
//   ifvd_class_sums   out[b,k,c] = sum_{p in class k} w_p * X[b,c,p]: the channel plane is read ONCE, coalesced, into LDS and gathered from
//                     there in sorted order; the four waves split the sorted index evenly (a big class does not serialise one wave) and runs
//                     cut by a wave boundary are put together in wave order.  Both networks in one launch (grid.z); the backward's two
//                     sums (alpha-weighted features, beta) in one launch (an extra "channel" whose plane is all ones)
//   ifvd_cos          per pixel, both networks: dot / norms against the class centre, the channels split over the four waves (64 pixels per
//                     workgroup: 2048 workgroups at 128 x 128 x 8, four independent loads per tensor in flight), the squared difference and
//                     the per-pixel gradient coefficients -- in pixel order for ifvd_bwd, in sorted order for the class sums
//   ifvd_bwd          dS[b,c,p] = g * ( alpha_p * mu_k[c] - gamma_p * S[b,c,p] + (A_k[c] - mu_k[c] * B_k) / (n_k + 1e-6) )
// with alpha = w/(|a||mu|), gamma = w*sim/|a|^2, beta = w*sim/|mu|^2, w = 20 (sim^S - sim^T)/(B*HW),
// A_k[c] = sum_{p in k} alpha_p S[c,p], B_k = sum_{p in k} beta_p.  Pixels without a class (label outside [0,K)) compare a feature with
// itself: similarity 1, gradient 0.  Deterministic: no float atomics, every sum in a fixed order.
// HBM-bound byte work at tap resolution (78 MB per tensor at config-2 sizes): forward 4 N e, backward 3 N e algorithmic bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cgd_device.h"

namespace sd {
namespace {

constexpr float kCosEps = 1e-8f;   // F.cosine_similarity's eps: each norm is clamped from below
constexpr int kGroupWaves = 16;    // ifvd_group: waves (= contiguous pixel ranges) per image
constexpr int kMaxKeys = 1024;     // classes + 1 ("no class") the grouping kernel takes
constexpr int kSumWaves = 8;       // ifvd_class_sums: waves per (image, channel) plane
constexpr int kPlaneLdsMax = 28672;  // pixels of a channel plane that fit the LDS image (112 KB of fp32, next to the class bins)

// lanes of the wave that are active and hold the same key as this lane
__device__ __forceinline__ unsigned long long same_key_mask(int key, bool active, int nbits) {
    unsigned long long m = __ballot(active);
    for (int b = 0; b < nbits; ++b) {
        const bool bit = (key >> b) & 1;
        const unsigned long long bm = __ballot(active && bit);
        m &= bit ? bm : ~bm;
    }
    return m;
}

// grid (B), 1024 threads.  LDS: counts[kGroupWaves][K+1], scan[1024], (stage:) order image [HW].  Stable: within a class the pixels keep their raster order.
// Outputs: order (sorted position -> pixel), pos (pixel -> sorted position), skey (sorted position -> class, K = none), offsets (run starts).
__global__ __launch_bounds__(1024) void ifvd_group(const int *__restrict__ cls, int *__restrict__ order, int *__restrict__ offsets,
                                                    int *__restrict__ pos, int *__restrict__ skey, int HW, int K, int stage) {
    extern __shared__ int gsh[];
    const int KP = K + 1, b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    int *counts = gsh, *scan = gsh + kGroupWaves * KP;
    const int *cl = cls + (size_t)b * HW;
    for (int i = t; i < kGroupWaves * KP; i += 1024) counts[i] = 0;
    __syncthreads();
    const int nbits = 32 - __clz(K);
    const int L = (((HW + kGroupWaves - 1) / kGroupWaves + 63) / 64) * 64;
    const int start = min(w * L, HW), end = min(start + L, HW);
    int *mine = counts + w * KP;
    constexpr int NB = 8;                                              // groups of 64 pixels whose labels are requested together
    auto load_keys = [&](int base, int (&key)[NB]) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int p = base + 64 * u + lane;
            const int c = p < end ? cl[p] : -1;
            key[u] = (c >= 0 && c < K) ? c : K;
        }
    };
    for (int base = start; base < end; base += 64 * NB) {
        int key[NB];
        load_keys(base, key);
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const bool active = base + 64 * u + lane < end;
            const unsigned long long m = same_key_mask(key[u], active, nbits);
            if (active && lane == __ffsll((long long)m) - 1) mine[key[u]] += __popcll(m);   // one lane per key: no conflict inside the wave
        }
    }
    __syncthreads();
    int total = 0;
    if (t < KP)
        for (int x = 0; x < kGroupWaves; ++x) total += counts[x * KP + t];
    scan[t] = total;
    for (int d = 1; d < 1024; d <<= 1) {
        __syncthreads();
        const int v = t >= d ? scan[t - d] : 0;
        __syncthreads();
        scan[t] += v;
    }
    if (t < KP) {
        int run = scan[t] - total;                                   // exclusive prefix over the classes
        offsets[(size_t)b * KP + t] = run;                           // offsets[K] = number of pixels with a class
        for (int x = 0; x < kGroupWaves; ++x) {
            const int c = counts[x * KP + t];
            counts[x * KP + t] = run;
            run += c;
        }
    }
    __syncthreads();
    // scatter: into an LDS image of `order` when it fits (one CU writes a whole image: 4-byte stores scattered over the image cost a cache
    // line each), written out in whole rows afterwards; straight to global memory otherwise
    int *ord_l = scan + 1024;
    for (int base = start; base < end; base += 64 * NB) {
        int key[NB];
        load_keys(base, key);
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int p = base + 64 * u + lane;
            const bool active = p < end;
            const unsigned long long m = same_key_mask(key[u], active, nbits);
            if (active) {
                const int at = mine[key[u]] + __popcll(m & ((1ull << lane) - 1ull));
                if (stage) ord_l[at] = p;
                else order[(size_t)b * HW + at] = p;
                pos[(size_t)b * HW + p] = at;
            }
            if (active && lane == __ffsll((long long)m) - 1) mine[key[u]] += __popcll(m);   // (after every lane of the key has read it: program order)
        }
    }
    __syncthreads();
    // the class at every sorted position: the run that contains it (binary search over the run starts, kept in scan[] as inclusive ends)
    for (int i = t; i < HW; i += 1024) {
        if (stage) order[(size_t)b * HW + i] = ord_l[i];
        int lo = 0, hi = K;                                          // first class whose run ends beyond i; K = no class
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (scan[mid] > i) hi = mid;
            else lo = mid + 1;
        }
        skey[(size_t)b * HW + i] = lo;
    }
}

// grid (C [+ 1 when WEIGHTED], B, tensors), 256 threads.  WEIGHTED: X * wsorted (weights in SORTED order); channel index C = the plane
// of ones against bsorted (-> outB[b][k]).  Dynamic LDS: off[K+1] | bins[waves][K] | the channel plane (use_lds: it fits).  Output [B][C][K].
// A wave walks its quarter of the sorted index 64 positions at a time: gather, then a SEGMENTED inclusive scan over the lanes (equal classes
// are adjacent: "lane - d is in my run" is a comparison with the run's start) and the last lane of every run adds the run's sum to the wave's
// own bin of that class -- one lane per class and group, groups in program order: no atomics, a fixed summation order.  Nothing depends on
// the length of a run: 150 runs of 109 pixels (random labels) cost what one run of 16384 does.
template <typename T, bool WEIGHTED>
__global__ __launch_bounds__(64 * kSumWaves) void ifvd_class_sums(const T *__restrict__ X0, const T *__restrict__ X1, float *__restrict__ out0,
                                                                   float *__restrict__ out1, const float *__restrict__ wsorted,
                                                                   const float *__restrict__ bsorted, float *__restrict__ outB,
                                                                   const int *__restrict__ order, const int *__restrict__ skey,
                                                                   const int *__restrict__ offsets, int C, int HW, int K, int mean_mode,
                                                                   int use_lds) {
    constexpr int NW = kSumWaves, NT = 64 * NW;
    extern __shared__ float csh[];
    int *off_l = reinterpret_cast<int *>(csh);
    float *bins = csh + (K + 1);
    float *plane_lds = bins + NW * K + ((4 - ((NW + 1) * K + 1) % 4) % 4);     // 16-byte aligned
    const int c = blockIdx.x, b = blockIdx.y;
    const bool ones = WEIGHTED && c == C;
    const T *X = blockIdx.z ? X1 : X0;
    float *out = blockIdx.z ? out1 : out0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const T *plane = X + ((size_t)b * C + (ones ? 0 : c)) * HW;
    const int *ord = order + (size_t)b * HW, *sk = skey + (size_t)b * HW, *off = offsets + (size_t)b * (K + 1);
    const float *ws = WEIGHTED ? (ones ? bsorted : wsorted) + (size_t)b * HW : nullptr;
    for (int k = threadIdx.x; k <= K; k += NT) off_l[k] = off[k];
    for (int k = threadIdx.x; k < NW * K; k += NT) bins[k] = 0.f;
    if (use_lds && !ones) {
        constexpr int VN = VecIO<T>::N;
        if (HW % VN == 0 && (reinterpret_cast<uintptr_t>(plane) & 15) == 0) {
            for (int p = threadIdx.x * VN; p < HW; p += NT * VN) {
                float v[VN];
                VecIO<T>::load(plane + p, v);
#pragma unroll
                for (int e = 0; e < VN; ++e) plane_lds[p + e] = v[e];
            }
        } else {
            for (int p = threadIdx.x; p < HW; p += NT) plane_lds[p] = VecIO<T>::load1(plane + p);
        }
    }
    __syncthreads();
    const int nvalid = off_l[K];
    const int Q = (((nvalid + NW - 1) / NW + 63) / 64) * 64;
    const int start = min(wave * Q, nvalid), end = min(start + Q, nvalid);
    float *mybins = bins + wave * K;
    constexpr int NG = 4;                       // groups of 64 sorted positions per step; the NEXT step's index / class / weight are in flight
    struct Req { int p[NG], k[NG]; float w[NG]; };
    auto request = [&](Req &r, int base) {
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const int i = base + 64 * u + lane;
            const bool active = i < end;
            r.p[u] = active ? ord[i] : 0;
            r.k[u] = active ? sk[i] : -1;
            r.w[u] = (WEIGHTED && active) ? ws[i] : 1.f;
        }
    };
    Req cur, nxt;
    request(cur, start);
    for (int base = start; base < end; base += 64 * NG) {
        request(nxt, base + 64 * NG);
        // every LDS read of the step first, then the scans, then the bin updates: the bins share the LDS array with the plane and the run
        // table, so a read placed after an update would have to wait for it -- and the four scan chains would run one after the other
        float v[NG];
        int rs[NG], re[NG];
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const int i = base + 64 * u + lane;
            v[u] = 0.f, rs[u] = i, re[u] = i;                          // inactive lanes: a run of their own (nothing joins, nothing is added)
            if (cur.k[u] >= 0) {
                v[u] = ones ? cur.w[u] : cur.w[u] * (use_lds ? plane_lds[cur.p[u]] : VecIO<T>::load1(plane + cur.p[u]));
                rs[u] = off_l[cur.k[u]], re[u] = off_l[cur.k[u] + 1];
            }
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                const float up = __shfl_up(v[u], d, 64);
                if (lane >= d && base + 64 * u + lane - d >= rs[u]) v[u] += up;
            }
        }
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const int i = base + 64 * u + lane;
            if (cur.k[u] >= 0 && (lane == 63 || i + 1 == re[u] || i + 1 == end)) mybins[cur.k[u]] += v[u];     // the run's last lane in this group
        }
        cur = nxt;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += NT) {
        float acc = 0.f;
#pragma unroll
        for (int x = 0; x < NW; ++x) acc += bins[x * K + k];          // wave order: fixed
        if (mean_mode) acc /= (float)(off_l[k + 1] - off_l[k]) + 1e-6f;
        if (ones) outB[(size_t)b * K + k] = acc;
        else out[((size_t)b * C + c) * K + k] = acc;                   // [B][C][K]: the per-pixel passes read one channel's K values per wave
    }
}

// grid (ceil(HW/64), B), 256 threads: lane = pixel, wave = a quarter of the channels.
template <typename T>
__global__ __launch_bounds__(256) void ifvd_cos(const T *__restrict__ S, const T *__restrict__ Tt, const int *__restrict__ cls,
                                                 const int *__restrict__ pos, const float *__restrict__ mean_s,
                                                 const float *__restrict__ mean_t, float *__restrict__ coef_px, float *__restrict__ coef_sorted,
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
        const size_t at_px = (size_t)b * HW + p, at_sorted = (size_t)b * HW + pos[(size_t)b * HW + p];
        coef_px[at_px] = alpha;
        coef_px[BHW + at_px] = gamma;
        coef_sorted[at_sorted] = alpha;
        coef_sorted[BHW + at_sorted] = beta;
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
                                                 const float *__restrict__ coef_px, const float *__restrict__ A, const float *__restrict__ Bk,
                                                 const int *__restrict__ offsets, const float *__restrict__ upstream, T *__restrict__ dS, int C,
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
    const float alpha = g * coef_px[(size_t)b * HW + p], gamma = g * coef_px[BHW + (size_t)b * HW + p];
    const int *off = offsets + (size_t)b * (K + 1);
    const float invn = g / ((float)(off[k + 1] - off[k]) + 1e-6f);
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
    if (B <= 0 || C <= 0 || HW <= 0 || K <= 0 || B > 65535 || C > 65534 || K + 1 > kMaxKeys) return SD_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(X) & (dtype == SD_F32 ? 3 : 1)) != 0) return SD_E_ALIGN;
    return SD_OK;
}

template <typename T, bool WEIGHTED>
int launch_class_sums(const T *X0, const T *X1, float *out0, float *out1, const float *wsorted, const float *bsorted, float *outB,
                      const int *order, const int *skey, const int *offsets, int B, int C, int HW, int K, int mean_mode, hipStream_t st) {
    const int use_lds = HW <= kPlaneLdsMax;
    const size_t lds = ((size_t)((kSumWaves + 1) * K + 1 + 3) + (use_lds ? (size_t)HW : 0)) * sizeof(float);
    auto kern = ifvd_class_sums<T, WEIGHTED>;
    if (lds > 48 * 1024) {
        static bool raised = false;                     // per instantiation
        if (!raised) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               ((kSumWaves + 1) * kMaxKeys + 4 + kPlaneLdsMax) * (int)sizeof(float));
            if (e != hipSuccess) return (int)e;
            raised = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3(C + (WEIGHTED ? 1 : 0), B, X1 ? 2 : 1), dim3(64 * kSumWaves), lds, st, X0, X1, out0, out1, wsorted, bsorted, outB, order,
                       skey, offsets, C, HW, K, mean_mode, use_lds);
    return (int)hipGetLastError();
}

}  // namespace
}  // namespace sd

extern "C" {

size_t sd_ifvd_workspace_bytes(int B, int HW) {
    if (B <= 0 || HW <= 0) return 0;
    return (size_t)((HW + 63) / 64) * B * sizeof(double) + 16;
}

int sd_ifvd_group(const int *cls, int B, int HW, int K, int *order, int *offsets, int *pos, int *skey, void *stream) {
    if (!cls || !order || !offsets || !pos || !skey) return SD_E_NULL;
    if (B <= 0 || HW <= 0 || K <= 0 || K + 1 > sd::kMaxKeys) return SD_E_SHAPE;
    const size_t base_ints = (size_t)sd::kGroupWaves * (K + 1) + 1024;
    const int stage = (base_ints + (size_t)HW) * sizeof(int) <= 160 * 1024;
    const size_t lds = (base_ints + (stage ? (size_t)HW : 0)) * sizeof(int);
    if (lds > 48 * 1024) {
        static bool raised = false;
        if (!raised) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sd::ifvd_group), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            raised = true;
        }
    }
    hipLaunchKernelGGL(sd::ifvd_group, dim3(B), dim3(1024), lds, static_cast<hipStream_t>(stream), cls, order, offsets, pos, skey, HW, K, stage);
    return (int)hipGetLastError();
}

int sd_ifvd_class_means(const void *S, const void *T, int dtype, const int *order, const int *skey, const int *offsets, float *mean_s, float *mean_t,
                        int B, int C, int HW, int K, void *stream) {
    int rc = sd::check_ifvd(S, dtype, B, C, HW, K);
    if (rc) return rc;
    if (!order || !skey || !offsets || !mean_s || (T && !mean_t)) return SD_E_NULL;
    if (T && (reinterpret_cast<uintptr_t>(T) & (dtype == SD_F32 ? 3 : 1))) return SD_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32)
        return sd::launch_class_sums<float, false>((const float *)S, (const float *)T, mean_s, mean_t, nullptr, nullptr, nullptr, order, skey, offsets,
                                                   B, C, HW, K, 1, st);
    return sd::launch_class_sums<sd::bf16_t, false>((const sd::bf16_t *)S, (const sd::bf16_t *)T, mean_s, mean_t, nullptr, nullptr, nullptr, order,
                                                    skey, offsets, B, C, HW, K, 1, st);
}

int sd_ifvd_cos(const void *S, const void *T, int dtype, const int *cls, const int *pos, const float *mean_s, const float *mean_t, float *coef_px,
                float *coef_sorted, float *loss, void *workspace, size_t workspace_bytes, int B, int C, int HW, int K, void *stream) {
    int rc = sd::check_ifvd(S, dtype, B, C, HW, K);
    if (rc) return rc;
    if (!T || !cls || !pos || !mean_s || !mean_t || !coef_px || !coef_sorted || !loss || !workspace) return SD_E_NULL;
    if (reinterpret_cast<uintptr_t>(T) & (dtype == SD_F32 ? 3 : 1)) return SD_E_ALIGN;
    if (workspace_bytes < sd_ifvd_workspace_bytes(B, HW) - 16 || (reinterpret_cast<uintptr_t>(workspace) & 7)) return SD_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int gx = (HW + 63) / 64;
    const long BHW = (long)B * HW;
    const float w_scale = 20.f / (float)BHW;            // d/dsim of 10 * mean (sim_s - sim_t)^2
    double *sums = static_cast<double *>(workspace);
    if (dtype == SD_F32)
        hipLaunchKernelGGL((sd::ifvd_cos<float>), dim3(gx, B), dim3(256), 0, st, (const float *)S, (const float *)T, cls, pos, mean_s, mean_t, coef_px,
                           coef_sorted, sums, C, HW, K, BHW, w_scale);
    else
        hipLaunchKernelGGL((sd::ifvd_cos<sd::bf16_t>), dim3(gx, B), dim3(256), 0, st, (const sd::bf16_t *)S, (const sd::bf16_t *)T, cls, pos, mean_s,
                           mean_t, coef_px, coef_sorted, sums, C, HW, K, BHW, w_scale);
    hipLaunchKernelGGL(sd::ifvd_loss, dim3(1), dim3(256), 0, st, sums, loss, gx * B, 10.f / (float)BHW);
    return (int)hipGetLastError();
}

int sd_ifvd_coef_sums(const void *S, int dtype, const int *order, const int *skey, const int *offsets, const float *coef_sorted, float *A, float *Bk,
                      int B, int C, int HW, int K, void *stream) {
    int rc = sd::check_ifvd(S, dtype, B, C, HW, K);
    if (rc) return rc;
    if (!order || !skey || !offsets || !coef_sorted || !A || !Bk) return SD_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float *alpha = coef_sorted, *beta = coef_sorted + (size_t)B * HW;
    if (dtype == SD_F32)
        return sd::launch_class_sums<float, true>((const float *)S, nullptr, A, nullptr, alpha, beta, Bk, order, skey, offsets, B, C, HW, K, 0, st);
    return sd::launch_class_sums<sd::bf16_t, true>((const sd::bf16_t *)S, nullptr, A, nullptr, alpha, beta, Bk, order, skey, offsets, B, C, HW, K, 0,
                                                   st);
}

int sd_ifvd_bwd(const void *X, int dtype, const int *cls, const float *mean, const float *coef_px, const float *A, const float *Bk,
                const int *offsets, const float *upstream, void *dS, int B, int C, int HW, int K, void *stream) {
    int rc = sd::check_ifvd(X, dtype, B, C, HW, K);
    if (rc) return rc;
    if (!cls || !mean || !coef_px || !A || !Bk || !offsets || !dS) return SD_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int gx = (HW + 63) / 64;
    if (dtype == SD_F32)
        hipLaunchKernelGGL((sd::ifvd_bwd<float>), dim3(gx, B), dim3(256), 0, st, (const float *)X, cls, mean, coef_px, A, Bk, offsets, upstream,
                           (float *)dS, C, HW, K, (long)B * HW);
    else
        hipLaunchKernelGGL((sd::ifvd_bwd<sd::bf16_t>), dim3(gx, B), dim3(256), 0, st, (const sd::bf16_t *)X, cls, mean, coef_px, A, Bk, offsets,
                           upstream, (sd::bf16_t *)dS, C, HW, K, (long)B * HW);
    return (int)hipGetLastError();
}

}  // extern "C"
