// cgd_tok.hip -- the CGD / CD criterion on TOKEN-MAJOR operands [B][P][C] (C contiguous), gfx950.
//
// The decoder features of a SegFormer head (decode_head.linear_c1..4, BASELINE config 5 / SURVEY a-16) leave their Linear as tokens
// [B, h*w, E].  The reference would first have to view them as [B, E, h, w] (the helper it keeps commented out, losses.py:300-318); on the
// GPU that view costs a transpose copy of the student tap, of the 3x wider teacher tap and of the gradient on every stage.  Here the
// criterion of losses.py:105-112 reads the tokens as they are:
//   row (b, j) = channel slots j*g .. j*g+g-1 (slot c' = channel perm[c'], virtual -1e9 pad beyond C) x all P pixels -- the same rows
//   and the same closed form as cgd_kl.hip, whose fp64 row kernel finishes the job.
// Forward: a lane owns ONE 16-byte channel vector position (4 fp32 / 8 bf16 consecutive channels) and ONE chunk of 16..64 pixels, walks
// down it U pixels per step keeping one online-softmax state PER CHANNEL (a row mixes channels, and under a shuffle the N channels of a
// vector belong to N different rows), and writes its N states as one contiguous 20 N-byte run of the CHANNEL-indexed, chunk-major
// partial array [b][chunk][channel] (the row kernel applies the shuffle).  Consecutive lanes hold consecutive vectors of a pixel, so
// loads and partial stores are whole contiguous rows.  cgd_tok_merge_chunks then folds each channel's chunks to at most kTokKeep.
// (Round-2 history, measured at config 5: the first version merged the r lanes of a vector position inside the workgroup -- 7 serial
// merges x N channels on 1/r of the lanes -- and let one wave per ROW fold all chunks: 117/45/34/34 us for the four stages; slot-major
// element stores of the partials cost more line writes than the operands cost line reads.  Now 83/29/14/8.5 us + ~5 us merge.)
// Backward: elementwise, one step of U pixels per lane (stage 2-4 launches were 1 wave per SIMD with 64-pixel lanes: 65/53/48 us -> 26/8/5).
// One rescale per U elements and channel: (2 + 2U)/U exponentials per element; the forward is VALU-paced (~17 instructions per element
// pair), ~76 % of the measured HBM ceiling at stage 1.  Algorithmic bytes: forward 2*N*e, backward 3*N*e.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cgd_device.h"

namespace sd {
namespace {

constexpr int kTokU = 4;          // forward: pixels per lane and step (independent 16-byte loads per operand in flight; one rescale per U elements)
template <typename T> constexpr int tok_u() { return VecIO<T>::N == 8 ? 2 : kTokU; }   // backward: pixels per lane (one step, no loop state)
constexpr int kMaxPermC = 2048;   // inverse-permutation table in LDS

struct TokGeo {
    int N, vpp, VS, r, nvb, pix_chunk, nchunk, threads;
};

template <typename T>
TokGeo tok_geometry(int B, int C, long P, bool fwd) {
    TokGeo q;
    q.N = VecIO<T>::N;
    q.vpp = C / q.N;                                   // vectors per pixel
    q.VS = q.vpp < 256 ? q.vpp : 256;                  // vector positions per workgroup
    q.nvb = (q.vpp + q.VS - 1) / q.VS;
    q.r = 256 / q.VS < 1 ? 1 : 256 / q.VS;             // lanes per vector position: fwd -- each owns a chunk; bwd -- pixel-interleaved in one chunk
    q.threads = (q.VS * q.r + 63) / 64 * 64;
    if (fwd) {
        // pixels per lane: 64 when that still makes >= 2048 waves, down to 16 otherwise (one chunk = one lane's run; measured per stage of
        // config 5: 64 / 16 / 16 / 16 are the fastest of 8..128)
        int L = 64;
        const long lanes = (long)B * q.vpp * P;
        while (L > 16 && lanes / (64L * L) < 2048) L /= 2;
        q.pix_chunk = L;
        q.nchunk = (int)((P + L - 1) / L);
    } else {
        // one step per lane: the backward has no per-chunk state beyond 2 N row constants (cached), and short lanes fill the small stages
        q.pix_chunk = q.r * tok_u<T>();
        q.nchunk = (int)((P + q.pix_chunk - 1) / q.pix_chunk);
    }
    return q;
}

// 16-byte vector kept RAW in registers (4 VGPRs whatever the storage type); elements are widened on use.  The forward holds 2*U of
// these per lane next to N online-softmax states: widening at load time (8 floats per bf16 vector) cost 136 VGPRs = 3 waves per SIMD.
template <typename T> struct RawIO;
template <> struct RawIO<float> {
    typedef float raw_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ raw_t load(const float *p) { return __builtin_nontemporal_load(reinterpret_cast<const raw_t *>(p)); }
    static __device__ __forceinline__ float elem(const raw_t &v, int i) { return v[i]; }
};
template <> struct RawIO<bf16_t> {
    typedef unsigned int raw_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ raw_t load(const bf16_t *p) { return __builtin_nontemporal_load(reinterpret_cast<const raw_t *>(p)); }
    static __device__ __forceinline__ float elem(const raw_t &v, int i) {
        return (i & 1) ? __uint_as_float(v[i >> 1] & 0xffff0000u) : __uint_as_float(v[i >> 1] << 16);
    }
};

// per-channel fold of U elements (cgd_device.h::fold with the element axis across pixels)
template <int U>
__device__ __forceinline__ void fold_channel(RowPart &st, const float (&s)[U], const float (&t)[U], float c2) {
    fold<U>(st, s, t, c2);
}

// grid: (ceil(nchunk / r) * nvb, B); the r lanes of a vector position each own one chunk of pix_chunk pixels.  Partials are indexed by
// CHANNEL (the row kernel applies the shuffle), chunk-major: part[(b*nchunk + k)*C + c] -- a lane's N states are 20*N contiguous bytes and
// the lanes of a pixel row write one contiguous run (slot-major element stores cost more line writes than the operands cost line reads).
template <typename T>
__global__ __launch_bounds__(256) void cgd_tok_fwd_partials(const T *__restrict__ S, const T *__restrict__ Tt, RowPart *__restrict__ part, int C,
                                                             long P, int VS, int r, int nvb, int pix_chunk, int nchunk, float c2) {
    constexpr int N = VecIO<T>::N, U = kTokU;
    const int b = blockIdx.y;
    const int kb = blockIdx.x / nvb, vb = blockIdx.x - kb * nvb;
    const int t = threadIdx.x;
    const int vl = t % VS, pr = t / VS;
    const int v = vb * VS + vl;
    const int k = kb * r + pr;
    if (!(pr < r && v * N < C && k < nchunk)) return;
    const long p_lo = (long)k * pix_chunk, p_hi = min(P, p_lo + pix_chunk);
    const T *ps = S + ((size_t)b * P) * C + (size_t)v * N;
    const T *pt = Tt + ((size_t)b * P) * C + (size_t)v * N;
    RowPart st[N];
#pragma unroll
    for (int i = 0; i < N; ++i) st[i] = {kNegBig, 0.f, kNegBig, 0.f, 0.f};
    // U raw vectors per operand in flight per lane; no cross-step prefetch (measured: 82.5 us vs 85.4 with it at stage 1 of config 5, and
    // 106 instead of 157 VGPRs) -- the other waves of the SIMD cover the latency, the loop is VALU-paced (~17 instructions per element pair)
    typedef typename RawIO<T>::raw_t raw_t;
    for (long p0 = p_lo; p0 < p_hi; p0 += U) {
        raw_t s[U], tt[U];
        bool in[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long p = p0 + u;
            in[u] = p < p_hi;
            const long pc = in[u] ? p : p_hi - 1;                // address clamped into the chunk, value masked below
            s[u] = RawIO<T>::load(ps + (size_t)pc * C);
            tt[u] = RawIO<T>::load(pt + (size_t)pc * C);
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            float sv[U], tv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {                        // a masked element contributes exp(-big) = 0 and (t - s) = 0
                sv[u] = in[u] ? RawIO<T>::elem(s[u], i) : kNegBig;
                tv[u] = in[u] ? RawIO<T>::elem(tt[u], i) : kNegBig;
            }
            fold_channel<U>(st[i], sv, tv, c2);
        }
    }
    typedef float f4 __attribute__((ext_vector_type(4)));
    float flat[5 * N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        flat[5 * i] = st[i].ms; flat[5 * i + 1] = st[i].zs; flat[5 * i + 2] = st[i].mt; flat[5 * i + 3] = st[i].zt; flat[5 * i + 4] = st[i].a;
    }
    f4 *dst = reinterpret_cast<f4 *>(part + ((size_t)b * nchunk + k) * C + (size_t)v * N);   // 20*N bytes, 16-byte aligned (N = 4 | 8)
#pragma unroll
    for (int j = 0; j < 5 * N / 4; ++j) dst[j] = f4{flat[4 * j], flat[4 * j + 1], flat[4 * j + 2], flat[4 * j + 3]};
}

// Fold the chunks k = z, z + kTokKeep, ... of every (image, channel) into the partial of chunk z, in place (grid.z = kTokKeep, each z owns a
// disjoint chunk set): the row kernel then reads at most kTokKeep partials per channel.  64 channels x 4 chunk lanes per workgroup -- a
// wave reads 64 consecutive records per chunk; sums accumulate in fp64.
constexpr int kTokKeep = 8;
__global__ __launch_bounds__(256) void cgd_tok_merge_chunks(RowPart *__restrict__ part, int C, int nchunk, float c2) {
    __shared__ double acc[3][3][64];
    __shared__ float mx[3][2][64];
    const int b = blockIdx.y, z = blockIdx.z;
    const int sl = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + sl;
    const bool ok = c < C;
    float ms = kNegBig, mt = kNegBig;
    double zs = 0, zt = 0, a = 0;
    RowPart *p = part + (size_t)b * nchunk * C + (ok ? c : 0);
    if (ok) {
        for (int k = z + kq * kTokKeep; k < nchunk; k += 4 * kTokKeep) {
            const RowPart q = p[(size_t)k * C];
            const float nms = fmaxf(ms, q.ms), nmt = fmaxf(mt, q.mt);
            const float rs = ex2((ms - nms) * c2), qs = ex2((q.ms - nms) * c2);
            const float rt = ex2((mt - nmt) * c2), qt = ex2((q.mt - nmt) * c2);
            zs = zs * (double)rs + (double)q.zs * (double)qs;
            zt = zt * (double)rt + (double)q.zt * (double)qt;
            a = a * (double)rt + (double)q.a * (double)qt;
            ms = nms; mt = nmt;
        }
    }
    if (kq > 0) {
        acc[kq - 1][0][sl] = zs; acc[kq - 1][1][sl] = zt; acc[kq - 1][2][sl] = a;
        mx[kq - 1][0][sl] = ms; mx[kq - 1][1][sl] = mt;
    }
    __syncthreads();
    if (kq == 0 && ok) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float qms = mx[j][0][sl], qmt = mx[j][1][sl];
            const float nms = fmaxf(ms, qms), nmt = fmaxf(mt, qmt);
            const float rs = ex2((ms - nms) * c2), qs = ex2((qms - nms) * c2);
            const float rt = ex2((mt - nmt) * c2), qt = ex2((qmt - nmt) * c2);
            zs = zs * (double)rs + acc[j][0][sl] * (double)qs;
            zt = zt * (double)rt + acc[j][1][sl] * (double)qt;
            a = a * (double)rt + acc[j][2][sl] * (double)qt;
            ms = nms; mt = nmt;
        }
        p[(size_t)z * C] = {ms, (float)zs, mt, (float)zt, (float)a};
    }
}

// dS = kk (2^{s c2 - lse2_s(row)} - 2^{t c2 - lse2_t(row)}), row = b*G + slot(c)/g
template <typename T, bool NT>
__global__ __launch_bounds__(256) void cgd_tok_bwd(const T *__restrict__ S, const T *__restrict__ Tt, const int32_t *__restrict__ perm,
                                                    const float *__restrict__ row_lse2, const float *__restrict__ upstream, T *__restrict__ dS,
                                                    int C, long P, int g, int G, int VS, int r, int nvb, int pix_chunk, float c2, float coef) {
    constexpr int N = VecIO<T>::N, U = tok_u<T>();
    extern __shared__ __attribute__((aligned(16))) unsigned char tok_smem[];   // int inv[C] when perm
    int *inv = reinterpret_cast<int *>(tok_smem);
    const int b = blockIdx.y;
    const int k = blockIdx.x / nvb, vb = blockIdx.x - k * nvb;
    const int t = threadIdx.x;
    const int vl = t % VS, pr = t / VS;
    const int v = vb * VS + vl;
    const bool lane_ok = pr < r && v * N < C;
    if (perm) {
        for (int i = t; i < C; i += blockDim.x) inv[perm[i]] = i;
        __syncthreads();
    }
    if (!lane_ok) return;
    const float kk = upstream ? coef * upstream[0] : coef;
    float ls[N], lt[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int c = v * N + i;
        const int row = b * G + (perm ? inv[c] : c) / g;
        ls[i] = row_lse2[2 * row];
        lt[i] = row_lse2[2 * row + 1];
    }
    const long p_lo = (long)k * pix_chunk, p_hi = min(P, p_lo + pix_chunk);
    const size_t base = ((size_t)b * P) * C + (size_t)v * N;
    for (long p0 = p_lo + pr; p0 < p_hi; p0 += (long)r * U) {
        float s[U][N], tt[U][N];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long p = p0 + (long)u * r;
            const long pc = p < p_hi ? p : p_hi - 1;
            VecIO<T>::load(S + base + (size_t)pc * C, s[u]);
            VecIO<T>::load(Tt + base + (size_t)pc * C, tt[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long p = p0 + (long)u * r;
            if (p < p_hi) {
                float d[N];
#pragma unroll
                for (int i = 0; i < N; ++i) d[i] = kk * (ex2(fmaf(s[u][i], c2, -ls[i])) - ex2(fmaf(tt[u][i], c2, -lt[i])));
                VecIO<T>::template store<NT>(dS + base + (size_t)p * C, d);
            }
        }
    }
}

int check_tok(const void *S, const void *Tt, int dtype, int B, int C, long P, int g, const int32_t *perm) {
    if (!S || !Tt) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (B <= 0 || C <= 0 || P <= 0 || g <= 0 || B > 65535) return SD_E_SHAPE;
    const int N = dtype == SD_F32 ? 4 : 8;
    if (C % N) return SD_E_UNSUPPORTED;                              // a pixel's channels must be whole 16-byte vectors
    if ((reinterpret_cast<uintptr_t>(S) | reinterpret_cast<uintptr_t>(Tt)) & 15) return SD_E_ALIGN;
    if (perm && C > kMaxPermC) return SD_E_UNSUPPORTED;
    return SD_OK;
}

template <typename T>
int tok_fwd(const void *S, const void *Tt, int B, int C, long P, int g, float inv_tau, float loss_scale, const int32_t *perm, float *row_lse2,
            float *row_kl, float *loss, void *ws, size_t ws_bytes, hipStream_t st) {
    const TokGeo q = tok_geometry<T>(B, C, P, true);
    if (ws_bytes < (size_t)B * C * q.nchunk * sizeof(RowPart) || (reinterpret_cast<uintptr_t>(ws) & 15)) return SD_E_WORKSPACE;
    const float c2 = inv_tau * 1.44269504088896340736f;
    RowPart *part = static_cast<RowPart *>(ws);
    hipLaunchKernelGGL((cgd_tok_fwd_partials<T>), dim3((unsigned)((q.nchunk + q.r - 1) / q.r * q.nvb), B), dim3(q.threads), 0, st, (const T *)S,
                       (const T *)Tt, part, C, P, q.VS, q.r, q.nvb, q.pix_chunk, q.nchunk, c2);
    if (q.nchunk > kTokKeep)
        hipLaunchKernelGGL(cgd_tok_merge_chunks, dim3((C + 63) / 64, B, kTokKeep), dim3(256), 0, st, part, C, q.nchunk, c2);
    launch_row_finalize_strided(part, row_lse2, row_kl, loss, B, C, g, q.nchunk > kTokKeep ? kTokKeep : q.nchunk, (long)q.nchunk * C, 1, C, perm, c2,
                                inv_tau, loss_scale, st);
    return (int)hipGetLastError();
}

template <typename T>
int tok_bwd(const void *S, const void *Tt, int B, int C, long P, int g, float inv_tau, float coef, const int32_t *perm, const float *row_lse2,
            const float *upstream, void *dS, hipStream_t st) {
    const TokGeo q = tok_geometry<T>(B, C, P, false);
    const float c2 = inv_tau * 1.44269504088896340736f;
    const int G = (C + g - 1) / g;
    hipLaunchKernelGGL((cgd_tok_bwd<T, true>), dim3((unsigned)(q.nchunk * q.nvb), B), dim3(q.threads), perm ? (size_t)C * sizeof(int) : 0, st, (const T *)S, (const T *)Tt, perm,
                       row_lse2, upstream, (T *)dS, C, P, g, G, q.VS, q.r, q.nvb, q.pix_chunk, c2, coef);
    return (int)hipGetLastError();
}

}  // namespace
}  // namespace sd

extern "C" {

size_t sd_cgd_kl_tok_workspace_bytes(int B, int C, long P) {
    if (B <= 0 || C <= 0 || P <= 0) return 0;
    // the fp32 geometry has the most vector positions per pixel, hence never fewer chunks than the bf16 one
    const sd::TokGeo a = sd::tok_geometry<float>(B, C, P, true), b = sd::tok_geometry<sd::bf16_t>(B, C, P, true);
    const int nchunk = a.nchunk > b.nchunk ? a.nchunk : b.nchunk;
    return (size_t)B * C * nchunk * sizeof(sd::RowPart) + 16;
}

int sd_cgd_kl_tok_fwd(const void *S, const void *T, int dtype, int B, int C, long P, int g, float inv_tau, float loss_scale, const int32_t *perm,
                      float *row_lse2, float *row_kl, float *loss, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = sd::check_tok(S, T, dtype, B, C, P, g, perm);
    if (rc) return rc;
    if (!row_lse2 || !row_kl || !loss || !workspace) return SD_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32) return sd::tok_fwd<float>(S, T, B, C, P, g, inv_tau, loss_scale, perm, row_lse2, row_kl, loss, workspace, workspace_bytes, st);
    return sd::tok_fwd<sd::bf16_t>(S, T, B, C, P, g, inv_tau, loss_scale, perm, row_lse2, row_kl, loss, workspace, workspace_bytes, st);
}

int sd_cgd_kl_tok_bwd(const void *S, const void *T, int dtype, int B, int C, long P, int g, float inv_tau, float coef, const int32_t *perm,
                      const float *row_lse2, const float *upstream, void *dS, void *stream) {
    int rc = sd::check_tok(S, T, dtype, B, C, P, g, perm);
    if (rc) return rc;
    if (!row_lse2 || !dS) return SD_E_NULL;
    if (reinterpret_cast<uintptr_t>(dS) & 15) return SD_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32) return sd::tok_bwd<float>(S, T, B, C, P, g, inv_tau, coef, perm, row_lse2, upstream, dS, st);
    return sd::tok_bwd<sd::bf16_t>(S, T, B, C, P, g, inv_tau, coef, perm, row_lse2, upstream, dS, st);
}

}  // extern "C"
