// cgd_tok.hip -- the CGD / CD criterion on TOKEN-MAJOR operands [B][P][C] (C contiguous), gfx950.
//
// The decoder features of a SegFormer head (decode_head.linear_c1..4, BASELINE config 5 / SURVEY a-16) leave their Linear as tokens
// [B, h*w, E].  The reference would first have to view them as [B, E, h, w] (the helper it keeps commented out, losses.py:300-318); on the
// GPU that view costs a transpose copy of the student tap, of the 3x wider teacher tap and of the gradient on every stage.  Here the
// criterion of losses.py:105-112 reads the tokens as they are:
//   row (b, j) = channel slots j*g .. j*g+g-1 (slot c' = channel perm[c'], virtual -1e9 pad beyond C) x all P pixels -- the same rows,
//   the same closed form and the same per-(channel slot, pixel chunk) partials as cgd_kl.hip, so its fp64 row finalisation is reused.
// Mapping: a lane owns ONE 16-byte channel vector position (4 fp32 / 8 bf16 consecutive channels) and walks down the pixels of its
// workgroup's chunk, U pixels per step, keeping one online-softmax state PER CHANNEL (a row mixes channels, and under a shuffle the N
// channels of a vector belong to N different rows); consecutive lanes hold consecutive vectors of a pixel, so every load instruction
// reads whole contiguous pixel rows.  One rescale per U elements and channel: (2 + 2U)/U exponentials per element.
// HBM-bound like R1: forward 2*N*e bytes, backward 3*N*e.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cgd_device.h"

namespace sd {
namespace {

constexpr int kTokU = 4;          // pixels per lane and step, fp32 (independent 16-byte loads per operand in flight); bf16: 2 -- 8 channel states per
                                  // lane leave no room for more without dropping below 4 waves per SIMD
template <typename T> constexpr int tok_u() { return VecIO<T>::N == 8 ? 2 : kTokU; }
constexpr int kMaxPermC = 2048;   // inverse-permutation table in LDS

struct TokGeo {
    int N, vpp, VS, r, nvb, pix_chunk, nchunk, threads;
};

template <typename T>
TokGeo tok_geometry(int C, long P) {
    TokGeo q;
    q.N = VecIO<T>::N;
    q.vpp = C / q.N;                                   // vectors per pixel
    q.VS = q.vpp < 256 ? q.vpp : 256;                  // vector positions per workgroup
    q.nvb = (q.vpp + q.VS - 1) / q.VS;
    q.r = 256 / q.VS < 1 ? 1 : 256 / q.VS;             // pixel lanes per vector position
    q.threads = (q.VS * q.r + 63) / 64 * 64;
    // ~64 pixels per lane and chunk, at least 4 chunks per image when the image is large enough: >= 1024 waves at the config-5 shapes
    const int U = tok_u<T>();
    long chunk = (long)q.r * U * (64 / U);
    while (chunk > (long)q.r * U && (P + chunk - 1) / chunk < 4) chunk /= 2;
    q.pix_chunk = (int)chunk;
    q.nchunk = (int)((P + chunk - 1) / chunk);
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

// grid: (nchunk * nvb, B); part[(b*C + slot)*nchunk + k]
template <typename T>
__global__ __launch_bounds__(256) void cgd_tok_fwd_partials(const T *__restrict__ S, const T *__restrict__ Tt, const int32_t *__restrict__ perm,
                                                             RowPart *__restrict__ part, int C, long P, int VS, int r, int nvb, int pix_chunk,
                                                             int nchunk, float c2) {
    constexpr int N = VecIO<T>::N, U = tok_u<T>();
    extern __shared__ __attribute__((aligned(16))) unsigned char tok_smem[];   // RowPart red[(r-1) * VS * N], then int inv[C] when perm
    RowPart *red = reinterpret_cast<RowPart *>(tok_smem);
    int *inv = reinterpret_cast<int *>(tok_smem + (size_t)(r - 1) * VS * N * sizeof(RowPart));
    const int b = blockIdx.y;
    const int k = blockIdx.x / nvb, vb = blockIdx.x - k * nvb;
    const int t = threadIdx.x;
    const int vl = t % VS, pr = t / VS;
    const int v = vb * VS + vl;
    const bool lane_ok = pr < r && v * N < C;
    if (perm) {
        for (int i = t; i < C; i += blockDim.x) inv[perm[i]] = i;
    }
    const long p_lo = (long)k * pix_chunk, p_hi = min(P, p_lo + pix_chunk);
    const T *ps = S + ((size_t)b * P) * C + (size_t)(lane_ok ? v : 0) * N;
    const T *pt = Tt + ((size_t)b * P) * C + (size_t)(lane_ok ? v : 0) * N;
    RowPart st[N];
#pragma unroll
    for (int i = 0; i < N; ++i) st[i] = {kNegBig, 0.f, kNegBig, 0.f, 0.f};
    if (lane_ok) {
        // software-pipelined: the vectors of step i+1 are requested before step i is folded (addresses clamped into the chunk, values masked)
        typedef typename RawIO<T>::raw_t raw_t;
        raw_t s[U], tt[U], sn[U], tn[U];
        auto request = [&](long p0, raw_t (&a)[U], raw_t (&bq)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long p = p0 + (long)u * r;
                const long pc = p < p_hi ? p : p_hi - 1;
                a[u] = RawIO<T>::load(ps + (size_t)pc * C);
                bq[u] = RawIO<T>::load(pt + (size_t)pc * C);
            }
        };
        request(p_lo + pr, s, tt);
        for (long p0 = p_lo + pr; p0 < p_hi; p0 += (long)r * U) {
            request(p0 + (long)r * U, sn, tn);
            bool in[U];
#pragma unroll
            for (int u = 0; u < U; ++u) in[u] = p0 + (long)u * r < p_hi;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                float sv[U], tv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {                    // a masked element contributes exp(-big) = 0 and (t - s) = 0
                    sv[u] = in[u] ? RawIO<T>::elem(s[u], i) : kNegBig;
                    tv[u] = in[u] ? RawIO<T>::elem(tt[u], i) : kNegBig;
                }
                fold_channel<U>(st[i], sv, tv, c2);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { s[u] = sn[u]; tt[u] = tn[u]; }
        }
    }
    // combine the r pixel lanes of a vector position (deterministic order), lane pr == 0 writes the N channel partials
    if (r > 1) {
        if (lane_ok && pr > 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) red[((pr - 1) * VS + vl) * N + i] = st[i];
        }
        __syncthreads();
    } else if (perm) {
        __syncthreads();                                          // inv[] complete
    }
    if (lane_ok && pr == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            RowPart w = st[i];
            for (int q = 1; q < r; ++q) merge(w, red[((q - 1) * VS + vl) * N + i], c2);
            const int c = v * N + i;
            const int slot = perm ? inv[c] : c;
            part[((size_t)b * C + slot) * nchunk + k] = w;
        }
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
    const TokGeo q = tok_geometry<T>(C, P);
    if (ws_bytes < (size_t)B * C * q.nchunk * sizeof(RowPart) || (reinterpret_cast<uintptr_t>(ws) & 15)) return SD_E_WORKSPACE;
    const float c2 = inv_tau * 1.44269504088896340736f;
    RowPart *part = static_cast<RowPart *>(ws);
    const size_t lds = (size_t)(q.r - 1) * q.VS * q.N * sizeof(RowPart) + (perm ? (size_t)C * sizeof(int) : 0);
    hipLaunchKernelGGL((cgd_tok_fwd_partials<T>), dim3((unsigned)(q.nchunk * q.nvb), B), dim3(q.threads), lds, st, (const T *)S, (const T *)Tt, perm,
                       part, C, P, q.VS, q.r, q.nvb, q.pix_chunk, q.nchunk, c2);
    launch_row_finalize(part, row_lse2, row_kl, loss, B, C, g, q.nchunk, c2, inv_tau, loss_scale, st);
    return (int)hipGetLastError();
}

template <typename T>
int tok_bwd(const void *S, const void *Tt, int B, int C, long P, int g, float inv_tau, float coef, const int32_t *perm, const float *row_lse2,
            const float *upstream, void *dS, hipStream_t st) {
    const TokGeo q = tok_geometry<T>(C, P);
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
    const sd::TokGeo a = sd::tok_geometry<float>(C, P), b = sd::tok_geometry<sd::bf16_t>(C, P);
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
