// cgd_up.hip -- CGD criterion with the bilinear up-sampling FUSED in (regime R2), gfx950.
//
// The reference first materialises both logits at label resolution
// (losses.py:101-102: F.interpolate(..., 'bilinear', align_corners=False) of [B,C,h,w] to
// [B,C,F*h,F*w]; a 16x / 64x blow-up: 2 x 1.26 GB at the headline config) and then runs the
// softmax/KL chain over them.  Here the up-sampled values exist only in registers:
//   forward : read the two TAP tensors (2*B*C*h*w*e bytes), interpolate on the fly, fold into the
//             same online-softmax row partials as the R1 kernel (cgd_device.h);
//   backward: recompute the interpolated values, form dS = k(p_s - p_t) in registers and apply
//             the TRANSPOSED interpolation as a gather (no atomics): vertically in registers while
//             walking down the band, horizontally through one LDS row per finished tap row.
// This kernel is exp/VALU-bound, not HBM-bound: its HBM traffic is ~1/16 of R1's.
//
// Geometry.  F = H/h = W/w in {2,4,8}.  With align_corners=False the F output rows
// Y in [F*j - F/2, F*j + F/2) lie between tap rows j-1 and j ("gap j", j = 0..h) with weight
// lambda_q = (q + 0.5)/F on row j, q = Y - (F*j - F/2); rows outside [0,h) clamp, so gap 0 and gap h
// are half gaps whose outputs equal the edge row.  Same along x.  One thread owns tap column kx,
// i.e. output columns F*kx .. F*kx+F-1 (second half of x-gap kx, first half of x-gap kx+1).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "cgd_device.h"
#include "up_device.h"

namespace sd {

namespace {

__device__ __forceinline__ RowPart block_combine_dyn(RowPart st, float c2) {
    __shared__ RowPart wave_part[16];
    const float ms = wave_max(st.ms), mt = wave_max(st.mt);
    const float rs = ex2((st.ms - ms) * c2), rt = ex2((st.mt - mt) * c2);
    RowPart w;
    w.ms = ms; w.mt = mt;
    w.zs = wave_sum(st.zs * rs);
    w.zt = wave_sum(st.zt * rt);
    w.a = wave_sum(st.a * rt);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane == 0) wave_part[wid] = w;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int i = 1; i < nw; ++i) merge(w, wave_part[i], c2);
    return w;
}

struct PlaneRef {
    int b, cs, ch;
    size_t base;
};
__device__ __forceinline__ PlaneRef plane_of(int slot, int C, int hw, const int32_t *perm) {
    PlaneRef p;
    p.b = slot / C;
    p.cs = slot - p.b * C;
    p.ch = perm ? perm[p.cs] : p.cs;
    p.base = ((size_t)p.b * C + p.ch) * (size_t)hw;
    return p;
}

// ---- forward ----------------------------------------------------------------------------------
// grid.x = B*C*nband; workgroup (slot, k) folds the output rows of gaps [k*R, min(h,(k+1)*R)) of one
// plane (the last band also takes the closing half gap h) and writes part[slot*nband + k].
// DUAL (round 4, SURVEY section 7 step 7 / BASELINE config 3: CGD + channel-wise logit KL on the same taps; reference call site opts.py:100-110
// evaluates the two criteria one after the other, each re-reading and re-interpolating both taps): every interpolated pair is folded into
// TWO online-softmax states -- temperature c2 for the first criterion's rows, c2b for the second's -- and a second partial array.  The
// rows of both criteria are built from per-(slot, band) partials, so any two group sizes share the pass; `perm` is the first criterion's.
template <typename T, int F, bool DUAL>
__global__ void cgd_up_fwd_partials(const T *__restrict__ s, const T *__restrict__ t, const int32_t *__restrict__ perm,
                                    RowPart *__restrict__ part, RowPart *__restrict__ part_b, int C, int h, int w, int R, int nband, float c2,
                                    float c2b) {
    constexpr int QB = (F >= 4) ? (16 / F) : F;  // output rows folded together (16 values, or all 4 for F=2)
    const int wg = blockIdx.x;
    const int k = wg % nband;
    const PlaneRef pl = plane_of(wg / nband, C, h * w, perm);
    const T *ps = s + pl.base, *pt = t + pl.base;
    const int kx = threadIdx.x;
    const bool active = kx < w;
    const int kxc = min(kx, w - 1);
    const int j0 = k * R;
    const int j1 = (k == nband - 1) ? h + 1 : min(h, j0 + R);

    RowPart st = {kNegBig, 0.f, kNegBig, 0.f, 0.f}, st_b = st;
    float sp[F], tp[F], sc[F], tc[F];
    {
        const int r = max(j0 - 1, 0);
        hrow<T, F>(ps + (size_t)r * w, kxc, w, sp);
        hrow<T, F>(pt + (size_t)r * w, kxc, w, tp);
    }
    for (int j = j0; j < j1; ++j) {
        const int r = min(j, h - 1);
        hrow<T, F>(ps + (size_t)r * w, kxc, w, sc);
        hrow<T, F>(pt + (size_t)r * w, kxc, w, tc);
        float dsv[F], dtv[F];
#pragma unroll
        for (int rx = 0; rx < F; ++rx) { dsv[rx] = sc[rx] - sp[rx]; dtv[rx] = tc[rx] - tp[rx]; }
        if (active) {
            if (j > 0 && j < h) {
#pragma unroll
                for (int qb = 0; qb < F; qb += QB) {
                    float vs[QB * F], vt[QB * F];
#pragma unroll
                    for (int q = 0; q < QB; ++q) {
                        const float lam = (qb + q + 0.5f) / F;
#pragma unroll
                        for (int rx = 0; rx < F; ++rx) {
                            vs[q * F + rx] = fmaf(lam, dsv[rx], sp[rx]);
                            vt[q * F + rx] = fmaf(lam, dtv[rx], tp[rx]);
                        }
                    }
                    fold<QB * F>(st, vs, vt, c2);
                    if constexpr (DUAL) fold<QB * F>(st_b, vs, vt, c2b);
                }
            } else {
                // half gap at the top (j == 0: q >= F/2) or bottom (j == h: q < F/2); sp == sc there
                // (both clamp to the edge row), so every output row equals the edge row.
#pragma unroll
                for (int q = 0; q < F / 2; ++q) {
                    fold<F>(st, sc, tc, c2);
                    if constexpr (DUAL) fold<F>(st_b, sc, tc, c2b);
                }
            }
        }
#pragma unroll
        for (int rx = 0; rx < F; ++rx) { sp[rx] = sc[rx]; tp[rx] = tc[rx]; }
    }
    const RowPart wsum = block_combine_dyn(st, c2);
    if (threadIdx.x == 0) part[wg] = wsum;
    if constexpr (DUAL) {
        __syncthreads();                                    // thread 0 has read the first combine's LDS slots
        const RowPart wsum_b = block_combine_dyn(st_b, c2b);
        if (threadIdx.x == 0) part_b[wg] = wsum_b;
    }
}

// ---- backward ---------------------------------------------------------------------------------
// Workgroup (slot, k) produces tap-gradient rows [y0, y1) = [k*R, min(h,(k+1)*R)) of one plane.
// It walks gaps j = y0 .. y1: gap j contributes to tap rows j-1 (weight 1-lambda) and j (weight lambda).
// DUAL: dS = k_a (p_s - p_t)_a + k_b (p_s - p_t)_b formed in registers from one interpolation, ONE transposed-interpolation gather and
// ONE tap-gradient store instead of two passes and an add.
template <typename T, int F, bool DUAL>
__global__ void cgd_up_bwd(const T *__restrict__ s, const T *__restrict__ t, const int32_t *__restrict__ perm,
                           const float *__restrict__ row_lse2, const float *__restrict__ upstream, T *__restrict__ ds, int C, int h,
                           int w, int R, int nband, int g, int G, float c2, float coef, const float *__restrict__ row_lse2_b,
                           const float *__restrict__ upstream_b, int g_b, int G_b, float c2b, float coef_b) {
    extern __shared__ float rowbuf[];  // 2 buffers of F*blockDim.x floats
    const int wg = blockIdx.x;
    const int k = wg % nband;
    const PlaneRef pl = plane_of(wg / nband, C, h * w, perm);
    const T *ps = s + pl.base, *pt = t + pl.base;
    T *pd = ds + pl.base;
    const int row = pl.b * G + pl.cs / g;
    const float ls = row_lse2[2 * row], lt = row_lse2[2 * row + 1];
    const float kk = upstream ? coef * upstream[0] : coef;
    float ls_b = 0.f, lt_b = 0.f, kk_b = 0.f;
    if constexpr (DUAL) {
        const int row_b = pl.b * G_b + pl.cs / g_b;
        ls_b = row_lse2_b[2 * row_b];
        lt_b = row_lse2_b[2 * row_b + 1];
        kk_b = upstream_b ? coef_b * upstream_b[0] : coef_b;
    }
    const int kx = threadIdx.x;
    const bool active = kx < w;
    const int kxc = min(kx, w - 1);
    const int y0 = k * R, y1 = min(h, y0 + R);
    const int W = F * w;
    const int bufstride = F * blockDim.x;

    float sp[F], tp[F], sc[F], tc[F], accA[F], accB[F];
#pragma unroll
    for (int rx = 0; rx < F; ++rx) accA[rx] = 0.f;
    {
        const int r = max(y0 - 1, 0);
        hrow<T, F>(ps + (size_t)r * w, kxc, w, sp);
        hrow<T, F>(pt + (size_t)r * w, kxc, w, tp);
    }
    int parity = 0;
    for (int j = y0; j <= y1; ++j) {
        const int r = min(j, h - 1);
        hrow<T, F>(ps + (size_t)r * w, kxc, w, sc);
        hrow<T, F>(pt + (size_t)r * w, kxc, w, tc);
        float dsv[F], dtv[F];
#pragma unroll
        for (int rx = 0; rx < F; ++rx) { dsv[rx] = sc[rx] - sp[rx]; dtv[rx] = tc[rx] - tp[rx]; accB[rx] = 0.f; }
        const bool top = (j == 0), bot = (j == h);
#pragma unroll
        for (int q = 0; q < F; ++q) {
            if ((top && q < F / 2) || (bot && q >= F / 2)) continue;  // rows outside the image
            const float lam = (q + 0.5f) / F;
            const float wa = top ? 0.f : (bot ? 1.f : 1.f - lam);
            const float wb = top ? 1.f : (bot ? 0.f : lam);
#pragma unroll
            for (int rx = 0; rx < F; ++rx) {
                const float S = fmaf(lam, dsv[rx], sp[rx]);
                const float Tv = fmaf(lam, dtv[rx], tp[rx]);
                float D = kk * (ex2(fmaf(S, c2, -ls)) - ex2(fmaf(Tv, c2, -lt)));
                if constexpr (DUAL) D = fmaf(kk_b, ex2(fmaf(S, c2b, -ls_b)) - ex2(fmaf(Tv, c2b, -lt_b)), D);
                accA[rx] = fmaf(wa, D, accA[rx]);
                accB[rx] = fmaf(wb, D, accB[rx]);
            }
        }
        if (j > y0) {  // tap row j-1 has now received both of its gaps: transpose along x through LDS
            float *buf = rowbuf + parity * bufstride;
#pragma unroll
            for (int rx = 0; rx < F; ++rx) buf[F * kx + rx] = accA[rx];
            __syncthreads();
            if (active) {
                float sum = 0.f;
#pragma unroll
                for (int q = 0; q < F; ++q) {
                    const float lam = (q + 0.5f) / F;
                    const int xl = F * kx - F / 2 + q;      // x-gap kx   : this column is the right tap
                    const int xr = F * kx + F / 2 + q;      // x-gap kx+1 : this column is the left tap
                    if (xl >= 0) sum = fmaf(kx == 0 ? 1.f : lam, buf[xl], sum);
                    if (xr < W) sum = fmaf(kx == w - 1 ? 1.f : 1.f - lam, buf[xr], sum);
                }
                VecIO<T>::store1(pd + (size_t)(j - 1) * w + kx, sum);
            }
            parity ^= 1;
        }
#pragma unroll
        for (int rx = 0; rx < F; ++rx) { accA[rx] = accB[rx]; sp[rx] = sc[rx]; tp[rx] = tc[rx]; }
    }
}

int g_band_rows = 16;  // tunable "cgd_up_band_rows": tap rows per workgroup

int factor_of(int h, int w, int H, int W) {
    if (h <= 0 || w <= 0 || H % h || W % w) return 0;
    const int f = H / h;
    if (W / w != f) return 0;
    if (f != 2 && f != 4 && f != 8) return 0;
    if (w > 1024 || (long)f * ((w + 63) / 64 * 64) > 8192) return 0;  // one workgroup spans the tap width; 2 LDS rows <= 64 KB
    return f;
}

struct UpGeo {
    int F, R, nband, threads;
};
UpGeo up_geometry(int h, int w, int H, int W) {
    UpGeo q;
    q.F = factor_of(h, w, H, W);
    q.R = g_band_rows < h ? g_band_rows : h;
    q.nband = (h + q.R - 1) / q.R;
    q.threads = (w + 63) / 64 * 64;
    return q;
}

int check_up(const void *s, const void *t, int dtype, int B, int C, int h, int w, int H, int W, int g) {
    if (!s || !t) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (B <= 0 || C <= 0 || h <= 0 || w <= 0 || g <= 0) return SD_E_SHAPE;
    if (!factor_of(h, w, H, W)) return SD_E_UNSUPPORTED;
    const size_t es = dtype == SD_F32 ? 4 : 2;
    if ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(t)) & (es - 1)) return SD_E_ALIGN;
    return SD_OK;
}

struct UpSecond {                 // the second criterion of a DUAL pass (part == nullptr / row_lse2 == nullptr: single criterion)
    RowPart *part;
    const float *row_lse2, *upstream;
    int g;
    float c2, coef;
};

template <typename T, int F>
void launch_fwd(const void *s, const void *t, const int32_t *perm, RowPart *part, int B, int C, int h, int w, const UpGeo &q, float c2,
                const UpSecond &b2, hipStream_t st) {
    const dim3 grid((unsigned)((long)B * C * q.nband)), blk(q.threads);
    if (b2.part)
        hipLaunchKernelGGL((cgd_up_fwd_partials<T, F, true>), grid, blk, 0, st, (const T *)s, (const T *)t, perm, part, b2.part, C, h, w, q.R,
                           q.nband, c2, b2.c2);
    else
        hipLaunchKernelGGL((cgd_up_fwd_partials<T, F, false>), grid, blk, 0, st, (const T *)s, (const T *)t, perm, part, (RowPart *)nullptr, C, h,
                           w, q.R, q.nband, c2, 0.f);
}
template <typename T, int F>
void launch_bwd(const void *s, const void *t, const int32_t *perm, const float *row_lse2, const float *upstream, void *ds, int B, int C,
                int h, int w, int g, const UpGeo &q, float c2, float coef, const UpSecond &b2, hipStream_t st) {
    const size_t lds = 2ull * F * q.threads * sizeof(float);
    const dim3 grid((unsigned)((long)B * C * q.nband)), blk(q.threads);
    if (b2.row_lse2)
        hipLaunchKernelGGL((cgd_up_bwd<T, F, true>), grid, blk, lds, st, (const T *)s, (const T *)t, perm, row_lse2, upstream, (T *)ds, C, h, w,
                           q.R, q.nband, g, (C + g - 1) / g, c2, coef, b2.row_lse2, b2.upstream, b2.g, (C + b2.g - 1) / b2.g, b2.c2, b2.coef);
    else
        hipLaunchKernelGGL((cgd_up_bwd<T, F, false>), grid, blk, lds, st, (const T *)s, (const T *)t, perm, row_lse2, upstream, (T *)ds, C, h, w,
                           q.R, q.nband, g, (C + g - 1) / g, c2, coef, (const float *)nullptr, (const float *)nullptr, 1, 1, 0.f, 0.f);
}

template <typename T>
void dispatch_fwd(int F, const void *s, const void *t, const int32_t *perm, RowPart *part, int B, int C, int h, int w, const UpGeo &q,
                  float c2, const UpSecond &b2, hipStream_t st) {
    if (F == 2) launch_fwd<T, 2>(s, t, perm, part, B, C, h, w, q, c2, b2, st);
    else if (F == 4) launch_fwd<T, 4>(s, t, perm, part, B, C, h, w, q, c2, b2, st);
    else launch_fwd<T, 8>(s, t, perm, part, B, C, h, w, q, c2, b2, st);
}
template <typename T>
void dispatch_bwd(int F, const void *s, const void *t, const int32_t *perm, const float *row_lse2, const float *upstream, void *ds, int B,
                  int C, int h, int w, int g, const UpGeo &q, float c2, float coef, const UpSecond &b2, hipStream_t st) {
    if (F == 2) launch_bwd<T, 2>(s, t, perm, row_lse2, upstream, ds, B, C, h, w, g, q, c2, coef, b2, st);
    else if (F == 4) launch_bwd<T, 4>(s, t, perm, row_lse2, upstream, ds, B, C, h, w, g, q, c2, coef, b2, st);
    else launch_bwd<T, 8>(s, t, perm, row_lse2, upstream, ds, B, C, h, w, g, q, c2, coef, b2, st);
}

}  // namespace

int cgd_up_tunable(const char *key, int set, int v) {
    if (strcmp(key, "cgd_up_band_rows")) return SD_E_UNSUPPORTED;
    if (!set) return g_band_rows;
    if (v < 1 || v > 4096) return SD_E_SHAPE;
    g_band_rows = v;
    return SD_OK;
}

}  // namespace sd

extern "C" {

int sd_cgd_kl_up_supported(int h, int w, int H, int W) { return sd::factor_of(h, w, H, W) ? 1 : 0; }

size_t sd_cgd_kl_up_workspace_bytes(int B, int C, int h, int w, int H, int W, int g) {
    if (B <= 0 || C <= 0 || g <= 0 || !sd::factor_of(h, w, H, W)) return 0;
    const sd::UpGeo q = sd::up_geometry(h, w, H, W);
    return (size_t)B * C * q.nband * sizeof(sd::RowPart) + 16;
}

int sd_cgd_kl_up_fwd(const void *s, const void *t, int dtype, int B, int C, int h, int w, int H, int W, int g, float inv_tau,
                     float loss_scale, const int32_t *perm, float *row_lse2, float *row_kl, float *loss, void *workspace,
                     size_t workspace_bytes, void *stream) {
    int rc = sd::check_up(s, t, dtype, B, C, h, w, H, W, g);
    if (rc) return rc;
    if (!row_lse2 || !row_kl || !loss || !workspace) return SD_E_NULL;
    const sd::UpGeo q = sd::up_geometry(h, w, H, W);
    const long nwg = (long)B * C * q.nband;
    if (nwg > 0x7fffffffL) return SD_E_SHAPE;
    if (workspace_bytes < (size_t)nwg * sizeof(sd::RowPart) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return SD_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float c2 = inv_tau * 1.44269504088896340736f;
    sd::RowPart *part = static_cast<sd::RowPart *>(workspace);
    const sd::UpSecond none = {nullptr, nullptr, nullptr, 1, 0.f, 0.f};
    if (dtype == SD_F32) sd::dispatch_fwd<float>(q.F, s, t, perm, part, B, C, h, w, q, c2, none, st);
    else sd::dispatch_fwd<sd::bf16_t>(q.F, s, t, perm, part, B, C, h, w, q, c2, none, st);
    sd::launch_row_finalize(part, row_lse2, row_kl, loss, B, C, g, q.nband, c2, inv_tau, loss_scale, st);
    return (int)hipGetLastError();
}

int sd_cgd_kl_up_fwd2(const void *s, const void *t, int dtype, int B, int C, int h, int w, int H, int W, const int32_t *perm, int g_a,
                      float inv_tau_a, float loss_scale_a, float *row_lse2_a, float *row_kl_a, float *loss_a, int g_b, float inv_tau_b,
                      float loss_scale_b, float *row_lse2_b, float *row_kl_b, float *loss_b, void *workspace, size_t workspace_bytes,
                      void *stream) {
    int rc = sd::check_up(s, t, dtype, B, C, h, w, H, W, g_a);
    if (rc) return rc;
    if (g_b <= 0) return SD_E_SHAPE;
    if (!row_lse2_a || !row_kl_a || !loss_a || !row_lse2_b || !row_kl_b || !loss_b || !workspace) return SD_E_NULL;
    const sd::UpGeo q = sd::up_geometry(h, w, H, W);
    const long nwg = (long)B * C * q.nband;
    if (nwg > 0x7fffffffL) return SD_E_SHAPE;
    if (workspace_bytes < 2 * (size_t)nwg * sizeof(sd::RowPart) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return SD_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float k = 1.44269504088896340736f;
    sd::RowPart *part = static_cast<sd::RowPart *>(workspace);
    const sd::UpSecond b2 = {part + nwg, nullptr, nullptr, g_b, inv_tau_b * k, 0.f};
    if (dtype == SD_F32) sd::dispatch_fwd<float>(q.F, s, t, perm, part, B, C, h, w, q, inv_tau_a * k, b2, st);
    else sd::dispatch_fwd<sd::bf16_t>(q.F, s, t, perm, part, B, C, h, w, q, inv_tau_a * k, b2, st);
    sd::launch_row_finalize(part, row_lse2_a, row_kl_a, loss_a, B, C, g_a, q.nband, inv_tau_a * k, inv_tau_a, loss_scale_a, st);
    sd::launch_row_finalize(part + nwg, row_lse2_b, row_kl_b, loss_b, B, C, g_b, q.nband, inv_tau_b * k, inv_tau_b, loss_scale_b, st);
    return (int)hipGetLastError();
}

int sd_cgd_kl_up_bwd(const void *s, const void *t, int dtype, int B, int C, int h, int w, int H, int W, int g, float inv_tau, float coef,
                     const int32_t *perm, const float *row_lse2, const float *upstream, void *ds, void *stream) {
    int rc = sd::check_up(s, t, dtype, B, C, h, w, H, W, g);
    if (rc) return rc;
    if (!row_lse2 || !ds) return SD_E_NULL;
    const sd::UpGeo q = sd::up_geometry(h, w, H, W);
    if ((long)B * C * q.nband > 0x7fffffffL) return SD_E_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float c2 = inv_tau * 1.44269504088896340736f;
    const sd::UpSecond none = {nullptr, nullptr, nullptr, 1, 0.f, 0.f};
    if (dtype == SD_F32) sd::dispatch_bwd<float>(q.F, s, t, perm, row_lse2, upstream, ds, B, C, h, w, g, q, c2, coef, none, st);
    else sd::dispatch_bwd<sd::bf16_t>(q.F, s, t, perm, row_lse2, upstream, ds, B, C, h, w, g, q, c2, coef, none, st);
    return (int)hipGetLastError();
}

int sd_cgd_kl_up_bwd2(const void *s, const void *t, int dtype, int B, int C, int h, int w, int H, int W, const int32_t *perm, int g_a,
                      float inv_tau_a, float coef_a, const float *row_lse2_a, const float *upstream_a, int g_b, float inv_tau_b, float coef_b,
                      const float *row_lse2_b, const float *upstream_b, void *ds, void *stream) {
    int rc = sd::check_up(s, t, dtype, B, C, h, w, H, W, g_a);
    if (rc) return rc;
    if (g_b <= 0) return SD_E_SHAPE;
    if (!row_lse2_a || !row_lse2_b || !ds) return SD_E_NULL;
    const sd::UpGeo q = sd::up_geometry(h, w, H, W);
    if ((long)B * C * q.nband > 0x7fffffffL) return SD_E_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float k = 1.44269504088896340736f;
    const sd::UpSecond b2 = {nullptr, row_lse2_b, upstream_b, g_b, inv_tau_b * k, coef_b};
    if (dtype == SD_F32) sd::dispatch_bwd<float>(q.F, s, t, perm, row_lse2_a, upstream_a, ds, B, C, h, w, g_a, q, inv_tau_a * k, coef_a, b2, st);
    else sd::dispatch_bwd<sd::bf16_t>(q.F, s, t, perm, row_lse2_a, upstream_a, ds, B, C, h, w, g_a, q, inv_tau_a * k, coef_a, b2, st);
    return (int)hipGetLastError();
}

}  // extern "C"
