// ce_up.hip -- supervised segmentation loss of the student with the bilinear up-sampling of the
// logits fused in, gfx950.
//
// Reference chain (mmseg/models/decode_heads/decode_head.py:217-237 -> losses/cross_entropy_loss.py:9-32,
// losses/accuracy.py:4-49): resize the [B,C,h,w] logits to label size (a [B,150,512,512] = 1.26 GB
// tensor at the headline config), log_softmax over C, NLL with ignore_index, reduction, plus a top-1
// argmax for the accuracy log variable; the backward re-reads / re-writes the same 1.26 GB three
// more times.  The round-1 rocprof of the KD step attributes ~8 ms/step to that chain.
//
//   forward : per output pixel an online softmax over the C channels of on-the-fly interpolated
//             logits; writes the per-pixel loss, its base-2 log-partition (for the backward) and counts
//             top-1 hits.  Nothing of size B*C*H*W is written.
//   backward: structured like cgd_up_bwd (one workgroup per channel plane and band of tap rows):
//             dlogit = g * (softmax - onehot) in registers, transposed interpolation vertically in
//             registers and horizontally through an LDS row; the per-pixel maps (lse, label, upstream)
//             are re-read per channel but stay L2-resident because consecutive workgroups handle the
//             SAME pixels for consecutive channels.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "cgd_device.h"
#include "up_device.h"

namespace sd {

namespace {

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;

// One element load the compiler can neither sink nor reorder (asm volatile), returned as raw bits; wait_pinned_loads() before the first use.
template <typename T> __device__ __forceinline__ unsigned pinned_load1(const T *p);
template <> __device__ __forceinline__ unsigned pinned_load1<float>(const float *p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <> __device__ __forceinline__ unsigned pinned_load1<bf16_t>(const bf16_t *p) {
    unsigned v;
    asm volatile("global_load_ushort %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// "memory" orders memory operations only: the register copies / arithmetic that consume the loaded registers could still be scheduled above
// the wait, so a scheduling barrier follows it (cdna_hip_programming.md section 5.4 rule 18).
__device__ __forceinline__ void wait_pinned_loads() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <typename T> __device__ __forceinline__ unsigned plain_load1(const T *p);          // the same raw bits through a load hipcc counts
template <> __device__ __forceinline__ unsigned plain_load1<float>(const float *p) { return __float_as_uint(*p); }
template <> __device__ __forceinline__ unsigned plain_load1<bf16_t>(const bf16_t *p) { return p->bits; }
template <typename T> __device__ __forceinline__ float raw_to_float(unsigned v);
template <> __device__ __forceinline__ float raw_to_float<float>(unsigned v) { return __uint_as_float(v); }
template <> __device__ __forceinline__ float raw_to_float<bf16_t>(unsigned v) { return __uint_as_float(v << 16); }

// Forward: a thread owns ONE output column (its two horizontal taps and weight are fixed) and the F output rows of G consecutive gaps
// (G+1 tap rows: the shared row is loaded once).  The class loop is innermost, in CHUNKS of 8 classes: the chunk's values are formed in
// registers, ONE max and ONE rescale exponential are spent per chunk and pixel (as cgd_device.h::fold does per 16 elements), then one
// fma + exponential + add per class -- 7 vector issues + 1 transcendental per pixel and class instead of the 13 + 1 of a per-class
// online update with the label pick and the arg-max compare inside the loop.  Those two leave the loop altogether:
//   * the label's logit is ONE interpolation per pixel of the label's own class plane (4 gathered taps), formed by the very same
//     fmaf sequence as in the loop, hence bit-equal to the value the loop saw;
//   * top-1: the loop only remembers in which chunk the running maximum last rose (strictly); a pixel whose label logit EQUALS the
//     maximum (the only candidates for a hit) re-forms that one chunk and takes the first class that attains the maximum -- exactly
//     the first-index rule of an arg-max, ties included.
// Arithmetic as up_device.h::hrow, operation for operation: fmaf(lambda, right - left, left).
// grid: (ceil(W / blockDim), ceil((h + 1) / G), B)
template <typename T, int F, int G>
__global__ __launch_bounds__(256) void ce_up_fwd_col(const T *__restrict__ s, const int32_t *__restrict__ label, float *__restrict__ loss_pix,
                                                      float *__restrict__ lse2_out, int *__restrict__ correct, int C, int h, int w,
                                                      int ignore_index) {
    constexpr int CH = 8, P = F * G;
    const int b = blockIdx.z;
    const int H = F * h, W = F * w;
    const int X = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = X < W;
    const int Xc = min(X, W - 1);
    const int kx = Xc / F, rx = Xc % F;
    const bool left = rx < F / 2;
    const int xa = left ? max(kx - 1, 0) : kx;
    const int xb = left ? kx : min(kx + 1, w - 1);
    const float lx = (left ? rx + F / 2 + 0.5f : rx - F / 2 + 0.5f) / F;
    const int j0 = blockIdx.y * G;                    // gaps j0 .. j0+G-1 (gap j lies between tap rows j-1 and j; gaps beyond h do not exist)
    const size_t plane = (size_t)h * w;
    const T *sb = s + (size_t)b * C * plane;
    const int32_t *lb = label + (size_t)b * H * W;
    // tap rows j0-1 .. j0+G-1, clamped into the map (the half gaps at the borders repeat the edge row)
    size_t roff[G + 1];
#pragma unroll
    for (int g = 0; g <= G; ++g) roff[g] = (size_t)min(max(j0 - 1 + g, 0), h - 1) * w;
    float m[P], z[P];
    int mc[P];
#pragma unroll
    for (int p = 0; p < P; ++p) { m[p] = kNegBig; z[p] = 0.f; mc[p] = 0; }

    // Raw taps of a chunk of CH classes: (G+1) tap rows x 2 tap columns per class, 48 registers at F = 4.  The taps of chunk i+1 are
    // requested BEFORE chunk i is folded and waited for after it: the kernel is otherwise bound by the latency of these (L1/L2-resident)
    // loads, not by its arithmetic -- measured 239 us against ~50 us of vector issue.  The loads are asm volatile (the compiler would sink
    // ordinary loads down to the register copy at the end of the iteration, i.e. behind the fold) and therefore waited for explicitly.
    struct Taps { unsigned a[CH][G + 1], b[CH][G + 1]; };
    auto request = [&](Taps &tp, int c0) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const T *pc = sb + (size_t)min(c0 + i, C - 1) * plane;       // classes beyond C are clamped here and masked in fold_chunk
#pragma unroll
            for (int g = 0; g <= G; ++g) {
                tp.a[i][g] = pinned_load1<T>(pc + roff[g] + xa);
                tp.b[i][g] = pinned_load1<T>(pc + roff[g] + xb);
            }
        }
    };
    auto fold_chunk = [&](const Taps &tp, int n, int ci) {
        float v[P][CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            float t[G + 1];
#pragma unroll
            for (int g = 0; g <= G; ++g) {
                const float a0 = raw_to_float<T>(tp.a[i][g]), b0 = raw_to_float<T>(tp.b[i][g]);
                t[g] = fmaf(lx, b0 - a0, a0);
            }
            const bool in = i < n;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float d = t[g + 1] - t[g];
#pragma unroll
                for (int q = 0; q < F; ++q) v[g * F + q][i] = in ? fmaf((q + 0.5f) / F, d, t[g]) : kNegBig;
            }
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            float cm = v[p][0];
#pragma unroll
            for (int i = 1; i < CH; ++i) cm = fmaxf(cm, v[p][i]);
            mc[p] = cm > m[p] ? ci : mc[p];
            const float nm = fmaxf(m[p], cm);
            float zz = z[p] * ex2((m[p] - nm) * kLog2e);
            const float off = -nm * kLog2e;
#pragma unroll
            for (int i = 0; i < CH; ++i) zz += ex2(fmaf(v[p][i], kLog2e, off));
            z[p] = zz;
            m[p] = nm;
        }
    };
    Taps cur, nxt;
    request(cur, 0);
    wait_pinned_loads();
    for (int c0 = 0, ci = 0; c0 < C; c0 += CH, ++ci) {
        request(nxt, c0 + CH);                     // (the request past the last chunk re-reads class C-1: harmless, never folded)
        fold_chunk(cur, min(CH, C - c0), ci);
        wait_pinned_loads();
        cur = nxt;
    }

    int hits = 0;
    if (active) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int j = j0 + g;
            if (j > h) continue;
#pragma unroll
            for (int q = 0; q < F; ++q) {
                const int Y = F * j - F / 2 + q, p = g * F + q;
                if (Y < 0 || Y >= H) continue;
                const size_t o = (size_t)b * H * W + (size_t)Y * W + X;
                const int lab = lb[(size_t)Y * W + X];
                const bool valid = lab != ignore_index && lab >= 0 && lab < C;
                float xl = 0.f;
                if (valid) {
                    const T *pc = sb + (size_t)lab * plane;
                    const float a0 = VecIO<T>::load1(pc + roff[g] + xa), b0 = VecIO<T>::load1(pc + roff[g] + xb);
                    const float a1 = VecIO<T>::load1(pc + roff[g + 1] + xa), b1 = VecIO<T>::load1(pc + roff[g + 1] + xb);
                    const float t0 = fmaf(lx, b0 - a0, a0), t1 = fmaf(lx, b1 - a1, a1);
                    xl = fmaf((q + 0.5f) / F, t1 - t0, t0);
                }
                const float l2 = __builtin_amdgcn_logf(z[p]);  // v_log_f32 = log2
                loss_pix[o] = valid ? (m[p] + l2 * kLn2 - xl) : 0.f;
                lse2_out[o] = fmaf(m[p], kLog2e, l2);
                if (valid && xl == m[p]) {
                    // the label attains the maximum: it is the arg-max unless an earlier class attains it too -- look at the chunk where
                    // the maximum first appeared
                    const int c0 = mc[p] * CH;
                    int first = -1;
                    for (int i = CH - 1; i >= 0; --i) {
                        const int c = c0 + i;
                        if (c >= C) continue;
                        const T *pc = sb + (size_t)c * plane;
                        const float a0 = VecIO<T>::load1(pc + roff[g] + xa), b0 = VecIO<T>::load1(pc + roff[g] + xb);
                        const float a1 = VecIO<T>::load1(pc + roff[g + 1] + xa), b1 = VecIO<T>::load1(pc + roff[g + 1] + xb);
                        const float t0 = fmaf(lx, b0 - a0, a0), t1 = fmaf(lx, b1 - a1, a1);
                        if (fmaf((q + 0.5f) / F, t1 - t0, t0) == m[p]) first = c;
                    }
                    hits += (first == lab) ? 1 : 0;
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) hits += __shfl_xor(hits, o, 64);
    if ((threadIdx.x & 63) == 0 && hits) atomicAdd(correct, hits);
}

// grid.x = B*nband*C with the channel fastest; block: round64(w) threads; dynamic LDS 2*F*blockDim floats.
template <typename T, int F, bool GMAP>
__global__ void ce_up_bwd(const T *__restrict__ s, const int32_t *__restrict__ label, const float *__restrict__ lse2,
                          const float *__restrict__ upstream, float gscale, T *__restrict__ ds, int C, int h, int w, int R, int nband,
                          int ignore_index) {
    extern __shared__ float rowbuf[];
    const int wg = blockIdx.x;
    const int c = wg % C;
    const int k = (wg / C) % nband;
    const int b = wg / (C * nband);
    const int kx = threadIdx.x;
    const bool active = kx < w;
    const int kxc = min(kx, w - 1);
    const int H = F * h, W = F * w;
    const size_t plane = (size_t)h * w;
    const T *pc = s + ((size_t)b * C + c) * plane;
    T *pd = ds + ((size_t)b * C + c) * plane;
    const size_t pix0 = (size_t)b * H * W;
    const float guni = GMAP ? 0.f : gscale * upstream[0];
    const int y0 = k * R, y1 = min(h, y0 + R);
    const int bufstride = F * blockDim.x;

    float sp[F], sc[F], accA[F], accB[F];
#pragma unroll
    for (int rx = 0; rx < F; ++rx) accA[rx] = 0.f;
    hrow<T, F>(pc + (size_t)max(y0 - 1, 0) * w, kxc, w, sp);
    int parity = 0;
    for (int j = y0; j <= y1; ++j) {
        hrow<T, F>(pc + (size_t)min(j, h - 1) * w, kxc, w, sc);
        float dv[F];
#pragma unroll
        for (int rx = 0; rx < F; ++rx) { dv[rx] = sc[rx] - sp[rx]; accB[rx] = 0.f; }
        const bool top = (j == 0), bot = (j == h);
#pragma unroll
        for (int q = 0; q < F; ++q) {
            if ((top && q < F / 2) || (bot && q >= F / 2)) continue;
            const float lam = (q + 0.5f) / F;
            const float wa = top ? 0.f : (bot ? 1.f : 1.f - lam);
            const float wb = top ? 1.f : (bot ? 0.f : lam);
            const size_t o = pix0 + (size_t)(F * j - F / 2 + q) * W + F * kxc;
#pragma unroll
            for (int rx = 0; rx < F; ++rx) {
                const int lab = label[o + rx];
                const float gp = GMAP ? gscale * upstream[o + rx] : guni;
                const float p = ex2(fmaf(fmaf(lam, dv[rx], sp[rx]), kLog2e, -lse2[o + rx]));
                const float D = (lab == ignore_index) ? 0.f : gp * (p - (lab == c ? 1.f : 0.f));
                accA[rx] = fmaf(wa, D, accA[rx]);
                accB[rx] = fmaf(wb, D, accB[rx]);
            }
        }
        if (j > y0) {
            float *buf = rowbuf + parity * bufstride;
#pragma unroll
            for (int rx = 0; rx < F; ++rx) buf[F * kx + rx] = accA[rx];
            __syncthreads();
            if (active) {
                float sum = 0.f;
#pragma unroll
                for (int q = 0; q < F; ++q) {
                    const float lam = (q + 0.5f) / F;
                    const int xl = F * kx - F / 2 + q;
                    const int xr = F * kx + F / 2 + q;
                    if (xl >= 0) sum = fmaf(kx == 0 ? 1.f : lam, buf[xl], sum);
                    if (xr < W) sum = fmaf(kx == w - 1 ? 1.f : 1.f - lam, buf[xr], sum);
                }
                VecIO<T>::store1(pd + (size_t)(j - 1) * w + kx, sum);
            }
            parity ^= 1;
        }
#pragma unroll
        for (int rx = 0; rx < F; ++rx) { accA[rx] = accB[rx]; sp[rx] = sc[rx]; }
    }
}

// ---- round 3: the same backward with CPW class planes per workgroup and every load requested one tap row ahead ---------------------------
// The one-class kernel above re-reads the per-pixel maps (label, log-partition, upstream) once PER CLASS -- 150 x 8 B x 2.1 M pixels = 2.5 GB
// of L2 traffic per launch at config 2, as 4-byte loads issued right in front of their use: 216 us against ~52 us of vector issue.  Here a
// workgroup owns CPW consecutive classes of a band: the pixel maps are loaded once per CPW classes as F-wide vectors (the F pixels a thread
// owns in an output row are contiguous and F*4-byte aligned), and the maps of gap j+1 as well as the taps of tap row j+1 are requested
// BEFORE gap j is computed (asm volatile loads, explicit wait: the compiler would otherwise sink them to their first use, as in the
// forward).  The upstream factor is folded into the two row weights per pixel (one multiply per pixel instead of one per pixel and class), so
// results agree with the one-class kernel to rounding, not bit for bit.
template <int N> struct PinRow;   // N consecutive dwords, N*4-byte aligned
template <> struct PinRow<2> {
    typedef unsigned v2 __attribute__((ext_vector_type(2)));
    v2 r;
    __device__ __forceinline__ void request(const void *p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r) : "v"(p) : "memory"); }
    __device__ __forceinline__ void load(const void *p) { r = *reinterpret_cast<const v2 *>(p); }      // a load hipcc counts
    __device__ __forceinline__ unsigned operator[](int i) const { return r[i]; }
};
template <> struct PinRow<4> {
    typedef unsigned v4 __attribute__((ext_vector_type(4)));
    v4 r;
    __device__ __forceinline__ void request(const void *p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory"); }
    __device__ __forceinline__ void load(const void *p) { r = *reinterpret_cast<const v4 *>(p); }
    __device__ __forceinline__ unsigned operator[](int i) const { return r[i]; }
};
template <> struct PinRow<8> {
    typedef unsigned v4 __attribute__((ext_vector_type(4)));
    v4 lo, hi;
    __device__ __forceinline__ void request(const void *p) {
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(lo) : "v"(p) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(hi) : "v"(p) : "memory");
    }
    __device__ __forceinline__ void load(const void *p) {
        lo = reinterpret_cast<const v4 *>(p)[0];
        hi = reinterpret_cast<const v4 *>(p)[1];
    }
    __device__ __forceinline__ unsigned operator[](int i) const { return i < 4 ? lo[i] : hi[i - 4]; }
};

// grid.x = B*nband*ngrp with the class group fastest; block: round64(w) threads; dynamic LDS CPW*2*F*blockDim floats.
// PF: the pixel maps of gap j+1 are requested while gap j is computed (F <= 4; at F = 8 the second set of 8 x 8 x 2..3 registers does not fit
// and the maps are requested at the top of their own gap, still as vectors and still once per CPW classes).
template <typename T, int F, bool GMAP, int CPW, bool PF>
__global__ __launch_bounds__(256) void ce_up_bwd_mc(const T *__restrict__ s, const int32_t *__restrict__ label, const float *__restrict__ lse2,
                             const float *__restrict__ upstream, float gscale, T *__restrict__ ds, int C, int h, int w, int R, int nband,
                             int ngrp, int ignore_index) {
    extern __shared__ float rowbuf[];   // [CPW][2][F * blockDim]
    const int wg = blockIdx.x;
    const int grp = wg % ngrp;
    const int k = (wg / ngrp) % nband;
    const int b = wg / (ngrp * nband);
    const int kx = threadIdx.x;
    const bool active = kx < w;
    const int kxc = min(kx, w - 1);
    const int H = F * h, W = F * w;
    const size_t plane = (size_t)h * w;
    const int c0 = grp * CPW;
    const T *pc[CPW];
#pragma unroll
    for (int i = 0; i < CPW; ++i) pc[i] = s + ((size_t)b * C + min(c0 + i, C - 1)) * plane;   // classes beyond C: clamped here, never stored
    const size_t pix0 = (size_t)b * H * W;
    const float guni = GMAP ? 0.f : gscale * upstream[0];
    const int y0 = k * R, y1 = min(h, y0 + R);
    const int bufstride = F * blockDim.x;
    const int xl0 = max(kxc - 1, 0), xr0 = min(kxc + 1, w - 1);

    struct Taps { unsigned a[CPW], b[CPW], c[CPW]; };
    struct Pix { PinRow<F> lab[F], lse[F], up[GMAP ? F : 1]; };
    auto request_taps = [&](Taps &t, int row) {
        const size_t ro = (size_t)row * w;
#pragma unroll
        for (int i = 0; i < CPW; ++i) {
            if (PF) {
                t.a[i] = pinned_load1<T>(pc[i] + ro + xl0);
                t.b[i] = pinned_load1<T>(pc[i] + ro + kxc);
                t.c[i] = pinned_load1<T>(pc[i] + ro + xr0);
            } else {      // counted loads only (see request_pix)
                t.a[i] = plain_load1<T>(pc[i] + ro + xl0);
                t.b[i] = plain_load1<T>(pc[i] + ro + kxc);
                t.c[i] = plain_load1<T>(pc[i] + ro + xr0);
            }
        }
    };
    auto request_pix = [&](Pix &px, int j) {
#pragma unroll
        for (int q = 0; q < F; ++q) {
            const int Y = min(max(F * j - F / 2 + q, 0), H - 1);            // rows outside the image (half gaps): clamped, never used
            const size_t o = pix0 + (size_t)Y * W + F * kxc;
            if (PF) {
                px.lab[q].request(label + o);
                px.lse[q].request(lse2 + o);
                if (GMAP) px.up[q].request(upstream + o);
            } else {
                // no prefetch: ordinary loads, which hipcc counts and waits for itself.  (A pinned load's destination counts as written when
                // the asm statement ends; at factor 8 -- ~300 registers -- the compiler parked such registers in AGPRs before the data had
                // landed and a late return overwrote an address: a memory fault.  tools/asm_pending_audit.py scans for exactly that.)
                px.lab[q].load(label + o);
                px.lse[q].load(lse2 + o);
                if (GMAP) px.up[q].load(upstream + o);
            }
        }
    };
    auto hrow_raw = [&](const Taps &t, int i, float (&o)[F]) {             // up_device.h::hrow on already-loaded taps
        const float a = raw_to_float<T>(t.a[i]), bb = raw_to_float<T>(t.b[i]), c = raw_to_float<T>(t.c[i]);
        const float dl = bb - a, dr = c - bb;
#pragma unroll
        for (int rx = 0; rx < F; ++rx) {
            if (rx < F / 2) o[rx] = fmaf((rx + F / 2 + 0.5f) / F, dl, a);
            else o[rx] = fmaf((rx - F / 2 + 0.5f) / F, dr, bb);
        }
    };

    float sp[CPW][F], accA[CPW][F];
    Taps tcur, tnxt;
    Pix pcur, pnxt;
    request_taps(tcur, max(y0 - 1, 0));
    request_taps(tnxt, min(y0, h - 1));
    if (PF) request_pix(pcur, y0);
    if (PF) wait_pinned_loads();
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        hrow_raw(tcur, i, sp[i]);
#pragma unroll
        for (int rx = 0; rx < F; ++rx) accA[i][rx] = 0.f;
    }
    tcur = tnxt;
    int parity = 0;
    for (int j = y0; j <= y1; ++j) {
        if (!PF) request_pix(pcur, j);
        request_taps(tnxt, min(j + 1, h - 1));      // the last iteration's requests re-read valid (clamped) addresses and are dropped
        if (PF) request_pix(pnxt, j + 1);
        float sc[CPW][F], dv[CPW][F], accB[CPW][F];
#pragma unroll
        for (int i = 0; i < CPW; ++i) {
            hrow_raw(tcur, i, sc[i]);
#pragma unroll
            for (int rx = 0; rx < F; ++rx) { accB[i][rx] = 0.f; dv[i][rx] = sc[i][rx] - sp[i][rx]; }
        }
        const bool top = (j == 0), bot = (j == h);
#pragma unroll
        for (int q = 0; q < F; ++q) {
            if ((top && q < F / 2) || (bot && q >= F / 2)) continue;
            const float lam = (q + 0.5f) / F;
            const float wa = top ? 0.f : (bot ? 1.f : 1.f - lam);
            const float wb = top ? 1.f : (bot ? 0.f : lam);
#pragma unroll
            for (int rx = 0; rx < F; ++rx) {
                const int lab = (int)pcur.lab[q][rx];
                const float nl = -__uint_as_float(pcur.lse[q][rx]);
                const float gp = (lab == ignore_index) ? 0.f : (GMAP ? gscale * __uint_as_float(pcur.up[q][rx]) : guni);
                // per pixel, shared by the CPW classes: the row weights with the upstream factor folded in, and the label relative to this group
                const float ga = wa * gp, gb = wb * gp;
                const int rel = lab - c0;
#pragma unroll
                for (int i = 0; i < CPW; ++i) {
                    const float p = ex2(fmaf(fmaf(lam, dv[i][rx], sp[i][rx]), kLog2e, nl));
                    const float D = p - (rel == i ? 1.f : 0.f);           // softmax - onehot: 7 vector issues + 1 exponential per (pixel, class)
                    accA[i][rx] = fmaf(ga, D, accA[i][rx]);
                    accB[i][rx] = fmaf(gb, D, accB[i][rx]);
                }
            }
        }
        if (j > y0) {
#pragma unroll
            for (int i = 0; i < CPW; ++i) {
                float *buf = rowbuf + (2 * i + parity) * bufstride;
#pragma unroll
                for (int rx = 0; rx < F; ++rx) buf[F * kx + rx] = accA[i][rx];
            }
            __syncthreads();
            if (active) {
#pragma unroll
                for (int i = 0; i < CPW; ++i) {
                    if (c0 + i >= C) continue;
                    const float *buf = rowbuf + (2 * i + parity) * bufstride;
                    float sum = 0.f;
#pragma unroll
                    for (int q = 0; q < F; ++q) {
                        const float lam = (q + 0.5f) / F;
                        const int xl = F * kx - F / 2 + q;
                        const int xr = F * kx + F / 2 + q;
                        if (xl >= 0) sum = fmaf(kx == 0 ? 1.f : lam, buf[xl], sum);
                        if (xr < W) sum = fmaf(kx == w - 1 ? 1.f : 1.f - lam, buf[xr], sum);
                    }
                    VecIO<T>::store1(ds + ((size_t)b * C + c0 + i) * plane + (size_t)(j - 1) * w + kx, sum);
                }
            }
            parity ^= 1;
        }
#pragma unroll
        for (int i = 0; i < CPW; ++i)
#pragma unroll
            for (int rx = 0; rx < F; ++rx) { accA[i][rx] = accB[i][rx]; sp[i][rx] = sc[i][rx]; }
        if (PF) {
            // The requests of this iteration are OLDER than its gradient stores (vmcnt retires in order): where the wave is known to have issued
            // exactly CPW stores, wait for all but those -- vmcnt(0) made every tap row wait for its own stores to be acknowledged (~1 us, nine
            // times per workgroup).  Wave-uniform conditions: a stored row (j > y0), a full class group, at least one active lane in the wave.
            if (j > y0 && c0 + CPW <= C && (int)(threadIdx.x & ~63u) < w) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CPW) : "memory");
                __builtin_amdgcn_sched_barrier(0);
            } else {
                wait_pinned_loads();
            }
        }
        tcur = tnxt;
        if (PF) pcur = pnxt;
    }
}

// Zero the hit counter with a kernel, not hipMemsetAsync: inside a captured hipGraph a memset becomes a memset NODE, and when two
// graphs are replayed concurrently on different streams (teacher graph || student-step graph, engine/trainer.py) the fill value of
// such nodes was observed to be taken from the OTHER graph's memset (counter came back as 0x01010101 + hits on ROCm 7.2 / gfx950).
__global__ void zero_counter(int *p) { *p = 0; }

int ce_factor(int h, int w, int H, int W) {
    if (h <= 0 || w <= 0 || H % h || W % w) return 0;
    const int f = H / h;
    if (W / w != f || (f != 2 && f != 4 && f != 8)) return 0;
    if (w > 1024 || (long)f * ((w + 63) / 64 * 64) > 8192) return 0;
    return f;
}

constexpr int kCeBand = 8;  // tap rows per workgroup

}  // namespace

int g_ce_bwd_multiclass = 1;   // tunable "ce_bwd_multiclass": 0 = the one-class-per-workgroup backward of rounds 1-2 (A/B, tests)

int ce_tunable(const char *key, int set, int v) {
    if (strcmp(key, "ce_bwd_multiclass")) return SD_E_UNSUPPORTED;
    if (!set) return g_ce_bwd_multiclass;
    if (v != 0 && v != 1) return SD_E_SHAPE;
    g_ce_bwd_multiclass = v;
    return SD_OK;
}

namespace {

}  // namespace
}  // namespace sd

extern "C" {

int sd_ce_up_supported(int h, int w, int H, int W) { return sd::ce_factor(h, w, H, W) ? 1 : 0; }

int sd_ce_up_fwd(const void *logits, const int32_t *label, float *loss_pix, float *pix_lse2, int *correct, int dtype, int B, int C, int h,
                 int w, int H, int W, int ignore_index, void *stream) {
    if (!logits || !label || !loss_pix || !pix_lse2 || !correct) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (B <= 0 || C <= 0 || B > 65535) return SD_E_SHAPE;
    const int F = sd::ce_factor(h, w, H, W);
    if (!F) return SD_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the forward keeps no state across gaps (the class loop is innermost): G gaps (h + 1 of them: the two border gaps are half gaps) per
    // workgroup row, one output column per thread
    const int threads = W >= 256 ? 256 : (W + 63) / 64 * 64;
    hipLaunchKernelGGL(sd::zero_counter, dim3(1), dim3(1), 0, st, correct);
#define SD_CE_FWD(TT, FF, GG)                                                                                                        \
    hipLaunchKernelGGL((sd::ce_up_fwd_col<TT, FF, GG>), dim3((W + threads - 1) / threads, (h + 1 + GG - 1) / GG, B), dim3(threads), 0, st, \
                       (const TT *)logits, label, loss_pix, pix_lse2, correct, C, h, w, ignore_index)
    if (dtype == SD_F32) {
        if (F == 2) SD_CE_FWD(float, 2, 4);
        else if (F == 4) SD_CE_FWD(float, 4, 2);
        else SD_CE_FWD(float, 8, 1);
    } else {
        if (F == 2) SD_CE_FWD(sd::bf16_t, 2, 4);
        else if (F == 4) SD_CE_FWD(sd::bf16_t, 4, 2);
        else SD_CE_FWD(sd::bf16_t, 8, 1);
    }
#undef SD_CE_FWD
    return (int)hipGetLastError();
}

int sd_ce_up_bwd(const void *logits, const int32_t *label, const float *pix_lse2, const float *upstream, int upstream_is_map, float gscale,
                 void *dlogits, int dtype, int B, int C, int h, int w, int H, int W, int ignore_index, void *stream) {
    if (!logits || !label || !pix_lse2 || !upstream || !dlogits) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (B <= 0 || C <= 0) return SD_E_SHAPE;
    const int F = sd::ce_factor(h, w, H, W);
    if (!F) return SD_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int R = sd::kCeBand < h ? sd::kCeBand : h, nband = (h + R - 1) / R, threads = (w + 63) / 64 * 64;
    // multi-class form (round 3): needs F*4-byte aligned pixel maps (16 bytes covers every F) -- torch allocations always are
    const bool aligned = ((reinterpret_cast<uintptr_t>(label) | reinterpret_cast<uintptr_t>(pix_lse2) |
                           (upstream_is_map ? reinterpret_cast<uintptr_t>(upstream) : 0)) & 15) == 0;
    if (aligned && threads <= 256 && sd::g_ce_bwd_multiclass) {
#define SD_CE_BWD_MC1(TT, FF, GG, CC, PP)                                                                                            \
    do {                                                                                                                             \
        const int ngrp = (C + CC - 1) / CC;                                                                                          \
        const long nwg = (long)B * nband * ngrp;                                                                                     \
        if (nwg > 0x7fffffffL) return SD_E_SHAPE;                                                                                    \
        const size_t lds = 2ull * CC * FF * threads * sizeof(float);                                                                 \
        hipLaunchKernelGGL((sd::ce_up_bwd_mc<TT, FF, GG, CC, PP>), dim3((unsigned)nwg), dim3(threads), lds, st, (const TT *)logits, label,  \
                           pix_lse2, upstream, gscale, (TT *)dlogits, C, h, w, R, nband, ngrp, ignore_index);                        \
    } while (0)
    // classes per workgroup: as many as fit 256 registers with nothing parked in AGPRs while loads are pending (tools/asm_pending_audit.py):
    // 6 with a uniform upstream gradient (the training step: 25 groups for 150 classes), 4 with a per-pixel one, 2 at factor 8
#define SD_CE_BWD_MC(TT, FF, CMAP, CUNI, PP)                                                                                       \
    do {                                                                                                                             \
        if (upstream_is_map) SD_CE_BWD_MC1(TT, FF, true, CMAP, PP);                                                                  \
        else SD_CE_BWD_MC1(TT, FF, false, CUNI, PP);                                                                                 \
    } while (0)
        if (dtype == SD_F32) {
            if (F == 2) SD_CE_BWD_MC(float, 2, 4, 4, true);
            else if (F == 4) SD_CE_BWD_MC(float, 4, 4, 6, true);
            else SD_CE_BWD_MC(float, 8, 2, 2, false);
        } else {
            if (F == 2) SD_CE_BWD_MC(sd::bf16_t, 2, 4, 4, true);
            else if (F == 4) SD_CE_BWD_MC(sd::bf16_t, 4, 4, 6, true);
            else SD_CE_BWD_MC(sd::bf16_t, 8, 2, 2, false);
        }
#undef SD_CE_BWD_MC1
#undef SD_CE_BWD_MC
        return (int)hipGetLastError();
    }
    const long nwg = (long)B * nband * C;
    if (nwg > 0x7fffffffL) return SD_E_SHAPE;
    const size_t lds = 2ull * F * threads * sizeof(float);
#define SD_CE_BWD(TT, FF)                                                                                                            \
    do {                                                                                                                             \
        if (upstream_is_map)                                                                                                         \
            hipLaunchKernelGGL((sd::ce_up_bwd<TT, FF, true>), dim3((unsigned)nwg), dim3(threads), lds, st, (const TT *)logits, label, pix_lse2, \
                               upstream, gscale, (TT *)dlogits, C, h, w, R, nband, ignore_index);                                    \
        else                                                                                                                         \
            hipLaunchKernelGGL((sd::ce_up_bwd<TT, FF, false>), dim3((unsigned)nwg), dim3(threads), lds, st, (const TT *)logits, label, pix_lse2, \
                               upstream, gscale, (TT *)dlogits, C, h, w, R, nband, ignore_index);                                    \
    } while (0)
    if (dtype == SD_F32) {
        if (F == 2) SD_CE_BWD(float, 2);
        else if (F == 4) SD_CE_BWD(float, 4);
        else SD_CE_BWD(float, 8);
    } else {
        if (F == 2) SD_CE_BWD(sd::bf16_t, 2);
        else if (F == 4) SD_CE_BWD(sd::bf16_t, 4);
        else SD_CE_BWD(sd::bf16_t, 8);
    }
#undef SD_CE_BWD
    return (int)hipGetLastError();
}

}  // extern "C"
