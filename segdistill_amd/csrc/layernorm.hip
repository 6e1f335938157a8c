// layernorm.hip -- LayerNorm over the last (channel) axis of token-major activations [rows, C],
// forward and backward, gfx950.
//
// The MiT encoders normalise tokens 83 times per KD step (every block's norm1/norm2, the SR-attention norm,
// the patch-embed norm and the stage norm: reference mix_transformer.py:96,139,169-171,214,336-365) with small
// C (32..512) and up to 131072 rows.  That is HBM-bound byte work (read x once, write y once), but the generic
// ATen kernels spend 38 us on average where the largest instance needs ~6-11 us of traffic; after the depth-wise
// and loss kernels were replaced LayerNorm was the largest single item of the step (11 % of GPU time).
//
// Mapping: a row is handled by a GROUP of G lanes (G = power of two <= 64, G*4*V >= C), each lane holding V
// float4 vectors of the row in registers; mean / variance (two-pass, in registers) by xor-shuffles inside the
// group, so a wave processes 64/G rows at once and small C does not idle lanes.
//   fwd:  y = (x - mean) * rstd * gamma + beta ; saves mean, rstd [rows]
//   bwd:  dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma ;
//         dgamma = sum_rows dy*xhat, dbeta = sum_rows dy  (per-workgroup partials, deterministic second pass)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cgd_device.h"

namespace sd {

namespace {

constexpr int kLnThreads = 256;

template <typename T> struct LV;
template <> struct LV<float> {
    static __device__ __forceinline__ float4 load(const float *p) { return *reinterpret_cast<const float4 *>(p); }
    static __device__ __forceinline__ void store(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
    static __device__ __forceinline__ float4 round(float4 v) { return v; }
};
template <> struct LV<bf16_t> {  // 4 bf16 = 8 bytes
    static __device__ __forceinline__ float4 load(const bf16_t *p) {
        const uint2 v = *reinterpret_cast<const uint2 *>(p);
        return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                           __uint_as_float(v.y & 0xffff0000u));
    }
    static __device__ __forceinline__ void store(bf16_t *p, float4 v) {
        uint2 o;
        o.x = (unsigned)f32_to_bf16(v.x) | ((unsigned)f32_to_bf16(v.y) << 16);
        o.y = (unsigned)f32_to_bf16(v.z) | ((unsigned)f32_to_bf16(v.w) << 16);
        *reinterpret_cast<uint2 *>(p) = o;
    }
    static __device__ __forceinline__ float4 round(float4 v) {  // value after a store/load round trip
        return make_float4(__uint_as_float((unsigned)f32_to_bf16(v.x) << 16), __uint_as_float((unsigned)f32_to_bf16(v.y) << 16),
                           __uint_as_float((unsigned)f32_to_bf16(v.z) << 16), __uint_as_float((unsigned)f32_to_bf16(v.w) << 16));
    }
};

template <int G> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Patch geometry of the spatial-reduction attention that consumes a LayerNorm output (round 3): token row (b, yy, xx) of a [B, H*W, C] map goes
// to row ((b, yy / r, xx / r), (yy % r, xx % r)) of the [B, (H/r)(W/r), r*r*C] patch matrix that the SR conv multiplies as a Linear
// (mix_transformer.py:75-84,121-124).  H, W, r powers of two (every MiT stage at 512 x 512): shifts only.  lr < 0: no patch output.
struct PatchGeom {
    int lw, lhw, lr;      // log2 W, log2 (H W), log2 r
};
__device__ __forceinline__ long patch_row(long row, const PatchGeom g) {
    const long b = row >> g.lhw;
    const int rem = (int)(row & ((1L << g.lhw) - 1)), yy = rem >> g.lw, xx = rem & ((1 << g.lw) - 1);
    const int lh = g.lhw - g.lw;
    return ((((b << (lh - g.lr)) + (yy >> g.lr)) << (g.lw - g.lr)) + (xx >> g.lr)) * (1L << (2 * g.lr)) + ((yy & ((1 << g.lr) - 1)) << g.lr) +
           (xx & ((1 << g.lr) - 1));
}

// grid: ceil(rows / rows_per_block); rows_per_block = kLnThreads / G
// (One row per row-group: batching 4 rows per group the way ln_bwd does measured 5-20 % SLOWER here -- 6.8 -> 8.0 us at 131072 x 32 f32 --
// the forward has no accumulators to carry and the larger grid already keeps 8 waves per SIMD in flight.)
// Residual form (res != nullptr): the row that is normalised is  xsum = x + scale * res  (scale = row_scale[row / rows_per_sample],
// the stochastic-depth factor of that sample, or 1), and xsum is written out as well -- the block's "x = x + drop_path(f(x))"
// followed by the next LayerNorm in ONE pass (reference mix_transformer.py:150-151).
template <typename T, int G, int V>
__global__ __launch_bounds__(kLnThreads) void ln_fwd(const T *__restrict__ x, const T *__restrict__ res, const float *__restrict__ row_scale,
                                                      long rows_per_sample, T *__restrict__ xsum, const float *__restrict__ gamma,
                                                      const float *__restrict__ beta, T *__restrict__ y, float *__restrict__ mean_out,
                                                      float *__restrict__ rstd_out, long rows, int C, float eps, T *__restrict__ y2,
                                                      PatchGeom pg) {
    const int gl = threadIdx.x % G;
    const long row = (long)blockIdx.x * (kLnThreads / G) + threadIdx.x / G;
    const bool live = row < rows;
    const int cv = C / 4;
    float4 v[V];
    float s = 0.f;
    const float sc = (res && row_scale && live) ? row_scale[row / rows_per_sample] : 1.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int j = gl + i * G;
        v[i] = (live && j < cv) ? LV<T>::load(x + row * C + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (res && live && j < cv) {
            const float4 rv = LV<T>::load(res + row * C + 4 * j);
            v[i].x = fmaf(sc, rv.x, v[i].x); v[i].y = fmaf(sc, rv.y, v[i].y); v[i].z = fmaf(sc, rv.z, v[i].z); v[i].w = fmaf(sc, rv.w, v[i].w);
            LV<T>::store(xsum + row * C + 4 * j, v[i]);
            v[i] = LV<T>::round(v[i]);   // normalise exactly what later passes will read back
        }
        s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const float mean = group_sum<G>(s) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int j = gl + i * G;
        if (j < cv) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += a * a + b * b + c * c + d * d;
        }
    }
    const float rstd = rsqrtf(group_sum<G>(q) / C + eps);
    if (!live) return;
    const long prow = y2 ? patch_row(row, pg) : 0;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int j = gl + i * G;
        if (j < cv) {
            const float4 g = *reinterpret_cast<const float4 *>(gamma + 4 * j), b = *reinterpret_cast<const float4 *>(beta + 4 * j);
            float4 o;
            o.x = (v[i].x - mean) * rstd * g.x + b.x;
            o.y = (v[i].y - mean) * rstd * g.y + b.y;
            o.z = (v[i].z - mean) * rstd * g.z + b.z;
            o.w = (v[i].w - mean) * rstd * g.w + b.w;
            LV<T>::store(y + row * C + 4 * j, o);
            if (y2) LV<T>::store(y2 + prow * C + 4 * j, o);       // the same values in patch order: the SR path's gather copy, folded in
        }
    }
    if (gl == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// Inference form with ROW MAPS (round 4: the Swin blocks of a frozen network, reference swin_transformer.py:195-262).  Output row r of image b
// (L_out rows per image) normalises
//     v = x[b][xi] (+ res[b][ri]),   xi = x_map ? x_map[r] : r,   ri = res_map ? res_map[r] : xi
// and, with res, also writes v to xsum[b][xi].  xi == L_in (one past the image's rows) means "padding": the output row is zeros and nothing
// else is read or written.  With x_map = the window-partition table this is  pad + cyclic shift + window_partition(norm1(x [+ pending]))  in
// one pass (:206-221); with res_map = its inverse it is  x + window_reverse / un-shift / un-pad (attention output)  followed by norm2
// (:232-250).  No mean / rstd outputs: there is no backward.
template <typename T, int G, int V>
__global__ __launch_bounds__(kLnThreads) void ln_map_fwd(const T *__restrict__ x, const T *__restrict__ res, T *__restrict__ xsum,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta, T *__restrict__ y,
                                                          const int *__restrict__ x_map, const int *__restrict__ res_map, long rows_out, int L_out,
                                                          int L_in, int L_res, int C, float eps) {
    const int gl = threadIdx.x % G;
    const long row = (long)blockIdx.x * (kLnThreads / G) + threadIdx.x / G;
    if (row >= rows_out) return;                       // whole row groups leave together (the sums below are group-wide shuffles)
    const int cv = C / 4;
    const long b = row / L_out;
    const int r = (int)(row - b * L_out);
    const int xi = x_map ? x_map[r] : r;
    if (xi >= L_in) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int j = gl + i * G;
            if (j < cv) LV<T>::store(y + row * C + 4 * j, make_float4(0.f, 0.f, 0.f, 0.f));
        }
        return;
    }
    const long xrow = b * L_in + xi;
    const long rrow = res ? b * L_res + (res_map ? res_map[r] : xi) : 0;
    float4 v[V];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int j = gl + i * G;
        v[i] = j < cv ? LV<T>::load(x + xrow * C + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (res && j < cv) {
            const float4 rv = LV<T>::load(res + rrow * C + 4 * j);
            v[i].x += rv.x; v[i].y += rv.y; v[i].z += rv.z; v[i].w += rv.w;
            LV<T>::store(xsum + xrow * C + 4 * j, v[i]);
            v[i] = LV<T>::round(v[i]);
        }
        s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const float mean = group_sum<G>(s) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int j = gl + i * G;
        if (j < cv) {
            const float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += a * a + bb * bb + c * c + d * d;
        }
    }
    const float rstd = rsqrtf(group_sum<G>(q) / C + eps);
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int j = gl + i * G;
        if (j < cv) {
            const float4 g = *reinterpret_cast<const float4 *>(gamma + 4 * j), bt = *reinterpret_cast<const float4 *>(beta + 4 * j);
            float4 o;
            o.x = (v[i].x - mean) * rstd * g.x + bt.x;
            o.y = (v[i].y - mean) * rstd * g.y + bt.y;
            o.z = (v[i].z - mean) * rstd * g.z + bt.z;
            o.w = (v[i].w - mean) * rstd * g.w + bt.w;
            LV<T>::store(y + row * C + 4 * j, o);
        }
    }
}

// grid: nblk workgroups, each walking rows blockIdx.x*RPB + k*gridDim.x*RPB ...; partial dgamma/dbeta per workgroup in
// part[blk][2][C] (gamma first).
// Residual form: dres (nullable) is the gradient that reaches the normalised row from its OTHER consumer (the residual path) and is
// added into dx; dr (nullable) receives row_scale * dx, the gradient of the scaled residual branch.
template <typename T, int G, int V>
__global__ __launch_bounds__(kLnThreads) void ln_bwd(const T *__restrict__ x, const T *__restrict__ dy, const float *__restrict__ gamma,
                                                      const float *__restrict__ mean_in, const float *__restrict__ rstd_in,
                                                      const T *__restrict__ dres, const float *__restrict__ row_scale, long rows_per_sample,
                                                      T *__restrict__ dx, T *__restrict__ dr, float *__restrict__ part, long rows, int C,
                                                      const T *__restrict__ dy2, PatchGeom pg) {
    extern __shared__ float red[];  // [RPB][2][C]
    constexpr int RPB = kLnThreads / G;
    const int gl = threadIdx.x % G, rg = threadIdx.x / G;
    const int cv = C / 4;
    float4 gam[V], ag[V], ab[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int j = gl + i * G;
        gam[i] = j < cv ? *reinterpret_cast<const float4 *>(gamma + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
        ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // U rows of this row-group are loaded before any of them is reduced: with one 16-byte vector per lane (the C <= 256 stages) a wave
    // otherwise keeps 1-2 KB in flight and the walk runs at the load latency, ~2 us per row (35 us for 131072 x 64 bf16 against an 8 us
    // HBM floor).  The rows are still accumulated in walk order, so the results do not depend on U.
    constexpr int U = V == 1 ? 4 : (V == 2 ? 2 : 1);
    const long stride = (long)gridDim.x * RPB;
    for (long row0 = (long)blockIdx.x * RPB + rg; row0 < rows; row0 += U * stride) {
        float4 xr[U][V], dr_in[U][V], er[U][V];
        float mu[U], rs[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long row = row0 + u * stride;
            const bool live = row < rows;
            mu[u] = live ? mean_in[row] : 0.f;
            rs[u] = live ? rstd_in[row] : 0.f;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const int j = gl + i * G;
                const bool on = live && j < cv;
                xr[u][i] = on ? LV<T>::load(x + row * C + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
                dr_in[u][i] = on ? LV<T>::load(dy + row * C + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (dy2 && on) {      // the gradient that arrives in PATCH order from the SR path: gathered and added here (was a scatter copy + an add)
                    const float4 d2 = LV<T>::load(dy2 + patch_row(row, pg) * C + 4 * j);
                    dr_in[u][i].x += d2.x; dr_in[u][i].y += d2.y; dr_in[u][i].z += d2.z; dr_in[u][i].w += d2.w;
                }
                er[u][i] = (on && dres) ? LV<T>::load(dres + row * C + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long row = row0 + u * stride;
            if (row >= rows) break;   // uniform over the row-group
            const float mean = mu[u], rstd = rs[u];
            float4 xh[V], g[V];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const int j = gl + i * G;
                if (j < cv) {
                    const float4 xv = xr[u][i], dv = dr_in[u][i];
                    xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
                    g[i] = make_float4(dv.x * gam[i].x, dv.y * gam[i].y, dv.z * gam[i].z, dv.w * gam[i].w);
                    s1 += g[i].x + g[i].y + g[i].z + g[i].w;
                    s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
                    ag[i].x += dv.x * xh[i].x; ag[i].y += dv.y * xh[i].y; ag[i].z += dv.z * xh[i].z; ag[i].w += dv.w * xh[i].w;
                    ab[i].x += dv.x; ab[i].y += dv.y; ab[i].z += dv.z; ab[i].w += dv.w;
                }
            }
            const float m1 = group_sum<G>(s1) / C, m2 = group_sum<G>(s2) / C;
            const float sc = (dr && row_scale) ? row_scale[row / rows_per_sample] : 1.f;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const int j = gl + i * G;
                if (j < cv) {
                    float4 o;
                    o.x = rstd * (g[i].x - m1 - xh[i].x * m2);
                    o.y = rstd * (g[i].y - m1 - xh[i].y * m2);
                    o.z = rstd * (g[i].z - m1 - xh[i].z * m2);
                    o.w = rstd * (g[i].w - m1 - xh[i].w * m2);
                    if (dres) {
                        const float4 e = er[u][i];
                        o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
                    }
                    LV<T>::store(dx + row * C + 4 * j, o);
                    if (dr) {
                        o = LV<T>::round(o);
                        LV<T>::store(dr + row * C + 4 * j, make_float4(sc * o.x, sc * o.y, sc * o.z, sc * o.w));
                    }
                }
            }
        }
    }
    // combine the RPB row-groups of this workgroup
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int j = gl + i * G;
        if (j < cv) {
            float *pg = red + ((size_t)rg * 2 + 0) * C + 4 * j, *pb = red + ((size_t)rg * 2 + 1) * C + 4 * j;
            pg[0] = ag[i].x; pg[1] = ag[i].y; pg[2] = ag[i].z; pg[3] = ag[i].w;
            pb[0] = ab[i].x; pb[1] = ab[i].y; pb[2] = ab[i].z; pb[3] = ab[i].w;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * C; e += kLnThreads) {
        float s = 0.f;
        for (int r = 0; r < RPB; ++r) s += red[(size_t)r * 2 * C + e];
        part[(size_t)blockIdx.x * 2 * C + e] = s;
    }
}

__global__ __launch_bounds__(256) void ln_param_reduce(const float *__restrict__ part, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                        int nblk, int C) {
    __shared__ float red[16][17];
    const int o = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + o;
    float acc = 0.f;
    if (e < 2 * C)
        for (int p = grp; p < nblk; p += 16) acc += part[(size_t)p * 2 * C + e];
    red[grp][o] = acc;
    __syncthreads();
    if (grp == 0 && e < 2 * C) {
        float t = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < 16; ++g2) t += red[g2][o];
        if (e < C) dgamma[e] = t;
        else dbeta[e - C] = t;
    }
}

struct LnPlan {
    int G, V;
};
LnPlan ln_plan(int C) {
    const int cv = C / 4;
    int G = 1;
    while (G < cv && G < 64) G <<= 1;
    if (G < 8) G = 8;
    const int V = (cv + G - 1) / G;
    return {G, V};
}
int ln_bwd_blocks(long rows, int G) {
    const long rpb = kLnThreads / G;
    long n = (rows + rpb - 1) / rpb;
    if (n > 512) n = 512;  // 2 workgroups per CU (1024 measured 5-10 % slower with the batched loads); keeps the parameter-gradient partial array (and its reduce pass) small
    return (int)(n < 1 ? 1 : n);
}

#define SD_LN_DISPATCH(CALL)                                                  \
    do {                                                                      \
        if (p.G == 8 && p.V == 1) { CALL(8, 1); }                             \
        else if (p.G == 16 && p.V == 1) { CALL(16, 1); }                      \
        else if (p.G == 32 && p.V == 1) { CALL(32, 1); }                      \
        else if (p.G == 64 && p.V == 1) { CALL(64, 1); }                      \
        else if (p.G == 64 && p.V == 2) { CALL(64, 2); }                      \
        else if (p.G == 64 && p.V <= 4) { CALL(64, 4); }                      \
        else return SD_E_UNSUPPORTED;                                         \
    } while (0)

template <typename T>
int ln_fwd_launch(const void *x, const void *res, const float *row_scale, long rows_per_sample, void *xsum, const float *gamma,
                  const float *beta, void *y, float *mean, float *rstd, long rows, int C, float eps, hipStream_t st, void *y2 = nullptr,
                  PatchGeom pg = PatchGeom{0, 0, -1}) {
    const LnPlan p = ln_plan(C);
#define SD_CALL(GG, VV)                                                                                                              \
    hipLaunchKernelGGL((ln_fwd<T, GG, VV>), dim3((unsigned)((rows + kLnThreads / GG - 1) / (kLnThreads / GG))), dim3(kLnThreads), 0, st, \
                       (const T *)x, (const T *)res, row_scale, rows_per_sample, (T *)xsum, gamma, beta, (T *)y, mean, rstd, rows, C, eps,  \
                       (T *)y2, pg)
    SD_LN_DISPATCH(SD_CALL);
#undef SD_CALL
    return (int)hipGetLastError();
}

template <typename T>
int ln_map_launch(const void *x, const void *res, void *xsum, const float *gamma, const float *beta, void *y, const int *x_map, const int *res_map,
                  long rows_out, int L_out, int L_in, int L_res, int C, float eps, hipStream_t st) {
    const LnPlan p = ln_plan(C);
#define SD_CALL(GG, VV)                                                                                                                  \
    hipLaunchKernelGGL((ln_map_fwd<T, GG, VV>), dim3((unsigned)((rows_out + kLnThreads / GG - 1) / (kLnThreads / GG))), dim3(kLnThreads), 0, st, \
                       (const T *)x, (const T *)res, (T *)xsum, gamma, beta, (T *)y, x_map, res_map, rows_out, L_out, L_in, L_res, C, eps)
    SD_LN_DISPATCH(SD_CALL);
#undef SD_CALL
    return (int)hipGetLastError();
}

template <typename T>
int ln_bwd_launch(const void *x, const void *dy, const float *gamma, const float *mean, const float *rstd, const void *dres,
                  const float *row_scale, long rows_per_sample, void *dx, void *dr, float *dgamma, float *dbeta, void *ws, size_t ws_bytes,
                  long rows, int C, hipStream_t st, const void *dy2 = nullptr, PatchGeom pg = PatchGeom{0, 0, -1}) {
    const LnPlan p = ln_plan(C);
    const int nblk = ln_bwd_blocks(rows, p.G);
    if (ws_bytes < (size_t)nblk * 2 * C * sizeof(float) || (reinterpret_cast<uintptr_t>(ws) & 15)) return SD_E_WORKSPACE;
    float *part = static_cast<float *>(ws);
    const size_t lds = (size_t)(kLnThreads / p.G) * 2 * C * sizeof(float);
    if (lds > 64 * 1024) return SD_E_UNSUPPORTED;
#define SD_CALL(GG, VV)                                                                                                              \
    hipLaunchKernelGGL((ln_bwd<T, GG, VV>), dim3(nblk), dim3(kLnThreads), lds, st, (const T *)x, (const T *)dy, gamma, mean, rstd,      \
                       (const T *)dres, row_scale, rows_per_sample, (T *)dx, (T *)dr, part, rows, C, (const T *)dy2, pg)
    SD_LN_DISPATCH(SD_CALL);
#undef SD_CALL
    // dgamma == dbeta == NULL: the partials [nblk][2][C] stay in the workspace for a deferred combine (sd_multi_slab_reduce)
    if (dgamma || dbeta) hipLaunchKernelGGL(ln_param_reduce, dim3((2 * C + 15) / 16), dim3(256), 0, st, part, dgamma, dbeta, nblk, C);
    return (int)hipGetLastError();
}

int check_ln(const void *a, const void *b, int dtype, long rows, int C) {
    if (!a || !b) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (rows <= 0 || C <= 0) return SD_E_SHAPE;
    if (C % 4 || C > 1024) return SD_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) return SD_E_ALIGN;
    return SD_OK;
}

// H, W, r powers of two, r | H, r | W, rows a multiple of H*W
int patch_geom(long rows, int H, int W, int r, PatchGeom *g) {
    auto lg = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
    const int lh = lg(H), lw = lg(W), lr = lg(r);
    if (H <= 0 || W <= 0 || r <= 1 || lh < 0 || lw < 0 || lr < 0 || lr > lh || lr > lw) return SD_E_UNSUPPORTED;
    if (rows % ((long)H * W)) return SD_E_SHAPE;
    g->lw = lw; g->lhw = lh + lw; g->lr = lr;
    return SD_OK;
}

}  // namespace
}  // namespace sd

extern "C" {

int sd_layernorm_patch_supported(int H, int W, int r) {
    sd::PatchGeom g;
    return sd::patch_geom((long)H * W, H, W, r, &g) == SD_OK ? 1 : 0;
}

/* LayerNorm (res == NULL) or residual-add + LayerNorm whose output is ALSO written in the patch order of the spatial-reduction attention
 * that consumes it, and the matching backward that gathers the patch-order gradient (dy_patches, may be NULL) into dy. */
int sd_add_layernorm_patch_fwd(const void *x, const void *res, const float *row_scale, long rows_per_sample, void *xsum, const float *gamma,
                               const float *beta, void *y, void *y_patches, float *mean, float *rstd, int dtype, long rows, int C, float eps, int H,
                               int W, int r, void *stream) {
    int rc = sd::check_ln(x, y, dtype, rows, C);
    if (rc) return rc;
    if (res && (rc = sd::check_ln(res, xsum, dtype, rows, C))) return rc;
    if (!gamma || !beta || !mean || !rstd || !y_patches) return SD_E_NULL;
    if (reinterpret_cast<uintptr_t>(y_patches) & 15) return SD_E_ALIGN;
    if (res && row_scale && (rows_per_sample <= 0 || rows % rows_per_sample)) return SD_E_SHAPE;
    sd::PatchGeom g;
    if ((rc = sd::patch_geom(rows, H, W, r, &g))) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32)
        return sd::ln_fwd_launch<float>(x, res, row_scale, rows_per_sample, xsum, gamma, beta, y, mean, rstd, rows, C, eps, st, y_patches, g);
    return sd::ln_fwd_launch<sd::bf16_t>(x, res, row_scale, rows_per_sample, xsum, gamma, beta, y, mean, rstd, rows, C, eps, st, y_patches, g);
}

int sd_add_layernorm_patch_bwd(const void *xsum, const void *dy, const void *dy_patches, const float *gamma, const float *mean, const float *rstd,
                               const void *dres, const float *row_scale, long rows_per_sample, void *dx, void *dr, float *dgamma, float *dbeta,
                               int dtype, long rows, int C, int H, int W, int r, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = sd::check_ln(xsum, dy, dtype, rows, C);
    if (rc) return rc;
    if (!gamma || !mean || !rstd || !dx || !workspace || (!dgamma != !dbeta)) return SD_E_NULL;
    if ((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(dres) | reinterpret_cast<uintptr_t>(dr) | reinterpret_cast<uintptr_t>(dy_patches)) & 15)
        return SD_E_ALIGN;
    if (row_scale && (rows_per_sample <= 0 || rows % rows_per_sample)) return SD_E_SHAPE;
    sd::PatchGeom g{0, 0, -1};
    if (dy_patches && (rc = sd::patch_geom(rows, H, W, r, &g))) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32)
        return sd::ln_bwd_launch<float>(xsum, dy, gamma, mean, rstd, dres, row_scale, rows_per_sample, dx, dr, dgamma, dbeta, workspace,
                                        workspace_bytes, rows, C, st, dy_patches, g);
    return sd::ln_bwd_launch<sd::bf16_t>(xsum, dy, gamma, mean, rstd, dres, row_scale, rows_per_sample, dx, dr, dgamma, dbeta, workspace,
                                         workspace_bytes, rows, C, st, dy_patches, g);
}

int sd_layernorm_supported(int C) { return (C > 0 && C % 4 == 0 && C <= 1024) ? 1 : 0; }

int sd_layernorm_bwd_blocks(long rows, int C) {
    if (rows <= 0 || C <= 0 || C % 4) return 0;
    return sd::ln_bwd_blocks(rows, sd::ln_plan(C).G);
}

size_t sd_layernorm_workspace_bytes(long rows, int C) {
    if (rows <= 0 || C <= 0 || C % 4) return 0;
    const sd::LnPlan p = sd::ln_plan(C);
    return (size_t)sd::ln_bwd_blocks(rows, p.G) * 2 * C * sizeof(float) + 16;
}

int sd_layernorm_fwd(const void *x, const float *gamma, const float *beta, void *y, float *mean, float *rstd, int dtype, long rows, int C,
                     float eps, void *stream) {
    int rc = sd::check_ln(x, y, dtype, rows, C);
    if (rc) return rc;
    if (!gamma || !beta || !mean || !rstd) return SD_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32) return sd::ln_fwd_launch<float>(x, nullptr, nullptr, 1, nullptr, gamma, beta, y, mean, rstd, rows, C, eps, st);
    return sd::ln_fwd_launch<sd::bf16_t>(x, nullptr, nullptr, 1, nullptr, gamma, beta, y, mean, rstd, rows, C, eps, st);
}

int sd_add_layernorm_fwd(const void *x, const void *res, const float *row_scale, long rows_per_sample, void *xsum, const float *gamma,
                         const float *beta, void *y, float *mean, float *rstd, int dtype, long rows, int C, float eps, void *stream) {
    int rc = sd::check_ln(x, y, dtype, rows, C);
    if (rc) return rc;
    rc = sd::check_ln(res, xsum, dtype, rows, C);
    if (rc) return rc;
    if (!gamma || !beta || !mean || !rstd) return SD_E_NULL;
    if (row_scale && (rows_per_sample <= 0 || rows % rows_per_sample)) return SD_E_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32)
        return sd::ln_fwd_launch<float>(x, res, row_scale, rows_per_sample, xsum, gamma, beta, y, mean, rstd, rows, C, eps, st);
    return sd::ln_fwd_launch<sd::bf16_t>(x, res, row_scale, rows_per_sample, xsum, gamma, beta, y, mean, rstd, rows, C, eps, st);
}

int sd_layernorm_map_fwd(const void *x, const void *res, void *xsum, const float *gamma, const float *beta, void *y, const int32_t *x_map,
                         const int32_t *res_map, int dtype, long images, int rows_out, int rows_in, int rows_res, int C, float eps, void *stream) {
    if (images <= 0 || rows_out <= 0 || rows_in <= 0) return SD_E_SHAPE;
    int rc = sd::check_ln(x, y, dtype, images * rows_out, C);
    if (rc) return rc;
    if (!gamma || !beta) return SD_E_NULL;
    if (res) {
        if (!xsum) return SD_E_NULL;
        if (rows_res <= 0) return SD_E_SHAPE;
        if ((reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(xsum)) & 15) return SD_E_ALIGN;
    }
    if (!x_map && rows_out != rows_in) return SD_E_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32)
        return sd::ln_map_launch<float>(x, res, xsum, gamma, beta, y, x_map, res_map, images * rows_out, rows_out, rows_in, rows_res, C, eps, st);
    return sd::ln_map_launch<sd::bf16_t>(x, res, xsum, gamma, beta, y, x_map, res_map, images * rows_out, rows_out, rows_in, rows_res, C, eps, st);
}

int sd_layernorm_bwd(const void *x, const void *dy, const float *gamma, const float *mean, const float *rstd, void *dx, float *dgamma,
                     float *dbeta, int dtype, long rows, int C, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = sd::check_ln(x, dy, dtype, rows, C);
    if (rc) return rc;
    if (!gamma || !mean || !rstd || !dx || !workspace || (!dgamma != !dbeta)) return SD_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32)
        return sd::ln_bwd_launch<float>(x, dy, gamma, mean, rstd, nullptr, nullptr, 1, dx, nullptr, dgamma, dbeta, workspace, workspace_bytes,
                                        rows, C, st);
    return sd::ln_bwd_launch<sd::bf16_t>(x, dy, gamma, mean, rstd, nullptr, nullptr, 1, dx, nullptr, dgamma, dbeta, workspace, workspace_bytes,
                                         rows, C, st);
}

int sd_add_layernorm_bwd(const void *xsum, const void *dy, const float *gamma, const float *mean, const float *rstd, const void *dres,
                         const float *row_scale, long rows_per_sample, void *dx, void *dr, float *dgamma, float *dbeta, int dtype, long rows,
                         int C, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = sd::check_ln(xsum, dy, dtype, rows, C);
    if (rc) return rc;
    if (!gamma || !mean || !rstd || !dx || !workspace || (!dgamma != !dbeta)) return SD_E_NULL;
    if ((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(dres) | reinterpret_cast<uintptr_t>(dr)) & 15) return SD_E_ALIGN;
    if (row_scale && (rows_per_sample <= 0 || rows % rows_per_sample)) return SD_E_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32)
        return sd::ln_bwd_launch<float>(xsum, dy, gamma, mean, rstd, dres, row_scale, rows_per_sample, dx, dr, dgamma, dbeta, workspace,
                                        workspace_bytes, rows, C, st);
    return sd::ln_bwd_launch<sd::bf16_t>(xsum, dy, gamma, mean, rstd, dres, row_scale, rows_per_sample, dx, dr, dgamma, dbeta, workspace,
                                         workspace_bytes, rows, C, st);
}

}  // extern "C"
