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
// vector belong to N different rows).  A workgroup is 32 vector positions x 8 chunk lanes; the chunk states of every channel are folded
// through LDS by the thread that owns the channel (all threads busy) and leave as ONE 20-byte record per (image, chunk block, channel),
// channel-contiguous.  One finish launch (a wave per row, fp64) turns the records of any number of stages into row statistics and losses.
// History, config 5, us per stage: round 2's first version merged inside the workgroup with 7 SERIAL merges on 1/r of the lanes and one
// wave per row over every chunk: 117/45/34/34; rounds 2-3: un-merged chunk-major records + merge + rows + loss launches: 115/51/27/19 with
// the scan itself 83/29/14/8.5 -- 31 MB of records next to stage 2's 100 MB of operands, four chained launches; round 4: this file.
// Backward: elementwise, one step of U pixels per lane (stage 2-4 launches were 1 wave per SIMD with 64-pixel lanes: 65/53/48 us -> 26/8/5).
// One rescale per U elements and channel: (2 + 2U)/U exponentials per element; the forward is VALU-paced (~17 instructions per element
// pair), ~76 % of the measured HBM ceiling at stage 1.  Algorithmic bytes: forward 2*N*e, backward 3*N*e.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cgd_tok_device.h"

namespace sd {
namespace {

constexpr int kTokU = 4;          // forward: pixels per lane and step (independent 16-byte loads per operand in flight; one rescale per U elements)
template <typename T> constexpr int tok_u() { return VecIO<T>::N == 8 ? 2 : kTokU; }   // backward: pixels per lane (one step, no loop state)
constexpr int kMaxPermC = 2048;   // inverse-permutation table in LDS

struct TokGeo {
    int N, vpp, VS, r, nvb, pix_chunk, nchunk, nkb, threads;
};

constexpr int kTokFwdVS = 32;     // forward: vector positions per workgroup; its 256 / VS chunk lanes are merged inside the workgroup

template <typename T>
TokGeo tok_geometry(int B, int C, long P, bool fwd) {
    TokGeo q;
    q.N = VecIO<T>::N;
    q.vpp = C / q.N;                                   // vectors per pixel
    if (fwd) {
        // a workgroup = VS vector positions x r chunks; the r chunk states of every channel are merged through LDS before they leave the
        // workgroup (transposed: thread m folds channel m's r states), so the partial array holds nkb = ceil(nchunk / r) records per
        // channel instead of nchunk -- at config 5's stage 2 the un-merged 16-pixel-chunk records were 31 MB next to 100 MB of operands
        q.VS = q.vpp < kTokFwdVS ? q.vpp : kTokFwdVS;
        q.nvb = (q.vpp + q.VS - 1) / q.VS;
        q.r = 256 / q.VS;
        q.threads = 256;
        // pixels per lane: 64 when that still makes >= 2048 waves, down to 16 otherwise (one chunk = one lane's run; measured per stage of
        // config 5: 64 / 16 / 16 / 16 are the fastest of 8..128)
        int L = 64;
        const long lanes = (long)B * q.vpp * P;
        while (L > 16 && lanes / (64L * L) < 2048) L /= 2;
        q.pix_chunk = L;
        q.nchunk = (int)((P + L - 1) / L);
        q.nkb = (q.nchunk + q.r - 1) / q.r;
    } else {
        q.VS = q.vpp < 256 ? q.vpp : 256;              // vector positions per workgroup
        q.nvb = (q.vpp + q.VS - 1) / q.VS;
        q.r = 256 / q.VS < 1 ? 1 : 256 / q.VS;         // lanes per vector position, pixel-interleaved in one chunk
        q.threads = (q.VS * q.r + 63) / 64 * 64;
        // one step per lane: the backward has no per-chunk state beyond 2 N row constants (cached), and short lanes fill the small stages
        q.pix_chunk = q.r * tok_u<T>();
        q.nchunk = (int)((P + q.pix_chunk - 1) / q.pix_chunk);
        q.nkb = q.nchunk;
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

// ---- job tables, BY VALUE in the kernel arguments (reduce.hip's pattern: scalar loads, nothing to copy, safe under graph capture) --------
// Round 4: the forward was FOUR chained launches per stage (scan -> merge -> rows -> loss), the last three ~4.5 us of
// pure latency each, and config 5 has four stages: 16 launches.  Now any number of stages (<= kTokMaxJobs) is ONE scan launch for the
// long-chunk stages, ONE for the 16-pixel-chunk stages, and ONE finish launch that folds every row's chunks in fp64 straight from the
// chunk-major partials and ends with the loss, computed by the workgroup whose arrival ticket is the last of its stage.
struct TokScanTable {
    const void *S[kTokMaxJobs];
    const void *T[kTokMaxJobs];
    RowPart *part[kTokMaxJobs];
    long P[kTokMaxJobs];
    int C[kTokMaxJobs], VS[kTokMaxJobs], r[kTokMaxJobs], nvb[kTokMaxJobs], pix_chunk[kTokMaxJobs], nchunk[kTokMaxJobs], nkb[kTokMaxJobs], bpi[kTokMaxJobs];
    float c2[kTokMaxJobs];
    int blk_begin[kTokMaxJobs + 1];
    int njobs;
};

// One workgroup: ceil(nchunk / r) * nvb of them per image; VS vector positions x r chunk lanes, each lane one chunk of pix_chunk pixels of
// its 16-byte channel vector.  The lane states go through LDS and thread m folds the r states of the workgroup's channel m in chunk order
// (every thread busy; stride-5 reads are conflict-free), so ONE 20-byte record per (image, chunk block kb, channel) leaves the workgroup,
// channel-contiguous: part[(b*nkb + kb)*C + c] (the finish kernel applies the shuffle).
// PRE (pix_chunk == 16, i.e. every stage below ~2048 waves of 64-pixel lanes): all 16 pixels of the chunk are requested before the first
// fold.  These launches are 1-3 waves per SIMD, so nothing else covers the load latency: four dependent rounds of (8 loads -> fold) ran at
// four memory latencies per lane.  Same folds in the same order as the stepped form.
template <typename T, bool PRE>
__global__ __launch_bounds__(256) void cgd_tok_fwd_partials(const TokScanTable tab, unsigned *__restrict__ counters, int ncounters) {
    constexpr int N = VecIO<T>::N, U = kTokU;
    __shared__ __attribute__((aligned(16))) float xch[256 * 5 * N];
    if (counters && blockIdx.x == 0 && threadIdx.x < (unsigned)ncounters) counters[threadIdx.x] = 0u;   // arrival tickets of the finish launch
    const int j = tok_find_job(tab, (int)blockIdx.x);
    const int lb = (int)blockIdx.x - tab.blk_begin[j];
    const int bpi = tab.bpi[j], nvb = tab.nvb[j], VS = tab.VS[j], r = tab.r[j], C = tab.C[j], pix_chunk = tab.pix_chunk[j], nchunk = tab.nchunk[j];
    const long P = tab.P[j];
    const float c2 = tab.c2[j];
    const int b = lb / bpi, rest = lb - b * bpi;
    const int kb = rest / nvb, vb = rest - kb * nvb;
    const int t = threadIdx.x;
    const int vl = t % VS, pr = t / VS;
    const int v = vb * VS + vl;
    const int k = kb * r + pr;
    const bool active = pr < r && v * N < C && k < nchunk;
    RowPart st[N];
#pragma unroll
    for (int i = 0; i < N; ++i) st[i] = {kNegBig, 0.f, kNegBig, 0.f, 0.f};
    if (active) {
        const long p_lo = (long)k * pix_chunk, p_hi = min(P, p_lo + pix_chunk);
        const T *ps = static_cast<const T *>(tab.S[j]) + ((size_t)b * P) * C + (size_t)v * N;
        const T *pt = static_cast<const T *>(tab.T[j]) + ((size_t)b * P) * C + (size_t)v * N;
        typedef typename RawIO<T>::raw_t raw_t;
        if constexpr (PRE) {
            constexpr int L = 4 * U;                                     // the launcher sends only pix_chunk == 16 here
            raw_t s[L], tt[L];
#pragma unroll
            for (int u = 0; u < L; ++u) {
                const long p = p_lo + u;
                const long pc = p < p_hi ? p : p_hi - 1;                 // address clamped into the chunk, value masked below
                s[u] = RawIO<T>::load(ps + (size_t)pc * C);
                tt[u] = RawIO<T>::load(pt + (size_t)pc * C);
            }
#pragma unroll
            for (int q = 0; q < L; q += U) {
                if (p_lo + q < p_hi) {                                   // the stepped form's loop condition: same folds, same order
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        float sv[U], tv[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const bool in = p_lo + q + u < p_hi;
                            sv[u] = in ? RawIO<T>::elem(s[q + u], i) : kNegBig;
                            tv[u] = in ? RawIO<T>::elem(tt[q + u], i) : kNegBig;
                        }
                        fold_channel<U>(st[i], sv, tv, c2);
                    }
                }
            }
        } else {
            // U raw vectors per operand in flight per lane; no cross-step prefetch (measured: 82.5 us vs 85.4 with it at stage 1 of config 5,
            // and 106 instead of 157 VGPRs) -- the other waves of the SIMD cover the latency, the loop is VALU-paced (~17 instructions per
            // element pair)
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
        }
    }
    // lane states -> LDS as [pr][vl][i][5] = record (pr * VS * N + channel-in-block) * 5; idle lanes publish the identity state
    typedef float f4 __attribute__((ext_vector_type(4)));
    float flat[5 * N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        flat[5 * i] = st[i].ms; flat[5 * i + 1] = st[i].zs; flat[5 * i + 2] = st[i].mt; flat[5 * i + 3] = st[i].zt; flat[5 * i + 4] = st[i].a;
    }
    f4 *mine = reinterpret_cast<f4 *>(xch + (size_t)t * 5 * N);
#pragma unroll
    for (int q = 0; q < 5 * N / 4; ++q) mine[q] = f4{flat[4 * q], flat[4 * q + 1], flat[4 * q + 2], flat[4 * q + 3]};
    __syncthreads();
    const int CB = VS * N, c = vb * CB + t;
    if (t < CB && c < C) {
        const int r_eff = min(r, nchunk - kb * r);                       // chunk lanes of this block that exist
        const float *q0 = xch + (size_t)t * 5;
        RowPart acc = {q0[0], q0[1], q0[2], q0[3], q0[4]};
        for (int p2 = 1; p2 < r_eff; ++p2) {
            const float *q = xch + ((size_t)p2 * CB + t) * 5;
            merge(acc, RowPart{q[0], q[1], q[2], q[3], q[4]}, c2);
        }
        tab.part[j][((size_t)b * tab.nkb[j] + kb) * C + c] = acc;
    }
}

// ---- finish: rows + loss of every stage in ONE launch ------------------------------------------------------------------------------
// One wave per row (b, jg): its partials are (chunk block k, slot s) for k < nkb, s in [jg*g, min(C, jg*g+g)) -- lane i takes (k, s) =
// (i / gw, i % gw), so consecutive lanes read the consecutive 20-byte records of one chunk block (a permutation scatters them: 1 iteration
// in 1000).  Each lane folds its records in walk order with fp64 sums (four requests in flight), the wave combines in fp64 (the closed
// form of cgd_fwd_rows).  The loss of a stage is written by the workgroup that draws the last arrival ticket of that stage, summing row_kl
// in the fixed order of cgd_fwd_loss: run-to-run identical.  Hand-off per cdna_hip_programming.md Guideline 16 (counter form, write-through
// payload): row_kl is stored `sc1` (a relaxed agent-scope atomic store: no dirty L2 line, so no release fence -- 1.7 us per workgroup
// saved), every storing wave drains, barrier, ONE lane draws the relaxed agent ticket; the last arriver acquires (agent), waits, barrier,
// then loads row_kl.  The tickets are zeroed by the first scan launch of the call (a kernel boundary earlier).
__global__ __launch_bounds__(256) void cgd_tok_finish(const TokFinTable tab, unsigned *counters) {
    __shared__ int is_last;
    __shared__ double acc[4];
    const int j = tok_find_job(tab, (int)blockIdx.x);
    const int lb = (int)blockIdx.x - tab.blk_begin[j];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int C = tab.C[j], g = tab.g[j], G = tab.G[j], nkb = tab.nkb[j];
    const int rows = tab.B[j] * G;
    const float c2 = tab.c2[j];
    float *row_kl = tab.row_kl[j];                       // re-read below by the last arriver: plain vector loads behind the acquire
    const int row = lb * 4 + w;
    if (row < rows) {                                    // wave-uniform
        const int b = row / G, jg = row - b * G;
        const int c_lo = jg * g, gw = min(C, c_lo + g) - c_lo;
        const int n = gw * nkb;
        const RowPart *pb = tab.part[j] + (size_t)b * nkb * C;
        const int32_t *perm = tab.perm[j];
        float ms = kNegBig, mt = kNegBig;
        double zs = 0, zt = 0, a = 0;
        auto fold1 = [&](const RowPart &q) {
            const float nms = fmaxf(ms, q.ms), nmt = fmaxf(mt, q.mt);
            const float rs = ex2((ms - nms) * c2), qs = ex2((q.ms - nms) * c2);
            const float rt = ex2((mt - nmt) * c2), qt = ex2((q.mt - nmt) * c2);
            zs = zs * (double)rs + (double)q.zs * (double)qs;
            zt = zt * (double)rt + (double)q.zt * (double)qt;
            a = a * (double)rt + (double)q.a * (double)qt;
            ms = nms; mt = nmt;
        };
        auto at = [&](int i) -> const RowPart * {
            const int k = i / gw, s = c_lo + (i - k * gw);
            return pb + (size_t)k * C + (perm ? perm[s] : s);
        };
        int i = lane;
        for (; i + 192 < n; i += 256) {
            const RowPart q0 = *at(i), q1 = *at(i + 64), q2 = *at(i + 128), q3 = *at(i + 192);
            fold1(q0); fold1(q1); fold1(q2); fold1(q3);
        }
        for (; i < n; i += 64) fold1(*at(i));
        const float wms = wave_max(ms), wmt = wave_max(mt);
        zs *= exp2((double)(ms - wms) * (double)c2);
        const double ft = exp2((double)(mt - wmt) * (double)c2);
        zt *= ft;
        a *= ft;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            zs += __shfl_xor(zs, o, 64);
            zt += __shfl_xor(zt, o, 64);
            a += __shfl_xor(a, o, 64);
        }
        if (lane == 0) {
            const double l2s = (double)wms * c2 + log2(zs), l2t = (double)wmt * c2 + log2(zt);
            const double ln2 = 0.69314718055994530942;
            tab.row_lse2[j][2 * row] = (float)l2s;
            tab.row_lse2[j][2 * row + 1] = (float)l2t;
            __hip_atomic_store(row_kl + row, (float)(a * (double)tab.inv_tau[j] / zt + (l2s - l2t) * ln2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // EVERY storing wave drains its stores before the barrier
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned nblk = (unsigned)(tab.blk_begin[j + 1] - tab.blk_begin[j]);
        const unsigned ticket = __hip_atomic_fetch_add(counters + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = ticket == nblk - 1u;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        is_last = last;
    }
    __syncthreads();
    if (is_last) {                                       // workgroup-uniform
        double v = 0;
        for (int i = threadIdx.x; i < rows; i += 256) v += (double)row_kl[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) acc[w] = v;
        __syncthreads();
        if (threadIdx.x == 0) tab.loss[j][0] = (float)(((acc[0] + acc[1]) + acc[2] + acc[3]) * (double)tab.loss_scale[j]);
    }
}

// dS = kk (2^{s c2 - lse2_s(row)} - 2^{t c2 - lse2_t(row)}), row = b*G + slot(c)/g; any number of stages in one launch
struct TokBwdTable {
    const void *S[kTokMaxJobs];
    const void *T[kTokMaxJobs];
    const int32_t *perm[kTokMaxJobs];
    const float *row_lse2[kTokMaxJobs];
    const float *upstream[kTokMaxJobs];
    void *dS[kTokMaxJobs];
    long P[kTokMaxJobs];
    int C[kTokMaxJobs], g[kTokMaxJobs], G[kTokMaxJobs], VS[kTokMaxJobs], r[kTokMaxJobs], nvb[kTokMaxJobs], pix_chunk[kTokMaxJobs], bpi[kTokMaxJobs];
    float c2[kTokMaxJobs], coef[kTokMaxJobs];
    int blk_begin[kTokMaxJobs + 1];
    int njobs;
};

template <typename T, bool NT>
__global__ __launch_bounds__(256) void cgd_tok_bwd(const TokBwdTable tab) {
    constexpr int N = VecIO<T>::N, U = tok_u<T>();
    extern __shared__ __attribute__((aligned(16))) unsigned char tok_smem[];   // int inv[C] when perm
    int *inv = reinterpret_cast<int *>(tok_smem);
    const int j = tok_find_job(tab, (int)blockIdx.x);
    const int lb = (int)blockIdx.x - tab.blk_begin[j];
    const int bpi = tab.bpi[j], nvb = tab.nvb[j], VS = tab.VS[j], r = tab.r[j], C = tab.C[j], g = tab.g[j], G = tab.G[j], pix_chunk = tab.pix_chunk[j];
    const long P = tab.P[j];
    const float c2 = tab.c2[j];
    const int32_t *perm = tab.perm[j];
    const float *row_lse2 = tab.row_lse2[j], *upstream = tab.upstream[j];
    const T *S = static_cast<const T *>(tab.S[j]), *Tt = static_cast<const T *>(tab.T[j]);
    T *dS = static_cast<T *>(tab.dS[j]);
    const int b = lb / bpi, rest = lb - b * bpi;
    const int k = rest / nvb, vb = rest - k * nvb;
    const int t = threadIdx.x;
    const int vl = t % VS, pr = t / VS;
    const int v = vb * VS + vl;
    const bool lane_ok = pr < r && v * N < C;
    if (perm) {
        for (int i = t; i < C; i += blockDim.x) inv[perm[i]] = i;
        __syncthreads();
    }
    if (!lane_ok) return;
    const float kk = upstream ? tab.coef[j] * upstream[0] : tab.coef[j];
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
int tok_fwd_multi(const sd_cgd_tok_fwd_job *jobs, int njobs, hipStream_t st) {
    TokScanTable scan[2] = {};         // [0]: stepped form (long chunks), [1]: PRE form (16-pixel chunks)
    TokFinTable fin = {};
    int nb[2] = {0, 0};
    int fb = 0;
    for (int i = 0; i < njobs; ++i) {
        const sd_cgd_tok_fwd_job &q = jobs[i];
        const TokGeo geo = tok_geometry<T>(q.B, q.C, q.P, true);
        const size_t need = tok_part_bytes(q.B, q.C, geo.nkb) + (i == 0 ? kTokTicketBytes : 0);
        if (q.workspace_bytes < need || (reinterpret_cast<uintptr_t>(q.workspace) & 15)) return SD_E_WORKSPACE;
        const float c2 = q.inv_tau * 1.44269504088896340736f;
        RowPart *part = static_cast<RowPart *>(q.workspace);
        TokScanTable &t = scan[geo.pix_chunk == 4 * kTokU ? 1 : 0];
        const int e = t.njobs++;
        t.S[e] = q.S; t.T[e] = q.T; t.part[e] = part; t.P[e] = q.P; t.C[e] = q.C; t.VS[e] = geo.VS; t.r[e] = geo.r; t.nvb[e] = geo.nvb;
        t.pix_chunk[e] = geo.pix_chunk; t.nchunk[e] = geo.nchunk; t.nkb[e] = geo.nkb; t.c2[e] = c2;
        t.bpi[e] = geo.nkb * geo.nvb;
        int &n = nb[&t == &scan[1] ? 1 : 0];
        t.blk_begin[e] = n;
        n += t.bpi[e] * q.B;
        t.blk_begin[e + 1] = n;
        const int G = (q.C + q.g - 1) / q.g;
        fin.part[i] = part; fin.perm[i] = q.perm; fin.row_lse2[i] = q.row_lse2; fin.row_kl[i] = q.row_kl; fin.loss[i] = q.loss;
        fin.B[i] = q.B; fin.C[i] = q.C; fin.g[i] = q.g; fin.G[i] = G; fin.nkb[i] = geo.nkb; fin.c2[i] = c2; fin.inv_tau[i] = q.inv_tau;
        fin.loss_scale[i] = q.loss_scale;
        fin.blk_begin[i] = fb;
        fb += (q.B * G + 3) / 4;
        fin.blk_begin[i + 1] = fb;
    }
    fin.njobs = njobs;
    const sd_cgd_tok_fwd_job &q0 = jobs[0];
    unsigned *tickets = reinterpret_cast<unsigned *>(static_cast<unsigned char *>(q0.workspace) +
                                                     tok_part_bytes(q0.B, q0.C, tok_geometry<T>(q0.B, q0.C, q0.P, true).nkb));
    bool zeroed = false;
    if (nb[0]) {
        hipLaunchKernelGGL((cgd_tok_fwd_partials<T, false>), dim3((unsigned)nb[0]), dim3(256), 0, st, scan[0], tickets, kTokMaxJobs);
        zeroed = true;
    }
    if (nb[1])
        hipLaunchKernelGGL((cgd_tok_fwd_partials<T, true>), dim3((unsigned)nb[1]), dim3(256), 0, st, scan[1], zeroed ? (unsigned *)nullptr : tickets,
                           kTokMaxJobs);
    return tok_finish_launch(fin, tickets, st);
}

template <typename T>
int tok_bwd_multi(const sd_cgd_tok_bwd_job *jobs, int njobs, hipStream_t st) {
    TokBwdTable t = {};
    int nblk = 0;
    size_t lds = 0;
    for (int i = 0; i < njobs; ++i) {
        const sd_cgd_tok_bwd_job &q = jobs[i];
        const TokGeo geo = tok_geometry<T>(q.B, q.C, q.P, false);
        t.S[i] = q.S; t.T[i] = q.T; t.perm[i] = q.perm; t.row_lse2[i] = q.row_lse2; t.upstream[i] = q.upstream; t.dS[i] = q.dS; t.P[i] = q.P;
        t.C[i] = q.C; t.g[i] = q.g; t.G[i] = (q.C + q.g - 1) / q.g; t.VS[i] = geo.VS; t.r[i] = geo.r; t.nvb[i] = geo.nvb; t.pix_chunk[i] = geo.pix_chunk;
        t.bpi[i] = geo.nchunk * geo.nvb;
        t.c2[i] = q.inv_tau * 1.44269504088896340736f;
        t.coef[i] = q.coef;
        t.blk_begin[i] = nblk;
        nblk += t.bpi[i] * q.B;
        t.blk_begin[i + 1] = nblk;
        if (q.perm && (size_t)q.C * sizeof(int) > lds) lds = (size_t)q.C * sizeof(int);
    }
    t.njobs = njobs;
    hipLaunchKernelGGL((cgd_tok_bwd<T, true>), dim3((unsigned)nblk), dim3(256), lds, st, t);
    return (int)hipGetLastError();
}

int check_fwd_jobs(const sd_cgd_tok_fwd_job *jobs, int njobs, int dtype) {
    if (!jobs) return SD_E_NULL;
    if (njobs <= 0 || njobs > kTokMaxJobs) return SD_E_SHAPE;
    for (int i = 0; i < njobs; ++i) {
        const sd_cgd_tok_fwd_job &q = jobs[i];
        int rc = check_tok(q.S, q.T, dtype, q.B, q.C, q.P, q.g, q.perm);
        if (rc) return rc;
        if (!q.row_lse2 || !q.row_kl || !q.loss || !q.workspace) return SD_E_NULL;
    }
    return SD_OK;
}

int check_bwd_jobs(const sd_cgd_tok_bwd_job *jobs, int njobs, int dtype) {
    if (!jobs) return SD_E_NULL;
    if (njobs <= 0 || njobs > kTokMaxJobs) return SD_E_SHAPE;
    for (int i = 0; i < njobs; ++i) {
        const sd_cgd_tok_bwd_job &q = jobs[i];
        int rc = check_tok(q.S, q.T, dtype, q.B, q.C, q.P, q.g, q.perm);
        if (rc) return rc;
        if (!q.row_lse2 || !q.dS) return SD_E_NULL;
        if (reinterpret_cast<uintptr_t>(q.dS) & 15) return SD_E_ALIGN;
    }
    return SD_OK;
}

}  // namespace

int tok_finish_launch(const TokFinTable &fin, unsigned *tickets, hipStream_t st) {
    hipLaunchKernelGGL(cgd_tok_finish, dim3((unsigned)fin.blk_begin[fin.njobs]), dim3(256), 0, st, fin, tickets);
    return (int)hipGetLastError();
}

}  // namespace sd

extern "C" {

size_t sd_cgd_kl_tok_workspace_bytes(int B, int C, long P) {
    if (B <= 0 || C <= 0 || P <= 0) return 0;
    const sd::TokGeo a = sd::tok_geometry<float>(B, C, P, true), b = sd::tok_geometry<sd::bf16_t>(B, C, P, true);
    const int nkb = a.nkb > b.nkb ? a.nkb : b.nkb;                     // enough for either storage type
    return sd::tok_part_bytes(B, C, nkb) + sd::kTokTicketBytes;        // merged partials + the finish launch's arrival tickets
}

int sd_cgd_kl_tok_max_jobs(void) { return sd::kTokMaxJobs; }

int sd_cgd_kl_tok_fwd_multi(const sd_cgd_tok_fwd_job *jobs, int njobs, int dtype, void *stream) {
    int rc = sd::check_fwd_jobs(jobs, njobs, dtype);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return dtype == SD_F32 ? sd::tok_fwd_multi<float>(jobs, njobs, st) : sd::tok_fwd_multi<sd::bf16_t>(jobs, njobs, st);
}

int sd_cgd_kl_tok_bwd_multi(const sd_cgd_tok_bwd_job *jobs, int njobs, int dtype, void *stream) {
    int rc = sd::check_bwd_jobs(jobs, njobs, dtype);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return dtype == SD_F32 ? sd::tok_bwd_multi<float>(jobs, njobs, st) : sd::tok_bwd_multi<sd::bf16_t>(jobs, njobs, st);
}

int sd_cgd_kl_tok_fwd(const void *S, const void *T, int dtype, int B, int C, long P, int g, float inv_tau, float loss_scale, const int32_t *perm,
                      float *row_lse2, float *row_kl, float *loss, void *workspace, size_t workspace_bytes, void *stream) {
    const sd_cgd_tok_fwd_job job = {S, T, perm, row_lse2, row_kl, loss, workspace, workspace_bytes, P, B, C, g, inv_tau, loss_scale, 0};
    return sd_cgd_kl_tok_fwd_multi(&job, 1, dtype, stream);
}

int sd_cgd_kl_tok_bwd(const void *S, const void *T, int dtype, int B, int C, long P, int g, float inv_tau, float coef, const int32_t *perm,
                      const float *row_lse2, const float *upstream, void *dS, void *stream) {
    const sd_cgd_tok_bwd_job job = {S, T, perm, row_lse2, upstream, dS, P, B, C, g, inv_tau, coef, 0};
    return sd_cgd_kl_tok_bwd_multi(&job, 1, dtype, stream);
}

}  // extern "C"
