// sra_attn.hip -- MiT "spatial-reduction" attention, forward and backward, gfx950.
//
// Reference mix_transformer.py:107-133: per head  O = softmax(scale * Q K^T) V  with N queries (16384 ... 256 tokens)
// but only KV = (H/32)*(W/32) keys (256 at 512x512) and head_dim 32 or 64.  With so few keys the op is nothing like
// LLM attention: the whole K and V of a head fit in LDS (64-128 KB of the CU's 160 KB), every query needs all of them, and
// nothing has to be tiled or re-scaled across key blocks.  ATen runs it either as bmm -> scale -> softmax -> bmm (a
// [B,heads,N,KV] score tensor, 134 MB per stage-1 layer, written and re-read five times fwd+bwd) or as a generic flash
// kernel whose backward parallelises over the 256 keys only.  Three kernel families share the design below:
//   sra_fwd / sra_bwd_dq / sra_bwd_dkv           exact f32-input MFMA (v_mfma_f32_32x32x2_f32) -- round 1; behind tunable sra_split_bf16 = 0,
//                                                 and the head_dim-64 backward of fp32 storage
//   sra_fwd_x3 / sra_bwd_dq_x3 / sra_bwd_dkv_x3  fp32 storage, split-bf16 products on the bf16 matrix pipe (fp32-grade; the shipped path)
//   sra_fwd_b16 / sra_bwd_dq_b16 / sra_bwd_dkv_b16  bf16 storage on the bf16 matrix pipe (config 5)
// The exact-f32 form:
//   * a wave owns 32 queries and computes the TRANSPOSED score tile S^T = K Q^T (keys x queries): in the MFMA C layout a
//     lane then holds one query column, so the softmax needs no cross-lane work except one exchange with lane^32, and the
//     probabilities are already in the B-operand layout of the second product O^T = V^T P^T -- no LDS round trip for P;
//   * K and V are staged in LDS once per workgroup (128 queries); q / dO rows stay in registers;
//   * backward = two kernels: dq (same S^T form, one key block at a time: S^T, dP^T, dS^T, dQ^T += K^T dS^T) and dk/dv
//     (S form: each wave keeps two 32-key column blocks of K and V in registers, query tiles stream through LDS; chunk
//     partials are combined by a deterministic second pass).
// q [B,N,heads*D] and kv [B,KV,2*heads*D] are the q / kv Linear outputs as they are (no head transpose copies); the output
// is [B,N,heads*D], what the proj Linear consumes.  Exponentials in base 2 (scale*log2e folded into q / k).
// (A first version with one thread per query and K/V rows fed through the scalar cache as SGPR operands of v_pk_fma_f32
// was limited by scalar-load latency x the ~100 SGPRs of a wave: 110-130 us per stage-1 layer, slower than the library.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "cgd_device.h"

namespace sd {
namespace {

constexpr int kSraThreads = 256;
constexpr int kSraKeys = 256;        // keys held in LDS (8 blocks of 32)
constexpr float kLog2e = 1.4426950408889634f;

typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
// MFMA 32x32x2 layouts (lane l): A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31]; C element e: row crow(e, l >> 5), col l & 31
__device__ __forceinline__ int crow(int e, int half) { return (e & 3) + 8 * (e >> 2) + 4 * half; }

// 4 consecutive elements
template <typename T> struct Q4;
template <> struct Q4<float> {
    static __device__ __forceinline__ float4 load(const float *p) { return *reinterpret_cast<const float4 *>(p); }
    static __device__ __forceinline__ void store(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
};
template <> struct Q4<bf16_t> {
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
};

// n consecutive elements (n % 4 == 0) of a row into registers
template <typename T, int NE> __device__ __forceinline__ void load_span(const T *p, float (&o)[NE]) {
#pragma unroll
    for (int i = 0; i < NE; i += 4) {
        const float4 v = Q4<T>::load(p + i);
        o[i] = v.x; o[i + 1] = v.y; o[i + 2] = v.z; o[i + 3] = v.w;
    }
}

// K and V of head h of image b -> LDS [rows][PD] (rows >= KV zero-filled up to `rows`)
template <typename T, int D, int PD>
__device__ __forceinline__ void stage_kv(const T *__restrict__ kv, float *Ks, float *Vs, int b, int h, int KV, int heads, int rows) {
    const int C = heads * D;
    constexpr int G = D / 4, U = 4;     // U row-vectors of K and of V in flight per thread: the copy is latency-bound otherwise
    const int total = rows * G, step = blockDim.x;
    for (int base = threadIdx.x; base < total; base += U * step) {
        float4 k4[U], v4[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * step;
            const int j = idx / G, c4 = (idx % G) * 4;
            k4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            v4[u] = k4[u];
            if (idx < total && j < KV) {
                const T *row = kv + ((size_t)b * KV + j) * 2 * C + h * D + c4;
                k4[u] = Q4<T>::load(row);
                v4[u] = Q4<T>::load(row + C);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * step;
            if (idx < total) {
                const int j = idx / G, c4 = (idx % G) * 4;
                *reinterpret_cast<float4 *>(Ks + j * PD + c4) = k4[u];
                *reinterpret_cast<float4 *>(Vs + j * PD + c4) = v4[u];
            }
        }
    }
}

// one 32x32 tile of the transposed product  (rows = the 32 LDS rows starting at row0, cols = this wave's 32 register rows):
// acc += M[row0 + i][:] . r[:]  with the reduction axis split as (half, t): lane l reads M[row0 + (l&31)][half*D/2 + t]
template <int D, int PD> __device__ __forceinline__ f32x16 tile_t(const float *M, int row0, const float (&r)[D / 2], int c, int half) {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float *row = M + (row0 + c) * PD + half * (D / 2);
#pragma unroll
    for (int t = 0; t < D / 2; t += 4) {
        const float4 a = *reinterpret_cast<const float4 *>(row + t);
        acc = mfma(a.x, r[t], acc);
        acc = mfma(a.y, r[t + 1], acc);
        acc = mfma(a.z, r[t + 2], acc);
        acc = mfma(a.w, r[t + 3], acc);
    }
    return acc;
}

// ---- forward: grid (ceil(N/(32*NW)), heads, B); a wave owns 32 queries, the NW waves of a workgroup share K/V in LDS ---------
// (NW = 8 puts two waves on every SIMD of the CU that holds the K/V copy: one wave's softmax overlaps the other's MFMAs)
template <typename T, int D, int NW, int QT>
__global__ __launch_bounds__(NW * 64) void sra_fwd(const T *__restrict__ q, const T *__restrict__ kv, T *__restrict__ out,
                                                    float *__restrict__ lse, int N, int KV, int heads, float cs /* scale*log2e */) {
    constexpr int PD = D + 4, DB = D / 32;
    extern __shared__ float smem[];
    const int nblk = (KV + 31) / 32;
    float *Ks = smem, *Vs = smem + nblk * 32 * PD;
    const int h = blockIdx.y, b = blockIdx.z, C = heads * D;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, c = l & 31, half = l >> 5;
    stage_kv<T, D, PD>(kv, Ks, Vs, b, h, KV, heads, nblk * 32);
    __syncthreads();
    // QT query tiles per wave: the staged K/V copy is amortised over QT * NW * 32 queries
    for (int qt = 0; qt < QT; ++qt) {
        const int n = (blockIdx.x * QT + qt) * (NW * 32) + w * 32 + c;
        const bool live = n < N;
        float qv[D / 2];
#pragma unroll
        for (int t = 0; t < D / 2; ++t) qv[t] = 0.f;
        if (live) load_span<T, D / 2>(q + ((size_t)b * N + n) * C + h * D + half * (D / 2), qv);
#pragma unroll
        for (int t = 0; t < D / 2; ++t) qv[t] *= cs;
        f32x16 S[8];
        float m = kNegBig;
#pragma unroll
        for (int blk = 0; blk < 8; ++blk) {
            if (blk < nblk) {
                S[blk] = tile_t<D, PD>(Ks, blk * 32, qv, c, half);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (blk * 32 + crow(e, half) >= KV) S[blk][e] = kNegBig;
                    m = fmaxf(m, S[blk][e]);
                }
            }
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float lsum = 0.f;
#pragma unroll
        for (int blk = 0; blk < 8; ++blk) {
            if (blk < nblk) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    S[blk][e] = ex2(S[blk][e] - m);
                    lsum += S[blk][e];
                }
            }
        }
        lsum += __shfl_xor(lsum, 32, 64);
        f32x16 O[DB];
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) O[db][e] = 0.f;
#pragma unroll
        for (int blk = 0; blk < 8; ++blk) {
            if (blk < nblk) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float *vrow = Vs + (blk * 32 + crow(e, half)) * PD + c;
#pragma unroll
                    for (int db = 0; db < DB; ++db) O[db] = mfma(vrow[db * 32], S[blk][e], O[db]);
                }
            }
        }
        if (live) {
            const float inv = 1.f / lsum;
            T *orow = out + ((size_t)b * N + n) * C + h * D;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    Q4<T>::store(orow + db * 32 + 8 * g + 4 * half,
                                 make_float4(O[db][4 * g] * inv, O[db][4 * g + 1] * inv, O[db][4 * g + 2] * inv, O[db][4 * g + 3] * inv));
            if (half == 0) lse[((size_t)b * heads + h) * N + n] = m + __builtin_amdgcn_logf(lsum);   // base-2 lse of the scaled scores
        }
    }
}

// ---- forward on the bf16 matrix pipe with fp32-grade arithmetic ("split-bf16") ------------------------------------------------
// Every fp32 operand x is split exactly into three bf16 planes x = hi + mid + lo (round-to-nearest residual splits) and a product is
// the six cross terms >= 2^-16 relative (mid.mid, lo.hi, hi.lo, mid.hi, hi.mid, hi.hi) on v_mfma_f32_32x32x16_bf16, accumulated in
// fp32, small terms first -- the dropped terms are <= 2^-24 relative, the rounding level of an fp32 fma chain itself (same scheme and
// same error bound as token_gemm.hip).  16 k per 32-cycle instruction x 6 = 12 cycles per k against 32 for v_mfma_f32_32x32x2_f32.
// Same ownership as sra_fwd (a wave owns 32 queries, S^T = K Q^T, P stays in registers between the two products), what changes:
//   * K is staged ONCE per workgroup as three bf16 planes [key][d] (row pitch 2 D + 16 bytes: conflict-free ds_read_b128), V as three
//     TRANSPOSED planes [d][key slot] whose key order inside a block of 32 is the order in which a lane of the 32x32 C layout holds the
//     rows of S^T (slot 16 s + 8 half + i <-> key (i & 3) + 8 (2 s + (i >> 2)) + 4 half): the 8 probabilities a lane owns for k-step s
//     ARE its B fragment, and the matching A fragment of V^T is one 16-byte LDS read;
//   * q is split once per tile in registers, P per k-step (8 values -> 3 x 4 VGPRs);
//   * head_dim 64: the six planes of 256 keys are 207 KB, more than the CU's 160 KB of LDS -- PHASED: the K planes are staged, S^T and
//     the softmax run, then the V planes replace them for the second product (two more barriers per tile, K / V re-read from L2).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int sra_slot(int r) { return 16 * (r >> 4) + 8 * ((r >> 2) & 1) + (r & 3) + 4 * ((r >> 3) & 1); }   // row of a 32-block -> k slot
__device__ __forceinline__ float bf16_lane(const u32x4 &v, int e) { return (e & 1) ? __uint_as_float(v[e >> 1] & 0xffff0000u) : __uint_as_float(v[e >> 1] << 16); }
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
// on PAIRS of values: one v_cvt_pk_bf16_f32 per pair and plane, packed residual subtractions (see token_gemm.hip::split8; same bits)
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 &h, bf16x8 &m, bf16x8 &l) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const f32x2 v = {x[e], x[e + 1]};
        const bf16x2 hh = __builtin_convertvector(v, bf16x2);
        const f32x2 r1 = v - __builtin_convertvector(hh, f32x2);
        const bf16x2 mm = __builtin_convertvector(r1, bf16x2);
        const f32x2 r2 = r1 - __builtin_convertvector(mm, f32x2);
        const bf16x2 ll = __builtin_convertvector(r2, bf16x2);
        h[e] = hh[0], h[e + 1] = hh[1], m[e] = mm[0], m[e + 1] = mm[1], l[e] = ll[0], l[e + 1] = ll[1];
    }
}
// acc += A . B over 16 k with A = ah + am + al, B = bh + bm + bl
__device__ __forceinline__ f32x16 mfma_x3(bf16x8 ah, bf16x8 am, bf16x8 al, bf16x8 bh, bf16x8 bm, bf16x8 bl, f32x16 c) {
    c = mfma16(am, bm, c);
    c = mfma16(al, bh, c);
    c = mfma16(ah, bl, c);
    c = mfma16(am, bh, c);
    c = mfma16(ah, bm, c);
    return mfma16(ah, bh, c);
}
template <int D> struct X3Geo {
    static constexpr int KP = 2 * D + 16;                                   // bytes per K-plane row
    static __host__ __device__ constexpr int vp(int rows) { return 2 * rows + 16; }   // bytes per V^T-plane row
    static __host__ __device__ constexpr size_t k_bytes(int rows) { return (size_t)3 * rows * KP; }
    static __host__ __device__ constexpr size_t v_bytes(int rows) { return (size_t)3 * D * vp(rows); }
};
// K of head h of image b -> three bf16 planes [rows][KP] (rows >= KV zero).  NG = groups of 8 values per thread at the full 256 keys:
// all their loads are issued before the first conversion (one memory latency per staging, not one per group).
template <typename T, int D, int NT>
__device__ __forceinline__ void stage_k_planes(const T *__restrict__ kv, unsigned char *Kimg, int b, int h, int KV, int heads, int rows,
                                               int col0 = 0 /* 0: K, heads*D: V */, float *f32_copy = nullptr /* [rows][D + 4] */) {
    constexpr int KP = X3Geo<D>::KP, G = D / 8, NG = (kSraKeys * G + NT - 1) / NT;
    const int C = heads * D;
    float x[NG][8];
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        const int idx = threadIdx.x + u * NT;
        const int j = idx / G, d0 = (idx % G) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) x[u][e] = 0.f;
        if (idx < rows * G && j < KV) load_span<T, 8>(kv + ((size_t)b * KV + j) * 2 * C + col0 + h * D + d0, x[u]);
    }
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        const int idx = threadIdx.x + u * NT;
        if (idx < rows * G) {
            const int j = idx / G, d0 = (idx % G) * 8;
            bf16x8 ph, pm, pl;
            split8(x[u], ph, pm, pl);
            unsigned char *dst = Kimg + (size_t)j * KP + d0 * 2;
            *reinterpret_cast<bf16x8 *>(dst) = ph;
            *reinterpret_cast<bf16x8 *>(dst + (size_t)rows * KP) = pm;
            *reinterpret_cast<bf16x8 *>(dst + (size_t)2 * rows * KP) = pl;
            if (f32_copy) {
                float *fr = f32_copy + (size_t)j * (D + 4) + d0;
                *reinterpret_cast<float4 *>(fr) = make_float4(x[u][0], x[u][1], x[u][2], x[u][3]);
                *reinterpret_cast<float4 *>(fr + 4) = make_float4(x[u][4], x[u][5], x[u][6], x[u][7]);
            }
        }
    }
}
// V of head h of image b -> three transposed bf16 planes [D][vp(rows)], keys of a 32-block in C-layout order (see above)
template <typename T, int D, int NT, int UG>
__device__ __forceinline__ void stage_v_planes(const T *__restrict__ kv, unsigned char *Vimg, int b, int h, int KV, int heads, int rows) {
    constexpr int NG = (D * (kSraKeys / 8) + NT - 1) / NT;                   // groups of 8 key slots per thread, UG of them in flight
    const int VP = X3Geo<D>::vp(rows);
    const int C = heads * D;
#pragma unroll
    for (int u0 = 0; u0 < NG; u0 += UG) {
        float x[UG][8];
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            const int idx = threadIdx.x + (u0 + u) * NT;
            const int d = idx % D, pg = idx / D;                             // pg: group of 8 key slots
            const int blk = pg >> 2, s = (pg >> 1) & 1, half = pg & 1;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int key = blk * 32 + (i & 3) + 8 * (2 * s + (i >> 2)) + 4 * half;
                x[u][i] = (u0 + u < NG && idx < D * (rows / 8) && key < KV) ? VecIO<T>::load1(kv + ((size_t)b * KV + key) * 2 * C + C + h * D + d) : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            const int idx = threadIdx.x + (u0 + u) * NT;
            if (u0 + u < NG && idx < D * (rows / 8)) {
                const int d = idx % D, pg = idx / D;
                bf16x8 ph, pm, pl;
                split8(x[u], ph, pm, pl);
                unsigned char *dst = Vimg + (size_t)d * VP + pg * 16;
                *reinterpret_cast<bf16x8 *>(dst) = ph;
                *reinterpret_cast<bf16x8 *>(dst + (size_t)D * VP) = pm;
                *reinterpret_cast<bf16x8 *>(dst + (size_t)2 * D * VP) = pl;
            }
        }
    }
}

// FULL: KV == 256 (every MiT stage at 512 x 512) -- eight whole key blocks, no masks, no branches: one basic block per phase, so the
// compiler can run the split of the next fragments under the MFMAs of the current ones.
template <typename T, int D, int NW, int QT, bool PHASED, bool FULL>
__global__ __launch_bounds__(NW * 64) void sra_fwd_x3(const T *__restrict__ q, const T *__restrict__ kv, T *__restrict__ out,
                                                       float *__restrict__ lse, int N, int KV, int heads, float cs /* scale*log2e */,
                                                       unsigned long long *__restrict__ stamps /* diagnostics, normally null */) {
    constexpr int KP = X3Geo<D>::KP, DB = D / 32, KS = D / 16;
    // -DSD_SRA_STAMPS (diagnostic build only, tools/sra_stamps.py): s_memtime at the phase boundaries of workgroup 0.  The product build has
    // neither the stores nor the branches (the undeclared export and the per-stamp `blockIdx == 0` test left the product kernel).
#ifdef SD_SRA_STAMPS
    int stamp_i = 0;
    auto stamp = [&]() {
        if (stamps && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (threadIdx.x & 63) == 0)
            stamps[(threadIdx.x >> 6) * 32 + stamp_i] = __builtin_amdgcn_s_memtime();
        ++stamp_i;
    };
#else
    (void)stamps;
    auto stamp = []() {};
#endif
    stamp();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_x3[];
    const int nblk = FULL ? 8 : (KV + 31) / 32, rows = nblk * 32;
    const int VP = X3Geo<D>::vp(rows);
    unsigned char *Kimg = smem_x3, *Vimg = PHASED ? smem_x3 : smem_x3 + X3Geo<D>::k_bytes(rows);
    const size_t kplane = (size_t)rows * KP, vplane = (size_t)D * VP;
    const int h = blockIdx.y, b = blockIdx.z, C = heads * D;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, c = l & 31, half = l >> 5;
    if (!PHASED) {
        stage_k_planes<T, D, NW * 64>(kv, Kimg, b, h, KV, heads, rows);
        stage_v_planes<T, D, NW * 64, (PHASED ? 2 : 4)>(kv, Vimg, b, h, KV, heads, rows);
        __syncthreads();
    }
    stamp();                                                                 // 1: staged
    for (int qt = 0; qt < QT; ++qt) {
        const int n = (blockIdx.x * QT + qt) * (NW * 32) + w * 32 + c;
        const bool live = n < N;
        // q row of this lane's query, k-steps of 16: elements ks*16 + 8*half .. +8 (B fragment: B[k = 8 half + i][j = c])
        float qx[KS][8];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int e = 0; e < 8; ++e) qx[ks][e] = 0.f;
            if (live) load_span<T, 8>(q + ((size_t)b * N + n) * C + h * D + ks * 16 + 8 * half, qx[ks]);
        }
        if (PHASED) {
            if (qt) __syncthreads();                                         // the V planes of the previous tile have been consumed
            stage_k_planes<T, D, NW * 64>(kv, Kimg, b, h, KV, heads, rows);
            __syncthreads();
        }
        bf16x8 qh[KS], qm[KS], ql[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int e = 0; e < 8; ++e) qx[ks][e] *= cs;
            split8(qx[ks], qh[ks], qm[ks], ql[ks]);
        }
        stamp();                                                             // q planes ready
        f32x16 S[8];
#pragma unroll
        for (int blk = 0; blk < 8; ++blk)
#pragma unroll
            for (int e = 0; e < 16; ++e) S[blk][e] = 0.f;
        // two key blocks at a time: their accumulation chains are independent, the MFMAs alternate between them
#pragma unroll
        for (int bp = 0; bp < 8; bp += 2) {
            if (FULL || bp < nblk) {
                const unsigned char *k0 = Kimg + (size_t)(bp * 32 + c) * KP + half * 16;
                const unsigned char *k1 = k0 + (size_t)32 * KP;
                const bool two = FULL || bp + 1 < nblk;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(k0 + ks * 32);
                    const bf16x8 am = *reinterpret_cast<const bf16x8 *>(k0 + ks * 32 + kplane);
                    const bf16x8 al = *reinterpret_cast<const bf16x8 *>(k0 + ks * 32 + 2 * kplane);
                    if (two) {
                        const bf16x8 bh = *reinterpret_cast<const bf16x8 *>(k1 + ks * 32);
                        const bf16x8 bm = *reinterpret_cast<const bf16x8 *>(k1 + ks * 32 + kplane);
                        const bf16x8 bl = *reinterpret_cast<const bf16x8 *>(k1 + ks * 32 + 2 * kplane);
                        f32x16 x = S[bp], y = S[bp + 1];
                        x = mfma16(am, qm[ks], x); y = mfma16(bm, qm[ks], y);
                        x = mfma16(al, qh[ks], x); y = mfma16(bl, qh[ks], y);
                        x = mfma16(ah, ql[ks], x); y = mfma16(bh, ql[ks], y);
                        x = mfma16(am, qh[ks], x); y = mfma16(bm, qh[ks], y);
                        x = mfma16(ah, qm[ks], x); y = mfma16(bh, qm[ks], y);
                        x = mfma16(ah, qh[ks], x); y = mfma16(bh, qh[ks], y);
                        S[bp] = x; S[bp + 1] = y;
                    } else {
                        S[bp] = mfma_x3(ah, am, al, qh[ks], qm[ks], ql[ks], S[bp]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);                               // keep the fragment reads of later blocks from piling up in registers
        }
        stamp();                                                             // S^T issued
        if (!FULL) {
#pragma unroll
            for (int blk = 0; blk < 8; ++blk) {
                if (blk * 32 + 32 > KV) {                                    // wave-uniform: only the ragged / absent blocks pay for the mask
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (blk * 32 + crow(e, half) >= KV) S[blk][e] = kNegBig;
                }
            }
        }
        float m = kNegBig;
#pragma unroll
        for (int blk = 0; blk < 8; ++blk)
#pragma unroll
            for (int e = 0; e < 16; ++e) m = fmaxf(m, S[blk][e]);
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float lsum = 0.f;
#pragma unroll
        for (int blk = 0; blk < 8; ++blk) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                S[blk][e] = ex2(S[blk][e] - m);                              // masked / absent keys: exp2(-1e30 - m) = 0
                lsum += S[blk][e];
            }
        }
        lsum += __shfl_xor(lsum, 32, 64);
        stamp();                                                             // softmax done
        if (PHASED) {
            __syncthreads();                                                 // every wave is done with the K planes
            stage_v_planes<T, D, NW * 64, (PHASED ? 2 : 4)>(kv, Vimg, b, h, KV, heads, rows);
            __syncthreads();
        }
        f32x16 O[DB];
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) O[db][e] = 0.f;
        // 16 k-steps (key block, s).  Software pipeline, one scheduling region per step: the V^T fragments of step i + 1 are read and the
        // probabilities of step i + 1 split while the MFMAs of step i run -- spelled out with sched_group_barrier (1 MFMA, then a share of
        // the ~44 vector instructions of a split) because left alone the scheduler issues the split first and the six MFMAs after it.
        bf16x8 ph[2], pm[2], pl[2], vh[2][DB], vm[2][DB], vl[2][DB];
        auto read_v = [&](int st, int buf) {
            const int blk = st >> 1, s = st & 1;
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const unsigned char *vrow = Vimg + (size_t)(db * 32 + c) * VP + (blk * 32 + s * 16 + half * 8) * 2;
                vh[buf][db] = *reinterpret_cast<const bf16x8 *>(vrow);
                vm[buf][db] = *reinterpret_cast<const bf16x8 *>(vrow + vplane);
                vl[buf][db] = *reinterpret_cast<const bf16x8 *>(vrow + 2 * vplane);
            }
        };
        // (taking the exponentials here too, step by step under the MFMAs, was slower: 35.8 vs 32.5 us at stage 1 -- the step becomes
        // issue-bound at ~76 vector instructions per 6 MFMAs)
        auto split_p = [&](int st, int buf) {
            float pv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) pv[i] = S[st >> 1][8 * (st & 1) + i];
            split8(pv, ph[buf], pm[buf], pl[buf]);
        };
        read_v(0, 0);
        split_p(0, 0);
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const int blk = st >> 1, cur = st & 1;
            if (FULL || blk < nblk) {
                if (st + 1 < 16 && (FULL || ((st + 1) >> 1) < nblk)) {       // never a fragment beyond the staged key blocks
                    read_v(st + 1, cur ^ 1);
                    split_p(st + 1, cur ^ 1);
                }
#pragma unroll
                for (int db = 0; db < DB; ++db) O[db] = mfma_x3(vh[cur][db], vm[cur][db], vl[cur][db], ph[cur], pm[cur], pl[cur], O[db]);
                __builtin_amdgcn_sched_group_barrier(0x100, 3 * DB, 0);      // the LDS reads of the next step first
#pragma unroll
                for (int i = 0; i < 6 * DB; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, (48 + 6 * DB - 1) / (6 * DB), 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        stamp();                                                             // PV issued
        if (live) {
            const float inv = 1.f / lsum;
            T *orow = out + ((size_t)b * N + n) * C + h * D;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    Q4<T>::store(orow + db * 32 + 8 * g + 4 * half,
                                 make_float4(O[db][4 * g] * inv, O[db][4 * g + 1] * inv, O[db][4 * g + 2] * inv, O[db][4 * g + 3] * inv));
            if (half == 0) lse[((size_t)b * heads + h) * N + n] = m + __builtin_amdgcn_logf(lsum);   // base-2 lse of the scaled scores
        }
        stamp();                                                             // stored
    }
}

// ---- forward for bf16 STORAGE on the bf16 matrix pipe (config 5) -------------------------------------------------------------------
// q, K and V are bf16 in memory, so each is ONE exact plane: K rows are copied as they are into LDS ([key][d], pitch 2D + 16 bytes), V goes
// in transposed with the key-slot order of sra_fwd_x3 ([d][key slot]), a lane's q fragment is a 16-byte global load.  S^T = K Q^T is exact
// products with fp32 accumulation; the scale is applied with the subtraction of the maximum (one fma); P is rounded to bf16 for the second
// product (as every bf16 attention does; row sums are taken before the rounding), accumulation fp32.  2 + 2 D/32 MFMAs per 32-key block
// against 16 + 16 D/32 f32-input ones at twice the cycles each; 71 KB of LDS at head_dim 64.
template <int D, int NW, int QT>
__global__ __launch_bounds__(NW * 64) void sra_fwd_b16(const bf16_t *__restrict__ q, const bf16_t *__restrict__ kv, bf16_t *__restrict__ out,
                                                        float *__restrict__ lse, int N, int KV, int heads, float cs /* scale*log2e */) {
    constexpr int KP = 2 * D + 16, DB = D / 32, KS = D / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b16[];
    const int nblk = (KV + 31) / 32, rows = nblk * 32;
    const int VP = 2 * rows + 16;
    unsigned char *Kimg = smem_b16, *Vimg = smem_b16 + (size_t)rows * KP;
    const int h = blockIdx.y, b = blockIdx.z, C = heads * D;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, c = l & 31, half = l >> 5;
    for (int idx = threadIdx.x; idx < rows * (D / 8); idx += NW * 64) {          // K: 16-byte pieces, rows >= KV zero
        const int j = idx / (D / 8), d0 = (idx % (D / 8)) * 8;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (j < KV) v = *reinterpret_cast<const u32x4 *>(kv + ((size_t)b * KV + j) * 2 * C + h * D + d0);
        *reinterpret_cast<u32x4 *>(Kimg + (size_t)j * KP + d0 * 2) = v;
    }
    for (int idx = threadIdx.x; idx < rows * (D / 8); idx += NW * 64) {          // V: 16-byte loads along d, 2-byte transposed stores
        const int j = idx / (D / 8), d0 = (idx % (D / 8)) * 8;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (j < KV) v = *reinterpret_cast<const u32x4 *>(kv + ((size_t)b * KV + j) * 2 * C + C + h * D + d0);
        const int r = j & 31, slot = (j & ~31) + 16 * (r >> 4) + 8 * ((r >> 2) & 1) + (r & 3) + 4 * ((r >> 3) & 1);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            *reinterpret_cast<unsigned short *>(Vimg + (size_t)(d0 + e) * VP + slot * 2) = (unsigned short)((e & 1) ? (v[e >> 1] >> 16) : (v[e >> 1] & 0xffffu));
    }
    __syncthreads();
    for (int qt = 0; qt < QT; ++qt) {
        const int n = (blockIdx.x * QT + qt) * (NW * 32) + w * 32 + c;
        const bool live = n < N;
        bf16x8 qf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (live) v = *reinterpret_cast<const u32x4 *>(q + ((size_t)b * N + n) * C + h * D + ks * 16 + 8 * half);
            qf[ks] = __builtin_bit_cast(bf16x8, v);
        }
        f32x16 S[8];
#pragma unroll
        for (int blk = 0; blk < 8; ++blk)
#pragma unroll
            for (int e = 0; e < 16; ++e) S[blk][e] = 0.f;
#pragma unroll
        for (int blk = 0; blk < 8; ++blk) {
            if (blk < nblk) {
                const unsigned char *krow = Kimg + (size_t)(blk * 32 + c) * KP + half * 16;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) S[blk] = mfma16(*reinterpret_cast<const bf16x8 *>(krow + ks * 32), qf[ks], S[blk]);
            }
        }
        float m = kNegBig;
#pragma unroll
        for (int blk = 0; blk < 8; ++blk) {
            if (blk * 32 + 32 > KV) {                             // wave-uniform: only ragged / absent blocks are masked
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (blk * 32 + crow(e, half) >= KV) S[blk][e] = kNegBig;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) m = fmaxf(m, S[blk][e]);
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        const float mc = -m * cs;
        float lsum = 0.f;
#pragma unroll
        for (int blk = 0; blk < 8; ++blk)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                S[blk][e] = ex2(fmaf(S[blk][e], cs, mc));         // masked / absent keys: exp2(-1e30 cs + ...) = 0
                lsum += S[blk][e];
            }
        lsum += __shfl_xor(lsum, 32, 64);
        f32x16 O[DB];
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) O[db][e] = 0.f;
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const int blk = st >> 1, s2 = st & 1;
            if (blk < nblk) {
                bf16x8 pf;
#pragma unroll
                for (int i = 0; i < 8; ++i) pf[i] = static_cast<__bf16>(S[blk][8 * s2 + i]);
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const unsigned char *vrow = Vimg + (size_t)(db * 32 + c) * VP + (blk * 32 + s2 * 16 + half * 8) * 2;
                    O[db] = mfma16(*reinterpret_cast<const bf16x8 *>(vrow), pf, O[db]);
                }
            }
        }
        if (live) {
            const float inv = 1.f / lsum;
            bf16_t *orow = out + ((size_t)b * N + n) * C + h * D;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    Q4<bf16_t>::store(orow + db * 32 + 8 * g + 4 * half,
                                      make_float4(O[db][4 * g] * inv, O[db][4 * g + 1] * inv, O[db][4 * g + 2] * inv, O[db][4 * g + 3] * inv));
            if (half == 0) lse[((size_t)b * heads + h) * N + n] = m * cs + __builtin_amdgcn_logf(lsum);   // base-2 lse of the scaled scores
        }
    }
}

// ---- backward, queries: dq and delta = sum_d dO*O.  Same grid / ownership as the forward ------------------------------------
template <typename T, int D, int NW>
__global__ __launch_bounds__(NW * 64) void sra_bwd_dq(const T *__restrict__ q, const T *__restrict__ kv, const T *__restrict__ out,
                                                           const T *__restrict__ dout, const float *__restrict__ lse, T *__restrict__ dq,
                                                           float *__restrict__ delta, int N, int KV, int heads, float cs, float scale) {
    constexpr int PD = D + 4, DB = D / 32;
    extern __shared__ float smem[];
    const int nblk = (KV + 31) / 32;
    float *Ks = smem, *Vs = smem + nblk * 32 * PD;
    const int h = blockIdx.y, b = blockIdx.z, C = heads * D;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, c = l & 31, half = l >> 5;
    const int n = blockIdx.x * (NW * 32) + w * 32 + c;
    const bool live = n < N;
    float qv[D / 2], gv[D / 2];
#pragma unroll
    for (int t = 0; t < D / 2; ++t) { qv[t] = 0.f; gv[t] = 0.f; }
    float dl = 0.f, L = 0.f;
    if (live) {
        const size_t row = ((size_t)b * N + n) * C + h * D + half * (D / 2);
        load_span<T, D / 2>(q + row, qv);
        load_span<T, D / 2>(dout + row, gv);
        float ov[D / 2];
        load_span<T, D / 2>(out + row, ov);
#pragma unroll
        for (int t = 0; t < D / 2; ++t) dl = fmaf(gv[t], ov[t], dl);
        L = lse[((size_t)b * heads + h) * N + n];
    }
    dl += __shfl_xor(dl, 32, 64);
#pragma unroll
    for (int t = 0; t < D / 2; ++t) qv[t] *= cs;
    stage_kv<T, D, PD>(kv, Ks, Vs, b, h, KV, heads, nblk * 32);
    __syncthreads();
    f32x16 G[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) G[db][e] = 0.f;
    for (int blk = 0; blk < nblk; ++blk) {
        f32x16 s = tile_t<D, PD>(Ks, blk * 32, qv, c, half);
        const f32x16 dp = tile_t<D, PD>(Vs, blk * 32, gv, c, half);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float p = (blk * 32 + crow(e, half) < KV) ? ex2(s[e] - L) : 0.f;
            s[e] = p * (dp[e] - dl);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float *krow = Ks + (blk * 32 + crow(e, half)) * PD + c;
#pragma unroll
            for (int db = 0; db < DB; ++db) G[db] = mfma(krow[db * 32], s[e], G[db]);
        }
    }
    if (!live) return;
    T *grow = dq + ((size_t)b * N + n) * C + h * D;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            Q4<T>::store(grow + db * 32 + 8 * g + 4 * half, make_float4(G[db][4 * g] * scale, G[db][4 * g + 1] * scale,
                                                                          G[db][4 * g + 2] * scale, G[db][4 * g + 3] * scale));
    if (half == 0) delta[((size_t)b * heads + h) * N + n] = dl;
}

// ---- backward, queries, split-bf16 for the two recomputed products (head_dim 32): S^T = K Q^T and dP^T = V dO^T run on bf16 planes of K and
// V ([key][d], staged once per workgroup) against planes of q and dO split once per tile; dQ^T += K^T dS^T stays on the exact f32 MFMA with
// an fp32 copy of K beside the planes (planes of K^T as well would need 173 KB of LDS; these three are 156 KB).  Per 32-key block:
// 24 x 32 + 16 x 64 cycles of matrix pipe against 48 x 64.
template <typename T, int NW, int QT>
__global__ __launch_bounds__(NW * 64) void sra_bwd_dq_x3(const T *__restrict__ q, const T *__restrict__ kv, const T *__restrict__ out,
                                                          const T *__restrict__ dout, const float *__restrict__ lse, T *__restrict__ dq,
                                                          float *__restrict__ delta, int N, int KV, int heads, float cs, float scale) {
    constexpr int D = 32, KP = X3Geo<D>::KP, PD = D + 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dq[];
    const int nblk = (KV + 31) / 32, rows = nblk * 32;
    const size_t plane = (size_t)rows * KP;
    unsigned char *Kp = smem_dq, *Vp = smem_dq + 3 * plane;
    float *Kf = reinterpret_cast<float *>(smem_dq + 6 * plane);
    const int h = blockIdx.y, b = blockIdx.z, C = heads * D;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, c = l & 31, half = l >> 5;
    stage_k_planes<T, D, NW * 64>(kv, Kp, b, h, KV, heads, rows, 0, Kf);
    stage_k_planes<T, D, NW * 64>(kv, Vp, b, h, KV, heads, rows, C, nullptr);
    __syncthreads();
    for (int qt = 0; qt < QT; ++qt) {
        const int n = (blockIdx.x * QT + qt) * (NW * 32) + w * 32 + c;
        const bool live = n < N;
        float qx[2][8], gx[2][8];
        float dl = 0.f, L = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float ox[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { qx[ks][e] = 0.f; gx[ks][e] = 0.f; ox[e] = 0.f; }
            if (live) {
                const size_t row = ((size_t)b * N + n) * C + h * D + ks * 16 + 8 * half;
                load_span<T, 8>(q + row, qx[ks]);
                load_span<T, 8>(dout + row, gx[ks]);
                load_span<T, 8>(out + row, ox);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) dl = fmaf(gx[ks][e], ox[e], dl);
        }
        if (live) L = lse[((size_t)b * heads + h) * N + n];
        dl += __shfl_xor(dl, 32, 64);
        bf16x8 qh[2], qm[2], ql[2], gh[2], gm[2], gl[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int e = 0; e < 8; ++e) qx[ks][e] *= cs;
            split8(qx[ks], qh[ks], qm[ks], ql[ks]);
            split8(gx[ks], gh[ks], gm[ks], gl[ks]);
        }
        f32x16 G;
#pragma unroll
        for (int e = 0; e < 16; ++e) G[e] = 0.f;
        for (int blk = 0; blk < nblk; ++blk) {
            f32x16 s = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dp = s;
            const unsigned char *kr = Kp + (size_t)(blk * 32 + c) * KP + half * 16, *vr = Vp + (size_t)(blk * 32 + c) * KP + half * 16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 kh = *reinterpret_cast<const bf16x8 *>(kr + ks * 32), km = *reinterpret_cast<const bf16x8 *>(kr + ks * 32 + plane),
                             kl = *reinterpret_cast<const bf16x8 *>(kr + ks * 32 + 2 * plane);
                const bf16x8 vh = *reinterpret_cast<const bf16x8 *>(vr + ks * 32), vm = *reinterpret_cast<const bf16x8 *>(vr + ks * 32 + plane),
                             vl = *reinterpret_cast<const bf16x8 *>(vr + ks * 32 + 2 * plane);
                s = mfma_x3(kh, km, kl, qh[ks], qm[ks], ql[ks], s);
                dp = mfma_x3(vh, vm, vl, gh[ks], gm[ks], gl[ks], dp);
            }
            if (blk * 32 + 32 <= KV) {                        // wave-uniform: whole blocks carry no mask
#pragma unroll
                for (int e = 0; e < 16; ++e) s[e] = ex2(s[e] - L) * (dp[e] - dl);
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float p = (blk * 32 + crow(e, half) < KV) ? ex2(s[e] - L) : 0.f;
                    s[e] = p * (dp[e] - dl);
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) G = mfma(Kf[(blk * 32 + crow(e, half)) * PD + c], s[e], G);
        }
        if (live) {
            T *grow = dq + ((size_t)b * N + n) * C + h * D;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                Q4<T>::store(grow + 8 * g + 4 * half, make_float4(G[4 * g] * scale, G[4 * g + 1] * scale, G[4 * g + 2] * scale, G[4 * g + 3] * scale));
            if (half == 0) delta[((size_t)b * heads + h) * N + n] = dl;
        }
    }
}

// ---- backward, keys.  grid (nchunk, heads, B): a workgroup walks `qchunk` queries (32 at a time through LDS); wave w keeps the
// key columns 64w .. 64w+63 (two 32-column blocks) of K and V in registers and accumulates their dK^T / dV^T tiles.
// part: [b][h][chunk][2 (k,v)][KV][D] fp32
template <typename T, int D>
__global__ __launch_bounds__(kSraThreads) void sra_bwd_dkv(const T *__restrict__ q, const T *__restrict__ kv, const T *__restrict__ dout,
                                                            const float *__restrict__ lse, const float *__restrict__ delta,
                                                            float *__restrict__ part, int N, int KV, int heads, int nchunk, int qchunk,
                                                            float cs, float scale) {
    constexpr int PD = D + 4, DB = D / 32, G4 = D / 4;
    __shared__ float Qs[32 * PD], Gs[32 * PD], Ls[32], Ds[32];
    const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z, C = heads * D;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, c = l & 31, half = l >> 5;
    float kreg[2][D / 2], vreg[2][D / 2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int j = 64 * w + 32 * cb + c;
#pragma unroll
        for (int t = 0; t < D / 2; ++t) { kreg[cb][t] = 0.f; vreg[cb][t] = 0.f; }
        if (j < KV) {
            const T *row = kv + ((size_t)b * KV + j) * 2 * C + h * D + half * (D / 2);
            load_span<T, D / 2>(row, kreg[cb]);
            load_span<T, D / 2>(row + C, vreg[cb]);
        }
#pragma unroll
        for (int t = 0; t < D / 2; ++t) kreg[cb][t] *= cs;
    }
    f32x16 aK[2][DB], aV[2][DB];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) { aK[cb][db][e] = 0.f; aV[cb][db][e] = 0.f; }
    const bool wave_live = 64 * w < KV;                       // waves whose key columns are all padding only help with the staging
    const int i0 = chunk * qchunk, i1 = min(N, i0 + qchunk);
    const T *qb = q + (size_t)b * N * C + h * D;
    const T *gb = dout + (size_t)b * N * C + h * D;
    const float *lb = lse + ((size_t)b * heads + h) * N;
    const float *db_ = delta + ((size_t)b * heads + h) * N;
    for (int t0 = i0; t0 < i1; t0 += 32) {
        __syncthreads();                                      // the previous tile has been consumed
        for (int idx = threadIdx.x; idx < 32 * G4; idx += kSraThreads) {
            const int r = idx / G4, c4 = (idx % G4) * 4;
            float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f), g4 = q4;
            if (t0 + r < i1) {
                q4 = Q4<T>::load(qb + (size_t)(t0 + r) * C + c4);
                g4 = Q4<T>::load(gb + (size_t)(t0 + r) * C + c4);
            }
            *reinterpret_cast<float4 *>(Qs + r * PD + c4) = q4;
            *reinterpret_cast<float4 *>(Gs + r * PD + c4) = g4;
        }
        if (threadIdx.x < 32) Ls[threadIdx.x] = (t0 + threadIdx.x < i1) ? lb[t0 + threadIdx.x] : 0.f;
        else if (threadIdx.x < 64) Ds[threadIdx.x - 32] = (t0 + threadIdx.x - 32 < i1) ? db_[t0 + threadIdx.x - 32] : 0.f;
        __syncthreads();
        if (!wave_live) continue;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            // S tile (rows = queries of the LDS tile, cols = this block's keys): A = Q rows from LDS, B = K columns in registers
            f32x16 s = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dp = s;
            const float *qrow = Qs + c * PD + half * (D / 2), *grow = Gs + c * PD + half * (D / 2);
#pragma unroll
            for (int t = 0; t < D / 2; t += 4) {
                const float4 a = *reinterpret_cast<const float4 *>(qrow + t), g = *reinterpret_cast<const float4 *>(grow + t);
                s = mfma(a.x, kreg[cb][t], s);      dp = mfma(g.x, vreg[cb][t], dp);
                s = mfma(a.y, kreg[cb][t + 1], s);  dp = mfma(g.y, vreg[cb][t + 1], dp);
                s = mfma(a.z, kreg[cb][t + 2], s);  dp = mfma(g.z, vreg[cb][t + 2], dp);
                s = mfma(a.w, kreg[cb][t + 3], s);  dp = mfma(g.w, vreg[cb][t + 3], dp);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = crow(e, half);
                const float p = ex2(s[e] - Ls[r]);            // padded query rows: q = dO = 0, lse = delta = 0 -> contribute exactly 0
                dp[e] = p * (dp[e] - Ds[r]);
                s[e] = p;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = crow(e, half);
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    aV[cb][db] = mfma(Gs[r * PD + db * 32 + c], s[e], aV[cb][db]);    // dV^T += dO^T P
                    aK[cb][db] = mfma(Qs[r * PD + db * 32 + c], dp[e], aK[cb][db]);   // dK^T += Q^T dS
                }
            }
        }
    }
    float *pk = part + (((size_t)b * heads + h) * nchunk + chunk) * 2 * KV * D;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int j = 64 * w + 32 * cb + c;
        if (j >= KV) continue;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float *dst = pk + (size_t)j * D + db * 32 + 8 * g + 4 * half;
                *reinterpret_cast<float4 *>(dst) = make_float4(aK[cb][db][4 * g] * scale, aK[cb][db][4 * g + 1] * scale,
                                                               aK[cb][db][4 * g + 2] * scale, aK[cb][db][4 * g + 3] * scale);
                *reinterpret_cast<float4 *>(dst + (size_t)KV * D) =
                    make_float4(aV[cb][db][4 * g], aV[cb][db][4 * g + 1], aV[cb][db][4 * g + 2], aV[cb][db][4 * g + 3]);
            }
    }
}

// ---- backward, keys, split-bf16 (head_dim 32).  Same ownership and partial layout as sra_bwd_dkv: wave w keeps key columns
// 64w .. 64w+63 -- here as bf16 planes of K (pre-scaled) and V in registers -- and 32-query tiles stream through LDS, split ONCE at
// staging into planes in two layouts: [q][d] (A operand of S = Q K^T and dP = dO V^T) and transposed [d][q slot] (A operand of
// dV^T += dO^T P and dK^T += Q^T dS; q slots in the order a lane of the S tile holds its rows, so the 8 values a lane owns per k-step are
// its B fragment).  All four products run as six bf16 MFMAs per 16 k: 48 x 32 cycles per (tile, key block) against 64 x 64.
template <typename T>
__global__ __launch_bounds__(kSraThreads) void sra_bwd_dkv_x3(const T *__restrict__ q, const T *__restrict__ kv, const T *__restrict__ dout,
                                                               const float *__restrict__ lse, const float *__restrict__ delta,
                                                               float *__restrict__ part, int N, int KV, int heads, int nchunk, int qchunk,
                                                               float cs, float scale) {
    constexpr int D = 32, RP = 80, PL = 32 * RP;              // plane row pitch (64 B of bf16 + 16), bytes per plane
    __shared__ __attribute__((aligned(16))) unsigned char Qp[3 * PL], Gp[3 * PL], QTp[3 * PL], GTp[3 * PL];
    __shared__ float Ls[32], Ds[32];
    const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z, C = heads * D;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, c = l & 31, half = l >> 5;
    bf16x8 kh[2][2], km[2][2], kl[2][2], vh[2][2], vm[2][2], vl[2][2];    // [key block][k-step of 16 over d]
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int j = 64 * w + 32 * cb + c;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float kx[8], vx[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { kx[e] = 0.f; vx[e] = 0.f; }
            if (j < KV) {
                const T *row = kv + ((size_t)b * KV + j) * 2 * C + h * D + ks * 16 + 8 * half;
                load_span<T, 8>(row, kx);
                load_span<T, 8>(row + C, vx);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) kx[e] *= cs;
            split8(kx, kh[cb][ks], km[cb][ks], kl[cb][ks]);
            split8(vx, vh[cb][ks], vm[cb][ks], vl[cb][ks]);
        }
    }
    f32x16 aK[2], aV[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) { aK[cb][e] = 0.f; aV[cb][e] = 0.f; }
    const bool wave_live = 64 * w < KV;                       // waves whose key columns are all padding only help with the staging
    const int i0 = chunk * qchunk, i1 = min(N, i0 + qchunk);
    const T *qb = q + (size_t)b * N * C + h * D;
    const T *gb = dout + (size_t)b * N * C + h * D;
    const float *lb = lse + ((size_t)b * heads + h) * N;
    const float *db_ = delta + ((size_t)b * heads + h) * N;
    // staging role of this thread: threads 0..127 split the Q tile, 128..255 the dO tile; query row sr, 8 consecutive d from sd0, and the
    // row's slot in the transposed planes
    const bool stage_q = threadIdx.x < 128;
    const int sr = (threadIdx.x & 127) >> 2, sd0 = (threadIdx.x & 3) * 8;
    const int sslot = 16 * (sr >> 4) + 8 * ((sr >> 2) & 1) + (sr & 3) + 4 * ((sr >> 3) & 1);   // r = (i&3) + 8(2s + (i>>2)) + 4 half -> 16 s + 8 half + i
    const T *sb = stage_q ? qb : gb;
    unsigned char *const P_ = stage_q ? Qp : Gp, *const PT_ = stage_q ? QTp : GTp;
    // the rows of tile i + 1 (and its lse / delta) are requested before tile i is computed: one exposed memory latency per workgroup
    float x[8], ld = 0.f;
    auto request = [&](int t0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = 0.f;
        if (t0 + sr < i1) load_span<T, 8>(sb + (size_t)(t0 + sr) * C + sd0, x);
        ld = 0.f;
        if (threadIdx.x < 32) { if (t0 + threadIdx.x < i1) ld = lb[t0 + threadIdx.x]; }
        else if (threadIdx.x < 64) { if (t0 + threadIdx.x - 32 < i1) ld = db_[t0 + threadIdx.x - 32]; }
    };
    request(i0);
    for (int t0 = i0; t0 < i1; t0 += 32) {
        bf16x8 p0, p1, p2;
        split8(x, p0, p1, p2);
        const float ld_now = ld;
        __syncthreads();                                      // the previous tile has been consumed
        {
            unsigned char *d_ = P_ + sr * RP + sd0 * 2;
            *reinterpret_cast<bf16x8 *>(d_) = p0; *reinterpret_cast<bf16x8 *>(d_ + PL) = p1; *reinterpret_cast<bf16x8 *>(d_ + 2 * PL) = p2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                unsigned char *t_ = PT_ + (sd0 + e) * RP + sslot * 2;
                *reinterpret_cast<__bf16 *>(t_) = p0[e]; *reinterpret_cast<__bf16 *>(t_ + PL) = p1[e]; *reinterpret_cast<__bf16 *>(t_ + 2 * PL) = p2[e];
            }
        }
        if (threadIdx.x < 32) Ls[threadIdx.x] = ld_now;
        else if (threadIdx.x < 64) Ds[threadIdx.x - 32] = ld_now;
        if (t0 + 32 < i1) request(t0 + 32);
        __syncthreads();
        if (!wave_live) continue;
        // (a software pipeline across the two key blocks -- the exponentials and splits of one block issued between the MFMAs of the other
        // with sched_group_barrier, six scheduling regions per tile -- was measured slower: 79 vs 70 us at stage 1, 334 instead of 265 registers)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            // S tile (rows = queries of the LDS tile, cols = this block's keys) and dP tile
            f32x16 s = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dp = s;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const unsigned char *qa = Qp + c * RP + ks * 32 + half * 16, *ga = Gp + c * RP + ks * 32 + half * 16;
                const bf16x8 qh_ = *reinterpret_cast<const bf16x8 *>(qa), qm_ = *reinterpret_cast<const bf16x8 *>(qa + PL),
                             ql_ = *reinterpret_cast<const bf16x8 *>(qa + 2 * PL);
                const bf16x8 gh_ = *reinterpret_cast<const bf16x8 *>(ga), gm_ = *reinterpret_cast<const bf16x8 *>(ga + PL),
                             gl_ = *reinterpret_cast<const bf16x8 *>(ga + 2 * PL);
                s = mfma_x3(qh_, qm_, ql_, kh[cb][ks], km[cb][ks], kl[cb][ks], s);
                dp = mfma_x3(gh_, gm_, gl_, vh[cb][ks], vm[cb][ks], vl[cb][ks], dp);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = crow(e, half);
                const float p = ex2(s[e] - Ls[r]);            // padded query rows: q = dO = 0, lse = delta = 0 -> contribute exactly 0
                dp[e] = p * (dp[e] - Ds[r]);
                s[e] = p;
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                float pv[8], dv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { pv[i] = s[8 * st + i]; dv[i] = dp[8 * st + i]; }
                bf16x8 ph, pm, pl, dh, dm, dl;
                split8(pv, ph, pm, pl);
                split8(dv, dh, dm, dl);
                const unsigned char *ga = GTp + c * RP + (16 * st + 8 * half) * 2, *qa = QTp + c * RP + (16 * st + 8 * half) * 2;
                const bf16x8 gh_ = *reinterpret_cast<const bf16x8 *>(ga), gm_ = *reinterpret_cast<const bf16x8 *>(ga + PL),
                             gl_ = *reinterpret_cast<const bf16x8 *>(ga + 2 * PL);
                const bf16x8 qh_ = *reinterpret_cast<const bf16x8 *>(qa), qm_ = *reinterpret_cast<const bf16x8 *>(qa + PL),
                             ql_ = *reinterpret_cast<const bf16x8 *>(qa + 2 * PL);
                aV[cb] = mfma_x3(gh_, gm_, gl_, ph, pm, pl, aV[cb]);    // dV^T += dO^T P
                aK[cb] = mfma_x3(qh_, qm_, ql_, dh, dm, dl, aK[cb]);    // dK^T += Q^T dS
            }
        }
    }
    float *pk = part + (((size_t)b * heads + h) * nchunk + chunk) * 2 * KV * D;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int j = 64 * w + 32 * cb + c;
        if (j >= KV) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float *dst = pk + (size_t)j * D + 8 * g + 4 * half;
            *reinterpret_cast<float4 *>(dst) = make_float4(aK[cb][4 * g] * scale, aK[cb][4 * g + 1] * scale, aK[cb][4 * g + 2] * scale,
                                                           aK[cb][4 * g + 3] * scale);
            *reinterpret_cast<float4 *>(dst + (size_t)KV * D) = make_float4(aV[cb][4 * g], aV[cb][4 * g + 1], aV[cb][4 * g + 2], aV[cb][4 * g + 3]);
        }
    }
}

// ---- backward for bf16 STORAGE on the bf16 matrix pipe (config 5).  Same ownership, partial layout and saved statistics as the kernels
// above; every operand is one exact bf16 plane (copies, no splits), P and dS are rounded to bf16 for the three gradient products (as every
// bf16 attention backward does), accumulation fp32, the scale enters through the exponent's fma.

template <int D, int NW, int QT>
__global__ __launch_bounds__(NW * 64) void sra_bwd_dq_b16(const bf16_t *__restrict__ q, const bf16_t *__restrict__ kv, const bf16_t *__restrict__ out,
                                                           const bf16_t *__restrict__ dout, const float *__restrict__ lse, bf16_t *__restrict__ dq,
                                                           float *__restrict__ delta, int N, int KV, int heads, float cs, float scale) {
    constexpr int KP = 2 * D + 16, DB = D / 32, KS = D / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dqb[];
    const int nblk = (KV + 31) / 32, rows = nblk * 32;
    const int TP = 2 * rows + 16;
    unsigned char *Kp = smem_dqb, *Vp = Kp + (size_t)rows * KP, *KTp = Vp + (size_t)rows * KP;
    const int h = blockIdx.y, b = blockIdx.z, C = heads * D;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, c = l & 31, half = l >> 5;
    for (int idx = threadIdx.x; idx < rows * (D / 8); idx += NW * 64) {
        const int j = idx / (D / 8), d0 = (idx % (D / 8)) * 8;
        u32x4 kx = {0u, 0u, 0u, 0u}, vx = kx;
        if (j < KV) {
            const bf16_t *row = kv + ((size_t)b * KV + j) * 2 * C + h * D + d0;
            kx = *reinterpret_cast<const u32x4 *>(row);
            vx = *reinterpret_cast<const u32x4 *>(row + C);
        }
        *reinterpret_cast<u32x4 *>(Kp + (size_t)j * KP + d0 * 2) = kx;
        *reinterpret_cast<u32x4 *>(Vp + (size_t)j * KP + d0 * 2) = vx;
        const int slot = (j & ~31) + sra_slot(j & 31);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            *reinterpret_cast<unsigned short *>(KTp + (size_t)(d0 + e) * TP + slot * 2) = (unsigned short)((e & 1) ? (kx[e >> 1] >> 16) : (kx[e >> 1] & 0xffffu));
    }
    __syncthreads();
    for (int qt = 0; qt < QT; ++qt) {
        const int n = (blockIdx.x * QT + qt) * (NW * 32) + w * 32 + c;
        const bool live = n < N;
        bf16x8 qf[KS], gf[KS];
        float dl = 0.f, L = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4 qx = {0u, 0u, 0u, 0u}, gx = qx, ox = qx;
            if (live) {
                const size_t row = ((size_t)b * N + n) * C + h * D + ks * 16 + 8 * half;
                qx = *reinterpret_cast<const u32x4 *>(q + row);
                gx = *reinterpret_cast<const u32x4 *>(dout + row);
                ox = *reinterpret_cast<const u32x4 *>(out + row);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) dl = fmaf(bf16_lane(gx, e), bf16_lane(ox, e), dl);
            qf[ks] = __builtin_bit_cast(bf16x8, qx);
            gf[ks] = __builtin_bit_cast(bf16x8, gx);
        }
        if (live) L = lse[((size_t)b * heads + h) * N + n];
        dl += __shfl_xor(dl, 32, 64);
        f32x16 G[DB];
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) G[db][e] = 0.f;
        for (int blk = 0; blk < nblk; ++blk) {
            f32x16 s = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dp = s;
            const unsigned char *kr = Kp + (size_t)(blk * 32 + c) * KP + half * 16, *vr = Vp + (size_t)(blk * 32 + c) * KP + half * 16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s = mfma16(*reinterpret_cast<const bf16x8 *>(kr + ks * 32), qf[ks], s);
                dp = mfma16(*reinterpret_cast<const bf16x8 *>(vr + ks * 32), gf[ks], dp);
            }
            const bool whole = blk * 32 + 32 <= KV;            // wave-uniform
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float p = ex2(fmaf(s[e], cs, -L));
                if (!whole && blk * 32 + crow(e, half) >= KV) p = 0.f;
                s[e] = p * (dp[e] - dl);
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                bf16x8 df;
#pragma unroll
                for (int i = 0; i < 8; ++i) df[i] = static_cast<__bf16>(s[8 * st + i]);
#pragma unroll
                for (int db = 0; db < DB; ++db)
                    G[db] = mfma16(*reinterpret_cast<const bf16x8 *>(KTp + (size_t)(db * 32 + c) * TP + (blk * 32 + 16 * st + 8 * half) * 2), df, G[db]);
            }
        }
        if (live) {
            bf16_t *grow = dq + ((size_t)b * N + n) * C + h * D;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    Q4<bf16_t>::store(grow + db * 32 + 8 * g + 4 * half, make_float4(G[db][4 * g] * scale, G[db][4 * g + 1] * scale,
                                                                                      G[db][4 * g + 2] * scale, G[db][4 * g + 3] * scale));
            if (half == 0) delta[((size_t)b * heads + h) * N + n] = dl;
        }
    }
}

template <int D>
__global__ __launch_bounds__(kSraThreads) void sra_bwd_dkv_b16(const bf16_t *__restrict__ q, const bf16_t *__restrict__ kv,
                                                                const bf16_t *__restrict__ dout, const float *__restrict__ lse,
                                                                const float *__restrict__ delta, float *__restrict__ part, int N, int KV, int heads,
                                                                int nchunk, int qchunk, float cs, float scale) {
    constexpr int KP = 2 * D + 16, TP = 80, DB = D / 32, KS = D / 16, PIECES = 32 * (D / 8), NIT = 2 * PIECES / kSraThreads;
    __shared__ __attribute__((aligned(16))) unsigned char Qp[32 * KP], Gp[32 * KP], QTp[D * TP], GTp[D * TP];
    __shared__ float Ls[32], Ds[32];
    const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z, C = heads * D;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, c = l & 31, half = l >> 5;
    bf16x8 kf[2][KS], vf[2][KS];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int j = 64 * w + 32 * cb + c;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4 kx = {0u, 0u, 0u, 0u}, vx = kx;
            if (j < KV) {
                const bf16_t *row = kv + ((size_t)b * KV + j) * 2 * C + h * D + ks * 16 + 8 * half;
                kx = *reinterpret_cast<const u32x4 *>(row);
                vx = *reinterpret_cast<const u32x4 *>(row + C);
            }
            kf[cb][ks] = __builtin_bit_cast(bf16x8, kx);
            vf[cb][ks] = __builtin_bit_cast(bf16x8, vx);
        }
    }
    f32x16 aK[2][DB], aV[2][DB];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) { aK[cb][db][e] = 0.f; aV[cb][db][e] = 0.f; }
    const bool wave_live = 64 * w < KV;
    const int i0 = chunk * qchunk, i1 = min(N, i0 + qchunk);
    const bf16_t *qb = q + (size_t)b * N * C + h * D;
    const bf16_t *gb = dout + (size_t)b * N * C + h * D;
    const float *lb = lse + ((size_t)b * heads + h) * N;
    const float *db_ = delta + ((size_t)b * heads + h) * N;
    // staging pieces of this thread: piece id = threadIdx.x + it * 256 over [Q pieces | dO pieces]; a piece = 8 consecutive d of one row
    u32x4 x[NIT];
    float ld = 0.f;
    auto request = [&](int t0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int id = threadIdx.x + it * kSraThreads, which = id / PIECES, pc = id - which * PIECES;
            const int r = pc / (D / 8), d0 = (pc % (D / 8)) * 8;
            x[it] = u32x4{0u, 0u, 0u, 0u};
            if (t0 + r < i1) x[it] = *reinterpret_cast<const u32x4 *>((which ? gb : qb) + (size_t)(t0 + r) * C + d0);
        }
        ld = 0.f;
        if (threadIdx.x < 32) { if (t0 + threadIdx.x < i1) ld = lb[t0 + threadIdx.x]; }
        else if (threadIdx.x < 64) { if (t0 + threadIdx.x - 32 < i1) ld = db_[t0 + threadIdx.x - 32]; }
    };
    request(i0);
    for (int t0 = i0; t0 < i1; t0 += 32) {
        __syncthreads();                                      // the previous tile has been consumed
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int id = threadIdx.x + it * kSraThreads, which = id / PIECES, pc = id - which * PIECES;
            const int r = pc / (D / 8), d0 = (pc % (D / 8)) * 8;
            *reinterpret_cast<u32x4 *>((which ? Gp : Qp) + r * KP + d0 * 2) = x[it];
            unsigned char *tb = (which ? GTp : QTp) + sra_slot(r) * 2;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                *reinterpret_cast<unsigned short *>(tb + (d0 + e) * TP) = (unsigned short)((e & 1) ? (x[it][e >> 1] >> 16) : (x[it][e >> 1] & 0xffffu));
        }
        if (threadIdx.x < 32) Ls[threadIdx.x] = ld;
        else if (threadIdx.x < 64) Ds[threadIdx.x - 32] = ld;
        if (t0 + 32 < i1) request(t0 + 32);
        __syncthreads();
        if (!wave_live) continue;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            f32x16 s = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dp = s;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s = mfma16(*reinterpret_cast<const bf16x8 *>(Qp + c * KP + ks * 32 + half * 16), kf[cb][ks], s);
                dp = mfma16(*reinterpret_cast<const bf16x8 *>(Gp + c * KP + ks * 32 + half * 16), vf[cb][ks], dp);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = crow(e, half);
                const float p = ex2(fmaf(s[e], cs, -Ls[r]));  // padded query rows: q = dO = 0, lse = delta = 0 -> contribute exactly 0
                dp[e] = p * (dp[e] - Ds[r]);
                s[e] = p;
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                bf16x8 pf, df;
#pragma unroll
                for (int i = 0; i < 8; ++i) { pf[i] = static_cast<__bf16>(s[8 * st + i]); df[i] = static_cast<__bf16>(dp[8 * st + i]); }
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const int off = (db * 32 + c) * TP + (16 * st + 8 * half) * 2;
                    aV[cb][db] = mfma16(*reinterpret_cast<const bf16x8 *>(GTp + off), pf, aV[cb][db]);    // dV^T += dO^T P
                    aK[cb][db] = mfma16(*reinterpret_cast<const bf16x8 *>(QTp + off), df, aK[cb][db]);    // dK^T += Q^T dS
                }
            }
        }
    }
    float *pk = part + (((size_t)b * heads + h) * nchunk + chunk) * 2 * KV * D;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int j = 64 * w + 32 * cb + c;
        if (j >= KV) continue;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float *dst = pk + (size_t)j * D + db * 32 + 8 * g + 4 * half;
                *reinterpret_cast<float4 *>(dst) = make_float4(aK[cb][db][4 * g] * scale, aK[cb][db][4 * g + 1] * scale, aK[cb][db][4 * g + 2] * scale,
                                                               aK[cb][db][4 * g + 3] * scale);
                *reinterpret_cast<float4 *>(dst + (size_t)KV * D) =
                    make_float4(aV[cb][db][4 * g], aV[cb][db][4 * g + 1], aV[cb][db][4 * g + 2], aV[cb][db][4 * g + 3]);
            }
    }
}

// dkv[b][j][which][h][d] = sum_chunk part[b][h][chunk][which][j][d].  one thread per 4 output elements
template <typename T>
__global__ __launch_bounds__(256) void sra_dkv_reduce(const float *__restrict__ part, T *__restrict__ dkv, int B, int KV, int heads, int D,
                                                       int nchunk) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int dv = D / 4;
    const size_t total = (size_t)B * KV * 2 * heads * dv;
    if (t >= total) return;
    const int d = (int)(t % dv) * 4;
    size_t r = t / dv;
    const int h = (int)(r % heads); r /= heads;
    const int which = (int)(r % 2); r /= 2;
    const int j = (int)(r % KV);
    const int b = (int)(r / KV);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    const float *p = part + ((((size_t)b * heads + h) * nchunk) * 2 + which) * KV * D + (size_t)j * D + d;
    int ch = 0;
    for (; ch + 8 <= nchunk; ch += 8) {                       // 8 independent loads in flight, summed in chunk order
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4 *>(p + (size_t)(ch + u) * 2 * KV * D);
#pragma unroll
        for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    for (; ch < nchunk; ++ch) {
        const float4 v = *reinterpret_cast<const float4 *>(p + (size_t)ch * 2 * KV * D);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    T *o = dkv + (((size_t)b * KV + j) * 2 + which) * heads * D + (size_t)h * D + d;
    VecIO<T>::store1(o, a.x); VecIO<T>::store1(o + 1, a.y); VecIO<T>::store1(o + 2, a.z); VecIO<T>::store1(o + 3, a.w);
}

int g_sra_wide_min = 1024;   // tunable "sra_wide_min": 8-wave workgroups (256 queries) from this many queries per (image, head) on
int g_sra_dkv_wgs = 256;     // tunable "sra_dkv_wgs": workgroups the dK/dV pass spreads its query chunks over (one per CU: r06_ab_sra_tiles.txt)
struct SraPlan {
    int nchunk, qchunk;
};
SraPlan sra_plan(int B, int N, int KV, int heads) {
    SraPlan p;
    const long groups = (long)B * heads;
    long want = (g_sra_dkv_wgs + groups - 1) / groups;
    const long most = (N + 127) / 128;                  // at least 128 queries (4 LDS tiles) per workgroup
    if (want > most) want = most;
    if (want < 1) want = 1;
    p.qchunk = (int)(((N + want - 1) / want + 31) / 32 * 32);
    p.nchunk = (N + p.qchunk - 1) / p.qchunk;
    return p;
}
template <int D> size_t sra_lds_bytes(int KV) { return (size_t)2 * ((KV + 31) / 32) * 32 * (D + 4) * sizeof(float); }
// K+V of 256 keys need 72 KB (D=32) / 136 KB (D=64) of the CU's 160 KB: raise the kernel's dynamic-LDS cap once per process
template <typename K> int sra_raise_lds(K kernel, bool &raised) {
    if (raised) return 0;
    const int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    raised = rc == 0;
    return rc;
}

int sra_check(const void *q, const void *kv, const void *o, int dtype, int B, int N, int KV, int heads, int D) {
    if (!q || !kv || !o) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (B <= 0 || N <= 0 || KV <= 0 || heads <= 0) return SD_E_SHAPE;
    if ((D != 32 && D != 64) || KV > kSraKeys) return SD_E_UNSUPPORTED;
    if (B > 65535 || heads > 65535) return SD_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(kv) | reinterpret_cast<uintptr_t>(o)) & 15) return SD_E_ALIGN;
    return SD_OK;
}

inline bool sra_wide(int N) { return N >= g_sra_wide_min; }   // 8-wave workgroups once there are enough queries to fill them

template <typename T, int D, int NW, int QT>
int sra_fwd_launch_nw(const void *q, const void *kv, void *out, float *lse, int B, int N, int KV, int heads, float scale, hipStream_t st) {
    const dim3 grid((N + NW * 32 * QT - 1) / (NW * 32 * QT), heads, B);
    const size_t lds = sra_lds_bytes<D>(KV);
    static bool raised = false;
    int rc = sra_raise_lds(sra_fwd<T, D, NW, QT>, raised);
    if (rc) return rc;
    hipLaunchKernelGGL((sra_fwd<T, D, NW, QT>), grid, dim3(NW * 64), lds, st, (const T *)q, (const T *)kv, (T *)out, lse, N, KV, heads,
                       scale * kLog2e);
    return (int)hipGetLastError();
}
unsigned long long *g_sra_stamps = nullptr;   // diagnostics (sd_debug_sra_stamps): [8 waves][32] s_memtime values of workgroup 0
int g_sra_split_bf16 = 1;   // tunable "sra_split_bf16": fp32 storage runs the products on the bf16 matrix pipe, operands split in three

template <typename T, int D, int NW, int QT, bool FULL>
int sra_fwd_x3_launch_full(const void *q, const void *kv, void *out, float *lse, int B, int N, int KV, int heads, float scale, hipStream_t st) {
    constexpr bool PHASED = D == 64;
    const dim3 grid((N + NW * 32 * QT - 1) / (NW * 32 * QT), heads, B);
    const int rows = (KV + 31) / 32 * 32;
    const size_t kb = X3Geo<D>::k_bytes(rows), vb = X3Geo<D>::v_bytes(rows);
    const size_t lds = PHASED ? (kb > vb ? kb : vb) : kb + vb;
    static bool raised = false;
    int rc = sra_raise_lds(sra_fwd_x3<T, D, NW, QT, PHASED, FULL>, raised);
    if (rc) return rc;
    hipLaunchKernelGGL((sra_fwd_x3<T, D, NW, QT, PHASED, FULL>), grid, dim3(NW * 64), lds, st, (const T *)q, (const T *)kv, (T *)out, lse, N,
                       KV, heads, scale * kLog2e, g_sra_stamps);
    return (int)hipGetLastError();
}
template <typename T, int D, int NW, int QT>
int sra_fwd_x3_launch_nw(const void *q, const void *kv, void *out, float *lse, int B, int N, int KV, int heads, float scale, hipStream_t st) {
    return KV == kSraKeys ? sra_fwd_x3_launch_full<T, D, NW, QT, true>(q, kv, out, lse, B, N, KV, heads, scale, st)
                          : sra_fwd_x3_launch_full<T, D, NW, QT, false>(q, kv, out, lse, B, N, KV, heads, scale, st);
}

template <int D, int NW, int QT>
int sra_fwd_b16_launch(const void *q, const void *kv, void *out, float *lse, int B, int N, int KV, int heads, float scale, hipStream_t st) {
    const dim3 grid((N + NW * 32 * QT - 1) / (NW * 32 * QT), heads, B);
    const int rows = (KV + 31) / 32 * 32;
    const size_t lds = (size_t)rows * (2 * D + 16) + (size_t)D * (2 * rows + 16);
    static bool raised = false;
    int rc = sra_raise_lds(sra_fwd_b16<D, NW, QT>, raised);
    if (rc) return rc;
    hipLaunchKernelGGL((sra_fwd_b16<D, NW, QT>), grid, dim3(NW * 64), lds, st, (const bf16_t *)q, (const bf16_t *)kv, (bf16_t *)out, lse, N, KV,
                       heads, scale * kLog2e);
    return (int)hipGetLastError();
}

int g_sra_bf16_mfma = 1;   // tunable "sra_bf16_mfma": bf16 storage runs the forward on the bf16 matrix pipe (P rounded to bf16)

template <typename T, int D>
int sra_fwd_launch(const void *q, const void *kv, void *out, float *lse, int B, int N, int KV, int heads, float scale, hipStream_t st) {
    if constexpr (sizeof(T) == 2) if (g_sra_bf16_mfma) {
        if (N >= 8192) return sra_fwd_b16_launch<D, 8, 2>(q, kv, out, lse, B, N, KV, heads, scale, st);
        return sra_wide(N) ? sra_fwd_b16_launch<D, 8, 1>(q, kv, out, lse, B, N, KV, heads, scale, st)
                           : sra_fwd_b16_launch<D, 4, 1>(q, kv, out, lse, B, N, KV, heads, scale, st);
    }
    if constexpr (sizeof(T) == 4) if (g_sra_split_bf16) {
        if (N >= 8192 && D == 32) return sra_fwd_x3_launch_nw<T, D, 8, 2>(q, kv, out, lse, B, N, KV, heads, scale, st);   // (PHASED restages per tile)
        return sra_wide(N) ? sra_fwd_x3_launch_nw<T, D, 8, 1>(q, kv, out, lse, B, N, KV, heads, scale, st)
                           : sra_fwd_x3_launch_nw<T, D, 4, 1>(q, kv, out, lse, B, N, KV, heads, scale, st);
    }
    if (N >= 8192) return sra_fwd_launch_nw<T, D, 8, 2>(q, kv, out, lse, B, N, KV, heads, scale, st);   // 512 queries per K/V copy
    return sra_wide(N) ? sra_fwd_launch_nw<T, D, 8, 1>(q, kv, out, lse, B, N, KV, heads, scale, st)
                       : sra_fwd_launch_nw<T, D, 4, 1>(q, kv, out, lse, B, N, KV, heads, scale, st);
}

template <typename T, int D>
int sra_bwd_launch(const void *q, const void *kv, const void *out, const void *dout, const float *lse, void *dq, void *dkv, int B, int N,
                   int KV, int heads, float scale, void *ws, hipStream_t st) {
    const SraPlan p = sra_plan(B, N, KV, heads);
    float *delta = static_cast<float *>(ws);
    float *part = delta + (((size_t)B * heads * N + 3) & ~(size_t)3);
    const size_t lds = sra_lds_bytes<D>(KV);
    if constexpr (sizeof(T) == 2) if (g_sra_bf16_mfma) {
        const int rows = (KV + 31) / 32 * 32;
        const size_t ldsb = (size_t)2 * rows * (2 * D + 16) + (size_t)D * (2 * rows + 16);
        static bool r82 = false, r81 = false, r41 = false;
        int rc;
        if (N >= 8192) {
            if ((rc = sra_raise_lds(sra_bwd_dq_b16<D, 8, 2>, r82))) return rc;
            hipLaunchKernelGGL((sra_bwd_dq_b16<D, 8, 2>), dim3((N + 511) / 512, heads, B), dim3(512), ldsb, st, (const bf16_t *)q, (const bf16_t *)kv,
                               (const bf16_t *)out, (const bf16_t *)dout, lse, (bf16_t *)dq, delta, N, KV, heads, scale * kLog2e, scale);
        } else if (sra_wide(N)) {
            if ((rc = sra_raise_lds(sra_bwd_dq_b16<D, 8, 1>, r81))) return rc;
            hipLaunchKernelGGL((sra_bwd_dq_b16<D, 8, 1>), dim3((N + 255) / 256, heads, B), dim3(512), ldsb, st, (const bf16_t *)q, (const bf16_t *)kv,
                               (const bf16_t *)out, (const bf16_t *)dout, lse, (bf16_t *)dq, delta, N, KV, heads, scale * kLog2e, scale);
        } else {
            if ((rc = sra_raise_lds(sra_bwd_dq_b16<D, 4, 1>, r41))) return rc;
            hipLaunchKernelGGL((sra_bwd_dq_b16<D, 4, 1>), dim3((N + 127) / 128, heads, B), dim3(256), ldsb, st, (const bf16_t *)q, (const bf16_t *)kv,
                               (const bf16_t *)out, (const bf16_t *)dout, lse, (bf16_t *)dq, delta, N, KV, heads, scale * kLog2e, scale);
        }
        hipLaunchKernelGGL((sra_bwd_dkv_b16<D>), dim3(p.nchunk, heads, B), dim3(kSraThreads), 0, st, (const bf16_t *)q, (const bf16_t *)kv,
                           (const bf16_t *)dout, lse, delta, part, N, KV, heads, p.nchunk, p.qchunk, scale * kLog2e, scale);
        const size_t total_b = (size_t)B * KV * 2 * heads * (D / 4);
        hipLaunchKernelGGL((sra_dkv_reduce<T>), dim3((unsigned)((total_b + 255) / 256)), dim3(256), 0, st, part, (T *)dkv, B, KV, heads, D, p.nchunk);
        return (int)hipGetLastError();
    }
    bool split = false;
    if constexpr (sizeof(T) == 4 && D == 32) split = g_sra_split_bf16 != 0;
    if (split) {
        if constexpr (D == 32) {
            const int rows = (KV + 31) / 32 * 32;
            const size_t lds3 = (size_t)6 * rows * X3Geo<32>::KP + (size_t)rows * 36 * sizeof(float);
            static bool r82 = false, r81 = false, r41 = false;
            int rc;
            if (N >= 8192) {
                if ((rc = sra_raise_lds(sra_bwd_dq_x3<T, 8, 2>, r82))) return rc;
                hipLaunchKernelGGL((sra_bwd_dq_x3<T, 8, 2>), dim3((N + 511) / 512, heads, B), dim3(512), lds3, st, (const T *)q, (const T *)kv,
                                   (const T *)out, (const T *)dout, lse, (T *)dq, delta, N, KV, heads, scale * kLog2e, scale);
            } else if (sra_wide(N)) {
                if ((rc = sra_raise_lds(sra_bwd_dq_x3<T, 8, 1>, r81))) return rc;
                hipLaunchKernelGGL((sra_bwd_dq_x3<T, 8, 1>), dim3((N + 255) / 256, heads, B), dim3(512), lds3, st, (const T *)q, (const T *)kv,
                                   (const T *)out, (const T *)dout, lse, (T *)dq, delta, N, KV, heads, scale * kLog2e, scale);
            } else {
                if ((rc = sra_raise_lds(sra_bwd_dq_x3<T, 4, 1>, r41))) return rc;
                hipLaunchKernelGGL((sra_bwd_dq_x3<T, 4, 1>), dim3((N + 127) / 128, heads, B), dim3(256), lds3, st, (const T *)q, (const T *)kv,
                                   (const T *)out, (const T *)dout, lse, (T *)dq, delta, N, KV, heads, scale * kLog2e, scale);
            }
        }
    } else {
        static bool raised4 = false, raised8 = false;
        int rc = sra_wide(N) ? sra_raise_lds(sra_bwd_dq<T, D, 8>, raised8) : sra_raise_lds(sra_bwd_dq<T, D, 4>, raised4);
        if (rc) return rc;
        if (sra_wide(N))
            hipLaunchKernelGGL((sra_bwd_dq<T, D, 8>), dim3((N + 255) / 256, heads, B), dim3(512), lds, st, (const T *)q, (const T *)kv,
                               (const T *)out, (const T *)dout, lse, (T *)dq, delta, N, KV, heads, scale * kLog2e, scale);
        else
            hipLaunchKernelGGL((sra_bwd_dq<T, D, 4>), dim3((N + 127) / 128, heads, B), dim3(256), lds, st, (const T *)q, (const T *)kv,
                               (const T *)out, (const T *)dout, lse, (T *)dq, delta, N, KV, heads, scale * kLog2e, scale);
    }
    const dim3 gk(p.nchunk, heads, B);
    if (split)
        hipLaunchKernelGGL((sra_bwd_dkv_x3<T>), gk, dim3(kSraThreads), 0, st, (const T *)q, (const T *)kv, (const T *)dout, lse, delta, part, N, KV,
                           heads, p.nchunk, p.qchunk, scale * kLog2e, scale);
    else
        hipLaunchKernelGGL((sra_bwd_dkv<T, D>), gk, dim3(kSraThreads), 0, st, (const T *)q, (const T *)kv, (const T *)dout, lse, delta, part, N,
                           KV, heads, p.nchunk, p.qchunk, scale * kLog2e, scale);
    const size_t total = (size_t)B * KV * 2 * heads * (D / 4);
    hipLaunchKernelGGL((sra_dkv_reduce<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, part, (T *)dkv, B, KV, heads, D,
                       p.nchunk);
    return (int)hipGetLastError();
}

}  // namespace

int sra_tunable(const char *key, int set, int v) {
    int *p = nullptr;
    if (!strcmp(key, "sra_split_bf16")) p = &g_sra_split_bf16;
    else if (!strcmp(key, "sra_bf16_mfma")) p = &g_sra_bf16_mfma;
    bool flag = true;
    if (!p) {
        flag = false;
        if (!strcmp(key, "sra_wide_min")) p = &g_sra_wide_min;
        else if (!strcmp(key, "sra_dkv_wgs")) p = &g_sra_dkv_wgs;
    }
    if (!p) return SD_E_UNSUPPORTED;
    if (set) {
        if (flag ? (v != 0 && v != 1) : (v < 1 || v > (1 << 20))) return SD_E_SHAPE;
        *p = v;
        return SD_OK;
    }
    return *p;
}
}  // namespace sd

extern "C" {

#ifdef SD_SRA_STAMPS
void sd_debug_sra_stamps(void *buf) { sd::g_sra_stamps = static_cast<unsigned long long *>(buf); }   // diagnostic build only (not in the header)
#endif

int sd_sra_supported(int head_dim) { return (head_dim == 32 || head_dim == 64) ? 1 : 0; }

size_t sd_sra_workspace_bytes(int B, int N, int KV, int heads, int D) {
    if (B <= 0 || N <= 0 || KV <= 0 || heads <= 0 || D <= 0) return 0;
    const sd::SraPlan p = sd::sra_plan(B, N, KV, heads);
    const size_t delta = (((size_t)B * heads * N + 3) & ~(size_t)3) * sizeof(float);
    return delta + (size_t)B * heads * p.nchunk * 2 * KV * D * sizeof(float) + 16;
}

int sd_sra_fwd(const void *q, const void *kv, void *out, float *lse, int dtype, int B, int N, int KV, int heads, int D, float scale,
               void *stream) {
    int rc = sd::sra_check(q, kv, out, dtype, B, N, KV, heads, D);
    if (rc) return rc;
    if (!lse) return SD_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32)
        return D == 32 ? sd::sra_fwd_launch<float, 32>(q, kv, out, lse, B, N, KV, heads, scale, st)
                       : sd::sra_fwd_launch<float, 64>(q, kv, out, lse, B, N, KV, heads, scale, st);
    return D == 32 ? sd::sra_fwd_launch<sd::bf16_t, 32>(q, kv, out, lse, B, N, KV, heads, scale, st)
                   : sd::sra_fwd_launch<sd::bf16_t, 64>(q, kv, out, lse, B, N, KV, heads, scale, st);
}

int sd_sra_bwd(const void *q, const void *kv, const void *out, const void *dout, const float *lse, void *dq, void *dkv, int dtype, int B,
               int N, int KV, int heads, int D, float scale, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = sd::sra_check(q, kv, out, dtype, B, N, KV, heads, D);
    if (rc) return rc;
    if (!dout || !lse || !dq || !dkv || !workspace) return SD_E_NULL;
    if ((reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(dq) | reinterpret_cast<uintptr_t>(dkv) |
         reinterpret_cast<uintptr_t>(workspace)) & 15)
        return SD_E_ALIGN;
    if (workspace_bytes < sd_sra_workspace_bytes(B, N, KV, heads, D)) return SD_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32)
        return D == 32 ? sd::sra_bwd_launch<float, 32>(q, kv, out, dout, lse, dq, dkv, B, N, KV, heads, scale, workspace, st)
                       : sd::sra_bwd_launch<float, 64>(q, kv, out, dout, lse, dq, dkv, B, N, KV, heads, scale, workspace, st);
    return D == 32 ? sd::sra_bwd_launch<sd::bf16_t, 32>(q, kv, out, dout, lse, dq, dkv, B, N, KV, heads, scale, workspace, st)
                   : sd::sra_bwd_launch<sd::bf16_t, 64>(q, kv, out, dout, lse, dq, dkv, B, N, KV, heads, scale, workspace, st);
}

}  // extern "C"
