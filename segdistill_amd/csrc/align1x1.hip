// align1x1.hip -- the trainable 1x1 feature-alignment projection of the student feature
// (documented at reference opts.py:25-27, commented-out at losses.py:258,332-333)
// as dense GEMMs on the gfx950 matrix cores.
//
//   forward   Y_b [Ct x P] = W [Ct x Cs] . X_b [Cs x P] + bias          (P = h*w pixels, NCHW)
//   bwd-data  dX_b [Cs x P] = W^T [Cs x Ct] . dY_b [Ct x P]
//   bwd-wgt   dW [Ct x Cs]  = sum_b dY_b [Ct x P] . X_b^T [P x Cs]       (split over (b, pixel chunks))
//   bwd-bias  db [Ct]       = sum_{b,p} dY_b[:, p]
//
// One kernel template serves all three products: C[M x N] = A[M x K] . B[K x N] with either
// operand given K-major ([K][X], X contiguous) or X-major ([X][K], K contiguous).  fp32 operands use
// v_mfma_f32_32x32x2_f32 (exact f32, bit-equal to an fmaf chain; 157 TFLOP/s dense peak); bf16
// STORAGE is widened to f32 on the way into LDS (same instruction, fp32 accumulation).
//
// Tiling for wave64: 256 threads = 4 waves in a 2x2 arrangement, each wave owns a 64x64 output
// tile = 2x2 MFMA 32x32 accumulators (64 acc registers); workgroup tile 128x128, K step 16.  Both
// operand tiles live K-major in LDS ([16][128+4] floats) so the per-lane operand reads
// (lane&31 -> row/col, lane>>5 -> k) are conflict-free ds_read_b32.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cgd_device.h"

namespace sd {

// fp32 operands go through the software-pipelined kernel of token_gemm.hip (one barrier per k-step, loads in flight behind the MFMAs,
// 16-byte row-major epilogue); the single-buffered gemm_mfma_* kernels below remain for bf16 STORAGE in NCHW layout (direct callers only:
// config 5's token-major taps take the token Linear + csrc/cgd_tok.hip route) and for the generic split-K Linear weight gradient.
int align_f32_fwd(const float *X, const float *W, const float *bias, float *Y, int B, int Cs, int Ct, long P, hipStream_t st);
int align_f32_bwd_data(const float *dY, const float *W, float *dX, int B, int Cs, int Ct, long P, hipStream_t st);
int align_f32_bwd_weight_slabs(const float *dY, const float *X, float *slabs, int B, int Cs, int Ct, long P, int nsplit, int klen, hipStream_t st);

namespace {

constexpr int BM = 128, BN = 128, BK = 16, PITCH = BM + 4;
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <typename T> __device__ __forceinline__ float ld1(const T *p);
template <> __device__ __forceinline__ float ld1<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float ld1<bf16_t>(const bf16_t *p) { return __uint_as_float((unsigned)p->bits << 16); }
template <typename T> __device__ __forceinline__ void st1(T *p, float v);
template <> __device__ __forceinline__ void st1<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void st1<bf16_t>(bf16_t *p, float v) { p->bits = f32_to_bf16(v); }

// Stage a [BK x 128] K-major tile into LDS from a source that is either K-major (src[k*ld + x]) or
// X-major (src[x*ld + k]).  Out-of-range elements are zero.  256 threads, 2048 elements -> 8 each.
template <typename T, bool KMAJOR>
__device__ __forceinline__ void stage(float (*dst)[PITCH], const T *__restrict__ src, long ld, int x0, int xmax, int k0, int kmax) {
    const int t = threadIdx.x;
    if constexpr (KMAJOR) {
        // thread -> (k = t/16, 8 consecutive x starting at (t%16)*8): coalesced along x
        const int k = t >> 4, xb = (t & 15) * 8;
        const bool kin = (k0 + k) < kmax;
        const T *row = src + (long)(k0 + k) * ld + x0 + xb;
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[k][xb + i] = (kin && x0 + xb + i < xmax) ? ld1<T>(row + i) : 0.f;
    } else {
        // thread -> (x = t/2, 8 consecutive k starting at (t%2)*8): each lane reads 32 contiguous bytes of its row
        const int x = t >> 1, kb = (t & 1) * 8;
        const bool xin = (x0 + x) < xmax;
        const T *row = src + (long)(x0 + x) * ld + k0 + kb;
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[kb + i][x] = (xin && k0 + kb + i < kmax) ? ld1<T>(row + i) : 0.f;
    }
}

// C_z[M x N] (+)= A_z . B_z over k in [k_begin, k_end);  z = blockIdx.z = batch*nsplit + split.
template <typename TA, typename TB, typename TC, bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(256) void gemm_mfma_f32(const TA *__restrict__ A, const TB *__restrict__ B, TC *__restrict__ C,
                                                      const float *__restrict__ bias, int M, int N, int K, long lda, long ldb, long ldc,
                                                      long strideA, long strideB, long strideC, int nsplit, int klen) {
    __shared__ float As[BK][PITCH];
    __shared__ float Bs[BK][PITCH];
    const int z = blockIdx.z, batch = z / nsplit, split = z - batch * nsplit;
    const int k_begin = split * klen, k_end = min(K, k_begin + klen);
    A += (long)batch * strideA;
    B += (long)batch * strideB;
    C += (long)z * strideC;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int r = lane & 31, kh = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // which of this wave's four 32x32 sub-tiles hold real rows/columns (skinny problems leave most of them empty;
    // issuing f32 MFMAs on zero padding would make a 32x32 weight gradient MFMA-bound instead of HBM-bound)
    const bool am0 = m0 + wm < M, am1 = m0 + wm + 32 < M;
    const bool bn0 = n0 + wn < N, bn1 = n0 + wn + 32 < N;
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        stage<TA, A_KMAJOR>(As, A, lda, m0, M, k0, k_end);
        stage<TB, B_KMAJOR>(Bs, B, ldb, n0, N, k0, k_end);
        __syncthreads();
        if (am0 && bn0) {
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                const float a0 = As[kk + kh][wm + r], a1 = As[kk + kh][wm + 32 + r];
                const float b0 = Bs[kk + kh][wn + r], b1 = Bs[kk + kh][wn + 32 + r];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                if (bn1) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                if (am1) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                if (am1 && bn1) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // C/D layout of the 32x32 accumulator: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                if (m < M && n < N) {
                    float v = acc[i][j][e];
                    if (bias) v += bias[m];
                    st1<TC>(C + (long)m * ldc + n, v);
                }
            }
        }
}

// ---- bf16-storage variant: v_mfma_f32_32x32x16_bf16 (16x the math rate of the f32-input MFMA) -------------------------------------
// Same tiling, arguments and split-K protocol as gemm_mfma_f32; used when the activations are stored in bf16 (config 5).  Products
// of bf16 values are exact in fp32 and accumulation is fp32, so the only numerical difference to the f32 kernel on widened inputs
// is that an fp32 master operand (the align weight) is rounded to bf16 on its way into LDS -- what autocast does for a conv.
// LDS tiles are [128 rows][32 k] bf16, k contiguous (pitch 40 -> conflict-free ds_read_b128): lane l = (r = l&31, h = l>>5) takes
// the 8 values k = 8h..8h+7 of row r as its A (or B) fragment.  A source whose k axis is strided in memory (KMAJOR: src[k*ld + x])
// is transposed on the way in: 16-byte loads along x, eight 2-byte LDS stores.
constexpr int BK16 = 64, PITCH16 = 72;   // 144-byte rows: 16-byte aligned, ds_read_b128 of 8 consecutive rows covers all 32 banks
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <typename T> __device__ __forceinline__ unsigned short to_bf16_bits(const T *p);
template <> __device__ __forceinline__ unsigned short to_bf16_bits<float>(const float *p) { return f32_to_bf16(*p); }
template <> __device__ __forceinline__ unsigned short to_bf16_bits<bf16_t>(const bf16_t *p) { return p->bits; }

// 8 consecutive source elements -> 8 bf16 (one 16-byte / two 16-byte loads; the pointer must be 16-byte aligned)
template <typename T> __device__ __forceinline__ uint4 load8_bf16(const T *p);
template <> __device__ __forceinline__ uint4 load8_bf16<bf16_t>(const bf16_t *p) { return *reinterpret_cast<const uint4 *>(p); }
template <> __device__ __forceinline__ uint4 load8_bf16<float>(const float *p) {
    const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
    uint4 o;
    o.x = (unsigned)f32_to_bf16(a.x) | ((unsigned)f32_to_bf16(a.y) << 16);
    o.y = (unsigned)f32_to_bf16(a.z) | ((unsigned)f32_to_bf16(a.w) << 16);
    o.z = (unsigned)f32_to_bf16(b.x) | ((unsigned)f32_to_bf16(b.y) << 16);
    o.w = (unsigned)f32_to_bf16(b.z) | ((unsigned)f32_to_bf16(b.w) << 16);
    return o;
}

// Stage a [128 x 64] tile (rows x0.., k k0..) into dst[128][PITCH16]; out-of-range elements are zero.  256 threads, 32 elements each.
// `vec`: the source rows are 16-byte aligned and every 8-element group lies on a 16-byte boundary (workgroup-uniform).
template <typename T, bool KMAJOR>
__device__ __forceinline__ void stage16(unsigned short (*dst)[PITCH16], const T *__restrict__ src, long ld, int x0, int xmax, int k0, int kmax,
                                        bool vec) {
    const int t = threadIdx.x;
    if constexpr (KMAJOR) {
        // src[k*ld + x]: a thread takes TWO consecutive k for 8 consecutive x (two 16-byte loads along x) and writes eight 4-byte
        // (k, k+1) pairs -- the transpose into the k-contiguous LDS image; two such blocks per thread
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int k = 2 * ((t >> 4) + 16 * it), xb = (t & 15) * 8;
            const T *row = src + (long)(k0 + k) * ld + x0 + xb;
            if (vec && k0 + k + 1 < kmax && x0 + xb + 7 < xmax) {
                const uint4 a = load8_bf16<T>(row), c = load8_bf16<T>(row + ld);
                const unsigned av[4] = {a.x, a.y, a.z, a.w}, cv[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    *reinterpret_cast<unsigned *>(&dst[xb + 2 * i][k]) = (av[i] & 0xffffu) | (cv[i] << 16);
                    *reinterpret_cast<unsigned *>(&dst[xb + 2 * i + 1][k]) = (av[i] >> 16) | (cv[i] & 0xffff0000u);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const bool xin = x0 + xb + i < xmax;
                    dst[xb + i][k] = (xin && k0 + k < kmax) ? to_bf16_bits<T>(row + i) : (unsigned short)0;
                    dst[xb + i][k + 1] = (xin && k0 + k + 1 < kmax) ? to_bf16_bits<T>(row + ld + i) : (unsigned short)0;
                }
            }
        }
    } else {
        // src[x*ld + k]: thread -> (x = t/2, 32 consecutive k starting at (t%2)*32): four 16-byte loads, four 16-byte LDS stores
        const int x = t >> 1, kb = (t & 1) * 32;
        const bool xin = (x0 + x) < xmax;
        const T *row = src + (long)(x0 + x) * ld + k0 + kb;
        if (vec && xin && k0 + kb + 31 < kmax) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4 *>(&dst[x][kb + 8 * i]) = load8_bf16<T>(row + 8 * i);
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i) dst[x][kb + i] = (xin && k0 + kb + i < kmax) ? to_bf16_bits<T>(row + i) : (unsigned short)0;
        }
    }
}

template <typename T> __device__ __forceinline__ bool rows_vectorisable(const T *src, long ld, int k_begin) {
    // every row start and every 8-group start 16-byte aligned: base, leading dimension and the k origin of this split
    constexpr int per16 = 16 / (int)sizeof(T);
    return (reinterpret_cast<uintptr_t>(src) & 15) == 0 && ld % 8 == 0 && k_begin % 8 == 0 && (per16 <= 8);
}

template <typename TA, typename TB, typename TC, bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(256) void gemm_mfma_bf16(const TA *__restrict__ A, const TB *__restrict__ B, TC *__restrict__ C,
                                                       const float *__restrict__ bias, int M, int N, int K, long lda, long ldb, long ldc,
                                                       long strideA, long strideB, long strideC, int nsplit, int klen) {
    __shared__ __attribute__((aligned(16))) unsigned short As[BM][PITCH16];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[BN][PITCH16];
    const int z = blockIdx.z, batch = z / nsplit, split = z - batch * nsplit;
    const int k_begin = split * klen, k_end = min(K, k_begin + klen);
    A += (long)batch * strideA;
    B += (long)batch * strideB;
    C += (long)z * strideC;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int r = lane & 31, kh = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const bool am0 = m0 + wm < M, am1 = m0 + wm + 32 < M;
    const bool bn0 = n0 + wn < N, bn1 = n0 + wn + 32 < N;
    const bool vecA = rows_vectorisable<TA>(A, lda, k_begin), vecB = rows_vectorisable<TB>(B, ldb, k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += BK16) {
        stage16<TA, A_KMAJOR>(As, A, lda, m0, M, k0, k_end, vecA);
        stage16<TB, B_KMAJOR>(Bs, B, ldb, n0, N, k0, k_end, vecB);
        __syncthreads();
        if (am0 && bn0) {
#pragma unroll
            for (int kq = 0; kq < BK16; kq += 16) {
                const bf16x8 a0 = *reinterpret_cast<const bf16x8 *>(&As[wm + r][kq + 8 * kh]);
                const bf16x8 a1 = *reinterpret_cast<const bf16x8 *>(&As[wm + 32 + r][kq + 8 * kh]);
                const bf16x8 b0 = *reinterpret_cast<const bf16x8 *>(&Bs[wn + r][kq + 8 * kh]);
                const bf16x8 b1 = *reinterpret_cast<const bf16x8 *>(&Bs[wn + 32 + r][kq + 8 * kh]);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
                if (bn1) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
                if (am1) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
                if (am1 && bn1) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                if (m < M && n < N) {
                    float v = acc[i][j][e];
                    if (bias) v += bias[m];
                    st1<TC>(C + (long)m * ldc + n, v);
                }
            }
        }
}

// dispatch: bf16 activations -> bf16 MFMA kernel, otherwise the exact f32 one (same arguments)
template <typename TA, typename TB, typename TC, bool A_KMAJOR, bool B_KMAJOR, typename... Args>
void launch_gemm(dim3 grid, hipStream_t st, Args... args) {
    if constexpr (sizeof(TA) == 2 || sizeof(TB) == 2)
        hipLaunchKernelGGL((gemm_mfma_bf16<TA, TB, TC, A_KMAJOR, B_KMAJOR>), grid, dim3(256), 0, st, args...);
    else
        hipLaunchKernelGGL((gemm_mfma_f32<TA, TB, TC, A_KMAJOR, B_KMAJOR>), grid, dim3(256), 0, st, args...);
}

// out[i] = sum_z slabs[z][i]  (deterministic split-K combine), i < n
template <typename TC>
__global__ __launch_bounds__(256) void slab_reduce(const float *__restrict__ slabs, TC *__restrict__ out, long n, int nz) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float acc = 0.f;
    for (int z = 0; z < nz; ++z) acc += slabs[(long)z * n + i];
    st1<TC>(out + i, acc);
}

// out[i] = sum_z slabs[z][i] for MANY slabs: block = 16 slab-groups x 16 outputs, LDS combine (deterministic order)
__global__ __launch_bounds__(256) void slab_reduce_wide(const float *__restrict__ slabs, float *__restrict__ out, long n, int nz,
                                                         float *__restrict__ out2 = nullptr, long n_first = 0) {
    __shared__ float red[16][17];
    const int o = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const long i = (long)blockIdx.x * 16 + o;
    float acc = 0.f;
    if (i < n)
        for (int z = grp; z < nz; z += 16) acc += slabs[(long)z * n + i];
    red[grp][o] = acc;
    __syncthreads();
    if (grp == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < 16; ++g2) t += red[g2][o];
        if (out2 && i >= n_first) out2[i - n_first] = t;   // tail of the slab (bias partials) goes to a second output
        else out[i] = t;
    }
}

// db[c] = sum over (b, p) of dY[b][c][p]; one workgroup per channel.  Rows are walked with eight 16-byte loads per lane in flight
// (the scalar one-load-at-a-time form ran at the load latency: 123 us for the [8, 150, 16384] bf16 class planes of config 5, 0.3 TB/s);
// the summation order is fixed by the geometry, not by timing.
template <typename T>
__global__ __launch_bounds__(256) void bias_grad(const T *__restrict__ dY, float *__restrict__ db, int B, int C, long P) {
    constexpr int VE = 16 / sizeof(T), U = 8;           // elements per 16-byte load; loads in flight per lane
    typedef unsigned int raw_t __attribute__((ext_vector_type(4)));
    const int c = blockIdx.x;
    float acc = 0.f;
    const bool vec = P % VE == 0 && (reinterpret_cast<uintptr_t>(dY) & 15) == 0;
    for (int b = 0; b < B; ++b) {
        const T *row = dY + ((long)b * C + c) * P;
        long p0 = 0;
        if (vec) {
            const long nv = P / VE, full = nv / (256 * U) * (256 * U);
            for (long v0 = 0; v0 < full; v0 += 256 * U) {
                raw_t r[U];
#pragma unroll
                for (int u = 0; u < U; ++u) r[u] = *reinterpret_cast<const raw_t *>(row + (v0 + u * 256 + threadIdx.x) * VE);
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (sizeof(T) == 4) acc += __uint_as_float(r[u][i]);
                        else acc += __uint_as_float(r[u][i] << 16) + __uint_as_float(r[u][i] & 0xffff0000u);
                    }
            }
            p0 = full * VE;
        }
        for (long p = p0 + threadIdx.x; p < P; p += 256) acc += ld1<T>(row + p);
    }
    __shared__ float part[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) db[c] = part[0] + part[1] + part[2] + part[3];
}

int wgrad_splits(int B, long P) {
    // tiles(M,N) are few (e.g. 4), so the pixel axis is split -- but every split writes (and the combine re-reads) a Ct x Cs slab, so no
    // further than ~64 slabs in all: 4 tiles x 64 = one workgroup per CU, 16 k-steps each at config 4
    int per_img = (int)((P + 2047) / 2048);
    if (per_img < 1) per_img = 1;
    while ((long)B * per_img < 64 && P / per_img > 512) per_img *= 2;
    return per_img;
}

template <typename T>
int align_fwd(const void *X, const void *W, const float *bias, void *Y, int B, int Cs, int Ct, long P, hipStream_t st) {
    dim3 grid((unsigned)((P + BN - 1) / BN), (Ct + BM - 1) / BM, B);
    launch_gemm<float, T, T, false, true>(grid, st, (const float *)W, (const T *)X, (T *)Y, bias, Ct,
                       (int)P, Cs, (long)Cs, P, P, 0L, (long)Cs * P, (long)Ct * P, 1, Cs);
    return (int)hipGetLastError();
}

template <typename T>
int align_bwd_data(const void *dY, const void *W, void *dX, int B, int Cs, int Ct, long P, hipStream_t st) {
    // A = W^T given as W [K=Ct][M=Cs] -> K-major
    dim3 grid((unsigned)((P + BN - 1) / BN), (Cs + BM - 1) / BM, B);
    launch_gemm<float, T, T, true, true>(grid, st, (const float *)W, (const T *)dY, (T *)dX, nullptr,
                       Cs, (int)P, Ct, (long)Cs, P, P, 0L, (long)Ct * P, (long)Cs * P, 1, Ct);
    return (int)hipGetLastError();
}

template <typename T>
int align_bwd_weight(const void *dY, const void *X, float *dW, float *db, void *ws, size_t ws_bytes, int B, int Cs, int Ct, long P,
                     hipStream_t st) {
    int nsplit = wgrad_splits(B, P);
    const int klen = (int)(((P + nsplit - 1) / nsplit + 31) / 32 * 32);    // multiple of the k-step of either kernel
    nsplit = (int)((P + klen - 1) / klen);                                  // after rounding: every split owns at least one pixel
    const int nz = B * nsplit;
    const long slab = (long)Ct * Cs;
    if (ws_bytes < (size_t)nz * slab * sizeof(float) || (reinterpret_cast<uintptr_t>(ws) & 15)) return SD_E_WORKSPACE;
    float *slabs = static_cast<float *>(ws);
    if constexpr (sizeof(T) == 4) {
        const int rc = align_f32_bwd_weight_slabs((const float *)dY, (const float *)X, slabs, B, Cs, Ct, P, nsplit, klen, st);
        if (rc) return rc;
    } else {
        // A = dY_b [M=Ct][K=P] (X-major), B = X_b^T given as X_b [N=Cs][K=P] (X-major); C = slab z
        dim3 grid((Cs + BN - 1) / BN, (Ct + BM - 1) / BM, nz);
        launch_gemm<T, T, float, false, false>(grid, st, (const T *)dY, (const T *)X, slabs, nullptr, Ct,
                           Cs, (int)P, P, P, (long)Cs, (long)Ct * P, (long)Cs * P, slab, nsplit, klen);
    }
    hipLaunchKernelGGL((slab_reduce<float>), dim3((unsigned)((slab + 255) / 256)), dim3(256), 0, st, slabs, dW, slab, nz);
    if (db) hipLaunchKernelGGL((bias_grad<T>), dim3(Ct), dim3(256), 0, st, (const T *)dY, db, B, Ct, P);
    return (int)hipGetLastError();
}

// Skinny weight gradient straight from global memory: the MFMA 32x32x2 operand layout (lane&31 -> row/col,
// lane>>5 -> k) IS a coalesced access pattern for K-major operands (32 consecutive floats of token t for lanes 0-31,
// of token t+1 for lanes 32-63), so no LDS staging / barriers are needed at all.  Each wave owns one 64x64 output
// region (2x2 accumulators); a workgroup covers up to four regions, and when the problem has fewer than four the
// spare waves split the token range instead (more slabs).  slab index = blockIdx.x * ksubs + ksub.
// Branch-free inner loop: TM / TN = number of 32-blocks of the region that exist (1 or 2; wave-uniform, compile-time here), column
// indices are CLAMPED into the matrix and the values of lanes beyond it zeroed by a select, the token range is walked in full steps
// of 2*U tokens (no predicate at all) plus one clamped + masked tail step.  All loads of a step are plain back-to-back
// global_load_dword instructions, and the loads of step i+1 are issued before the MFMAs of step i (two register sets): the grid
// holds about one wave per SIMD (more waves = more slabs to combine), so nothing else hides the HBM latency.
template <int TM, int TN, int U>
struct WgradFrag {
    float a[TM][U], b[TN][U];
};

template <typename T, int TM, int TN, int U, bool TAIL>
__device__ __forceinline__ void wgrad_load(WgradFrag<TM, TN, U> &f, const T *__restrict__ dY, const T *__restrict__ X, long t0, int kh,
                                           long k_end, int M, int N, const int (&ca)[2], const int (&cb)[2], const bool (&am)[2],
                                           const bool (&bn)[2]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        long t = t0 + 2 * u + kh;
        bool in = true;
        if (TAIL) {
            in = t < k_end;
            t = in ? t : k_end - 1;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float v = ld1<T>(dY + t * M + ca[i]);
            f.a[i][u] = (in && am[i]) ? v : 0.f;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float v = ld1<T>(X + t * N + cb[j]);
            f.b[j][u] = (in && bn[j]) ? v : 0.f;
        }
    }
}

template <int TM, int TN, int U>
__device__ __forceinline__ void wgrad_multiply(const WgradFrag<TM, TN, U> &f, f32x16 (&acc)[2][2], bool do_bias, float (&bs)[2]) {
    if (do_bias) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < TM; ++i) bs[i] += f.a[i][u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][u], f.b[j][u], acc[i][j], 0, 0, 0);
}

template <typename T, int TM, int TN>
__device__ __forceinline__ void wgrad_walk(const T *__restrict__ dY, const T *__restrict__ X, long k_begin, long k_end, int kh, int M, int N,
                                           const int (&ca)[2], const int (&cb)[2], const bool (&am)[2], const bool (&bn)[2],
                                           f32x16 (&acc)[2][2], bool do_bias, float (&bs)[2]) {
    constexpr int U = 16;   // token pairs per step: up to 64 dword loads in flight per lane and register set
    WgradFrag<TM, TN, U> fa, fb;
    const long full_end = k_begin + (k_end - k_begin) / (2 * U) * (2 * U);
    long t0 = k_begin;
    if (t0 < full_end) {
        wgrad_load<T, TM, TN, U, false>(fa, dY, X, t0, kh, k_end, M, N, ca, cb, am, bn);
        while (true) {
            const long t1 = t0 + 2 * U;
            if (t1 >= full_end) { wgrad_multiply<TM, TN, U>(fa, acc, do_bias, bs); break; }
            wgrad_load<T, TM, TN, U, false>(fb, dY, X, t1, kh, k_end, M, N, ca, cb, am, bn);
            wgrad_multiply<TM, TN, U>(fa, acc, do_bias, bs);
            const long t2 = t1 + 2 * U;
            if (t2 >= full_end) { wgrad_multiply<TM, TN, U>(fb, acc, do_bias, bs); break; }
            wgrad_load<T, TM, TN, U, false>(fa, dY, X, t2, kh, k_end, M, N, ca, cb, am, bn);
            wgrad_multiply<TM, TN, U>(fb, acc, do_bias, bs);
            t0 = t2;
        }
    }
    if (full_end < k_end) {
        wgrad_load<T, TM, TN, U, true>(fa, dY, X, full_end, kh, k_end, M, N, ca, cb, am, bn);
        wgrad_multiply<TM, TN, U>(fa, acc, do_bias, bs);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void linear_wgrad_direct(const T *__restrict__ dY, const T *__restrict__ X, float *__restrict__ slabs, int M,
                                                            int N, long Tn, int klen, int regions_m, int regions_n, int regions_per_wg,
                                                            int with_bias) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform values stay scalar
    const int r = lane & 31, kh = lane >> 5;
    const int ksubs = 4 / regions_per_wg;
    const int region = blockIdx.y * regions_per_wg + wave % regions_per_wg;
    const int ksub = wave / regions_per_wg;
    if (region >= regions_m * regions_n) return;
    const int m0 = (region / regions_n) * 64, n0 = (region % regions_n) * 64;
    const long k_begin = (long)blockIdx.x * klen + (long)ksub * (klen / ksubs);
    const long k_end = min(Tn, ksub == ksubs - 1 ? (long)(blockIdx.x + 1) * klen : k_begin + klen / ksubs);
    const bool am[2] = {m0 + r < M, m0 + 32 + r < M}, bn[2] = {n0 + r < N, n0 + 32 + r < N};
    const bool am0 = am[0], am1 = am[1];
    const int ca[2] = {min(m0 + r, M - 1), min(m0 + 32 + r, M - 1)}, cb[2] = {min(n0 + r, N - 1), min(n0 + 32 + r, N - 1)};
    const bool tm1 = m0 + 32 < M, tn1 = n0 + 32 < N;  // wave-uniform: does the second 32-block exist at all
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // bias gradient = column sums of dY: the A operands ARE dY, so the waves of the first n-region add them up on the side
    const bool do_bias = with_bias && (region % regions_n) == 0;
    float bs[2] = {0.f, 0.f};
    if (k_begin < k_end) {
        if (tm1 && tn1) wgrad_walk<T, 2, 2>(dY, X, k_begin, k_end, kh, M, N, ca, cb, am, bn, acc, do_bias, bs);
        else if (tm1) wgrad_walk<T, 2, 1>(dY, X, k_begin, k_end, kh, M, N, ca, cb, am, bn, acc, do_bias, bs);
        else if (tn1) wgrad_walk<T, 1, 2>(dY, X, k_begin, k_end, kh, M, N, ca, cb, am, bn, acc, do_bias, bs);
        else wgrad_walk<T, 1, 1>(dY, X, k_begin, k_end, kh, M, N, ca, cb, am, bn, acc, do_bias, bs);
    }
    float bs0 = bs[0], bs1 = bs[1];
    const long slab_elems = (long)M * N + (with_bias ? M : 0);   // a slab = the M x N partial, then (optionally) M bias partials
    float *C = slabs + ((long)blockIdx.x * ksubs + ksub) * slab_elems;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                if (m < M && n < N) C[(long)m * N + n] = acc[i][j][e];
            }
        }
    if (do_bias) {
        bs0 += __shfl_xor(bs0, 32, 64);   // the two token parities
        bs1 += __shfl_xor(bs1, 32, 64);
        if (kh == 0) {
            if (am0) C[(long)M * N + m0 + r] = bs0;
            if (am1) C[(long)M * N + m0 + 32 + r] = bs1;
        }
    }
}

// dW[M x N] = dY^T . X for token-major dY [T][M], X [T][N] (the weight gradient of nn.Linear):
// both operands K-major, K = T split over workgroups, partial slabs combined deterministically.
struct WgradPlan {
    bool direct, tn;
    int nsplit, klen, regions_m, regions_n, regions_per_wg, nslabs;
};

WgradPlan linear_wgrad_plan(long T, int M, int N, bool bf16 = false) {
    WgradPlan p{};
    p.regions_m = (M + 63) / 64;
    p.regions_n = (N + 63) / 64;
    const int regions = p.regions_m * p.regions_n;
    p.direct = T >= 8192 && regions <= 16;      // many tokens, small weight: the HBM-bound tall-skinny case
    // bf16 storage halves the bytes while the direct kernel still multiplies on the f32-input MFMA: beyond ~32 flop/byte
    // (2MN/(M+N) per bf16 element pair) it is compute-bound there, and the bf16-MFMA split-K GEMM wins
    if (bf16 && (long)M * N > 32L * (M + N)) p.direct = false;
    if (p.direct) {
        p.regions_per_wg = regions >= 4 ? 4 : (regions >= 2 ? 2 : 1);
        const int wg_y = (regions + p.regions_per_wg - 1) / p.regions_per_wg;
        long nsplit = 256 / wg_y;  // x (4 / regions_per_wg) k-sub-ranges: ~1024 waves, and few enough slabs that the combine stays cheap
        if (nsplit > T / 128) nsplit = T / 128;
        {
            const long cap = wgrad_slab_cap(T, M, N, bf16 ? 2 : 4) / (4 / p.regions_per_wg);     // slabs = nsplit x k-sub-ranges
            if (nsplit > cap) nsplit = cap;
        }
        if (nsplit < 1) nsplit = 1;
        long klen = ((T + nsplit - 1) / nsplit + 63) / 64 * 64;  // multiple of 64: every k-sub-range stays even-aligned
        p.nsplit = (int)((T + klen - 1) / klen);
        p.klen = (int)klen;
        p.nslabs = p.nsplit * (4 / p.regions_per_wg);
        return p;
    }
    if (bf16 && M % 8 == 0 && N % 8 == 0) {
        // round 4: both operands are token-major (the reduction index is the slow one) -- csrc/wgrad_tn.hip copies the tiles as they are and
        // transposes in the LDS read (ds_read_b64_tr_b16); its own split plan (fewer, longer splits: fewer slabs to combine)
        p.tn = true;
        wgrad_tn_plan(T, M, N, &p.nsplit, &p.klen);
        p.nslabs = p.nsplit;
        return p;
    }
    const long tiles = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    long nsplit = 1024 / tiles;
    if (nsplit > T / 256) nsplit = T / 256;
    if (nsplit > wgrad_slab_cap(T, M, N, bf16 ? 2 : 4)) nsplit = wgrad_slab_cap(T, M, N, bf16 ? 2 : 4);
    if (nsplit < 1) nsplit = 1;
    long klen = ((T + nsplit - 1) / nsplit + BK - 1) / BK * BK;
    p.nsplit = (int)((T + klen - 1) / klen);
    p.klen = (int)klen;
    p.nslabs = p.nsplit;
    return p;
}

template <typename T>
int linear_wgrad(const void *dY, const void *X, float *dW, float *dbias, void *ws, size_t ws_bytes, long Tn, int M, int N, hipStream_t st) {
    const WgradPlan p = linear_wgrad_plan(Tn, M, N, sizeof(T) == 2);
    if (dbias && !p.direct) return SD_E_UNSUPPORTED;   // ask sd_linear_wgrad_fuses_bias_dtype() first
    const long slab = (long)M * N + (dbias ? M : 0);
    const bool tn = p.tn && wgrad_tn_supported(Tn, M, N, dY, X);
    if (!p.direct && p.nsplit == 1) {
        if (tn) return wgrad_tn_launch(dY, X, dW, Tn, M, N, 1, p.klen, st);
        dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, 1);
        launch_gemm<T, T, float, true, true>(grid, st, (const T *)dY, (const T *)X, dW, nullptr, M, N,
                           (int)Tn, (long)M, (long)N, (long)N, 0L, 0L, 0L, 1, (int)Tn);
        return (int)hipGetLastError();
    }
    if (ws_bytes < (size_t)p.nslabs * slab * sizeof(float) || (reinterpret_cast<uintptr_t>(ws) & 15)) return SD_E_WORKSPACE;
    float *slabs = static_cast<float *>(ws);
    if (p.direct) {
        const int regions = p.regions_m * p.regions_n;
        dim3 grid(p.nsplit, (regions + p.regions_per_wg - 1) / p.regions_per_wg);
        hipLaunchKernelGGL((linear_wgrad_direct<T>), grid, dim3(256), 0, st, (const T *)dY, (const T *)X, slabs, M, N, Tn, p.klen, p.regions_m,
                           p.regions_n, p.regions_per_wg, dbias ? 1 : 0);
    } else if (tn) {
        int rc = wgrad_tn_launch(dY, X, slabs, Tn, M, N, p.nsplit, p.klen, st);
        if (rc) return rc;
    } else {
        dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, p.nsplit);
        launch_gemm<T, T, float, true, true>(grid, st, (const T *)dY, (const T *)X, slabs, nullptr, M, N,
                           (int)Tn, (long)M, (long)N, (long)N, 0L, 0L, slab, p.nsplit, p.klen);
    }
    hipLaunchKernelGGL(slab_reduce_wide, dim3((unsigned)((slab + 15) / 16)), dim3(256), 0, st, slabs, dW, slab, p.nslabs, dbias, (long)M * N);
    return (int)hipGetLastError();
}

// Y[M x N] = X[M x K] . W[N x K]^T + bias[N] for LONG K and a small output (the SR-attention patch projection:
// K = r*r*C up to 4096, N = C, M = B*256 tokens): the library picks an un-split 32x64 tiling with 64 workgroups;
// here K is split over workgroups (both operands X-major, f32 MFMA slabs) and the combine adds the bias.
__global__ __launch_bounds__(256) void slab_reduce_bias(const float *__restrict__ slabs, float *__restrict__ out, const float *__restrict__ bias,
                                                         long n, int nz, int N) {
    __shared__ float red[16][17];
    const int o = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const long i = (long)blockIdx.x * 16 + o;
    float acc = 0.f;
    if (i < n)
        for (int z = grp; z < nz; z += 16) acc += slabs[(long)z * n + i];
    red[grp][o] = acc;
    __syncthreads();
    if (grp == 0 && i < n) {
        float t = bias ? bias[i % N] : 0.f;
#pragma unroll
        for (int g2 = 0; g2 < 16; ++g2) t += red[g2][o];
        out[i] = t;
    }
}


int check_align(const void *a, const void *b, const void *c, int dtype, int B, int Cs, int Ct, int h, int w) {
    if (!a || !b || !c) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (B <= 0 || Cs <= 0 || Ct <= 0 || h <= 0 || w <= 0 || B > 65535) return SD_E_SHAPE;
    return SD_OK;
}

}  // namespace
}  // namespace sd

extern "C" {

size_t sd_align1x1_workspace_bytes(int B, int Cs, int Ct, int h, int w) {
    if (B <= 0 || Cs <= 0 || Ct <= 0 || h <= 0 || w <= 0) return 0;
    const long P = (long)h * w;
    return (size_t)B * sd::wgrad_splits(B, P) * Ct * Cs * sizeof(float) + 16;
}

size_t sd_linear_wgrad_workspace_bytes(long tokens, int out_features, int in_features) {
    if (tokens <= 0 || out_features <= 0 || in_features <= 0) return 0;
    const sd::WgradPlan p = sd::linear_wgrad_plan(tokens, out_features, in_features), q = sd::linear_wgrad_plan(tokens, out_features, in_features, true);
    const int nslabs = p.nslabs > q.nslabs ? p.nslabs : q.nslabs;    // either storage type
    return (size_t)nslabs * ((size_t)out_features * in_features + out_features) * sizeof(float) + 16;
}

size_t sd_linear_nchw_workspace_bytes(int B, long P, int in_features, int out_features) {
    return sd::linear_nchw_workspace_bytes(B, P, in_features, out_features);
}

int sd_linear_nchw_fwd(const void *X, const float *W, const float *bias, void *Y, int dtype, int B, long P, int in_features, int out_features,
                       void *stream) {
    if (!X || !W || !Y) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (B <= 0 || P <= 0 || in_features <= 0 || out_features <= 0 || B > 65535) return SD_E_SHAPE;
    if (dtype == SD_F32) return sd::linear_nchw_f32_fwd((const float *)X, W, bias, (float *)Y, B, P, in_features, out_features, stream);
    // bf16 storage: A = W [m][k] (fp32 master, rounded on its way into LDS), B = tokens_b [n][k], C_b [m][n]
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((P + sd::BN - 1) / sd::BN), (out_features + sd::BM - 1) / sd::BM, B);
    sd::launch_gemm<float, sd::bf16_t, sd::bf16_t, false, false>(grid, st, W, (const sd::bf16_t *)X, (sd::bf16_t *)Y, bias, out_features, (int)P, in_features,
                                                                 (long)in_features, (long)in_features, P, 0L, P * in_features, P * out_features, 1, in_features);
    return (int)hipGetLastError();
}

int sd_linear_nchw_bwd_data(const void *dY, const float *W, void *dX, int dtype, int B, long P, int in_features, int out_features, void *stream) {
    if (!dY || !W || !dX) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (B <= 0 || P <= 0 || in_features <= 0 || out_features <= 0 || B > 65535) return SD_E_SHAPE;
    if (dtype == SD_F32) return sd::linear_nchw_f32_bwd_data((const float *)dY, W, (float *)dX, B, P, in_features, out_features, stream);
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid((in_features + sd::BN - 1) / sd::BN, (unsigned)((P + sd::BM - 1) / sd::BM), B);
    sd::launch_gemm<sd::bf16_t, float, sd::bf16_t, true, true>(grid, st, (const sd::bf16_t *)dY, W, (sd::bf16_t *)dX, nullptr, (int)P, in_features, out_features,
                                                               P, (long)in_features, (long)in_features, P * out_features, 0L, P * in_features, 1, out_features);
    return (int)hipGetLastError();
}

int sd_linear_nchw_bwd_weight(const void *dY, const void *X, float *dW, float *dbias, int dtype, int B, long P, int in_features, int out_features,
                              void *workspace, size_t workspace_bytes, void *stream) {
    if (!dY || !X || !dW || !workspace) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (B <= 0 || P <= 0 || in_features <= 0 || out_features <= 0 || B > 65535) return SD_E_SHAPE;
    if (dtype == SD_F32)
        return sd::linear_nchw_f32_bwd_weight((const float *)dY, (const float *)X, dW, dbias, B, P, in_features, out_features, workspace, workspace_bytes, stream);
    if (workspace_bytes < sd::linear_nchw_workspace_bytes(B, P, in_features, out_features) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return SD_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int nsplit = sd::pred_splits(B, P);
    const int klen = (int)(((P + nsplit - 1) / nsplit + 63) / 64 * 64);
    nsplit = (int)((P + klen - 1) / klen);
    const long slab = (long)out_features * in_features;
    float *slabs = static_cast<float *>(workspace);
    // slab z = dY_b[:, chunk] . X_b[chunk, :]: A = dY_b [m][k] (pixels contiguous), B = tokens_b [k][n]
    dim3 grid((in_features + sd::BN - 1) / sd::BN, (out_features + sd::BM - 1) / sd::BM, B * nsplit);
    sd::launch_gemm<sd::bf16_t, sd::bf16_t, float, false, true>(grid, st, (const sd::bf16_t *)dY, (const sd::bf16_t *)X, slabs, nullptr, out_features, in_features,
                                                                (int)P, P, (long)in_features, (long)in_features, P * out_features, P * in_features, slab, nsplit,
                                                                klen);
    hipLaunchKernelGGL((sd::slab_reduce<float>), dim3((unsigned)((slab + 255) / 256)), dim3(256), 0, st, slabs, dW, slab, B * nsplit);
    if (dbias) hipLaunchKernelGGL((sd::bias_grad<sd::bf16_t>), dim3(out_features), dim3(256), 0, st, (const sd::bf16_t *)dY, dbias, B, out_features, P);
    return (int)hipGetLastError();
}

int sd_linear_wgrad_fuses_bias_dtype(int dtype, long tokens, int out_features, int in_features) {
    if (tokens <= 0 || out_features <= 0 || in_features <= 0) return 0;
    return sd::linear_wgrad_plan(tokens, out_features, in_features, dtype == SD_BF16).direct ? 1 : 0;
}

int sd_linear_wgrad_slabs(int dtype, long tokens, int out_features, int in_features) {
    if (tokens <= 0 || out_features <= 0 || in_features <= 0 || (dtype != SD_F32 && dtype != SD_BF16)) return 0;
    const sd::WgradPlan p = sd::linear_wgrad_plan(tokens, out_features, in_features, dtype == SD_BF16);
    return p.direct ? p.nslabs : 0;
}

/* the split-K plan of the generic (non tall-skinny) weight gradient: number of [out x in] slabs sd_linear_wgrad_generic_partials writes (>= 2),
 * or 0 when that plan is a single un-split GEMM (nothing to defer) or the direct plan applies */
int sd_linear_wgrad_generic_slabs(int dtype, long tokens, int out_features, int in_features) {
    if (tokens <= 0 || out_features <= 0 || in_features <= 0 || (dtype != SD_F32 && dtype != SD_BF16)) return 0;
    const sd::WgradPlan p = sd::linear_wgrad_plan(tokens, out_features, in_features, dtype == SD_BF16);
    return (!p.direct && p.nsplit > 1) ? p.nslabs : 0;
}

/* the GEMM of sd_linear_wgrad's generic plan WITHOUT its slab combine: the caller sums the slabs (sd_multi_slab_reduce), typically deferred to
 * the end of the backward -- under bf16 storage (config 5) ~54 such combines ran as separate launches per step */
int sd_linear_wgrad_generic_partials(const void *dY, const void *X, int dtype, long tokens, int out_features, int in_features, void *workspace,
                                     size_t workspace_bytes, void *stream) {
    if (!dY || !X || !workspace) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (tokens <= 0 || tokens > 0x7fffffffL || out_features <= 0 || in_features <= 0) return SD_E_SHAPE;
    const int M = out_features, N = in_features;
    const sd::WgradPlan p = sd::linear_wgrad_plan(tokens, M, N, dtype == SD_BF16);
    if (p.direct || p.nsplit <= 1) return SD_E_UNSUPPORTED;
    const long slab = (long)M * N;
    if (workspace_bytes < (size_t)p.nslabs * slab * sizeof(float) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return SD_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *slabs = static_cast<float *>(workspace);
    dim3 grid((N + sd::BN - 1) / sd::BN, (M + sd::BM - 1) / sd::BM, p.nsplit);
    if (dtype == SD_F32)
        sd::launch_gemm<float, float, float, true, true>(grid, st, (const float *)dY, (const float *)X, slabs, nullptr, M, N, (int)tokens, (long)M, (long)N,
                                                         (long)N, 0L, 0L, slab, p.nsplit, p.klen);
    else if (p.tn && sd::wgrad_tn_supported(tokens, M, N, dY, X))
        return sd::wgrad_tn_launch(dY, X, slabs, tokens, M, N, p.nsplit, p.klen, st);
    else
        sd::launch_gemm<sd::bf16_t, sd::bf16_t, float, true, true>(grid, st, (const sd::bf16_t *)dY, (const sd::bf16_t *)X, slabs, nullptr, M, N,
                                                                   (int)tokens, (long)M, (long)N, (long)N, 0L, 0L, slab, p.nsplit, p.klen);
    return (int)hipGetLastError();
}

int sd_linear_wgrad_partials(const void *dY, const void *X, int dtype, long tokens, int out_features, int in_features, int with_bias,
                             void *workspace, size_t workspace_bytes, void *stream) {
    if (!dY || !X || !workspace) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (tokens <= 0 || out_features <= 0 || in_features <= 0) return SD_E_SHAPE;
    const int M = out_features, N = in_features;
    const sd::WgradPlan p = sd::linear_wgrad_plan(tokens, M, N, dtype == SD_BF16);
    if (!p.direct) return SD_E_UNSUPPORTED;
    const long slab = (long)M * N + (with_bias ? M : 0);
    if (workspace_bytes < (size_t)p.nslabs * slab * sizeof(float) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return SD_E_WORKSPACE;
    const int regions = p.regions_m * p.regions_n;
    dim3 grid(p.nsplit, (regions + p.regions_per_wg - 1) / p.regions_per_wg);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *slabs = static_cast<float *>(workspace);
    if (dtype == SD_F32)
        hipLaunchKernelGGL((sd::linear_wgrad_direct<float>), grid, dim3(256), 0, st, (const float *)dY, (const float *)X, slabs, M, N, tokens,
                           p.klen, p.regions_m, p.regions_n, p.regions_per_wg, with_bias ? 1 : 0);
    else
        hipLaunchKernelGGL((sd::linear_wgrad_direct<sd::bf16_t>), grid, dim3(256), 0, st, (const sd::bf16_t *)dY, (const sd::bf16_t *)X, slabs,
                           M, N, tokens, p.klen, p.regions_m, p.regions_n, p.regions_per_wg, with_bias ? 1 : 0);
    return (int)hipGetLastError();
}

int sd_linear_wgrad_fuses_bias(long tokens, int out_features, int in_features) {
    if (tokens <= 0 || out_features <= 0 || in_features <= 0) return 0;
    return sd::linear_wgrad_plan(tokens, out_features, in_features).direct ? 1 : 0;
}

int sd_linear_wgrad(const void *dY, const void *X, float *dW, float *dbias, int dtype, long tokens, int out_features, int in_features,
                    void *workspace, size_t workspace_bytes, void *stream) {
    if (!dY || !X || !dW) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (tokens <= 0 || tokens > 0x7fffffffL || out_features <= 0 || in_features <= 0) return SD_E_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32) return sd::linear_wgrad<float>(dY, X, dW, dbias, workspace, workspace_bytes, tokens, out_features, in_features, st);
    return sd::linear_wgrad<sd::bf16_t>(dY, X, dW, dbias, workspace, workspace_bytes, tokens, out_features, in_features, st);
}

int sd_align1x1_fwd(const void *X, const float *W, const float *bias, void *Y, int dtype, int B, int Cs, int Ct, int h, int w, void *stream) {
    int rc = sd::check_align(X, W, Y, dtype, B, Cs, Ct, h, w);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long P = (long)h * w;
    if (dtype == SD_F32) return sd::align_f32_fwd((const float *)X, W, bias, (float *)Y, B, Cs, Ct, P, st);
    return sd::align_fwd<sd::bf16_t>(X, W, bias, Y, B, Cs, Ct, P, st);
}

int sd_align1x1_bwd_data(const void *dY, const float *W, void *dX, int dtype, int B, int Cs, int Ct, int h, int w, void *stream) {
    int rc = sd::check_align(dY, W, dX, dtype, B, Cs, Ct, h, w);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long P = (long)h * w;
    if (dtype == SD_F32) return sd::align_f32_bwd_data((const float *)dY, W, (float *)dX, B, Cs, Ct, P, st);
    return sd::align_bwd_data<sd::bf16_t>(dY, W, dX, B, Cs, Ct, P, st);
}

int sd_align1x1_bwd_weight(const void *dY, const void *X, float *dW, float *dbias, int dtype, int B, int Cs, int Ct, int h, int w,
                           void *workspace, size_t workspace_bytes, void *stream) {
    int rc = sd::check_align(dY, X, dW, dtype, B, Cs, Ct, h, w);
    if (rc) return rc;
    if (!workspace) return SD_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long P = (long)h * w;
    if (dtype == SD_F32) return sd::align_bwd_weight<float>(dY, X, dW, dbias, workspace, workspace_bytes, B, Cs, Ct, P, st);
    return sd::align_bwd_weight<sd::bf16_t>(dY, X, dW, dbias, workspace, workspace_bytes, B, Cs, Ct, P, st);
}

}  // extern "C"
