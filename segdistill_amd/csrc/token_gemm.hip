// token_gemm.hip -- the Linear products of the MiT encoders / SegFormer heads on TOKEN-MAJOR activations, as exact-f32 GEMMs on the
// gfx950 matrix cores (v_mfma_f32_32x32x2_f32: bit-equal to an fmaf chain, 157 TFLOP/s dense).
//
//   forward    Y [T x N] = act( X [T x K] . W [N x K]^T + bias[N] ) (+ R [T x N])      reference mix_transformer.py:24-27,48-55,75-84,
//   bwd-data   dX[T x K] = dY[T x N] . W [N x K]                                        107-133 (nn.Linear fwd / autograd mm of every q, kv,
//                                                                                       proj, fc1, fc2), segformer_head.py:22-33,75-98
// Why not the library: in these networks T is huge (131072 ... 2048 tokens) while K, N are 32 ... 2048; the library's heuristics pick
// 256x256 / 32x64 macro-tiles that run the K = 32 ... 128 products of stages 1-2 at 5x their HBM time and most others at 30-40 % of the
// f32-input MFMA rate (profiles/r01_train_step_kernels_final.txt: ~5 ms of `Cijk_*` in a 13.1 ms step).
//
// Structure (one template, three tile shapes):
//  * A operand (activations) always [rows][k] with k contiguous; B either [n][k] (forward: the weight as stored) or [k][n] (bwd-data: the
//    same weight read the other way).  k-contiguous tiles live in LDS as [row][BK + 4] floats and are read with ONE ds_read_b128 per four
//    MFMAs (lane (r, kh) takes k = 8q + 4kh .. +3 of row r; MFMA e of quad q then contracts k = 8q + e and 8q + 4 + e -- any bijection of k
//    is fine as long as both operands use the same one); an [k][n] tile is read with ds_read_b32 (32 consecutive n per half-wave).
//  * global -> registers -> LDS double buffering: the 16-byte loads of tile kt+1 are issued BEFORE the MFMAs of tile kt and written to the
//    other LDS buffer after them: ONE barrier per k-step, HBM/L2 latency hidden behind 16 ... 64 MFMAs (64 cycles each) per wave.
//  * epilogue in the accumulator layout (col = lane & 31 -> 128-byte row segments): + bias, exact GELU, + residual.
// 256 threads = 4 waves; tiles 128x128 (waves 2x2, 2x2 MFMA tiles each), 128x64 (2x2, 2x1), 256x32 (4x1, 2x1) chosen by N.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sd_common.h"

namespace sd {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 32, KP = BK + 4;   // k-contiguous LDS rows: 36 floats = 144 B (16-byte aligned, conflict-free ds_read_b128 / ds_write_b128)

__device__ __forceinline__ float gelu_exact(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }

// ---- staging ---------------------------------------------------------------------------------------------------------------------------
// k-contiguous source tile: ROWS rows x BK k.  256 threads; thread -> (row = t/8 + 32 i, 16-byte chunk c = t%8).  Rows beyond `rmax` are
// clamped (their products are never stored), k beyond `K` is zero.  VEC: rows are 16-byte aligned and K % 4 == 0 (block-uniform).
template <int ROWS> struct KTile { float4 v[ROWS / 32]; };

template <int ROWS, bool VEC>
__device__ __forceinline__ void load_ktile(KTile<ROWS> &f, const float *__restrict__ src, long ld, long r0, long rmax, int k0, int K) {
    const int t = threadIdx.x, c = (t & 7) * 4;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        long row = r0 + (t >> 3) + 32 * i;
        row = row < rmax ? row : rmax - 1;
        const float *p = src + row * ld + k0 + c;
        if (VEC) {
            f.v[i] = (k0 + c < K) ? *reinterpret_cast<const float4 *>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            f.v[i].x = (k0 + c + 0 < K) ? p[0] : 0.f;
            f.v[i].y = (k0 + c + 1 < K) ? p[1] : 0.f;
            f.v[i].z = (k0 + c + 2 < K) ? p[2] : 0.f;
            f.v[i].w = (k0 + c + 3 < K) ? p[3] : 0.f;
        }
    }
}

template <int ROWS>
__device__ __forceinline__ void store_ktile(const KTile<ROWS> &f, float *dst /* [ROWS][KP] */) {
    const int t = threadIdx.x, c = (t & 7) * 4;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) *reinterpret_cast<float4 *>(dst + ((t >> 3) + 32 * i) * KP + c) = f.v[i];
}

// n-contiguous source tile (bwd-data B operand): BK k-rows x COLS n.  thread -> (k = t / (COLS/4) + (1024/COLS) i, chunk = t % (COLS/4)).
template <int COLS> struct NTile { float4 v[BK * COLS / 1024]; };

template <int COLS, bool VEC>
__device__ __forceinline__ void load_ntile(NTile<COLS> &f, const float *__restrict__ src, long ld, int n0, int N, int k0, int K) {
    constexpr int CH = COLS / 4, KSTEP = 256 / CH;
    const int t = threadIdx.x, c = (t % CH) * 4, kk = t / CH;
#pragma unroll
    for (int i = 0; i < BK / KSTEP; ++i) {
        int k = k0 + kk + KSTEP * i;
        const bool kin = k < K;
        k = kin ? k : K - 1;
        const float *p = src + (long)k * ld + n0 + c;
        if (VEC) {
            f.v[i] = (kin && n0 + c < N) ? *reinterpret_cast<const float4 *>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            f.v[i].x = (kin && n0 + c + 0 < N) ? p[0] : 0.f;
            f.v[i].y = (kin && n0 + c + 1 < N) ? p[1] : 0.f;
            f.v[i].z = (kin && n0 + c + 2 < N) ? p[2] : 0.f;
            f.v[i].w = (kin && n0 + c + 3 < N) ? p[3] : 0.f;
        }
    }
}

template <int COLS>
__device__ __forceinline__ void store_ntile(const NTile<COLS> &f, float *dst /* [BK][COLS + 4] */) {
    constexpr int CH = COLS / 4, KSTEP = 256 / CH;
    const int t = threadIdx.x, c = (t % CH) * 4, kk = t / CH;
#pragma unroll
    for (int i = 0; i < BK / KSTEP; ++i) *reinterpret_cast<float4 *>(dst + (kk + KSTEP * i) * (COLS + 4) + c) = f.v[i];
}

// ---- the kernel ------------------------------------------------------------------------------------------------------------------------
// C[M x N] = epilogue( A[M x K] . B ),  B(k, n) = BT ? Bm[n * ldb + k] : Bm[k * ldb + n].
// ACT: 0 none, 1 exact GELU.  bias (per n) and residual (C-shaped) optional.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool BT, bool VEC>
__global__ __launch_bounds__(256) void token_gemm_f32(const float *__restrict__ A, const float *__restrict__ Bm, float *__restrict__ C,
                                                       const float *__restrict__ bias, const float *__restrict__ residual, long M, int N, int K,
                                                       long lda, long ldb, long ldc, int act, int tiles_n) {
    static_assert(WAVES_M * WAVES_N == 4, "four waves");
    constexpr int TM = BM / (32 * WAVES_M), TN = BN / (32 * WAVES_N);
    constexpr int A_ELEMS = BM * KP;
    constexpr int B_ELEMS = BT ? BN * KP : BK * (BN + 4);
    extern __shared__ __attribute__((aligned(16))) float lds[];        // [2][A_ELEMS + B_ELEMS]
    // XCD-aware order: consecutive workgroup ids land on different XCDs; give each XCD a contiguous band of M-tiles (they share the B
    // operand through that XCD's L2 and, for one M-tile, the A rows across the N-tiles)
    const long nblk = gridDim.x;
    const long id = blockIdx.x;
    const long q = nblk / 8, rem = nblk % 8, xcd = id % 8;
    const long tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + id / 8;
    const long tm_idx = tile / tiles_n;
    const int tn_idx = (int)(tile - tm_idx * tiles_n);
    const long m0 = tm_idx * BM;
    const int n0 = tn_idx * BN;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave / WAVES_N) * (TM * 32), wn = (wave % WAVES_N) * (TN * 32);
    const int r = lane & 31, kh = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // n-tiles of this wave that hold at least one real column (N = 150, 160 ...: do not multiply padding)
    bool live[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) live[j] = n0 + wn + 32 * j < N;

    KTile<BM> fa;
    KTile<BN> fbt;
    NTile<BN> fbn;
    const int nk = (K + BK - 1) / BK;
    load_ktile<BM, VEC>(fa, A, lda, m0, M, 0, K);
    if (BT) load_ktile<BN, VEC>(fbt, Bm, ldb, n0, N, 0, K);
    else load_ntile<BN, VEC>(fbn, Bm, ldb, n0, N, 0, K);
    store_ktile<BM>(fa, lds);
    if (BT) store_ktile<BN>(fbt, lds + A_ELEMS);
    else store_ntile<BN>(fbn, lds + A_ELEMS);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const float *As = lds + (kt & 1) * (A_ELEMS + B_ELEMS);
        const float *Bs = As + A_ELEMS;
        const bool more = kt + 1 < nk;
        if (more) {
            load_ktile<BM, VEC>(fa, A, lda, m0, M, (kt + 1) * BK, K);
            if (BT) load_ktile<BN, VEC>(fbt, Bm, ldb, n0, N, (kt + 1) * BK, K);
            else load_ntile<BN, VEC>(fbn, Bm, ldb, n0, N, (kt + 1) * BK, K);
        }
#pragma unroll
        for (int q4 = 0; q4 < BK / 8; ++q4) {
            float4 a[TM];
            float b[TN][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4 *>(As + (wm + 32 * i + r) * KP + 8 * q4 + 4 * kh);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (BT) {
                    const float4 v = *reinterpret_cast<const float4 *>(Bs + (wn + 32 * j + r) * KP + 8 * q4 + 4 * kh);
                    b[j][0] = v.x, b[j][1] = v.y, b[j][2] = v.z, b[j][3] = v.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[j][e] = Bs[(8 * q4 + 4 * kh + e) * (BN + 4) + wn + 32 * j + r];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float av = e == 0 ? a[i].x : (e == 1 ? a[i].y : (e == 2 ? a[i].z : a[i].w));
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        if (live[j]) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[j][e], acc[i][j], 0, 0, 0);
                }
            }
        }
        if (more) {
            float *An = lds + ((kt + 1) & 1) * (A_ELEMS + B_ELEMS);
            store_ktile<BM>(fa, An);
            if (BT) store_ktile<BN>(fbt, An + A_ELEMS);
            else store_ntile<BN>(fbn, An + A_ELEMS);
        }
        __syncthreads();
    }
    // epilogue; C/D layout of the 32x32 accumulator: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn + 32 * j + r;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long m = m0 + wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * kh;
                if (m < M) {
                    float v = acc[i][j][e] + bv;
                    if (act == 1) v = gelu_exact(v);
                    if (residual) v += residual[m * ldc + n];
                    C[m * ldc + n] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool BT>
int launch(const float *A, const float *Bm, float *C, const float *bias, const float *residual, long M, int N, int K, long lda, long ldb, long ldc,
           int act, hipStream_t st) {
    constexpr size_t lds_bytes = 2 * (size_t)(BM * KP + (BT ? BN * KP : BK * (BN + 4))) * sizeof(float);
    const long tiles_m = (M + BM - 1) / BM;
    const int tiles_n = (N + BN - 1) / BN;
    const long nblk = tiles_m * tiles_n;
    if (nblk > 0x7fffffffL) return SD_E_SHAPE;
    // 16-byte loads need aligned rows: base pointers, leading dimensions, and (k-contiguous tiles) K % 4, (n-contiguous) N % 4
    const bool vec = (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(Bm) & 15) == 0 && lda % 4 == 0 && ldb % 4 == 0 &&
                     K % 4 == 0 && (BT || N % 4 == 0);
    auto kern = vec ? token_gemm_f32<BM, BN, WAVES_M, WAVES_N, BT, true> : token_gemm_f32<BM, BN, WAVES_M, WAVES_N, BT, false>;
    if (lds_bytes > 64 * 1024) {
        static bool raised[2] = {false, false};      // per instantiation (this function template) and load flavour; idempotent, so a race is harmless
        if (!raised[vec]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return (int)e;
            raised[vec] = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds_bytes, st, A, Bm, C, bias, residual, M, N, K, lda, ldb, ldc, act, tiles_n);
    return (int)hipGetLastError();
}

template <bool BT>
int dispatch(const float *A, const float *Bm, float *C, const float *bias, const float *residual, long M, int N, int K, long lda, long ldb, long ldc,
             int act, hipStream_t st) {
    if (N <= 32) return launch<256, 32, 4, 1, BT>(A, Bm, C, bias, residual, M, N, K, lda, ldb, ldc, act, st);
    if (N <= 64) return launch<128, 64, 2, 2, BT>(A, Bm, C, bias, residual, M, N, K, lda, ldb, ldc, act, st);
    return launch<128, 128, 2, 2, BT>(A, Bm, C, bias, residual, M, N, K, lda, ldb, ldc, act, st);
}

}  // namespace
}  // namespace sd

extern "C" {

int sd_linear_fwd(const void *X, const float *W, long w_row_stride, const float *bias, const void *residual, void *Y, int dtype, long tokens,
                  int in_features, int out_features, int act, void *stream) {
    if (!X || !W || !Y) return SD_E_NULL;
    if (dtype != SD_F32) return SD_E_DTYPE;
    if (tokens <= 0 || in_features <= 0 || out_features <= 0) return SD_E_SHAPE;
    if (w_row_stride == 0) w_row_stride = in_features;
    if (w_row_stride < in_features) return SD_E_SHAPE;
    if (act != 0 && act != 1) return SD_E_UNSUPPORTED;
    return sd::dispatch<true>((const float *)X, W, (float *)Y, bias, (const float *)residual, tokens, out_features, in_features, in_features,
                              w_row_stride, out_features, act, static_cast<hipStream_t>(stream));
}

int sd_linear_bwd_data(const void *dY, const float *W, long w_row_stride, void *dX, int dtype, long tokens, int in_features, int out_features,
                       void *stream) {
    if (!dY || !W || !dX) return SD_E_NULL;
    if (dtype != SD_F32) return SD_E_DTYPE;
    if (tokens <= 0 || in_features <= 0 || out_features <= 0) return SD_E_SHAPE;
    if (w_row_stride == 0) w_row_stride = in_features;
    if (w_row_stride < in_features) return SD_E_SHAPE;
    // C[T x in] = dY[T x out] . W[out x in]: K = out_features, B(k, n) = W[k * ldw + n]
    return sd::dispatch<false>((const float *)dY, W, (float *)dX, nullptr, nullptr, tokens, in_features, out_features, out_features, w_row_stride,
                               in_features, 0, static_cast<hipStream_t>(stream));
}

}  // extern "C"
