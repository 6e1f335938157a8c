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
#include <string.h>

#include <type_traits>

#include "sd_common.h"

namespace sd {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int BK = 32, KP = BK + 4;   // k-contiguous LDS rows: 36 floats = 144 B (16-byte aligned, conflict-free ds_read_b128 / ds_write_b128)

__device__ __forceinline__ float gelu_exact(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }

// ---- staging ---------------------------------------------------------------------------------------------------------------------------
// k-contiguous source tile: ROWS rows x BK k.  256 threads; thread -> (row = t/8 + 32 i, 16-byte chunk c = t%8).  Rows beyond `rmax` are
// clamped (their products are never stored), k beyond `K` is zero.  VEC: rows are 16-byte aligned and K % 4 == 0 (block-uniform).
template <int ROWS> struct KTile { float4 v[ROWS / 32]; };

// A 16-byte global load the compiler can neither sink nor reorder (asm volatile): hipcc's scheduler / sinking moves ordinary loads down to
// their first use -- here the LDS stores half a k-step later -- which exposes the whole memory latency in every step.  The price: the
// compiler does not count these loads in vmcnt, so the consumer must be preceded by an explicit wait (wait_loads() below).
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 pinned_load16(const float *p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void wait_loads() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);      // "memory" orders memory operations only; nothing that reads the loaded registers may move above the wait
}

template <int ROWS, bool VEC, bool FULLK>
__device__ __forceinline__ void load_ktile(KTile<ROWS> &f, const float *__restrict__ src, long ld, long r0, long rmax, int k0, int K, bool live = true) {
    const int t = threadIdx.x, c = (t & 7) * 4;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        long row = r0 + (t >> 3) + 32 * i;
        row = row < rmax ? row : rmax - 1;
        // branch-free: the address is clamped into the row, the value of an out-of-range k is zeroed by a select afterwards
        if (VEC && FULLK) {
            // K % 32 == 0: every chunk of every k-step exists.  !live (block-uniform): a placeholder request of one shared line, so that the
            // count of outstanding operations -- the hand-counted vmcnt waits -- is the same in every round
            f.v[i] = pinned_load16(live ? src + row * ld + k0 + c : src);
        } else if (VEC) {
            const bool in = k0 + c < K;                                   // K % 4 == 0: a chunk is entirely in or out
            const float4 v = *reinterpret_cast<const float4 *>(src + row * ld + (in ? k0 + c : 0));
            f.v[i] = in ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            const float *p = src + row * ld;
            const int k = k0 + c;
            const float x = p[min(k, K - 1)], y = p[min(k + 1, K - 1)], z = p[min(k + 2, K - 1)], w = p[min(k + 3, K - 1)];
            f.v[i] = make_float4(k < K ? x : 0.f, k + 1 < K ? y : 0.f, k + 2 < K ? z : 0.f, k + 3 < K ? w : 0.f);
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

template <int COLS, bool VEC, bool FULLK>
__device__ __forceinline__ void load_ntile(NTile<COLS> &f, const float *__restrict__ src, long ld, int n0, int N, int k0, int K) {
    constexpr int CH = COLS / 4, KSTEP = 256 / CH;
    const int t = threadIdx.x, c = (t % CH) * 4, kk = t / CH;
#pragma unroll
    for (int i = 0; i < BK / KSTEP; ++i) {
        int k = k0 + kk + KSTEP * i;
        const bool kin = FULLK || k < K;
        k = kin ? k : K - 1;
        const float *p = src + (long)k * ld;
        const int n = n0 + c;
        if (VEC && FULLK) {
            // columns beyond N are clamped, not zeroed: they only feed output columns that are never stored
            f.v[i] = pinned_load16(p + (n < N ? n : N - 4));
        } else if (VEC) {
            const bool in = kin && n < N;                                 // N % 4 == 0: a chunk is entirely in or out
            const float4 v = *reinterpret_cast<const float4 *>(p + (n < N ? n : 0));
            f.v[i] = in ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            const float x = p[min(n, N - 1)], y = p[min(n + 1, N - 1)], z = p[min(n + 2, N - 1)], w = p[min(n + 3, N - 1)];
            f.v[i] = make_float4(kin && n < N ? x : 0.f, kin && n + 1 < N ? y : 0.f, kin && n + 2 < N ? z : 0.f, kin && n + 3 < N ? w : 0.f);
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
// C_z[M x N] = epilogue( A_z[M x K] . B_z ),  z = blockIdx.y (operand strides sA / sB / sC elements, 0 = shared),
//   A(m, k) = AKC ? A[m * lda + k] : A[k * lda + m],   B(k, n) = BT ? Bm[n * ldb + k] : Bm[k * ldb + n].
// EPI: bit 0 = + residual (C-shaped), bit 1 = exact GELU.  bias optional: per column n, or per row m when BIAS_ROW (the NCHW 1x1 projection).
// BFRAG (round 3; X3 + FULLK + AKC only): the B operand arrives PRE-SPLIT -- three bf16 planes in the fragment-contiguous layout written by
// presplit_planes below (one 1 KB run per (32-column block, 16-deep k-step, plane): lane l's 16 bytes ARE its MFMA B fragment) -- and is
// loaded straight from global memory / L2 into registers, one k16 step ahead: no LDS for B, no split arithmetic for B.  `Bm` is then the
// planes buffer and `ldb` its number of 32-column blocks.
// APL (round 3; X3 + FULLK + AKC + BT only): the A operand arrives PRE-SPLIT as three ROW-MAJOR bf16 planes [plane][M][K] (presplit_planes, dir 2)
// and is staged through LDS as planes (rows of 32 k = 64 bytes, pitch 80: conflict-free ds_read_b128), so its MFMA fragments are plain 16-byte LDS
// reads -- for a tile whose every wave needs ALL of A's rows (the 160-class-row tile of linear_pred: five row blocks per wave) this removes 5/6 of
// the split arithmetic, which was 2040 vector-issue cycles per k-step against 1920 of matrix work.  `A` is then the planes buffer.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool BT, bool VEC, bool FULLK, int EPI, bool AKC = true, bool BIAS_ROW = false, bool X3 = false,
          bool BFRAG = false, bool APL = false>
__global__ __launch_bounds__(256) void token_gemm_f32(const float *__restrict__ A, const float *__restrict__ Bm, float *__restrict__ C,
                                                       const float *__restrict__ bias, const float *__restrict__ residual, long M, int N, int K,
                                                       long lda, long ldb, long ldc, int tiles_n, int vec_out, long sA, long sB, long sC,
                                                       int nsplit, int klen) {
    static_assert(WAVES_M * WAVES_N == 4, "four waves");
    constexpr int TM = BM / (32 * WAVES_M), TN = BN / (32 * WAVES_N);
    constexpr int APITCH = 20;                                   // APL: one plane row = 32 bf16 + 16 bytes of pad = 20 floats
    constexpr int A_ELEMS = APL ? 3 * BM * APITCH : (AKC ? BM * KP : BK * (BM + 4));
    constexpr int B_ELEMS = BFRAG ? 0 : (BT ? BN * KP : BK * (BN + 4));
    static_assert(!APL || (X3 && FULLK && VEC && AKC && BT && !BFRAG), "pre-split A planes: split-bf16 mode, whole k-steps, k-contiguous operands");
    static_assert(!BFRAG || (X3 && FULLK && VEC && AKC), "pre-split B planes: split-bf16 mode, whole k-steps, k-contiguous A");
    extern __shared__ __attribute__((aligned(16))) float lds[];        // [2][A_ELEMS + B_ELEMS]
#ifdef SD_GEMM_STAMPS
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
    // z = batch * nsplit + split: a split covers k in [split * klen, min(K, (split + 1) * klen)) and writes its own C slab (split-K)
    {
        const int zb = blockIdx.y / nsplit, sp = blockIdx.y - zb * nsplit;
        const long k0 = (long)sp * klen;
        A += (long)zb * sA + (AKC ? k0 : k0 * lda);
        Bm += (long)zb * sB + (BT ? k0 : k0 * ldb);
        C += (long)blockIdx.y * sC;
        if (residual) residual += (long)blockIdx.y * sC;
        K = (int)min((long)klen, (long)K - k0);
    }
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

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform values stay scalar
    const int wm = (wave / WAVES_N) * (TM * 32), wn = (wave % WAVES_N) * (TN * 32);
    const int r = lane & 31, kh = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // NB no "skip the MFMAs of padding columns" test anywhere: each wave has its own SIMD, so a wave multiplying padding (N = 150, 160 in the
    // last tile column) delays nobody, whereas ANY branch inside the k-step splits it into basic blocks and the compiler then sinks the global
    // loads down to their use (measured: loads issued right in front of the LDS stores, full latency exposed every step)

    KTile<BM> fa;
    NTile<BM> fan;
    KTile<BN> fbt;
    NTile<BN> fbn;
    const int nk = (K + BK - 1) / BK;
    // APL staging: 3 planes x BM rows x 4 chunks of 16 bytes per k-step; chunk q -> (plane, row, c); the surplus threads of the last round
    // repeat the last chunk (same address, same data: no branch)
    constexpr int NAPL = APL ? (3 * BM * 4 + 255) / 256 : 1;
    f32x4 fap[NAPL];
    auto load_tiles = [&](int k0, bool live = true) {
        if constexpr (APL) {
#pragma unroll
            for (int i = 0; i < NAPL; ++i) {
                const int q = min((int)threadIdx.x + 256 * i, 3 * BM * 4 - 1);
                const int pl = q / (BM * 4), rem = q - pl * (BM * 4), row = rem >> 2, c = rem & 3;
                const long gr = min(m0 + row, M - 1);
                const char *src = reinterpret_cast<const char *>(A) + (((long)pl * M + gr) * K + k0) * 2 + c * 16;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(fap[i]) : "v"(src) : "memory");
            }
        } else if (AKC) load_ktile<BM, VEC, FULLK>(fa, A, lda, m0, M, k0, K, live);
        else load_ntile<BM, VEC, FULLK>(fan, A, lda, (int)m0, (int)M, k0, K);
        if (BFRAG) return;
        if (BT) load_ktile<BN, VEC, FULLK>(fbt, Bm, ldb, n0, N, k0, K);
        else load_ntile<BN, VEC, FULLK>(fbn, Bm, ldb, n0, N, k0, K);
    };
    auto store_tiles = [&](float *buf) {
        if constexpr (APL) {
#pragma unroll
            for (int i = 0; i < NAPL; ++i) {
                const int q = min((int)threadIdx.x + 256 * i, 3 * BM * 4 - 1);
                const int pl = q / (BM * 4), rem = q - pl * (BM * 4), row = rem >> 2, c = rem & 3;
                *reinterpret_cast<f32x4 *>(buf + (pl * BM + row) * APITCH + c * 4) = fap[i];
            }
        } else if (AKC) store_ktile<BM>(fa, buf);
        else store_ntile<BM>(fan, buf);
        if (BFRAG) return;
        if (BT) store_ktile<BN>(fbt, buf + A_ELEMS);
        else store_ntile<BN>(fbn, buf + A_ELEMS);
    };
#ifdef SD_DIAG_NOPROLOGUE
    // diagnostic A/B build only (tools/gemm_nosplit_probe.py; WRONG numerics): the first tile's operands are NOT requested and not waited for --
    // the time this build saves is an upper bound on what a persistent tile loop can buy by requesting the next tile's first operands under the
    // current tile's epilogue (profiles/r06_gemm_noprologue_probe.txt: 3.5 %)
    if constexpr (BFRAG) {
#pragma unroll
        for (int i = 0; i < BM / 32; ++i) fa.v[i] = make_float4(1.f, 1.f, 1.f, 1.f);
    } else {
        load_tiles(0);
    }
#else
    load_tiles(0);
#endif
    if (FULLK && !BFRAG) wait_loads();
    if (!BFRAG) {
        store_tiles(lds);
        __syncthreads();
    }
    // Fragments of one quad (8 consecutive k): per lane 4 k-values of TM A rows and TN B columns.
    struct Frag { float a[TM][4], b[TN][4]; };
    auto read_frag = [&](Frag &f, const float *As, const float *Bs, int q4) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (AKC) {
                const float4 v = *reinterpret_cast<const float4 *>(As + (wm + 32 * i + r) * KP + 8 * q4 + 4 * kh);
                f.a[i][0] = v.x, f.a[i][1] = v.y, f.a[i][2] = v.z, f.a[i][3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) f.a[i][e] = As[(8 * q4 + 4 * kh + e) * (BM + 4) + wm + 32 * i + r];
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (BT) {
                const float4 v = *reinterpret_cast<const float4 *>(Bs + (wn + 32 * j + r) * KP + 8 * q4 + 4 * kh);
                f.b[j][0] = v.x, f.b[j][1] = v.y, f.b[j][2] = v.z, f.b[j][3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) f.b[j][e] = Bs[(8 * q4 + 4 * kh + e) * (BN + 4) + wn + 32 * j + r];
            }
        }
    };
    auto mma = [&](const Frag &f) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][e], f.b[j][e], acc[i][j], 0, 0, 0);
    };
    // Software pipeline of one k-step (4 quads of 16 ... 4 MFMAs per wave):
    //   quad 0   | issue the global loads of tile kt+1, MFMAs
    //   quad 1-2 | the loads have landed behind ~2 quads of MFMAs: write them to the OTHER LDS buffer (free since the barrier of step kt-1);
    //            | ds_write issue slots between MFMAs are free -- the matrix pipe is busy 64 cycles per instruction
    //   quad 3   | its fragments are already in registers: BARRIER first (tile kt+1 visible, tile kt no longer read by anyone), read the
    //            | first fragments of tile kt+1, THEN the MFMAs of quad 3 -- the barrier wait and the LDS latency of the next step's first
    //            | read sit behind 16 MFMAs (1024 cycles) instead of in front of them.  One barrier per k-step.
    // Two co-resident workgroups run this same program nearly in lockstep, so a bubble in one is not filled by the other (measured: 68 % MFMA
    // utilisation with the plain load / multiply / store / barrier order).
    if constexpr (BFRAG) {
        // ---- split-bf16 products with a PRE-SPLIT B operand ------------------------------------------------------------------------------
        // Per k32 tile and wave: 2 x TM fragments of A are read from LDS (eight CONSECUTIVE k per lane: k = 16 s + 8 (lane >> 5) .. + 7, the
        // k order of the planes) and split in registers (the only vector arithmetic left: ~170 instructions against 48 MFMAs), 2 x 3 TN
        // B fragments come from global memory as 16-byte loads.  vmcnt is counted by hand (asm loads are invisible to hipcc): the A tile
        // of step kt+1 (NA loads) is requested first, then B(kt, 1) (NB loads); B(kt+1, 0) follows the LDS stores.
        constexpr int NA = BM / 32, NB = 3 * TN;
        struct APl { bf16x8 h[TM], m[TM], l[TM]; };
        struct BPl { bf16x8 h[TN], m[TN], l[TN]; };
        const int KS = K / 16, nblocks = (int)ldb;
        const char *bbase[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nb = min((n0 + wn) / 32 + j, nblocks - 1);                 // blocks beyond N: clamped (their columns are never stored)
            bbase[j] = reinterpret_cast<const char *>(Bm) + ((size_t)nb * KS) * 3072 + lane * 16;
        }
        auto load_b = [&](BPl &b, int ks) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const char *q = bbase[j] + (size_t)ks * 3072;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(b.h[j]) : "v"(q) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(b.m[j]) : "v"(q) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off offset:2048" : "=v"(b.l[j]) : "v"(q) : "memory");
            }
        };
        // Conversions in pairs (one v_cvt_pk_bf16_f32 per two values); residuals as one packed subtraction per pair.  (Measured and dropped,
        // same box: the residuals as scalar v_sub_f32 kept apart by opaque asm operands -- a packed f32 instruction beside MFMAs is priced
        // ~13 cycles above the two scalar ones it replaces, MI355X_MICROARCH.md -- 106.8 vs 105.1 / 107.1 us at 256 -> 256 over 131072
        // tokens, no shape moved by more than the run-to-run noise: the asm operands cost s_nop pads and scheduling freedom.)
        auto split8 = [&](const float (&x)[8], bf16x8 &h, bf16x8 &m, bf16x8 &l) {
#ifdef SD_DIAG_NOSPLIT
            // diagnostic A/B build only (tools/gemm_nosplit_probe.py; WRONG numerics): one conversion per pair and NO residual arithmetic -- the time
            // this build saves is an upper bound on what moving the split of the activation operand out of the k-loop (into its producer) can buy
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const f32x2 v = {x[e], x[e + 1]};
                const bf16x2 hh = __builtin_convertvector(v, bf16x2);
                h[e] = hh[0], h[e + 1] = hh[1], m[e] = hh[0], m[e + 1] = hh[1], l[e] = hh[0], l[e + 1] = hh[1];
            }
            return;
#endif
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
        };
        auto read_split_a = [&](APl &P, const float *As, int s2) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float *q = As + (wm + 32 * i + r) * KP + 16 * s2 + 8 * kh;
                const float4 v0 = *reinterpret_cast<const float4 *>(q), v1 = *reinterpret_cast<const float4 *>(q + 4);
                const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                split8(x, P.h[i], P.m[i], P.l[i]);
            }
        };
        auto mfma_planes = [&](const APl &A_, const BPl &B_) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x16 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_.m[i], B_.m[j], c, 0, 0, 0);      // small terms first
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_.l[i], B_.h[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_.h[i], B_.l[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_.m[i], B_.h[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_.h[i], B_.m[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_.h[i], B_.h[j], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        };
        auto wait_vm = [&](auto n) {       // all but the n youngest vector-memory operations of this wave are done
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(n)::value) : "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        // One scheduling region = the LDS reads + split of the NEXT A fragments and the 6 TM TN MFMAs on the CURRENT planes, interleaved by
        // sched_group_barrier: an MFMA holds the SIMD's vector issue for 8 of its 32 cycles, so ~4 vector instructions ride in each gap.
        auto interleave = [&]() {
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * TM, 0);      // the fragment reads first: their latency hides behind the first MFMAs
#pragma unroll
            for (int q = 0; q < 6 * TM * TN; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
        };
        // (Measured and dropped, round 3: the A fragments straight from global memory as well -- every wave requests its own rows a whole k-step
        // ahead into two register sets; no LDS, no barrier in the main loop, 252 registers, audit clean.  Same box, planes forward: 123.4 vs
        // 103.9 us at 256 -> 256 over 131072 tokens, 20.9 vs 17.2 us at 32 -> 32, 22.9 vs 18.7 at 160 -> 640 over 8192 -- slower on all 47 shapes
        // of tools/gemm_bench.py.  Each row of A is then fetched by both waves of a tile row, 16 bytes per lane from 32 different lines per
        // instruction: the vector-memory path, not the barrier, becomes the limit.)
        // Schedule of k-step kt (tile kt in LDS stage kt & 1; B0 = B(kt, 0) and B1 = B(kt, 1) are register sets):
        //   top     request B(kt, 1)                                   wait B(kt, 0)      [queue: B(kt,0), A(kt+1) | B(kt,1)]
        //   half 0  { read + split A(kt, half 1) -> PA1  ||  MFMAs on (PA0, B0) }
        //   middle  wait A(kt+1) (requested one whole step ago), store it to the other stage, request B(kt+1, 0); barrier; request A(kt+2)
        //   half 1  wait B(kt, 1)  { read + split A(kt+1, half 0) -> PA0  ||  MFMAs on (PA1, B1) }
        // vmcnt retires in order, so B(kt+1, 0) is requested BEFORE A(kt+2): the wait for it at the next top does not drag the A tile along,
        // which keeps its full step of flight time (the round-2 order -- A requested at the top, consumed half a step later -- ran each
        // k-step at the HBM latency: 3.2 us against 1.3 us of matrix work).
        // tools/gemm_stamps.py (256 -> 256 over 131072 tokens, cycles per k-step and wave): wait B(kt, 0) 614 | half 0 1058 | wait A 48 | store +
        // requests + barrier 1294 | wait B(kt, 1) 63 | half 1 1039 = 4117 against 1536 of MFMA issue, two waves per SIMD; prologue 5500 and
        // last step + epilogue 9300 of a 43900-cycle workgroup lifetime.  The fragment loads hit in L2 yet take ~3000 cycles: the vector L1
        // returns a CU's loads in order, so they queue behind the other waves' A-tile loads from HBM.  (Tried and dropped: a THIRD B register
        // set, every request a whole step ahead -- the rotation needs the k-loop unrolled by three and hipcc then allocates > 256 registers
        // (200 AGPRs as spill space at one workgroup per CU, 357-635 scratch spills when held to two), pending loads among the spilled.)
        // (Tried and dropped: different `s_setprio` levels, by hardware wave slot, for the two waves of a SIMD in their MFMA halves, so that they
        // alternate instead of running in lockstep -- same box, slower on 44 of 47 shapes: 114.1 vs 109.0 us at 256 -> 256 over 131072 tokens,
        // 56.1 vs 46.9 at 320 -> 1280 over 8192; the lower-priority wave's split instructions starve and its own MFMAs start late.)
        APl PA0, PA1;
        BPl B0, B1;
#ifdef SD_DIAG_NOPROLOGUE
#pragma unroll
        for (int j = 0; j < TN; ++j) B0.h[j] = B0.m[j] = B0.l[j] = bf16x8{};
#else
        load_b(B0, 0);                                  // (the A tile of step 0 was requested above)
        wait_vm(std::integral_constant<int, 0>{});
#endif
        store_tiles(lds);
        if (nk > 1) load_tiles(BK);                     // A(1): consumed in the middle of step 0
        __syncthreads();
        read_split_a(PA0, lds, 0);
        // -DSD_GEMM_STAMPS (diagnostic A/B build only, tools/gemm_stamps.py): s_memtime at the phase boundaries of every k-step, summed per
        // phase; wave 0 of every 64th workgroup writes its sums to the buffer that build's launcher passes in `residual` (unused by EPI = 0)
#ifdef SD_GEMM_STAMPS
        unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime(), tbegin = tlast;
#define SD_STAMP(i)                                                    \
    do {                                                               \
        const unsigned long long tnow = __builtin_amdgcn_s_memtime();  \
        tph[i] += tnow - tlast;                                        \
        tlast = tnow;                                                  \
    } while (0)
#else
#define SD_STAMP(i)
#endif
        for (int kt = 0; kt + 1 < nk; ++kt) {
            const float *cur = lds + (kt & 1) * A_ELEMS;
            float *nxt = lds + ((kt + 1) & 1) * A_ELEMS;
            load_b(B1, 2 * kt + 1);                     // NB
            wait_vm(std::integral_constant<int, NA + NB>{});     // B0 = B(kt, 0) landed; A(kt+1) and B(kt, 1) may still fly
            SD_STAMP(0);
            read_split_a(PA1, cur, 1);
            mfma_planes(PA0, B0);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            SD_STAMP(1);
            wait_vm(std::integral_constant<int, NB>{});          // A(kt+1) landed
            SD_STAMP(2);
            store_tiles(nxt);
            load_b(B0, 2 * kt + 2);                     // NB; B0's registers were last read by the MFMAs above
            // the A tile is requested BEHIND the barrier: all four waves' fragment requests (L2 hits) are then in the CU's in-order return
            // queue ahead of the workgroup's HBM requests instead of interleaved with them (same box, 47 shapes of tools/gemm_bench.py: 3-7 %
            // on the latency-bound products -- 2048 x 1024 -> 256: 37.0 vs 39.5 us, 4096 -> 64: 133 vs 144 -- and +-1 % on the wide ones)
            __syncthreads();
            load_tiles((kt + 2) * BK, kt + 2 < nk);             // NA; past the last tile: placeholder requests (no branch: one basic block)
            SD_STAMP(3);
            wait_vm(std::integral_constant<int, NA + NB>{});     // B1 = B(kt, 1) landed
            SD_STAMP(4);
            read_split_a(PA0, nxt, 0);
            mfma_planes(PA1, B1);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            SD_STAMP(5);
        }
#ifdef SD_GEMM_STAMPS
        if (residual && (blockIdx.x & 63) == 0 && threadIdx.x == 0) {
            unsigned long long *o = reinterpret_cast<unsigned long long *>(const_cast<float *>(residual)) + (blockIdx.x >> 6) * 12;
#pragma unroll
            for (int i = 0; i < 6; ++i) o[i] = tph[i];
            o[6] = __builtin_amdgcn_s_memtime() - tbegin;
            o[7] = (unsigned long long)(nk - 1);
            o[8] = tbegin - t_entry;                                  // prologue: first requests -> first fragments split
            o[9] = t_entry;                                           // absolute, for the workgroup timeline
        }
#endif
#undef SD_STAMP
        {
            const float *cur = lds + ((nk - 1) & 1) * A_ELEMS;
            load_b(B1, 2 * nk - 1);
            wait_vm(std::integral_constant<int, NB>{});          // B0 landed (and everything older)
            // the last round's placeholder A requests are dead values to the compiler, which would hand their registers out while the loads
            // were still in flight (tools/asm_pending_audit.py): keep them alive up to this wait
#pragma unroll
            for (int i = 0; i < NA; ++i) asm volatile("" ::"v"(fa.v[i].x), "v"(fa.v[i].y), "v"(fa.v[i].z), "v"(fa.v[i].w));
            read_split_a(PA1, cur, 1);
            mfma_planes(PA0, B0);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            wait_vm(std::integral_constant<int, 0>{});
            mfma_planes(PA1, B1);
            __syncthreads();
        }
    } else if constexpr (APL) {
        // ---- split-bf16 products, A from pre-split LDS planes, B (fp32 in LDS) split in registers --------------------------------------------
        struct Pl { bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN]; };
        auto split8 = [&](const float (&x)[8], bf16x8 &h, bf16x8 &m, bf16x8 &l) {
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
        };
        auto fetch = [&](Pl &P, const float *As, const float *Bs, int s2) {       // k = 16 s2 + 8 kh .. + 7 for both operands
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float *q = As + (wm + 32 * i + r) * APITCH + 8 * s2 + 4 * kh;
                P.ah[i] = *reinterpret_cast<const bf16x8 *>(q);
                P.am[i] = *reinterpret_cast<const bf16x8 *>(q + BM * APITCH);
                P.al[i] = *reinterpret_cast<const bf16x8 *>(q + 2 * BM * APITCH);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float *q = Bs + (wn + 32 * j + r) * KP + 16 * s2 + 8 * kh;
                const float4 v0 = *reinterpret_cast<const float4 *>(q), v1 = *reinterpret_cast<const float4 *>(q + 4);
                const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                split8(x, P.bh[j], P.bm[j], P.bl[j]);
            }
        };
        auto mfma_planes = [&](const Pl &P) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x16 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P.am[i], P.bm[j], c, 0, 0, 0);      // small terms first
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P.al[i], P.bh[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P.ah[i], P.bl[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P.am[i], P.bh[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P.ah[i], P.bm[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P.ah[i], P.bh[j], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        };
        // One workgroup per CU (113 KB of LDS) = one wave per SIMD: nobody covers a wait, so the tile of step kt+2 is requested as soon as the
        // staging registers are free (right after tile kt+1 went to LDS) and has a whole k-step -- 60 MFMAs, ~0.9 us -- to arrive; requested at
        // the top of its own step (round-2 order) it had half of that and every k-step stalled on HBM latency (250 -> see profiles/).
        auto keep_alive = [&]() {      // pending requests whose values nobody reads are dead to the compiler: pin their registers up to the wait
#pragma unroll
            for (int i = 0; i < NAPL; ++i) asm volatile("" ::"v"(fap[i]));
#pragma unroll
            for (int i = 0; i < BN / 32; ++i) asm volatile("" ::"v"(fbt.v[i].x), "v"(fbt.v[i].y), "v"(fbt.v[i].z), "v"(fbt.v[i].w));
        };
        Pl P0, P1;
        if (nk > 1) load_tiles(BK);
        fetch(P0, lds, lds + A_ELEMS, 0);
        // staging instructions of one k-step per thread: LDS stores, then the global requests of the tile after next
        constexpr int NST = NAPL + BN / 32;
        for (int kt = 0; kt + 1 < nk; ++kt) {
            const float *cur = lds + (kt & 1) * (A_ELEMS + B_ELEMS);
            float *nxt = lds + ((kt + 1) & 1) * (A_ELEMS + B_ELEMS);
            wait_loads();                               // tile kt+1 has had a whole k-step to land
            // ONE scheduling region: the MFMAs on the current planes with, in their gaps, the fragment reads + split of the next planes, the LDS
            // stores of tile kt+1 and the requests of tile kt+2 -- a wave alone on its SIMD cannot afford a phase in which the matrix pipe waits
            // for its stores and requests (that phase structure ran at 39 % of the matrix-pipe time)
            fetch(P1, cur, cur + A_ELEMS, 1);
            mfma_planes(P0);
            store_tiles(nxt);
            load_tiles(min(kt + 2, nk - 1) * BK);       // past the end: a repeat of the last tile (1 / nk of the traffic), dropped below
            __builtin_amdgcn_sched_group_barrier(0x100, 3 * TM + 2 * TN, 0);
#pragma unroll
            for (int q = 0; q < 6 * TM * TN; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (q < NST) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                else if (q < 2 * NST) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            fetch(P0, nxt, nxt + A_ELEMS, 0);
            mfma_planes(P1);
            __builtin_amdgcn_sched_group_barrier(0x100, 3 * TM + 2 * TN, 0);
#pragma unroll
            for (int q = 0; q < 6 * TM * TN; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        {
            const float *cur = lds + ((nk - 1) & 1) * (A_ELEMS + B_ELEMS);
            fetch(P1, cur, cur + A_ELEMS, 1);
            mfma_planes(P0);
            __builtin_amdgcn_sched_barrier(0);
            if (nk > 1) {
                wait_loads();
                keep_alive();
            }
            mfma_planes(P1);
            __syncthreads();
        }
    } else if constexpr (X3) {
        // ---- split-bf16 products ("bf16x3"): every fp32 operand value is split EXACTLY into three bf16 terms x = hi + mid + lo (8 + 8 + 8
        // significand bits; each residual of a round-to-nearest is exactly representable), and a product a.b is formed from the six
        // bf16 x bf16 products whose weight is >= 2^-16 of the leading one -- hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid -- on
        // v_mfma_f32_32x32x16_bf16 (products exact, fp32 accumulation).  The dropped terms (mid.lo, lo.mid, lo.lo) are <= 2^-24 relative:
        // the rounding level of an fp32 fma chain itself (tests hold this path to the SAME error bound as the f32-input MFMA path).
        // 16 k per instruction at 32 cycles, six instructions: 12 cycles per k against 32 for v_mfma_f32_32x32x2_f32.
        // The split happens on the fragments (registers), so LDS holds plain fp32 tiles exactly as in the f32 path.  (A variant that
        // splits ONCE at staging into three bf16 LDS planes per operand -- no redundant splits, the multiply phase nothing but fragment
        // reads and MFMAs -- measured the same: 121.9 vs 119.1 us at 256 -> 256 over 131072 tokens.  With a third of the matrix-pipe
        // time this mode is bound by the operand stream (268 MB of HBM traffic = 43 us at that shape, one tile of loads in flight per
        // workgroup), not by the split arithmetic; it was removed again.)
        struct Frag2 { float a[TM][8], b[TN][8]; };
        auto read_frag2 = [&](Frag2 &f, const float *As, const float *Bs, int s2) {
            Frag lo4, hi4;
            read_frag(lo4, As, Bs, 2 * s2);
            read_frag(hi4, As, Bs, 2 * s2 + 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int i = 0; i < TM; ++i) { f.a[i][e] = lo4.a[i][e]; f.a[i][4 + e] = hi4.a[i][e]; }
#pragma unroll
                for (int j = 0; j < TN; ++j) { f.b[j][e] = lo4.b[j][e]; f.b[j][4 + e] = hi4.b[j][e]; }
            }
        };
        // Written on PAIRS of values (two-element vectors): the conversions become one v_cvt_pk_bf16_f32 per pair and the residuals one
        // v_pk_add_f32 per pair.  The scalar form left that pairing to the SLP vectoriser, which packed 40 of the 128 subtractions of a
        // k-step and emitted 128 instead of 96 conversions (383 vector instructions per k-step against 48 MFMAs).  Same operations, same bits.
        auto split8 = [&](const float (&x)[8], bf16x8 &h, bf16x8 &m, bf16x8 &l) {
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
        };
        // The split of the NEXT fragments is independent of the MFMAs on the CURRENT planes: both are issued in one scheduling region so
        // that the ~230 vector instructions of a split run in the shadow of the 24 MFMAs (an MFMA holds the SIMD's vector issue for 8 of
        // its 32 cycles only).  Before this the two phases alternated: PMC showed matrix pipe 39 % + vector ALU 48 % busy, never together.
        // (De-phasing the co-resident workgroups with an initial s_sleep on every other one changed nothing: 117.8 vs 113.5 us.  A form with
        // the WEIGHT split beforehand -- its fragments loaded as 16-byte rows of bf16 planes straight from global memory, the activations
        // split once at staging into LDS planes, no vector arithmetic left in the multiply loop -- was SLOWER: 142 vs 119 us at 256 -> 256
        // over 131072 tokens, 72 vs 54 us at 320 -> 1280 over 8192: fragment-shaped global loads touch 32 cache lines per wave instruction.)
        struct Planes { bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN]; };
        auto split_frag = [&](const Frag2 &f, Planes &P) {
#pragma unroll
            for (int i = 0; i < TM; ++i) split8(f.a[i], P.ah[i], P.am[i], P.al[i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) split8(f.b[j], P.bh[j], P.bm[j], P.bl[j]);
        };
        auto mfma_planes = [&](const Planes &P) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x16 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P.am[i], P.bm[j], c, 0, 0, 0);      // small terms first
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P.al[i], P.bh[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P.ah[i], P.bl[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P.am[i], P.bh[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P.ah[i], P.bm[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P.ah[i], P.bh[j], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        };
        Frag2 g;
        Planes P0, P1;
        read_frag2(g, lds, lds + A_ELEMS, 0);
        split_frag(g, P0);
        for (int kt = 0; kt + 1 < nk; ++kt) {
            const float *cur = lds + (kt & 1) * (A_ELEMS + B_ELEMS);
            float *nxt = lds + ((kt + 1) & 1) * (A_ELEMS + B_ELEMS);
            load_tiles((kt + 1) * BK);
            __builtin_amdgcn_sched_barrier(0);
            read_frag2(g, cur, cur + A_ELEMS, 1);
            split_frag(g, P1);
            mfma_planes(P0);
            __builtin_amdgcn_sched_barrier(0);
            if (FULLK) wait_loads();                      // the pinned (asm) loads are not counted by the compiler
            store_tiles(nxt);
            __syncthreads();
            read_frag2(g, nxt, nxt + A_ELEMS, 0);
            split_frag(g, P0);
            mfma_planes(P1);
        }
        {
            const float *cur = lds + ((nk - 1) & 1) * (A_ELEMS + B_ELEMS);
            read_frag2(g, cur, cur + A_ELEMS, 1);
            split_frag(g, P1);
            mfma_planes(P0);
            mfma_planes(P1);
            __syncthreads();
        }
    } else {
    Frag f0, f1;
    read_frag(f0, lds, lds + A_ELEMS, 0);
    for (int kt = 0; kt + 1 < nk; ++kt) {
        const float *cur = lds + (kt & 1) * (A_ELEMS + B_ELEMS);
        float *nxt = lds + ((kt + 1) & 1) * (A_ELEMS + B_ELEMS);
        load_tiles((kt + 1) * BK);
        __builtin_amdgcn_sched_barrier(0);      // keep the loads HERE: the scheduler otherwise sinks them to just above the LDS stores
        read_frag(f1, cur, cur + A_ELEMS, 1);
        mma(f0);
        read_frag(f0, cur, cur + A_ELEMS, 2);
        mma(f1);
        __builtin_amdgcn_sched_barrier(0);
        if (FULLK) wait_loads();
        store_tiles(nxt);
        read_frag(f1, cur, cur + A_ELEMS, 3);
        mma(f0);
        __syncthreads();
        read_frag(f0, nxt, nxt + A_ELEMS, 0);
        mma(f1);
    }
    {
        const float *cur = lds + ((nk - 1) & 1) * (A_ELEMS + B_ELEMS);
        read_frag(f1, cur, cur + A_ELEMS, 1);
        mma(f0);
        read_frag(f0, cur, cur + A_ELEMS, 2);
        mma(f1);
        read_frag(f1, cur, cur + A_ELEMS, 3);
        mma(f0);
        mma(f1);
        __syncthreads();
    }
    }   // !X3
    // ---- epilogue (EPI: bit 0 = + residual, bit 1 = exact GELU; compile-time, so the plain Linear carries none of their registers) ----
    constexpr bool RES = (EPI & 1) != 0, ACT = (EPI & 2) != 0;
    if (vec_out) {
        // Row-major through LDS (the staging buffers are free after the loop's last barrier): a wave parks 32 rows of its tile in a
        // private [32][TN*32 + 4] image (ds_write_b32 straight from the accumulator layout, conflict-free), reads whole rows back and
        // stores 16 bytes per lane -- full 128/256-byte row segments, a quarter of the store instructions; bias / residual are applied
        // row-wise with 16-byte loads.  Needs 16-byte aligned rows of C, of the residual and of the bias (checked by the launcher).
        constexpr int WCOLS = TN * 32, WP = WCOLS + 4, LPR = WCOLS / 4, RPI = 64 / LPR;   // lanes per row, rows per store instruction
        float *img = lds + wave * (32 * WP);
        const int lr = lane / LPR, lc = (lane % LPR) * 4;
        const int ncol = n0 + wn + lc;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!BIAS_ROW && bias && ncol < N) bv = *reinterpret_cast<const float4 *>(bias + ncol);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) img[((e & 3) + 8 * (e >> 2) + 4 * kh) * WP + 32 * j + r] = acc[i][j][e];
            // (no workgroup barrier: the image is private to the wave and a wave's LDS operations execute in order; the main loop's last
            // barrier has already retired every other wave's reads of the staging buffers this image overlays)
            auto rows_out = [&](auto inner) {
                constexpr bool IN = decltype(inner)::value;          // interior tile: no per-row / per-column predicate at all
#pragma unroll
                for (int rr = 0; rr < 32; rr += RPI) {
                    const long m = m0 + wm + 32 * i + rr + lr;
                    float4 v = *reinterpret_cast<const float4 *>(img + (rr + lr) * WP + lc);
                    if (BIAS_ROW) {
                        const float br = bias ? bias[m < M ? m : M - 1] : 0.f;
                        v.x += br, v.y += br, v.z += br, v.w += br;
                    } else {
                        v.x += bv.x, v.y += bv.y, v.z += bv.z, v.w += bv.w;
                    }
                    if (ACT) v.x = gelu_exact(v.x), v.y = gelu_exact(v.y), v.z = gelu_exact(v.z), v.w = gelu_exact(v.w);
                    if (IN || (m < M && ncol < N)) {
                        if (RES) {
                            const float4 rv = *reinterpret_cast<const float4 *>(residual + m * ldc + ncol);
                            v.x += rv.x, v.y += rv.y, v.z += rv.z, v.w += rv.w;
                        }
                        *reinterpret_cast<float4 *>(C + m * ldc + ncol) = v;
                    }
                }
            };
            if (m0 + BM <= M && n0 + BN <= N) rows_out(std::true_type{});
            else rows_out(std::false_type{});
        }
#ifdef SD_GEMM_STAMPS
        if (BFRAG && EPI == 0 && residual && (blockIdx.x & 63) == 0 && threadIdx.x == 0) {
            unsigned long long *o = reinterpret_cast<unsigned long long *>(const_cast<float *>(residual)) + (blockIdx.x >> 6) * 12;
            o[10] = __builtin_amdgcn_s_memtime();                     // absolute end (stores issued, not necessarily drained)
        }
#endif
        return;
    }
    // unaligned output (N % 4 != 0: the 150 classes of linear_pred): element stores in the accumulator layout -- col = lane & 31,
    // row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn + 32 * j + r;
        const bool nin = n < N;
        const int nc = nin ? n : N - 1;
        const float bv = (bias && !BIAS_ROW) ? bias[nc] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long m = m0 + wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * kh;
                const long mc = m < M ? m : M - 1;
                float v = acc[i][j][e] + (BIAS_ROW ? (bias ? bias[mc] : 0.f) : bv);
                if (ACT) v = gelu_exact(v);
                if (RES) v += residual[mc * ldc + nc];
                if (nin && m < M) C[m * ldc + nc] = v;
            }
        }
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool BT, int EPI, bool AKC = true, bool BIAS_ROW = false, bool X3 = false, bool BFRAG = false,
          bool APL = false>
int launch_epi(const float *A, const float *Bm, float *C, const float *bias, const float *residual, long M, int N, int K, long lda, long ldb,
               long ldc, hipStream_t st, int batch = 1, long sA = 0, long sB = 0, long sC = 0, int nsplit = 1, int klen = 0) {
    if (klen <= 0 || nsplit <= 1) { nsplit = 1; klen = K; }
    // two staging buffers, or ONE when the whole reduction is a single k-step (K <= 32: the stage-1 products -- HBM-bound, so what matters
    // there is how many workgroups, i.e. bytes in flight, a CU holds)
    // The row-major epilogue parks 4 waves x 32 rows x (TN*32 + 4) floats in the same LDS: with ONE staging buffer of two [k][m] / [k][n]
    // operands (bwd-data of the class-plane Linear with <= 32 classes: 33792 B) that image (34816 B) is the larger of the two.
    constexpr size_t epi_bytes = (size_t)4 * 32 * ((BN / (32 * WAVES_N)) * 32 + 4) * sizeof(float);
    const size_t stage_bytes = (klen > BK ? 2 : 1) * (size_t)((APL ? 3 * BM * 20 : (AKC ? BM * KP : BK * (BM + 4))) + (BFRAG ? 0 : (BT ? BN * KP : BK * (BN + 4)))) * sizeof(float);
    const size_t lds_bytes = stage_bytes > epi_bytes ? stage_bytes : epi_bytes;
    const long tiles_m = (M + BM - 1) / BM;
    const int tiles_n = (N + BN - 1) / BN;
    const long nblk = tiles_m * tiles_n;
    if (nblk > 0x7fffffffL || batch < 1 || (long)batch * nsplit > 65535) return SD_E_SHAPE;
    // 16-byte loads need aligned rows: base pointers, leading dimensions, batch strides, and the contiguous axis of each operand a multiple of 4
    const bool vec = (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(Bm) & 15) == 0 && lda % 4 == 0 && ldb % 4 == 0 &&
                     sA % 4 == 0 && sB % 4 == 0 && ((AKC || BT) ? klen % 4 == 0 : true) && (AKC ? K % 4 == 0 : M % 4 == 0) && (BT ? K % 4 == 0 : N % 4 == 0);
    // (a k-contiguous operand needs k origins on 16-byte boundaries; [k][m] / [k][n] operands vectorise along m / n whatever K is)
    const bool fullk = vec && K % BK == 0 && klen % BK == 0 && (AKC || M >= 4) && (BT || N >= 4);
    if constexpr (BFRAG) {
        // pre-split planes: A alone decides alignment; whole k-steps, no split-K
        const bool ok = (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(Bm) & 15) == 0 && lda % 4 == 0 && K % BK == 0 &&
                        nsplit == 1 && batch == 1 && ldb > 0;
        if (!ok) return SD_E_UNSUPPORTED;
        auto kernf = token_gemm_f32<BM, BN, WAVES_M, WAVES_N, BT, true, true, EPI, AKC, BIAS_ROW, true, true>;
        if (lds_bytes > 64 * 1024) {      // the row-major epilogue image of a 256-column tile: 66 KB
            static bool raised = false;   // per instantiation; idempotent, so a race is harmless
            if (!raised) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
                if (e != hipSuccess) return (int)e;
                raised = true;
            }
        }
        const int vo = ((reinterpret_cast<uintptr_t>(C) & 15) == 0 && ldc % 4 == 0 && N % 4 == 0 &&
                        (!residual || (reinterpret_cast<uintptr_t>(residual) & 15) == 0) &&
                        (BIAS_ROW || !bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0)) ? 1 : 0;
        hipLaunchKernelGGL(kernf, dim3((unsigned)nblk, 1), dim3(256), lds_bytes, st, A, Bm, C, bias, residual, M, N, K, lda, ldb, ldc, tiles_n, vo, 0L, 0L,
                           0L, 1, K);
        return (int)hipGetLastError();
    }
    else {
    // (pre-split A planes: the B operand alone decides the alignment; the caller guarantees whole k-steps and no split-K)
    if (APL && !((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(Bm) & 15) == 0 && ldb % 4 == 0 && sB % 4 == 0 && K % BK == 0 &&
                 nsplit == 1 && sA == 0))
        return SD_E_UNSUPPORTED;
    auto pick = [&]() {
        if constexpr (APL) return token_gemm_f32<BM, BN, WAVES_M, WAVES_N, BT, true, true, EPI, AKC, BIAS_ROW, X3, false, true>;
        else
            return fullk ? token_gemm_f32<BM, BN, WAVES_M, WAVES_N, BT, true, true, EPI, AKC, BIAS_ROW, X3>
                         : (vec ? token_gemm_f32<BM, BN, WAVES_M, WAVES_N, BT, true, false, EPI, AKC, BIAS_ROW, X3>
                                : token_gemm_f32<BM, BN, WAVES_M, WAVES_N, BT, false, false, EPI, AKC, BIAS_ROW, X3>);
    };
    auto kern = pick();
    if (lds_bytes > 64 * 1024) {
        static bool raised[3] = {false, false, false};   // per instantiation (this function template) and load flavour; idempotent, so a race is harmless
        const int flavour = fullk ? 2 : (vec ? 1 : 0);
        if (!raised[flavour]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return (int)e;
            raised[flavour] = true;
        }
    }
    // row-major 16-byte epilogue: rows of C (and of the residual, a per-column bias vector) 16-byte aligned and N % 4 == 0
    const int vec_out = ((reinterpret_cast<uintptr_t>(C) & 15) == 0 && ldc % 4 == 0 && sC % 4 == 0 && N % 4 == 0 &&
                         (!residual || (reinterpret_cast<uintptr_t>(residual) & 15) == 0) &&
                         (BIAS_ROW || !bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0)) ? 1 : 0;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)(batch * nsplit)), dim3(256), lds_bytes, st, A, Bm, C, bias, residual, M, N, K, lda, ldb,
                       ldc, tiles_n, vec_out, sA, sB, sC, nsplit, klen);
    return (int)hipGetLastError();
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool BT>
int launch(const float *A, const float *Bm, float *C, const float *bias, const float *residual, long M, int N, int K, long lda, long ldb, long ldc,
           int act, hipStream_t st) {
    if constexpr (!BT) {    // bwd-data: never an epilogue
        return launch_epi<BM, BN, WAVES_M, WAVES_N, BT, 0>(A, Bm, C, bias, nullptr, M, N, K, lda, ldb, ldc, st);
    } else {
        const int epi = (residual ? 1 : 0) | (act == 1 ? 2 : 0);
        switch (epi) {
            case 0: return launch_epi<BM, BN, WAVES_M, WAVES_N, BT, 0>(A, Bm, C, bias, residual, M, N, K, lda, ldb, ldc, st);
            case 1: return launch_epi<BM, BN, WAVES_M, WAVES_N, BT, 1>(A, Bm, C, bias, residual, M, N, K, lda, ldb, ldc, st);
            case 2: return launch_epi<BM, BN, WAVES_M, WAVES_N, BT, 2>(A, Bm, C, bias, residual, M, N, K, lda, ldb, ldc, st);
            default: return launch_epi<BM, BN, WAVES_M, WAVES_N, BT, 3>(A, Bm, C, bias, residual, M, N, K, lda, ldb, ldc, st);
        }
    }
}

// split-bf16 arithmetic (mode 1): only the 128x128 tile (these are the MFMA-bound, wide products) and the plain / + residual epilogues.
// (128x64 and 64x64 tiles for the stage 2-3 products that leave CUs empty -- 8192 x 320 -> 320 is 192 tiles of 128x128 -- were measured at
// 22.0 / 21.8 / 20.4 us for that shape and 14.9 / 15.3 / 15.7 us for 32768 x 128 -> 128: tile count is not what bounds them.)
template <bool BT>
int dispatch_x3(const float *A, const float *Bm, float *C, const float *bias, const float *residual, long M, int N, int K, long lda, long ldb,
                long ldc, hipStream_t st) {
    if (residual) return launch_epi<128, 128, 2, 2, BT, 1, true, false, true>(A, Bm, C, bias, residual, M, N, K, lda, ldb, ldc, st);
    return launch_epi<128, 128, 2, 2, BT, 0, true, false, true>(A, Bm, C, bias, nullptr, M, N, K, lda, ldb, ldc, st);
}

// ---- pre-split weights (round 3) -------------------------------------------------------------------------------------------------------
// B(k, n) of a Linear product as three bf16 planes (hi = rn(x), mid = rn(x - hi), lo = rn(x - hi - mid): the split of the X3 kernels, the same
// bits) in FRAGMENT-CONTIGUOUS order: for every (32-column block nb, 16-deep k-step ks, plane) one 1 KB run whose lane-l 16 bytes hold
// B(16 ks + 8 (l >> 5) + e, 32 nb + (l & 31)), e = 0..7 -- exactly lane l's B operand of v_mfma_f32_32x32x16_bf16.  Columns beyond the
// matrix are zero.   dir 1 (forward):  B(k, n) = W[n][k]   (Nd = out_features, Kd = in_features)
//                    dir 0 (bwd-data): B(k, n) = W[k][n]   (Nd = in_features,  Kd = out_features)
constexpr int kMaxPresplit = 64;
struct PresplitTable {   // by value in the kernel arguments (no device-side table: safe under graph capture)
    const float *W[kMaxPresplit];
    uint16_t *out[kMaxPresplit];
    int ldw[kMaxPresplit], Nd[kMaxPresplit], Kd[kMaxPresplit], dir[kMaxPresplit];
    int blk_begin[kMaxPresplit + 1];
    int njobs;
};

__global__ __launch_bounds__(256) void presplit_planes(const PresplitTable t) {
    int lo = 0, hi = t.njobs - 1;
    while (lo < hi) {        // wave-uniform binary search
        const int mid = (lo + hi + 1) >> 1;
        if ((int)blockIdx.x >= t.blk_begin[mid]) lo = mid;
        else hi = mid - 1;
    }
    const int j = lo;
    if (t.dir[j] == 2) {
        // ROW-MAJOR planes [plane][Nd rows][Kd] of W [Nd][ldw]: the A operand of the APL kernels (staged through LDS as plain 2-D tiles)
        const int kc = t.Kd[j] / 8;
        const long g = (long)((int)blockIdx.x - t.blk_begin[j]) * 256 + threadIdx.x;
        if (g >= (long)t.Nd[j] * kc) return;
        const int row = (int)(g / kc), c8 = (int)(g - (long)row * kc);
        const float *src = t.W[j] + (long)row * t.ldw[j] + 8 * c8;
        bf16x8 h, m, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = src[e];
            const __bf16 hh = static_cast<__bf16>(x);
            const float r1 = x - static_cast<float>(hh);
            const __bf16 mm = static_cast<__bf16>(r1);
            h[e] = hh, m[e] = mm, l[e] = static_cast<__bf16>(r1 - static_cast<float>(mm));
        }
        const size_t plane = (size_t)t.Nd[j] * t.Kd[j];
        uint16_t *dst = t.out[j] + (size_t)row * t.Kd[j] + 8 * c8;
        *reinterpret_cast<bf16x8 *>(dst) = h;
        *reinterpret_cast<bf16x8 *>(dst + plane) = m;
        *reinterpret_cast<bf16x8 *>(dst + 2 * plane) = l;
        return;
    }
    const int KS = t.Kd[j] / 16, NBk = (t.Nd[j] + 31) / 32;
    const long f = (long)((int)blockIdx.x - t.blk_begin[j]) * 4 + (threadIdx.x >> 6);      // fragment = (nb, ks)
    if (f >= (long)NBk * KS) return;
    const int lane = threadIdx.x & 63;
    const int nb = (int)(f / KS), ks = (int)(f - (long)nb * KS);
    const int n = 32 * nb + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
    const float *__restrict__ W = t.W[j];
    const long ldw = t.ldw[j];
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = 0.f;
    if (n < t.Nd[j]) {
        if (t.dir[j]) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = W[(long)n * ldw + k0 + e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = W[(long)(k0 + e) * ldw + n];
        }
    }
    bf16x8 h, m, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 hh = static_cast<__bf16>(x[e]);
        const float r1 = x[e] - static_cast<float>(hh);
        const __bf16 mm = static_cast<__bf16>(r1);
        const float r2 = r1 - static_cast<float>(mm);
        h[e] = hh, m[e] = mm, l[e] = static_cast<__bf16>(r2);
    }
    bf16x8 *dst = reinterpret_cast<bf16x8 *>(t.out[j]) + (size_t)f * 3 * 64 + lane;
    dst[0] = h;
    dst[64] = m;
    dst[128] = l;
}

#ifdef SD_GEMM_STAMPS
const float *g_gemm_stamp_buf = nullptr;     // diagnostic build: device buffer of 8 x u64 per 64 workgroups (sd_debug_gemm_stamps)
#endif

int g_planes_tile = 0;        // tunable "planes_tile"

// X3 products on a pre-split B operand (planes: `nblocks` 32-column blocks x K/16 k-steps)
int dispatch_planes(const float *A, const void *planes, int nblocks, float *C, const float *bias, const float *residual, long M, int N, int K,
                    hipStream_t st) {
    const float *Bp = static_cast<const float *>(planes);
    // 128 x 128 tiles, two workgroups per CU.  (Measured and dropped: 128 x 256 tiles -- each wave 64 x 128, 96 MFMAs per k-step against the
    // same A split, ~330 registers, ONE workgroup per CU -- 109.7 vs 105.1 us at 256 -> 256 over 131072 tokens, 47.3 vs 43.6 us at 64 -> 256,
    // 55.7 vs 45.4 us at 320 -> 1280 over 8192: PMC shows the same ~50 % matrix-pipe occupancy with less co-resident work to cover the waits.)
    // 64-row tiles (each wave 32 x 64: the same MFMAs per split fragment, twice the workgroups) where 128 x 128 tiles leave the chip short
    // of work: fewer than 160 of them (8192 tokens x 256 columns = 128 workgroups on 256 CUs, one wave per SIMD running its k-steps at the
    // memory latency), or fewer than 384 with a short reduction.  Same box, planes forward, 128 -> 64 rows (library): 2048 x 256 -> 1024
    // 14.3 -> 11.4 us (14.1), 8192 x 256 -> 256 14.1 -> 11.6 (13.3), 8192 x 160 -> 640 18.8 -> 17.1 (21.0) and its input gradient 27.1 ->
    // 20.7 (17.4); but 8192 x 1280 -> 320 (192 tiles) 50.9 -> 58.3 and 32768 x 512 -> 128 28.0 -> 30.1: the weight fragments are then fetched by
    // twice as many workgroups.  Tunable "planes_tile": 0 = this rule, 128 / 64 = force (tests, A/B: SEGDISTILL_PLANES_TILE).
    const long tiles128 = ((M + 127) / 128) * ((N + 127) / 128);
    const bool small = g_planes_tile == 64 || (g_planes_tile == 0 && M > 64 && (tiles128 < 160 || (tiles128 < 384 && K <= 256)));
    if (small) {
        if (residual) return launch_epi<64, 128, 2, 2, true, 1, true, false, true, true>(A, Bp, C, bias, residual, M, N, K, K, nblocks, N, st);
        return launch_epi<64, 128, 2, 2, true, 0, true, false, true, true>(A, Bp, C, bias, nullptr, M, N, K, K, nblocks, N, st);
    }
    if (residual) return launch_epi<128, 128, 2, 2, true, 1, true, false, true, true>(A, Bp, C, bias, residual, M, N, K, K, nblocks, N, st);
#ifdef SD_GEMM_STAMPS
    return launch_epi<128, 128, 2, 2, true, 0, true, false, true, true>(A, Bp, C, bias, g_gemm_stamp_buf, M, N, K, K, nblocks, N, st);
#else
    return launch_epi<128, 128, 2, 2, true, 0, true, false, true, true>(A, Bp, C, bias, nullptr, M, N, K, K, nblocks, N, st);
#endif
}

template <bool BT>
int dispatch(const float *A, const float *Bm, float *C, const float *bias, const float *residual, long M, int N, int K, long lda, long ldb, long ldc,
             int act, hipStream_t st) {
    // widest column tile whose padding wastes at most a quarter of the MFMAs (N = 150 / 160: 3 x 64 instead of 2 x 128)
    auto waste_ok = [&](int bn) { return (long)((N + bn - 1) / bn) * bn * 4 <= (long)N * 5; };
    if (N > 64 && waste_ok(128)) return launch<128, 128, 2, 2, BT>(A, Bm, C, bias, residual, M, N, K, lda, ldb, ldc, act, st);
    if (N > 32) return launch<128, 64, 2, 2, BT>(A, Bm, C, bias, residual, M, N, K, lda, ldb, ldc, act, st);
    return launch<256, 32, 4, 1, BT>(A, Bm, C, bias, residual, M, N, K, lda, ldb, ldc, act, st);
}

}  // namespace

// ---- the NCHW 1x1 feature-align projection in fp32 on the same kernel (entry points in align1x1.hip; SURVEY a-15) --------------------------
//   forward   Y_b [Ct x P] = W [Ct x Cs] . X_b [Cs x P] + bias[Ct]      A = W (k-contiguous, shared by the batch), B = X_b ([k][n]), row bias
//   bwd-data  dX_b [Cs x P] = W^T . dY_b [Ct x P]                        A = W read as [k][m] (m-contiguous), B = dY_b ([k][n])
//   bwd-wgt   slab_z [Ct x Cs] = dY_b[:, chunk] . X_b[:, chunk]^T        A = dY_b ([m][k]), B = X_b ([n][k]); z = (image, pixel chunk), combined
//                                                                         by the caller's deterministic slab reduction
int g_align_split_bf16 = 1;   // tunable "align_split_bf16": the three align products in split-bf16 arithmetic (same error bound as f32 MFMA)

int align_f32_fwd(const float *X, const float *W, const float *bias, float *Y, int B, int Cs, int Ct, long P, hipStream_t st) {
    if (g_align_split_bf16) {
        // short reductions (Cs <= 128): the streaming kernel with W resident in registers (csrc/align_stream.hip); SD_E_UNSUPPORTED = not its shape
        const int rc = align_f32_fwd_stream(X, W, bias, Y, B, Cs, Ct, P, st);
        if (rc != SD_E_UNSUPPORTED) return rc;
    }
    if (g_align_split_bf16 && Cs % 32 == 0)
        return launch_epi<128, 128, 2, 2, false, 0, true, true, true>(W, X, Y, bias, nullptr, Ct, (int)P, Cs, Cs, P, P, st, B, 0L, (long)Cs * P,
                                                                      (long)Ct * P);
    return launch_epi<128, 128, 2, 2, false, 0, true, true>(W, X, Y, bias, nullptr, Ct, (int)P, Cs, Cs, P, P, st, B, 0L, (long)Cs * P, (long)Ct * P);
}

int align_f32_bwd_data(const float *dY, const float *W, float *dX, int B, int Cs, int Ct, long P, hipStream_t st) {
    if (g_align_split_bf16 && Ct % 32 == 0)
        return launch_epi<128, 128, 2, 2, false, 0, false, false, true>(W, dY, dX, nullptr, nullptr, Cs, (int)P, Ct, Cs, P, P, st, B, 0L, (long)Ct * P,
                                                                        (long)Cs * P);
    return launch_epi<128, 128, 2, 2, false, 0, false, false>(W, dY, dX, nullptr, nullptr, Cs, (int)P, Ct, Cs, P, P, st, B, 0L, (long)Ct * P,
                                                              (long)Cs * P);
}

int align_f32_bwd_weight_slabs(const float *dY, const float *X, float *slabs, int B, int Cs, int Ct, long P, int nsplit, int klen, hipStream_t st) {
    // z = image * nsplit + pixel chunk; slab z = dY_b[:, chunk] . X_b[:, chunk]^T  (both operands k-contiguous along the pixels)
    if (g_align_split_bf16 && klen % 32 == 0 && P % 32 == 0)
        return launch_epi<128, 128, 2, 2, true, 0, true, false, true>(dY, X, slabs, nullptr, nullptr, Ct, Cs, (int)P, P, P, Cs, st, B, (long)Ct * P,
                                                                      (long)Cs * P, (long)Ct * Cs, nsplit, klen);
    return launch_epi<128, 128, 2, 2, true, 0, true, false>(dY, X, slabs, nullptr, nullptr, Ct, Cs, (int)P, P, P, Cs, st, B, (long)Ct * P, (long)Cs * P,
                                                            (long)Ct * Cs, nsplit, klen);
}

// ---- a Linear from TOKEN-MAJOR features to NCHW planes: the SegFormer head's `linear_pred` (1x1 conv, segformer_head.py:73,96) -------------
// The head's fused feature map is token-major ([B, P, E], P = h*w) while everything that reads the logits (the fused up-sample + CE kernel, the
// fused up-sample + CGD kernel) wants contiguous class planes [B, classes, P].  As a token Linear the product needs a 79 MB transpose copy of
// its output and another of the incoming gradient (80 us each at config 2); with the operand roles swapped the same kernel writes / reads the
// planes directly:
//   forward   out_b [classes x P] = W [classes x E] . tokens_b^T + bias     A = W ([m][k]), B = tokens_b ([n][k]), per-row bias, batch = images
//   bwd-data  dX_b  [P x E]       = dOut_b^T . W                            A = dOut_b read as [k][m] (m contiguous), B = W ([k][n])
//   bwd-wgt   slab_z [classes x E] = dOut_b[:, chunk] . tokens_b[chunk, :]   A = dOut_b ([m][k]), B = tokens_b ([k][n]); z = (image, pixel chunk)
__global__ __launch_bounds__(256) void plane_rowsum_partials(const float *__restrict__ dOut, float *__restrict__ part, int C, long P) {
    const int c = blockIdx.x, b = blockIdx.y;
    const float *row = dOut + ((long)b * C + c) * P;
    float acc = 0.f;
    if ((P & 3) == 0 && (reinterpret_cast<uintptr_t>(row) & 15) == 0) {
        for (long p = 4L * threadIdx.x; p < P; p += 1024) {
            const float4 v = *reinterpret_cast<const float4 *>(row + p);
            acc += (v.x + v.y) + (v.z + v.w);
        }
    } else {
        for (long p = threadIdx.x; p < P; p += 256) acc += row[p];
    }
    __shared__ float red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[(long)b * C + c] = (red[0] + red[1]) + (red[2] + red[3]);
}
// dW = sum of the nz slabs (fixed order); the last workgroup also folds the per-image bias partials
__global__ __launch_bounds__(256) void pred_wgrad_reduce(const float *__restrict__ slabs, float *__restrict__ dW, long n, int nz,
                                                          const float *__restrict__ bias_part, float *__restrict__ db, int C, int B) {
    if ((int)blockIdx.x == (int)gridDim.x - 1) {
        if (db)
            for (int c = threadIdx.x; c < C; c += 256) {
                float a = 0.f;
                for (int b = 0; b < B; ++b) a += bias_part[(long)b * C + c];
                db[c] = a;
            }
        return;
    }
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float a = 0.f;
    for (int z = 0; z < nz; ++z) a += slabs[(long)z * n + i];
    dW[i] = a;
}

int g_pred_split_bf16 = 1;   // follows tunable "align_split_bf16" (set together): split-bf16 arithmetic for the three products

int pred_splits(int B, long P) {
    int per_img = (int)((P + 2047) / 2048);
    if (per_img < 1) per_img = 1;
    while ((long)B * per_img < 64 && P / per_img > 512) per_img *= 2;
    return per_img;
}

int token_gemm_tunable(const char *key, int set, int v) {
    if (!strcmp(key, "planes_tile")) {
        if (!set) return g_planes_tile;
        if (v != 0 && v != 64 && v != 128) return SD_E_SHAPE;
        g_planes_tile = v;
        return SD_OK;
    }
    if (strcmp(key, "align_split_bf16")) return SD_E_UNSUPPORTED;
    if (set) {
        if (v != 0 && v != 1) return SD_E_SHAPE;
        g_align_split_bf16 = v;
        g_pred_split_bf16 = v;
        return SD_OK;
    }
    return g_align_split_bf16;
}

size_t linear_nchw_workspace_bytes(int B, long P, int in_features, int out_features) {
    if (B <= 0 || P <= 0 || in_features <= 0 || out_features <= 0) return 0;
    const int nsplit = pred_splits(B, P);
    return ((size_t)B * nsplit * out_features * in_features + (size_t)B * out_features) * sizeof(float) + 16;
}

int linear_nchw_f32_fwd(const float *X, const float *W, const float *bias, float *Y, int B, long P, int in_features, int out_features, void *stream) {
    if (!X || !W || !Y) return SD_E_NULL;
    if (B <= 0 || P <= 0 || in_features <= 0 || out_features <= 0 || B > 65535) return SD_E_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long sB = P * in_features, sC = P * out_features;
    if (g_pred_split_bf16 && in_features % 32 == 0) {
        // 129 ... 160 output rows (the 150 classes): ONE 160-row tile (five 32-row MFMA blocks per wave, four waves along the pixels) instead of
        // two 128-row tiles of which the second is 83 % padding
        if (out_features > 128 && out_features <= 160)
            return launch_epi<160, 128, 1, 4, true, 0, true, true, true>(W, X, Y, bias, nullptr, out_features, (int)P, in_features, in_features,
                                                                             in_features, P, st, B, 0L, sB, sC);
        return launch_epi<128, 128, 2, 2, true, 0, true, true, true>(W, X, Y, bias, nullptr, out_features, (int)P, in_features, in_features, in_features, P,
                                                                         st, B, 0L, sB, sC);
    }
    return launch_epi<128, 128, 2, 2, true, 0, true, true>(W, X, Y, bias, nullptr, out_features, (int)P, in_features, in_features, in_features, P, st, B,
                                                               0L, sB, sC);
}

// the same forward with W given as pre-split ROW-MAJOR planes (sd_presplit_multi, row_planes): the 160-row tile, A fragments from LDS planes
int linear_nchw_f32_fwd_planes(const float *X, const void *w_row_planes, const float *bias, float *Y, int B, long P, int in_features, int out_features,
                               void *stream) {
    if (!X || !w_row_planes || !Y) return SD_E_NULL;
    if (B <= 0 || P <= 0 || in_features <= 0 || out_features <= 0 || B > 65535) return SD_E_SHAPE;
    if (in_features % 32 || out_features > 160) return SD_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long sB = P * in_features, sC = P * out_features;
    return launch_epi<160, 128, 1, 4, true, 0, true, true, true, false, true>(static_cast<const float *>(w_row_planes), X, Y, bias, nullptr, out_features, (int)P,
                                                                                  in_features, in_features, in_features, P, st, B, 0L, sB, sC);
}

int linear_nchw_f32_bwd_data(const float *dY, const float *W, float *dX, int B, long P, int in_features, int out_features, void *stream) {
    if (!dY || !W || !dX) return SD_E_NULL;
    if (B <= 0 || P <= 0 || in_features <= 0 || out_features <= 0 || B > 65535) return SD_E_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // C_b [P x in] = A_b . B with A_b = dY_b read as [k = class][m = pixel], B = W [k = class][n = in]
    if (g_pred_split_bf16)
        return launch_epi<128, 128, 2, 2, false, 0, false, false, true>(dY, W, dX, nullptr, nullptr, P, in_features, out_features, P, in_features,
                                                                            in_features, st, B, P * out_features, 0L, P * in_features);
    return launch_epi<128, 128, 2, 2, false, 0, false, false>(dY, W, dX, nullptr, nullptr, P, in_features, out_features, P, in_features, in_features, st, B,
                                                                  P * out_features, 0L, P * in_features);
}

int linear_nchw_f32_bwd_weight(const float *dY, const float *X, float *dW, float *dbias, int B, long P, int in_features, int out_features, void *workspace,
                              size_t workspace_bytes, void *stream) {
    if (!dY || !X || !dW || !workspace) return SD_E_NULL;
    if (B <= 0 || P <= 0 || in_features <= 0 || out_features <= 0 || B > 65535) return SD_E_SHAPE;
    if (workspace_bytes < linear_nchw_workspace_bytes(B, P, in_features, out_features) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return SD_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int nsplit = pred_splits(B, P);
    const int klen = (int)(((P + nsplit - 1) / nsplit + 31) / 32 * 32);
    nsplit = (int)((P + klen - 1) / klen);
    const long slab = (long)out_features * in_features;
    float *slabs = static_cast<float *>(workspace);
    float *bias_part = slabs + (size_t)B * pred_splits(B, P) * slab;
    int rc;
    if (g_pred_split_bf16 && klen % 32 == 0 && P % 32 == 0 && out_features > 128 && out_features <= 160)   // one 160-row tile, see the forward
        rc = launch_epi<160, 128, 1, 4, false, 0, true, false, true>(dY, X, slabs, nullptr, nullptr, out_features, in_features, (int)P, P, in_features,
                                                                         in_features, st, B, P * out_features, P * in_features, slab, nsplit, klen);
    else if (g_pred_split_bf16 && klen % 32 == 0 && P % 32 == 0)
        rc = launch_epi<128, 128, 2, 2, false, 0, true, false, true>(dY, X, slabs, nullptr, nullptr, out_features, in_features, (int)P, P, in_features,
                                                                         in_features, st, B, P * out_features, P * in_features, slab, nsplit, klen);
    else
        rc = launch_epi<128, 128, 2, 2, false, 0, true, false>(dY, X, slabs, nullptr, nullptr, out_features, in_features, (int)P, P, in_features, in_features,
                                                                   st, B, P * out_features, P * in_features, slab, nsplit, klen);
    if (rc) return rc;
    if (dbias) hipLaunchKernelGGL(plane_rowsum_partials, dim3(out_features, B), dim3(256), 0, st, dY, bias_part, out_features, P);
    hipLaunchKernelGGL(pred_wgrad_reduce, dim3((unsigned)((slab + 255) / 256 + 1)), dim3(256), 0, st, slabs, dW, slab, B * nsplit, bias_part, dbias,
                       out_features, B);
    return (int)hipGetLastError();
}

}  // namespace sd

extern "C" {

int sd_linear_fwd(const void *X, const float *W, long w_row_stride, const float *bias, const void *residual, void *Y, int dtype, long tokens,
                  int in_features, int out_features, int act, int mode, void *stream) {
    if (!X || !W || !Y) return SD_E_NULL;
    if (dtype != SD_F32) return SD_E_DTYPE;
    if (tokens <= 0 || in_features <= 0 || out_features <= 0) return SD_E_SHAPE;
    if (w_row_stride == 0) w_row_stride = in_features;
    if (w_row_stride < in_features) return SD_E_SHAPE;
    if (act != 0 && act != 1) return SD_E_UNSUPPORTED;
    if (mode != 0 && mode != 1) return SD_E_UNSUPPORTED;
    if (mode == 1) {
        if (act != 0) return SD_E_UNSUPPORTED;
        return sd::dispatch_x3<true>((const float *)X, W, (float *)Y, bias, (const float *)residual, tokens, out_features, in_features, in_features,
                                     w_row_stride, out_features, static_cast<hipStream_t>(stream));
    }
    return sd::dispatch<true>((const float *)X, W, (float *)Y, bias, (const float *)residual, tokens, out_features, in_features, in_features,
                              w_row_stride, out_features, act, static_cast<hipStream_t>(stream));
}

int sd_linear_bwd_data(const void *dY, const float *W, long w_row_stride, void *dX, int dtype, long tokens, int in_features, int out_features,
                       int mode, void *stream) {
    if (!dY || !W || !dX) return SD_E_NULL;
    if (dtype != SD_F32) return SD_E_DTYPE;
    if (tokens <= 0 || in_features <= 0 || out_features <= 0) return SD_E_SHAPE;
    if (w_row_stride == 0) w_row_stride = in_features;
    if (w_row_stride < in_features) return SD_E_SHAPE;
    // C[T x in] = dY[T x out] . W[out x in]: K = out_features, B(k, n) = W[k * ldw + n]
    if (mode != 0 && mode != 1) return SD_E_UNSUPPORTED;
    if (mode == 1)
        return sd::dispatch_x3<false>((const float *)dY, W, (float *)dX, nullptr, nullptr, tokens, in_features, out_features, out_features, w_row_stride,
                                      in_features, static_cast<hipStream_t>(stream));
    return sd::dispatch<false>((const float *)dY, W, (float *)dX, nullptr, nullptr, tokens, in_features, out_features, out_features, w_row_stride,
                               in_features, 0, static_cast<hipStream_t>(stream));
}

size_t sd_presplit_bytes(int n_cols, int k_depth) {
    if (n_cols <= 0 || k_depth <= 0 || k_depth % 16) return 0;
    return (size_t)((n_cols + 31) / 32) * (k_depth / 16) * 3 * 1024;
}

int sd_presplit_multi(const sd_presplit_job *jobs, int njobs, void *stream) {
    if (njobs < 0) return SD_E_SHAPE;
    if (njobs == 0) return SD_OK;
    if (!jobs) return SD_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    sd::PresplitTable t{};
    int cnt = 0;
    long blocks = 0;
    auto flush = [&]() {
        if (!cnt) return;
        t.blk_begin[cnt] = (int)blocks;
        t.njobs = cnt;
        hipLaunchKernelGGL(sd::presplit_planes, dim3((unsigned)blocks), dim3(256), 0, st, t);
        cnt = 0;
        blocks = 0;
    };
    for (int q = 0; q < njobs; ++q) {
        const sd_presplit_job &jb = jobs[q];
        if (!jb.W || (!jb.fwd_planes && !jb.bwd_planes && !jb.row_planes)) return SD_E_NULL;
        if (jb.out_features <= 0 || jb.in_features <= 0 || jb.w_row_stride < jb.in_features) return SD_E_SHAPE;
        if ((reinterpret_cast<uintptr_t>(jb.fwd_planes) | reinterpret_cast<uintptr_t>(jb.bwd_planes) | reinterpret_cast<uintptr_t>(jb.row_planes)) & 15)
            return SD_E_ALIGN;
        for (int dir = 2; dir >= 0; --dir) {
            void *out = dir == 2 ? jb.row_planes : (dir ? jb.fwd_planes : jb.bwd_planes);
            if (!out) continue;
            const int Nd = dir ? jb.out_features : jb.in_features, Kd = dir ? jb.in_features : jb.out_features;
            if (dir == 2) {
                if (Kd % 8 || (reinterpret_cast<uintptr_t>(jb.W) & 3)) return SD_E_UNSUPPORTED;
                if (cnt == sd::kMaxPresplit) flush();
                t.W[cnt] = jb.W; t.out[cnt] = static_cast<uint16_t *>(out); t.ldw[cnt] = (int)jb.w_row_stride;
                t.Nd[cnt] = Nd; t.Kd[cnt] = Kd; t.dir[cnt] = 2; t.blk_begin[cnt] = (int)blocks;
                blocks += ((long)Nd * (Kd / 8) + 255) / 256;
                ++cnt;
                continue;
            }
            if (Kd % 16) return SD_E_UNSUPPORTED;
            if (cnt == sd::kMaxPresplit) flush();
            t.W[cnt] = jb.W;
            t.out[cnt] = static_cast<uint16_t *>(out);
            t.ldw[cnt] = (int)jb.w_row_stride;
            t.Nd[cnt] = Nd;
            t.Kd[cnt] = Kd;
            t.dir[cnt] = dir;
            t.blk_begin[cnt] = (int)blocks;
            blocks += ((long)((Nd + 31) / 32) * (Kd / 16) + 3) / 4;
            if (blocks > 0x7fffffffL) return SD_E_SHAPE;
            ++cnt;
        }
    }
    flush();
    return (int)hipGetLastError();
}

size_t sd_presplit_rows_bytes(int out_features, int in_features) {
    if (out_features <= 0 || in_features <= 0 || in_features % 8) return 0;
    return (size_t)out_features * in_features * 3 * 2;
}

int sd_linear_nchw_fwd_planes(const void *X, const void *w_row_planes, const float *bias, void *Y, int dtype, int B, long P, int in_features,
                              int out_features, void *stream) {
    if (dtype != SD_F32) return SD_E_DTYPE;
    return sd::linear_nchw_f32_fwd_planes((const float *)X, w_row_planes, bias, (float *)Y, B, P, in_features, out_features, stream);
}

#ifdef SD_GEMM_STAMPS
int sd_debug_gemm_stamps(const void *buf) {      // diagnostic A/B build only (not in the header): where sd_linear_fwd_planes writes its phase sums
    sd::g_gemm_stamp_buf = static_cast<const float *>(buf);
    return SD_OK;
}
#endif

int sd_linear_fwd_planes(const void *X, const void *fwd_planes, const float *bias, const void *residual, void *Y, int dtype, long tokens,
                         int in_features, int out_features, void *stream) {
    if (!X || !fwd_planes || !Y) return SD_E_NULL;
    if (dtype != SD_F32) return SD_E_DTYPE;
    if (tokens <= 0 || in_features <= 0 || out_features <= 0) return SD_E_SHAPE;
    if (in_features % 32) return SD_E_UNSUPPORTED;
    return sd::dispatch_planes((const float *)X, fwd_planes, (out_features + 31) / 32, (float *)Y, bias, (const float *)residual, tokens, out_features,
                               in_features, static_cast<hipStream_t>(stream));
}

int sd_linear_bwd_data_planes(const void *dY, const void *bwd_planes, void *dX, int dtype, long tokens, int in_features, int out_features,
                              void *stream) {
    if (!dY || !bwd_planes || !dX) return SD_E_NULL;
    if (dtype != SD_F32) return SD_E_DTYPE;
    if (tokens <= 0 || in_features <= 0 || out_features <= 0) return SD_E_SHAPE;
    if (out_features % 32) return SD_E_UNSUPPORTED;
    // C[T x in] = dY[T x out] . B,  B(k, n) = W[k][n]: K = out_features, N = in_features
    return sd::dispatch_planes((const float *)dY, bwd_planes, (in_features + 31) / 32, (float *)dX, nullptr, nullptr, tokens, in_features, out_features,
                               static_cast<hipStream_t>(stream));
}

/* dW [out x in] = dY^T . X over FEW tokens (< 8192) or a LARGE weight (more than 16 regions of 64 x 64): the products for which
 * sd_linear_wgrad's direct tall-skinny plan does not apply and the library picks un-split 32 x 32 tiles on ~100 workgroups (63 us for
 * 1024 x 256 over 2048 tokens against ~7 us of matrix work).  Split-K over the tokens on the pipelined kernel above (A = dY read as [k][m],
 * B = X as [k][n], split-bf16 arithmetic when the shapes allow whole k-steps, exact f32 MFMA otherwise); the caller combines the slabs
 * (sd_multi_slab_reduce: deterministic, and deferrable to the end of the backward like every other parameter gradient). */
static int wgrad_splitk_plan(long tokens, int M, int N, int *klen_out) {
    if (tokens <= 0 || M <= 0 || N <= 0 || tokens > 0x7fffffffL || M % 4 || N % 4) return 0;
    const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
    // where it pays (tools/wgrad_splitk_bench.py, us library / split-K kernel, MI355X): 8192 tokens 640 x 160: 52 / 36; 2048 tokens 256 x 256:
    // 14.8 / 9.2, 512 x 256: 15.3 / 10.9, 128 x 64: 18.7 / 9.1 -- but 1024 x 256: 15.7 / 19.3, 32 x 2048: 15.3 / 19.0 (16 tiles x 22 slabs of
    // two k-steps each: all prologue, and 22 MB of slabs): few tokens AND many tiles stay with the library
    if (tokens < 8192 && tiles > 8) return 0;
    long nsplit = (384 + tiles - 1) / tiles;                    // ~1.5 workgroups per CU
    if (nsplit > tokens / 64) nsplit = tokens / 64;             // at least two k-steps per slab
    if (nsplit < 1) nsplit = 1;
    long klen = ((tokens + nsplit - 1) / nsplit + 31) / 32 * 32;
    nsplit = (tokens + klen - 1) / klen;
    if (nsplit > 512) return 0;
    *klen_out = (int)klen;
    return (int)nsplit;
}

int sd_linear_wgrad_splitk_slabs(long tokens, int out_features, int in_features) {
    int klen;
    return wgrad_splitk_plan(tokens, out_features, in_features, &klen);
}

int sd_linear_wgrad_splitk(const float *dY, const float *X, float *slabs, size_t slabs_bytes, long tokens, int out_features, int in_features,
                           void *stream) {
    if (!dY || !X || !slabs) return SD_E_NULL;
    int klen = 0;
    const int nsplit = wgrad_splitk_plan(tokens, out_features, in_features, &klen);
    if (!nsplit) return SD_E_UNSUPPORTED;
    const long slab = (long)out_features * in_features;
    if (slabs_bytes < (size_t)nsplit * slab * sizeof(float) || (reinterpret_cast<uintptr_t>(slabs) & 15)) return SD_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int M = out_features, N = in_features, K = (int)tokens;
    if (sd::g_align_split_bf16 && K % 32 == 0)
        return sd::launch_epi<128, 128, 2, 2, false, 0, false, false, true>(dY, X, slabs, nullptr, nullptr, M, N, K, M, N, N, st, 1, 0L, 0L, slab, nsplit,
                                                                            klen);
    return sd::launch_epi<128, 128, 2, 2, false, 0, false, false>(dY, X, slabs, nullptr, nullptr, M, N, K, M, N, N, st, 1, 0L, 0L, slab, nsplit, klen);
}

}  // extern "C"
