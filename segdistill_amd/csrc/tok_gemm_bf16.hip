// tok_gemm_bf16.hip -- nn.Linear on token-major activations under bf16 storage (BASELINE config 5), forward:
//     Y [tokens][out] = X [tokens][in] . W [out][in]^T + bias[out]          bf16 in and out, fp32 accumulation, gfx950.
// reference: every q / kv / proj / fc1 / fc2 Linear of the MiT encoders (mmseg/models/backbones/mix_transformer.py:24-27,48-55,75-84,107-133), the
// SR convolution as a patch GEMM (:66-70) and the MLP projections of the SegFormer head (decode_heads/segformer_head.py:22-33), as autocast runs them.
//
// Why not the library: the 41-layer B4 teacher alone issues ~200 of these per step on 8192 ... 2048 tokens with K, N in 64 ... 2048 -- a few GFLOP
// each, 2-4 us of matrix-pipe time -- and hipBLASLt's 160x64 / 160x256 / 256x256 macro-tiles finish them in 10-21 us (profiles/r05_step_shapes_cfg5.txt):
// they are latency-bound (launch + first operands + a k-loop of 5 ... 20 cold steps + epilogue), not throughput-bound.  This kernel is built for
// that regime:
//   * tile 128 tokens x 64 (or 128) output channels, 256 threads: 8192 x 320 -> 320 is 320 workgroups, all co-resident (<= 72 KB of LDS each);
//   * both operands are k-contiguous rows: a k-step is 64 k = one 128-byte line per row, brought in by LDS-DMA (global_load_lds_dwordx4: no staging
//     registers) into a ring of NS stages; everything up to NS - 1 stages ahead is requested before the first MFMA, so K = 320 costs ONE exposed
//     memory latency, not five; the waits are counted (s_waitcnt vmcnt(n)), one barrier per k-step;
//   * LDS image of a tile: 128-byte rows, 16-byte chunk c of row r at  (r >> 1) * 256 + (((r & 1) ^ (r >> 3 & 1)) << 7) + ((c ^ (r & 7)) << 4):
//     the 16 rows of one ds_read_b128 lane group (rows distinct mod 16) hit the 16 different 16-byte slots of the 256-byte bank row -- conflict-free
//     fragment reads of 128-byte rows (the plain (r & 7) XOR is two-way: rows r and r + 8 share a slot);
//   * MFMA v_mfma_f32_32x32x16_bf16 with A = W rows, B = X rows: the accumulator has the TOKEN on the lane and four consecutive output channels in
//     consecutive registers, so the epilogue packs 8 bytes per lane (+ bias in fp32, one rounding), parks the tile in LDS and stores full rows.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "sd_common.h"

namespace sd {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int kBK = 64;             // k per step: one 128-byte line per operand row

int g_force_variant = -1;           // A/B tunable (-1 = the dispatch below)

// global -> LDS, 16 bytes per lane (align_tok.hip): M0 = wave-uniform LDS byte address of lane 0's 16 bytes, lane l lands at M0 + 16 l; the source is a
// wave-uniform base (SGPR pair) + a per-lane unsigned byte offset.  From inline asm, so that the waits are the hand-counted ones below.
// hazard (gfx9): a VALU instruction that WRITES an SGPR (the v_readlane that restores a spilled base pointer, a v_readfirstlane) followed by a
// vector-memory instruction that READS it needs 5 wait states; hipcc inserts them for its own instructions, not in front of inline asm -- the load
// then goes to a stale address (round 6: a memory fault in head_tail.hip as soon as a spilled pointer was involved).  Every asm load with a
// scalar operand therefore carries its own wait states (tools/asm_sgpr_hazard_scan.py checks the built code).
__device__ __forceinline__ void dma16(const void *base, unsigned lane_off, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(base), "s"(lds_byte) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Tile BM tokens x BN channels, WM x WN waves: a wave owns BM / WM tokens (TN blocks of 32) x BN / WN channels (TM blocks of 32).
template <int BM, int BN, int WM, int WN, int NS>
struct TileCfg {
    static constexpr int NW = WM * WN, NT = 64 * NW;
    static constexpr int TN = BM / (32 * WM), TM = BN / (32 * WN);
    static constexpr int kXB = BM * 128, kWB = BN * 128, kStage = kXB + kWB;
    static constexpr int kPieces = kStage / 1024, kPPW = (kPieces + NW - 1) / NW;      // 1 KB DMA pieces per stage / per wave
    static constexpr int kPitch = BN * 2 + 16;                                         // output image rows: BN channels bf16 + 16 bytes of pad
    static constexpr int kImg = BM * kPitch;
    static constexpr int kRing = NS * kStage > kImg ? NS * kStage : kImg;
    static constexpr int kLds = kRing + BN * 4;
    static constexpr int kWgPerCu = 160 * 1024 / kLds >= 2 ? 2 : 1;
    static constexpr int kWavesPerSimd = (kWgPerCu * NW + 3) / 4;
    static_assert(BM % (32 * WM) == 0 && BN % (32 * WN) == 0, "whole 32 x 32 blocks per wave");
    static_assert(NS >= 2 && NS <= 4 && (NS - 2) * kPPW <= 63, "ring depth / vmcnt range");
    static_assert(kLds <= 160 * 1024, "LDS");
};

template <int BM, int BN, int WM, int WN, int NS>
__global__ __launch_bounds__(64 * WM * WN, (TileCfg<BM, BN, WM, WN, NS>::kWavesPerSimd)) void tok_gemm_bf16_kernel(
    const bf16_t *__restrict__ X, const bf16_t *__restrict__ W, const void *__restrict__ bias, int bias_is_bf16, bf16_t *__restrict__ Y, long T, int N,
    int K, int tiles_n) {
    typedef TileCfg<BM, BN, WM, WN, NS> C;
    constexpr int TM = C::TM, TN = C::TN, NT = C::NT, kPPW = C::kPPW, kStage = C::kStage, kXB = C::kXB, kPitch = C::kPitch;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[C::kLds];

    // XCD-aware order (bijective): consecutive workgroup ids land on different XCDs; every XCD gets a contiguous band of tiles, channel tiles of one
    // token tile next to each other (they share the X rows through that XCD's L2)
    const long nblk = gridDim.x, id = blockIdx.x;
    const long qx = nblk / 8, rem = nblk % 8, xcd = id % 8;
    const long tile = (xcd < rem ? xcd * (qx + 1) : rem * (qx + 1) + (xcd - rem) * qx) + id / 8;
    const long tm = tile / tiles_n;
    const int tn = (int)(tile - tm * tiles_n);
    const long m0 = tm * BM;
    const int n0 = tn * BN;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;

    // bias table (fp32) behind the ring; visible to everybody after the first barrier of the k-loop
    float *bias_l = reinterpret_cast<float *>(lds + C::kRing);
    for (int c = t; c < BN; c += NT) {
        float b = 0.f;
        if (bias) {
            if (bias_is_bf16) {
                const uint32_t bits = (uint32_t) reinterpret_cast<const uint16_t *>(bias)[n0 + c] << 16;
                b = __builtin_bit_cast(float, bits);
            } else {
                b = reinterpret_cast<const float *>(bias)[n0 + c];
            }
        }
        bias_l[c] = b;
    }

    // ---- DMA plan: piece q of a stage = physical bytes [1024 q, 1024 q + 1024) = 8 rows of one operand tile (the first BM / 8 pieces: X, the rest: W);
    // a wave beyond the stage's last piece repeats it (same bytes to the same place: every wave issues the same count, the waits are uniform)
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    unsigned src_off[kPPW];
    const long tlast = T - 1;
#pragma unroll
    for (int u = 0; u < kPPW; ++u) {
        const int q = min(wave * kPPW + u, C::kPieces - 1);
        const bool is_x = q < kXB / 1024;
        const int qq = is_x ? q : q - kXB / 1024;
        const int o = 16 * lane;
        const int pr2 = 4 * qq + (o >> 8), half = (o >> 7) & 1, pc = (o >> 4) & 7;
        const int row = 2 * pr2 + (half ^ ((pr2 >> 2) & 1));
        const int chunk = pc ^ (row & 7);
        const long grow = is_x ? min(m0 + row, tlast) : (long)min(n0 + row, N - 1);
        src_off[u] = (unsigned)(grow * K * 2 + chunk * 16);
    }
    auto issue = [&](int ks) {          // stage ks % NS <- k-step ks of both tiles
        const bf16_t *xk = X + (size_t)ks * kBK, *wk = W + (size_t)ks * kBK;
        const unsigned sb = lds0 + (unsigned)(ks % NS) * (unsigned)kStage;
#pragma unroll
        for (int u = 0; u < kPPW; ++u) {
            const int q = min(wave * kPPW + u, C::kPieces - 1);
            dma16(q < kXB / 1024 ? (const void *)xk : (const void *)wk, src_off[u], __builtin_amdgcn_readfirstlane(sb + 1024u * (unsigned)q));
        }
    };

    // ---- fragment addresses: lane (r, h) reads chunk 2 s + h of row base + r; (2 s + h) ^ (r & 7) = (2 s) ^ (h ^ (r & 7))
    const unsigned lane_c = (unsigned)((r >> 1) * 256 + (((r & 1) ^ ((r >> 3) & 1)) << 7) + ((h ^ (r & 7)) << 4));
    const unsigned xrd = lane_c + (unsigned)wm * (TN * 32 * 128);
    const unsigned wrd = lane_c + (unsigned)kXB + (unsigned)wn * (TM * 32 * 128);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = K / kBK;
    __builtin_amdgcn_sched_barrier(0);      // the bias loads (compiler-counted vector-memory operations) are waited for above this line
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) issue(s);

    for (int k = 0; k < nk; ++k) {
        // stage k has landed when at most the stages requested after it are in flight: min(NS - 2, nk - 1 - k) of them
        const int ahead = min(NS - 2, nk - 1 - k);
        if (NS >= 4 && ahead == 2) wait_vm<2 * kPPW>();
        else if (NS >= 3 && ahead == 1) wait_vm<kPPW>();
        else wait_vm<0>();
        __syncthreads();                    // everyone's part of stage k is there; nobody reads stage (k - 1) % NS any more
        if (k + NS - 1 < nk) issue(k + NS - 1);
        const unsigned char *st = lds + (size_t)(k % NS) * kStage;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 wf[TM], xf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) wf[i] = *reinterpret_cast<const bf16x8 *>(st + ((wrd + 4096u * i) ^ (unsigned)(s << 5)));
#pragma unroll
            for (int j = 0; j < TN; ++j) xf[j] = *reinterpret_cast<const bf16x8 *>(st + ((xrd + 4096u * j) ^ (unsigned)(s << 5)));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();                        // the ring is free (the last wait was vmcnt(0)): it becomes the output image

    // ---- epilogue: D[channel][token] -- token on the lane, channels (e & 3) + 8 (e >> 2) + 4 h in the registers
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int tok = wm * (TN * 32) + 32 * j + r;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = wn * (TM * 32) + 32 * i + 8 * q + 4 * h;
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bias_l + nl);
                const bf16x2 lo = __builtin_convertvector((f32x2){acc[i][j][4 * q] + b4[0], acc[i][j][4 * q + 1] + b4[1]}, bf16x2);
                const bf16x2 hi = __builtin_convertvector((f32x2){acc[i][j][4 * q + 2] + b4[2], acc[i][j][4 * q + 3] + b4[3]}, bf16x2);
                const u32x2 v = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
                *reinterpret_cast<u32x2 *>(lds + tok * kPitch + nl * 2) = v;
            }
        }
    __syncthreads();
    constexpr int kCPR = BN / 8;            // 16-byte chunks per output row
    for (int idx = t; idx < BM * kCPR; idx += NT) {
        const int row = idx / kCPR, c = idx - row * kCPR;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(lds + row * kPitch + c * 16);
        if (m0 + row < T) *reinterpret_cast<u32x4 *>(Y + (size_t)(m0 + row) * N + n0 + c * 8) = v;
    }
}

bool shape_ok(long T, int K, int N) {
    // whole k-steps, whole 64-channel tiles, 32-bit per-lane byte offsets into X and W
    return T > 0 && K >= kBK && K % kBK == 0 && N >= 64 && N % 64 == 0 && (double)T * K * 2 < 4.0e9 && (double)N * K * 2 < 4.0e9;
}

struct Args {
    const void *X, *W, *bias;
    int bias_is_bf16;
    void *Y;
    long T;
    int N, K;
    hipStream_t st;
};

template <int BM, int BN, int WM, int WN, int NS>
int launch(const Args &a) {
    if (a.N % BN) return SD_E_UNSUPPORTED;
    const long tiles_m = (a.T + BM - 1) / BM;
    const int tiles_n = a.N / BN;
    const long nblk = tiles_m * tiles_n;
    if (nblk > 0x7fffffffL) return SD_E_SHAPE;
    hipLaunchKernelGGL((tok_gemm_bf16_kernel<BM, BN, WM, WN, NS>), dim3((unsigned)nblk), dim3(64 * WM * WN), 0, a.st, (const bf16_t *)a.X,
                       (const bf16_t *)a.W, a.bias, a.bias_is_bf16, (bf16_t *)a.Y, a.T, a.N, a.K, tiles_n);
    return (int)hipGetLastError();
}

// The tile variants kept (tools/bf16_gemm_bench.py --variants).  Measured and dropped (profiles/r05_bf16_gemm_variants.txt): 256 x 160, 128 x 160,
// 64 x 160, 256 x 128 and 256 x 64 tiles -- fewer, larger workgroups read fewer bytes in total but finish in ONE round, so every workgroup's output
// stores drain together behind the last MFMA instead of under the next round's k-loop (8192 x 320 -> 1280: 13.8 - 18.4 us against 14.5; -> 320: 9.1 -
// 13.3 against 7.3); a third ring stage buys nothing (the shallow ring's extra co-resident workgroup does more for latency).
constexpr int kVariants = 5;
int launch_variant(int v, const Args &a) {
    switch (v) {
        case 0: return launch<128, 64, 2, 2, 2>(a);
        case 1: return launch<128, 128, 2, 2, 2>(a);
        case 2: return launch<128, 64, 2, 2, 3>(a);
        case 3: return launch<64, 64, 2, 1, 2>(a);
        case 4: return launch<64, 128, 2, 2, 2>(a);
        default: return SD_E_UNSUPPORTED;
    }
}

}  // namespace

int tok_gemm_bf16_tunable(const char *key, int set, int v) {
    if (strcmp(key, "tok_gemm_bf16_variant")) return SD_E_UNSUPPORTED;
    if (!set) return g_force_variant;
    if (v < -1 || v >= kVariants) return SD_E_SHAPE;
    g_force_variant = v;
    return SD_OK;
}

}  // namespace sd

extern "C" {

int sd_linear_bf16_fwd_supported(long tokens, int in_features, int out_features) { return sd::shape_ok(tokens, in_features, out_features) ? 1 : 0; }

int sd_linear_bf16_fwd(const void *X, const void *W, const void *bias, int bias_dtype, void *Y, long tokens, int in_features, int out_features,
                       void *stream) {
    using namespace sd;
    if (!X || !W || !Y) return SD_E_NULL;
    if (!shape_ok(tokens, in_features, out_features)) return SD_E_UNSUPPORTED;
    if (bias && bias_dtype != SD_F32 && bias_dtype != SD_BF16) return SD_E_DTYPE;
    if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(Y)) & 15) return SD_E_ALIGN;
    const Args a = {X, W, bias, bias_dtype == SD_BF16 ? 1 : 0, Y, tokens, out_features, in_features, reinterpret_cast<hipStream_t>(stream)};
    if (g_force_variant >= 0) {
        const int rc = launch_variant(g_force_variant, a);
        if (rc != SD_E_UNSUPPORTED) return rc;          // a forced variant that does not divide N: the dispatch below
    }
    // Measured (tools/bf16_gemm_bench.py, profiles/r05_bf16_gemm_bench.txt, r05_bf16_gemm_variants.txt): these launches cost a fixed ~4 us (launch, first
    // operands, epilogue) + ~0.5 us per 64-deep k-step -- the rate at which a CU fills its LDS from L2 (~70 GB/s) over the one or two workgroups it
    // holds -- so the tile shape moves them by a few per cent only: 128-channel tiles (X re-read half as often) from 32768 tokens on, 64-channel tiles
    // (more workgroups in flight) below.
    const int N = out_features;
    return launch_variant((N % 128 == 0 && tokens >= 32768) ? 1 : 0, a);
}

}  // extern "C"
