// wgrad_tn.hip -- the weight gradient of a token-major Linear under bf16 storage, dW [M x N] = dY [T x M]^T . X [T x N], as split-K fp32 slabs
// on the bf16 matrix pipe with HARDWARE-TRANSPOSED operand reads, gfx950.
//
// reference: autograd's `grad_output.t().mm(input)` of every nn.Linear of the MiT encoders / SegFormer heads (mix_transformer.py:24-27,48-55,
// 75-84,107-133; segformer_head.py:22-33) and of the token-major feature-align projection of BASELINE config 5 (opts.py:25-27).
// Both operands are stored token-major, i.e. the REDUCTION index (the token) is the slow one: an MFMA fragment wants 8 consecutive k of one
// row m / column n, which in memory are 8 elements M (N) apart.  Round 1's gemm_mfma_bf16<.., K-major, K-major> transposed them with 2-byte
// LDS stores and ran config 5's align gradient (768 x 256 over 131072 tokens) in 239 us against 42 us of HBM time (profiles of round 4:
// gpurun_out/configs/step_shapes_cfg5.txt) -- the top row of that step's GEMM time.  Here:
//   * tiles [32 tokens][128 m] and [32 tokens][128 n] are copied into LDS as they are (16-byte global loads, 16-byte LDS stores, 256-byte rows
//     with the XOR swizzle of cdna_hip_programming.md T10 (b)), double-buffered, the next tile's loads in flight during the current MFMAs;
//   * every operand fragment is two `ds_read_b64_tr_b16` (a 4-row x 16-column block delivered column-major: the transpose is free);
//   * 128 x 128 output tile, four waves 2 x 2, `v_mfma_f32_32x32x16_bf16`, fp32 accumulation, one fp32 slab per k-split (deterministic
//     combine by sd_multi_slab_reduce); workgroups of one k-split are placed on ONE XCD (consecutive logical ids share an XCD), so the tiles
//     that share a token range share its rows through that XCD's L2 instead of re-reading them from HBM.
// HBM-bound: dY and X once (2 T (M + N) bytes) + nsplit slabs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "sd_common.h"

namespace sd {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TBM = 128, TBN = 128, TBK = 32;
constexpr int kStageBytes = TBK * 256;                 // one operand tile: 32 rows of 128 bf16

// byte offset of 16-byte chunk `ch` (0..15) of row `row` in a tile of 256-byte rows (T10 image (b): conflict-free row stores AND transposed reads)
__device__ __forceinline__ int swz(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// the 8 k-values (rows r0 .. r0+7) of column block c0 (16 columns = chunks c0, c0 + 1) as one MFMA fragment: two transposed reads
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char *tile, int r0, int c0, int li) {
    const int q = li >> 2, p = li & 3;
    typedef s16x4 __attribute__((address_space(3))) * lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tile + swz(r0 + q, c0 + (p >> 1)) + 8 * (p & 1)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tile + swz(r0 + 4 + q, c0 + (p >> 1)) + 8 * (p & 1)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// C_z[M x N] = sum over tokens k of split z of A[k][m] B[k][n].  grid.x = tiles_m * tiles_n * nsplit (XCD-remapped); klen % TBK == 0;
// M % 8 == 0 and N % 8 == 0 (rows are whole 16-byte chunks), A and B 16-byte aligned.
__global__ __launch_bounds__(256) void wgrad_tn_bf16(const bf16_t *__restrict__ A, const bf16_t *__restrict__ B, float *__restrict__ C, int M, int N,
                                                      long T, int klen, int tiles_m, int tiles_n, int nsplit) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * kStageBytes];     // [stage][A | B]
    // consecutive LOGICAL ids on one XCD (bijective remap, cdna_hip_programming.md section 5): the tiles of a k-split then share an L2
    const long nblk = gridDim.x, id = blockIdx.x;
    const long qd = nblk / 8, rem = nblk % 8, xcd = id % 8;
    const long L = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + id / 8;
    const int tiles = tiles_m * tiles_n;
    const int split = (int)(L / tiles), tile = (int)(L - (long)split * tiles);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * TBM, n0 = tn * TBN;
    const long k_begin = (long)split * klen, k_end = min(T, k_begin + klen);
    const int nk = (int)((k_end - k_begin + TBK - 1) / TBK);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int h = lane >> 5, g = (lane >> 4) & 1, li = lane & 15;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // staging: thread -> (row = t / 16 (+16), chunk = t % 16) of each operand tile
    const int srow = t >> 4, sch = t & 15;
    const bool a_in = m0 + 8 * sch < M, b_in = n0 + 8 * sch < N;          // whole chunks: M, N % 8 == 0
    const bf16_t *pa = A + (size_t)(m0 + (a_in ? 8 * sch : 0));
    const bf16_t *pb = B + (size_t)(n0 + (b_in ? 8 * sch : 0));
    u32x4 ra[2], rb[2];
    auto load_regs = [&](int kt) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long k = k_begin + (long)kt * TBK + srow + 16 * u;
            const bool kin = k < k_end;
            const long kc = kin ? k : k_end - 1;                             // clamped address, zeroed value
            const u32x4 va = *reinterpret_cast<const u32x4 *>(pa + (size_t)kc * M);
            const u32x4 vb = *reinterpret_cast<const u32x4 *>(pb + (size_t)kc * N);
            const u32x4 z = {0u, 0u, 0u, 0u};
            ra[u] = (kin && a_in) ? va : z;
            rb[u] = (kin && b_in) ? vb : z;
        }
    };
    auto store_regs = [&](int stage) {
        unsigned char *sa = lds + (size_t)stage * 2 * kStageBytes, *sb = sa + kStageBytes;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            *reinterpret_cast<u32x4 *>(sa + swz(srow + 16 * u, sch)) = ra[u];
            *reinterpret_cast<u32x4 *>(sb + swz(srow + 16 * u, sch)) = rb[u];
        }
    };
    if (nk > 0) {
        load_regs(0);
        store_regs(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_regs(kt + 1);                                  // in flight during the MFMAs below
        const unsigned char *sa = lds + (size_t)(kt & 1) * 2 * kStageBytes, *sb = sa + kStageBytes;
#pragma unroll
        for (int s = 0; s < TBK / 16; ++s) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = tr_frag(sa, 16 * s + 8 * h, (wm + 32 * i) / 8 + 2 * g, li);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = tr_frag(sb, 16 * s + 8 * h, (wn + 32 * j) / 8 + 2 * g, li);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_regs((kt + 1) & 1);                           // the other stage: last read before the previous barrier
        __syncthreads();
    }
    // slab z = split: C layout of the 32x32 MFMA -- col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float *Cz = C + (size_t)split * M * N;
    const int col = lane & 31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + 32 * j + col;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < M && n < N) Cz[(size_t)m * N + n] = acc[i][j][e];
            }
        }
}

// ---- the same product with the tiles brought in by LDS-DMA into a ring of four stages ----------------------------------------------------
// The register-staged kernel above keeps ONE tile (16 KB) in flight per workgroup and runs every k-step at the memory latency: PMC says its
// HBM traffic is ideal (FETCH_SIZE x 2 = dY + X once: the tiles of a k-split do share their rows through the XCD's L2), yet 768 x 256 over
// 131072 tokens took 132 us = 24 GB/s per CU.  Here `global_load_lds_dwordx4` copies global -> LDS without staging registers (the XOR swizzle
// is applied to the per-lane SOURCE address: the LDS image of one wave-instruction is lane-linear), THREE tiles are in flight behind the one
// being multiplied, one raw barrier per k-step.  The DMA is issued from inline asm: hipcc puts `s_waitcnt vmcnt(0)` in front of every LDS
// read that follows a `__builtin_amdgcn_global_load_lds` it can see, which would drain the ring; the counts are kept by hand -- every
// iteration issues exactly kDmaPerTile instructions per wave (past the last tile: a harmless re-load of it into a free stage), so the wait
// that retires tile kt is always vmcnt(2 * kDmaPerTile).  Whole tiles only: T % 32 == 0 (the launcher checks).
constexpr int kRing = 4;                               // stages; per wave and tile: rows 8w .. 8w+7 of every 128-column image, 4 rows (1 KB) per DMA instruction

__device__ __forceinline__ void dma16(const void *gptr, unsigned lds_byte) {
    // M0 = wave-uniform LDS byte address of lane 0's 16 bytes; lane l lands at M0 + 16 l
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_byte) : "memory");   // M0 is reserved in LLVM (a clobber entry is rejected with a warning); nothing else in this kernel uses it (checked in the ISA)
}

// IM = images of 128 columns per operand tile: 1 -> 128 x 128 output tiles (64 x 64 per wave), 2 -> 256 x 256 (128 x 128 per wave, 256 accumulator
// registers, one workgroup per CU).  PMC of the IM = 1 form at 768 x 256 over 131072 tokens (profiles/r04_pmc_wgrad_tn.json): no LDS bank conflict,
// LDS array ~12 % busy, matrix pipe 25 % busy, FETCH_SIZE = dY + X once -- the kernel runs at what ~96 KB in flight per CU buy from L2 at this
// latency (804 MB of L2 -> CU traffic in 93 us): the big tile halves that traffic per MFMA (402 MB) instead of deepening the ring.
// nblk / id: the number of workgroups of THIS product and this workgroup's index among them (the whole grid for a single launch; a slice of it in
// the grouped launch below, whose slices start at multiples of 8 so that id % 8 is still the XCD the workgroup runs on)
template <int IM>
__device__ __forceinline__ void wgrad_ring_body(const bf16_t *__restrict__ A, const bf16_t *__restrict__ B, float *__restrict__ C, int M, int N, long T,
                                                int klen, int tiles_m, int tiles_n, int nsplit, long nblk, long id, const bool BIAS = false) {
    // BIAS (IM = 1 only): the slab is [M x N] followed by M column sums of A = dY over the split's tokens -- the Linear's bias gradient, taken from the
    // tiles the tn == 0 workgroups stage anyway (the batched column-sum pass re-read every dY: 0.11 ms per config-5 step)
    constexpr int BT = 128 * IM;                       // tile extent along m and along n
    constexpr int TW = 2 * IM;                         // 32-row (-column) MFMA blocks per wave and axis
    constexpr int kStage = 2 * IM * kStageBytes;       // A images, then B images
    constexpr int kDma = 4 * IM;                       // DMA instructions per wave and tile
    __shared__ __attribute__((aligned(1024))) unsigned char lds[kRing * kStage];      // 64 KB (IM = 1) / 128 KB (IM = 2)
    const long qd = nblk / 8, rem = nblk % 8, xcd = id % 8;
    const long L = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + id / 8;
    const int tiles = tiles_m * tiles_n;
    const int split = (int)(L / tiles), tile = (int)(L - (long)split * tiles);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BT, n0 = tn * BT;
    const long k_begin = (long)split * klen, k_end = min(T, k_begin + klen);
    const int nk = (int)((k_end - k_begin) / TBK);
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = (wave >> 1) * (BT / 2), wn = (wave & 1) * (BT / 2);
    const int h = lane >> 5, g = (lane >> 4) & 1, li = lane & 15;

    f32x16 acc[TW][TW];
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // DMA geometry: per 128-column image, instruction u (0, 1) of wave w moves tile rows 8w + 4u .. + 3; lane -> (row r = 8w + 4u + lane / 16, LDS
    // chunk c' = lane % 16), which must hold source chunk c' ^ f(r) (swz is an involution in the chunk index).  Chunks beyond M / N: clamped to a
    // valid chunk -- they only feed output columns that are never stored.
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    const bf16_t *ga[IM][2], *gb[IM][2];
    unsigned la[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int r = 8 * wave + 4 * u + (lane >> 4);
        const int ch = (lane & 15) ^ (((r & 3) << 2) | ((r >> 2) & 3));
#pragma unroll
        for (int im = 0; im < IM; ++im) {
            const int ma = m0 + 128 * im + 8 * ch, nb = n0 + 128 * im + 8 * ch;
            ga[im][u] = A + (size_t)(k_begin + r) * M + (ma < M ? ma : 0);
            gb[im][u] = B + (size_t)(k_begin + r) * N + (nb < N ? nb : 0);
        }
        la[u] = 256u * (unsigned)(8 * wave + 4 * u);
    }
    auto issue = [&](int kt) {                          // tile kt (clamped) -> stage kt % kRing; kDma instructions
        const int kc = kt < nk ? kt : nk - 1;
        const unsigned st = lds0 + (unsigned)(kt % kRing) * (unsigned)kStage;
#pragma unroll
        for (int im = 0; im < IM; ++im)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                dma16(ga[im][u] + (size_t)kc * TBK * M, __builtin_amdgcn_readfirstlane(st + (unsigned)im * kStageBytes + la[u]));
                dma16(gb[im][u] + (size_t)kc * TBK * N, __builtin_amdgcn_readfirstlane(st + (unsigned)(IM + im) * kStageBytes + la[u]));
            }
    };
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (nk > 0) {
        issue(0);
        issue(1);
        issue(2);
    }
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kDma) : "memory");            // all but the two youngest tiles: tile kt has landed (this wave's part)
        __builtin_amdgcn_s_barrier();                                                // ... everyone's part; and stage (kt + 3) % 4 is no longer read
        __builtin_amdgcn_sched_barrier(0);
        issue(kt + 3);
        const unsigned char *sa = lds + (size_t)(kt % kRing) * kStage, *sb = sa + IM * kStageBytes;
        if (IM == 1 && BIAS && tn == 0) {
            // thread t: chunk t % 16 (8 columns) of rows 2 (t / 16), + 1 of the 32-row tile
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(sa + swz(2 * (t >> 4) + u, t & 15));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bsum[2 * e] += __uint_as_float(v[e] << 16);
                    bsum[2 * e + 1] += __uint_as_float(v[e] & 0xffff0000u);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < TBK / 16; ++s) {
            bf16x8 fa[TW], fb[TW];
#pragma unroll
            for (int i = 0; i < TW; ++i) {
                const int cm = wm + 32 * i;                                          // column of the A tile = row block of the output
                fa[i] = tr_frag(sa + (cm >> 7) * kStageBytes, 16 * s + 8 * h, (cm & 127) / 8 + 2 * g, li);
            }
#pragma unroll
            for (int j = 0; j < TW; ++j) {
                const int cn = wn + 32 * j;
                fb[j] = tr_frag(sb + (cn >> 7) * kStageBytes, 16 * s + 8 * h, (cn & 127) / 8 + 2 * g, li);
            }
#pragma unroll
            for (int i = 0; i < TW; ++i)
#pragma unroll
                for (int j = 0; j < TW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the tail's placeholder DMAs must not land after the workgroup's LDS is gone
    float *Cz = C + (size_t)split * ((size_t)M * N + (BIAS ? M : 0));
    if (IM == 1 && BIAS && tn == 0) {
        __syncthreads();                                 // every wave is done with the ring: 8 KB of it become the fold buffer [16 row pairs][128 columns]
        float *red = reinterpret_cast<float *>(lds);
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(t >> 4) * TBM + 8 * (t & 15) + e] = bsum[e];
        __syncthreads();
        if (t < TBM) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) v += red[q * TBM + t];
            if (m0 + t < M) Cz[(size_t)M * N + m0 + t] = v;
        }
    }
    const int col = lane & 31;
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            const int n = n0 + wn + 32 * j + col;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < M && n < N) Cz[(size_t)m * N + n] = acc[i][j][e];
            }
        }
}

template <int IM>
__global__ __launch_bounds__(256) void wgrad_tn_bf16_ring(const bf16_t *__restrict__ A, const bf16_t *__restrict__ B, float *__restrict__ C, int M,
                                                           int N, long T, int klen, int tiles_m, int tiles_n, int nsplit) {
    wgrad_ring_body<IM>(A, B, C, M, N, T, klen, tiles_m, tiles_n, nsplit, (long)gridDim.x, (long)blockIdx.x);
}

// ---- many products in ONE launch (round 5) -------------------------------------------------------------------------------------------------
// The weight gradients of a backward are needed by nobody before the optimizer; launched one by one between the input-gradient GEMMs they sat in
// the critical chain (54 launches of 9 ... 74 us per config-5 step, the stage 3-4 ones on 16 ... 100 workgroups) and each planned its k-splits as
// if it had the chip to itself (~768 workgroups per product: 832 MB of slabs next to 1.48 GB of operands per step).  Deferred to the end of the
// backward (segdistill_amd/deferred.py) they run as one launch whose k-splits are planned over ALL of them: ~1536 workgroups in total, dealt in
// proportion to each product's tile-k-steps -- a 2048-token product gets ONE split (its gradient is written straight to its destination, no slab at
// all), a 131072-token one ~30.  The table travels by value in the kernel arguments (safe under graph capture).
constexpr int kMultiMax = 32;
struct WgradMultiTable {
    const void *A[kMultiMax], *B[kMultiMax];
    float *C[kMultiMax];
    long T[kMultiMax];
    int M[kMultiMax], N[kMultiMax], klen[kMultiMax], tiles_m[kMultiMax], tiles_n[kMultiMax], nsplit[kMultiMax];
    int blk_begin[kMultiMax + 1];       // multiples of 8
    int nblk[kMultiMax];                // real workgroups of the job (the slice is padded up to a multiple of 8)
    int bias[kMultiMax];                // fp32 form: the slab is [M x N] followed by M column sums of dY
    int njobs;
};

__global__ __launch_bounds__(256) void wgrad_tn_bf16_ring_multi(const WgradMultiTable t) {
    int lo = 0, hi = t.njobs - 1;
    while (lo < hi) {                   // wave-uniform binary search
        const int mid = (lo + hi + 1) >> 1;
        if ((int)blockIdx.x >= t.blk_begin[mid]) lo = mid;
        else hi = mid - 1;
    }
    const int j = lo;
    const long id = (long)blockIdx.x - t.blk_begin[j];
    if (id >= t.nblk[j]) return;        // padding of the slice
    wgrad_ring_body<1>((const bf16_t *)t.A[j], (const bf16_t *)t.B[j], t.C[j], t.M[j], t.N[j], t.T[j], t.klen[j], t.tiles_m[j], t.tiles_n[j], t.nsplit[j], (long)t.nblk[j], id,
                       t.bias[j] != 0);
}

// ---- fp32 storage: the same product in split-bf16 arithmetic ("bf16x3", fp32-grade: token_gemm.hip section 3.9 of DESIGN.md) ---------------
// dY and X arrive as fp32.  A staging thread owns 8 consecutive floats of a tile row (two 16-byte loads), splits them EXACTLY into three bf16
// planes (hi = rn(x), mid = rn(x - hi), lo = rn(x - hi - mid): the bits token_gemm.hip derives) and stores one 16-byte chunk per plane into
// three LDS images of the bf16 layout above; fragments are transposed reads of those planes and every (i, j, k16) keeps the six products of
// weight >= 2^-16, small terms first, fp32 accumulation.  Round 3's tall-skinny kernel (linear_wgrad_direct, exact f32 MFMA: 64 matrix-pipe
// cycles per k where this needs 12 per plane pair x 6 = 24) ran the SegFormer head's 256 x 256 / 160 / 64 / 32 weight gradients over 131072
// tokens at 225 / ~150 / ~60 / 42 us -- 0.48 of its 0.675 ms per config-2 step.  One LDS stage (48 KB: three workgroups per CU), the next tile's
// loads in flight in registers during the MFMAs.  BN_ = 128 / 64 / 32 columns of X per tile (the head's in_features are 256, 160, 64, 32).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split8_planes(const f32x4 &v0, const f32x4 &v1, u32x4 &h, u32x4 &m, u32x4 &l) {
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const f32x2 v = {x[e], x[e + 1]};
        const bf16x2 hh = __builtin_convertvector(v, bf16x2);
        const f32x2 r1 = v - __builtin_convertvector(hh, f32x2);
        const bf16x2 mm = __builtin_convertvector(r1, bf16x2);
        const f32x2 r2 = r1 - __builtin_convertvector(mm, f32x2);
        const bf16x2 ll = __builtin_convertvector(r2, bf16x2);
        h[e >> 1] = __builtin_bit_cast(unsigned, hh);
        m[e >> 1] = __builtin_bit_cast(unsigned, mm);
        l[e >> 1] = __builtin_bit_cast(unsigned, ll);
    }
}

// BIAS: the slab of split z is [M x N] followed by M column sums of dY over the split's tokens (the Linear's bias gradient: every dY value passes
// through the staging registers of exactly one thread of the tn == 0 workgroup of its row tile, in fp32 -- the separate pass that re-read every dY for
// the bias gradients was 98 us per config-2 step at full HBM speed).
// (BIAS is a compile-time constant in the single launches and a per-job flag in the grouped one; nblk / id as in wgrad_ring_body)
template <int BN_, int WM, int WN>
__device__ __forceinline__ void wgrad_x3_body(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int M, int N, long T,
                                              int klen, int tiles_m, int tiles_n, int nsplit, const bool BIAS, long nblk, long id) {
    constexpr int TM = TBM / (32 * WM), TN = BN_ / (32 * WN);
    constexpr int BCH = BN_ / 8;                                     // 8-float chunks per B row
    constexpr int NB = (32 * BCH + 255) / 256;                       // B chunks per thread (1 for BN_ <= 64, 2 for 128)
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "four waves");
    __shared__ __attribute__((aligned(16))) unsigned char lds[6 * kStageBytes];        // A planes h, m, l | B planes h, m, l
    const long qd = nblk / 8, rem = nblk % 8, xcd = id % 8;
    const long L = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + id / 8;
    const int tiles = tiles_m * tiles_n;
    const int split = (int)(L / tiles), tile = (int)(L - (long)split * tiles);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * TBM, n0 = tn * BN_;
    const long k_begin = (long)split * klen, k_end = min(T, k_begin + klen);
    const int nk = (int)((k_end - k_begin + TBK - 1) / TBK);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = (wave / WN) * (32 * TM), wn = (wave % WN) * (32 * TN);
    const int h = lane >> 5, g = (lane >> 4) & 1, li = lane & 15;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // staging: A chunk e = t + 256 u (u = 0, 1) -> (row = e / 16, ch = e % 16); B chunk e = t + 256 u (u < NB) -> (row = e / BCH, ch = e % BCH)
    f32x4 ra[2][2], rb[NB][2];
    f32x4 bsum[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};     // BIAS: this thread's share of 8 column sums
    auto load_regs = [&](int kt) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = t + 256 * u, row = e >> 4, ch = e & 15;
            const long k = k_begin + (long)kt * TBK + row;
            const bool in = k < k_end && m0 + 8 * ch < M;
            const float *p = A + (size_t)(in ? k : k_begin) * M + (in ? m0 + 8 * ch : 0);
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(p), v1 = *reinterpret_cast<const f32x4 *>(p + 4);
            ra[u][0] = in ? v0 : z;
            ra[u][1] = in ? v1 : z;
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int e = t + 256 * u, row = e / BCH, ch = e % BCH;
            const long k = k_begin + (long)kt * TBK + row;
            const bool in = row < TBK && k < k_end && n0 + 8 * ch < N;
            const float *p = B + (size_t)(in ? k : k_begin) * N + (in ? n0 + 8 * ch : 0);
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(p), v1 = *reinterpret_cast<const f32x4 *>(p + 4);
            rb[u][0] = in ? v0 : z;
            rb[u][1] = in ? v1 : z;
        }
    };
    auto store_regs = [&]() {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = t + 256 * u, row = e >> 4, ch = e & 15;
            if (BIAS) {
                bsum[u][0] += ra[u][0];
                bsum[u][1] += ra[u][1];
            }
            u32x4 ph, pm, pl;
            split8_planes(ra[u][0], ra[u][1], ph, pm, pl);
            unsigned char *d = lds + swz(row, ch);
            *reinterpret_cast<u32x4 *>(d) = ph;
            *reinterpret_cast<u32x4 *>(d + kStageBytes) = pm;
            *reinterpret_cast<u32x4 *>(d + 2 * kStageBytes) = pl;
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int e = t + 256 * u, row = e / BCH, ch = e % BCH;
            if (row < TBK) {
                u32x4 ph, pm, pl;
                split8_planes(rb[u][0], rb[u][1], ph, pm, pl);
                unsigned char *d = lds + 3 * kStageBytes + swz(row, ch);
                *reinterpret_cast<u32x4 *>(d) = ph;
                *reinterpret_cast<u32x4 *>(d + kStageBytes) = pm;
                *reinterpret_cast<u32x4 *>(d + 2 * kStageBytes) = pl;
            }
        }
    };
    if (nk > 0) load_regs(0);
    for (int kt = 0; kt < nk; ++kt) {
        store_regs();                                                        // tile kt: split + 16-byte LDS stores
        __syncthreads();
        if (kt + 1 < nk) load_regs(kt + 1);                                  // in flight during the MFMAs below
        const unsigned char *sa = lds, *sb = lds + 3 * kStageBytes;
#pragma unroll
        for (int s = 0; s < TBK / 16; ++s) {
            bf16x8 fa[TM][3], fb[TN][3];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) fa[i][pl] = tr_frag(sa + pl * kStageBytes, 16 * s + 8 * h, (wm + 32 * i) / 8 + 2 * g, li);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) fb[j][pl] = tr_frag(sb + pl * kStageBytes, 16 * s + 8 * h, (wn + 32 * j) / 8 + 2 * g, li);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x16 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], c, 0, 0, 0);      // small terms first (token_gemm.hip's order)
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        }
        __syncthreads();                                                     // every wave is done reading before the next tile is stored
    }
    float *Cz = C + (size_t)split * ((size_t)M * N + (BIAS ? M : 0));
    const int col = lane & 31;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn + 32 * j + col;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < M && n < N) Cz[(size_t)m * N + n] = acc[i][j][e];
            }
        }
    if (BIAS && tn == 0) {
        // thread t staged chunk t % 16 (8 columns) of rows t / 16 and t / 16 + 16 of every tile: 16 threads share a chunk; fixed-order fold through LDS
        // (the staging buffers are free: the main loop ended on a barrier)
        float *red = reinterpret_cast<float *>(lds);                 // [16 row groups][128 columns]
        const f32x4 s0 = bsum[0][0] + bsum[1][0], s1 = bsum[0][1] + bsum[1][1];
        float *r = red + (t >> 4) * TBM + 8 * (t & 15);
        *reinterpret_cast<f32x4 *>(r) = s0;
        *reinterpret_cast<f32x4 *>(r + 4) = s1;
        __syncthreads();
        if (t < TBM) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) v += red[q * TBM + t];
            if (m0 + t < M) Cz[(size_t)M * N + m0 + t] = v;
        }
    }
}

template <int BN_, int WM, int WN, bool BIAS>
__global__ __launch_bounds__(256) void wgrad_tn_x3(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int M, int N, long T,
                                                    int klen, int tiles_m, int tiles_n, int nsplit) {
    wgrad_x3_body<BN_, WM, WN>(A, B, C, M, N, T, klen, tiles_m, tiles_n, nsplit, BIAS, (long)gridDim.x, (long)blockIdx.x);
}

// the fp32 weight gradients of a backward in one launch per tile width (the grouped form of wgrad_tn_bf16_ring_multi above; bias: per job)
template <int BN_, int WM, int WN>
__global__ __launch_bounds__(256) void wgrad_tn_x3_multi(const WgradMultiTable t) {
    int lo = 0, hi = t.njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int)blockIdx.x >= t.blk_begin[mid]) lo = mid;
        else hi = mid - 1;
    }
    const int j = lo;
    const long id = (long)blockIdx.x - t.blk_begin[j];
    if (id >= t.nblk[j]) return;
    wgrad_x3_body<BN_, WM, WN>((const float *)t.A[j], (const float *)t.B[j], t.C[j], t.M[j], t.N[j], t.T[j], t.klen[j], t.tiles_m[j], t.tiles_n[j],
                               t.nsplit[j], t.bias[j] != 0, (long)t.nblk[j], id);
}

}  // namespace

// ---- plan + launcher shared with align1x1.hip (the generic weight-gradient entry points) -------------------------------------------------
// Split-K weight gradients write one M x N fp32 slab per split and the combine reads it back; the most splits a bf16-storage product may take so
// that the slab bytes stay within `wgrad_slab_ratio` % (tunable, 0 = no cap) of its operand bytes 2 T (M + N).  Config 5, same box
// (profiles/r04_ab_cfg5_slab_ratio.txt): no cap 693.8 imgs/s, 200 % 705.3, 100 % 712.5 / 705, 50 % 719.2 / 712.4, 35 % 718.2, 25 % 714.1, 12 % 692.0.
// fp32 storage (es == 4) is NOT capped: config 2, same box, no cap 781.6 / 784.5 imgs/s, 100 % 783.1, 60 % 780.7, 40 % 771.9, 20 % 744.6 -- its
// split-bf16 / f32-MFMA weight-gradient kernels need the parallelism more than they pay for the slabs.
int g_slab_ratio = 40;
int g_multi_wgs = 0;        // tunable "wgrad_multi_wgs" (A/B): workgroups per grouped launch, 0 = kMultiTargetWgs / kMultiTargetWgsF32 below
long wgrad_slab_cap(long T, int M, int N, int es) {
    if (es != 2 || g_slab_ratio <= 0) return 1L << 40;
    const long cap = (long)((double)g_slab_ratio * 0.01 * (double)T * (M + N) * 2.0 / ((double)M * N * 4.0));
    return cap < 1 ? 1 : cap;
}
int g_wgrad_ring = 1;      // tunable "wgrad_tn_ring": 0 = the register-staged kernel everywhere, 1 = the rules below, 2 = 256 x 256 tiles wherever legal (A/B, tests)
bool wgrad_tn_supported(long T, int M, int N, const void *dY, const void *X) {
    return T > 0 && M > 0 && N > 0 && M % 8 == 0 && N % 8 == 0 && ((reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(X)) & 15) == 0;
}

// big (256 x 256) tiles of the ring kernel: both extents at least 256, whole 32-token tiles,
// and at least 32 k-steps per workgroup at one workgroup per CU -- measured (profiles/r04_kernels_wgrad_bf16.txt): 96 -> 86 us at 768 x 256 over
// 131072 tokens (49 k-steps), but 28 -> 29 us over 32768 tokens (13 k-steps): below that the unoverlapped prologue and 256 KB slab store of a
// lone workgroup per CU cost more than the halved L2 -> CU traffic saves.
static bool wgrad_tn_big(long T, int M, int N) {
    if (!g_wgrad_ring || M < 256 || N < 256 || T % TBK != 0) return false;
    const long tiles = (long)((M + 255) / 256) * ((N + 255) / 256);
    const long ns = 256 / tiles < 1 ? 1 : 256 / tiles;
    return g_wgrad_ring == 2 ? T >= 8 * TBK : T / ns >= 32 * TBK;
}

// number of k-splits (= slabs): ~3 workgroups per CU (one per CU for the big tiles), every split at least two k-steps, at most 256 slabs
void wgrad_tn_plan(long T, int M, int N, int *nsplit, int *klen) {
    const bool big = wgrad_tn_big(T, M, N);
    const int bt = big ? 256 : TBM;
    const long tiles = (long)((M + bt - 1) / bt) * ((N + bt - 1) / bt);
    long ns = (big ? 256 : 768) / tiles;
    if (ns > T / ((big ? 4 : 2) * TBK)) ns = T / ((big ? 4 : 2) * TBK);
    if (ns > 256) ns = 256;
    // Every split writes an M x N fp32 slab and the combine reads it back: for the few-token / large-weight Linears (2048 tokens x 1024 x 512:
    // 22 splits of 3 k-steps each) that was 46 MB of slabs each way next to 6 MB of operands -- 2.9 GB of slab traffic per config-5 step against
    // 1.4 GB of operands.  wgrad_slab_cap: no more slab bytes than a set percentage of the operand bytes.
    const long cap = wgrad_slab_cap(T, M, N, 2);
    if (ns > cap) ns = cap;
    if (ns < 1) ns = 1;
    const long kl = ((T + ns - 1) / ns + TBK - 1) / TBK * TBK;
    *klen = (int)kl;
    *nsplit = (int)((T + kl - 1) / kl);
}

int wgrad_tn_tunable(const char *key, int set, int v) {
    if (!strcmp(key, "wgrad_slab_ratio")) {
        if (!set) return g_slab_ratio;
        if (v < 0 || v > 100000) return SD_E_SHAPE;
        g_slab_ratio = v;
        return SD_OK;
    }
    if (!strcmp(key, "wgrad_multi_wgs")) {
        if (!set) return g_multi_wgs;
        if (v < 0 || v > 65536) return SD_E_SHAPE;
        g_multi_wgs = v;
        return SD_OK;
    }
    if (strcmp(key, "wgrad_tn_ring")) return SD_E_UNSUPPORTED;
    if (!set) return g_wgrad_ring;
    if (v < 0 || v > 2) return SD_E_SHAPE;
    g_wgrad_ring = v;
    return SD_OK;
}

int wgrad_tn_launch(const void *dY, const void *X, float *slabs, long T, int M, int N, int nsplit, int klen, hipStream_t st) {
    const bool big = wgrad_tn_big(T, M, N) && klen >= 3 * TBK;
    const int bt = big ? 256 : TBM;
    const int tiles_m = (M + bt - 1) / bt, tiles_n = (N + bt - 1) / bt;
    const long nblk = (long)tiles_m * tiles_n * nsplit;
    if (nblk > 0x7fffffffL) return SD_E_SHAPE;
    const bf16_t *a = (const bf16_t *)dY, *b = (const bf16_t *)X;
    if (big) {
        hipLaunchKernelGGL(wgrad_tn_bf16_ring<2>, dim3((unsigned)nblk), dim3(256), 0, st, a, b, slabs, M, N, T, klen, tiles_m, tiles_n, nsplit);
    } else if (T % TBK == 0 && klen >= 3 * TBK && g_wgrad_ring) {
        hipLaunchKernelGGL(wgrad_tn_bf16_ring<1>, dim3((unsigned)nblk), dim3(256), 0, st, a, b, slabs, M, N, T, klen, tiles_m, tiles_n, nsplit);
    } else {
        hipLaunchKernelGGL(wgrad_tn_bf16, dim3((unsigned)nblk), dim3(256), 0, st, a, b, slabs, M, N, T, klen, tiles_m, tiles_n, nsplit);
    }
    return (int)hipGetLastError();
}

// ---- the grouped launch: eligibility, the joint plan, the launch (bf16: the ring kernel, 128 x 128 tiles; fp32: wgrad_tn_x3, 128 x {128, 64, 32})
bool wgrad_tn_multi_ok(bool bf16, long T, int M, int N) {
    if (M <= 0 || N <= 0 || M % 8 || N % 8 || T <= 0 || T > 0x7fffffffL) return false;
    return bf16 ? (T >= 3 * TBK && T % TBK == 0) : true;
}
static int x3_bn(int N) { return N > 64 ? 128 : (N > 32 ? 64 : 32); }

// workgroups per launch group: bf16 64 KB of LDS (two per CU, three rounds), fp32 48 KB (three per CU, four rounds: 1024 / 1536 / 2304 / 3072 / 4608 measured at 827.6 / 836.0 / 833.5 / 842.1 / 837.9 imgs/s on config 2, same box)
constexpr long kMultiTargetWgs = 1536, kMultiTargetWgsF32 = 3072;

// k-splits of every job, planned together: work = tiles x k-steps; a workgroup should get total / target k-steps (at least 8); bf16: every split at
// least three k-steps (the ring's depth) and no job more slab bytes than wgrad_slab_cap allows; fp32: at least two k-steps, no cap (see g_slab_ratio)
void wgrad_tn_multi_plan(bool bf16, const long *T, const int *M, const int *N, int *nsplit, int njobs) {
    double total = 0.0;
    for (int j = 0; j < njobs; ++j) {
        const int bn = bf16 ? TBN : x3_bn(N[j]);
        total += (double)((M[j] + TBM - 1) / TBM) * ((N[j] + bn - 1) / bn) * (double)((T[j] + TBK - 1) / TBK);
    }
    double per_wg = total / (double)(g_multi_wgs > 0 ? (long)g_multi_wgs : (bf16 ? kMultiTargetWgs : kMultiTargetWgsF32));
    if (per_wg < 8.0) per_wg = 8.0;
    for (int j = 0; j < njobs; ++j) {
        const long ksteps = (T[j] + TBK - 1) / TBK;
        long ns = (long)((double)ksteps / per_wg + 0.5);
        const long most = ksteps / (bf16 ? 3 : 2);
        if (ns > most) ns = most;
        const long cap = wgrad_slab_cap(T[j], M[j], N[j], bf16 ? 2 : 4);
        if (ns > cap) ns = cap;
        if (ns > 256) ns = 256;
        if (ns < 1) ns = 1;
        const long kl = ((ksteps + ns - 1) / ns) * TBK;
        nsplit[j] = (int)((T[j] + kl - 1) / kl);
    }
}

template <typename F>
static int multi_fill_and_launch(bool bf16, const int *sel, int nsel, const void *const *dY, const void *const *X, float *const *slabs, const long *T,
                                 const int *M, const int *N, const int *nsplit, const int *bias, int bn, F launch) {
    int order[kMultiMax];
    for (int base = 0; base < nsel; base += kMultiMax) {
        const int n = nsel - base < kMultiMax ? nsel - base : kMultiMax;
        for (int i = 0; i < n; ++i) order[i] = sel[base + i];
        for (int i = 1; i < n; ++i) {       // longest workgroups first (k-steps per split, descending): the launch's tail is made of short ones
            const int v = order[i];
            const long kv = T[v] / nsplit[v];
            int q = i - 1;
            while (q >= 0 && T[order[q]] / nsplit[order[q]] < kv) { order[q + 1] = order[q]; --q; }
            order[q + 1] = v;
        }
        WgradMultiTable t;
        memset(&t, 0, sizeof(t));
        long blk = 0;
        for (int i = 0; i < n; ++i) {
            const int j = order[i];
            if (!wgrad_tn_multi_ok(bf16, T[j], M[j], N[j]) || nsplit[j] < 1) return SD_E_UNSUPPORTED;
            const long ksteps = (T[j] + TBK - 1) / TBK;
            const long kl = ((ksteps + nsplit[j] - 1) / nsplit[j]) * TBK;
            if ((T[j] + kl - 1) / kl != nsplit[j] || (bf16 && kl < 3 * TBK)) return SD_E_SHAPE;       // not a plan of wgrad_tn_multi_plan
            if ((reinterpret_cast<uintptr_t>(dY[j]) | reinterpret_cast<uintptr_t>(X[j]) | reinterpret_cast<uintptr_t>(slabs[j])) & 15) return SD_E_ALIGN;
            t.A[i] = dY[j];
            t.B[i] = X[j];
            t.C[i] = slabs[j];
            t.T[i] = T[j];
            t.M[i] = M[j];
            t.N[i] = N[j];
            t.klen[i] = (int)kl;
            t.tiles_m[i] = (M[j] + TBM - 1) / TBM;
            t.tiles_n[i] = (N[j] + bn - 1) / bn;
            t.nsplit[i] = nsplit[j];
            t.bias[i] = bias[j];
            const long nb = (long)t.tiles_m[i] * t.tiles_n[i] * nsplit[j];
            if (nb > 0x3fffffffL) return SD_E_SHAPE;
            t.nblk[i] = (int)nb;
            t.blk_begin[i] = (int)blk;
            blk += (nb + 7) / 8 * 8;
            if (blk > 0x7fffffffL) return SD_E_SHAPE;
        }
        t.blk_begin[n] = (int)blk;
        t.njobs = n;
        launch(t, (unsigned)blk);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return SD_OK;
}

int wgrad_tn_multi_launch(bool bf16, const void *const *dY, const void *const *X, float *const *slabs, const long *T, const int *M, const int *N,
                          const int *nsplit, const int *bias, int njobs, hipStream_t st) {
    int *sel = new int[njobs];
    int rc = SD_OK;
    if (bf16) {
        for (int j = 0; j < njobs; ++j) sel[j] = j;
        rc = multi_fill_and_launch(true, sel, njobs, dY, X, slabs, T, M, N, nsplit, bias, TBN, [&](const WgradMultiTable &t, unsigned blk) {
            hipLaunchKernelGGL(wgrad_tn_bf16_ring_multi, dim3(blk), dim3(256), 0, st, t);
        });
    } else {
        for (int bn = 128; bn >= 32 && rc == SD_OK; bn >>= 1) {        // one launch group per tile width
            int n = 0;
            for (int j = 0; j < njobs; ++j)
                if (x3_bn(N[j]) == bn) sel[n++] = j;
            if (!n) continue;
            rc = multi_fill_and_launch(false, sel, n, dY, X, slabs, T, M, N, nsplit, bias, bn, [&](const WgradMultiTable &t, unsigned blk) {
                if (bn == 128) hipLaunchKernelGGL((wgrad_tn_x3_multi<128, 2, 2>), dim3(blk), dim3(256), 0, st, t);
                else if (bn == 64) hipLaunchKernelGGL((wgrad_tn_x3_multi<64, 2, 2>), dim3(blk), dim3(256), 0, st, t);
                else hipLaunchKernelGGL((wgrad_tn_x3_multi<32, 4, 1>), dim3(blk), dim3(256), 0, st, t);
            });
        }
    }
    delete[] sel;
    return rc;
}

// fp32 storage: which products take wgrad_tn_x3, and with how many k-splits (0: not this kernel's)
int wgrad_tn_x3_plan(long T, int M, int N, int *klen, int *bn) {
    if (T < 8192 || T > 0x7fffffffL || M < 128 || N < 32 || M % 8 || N % 8) return 0;          // tall-skinny, at least one full tile row of dY
    const int BNt = N > 64 ? 128 : (N > 32 ? 64 : 32);
    const long tiles = (long)((M + TBM - 1) / TBM) * ((N + BNt - 1) / BNt);
    if (tiles > 32) return 0;                                                                    // large weights: the split-K plan of token_gemm.hip
    long ns = 768 / tiles;
    if (ns > T / (4 * TBK)) ns = T / (4 * TBK);
    if (ns > 256) ns = 256;
    if (ns < 1) ns = 1;
    const long kl = ((T + ns - 1) / ns + TBK - 1) / TBK * TBK;
    *klen = (int)kl;
    *bn = BNt;
    return (int)((T + kl - 1) / kl);
}

}  // namespace sd

extern "C" {

/* fp32 token-major Linear weight gradient dW = dY^T . X as split-K slabs in split-bf16 arithmetic on transposed LDS reads (see above): number of
 * [out x in] fp32 slabs sd_linear_wgrad_tn writes for this shape, 0 when the shape is not this kernel's (few tokens, out_features < 128,
 * features not multiples of 8, more than 32 tiles).  The caller combines the slabs (sd_multi_slab_reduce); with_bias: each slab is [out x in] followed by
 * `out` column sums of dY over the split's tokens (the bias gradient), i.e. out * in + out floats. */
int sd_linear_wgrad_tn_slabs(long tokens, int out_features, int in_features) {
    int klen, bn;
    return sd::wgrad_tn_x3_plan(tokens, out_features, in_features, &klen, &bn);
}

int sd_linear_wgrad_tn(const float *dY, const float *X, float *slabs, size_t slabs_bytes, long tokens, int out_features, int in_features, int with_bias,
                       void *stream) {
    if (!dY || !X || !slabs) return SD_E_NULL;
    int klen = 0, bn = 0;
    const int M = out_features, N = in_features;
    const int nsplit = sd::wgrad_tn_x3_plan(tokens, M, N, &klen, &bn);
    if (!nsplit) return SD_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(X)) & 15) return SD_E_ALIGN;
    if (slabs_bytes < (size_t)nsplit * ((size_t)M * N + (with_bias ? M : 0)) * sizeof(float)) return SD_E_WORKSPACE;
    const int tiles_m = (M + sd::TBM - 1) / sd::TBM, tiles_n = (N + bn - 1) / bn;
    const dim3 grid((unsigned)((long)tiles_m * tiles_n * nsplit));
    hipStream_t st = static_cast<hipStream_t>(stream);
#define SD_X3(BNN, WMM, WNN)                                                                                                                       \
    do {                                                                                                                                           \
        if (with_bias) hipLaunchKernelGGL((sd::wgrad_tn_x3<BNN, WMM, WNN, true>), grid, dim3(256), 0, st, dY, X, slabs, M, N, tokens, klen, tiles_m, tiles_n, nsplit); \
        else hipLaunchKernelGGL((sd::wgrad_tn_x3<BNN, WMM, WNN, false>), grid, dim3(256), 0, st, dY, X, slabs, M, N, tokens, klen, tiles_m, tiles_n, nsplit);         \
    } while (0)
    if (bn == 128) SD_X3(128, 2, 2);
    else if (bn == 64) SD_X3(64, 2, 2);
    else SD_X3(32, 4, 1);
#undef SD_X3
    return (int)hipGetLastError();
}


int sd_linear_wgrad_tn_multi_supported(int dtype, long tokens, int out_features, int in_features) {
    if (dtype != SD_F32 && dtype != SD_BF16) return 0;
    return sd::wgrad_tn_multi_ok(dtype == SD_BF16, tokens, out_features, in_features) ? 1 : 0;
}

int sd_linear_wgrad_tn_multi_plan(sd_wgrad_job *jobs, int njobs, int dtype) {
    if (!jobs) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (njobs <= 0 || njobs > 4096) return SD_E_SHAPE;
    const bool bf16 = dtype == SD_BF16;
    long *T = new long[njobs];
    int *M = new int[njobs], *N = new int[njobs], *ns = new int[njobs];
    int rc = SD_OK;
    for (int j = 0; j < njobs; ++j) {
        T[j] = jobs[j].tokens, M[j] = jobs[j].out_features, N[j] = jobs[j].in_features;
        if (!sd::wgrad_tn_multi_ok(bf16, T[j], M[j], N[j])) rc = SD_E_UNSUPPORTED;
    }
    if (rc == SD_OK) {
        sd::wgrad_tn_multi_plan(bf16, T, M, N, ns, njobs);
        for (int j = 0; j < njobs; ++j) jobs[j].nsplit = ns[j];
    }
    delete[] T;
    delete[] M;
    delete[] N;
    delete[] ns;
    return rc;
}

int sd_linear_wgrad_tn_multi(const sd_wgrad_job *jobs, int njobs, int dtype, void *stream) {
    if (!jobs) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (njobs <= 0 || njobs > 4096) return SD_E_SHAPE;
    const void **dY = new const void *[njobs], **X = new const void *[njobs];
    float **C = new float *[njobs];
    long *T = new long[njobs];
    int *M = new int[njobs], *N = new int[njobs], *ns = new int[njobs], *bias = new int[njobs];
    int rc = SD_OK;
    for (int j = 0; j < njobs; ++j) {
        if (!jobs[j].dY || !jobs[j].X || !jobs[j].slabs) rc = SD_E_NULL;
        dY[j] = jobs[j].dY, X[j] = jobs[j].X, C[j] = jobs[j].slabs;
        T[j] = jobs[j].tokens, M[j] = jobs[j].out_features, N[j] = jobs[j].in_features, ns[j] = jobs[j].nsplit, bias[j] = jobs[j].with_bias;
    }
    if (rc == SD_OK) rc = sd::wgrad_tn_multi_launch(dtype == SD_BF16, dY, X, C, T, M, N, ns, bias, njobs, static_cast<hipStream_t>(stream));
    delete[] dY;
    delete[] X;
    delete[] C;
    delete[] T;
    delete[] M;
    delete[] N;
    delete[] ns;
    delete[] bias;
    return rc;
}

}  // extern "C"
