// align_stream.hip -- the NCHW 1x1 feature-alignment projection, forward, for SHORT reductions (Cs <= 128: BASELINE config 4, 128 -> 512):
//   Y_b [Ct x P] = W [Ct x Cs] . X_b [Cs x P] + bias          reference opts.py:25-27, losses.py:258,332-333,373-374 (`self.ff`, a 1x1 Conv2d)
// in split-bf16 arithmetic (every fp32 operand split exactly into three bf16 terms, six products, fp32 accumulation: fp32-grade, the same
// arithmetic as csrc/token_gemm.hip's X3 kernels and held to the same bound by tests/test_align_gpu.py).
//
// Why a kernel of its own (round 6): on the generic pipelined 128 x 128 GEMM this product is four k-steps of 32 -- a workgroup's life is its
// prologue and epilogue -- and ran at 21 % of either roof (49.7 us; 84 MB of HBM traffic = 10.5 us, 25.8 GF of bf16 MFMA work = 10.3 us).
// Here NOTHING of W is ever re-read or re-split: a wave owns 32 output channels and keeps their W rows -- all of K, already split into the three
// bf16 planes, in MFMA A-fragment order -- in REGISTERS (K / 16 x 12 VGPRs = 96 at K = 128) for its whole life, and the workgroup (4 waves = 128
// channels) streams pixel tiles of its image range past them:
//   * X tile [K][64 pixels] fp32: one 8 k x 4 pixel block per thread (8 coalesced 16-byte loads, requested a whole tile ahead), split ONCE per
//     workgroup and parked in LDS as three bf16 planes [pixel][k] (256-byte rows, 16-byte chunks XOR-swizzled so that both the 4-pixel-strided
//     writes and the fragment reads of 16 consecutive pixels hit 16 different chunks): a B fragment is one ds_read_b128;
//   * per tile and wave: 2 pixel blocks x K/16 k-steps x 6 = 96 v_mfma_f32_32x32x16_bf16 against 48 fragment reads, no other arithmetic;
//   * D[channel][pixel] leaves the matrix pipe with the PIXEL on the lane: every store instruction writes two full 128-byte lines of Y; the bias
//     is the accumulators' start value.
// The four channel slices of one pixel range are consecutive logical workgroups on ONE XCD (they share X through that L2): HBM sees X once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <type_traits>

#include "sd_common.h"

namespace sd {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int kPix = 64;                     // pixels per tile
constexpr int kPitch = 256;                  // bytes per pixel row of a plane (K <= 128 bf16)
constexpr int kPlane = kPix * kPitch;        // 16 KB
constexpr int kChan = 128;                   // channels per workgroup
constexpr int kImgPitch = 32;                // floats per row of a wave's private output image [32 channel rows][32 pixels]: 128-byte rows put the four
                                             // row pieces of every ds_read_b128 lane group on disjoint quarters of the 64 banks (a padded pitch does not)
constexpr int kImgBytes = 32 * kImgPitch * 4;
constexpr int kLdsBytes = 3 * kPlane + 4 * kImgBytes;      // 49152 + 16384 = 65536 B: two workgroups per CU
constexpr int kStores = 8;                   // 16-byte stores per thread and tile

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

// chunk swizzle key of pixel row px = 16 a + 4 b + c (0..63): 4 b + (c ^ (2 (a & 1) | (b >> 1))).
//  * fragment reads (ds_read_b128: four groups of 16 lanes, banks = 16-byte chunk within the 256-byte row): the 16 pixels of every group -- they mix
//    two aligned blocks of 16, e.g. {0-3, 12-15, 20-27} -- get 16 different keys;
//  * staging writes (ds_write_b128: eight groups of 8 consecutive lanes = pixels 4 g + j, g = 0..7 or 8..15, banks = chunk mod 8): 8 different keys mod 8.
__device__ __forceinline__ int key_of(int px) {
    const int a = px >> 4, b = (px >> 2) & 3, c = px & 3;
    return 4 * b + (c ^ (((a & 1) << 1) | (b >> 1)));
}

// a 16-byte load the compiler neither sinks nor reorders (waited for by the explicit s_waitcnt below): wave-uniform base in scalar registers + a
// 32-bit byte offset per lane -- one VGPR of address per request instead of two per row
// hazard (gfx9): a VALU instruction that WRITES an SGPR (the v_readlane that restores a spilled base pointer, a v_readfirstlane) followed by a
// vector-memory instruction that READS it needs 5 wait states; hipcc inserts them for its own instructions, not in front of inline asm -- the load
// then goes to a stale address (round 6: a memory fault in head_tail.hip as soon as a spilled pointer was involved).  Every asm load with a
// scalar operand therefore carries its own wait states (tools/asm_sgpr_hazard_scan.py checks the built code).
__device__ __forceinline__ f32x4 pinned_load16(const float *base, unsigned off) {
    f32x4 v;
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(off), "s"(base) : "memory");
    return v;
}
__device__ __forceinline__ void wait_loads() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

#ifdef SD_ALIGN_STREAM_STAMPS
// diagnostic build only (tools/align_stream_bench.py --stamps): s_memtime sums per phase of wave 0 of logical workgroup 0
__device__ unsigned long long g_stream_stamps[16];
#define SD_ST(i)                                                        \
    do {                                                                \
        const unsigned long long tnow = __builtin_amdgcn_s_memtime();   \
        ph[i] += tnow - tlast;                                          \
        tlast = tnow;                                                   \
    } while (0)
#else
#define SD_ST(i)
#endif

// grid.x = slices * B * chunks (XCD-remapped); a workgroup = channel slice `ms`, image b, pixel tiles [chunk * tpw, (chunk + 1) * tpw)
template <int KS>
__global__ __launch_bounds__(256, 2) void align_fwd_stream(const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias,
                                                            float *__restrict__ Y, int Cs, int Ct, long P, int tiles_per_img, int tpw,
                                                            int chunks_per_img, int slices) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];       // [3 planes][64 px][256 B] | [4 waves][32 rows][kImgPitch floats]
    const long nblk = gridDim.x, id = blockIdx.x;
    const long qd = nblk / 8, rem = nblk % 8, xcd = id % 8;
    const long L = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + id / 8;      // consecutive logical ids share an XCD
    const int ms = (int)(L % slices);
    const long rest = L / slices;
    const int chunk = (int)(rest % chunks_per_img), b = (int)(rest / chunks_per_img);
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int col = lane & 31, kg = lane >> 5;
    const int m0 = ms * kChan + wave * 32;

    // ---- staging geometry: thread -> (k group of 8 = t >> 4, pixel group of 4 = t & 15); K = 16 KS -> 2 KS k groups ----
    const int kgp = t >> 4, pxg = t & 15;
    const bool stager = kgp < 2 * KS;
    const float *Xb = X + (size_t)b * Cs * P;
    float *Yb = Y + (size_t)b * Ct * P;
    const int tile_begin = chunk * tpw, tile_end = min(tile_begin + tpw, tiles_per_img);
    if (tile_begin >= tile_end) return;
#ifdef SD_ALIGN_STREAM_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
    const unsigned long long tstart = tlast;
#endif
    f32x4 xr[8];
    const unsigned xrow = (unsigned)((long)(8 * (stager ? kgp : 0)) * P * 4);       // byte offset of this thread's first k row (Cs * P * 4 < 2^32: launcher)
    auto request = [&](int tile) {
        long p = (long)tile * kPix + 4 * pxg;
        if (p >= P) p = 0;                         // a pixel group past the image: any valid address (its columns are never stored)
        const unsigned off = xrow + (unsigned)(p * 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) xr[j] = pinned_load16(Xb + (size_t)j * P, off);
    };
    request(tile_begin);                           // first: in flight while the W rows are fetched and split

    // ---- this wave's 32 rows of W, all of K, split once into A fragments (lane: row m0 + col, k = 16 s + 8 kg .. + 7) ----
    bf16x8 ah[KS], am[KS], al[KS];
    {
        const float *wrow = W + (size_t)min(m0 + col, Ct - 1) * Cs + 8 * kg;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(wrow + 16 * s), v1 = *reinterpret_cast<const f32x4 *>(wrow + 16 * s + 4);
            const float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            split8(x, ah[s], am[s], al[s]);
        }
    }
    // the wave's 32 bias values are wave-uniform addresses: scalar registers.  Accumulator register e holds channel row (e & 3) + 8 (e >> 2) + 4 kg.
    float bs[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) bs[i] = bias ? bias[min(m0 + i, Ct - 1)] : 0.f;

    // INTERIOR (block-uniform): every channel row and every pixel of the workgroup's tiles exists -> unguarded stores, whose number per thread is
    // then known: the wait for the next tile's operands leaves exactly those youngest operations (the stores) in flight instead of draining them
    SD_ST(0);                                                  // prologue: W rows fetched and split, first tile requested
    auto body = [&](auto interior) {
        constexpr bool IN = decltype(interior)::value;
        for (int tile = tile_begin; tile < tile_end; ++tile) {
            if (IN && tile > tile_begin) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kStores) : "memory");
                __builtin_amdgcn_sched_barrier(0);
            } else {
                wait_loads();
            }
            SD_ST(1);                                          // waited for the tile's operands
            if (stager) {
                // 8 k x 4 pixels in registers -> per pixel one 16-byte chunk (8 consecutive k) of each plane
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x[8] = {xr[0][j], xr[1][j], xr[2][j], xr[3][j], xr[4][j], xr[5][j], xr[6][j], xr[7][j]};
                    bf16x8 h, m, l;
                    split8(x, h, m, l);
                    const int px = 4 * pxg + j;
                    unsigned char *d = lds + px * kPitch + ((kgp ^ key_of(px)) << 4);
                    *reinterpret_cast<bf16x8 *>(d) = h;
                    *reinterpret_cast<bf16x8 *>(d + kPlane) = m;
                    *reinterpret_cast<bf16x8 *>(d + 2 * kPlane) = l;
                }
            }
            SD_ST(2);                                          // split + LDS stores
            __syncthreads();
            SD_ST(3);                                          // barrier
            // the next tile's operands: in flight during the MFMAs and stores below (past the last tile: a repeat of it, so that every iteration
            // issues the same number of operations -- the counted wait above)
            request(tile + 1 < tile_end ? tile + 1 : tile);
            f32x16 acc[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[nt][e] = kg ? bs[(e & 3) + 8 * (e >> 2) + 4] : bs[(e & 3) + 8 * (e >> 2)];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int px = 32 * nt + col;
                    const unsigned char *q = lds + px * kPitch + (((2 * s + kg) ^ key_of(px)) << 4);
                    const bf16x8 bh = *reinterpret_cast<const bf16x8 *>(q), bm = *reinterpret_cast<const bf16x8 *>(q + kPlane),
                                 bl = *reinterpret_cast<const bf16x8 *>(q + 2 * kPlane);
                    f32x16 c = acc[nt];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[s], bm, c, 0, 0, 0);      // small terms first (token_gemm.hip's order)
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s], bh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bl, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[s], bh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bm, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bh, c, 0, 0, 0);
                    acc[nt] = c;
                }
            }
#ifdef SD_ALIGN_STREAM_STAMPS
            asm volatile("" ::"v"(acc[0][0]), "v"(acc[1][15]));
#endif
            SD_ST(4);                                          // requests + fragment reads + MFMAs
            // ---- stores.  D[channel][pixel] has the pixel on the lane: stored as it is, every instruction would move 4 bytes per lane (32 store
            // instructions per tile and wave; measured: 2800 cycles per tile in the store phase, the 6-bit vmcnt queue full of 256-byte stores).
            // Instead each 32 x 32 block is parked in the wave's PRIVATE LDS image (ds_write_b32 straight from the accumulator layout, conflict-free;
            // no barrier: a wave's LDS operations execute in order) and read back row-wise: 16 bytes per lane, 8 lanes = one full 128-byte line
            // of a channel row, 8 rows per instruction, 4 store instructions per block.
            float *img = reinterpret_cast<float *>(lds + 3 * kPlane) + wave * (32 * kImgPitch);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                for (int e = 0; e < 16; ++e) img[((e & 3) + 8 * (e >> 2) + 4 * kg) * kImgPitch + col] = acc[nt][e];
                const int c4 = (lane & 7) * 4, rsub = lane >> 3;
                const long p = (long)tile * kPix + 32 * nt + c4;
                const unsigned voff = (unsigned)(rsub * P + p);                     // elements; Ct * P < 2^30 (launcher)
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(img + (8 * it + rsub) * kImgPitch + c4);
                    float *rowp = Yb + (size_t)(m0 + 8 * it) * P;                   // wave-uniform: scalar base + one 32-bit lane offset
                    if (IN || (p < P && m0 + 8 * it + rsub < Ct)) *reinterpret_cast<f32x4 *>(rowp + voff) = v;
                }
            }
            SD_ST(5);                                          // stores issued
            __syncthreads();                                   // every wave is done with the planes before the next tile overwrites them
            SD_ST(6);                                          // barrier
        }
        wait_loads();                                          // the surplus request of the last iteration
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(xr[j]));          // ... whose registers stay allocated until it has landed
    };
    if (ms * kChan + kChan <= Ct && (long)tile_end * kPix <= P) body(std::true_type{});
    else body(std::false_type{});
#ifdef SD_ALIGN_STREAM_STAMPS
    if (L == 0 && t == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i) g_stream_stamps[i] = ph[i];
        g_stream_stamps[7] = __builtin_amdgcn_s_memtime() - tstart;
        g_stream_stamps[8] = (unsigned long long)(tile_end - tile_begin);
    }
#endif
}

}  // namespace

int g_align_stream = 1;        // tunable "align_stream" (A/B, tests): 0 = the generic pipelined GEMM of token_gemm.hip for every shape

// -> SD_E_UNSUPPORTED when the shape is not this kernel's (the caller then takes the generic path)
int align_f32_fwd_stream(const float *X, const float *W, const float *bias, float *Y, int B, int Cs, int Ct, long P, hipStream_t st) {
    if (!g_align_stream || Cs % 16 != 0 || Cs > 128 || P % 4 != 0 || P <= 0 || B <= 0 || Ct <= 0) return SD_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W)) & 15) return SD_E_UNSUPPORTED;
    const int slices = (Ct + kChan - 1) / kChan;
    const long tiles_per_img = (P + kPix - 1) / kPix;
    if (tiles_per_img > 0x7fffffffL) return SD_E_UNSUPPORTED;
    // ~512 workgroups (two per CU in one round); at least 2 tiles per workgroup where there are that many (the W rows are split once per workgroup)
    long tpw = (long)slices * B * tiles_per_img / 512;
    tpw = tpw < 2 ? 2 : (tpw > 16 ? 16 : tpw);
    if (tpw > tiles_per_img) tpw = tiles_per_img;
    const long chunks = (tiles_per_img + tpw - 1) / tpw;
    const long nblk = (long)slices * B * chunks;
    if (nblk > 0x7fffffffL || (long)Cs * P >= (1L << 30) || (long)Ct * P >= (1L << 30)) return SD_E_UNSUPPORTED;
    const dim3 g((unsigned)nblk), blk(256);
#define SD_AS(KS_)                                                                                                                                   \
    do {                                                                                                                                          \
        static bool raised = false;       /* per instantiation; idempotent, so a race is harmless */                                             \
        if (!raised) {                                                                                                                            \
            hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(align_fwd_stream<KS_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                kLdsBytes);                                                                                       \
            if (e_ != hipSuccess) return (int)e_;                                                                                                 \
            raised = true;                                                                                                                        \
        }                                                                                                                                         \
        hipLaunchKernelGGL((align_fwd_stream<KS_>), g, blk, kLdsBytes, st, X, W, bias, Y, Cs, Ct, P, (int)tiles_per_img, (int)tpw, (int)chunks,   \
                           slices);                                                                                                               \
    } while (0)
    switch (Cs / 16) {
        case 1: SD_AS(1); break;
        case 2: SD_AS(2); break;
        case 3: SD_AS(3); break;
        case 4: SD_AS(4); break;
        case 5: SD_AS(5); break;
        case 6: SD_AS(6); break;
        case 7: SD_AS(7); break;
        default: SD_AS(8); break;
    }
#undef SD_AS
    return (int)hipGetLastError();
}

#ifdef SD_ALIGN_STREAM_STAMPS
extern "C" int sd_debug_align_stream_stamps(unsigned long long *out16) {       // diagnostic build only (not in the header)
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stream_stamps), 16 * sizeof(unsigned long long));
}
#endif

int align_stream_tunable(const char *key, int set, int v) {
    if (strcmp(key, "align_stream")) return SD_E_UNSUPPORTED;
    if (!set) return g_align_stream;
    if (v < 0 || v > 1) return SD_E_SHAPE;
    g_align_stream = v;
    return SD_OK;
}

}  // namespace sd
