// head_tail.hip -- the tail of a FROZEN SegFormer head in one pass (round 6), fp32 token-major branch maps in, class planes out:
//   logits [B, classes, H, W] = W_p . relu( scale * (z1 + up2(z2) + up4(z3) + up8(z4) + b_f) + shift ) + b_p
// reference segformer_head.py:75-98 (`_c = self.linear_fuse(torch.cat([_c4, _c3, _c2, _c1], dim=1)); x = self.dropout(_c); x = self.linear_pred(x)`)
// for a head in eval mode nobody differentiates through (the teacher).  decode_heads/segformer_head.py already runs the fuse conv per branch at
// native resolution (z_i = W_i c_i); as two kernels the summed, normalised map [B, H W, E] -- 403 MB at E = 768, 8 x 128 x 128 -- is written by
// csrc/headfuse.hip::upsum_fwd_strip and read back by the linear_pred product.  Here it only exists as a workgroup's LDS tile.
// Arithmetic: the interpolation / sum / affine / ReLU operation for operation as upsum_fwd_strip (same clamped taps, same fmaf order), the product
// in split-bf16 arithmetic as csrc/token_gemm.hip (three bf16 planes per operand, six products, small terms first, fp32 accumulation, k
// ascending); W_p arrives pre-split as row-major planes [3][classes][E] (sd_presplit_multi, row_planes -- what sd_linear_nchw_fwd_planes takes).
//
// A workgroup (256 threads) owns 4 rows x 32 columns of one image (128 pixels) and walks E in chunks of 32 channels:
//   * W_p's chunk [160 rows][32 k] x 3 planes goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers; the 80-byte row pitch is
//     kept by letting every fifth lane of a 1 KB run land on the 16 bytes of padding), issued at the top of the chunk under the sum stage;
//   * the coarse branches' taps of the tile and chunk -- 4 x 18, 3 x 10 and 2 x 6 coarse pixels, 14.6 KB -- are parked in LDS by DMA a chunk ahead (under
//     the matrix stage): read from global memory per thread they were 72 KB of loads per chunk and workgroup for that footprint, a third of a wave's
//     chunk spent waiting to issue them (profiles/r06_head_tail_stamps.txt);
//   * sum stage: thread -> (4 channels, 4 consecutive pixels of one row): z1 (4 vectors) and the chunk's scale / shift / bias requested a chunk ahead
//     into registers, the 18 tap vectors read from LDS; 16 results split into the pixel tile [3][128 px][32 k];
//   * matrix stage: wave w = tile row w (32 consecutive pixels) against all 160 class rows: D[class][pixel] has the PIXEL on the lane, so every
//     store instruction writes two full 128-byte lines of the class planes.
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
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int kTH = 4, kTW = 32, kTM = kTH * kTW;   // pixel tile: 4 rows x 32 columns
constexpr int kKC = 32;                             // channels per chunk
constexpr int kPitch = 2 * kKC + 16;                // 80 bytes per LDS row
constexpr int kPPlane = kTM * kPitch;               // 10240
constexpr int kRows = 160;                          // class rows of the product (5 MFMA row blocks)
constexpr int kWPlane = kRows * kPitch;             // 12800
constexpr int kWRuns = (3 * kWPlane + 1023) / 1024;  // 1 KB DMA runs covering the three W planes: 38 (the last one half used)
constexpr int kWRunsPerWave = (kWRuns + 3) / 4;     // 10

__device__ __forceinline__ void split2(float x0, float x1, bf16x2 &h, bf16x2 &m, bf16x2 &l) {
    const f32x2 v = {x0, x1};
    h = __builtin_convertvector(v, bf16x2);
    const f32x2 r1 = v - __builtin_convertvector(h, f32x2);
    m = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
    l = __builtin_convertvector(r2, bf16x2);
}

// a 16-byte load the compiler neither sinks nor reorders (wave-uniform base + 32-bit byte offset per lane), waited for by wait_loads(); "+v": the
// destination is the loop-carried variable's own register (tools/asm_pending_audit.py)
// hazard (gfx9): a VALU instruction that WRITES an SGPR (the v_readlane that restores a spilled base pointer, a v_readfirstlane) followed by a
// vector-memory instruction that READS it needs 5 wait states; hipcc inserts them for its own instructions, not in front of inline asm -- the load
// then goes to a stale address (round 6: a memory fault in head_tail.hip as soon as a spilled pointer was involved).  Every asm load with a
// scalar operand therefore carries its own wait states (tools/asm_sgpr_hazard_scan.py checks the built code).
__device__ __forceinline__ void pinned_load16(f32x4 &v, const void *base, unsigned off) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "+v"(v) : "v"(off), "s"(base) : "memory");
}
// global -> LDS, 16 bytes per lane (tok_gemm_bf16.hip): M0 = wave-uniform LDS byte address of lane 0's 16 bytes, lane l lands at M0 + 16 l
__device__ __forceinline__ void dma16(const void *base, unsigned lane_off, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(base), "s"(lds_byte) : "memory");
}
__device__ __forceinline__ const void *uniform_ptr(const void *p) {       // a wave-uniform pointer the compiler holds in vector registers -> scalar
    const unsigned long long v = (unsigned long long)(uintptr_t)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const void *)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void wait_loads() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

#ifdef SD_HEAD_TAIL_STAMPS
// diagnostic build only (tools/head_tail_bench.py --stamps): s_memtime sums per phase of wave 0 of logical workgroup 0
__device__ unsigned long long g_head_tail_stamps[16];
#define SD_ST(i)                                                        \
    do {                                                                \
        const unsigned long long tnow = __builtin_amdgcn_s_memtime();   \
        ph[i] += tnow - tlast;                                          \
        tlast = tnow;                                                   \
    } while (0)
#else
#define SD_ST(i)
#endif

// one coarse branch of csrc/headfuse.hip::add_branch_strip on already loaded taps: vertical lerp of NC columns, then the four pixels
template <int F, int NC>
__device__ __forceinline__ void add_branch(float (&acc)[4][4], const f32x4 (&top)[NC], const f32x4 (&bot)[NC], float ly, const float (&lx)[4]) {
    float col[NC][4];
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) col[j][i] = fmaf(ly, bot[j][i] - top[j][i], top[j][i]);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        constexpr int kOff2[4] = {0, 1, 1, 2}, kOff4[4] = {0, 0, 1, 1};
        const int o = F == 2 ? kOff2[p] : (F == 4 ? kOff4[p] : 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[p][i] += fmaf(lx[p], col[o + 1][i] - col[o][i], col[o][i]);
    }
}

// coarse-branch taps of one tile and chunk, parked in LDS (128 bytes = 32 channels per coarse pixel; rows x columns the tile's 4 x 32 pixels can touch):
constexpr int kT2R = 4, kT2C = 18, kT3R = 3, kT3C = 10, kT4R = 2, kT4C = 6;
constexpr int kT2Off = 0, kT3Off = kT2R * kT2C * 128, kT4Off = kT3Off + kT3R * kT3C * 128, kTapBytes = kT4Off + kT4R * kT4C * 128;   // 9216, 13056, 14592
constexpr int kTapRuns = 9 + 4 + 2;                  // 1 KB DMA runs (8 coarse pixels each): x2 72 px, x4 30 px (last run 6 px), x8 12 px (last run 4 px)

// grid.x = B * (H / 4) * (W / 32) (XCD-remapped: consecutive tiles of an image share one L2)
__global__ __launch_bounds__(256, 2) void head_tail_x3(const float *__restrict__ z1, const float *__restrict__ z2, const float *__restrict__ z3,
                                                       const float *__restrict__ z4, const float *__restrict__ fbias,
                                                       const float *__restrict__ scale, const float *__restrict__ shift,
                                                       const unsigned char *__restrict__ wplanes, const float *__restrict__ pbias,
                                                       float *__restrict__ out, int H, int W, int E, int classes, int tiles_x,
                                                       int tiles_per_img) {
    // pixel planes [3][128][80 B] | W planes [3][classes][80 B] | taps [14592 B].  (Fragment reads of class rows classes .. 159 run into the next
    // plane / the taps region: their products are never stored.)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wplane = classes * kPitch;
    unsigned char *ldsP = lds, *ldsW = lds + 3 * kPPlane, *ldsT = ldsW + 3 * wplane;
    const long nblk = gridDim.x, id = blockIdx.x;
    const long qd = nblk / 8, rem = nblk % 8, xcd = id % 8;
    const long L = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + id / 8;
    const int b = (int)(L / tiles_per_img), tr = (int)(L % tiles_per_img);
    const int y0 = (tr / tiles_x) * kTH, x0 = (tr % tiles_x) * kTW;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int col = lane & 31, kg = lane >> 5;

    // ---- sum-stage geometry: thread -> (channel quad c4, tile row r, 4 pixels from column xs) ----
    const int c4 = t & 7, strip = t >> 3;
    const int r = strip >> 3, xs = (strip & 7) * 4;
    const int Y = y0 + r, X0 = x0 + xs;
    const size_t HW = (size_t)H * W;
    const float *z1b = z1 + (size_t)b * HW * E, *z2b = z2 + (size_t)b * (HW / 4) * E, *z3b = z3 + (size_t)b * (HW / 16) * E,
                *z4b = z4 + (size_t)b * (HW / 64) * E;
    const unsigned o1 = (((unsigned)(Y * W + X0)) * (unsigned)E + 4u * c4) * 4u;
    // origins of the parked tap regions (un-clamped coarse coordinates of slot (0, 0)); y0 % 4 == 0, x0 % 32 == 0
    const int oy2 = y0 / 2 - 1, ox2 = x0 / 2 - 1, oy3 = y0 / 4 - 1, ox3 = x0 / 4 - 1, oy4 = (y0 & 4) ? y0 / 8 : y0 / 8 - 1, ox4 = x0 / 8 - 1;
    // this thread's taps inside them: csrc/headfuse.hip::add_branch_strip's (un-clamped) top-left tap minus the origin, and its vertical weight
    //   x2: floor((Y + .5) / 2 - .5) - oy2 = (r + 1) >> 1, weight .75 / .25;  x4: r >> 1, weights .625 .875 .125 .375;  x8: 0, (.0625 | .5625) + r / 8
    const unsigned ta2 = (unsigned)(kT2Off + (((r + 1) >> 1) * kT2C + xs / 2) * 128 + 16 * c4);
    const unsigned ta3 = (unsigned)(kT3Off + ((r >> 1) * kT3C + xs / 4) * 128 + 16 * c4);
    const unsigned ta4 = (unsigned)(kT4Off + ((int)floorf((xs + 0.5f) / 8 - 0.5f) + 1) * 128 + 16 * c4);
    const float ly2 = (r & 1) ? 0.25f : 0.75f;
    const float ly3 = r == 0 ? 0.625f : (r == 1 ? 0.875f : (r == 2 ? 0.125f : 0.375f));
    const float ly4 = ((y0 & 4) ? 0.0625f : 0.5625f) + 0.125f * r;
    const float lx2[4] = {0.75f, 0.25f, 0.75f, 0.25f}, lx3[4] = {0.625f, 0.875f, 0.125f, 0.375f};
    float lx4[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) lx4[p] = ((X0 & 4) ? 0.0625f : 0.5625f) + 0.125f * p;

    // ---- W_p by LDS-DMA: wave w issues the 1 KB runs w, w + 4, ...; lane l of run u lands at LDS byte o = 1024 u + 16 l of the W region
    //      [3 planes][classes rows][80 B]: plane o / (80 classes), row (o % (80 classes)) / 80, 16-byte piece (o % 80) / 16 -- piece 4 is the row's
    //      padding (any valid source); lanes beyond the region are switched off ----
    const int wbytes = 3 * wplane, wruns = (wbytes + 1023) / 1024;
    unsigned wsrc[kWRunsPerWave];
#pragma unroll
    for (int i = 0; i < kWRunsPerWave; ++i) {
        const int oraw = 1024 * (wave + 4 * i) + 16 * lane, o = min(oraw, wbytes - 16);
        const int pl = o / wplane, ro = o % wplane, row = ro / kPitch, pc = min((ro % kPitch) >> 4, 3);
        wsrc[i] = (((unsigned)pl * (unsigned)classes + (unsigned)row) * (unsigned)E + 8u * pc) * 2u;
    }
    // ---- the taps by LDS-DMA: run u = wave + 4 i: 0..8 the x2 map, 9..12 x4, 13..14 x8; lane -> (coarse pixel 8 (run in map) + (lane >> 3), 16-byte
    //      piece lane & 7); the slot's content is the map at the CLAMPED coordinates (what add_branch_strip loads) ----
    unsigned tsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int u = wave + 4 * i;
        const int rig = u < 9 ? u : (u < 13 ? u - 9 : u - 13), nc = u < 9 ? kT2C : (u < 13 ? kT3C : kT4C), np = u < 9 ? kT2R * kT2C : (u < 13 ? kT3R * kT3C : kT4R * kT4C);
        const int f = u < 9 ? 2 : (u < 13 ? 4 : 8), oy = u < 9 ? oy2 : (u < 13 ? oy3 : oy4), ox = u < 9 ? ox2 : (u < 13 ? ox3 : ox4);
        const int px = 8 * rig + (lane >> 3), pxc = min(px, np - 1);
        const int cy = min(max(oy + pxc / nc, 0), H / f - 1), cx = min(max(ox + pxc % nc, 0), W / f - 1);
        tsrc[i] = (((unsigned)(cy * (W / f) + cx)) * (unsigned)E + 4u * (lane & 7)) * 4u;
    }
    auto dma_taps = [&](int kc) {
        const unsigned tbase = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ldsT), cbb = (unsigned)(kc * kKC) * 4u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int u = wave + 4 * i;                   // wave-uniform: one of the three branches per run
            const int px = lane >> 3;
            if (u < 9) {
                dma16(z2b, tsrc[i] + cbb, __builtin_amdgcn_readfirstlane(tbase + (unsigned)(kT2Off + 1024 * u)));
            } else if (u < 13) {
                if (8 * (u - 9) + px < kT3R * kT3C) dma16(z3b, tsrc[i] + cbb, __builtin_amdgcn_readfirstlane(tbase + (unsigned)(kT3Off + 1024 * (u - 9))));
            } else if (u < kTapRuns) {
                if (8 * (u - 13) + px < kT4R * kT4C) dma16(z4b, tsrc[i] + cbb, __builtin_amdgcn_readfirstlane(tbase + (unsigned)(kT4Off + 1024 * (u - 13))));
            }
        }
    };

    // accumulators: D row = class 32 mt + (e & 3) + 8 (e >> 2) + 4 kg
    f32x16 acc[5];
#pragma unroll
    for (int mt = 0; mt < 5; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mt][e] = 0.f;

    // requested a chunk ahead into registers: z1 (4 vectors) and the chunk's scale / shift / bias (this thread's 4 channels)
    const bool hasb = fbias != nullptr;
    f32x4 q1[4], qs, qh, qb;
    {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) q1[i] = z;
        qs = z, qh = z, qb = z;
    }
    auto request = [&](int kc) {
        const unsigned cbb = (unsigned)(kc * kKC) * 4u, tb = cbb + 16u * c4;
#pragma unroll
        for (int p = 0; p < 4; ++p) pinned_load16(q1[p], z1b, o1 + (unsigned)p * (unsigned)E * 4u + cbb);
        pinned_load16(qs, scale, tb);
        pinned_load16(qh, shift, tb);
        pinned_load16(qb, hasb ? fbias : shift, tb);      // always 7 requests: the counted wait below is exact (without a bias the values are unused)
    };
    request(0);
    dma_taps(0);
    wait_loads();
    __syncthreads();                                     // chunk 0's taps visible
#ifdef SD_HEAD_TAIL_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
    const unsigned long long tstart = tlast;
#endif

    const int nchunk = E / kKC;
    for (int kc = 0; kc < nchunk; ++kc) {
        // -- W_p chunk -> LDS by DMA (the barrier at the end of the previous iteration freed both tiles); in flight under the sum stage --
        {
            const unsigned wbase = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ldsW);
#pragma unroll
            for (int i = 0; i < kWRunsPerWave; ++i)
                if (wave + 4 * i < wruns && 1024 * (wave + 4 * i) + 16 * lane < wbytes)      // (the last run is partial: its lanes beyond the region stay off)
                    dma16(wplanes, wsrc[i] + (unsigned)(kc * kKC) * 2u, __builtin_amdgcn_readfirstlane(wbase + 1024u * (unsigned)(wave + 4 * i)));
        }
        SD_ST(0);                                        // W chunk DMA issued
        // -- z1 + bias, then the branches in upsum_fwd_strip's order; the coarse taps come from LDS --
        float a[4][4];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i) a[p][i] = hasb ? q1[p][i] + qb[i] : q1[p][i];
        {
            f32x4 top[4], bot[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                top[j] = *reinterpret_cast<const f32x4 *>(ldsT + ta2 + 128 * j);
                bot[j] = *reinterpret_cast<const f32x4 *>(ldsT + ta2 + 128 * (j + kT2C));
            }
            add_branch<2, 4>(a, top, bot, ly2, lx2);
        }
        {
            f32x4 top[3], bot[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                top[j] = *reinterpret_cast<const f32x4 *>(ldsT + ta3 + 128 * j);
                bot[j] = *reinterpret_cast<const f32x4 *>(ldsT + ta3 + 128 * (j + kT3C));
            }
            add_branch<4, 3>(a, top, bot, ly3, lx3);
        }
        {
            f32x4 top[2], bot[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                top[j] = *reinterpret_cast<const f32x4 *>(ldsT + ta4 + 128 * j);
                bot[j] = *reinterpret_cast<const f32x4 *>(ldsT + ta4 + 128 * (j + kT4C));
            }
            add_branch<8, 2>(a, top, bot, ly4, lx4);
        }
        const f32x4 sc = qs, sh = qh;
        // every register of this chunk's request has been consumed: the next chunk's goes out now
        __builtin_amdgcn_sched_barrier(0);
#ifdef SD_HEAD_TAIL_STAMPS
        asm volatile("" ::"v"(a[0][0]), "v"(a[3][3]));
#endif
        SD_ST(1);                                        // sum stage
        request(kc + 1 < nchunk ? kc + 1 : kc);          // (past the last chunk: a repeat, never used)
        __builtin_amdgcn_sched_barrier(0);
        SD_ST(2);                                        // requests issued
        // -- affine + ReLU, split, pixel tile --
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float g[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) g[i] = fmaxf(fmaf(a[p][i], sc[i], sh[i]), 0.f);
            bf16x2 h0, m0, l0, h1, m1, l1;
            split2(g[0], g[1], h0, m0, l0);
            split2(g[2], g[3], h1, m1, l1);
            const bf16x4 hh = {h0[0], h0[1], h1[0], h1[1]}, mm = {m0[0], m0[1], m1[0], m1[1]}, ll = {l0[0], l0[1], l1[0], l1[1]};
            unsigned char *q = ldsP + (32 * r + xs + p) * kPitch + 8 * c4;
            *reinterpret_cast<uint2 *>(q) = __builtin_bit_cast(uint2, hh);
            *reinterpret_cast<uint2 *>(q + kPPlane) = __builtin_bit_cast(uint2, mm);
            *reinterpret_cast<uint2 *>(q + 2 * kPPlane) = __builtin_bit_cast(uint2, ll);
        }
        SD_ST(3);                                        // affine + split + LDS stores
        asm volatile("s_waitcnt vmcnt(7)" ::: "memory");    // the W chunk has landed: it is older than the 7 requests of the next chunk, which stay in flight
        SD_ST(4);                                        // wait for the W chunk
        __syncthreads();                                 // pixel tile + W chunk visible; every thread has read this chunk's taps
        SD_ST(5);                                        // barrier 1
        dma_taps(kc + 1 < nchunk ? kc + 1 : kc);         // the next chunk's taps: in flight under the matrix stage
        // -- matrix stage: A = W_p rows (classes), B = this wave's 32 pixels --
#pragma unroll
        for (int s = 0; s < kKC / 16; ++s) {
            const unsigned char *qb_ = ldsP + (32 * wave + col) * kPitch + 32 * s + 16 * kg;
            const bf16x8 bh = *reinterpret_cast<const bf16x8 *>(qb_), bm = *reinterpret_cast<const bf16x8 *>(qb_ + kPPlane),
                         bl = *reinterpret_cast<const bf16x8 *>(qb_ + 2 * kPPlane);
#pragma unroll
            for (int mt = 0; mt < 5; ++mt) {
                const unsigned char *qa = ldsW + (32 * mt + col) * kPitch + 32 * s + 16 * kg;
                const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(qa), am = *reinterpret_cast<const bf16x8 *>(qa + wplane),
                             al = *reinterpret_cast<const bf16x8 *>(qa + 2 * wplane);
                f32x16 c = acc[mt];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);      // small terms first (token_gemm.hip's order)
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
                acc[mt] = c;
            }
        }
#ifdef SD_HEAD_TAIL_STAMPS
        asm volatile("" ::"v"(acc[0][0]), "v"(acc[4][15]));
#endif
        SD_ST(6);                                        // fragment reads + MFMAs
        // the wait sits here, not at the loop top: whatever copies the compiler makes of the loop-carried request registers at the back edge read
        // arrived data (tools/asm_pending_audit.py)
        wait_loads();
        __syncthreads();                                 // fragment reads done, the next chunk's taps visible: the tiles may be overwritten
        SD_ST(7);                                        // wait for the next chunk's operands + barrier 2
    }
#ifdef SD_HEAD_TAIL_STAMPS
    if (L == 0 && t == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) g_head_tail_stamps[i] = ph[i];
        g_head_tail_stamps[8] = __builtin_amdgcn_s_memtime() - tstart;
        g_head_tail_stamps[9] = (unsigned long long)nchunk;
    }
#endif
    // ---- epilogue: class planes; lane = pixel x0 + col of row y0 + wave ----
    float *ob = out + (size_t)b * classes * HW + (size_t)(y0 + wave) * W + x0 + col;
#pragma unroll
    for (int mt = 0; mt < 5; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int cls = 32 * mt + (e & 3) + 8 * (e >> 2) + 4 * kg;
            if (cls < classes) ob[(size_t)cls * HW] = acc[mt][e] + (pbias ? pbias[cls] : 0.f);
        }
}

}  // namespace
}  // namespace sd

extern "C" {

#ifdef SD_HEAD_TAIL_STAMPS
int sd_debug_head_tail_stamps(unsigned long long *out16) {       // diagnostic build only (not in the header)
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(sd::g_head_tail_stamps), 16 * sizeof(unsigned long long));
}
#endif

int sd_head_tail_supported(int H, int W, int E, int classes) {
    return (H > 0 && W > 0 && H % 8 == 0 && W % sd::kTW == 0 && E % sd::kKC == 0 && E <= 1024 && classes >= 1 && classes <= sd::kRows &&
            (long)H * W * E < (1L << 30))
               ? 1
               : 0;
}

int sd_head_tail_f32(const float *z1, const float *z2, const float *z3, const float *z4, const float *fuse_bias /* or NULL */, const float *scale,
                     const float *shift, const void *pred_row_planes, const float *pred_bias /* or NULL */, float *logits, int B, int H, int W, int E,
                     int classes, void *stream) {
    if (!z1 || !z2 || !z3 || !z4 || !scale || !shift || !pred_row_planes || !logits) return SD_E_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || E <= 0 || classes <= 0) return SD_E_SHAPE;
    if (!sd_head_tail_supported(H, W, E, classes)) return SD_E_UNSUPPORTED;
    if ((long)B * (H / sd::kTH) * (W / sd::kTW) > 0x7fffffffL) return SD_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(z1) | reinterpret_cast<uintptr_t>(z2) | reinterpret_cast<uintptr_t>(z3) | reinterpret_cast<uintptr_t>(z4) |
         reinterpret_cast<uintptr_t>(fuse_bias) | reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift) |
         reinterpret_cast<uintptr_t>(pred_row_planes) | reinterpret_cast<uintptr_t>(logits)) & 15)
        return SD_E_ALIGN;
    const int tx = W / sd::kTW, tpi = tx * (H / sd::kTH);
    const size_t ldsb = (size_t)3 * sd::kPPlane + (size_t)3 * classes * sd::kPitch + (size_t)sd::kTapBytes;     // 81312 B at 150 classes: two workgroups per CU
    static bool raised = false;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sd::head_tail_x3), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        raised = true;
    }
    hipLaunchKernelGGL(sd::head_tail_x3, dim3((unsigned)((long)B * tpi)), dim3(256), ldsb, static_cast<hipStream_t>(stream), z1, z2, z3, z4, fuse_bias,
                       scale, shift, static_cast<const unsigned char *>(pred_row_planes), pred_bias, logits, H, W, E, classes, tx, tpi);
    return (int)hipGetLastError();
}

}  // extern "C"
