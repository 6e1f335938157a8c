// mixffn_tail.hip -- the second half of a FROZEN Mix-FFN in one pass over token-major fp32 activations (round 6):
//   Y [B, H*W, Cout] = fc2( GELU( dwconv3x3(h) + b_dw ) ) + b_2,      h [B, H*W, Ch] = fc1's output, Ch = 4 Cout
// reference mix_transformer.py:20-55 (`x = self.fc1(x); x = self.dwconv(x, H, W); x = self.act(x); ...; x = self.fc2(x)`), the teacher's
// stages 1-2 (131072 / 32768 tokens, Ch = 256 / 512): as two kernels the activated hidden map -- the largest tensor of the block -- is written by
// the depthwise pass and read back by the GEMM (2 x 134 MB per stage-1 block of Segformer-B2, 81 us + 54 us); here it only ever exists as one
// workgroup's LDS tile.  Arithmetic: the convolution / GELU exactly as csrc/dwconv.hip (same fmaf order, same gelu_erf), the product in
// split-bf16 arithmetic exactly as csrc/token_gemm.hip's X3 kernels (three bf16 planes per operand, six products, small terms first, fp32
// accumulation, k ascending).
//
// A workgroup (256 threads) owns an 8 x 16 pixel patch of one image (128 GEMM rows) and walks the hidden channels in chunks of 32:
//   * depthwise stage: thread -> (4 channels, 4 consecutive pixels of one patch row): 3 x 6 16-byte loads of h (halo columns / rows come from the
//     neighbouring patches' lines: L2), 144 FMAs, 16 GELUs, the 16 results split into the three bf16 planes and parked in LDS as the GEMM's A tile
//     [plane][128 rows][32 k] (80-byte rows: conflict-free 8-byte writes and 16-byte fragment reads);
//   * fc2's weight chunk [Cout][32 k] is split the same way into the B tile (8 elements per thread);
//   * matrix stage: wave w multiplies rows 32 w .. 32 w + 31 with all Cout columns: 2 k16-steps x 6 products x Cout / 32 MFMAs per chunk.
// Accumulators start at b_2 and leave in the MFMA layout (column on the lane: 128-byte row segments of Y).
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
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int kPH = 8, kPW = 16, kPM = kPH * kPW;   // patch: 8 rows x 16 columns = 128 GEMM rows
constexpr int kKC = 32;                             // hidden channels per chunk
constexpr int kPitch = 2 * kKC + 16;                // 80 bytes per LDS row
constexpr int kAPlane = kPM * kPitch;               // 10240

// csrc/dwconv.hip::gelu_erf, operation for operation (branch-free erfc form; held to 6e-7 by tests/test_dwconv_gpu.py)
__device__ __forceinline__ float gelu_erf(float v) {
    const float a = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.39f, a, 1.f));
    float p = -0.22753699123859406f;
    p = fmaf(p, t, 0.8866638541221619f);
    p = fmaf(p, t, -0.6353994011878967f);
    p = fmaf(p, t, 0.6495586633682251f);
    p = fmaf(p, t, 0.09138870239257812f);
    p = fmaf(p, t, 0.2353251725435257f);
    const float q = p * t * __builtin_amdgcn_exp2f(a * a * -1.4426950408889634f);
    return 0.5f * v * (v < 0.f ? q : 2.f - q);
}

// x = hi + mid + lo exactly-ish (hi = rn(x), mid = rn(x - hi), lo = rn(x - hi - mid)): token_gemm.hip's split, two elements at a time
__device__ __forceinline__ void split2(float x0, float x1, bf16x2 &h, bf16x2 &m, bf16x2 &l) {
    const f32x2 v = {x0, x1};
    h = __builtin_convertvector(v, bf16x2);
    const f32x2 r1 = v - __builtin_convertvector(h, f32x2);
    m = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
    l = __builtin_convertvector(r2, bf16x2);
}

// a 16-byte load the compiler neither sinks nor reorders (wave-uniform base in scalar registers + a 32-bit byte offset per lane), waited for by the
// explicit s_waitcnt of wait_loads(); tools/asm_pending_audit.py checks that nothing touches the destination registers in between
// ("+v": the destination IS the loop-carried variable's register -- with "=v" the compiler defines a fresh value and copies it into the loop
// variable right behind the request, i.e. before the data has arrived)
// hazard (gfx9): a VALU instruction that WRITES an SGPR (the v_readlane that restores a spilled base pointer, a v_readfirstlane) followed by a
// vector-memory instruction that READS it needs 5 wait states; hipcc inserts them for its own instructions, not in front of inline asm -- the load
// then goes to a stale address (round 6: a memory fault in head_tail.hip as soon as a spilled pointer was involved).  Every asm load with a
// scalar operand therefore carries its own wait states (tools/asm_sgpr_hazard_scan.py checks the built code).
__device__ __forceinline__ void pinned_load16(f32x4 &v, const void *base, unsigned off) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "+v"(v) : "v"(off), "s"(base) : "memory");
}
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pinned_load8(u32x2 &v, const void *base, unsigned off) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2" : "+v"(v) : "v"(off), "s"(base) : "memory");
}
__device__ __forceinline__ void wait_loads() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

#ifdef SD_MIXFFN_TAIL_STAMPS
// diagnostic build only (tools/mixffn_tail_bench.py --stamps): s_memtime sums per phase of wave 0 of logical workgroup 0
__device__ unsigned long long g_mixffn_tail_stamps[16];
#define SD_ST(i)                                                        \
    do {                                                                \
        const unsigned long long tnow = __builtin_amdgcn_s_memtime();   \
        ph[i] += tnow - tlast;                                          \
        tlast = tnow;                                                   \
    } while (0)
#else
#define SD_ST(i)
#endif

// grid.x = B * (H / 8) * (W / 16) (XCD-remapped: consecutive logical patches of an image share one L2)
// T = float: h, Y fp32, the product in split-bf16 arithmetic (NP = 3 planes per operand, six products).  T = bf16_t (the network under bf16 autocast):
// h, Y bf16, the activated map and fc2's weight rounded to bf16 -- exactly what the two-kernel route hands the bf16 GEMM -- one plane, one product.
template <int NT, typename T>   // Cout = 32 NT
__global__ __launch_bounds__(256, NT <= 2 ? 2 : 1) void mixffn_tail_x3(const T *__restrict__ h, const float *__restrict__ dww,
                                                                       const float *__restrict__ dwb, const float *__restrict__ W2,
                                                                       const float *__restrict__ b2, T *__restrict__ Y, int H, int W, int Ch,
                                                                       int patches_x, int patches_per_img) {
    constexpr int COUT = 32 * NT;
    constexpr int kBPlane = COUT * kPitch;
    constexpr bool F32 = sizeof(T) == 4;
    constexpr int NP = F32 ? 3 : 1;
    constexpr unsigned ES = sizeof(T);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // A planes [NP][128][80 B] | B planes [NP][COUT][80 B] | taps [Ch][9] | bias [Ch]
    unsigned char *ldsA = lds, *ldsB = lds + NP * kAPlane;
    float *ldsT = reinterpret_cast<float *>(lds + NP * kAPlane + NP * kBPlane), *ldsBias = ldsT + 9 * Ch;
    const long nblk = gridDim.x, id = blockIdx.x;
    const long qd = nblk / 8, rem = nblk % 8, xcd = id % 8;
    const long L = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + id / 8;
    const int b = (int)(L / patches_per_img), pr = (int)(L % patches_per_img);
    const int y0 = (pr / patches_x) * kPH, x0 = (pr % patches_x) * kPW;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int col = lane & 31, kg = lane >> 5;

    // ---- depthwise geometry: thread -> (channel quad c4, patch row r, 4 pixels from column xs) ----
    const int c4 = t & 7, strip = t >> 3;
    const int r = strip >> 2, xs = (strip & 3) * 4;
    const int yy = y0 + r, xx = x0 + xs;
    const T *hb = h + (size_t)b * H * W * Ch;          // wave-uniform; the lane's part of the address is a 32-bit byte offset (H W Ch < 2^30)
    unsigned roff[3], coff[6];
    bool rok[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = yy + ky - 1;
        rok[ky] = iy >= 0 && iy < H;
        roff[ky] = ((unsigned)(min(max(iy, 0), H - 1) * W) * (unsigned)Ch + 4u * c4) * ES;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) coff[j] = (unsigned)min(max(xx + j - 1, 0), W - 1) * (unsigned)Ch * ES;
    const bool lok = xx >= 1, rgt = xx + 4 < W;        // halo columns inside the image row

    // ---- fc2 weight staging: thread -> (output channel n = t >> 2 (+ 64 per round), 8 k from 8 (t & 3)) ----
    const int wn = t >> 2, wk = (t & 3) * 8;
    constexpr int RD = NT > 2 ? 2 : 1;
    const unsigned w2off = ((unsigned)wn * (unsigned)Ch + wk) * 4u;

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float bv = b2[32 * nt + col];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nt][e] = bv;
    }

    // the depthwise taps and bias of ALL hidden channels are parked in LDS once (9 Ch + Ch floats): per chunk a thread reads its 36 + 4 from there
    for (int i = t; i < 9 * Ch / 4; i += 256) reinterpret_cast<f32x4 *>(ldsT)[i] = reinterpret_cast<const f32x4 *>(dww)[i];
    for (int i = t; i < Ch / 4; i += 256) reinterpret_cast<f32x4 *>(ldsBias)[i] = reinterpret_cast<const f32x4 *>(dwb)[i];

    // the global operands of one chunk: 18 input vectors and this thread's piece of fc2's weight chunk.  Requested a chunk AHEAD -- right after the
    // previous chunk's convolution has consumed these registers -- so that their latency runs under the GELU / split / LDS / matrix stages instead
    // of in front of every chunk.
    typedef typename std::conditional<F32, f32x4, u32x2>::type raw_t;      // 4 channels of one pixel
    raw_t raw[3][6];
    f32x4 w2r[2 * RD];
    {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int j = 0; j < 6; ++j) raw[ky][j] = raw_t{};
#pragma unroll
        for (int i = 0; i < 2 * RD; ++i) w2r[i] = z;
    }
    auto request = [&](int kc) {
        const unsigned cbb = (unsigned)(kc * kKC) * 4u, cbe = (unsigned)(kc * kKC) * ES;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                if constexpr (F32) pinned_load16(raw[ky][j], hb, roff[ky] + coff[j] + cbe);
                else pinned_load8(raw[ky][j], hb, roff[ky] + coff[j] + cbe);
            }
#pragma unroll
        for (int rd = 0; rd < RD; ++rd) {
            const unsigned o = w2off + (unsigned)(64 * rd) * (unsigned)Ch * 4u + cbb;
            pinned_load16(w2r[2 * rd], W2, o);
            pinned_load16(w2r[2 * rd + 1], W2, o + 16u);
        }
    };
    request(0);
    __syncthreads();                                     // taps / bias visible
    wait_loads();

    const int nchunk = Ch / kKC;
#ifdef SD_MIXFFN_TAIL_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
    const unsigned long long tstart = tlast;
#endif
    for (int kc = 0; kc < nchunk; ++kc) {
        // taps of a row outside the image are zeroed (branch-free: the halo rows are clamped, valid addresses)
        float wreg[36];
        {
            const f32x4 *tq = reinterpret_cast<const f32x4 *>(ldsT + (kc * kKC + 4 * c4) * 9);
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const f32x4 v = tq[i];
                wreg[4 * i] = v[0], wreg[4 * i + 1] = v[1], wreg[4 * i + 2] = v[2], wreg[4 * i + 3] = v[3];
            }
#pragma unroll
            for (int i = 0; i < 36; ++i) {
                const int k = i % 9;
                if (k < 3) wreg[i] = rok[0] ? wreg[i] : 0.f;
                if (k >= 6) wreg[i] = rok[2] ? wreg[i] : 0.f;
            }
        }
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(ldsBias + kc * kKC + 4 * c4);
        // -- depthwise 3x3 + bias (dwconv.hip's order: ky, kx, pixel, channel) --
        float a[4][4];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i) a[p][i] = bv[i];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            f32x4 cv[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                if constexpr (F32) cv[j] = raw[ky][j];
                else {
                    const u32x2 v = raw[ky][j];
                    cv[j] = f32x4{__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16),
                                  __uint_as_float(v[1] & 0xffff0000u)};
                }
            }
            if (!lok) cv[0] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!rgt) cv[5] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int k = 3 * ky + kx;
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int i = 0; i < 4; ++i) a[p][i] = fmaf(wreg[9 * i + k], cv[p + kx][i], a[p][i]);
            }
        }
#ifdef SD_MIXFFN_TAIL_STAMPS
        asm volatile("" ::"v"(a[0][0]), "v"(a[3][3]));
#endif
        SD_ST(0);                                        // taps from LDS + convolution
        // -- fc2 weight chunk: split, straight into the B tile (the barrier at the end of the previous iteration freed both tiles) --
#pragma unroll
        for (int rd = 0; rd < RD; ++rd) {
            const f32x4 u = w2r[2 * rd], v = w2r[2 * rd + 1];
            bf16x2 hp[4], mp[4], lp[4];
            split2(u[0], u[1], hp[0], mp[0], lp[0]);
            split2(u[2], u[3], hp[1], mp[1], lp[1]);
            split2(v[0], v[1], hp[2], mp[2], lp[2]);
            split2(v[2], v[3], hp[3], mp[3], lp[3]);
            bf16x8 bp[3];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bp[0][2 * q] = hp[q][0], bp[0][2 * q + 1] = hp[q][1];
                bp[1][2 * q] = mp[q][0], bp[1][2 * q + 1] = mp[q][1];
                bp[2][2 * q] = lp[q][0], bp[2][2 * q + 1] = lp[q][1];
            }
            if (wn + 64 * rd < COUT) {
                unsigned char *q = ldsB + (wn + 64 * rd) * kPitch + 2 * wk;
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<bf16x8 *>(q + pl * kBPlane) = bp[pl];
            }
        }
        // every register of this chunk's request has been consumed: the next chunk's goes out now
        __builtin_amdgcn_sched_barrier(0);
        SD_ST(1);                                        // weight chunk split + LDS stores
        request(kc + 1 < nchunk ? kc + 1 : kc);          // (past the last chunk: a repeat, never used -- the loop body stays branch-free)
        __builtin_amdgcn_sched_barrier(0);
        SD_ST(2);                                        // requests issued
        // -- GELU, split, A tile: pixel by pixel --
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float g[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) g[i] = gelu_erf(a[p][i]);
            bf16x2 h0, m0, l0, h1, m1, l1;
            split2(g[0], g[1], h0, m0, l0);
            split2(g[2], g[3], h1, m1, l1);
            const bf16x4 hh = {h0[0], h0[1], h1[0], h1[1]}, mm = {m0[0], m0[1], m1[0], m1[1]}, ll = {l0[0], l0[1], l1[0], l1[1]};
            unsigned char *q = ldsA + (16 * r + xs + p) * kPitch + 8 * c4;
            *reinterpret_cast<uint2 *>(q) = __builtin_bit_cast(uint2, hh);
            if constexpr (F32) {
                *reinterpret_cast<uint2 *>(q + kAPlane) = __builtin_bit_cast(uint2, mm);
                *reinterpret_cast<uint2 *>(q + 2 * kAPlane) = __builtin_bit_cast(uint2, ll);
            }
        }
        SD_ST(3);                                        // GELU + split + LDS stores
        __syncthreads();
        SD_ST(4);                                        // barrier 1
        // -- matrix stage --
#pragma unroll
        for (int s = 0; s < kKC / 16; ++s) {
            const unsigned char *qa = ldsA + (32 * wave + col) * kPitch + 32 * s + 16 * kg;
            const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(qa);
            bf16x8 am = ah, al = ah;
            if constexpr (F32) am = *reinterpret_cast<const bf16x8 *>(qa + kAPlane), al = *reinterpret_cast<const bf16x8 *>(qa + 2 * kAPlane);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const unsigned char *qb = ldsB + (32 * nt + col) * kPitch + 32 * s + 16 * kg;
                const bf16x8 bh = *reinterpret_cast<const bf16x8 *>(qb);
                f32x16 c = acc[nt];
                if constexpr (F32) {
                    const bf16x8 bm = *reinterpret_cast<const bf16x8 *>(qb + kBPlane), bl = *reinterpret_cast<const bf16x8 *>(qb + 2 * kBPlane);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);      // small terms first (token_gemm.hip's order)
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
                acc[nt] = c;
            }
        }
#ifdef SD_MIXFFN_TAIL_STAMPS
        asm volatile("" ::"v"(acc[0][0]), "v"(acc[NT - 1][15]));
#endif
        SD_ST(5);                                        // fragment reads + MFMAs
        // the wait sits HERE, not at the loop top: whatever copies the compiler makes of the loop-carried request registers at the back edge then read
        // arrived data (tools/asm_pending_audit.py)
        wait_loads();
        SD_ST(6);                                        // wait for the next chunk's operands
        __syncthreads();                                 // this chunk's fragment reads are done: the tiles may be overwritten
        SD_ST(7);                                        // barrier 2
    }
#ifdef SD_MIXFFN_TAIL_STAMPS
    if (L == 0 && t == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) g_mixffn_tail_stamps[i] = ph[i];
        g_mixffn_tail_stamps[8] = __builtin_amdgcn_s_memtime() - tstart;
        g_mixffn_tail_stamps[9] = (unsigned long long)nchunk;
    }
#endif
    // ---- epilogue: D row m = 32 wave + (e & 3) + 8 (e >> 2) + 4 kg -> patch pixel (m >> 4, m & 15); column 32 nt + col ----
    T *Yb = Y + (size_t)b * H * W * COUT;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = 32 * wave + (e & 3) + 8 * (e >> 2) + 4 * kg;
        T *yo = Yb + ((size_t)(y0 + (m >> 4)) * W + x0 + (m & 15)) * COUT + col;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if constexpr (F32) yo[32 * nt] = acc[nt][e];
            else yo[32 * nt].bits = f32_to_bf16(acc[nt][e]);
        }
    }
}

template <int NT, typename T>
int launch_tail(const void *h, const float *dww, const float *dwb, const float *W2, const float *b2, void *Y, int B, int H, int W, int Ch,
                hipStream_t st) {
    constexpr int NP = sizeof(T) == 4 ? 3 : 1;
    const int px = W / kPW, ppi = px * (H / kPH);
    const size_t ldsb = (size_t)NP * kAPlane + (size_t)NP * 32 * NT * kPitch + (size_t)40 * Ch;
    if (ldsb > 64 * 1024) {
        static bool raised = false;
        if (!raised) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&mixffn_tail_x3<NT, T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            raised = true;
        }
    }
    if (ldsb > 160 * 1024) return SD_E_UNSUPPORTED;
    hipLaunchKernelGGL((mixffn_tail_x3<NT, T>), dim3((unsigned)((long)B * ppi)), dim3(256), ldsb, st, (const T *)h, dww, dwb, W2, b2, (T *)Y, H, W, Ch, px,
                       ppi);
    return (int)hipGetLastError();
}

}  // namespace
}  // namespace sd

extern "C" {

#ifdef SD_MIXFFN_TAIL_STAMPS
int sd_debug_mixffn_tail_stamps(unsigned long long *out16) {       // diagnostic build only (not in the header)
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(sd::g_mixffn_tail_stamps), 16 * sizeof(unsigned long long));
}
#endif

int sd_mixffn_tail_supported(int H, int W, int hidden, int out_features) {
    return (H > 0 && W > 0 && H % sd::kPH == 0 && W % sd::kPW == 0 && hidden % sd::kKC == 0 && (out_features == 64 || out_features == 128) &&
            (long)H * W * hidden < (1L << 30))
               ? 1
               : 0;
}

int sd_mixffn_tail(const void *h, const float *dw_weight, const float *dw_bias, const float *fc2_weight, const float *fc2_bias, void *y, int dtype, int B,
                   int H, int W, int hidden, int out_features, void *stream) {
    if (!h || !dw_weight || !dw_bias || !fc2_weight || !fc2_bias || !y) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (B <= 0 || H <= 0 || W <= 0 || hidden <= 0 || out_features <= 0) return SD_E_SHAPE;
    if (!sd_mixffn_tail_supported(H, W, hidden, out_features)) return SD_E_UNSUPPORTED;
    if ((long)B * (H / sd::kPH) * (W / sd::kPW) > 0x7fffffffL) return SD_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(dw_weight) | reinterpret_cast<uintptr_t>(dw_bias) |
         reinterpret_cast<uintptr_t>(fc2_weight) | reinterpret_cast<uintptr_t>(fc2_bias) | reinterpret_cast<uintptr_t>(y)) & 15)
        return SD_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32)
        return out_features == 64 ? sd::launch_tail<2, float>(h, dw_weight, dw_bias, fc2_weight, fc2_bias, y, B, H, W, hidden, st)
                                  : sd::launch_tail<4, float>(h, dw_weight, dw_bias, fc2_weight, fc2_bias, y, B, H, W, hidden, st);
    return out_features == 64 ? sd::launch_tail<2, sd::bf16_t>(h, dw_weight, dw_bias, fc2_weight, fc2_bias, y, B, H, W, hidden, st)
                              : sd::launch_tail<4, sd::bf16_t>(h, dw_weight, dw_bias, fc2_weight, fc2_bias, y, B, H, W, hidden, st);
}

}  // extern "C"
