// window_attn.hip -- window multi-head self-attention of a Swin block, forward only (the frozen teacher of BASELINE config 4), gfx950.
//
// Reference: mmseg/models/backbones/swin_transformer.py:119-153 (WindowAttention.forward): per window of N = Wh*Ww tokens and per head
//   attn = softmax((q * scale) k^T + relative_position_bias[head] + mask[window % nW]) ; out = attn v
// with q, k, v sliced out of ONE Linear's output [windows, N, 3, heads, D].
//
// Shape of the problem: N = 49, D = 32 -- 49 x 49 x 32 products, thousands of (window, head) pairs, 290 MB of operands per stage-1 block.  The
// framework's fused attention wants a materialised [windows, heads, N, N] additive tensor (111 MB per stage-1 block, read every block), runs a
// finiteness scan over it per call and takes permuted copies of q / k / v; the unshifted blocks fell to the three-kernel math path.  Here one
// wave owns one (window, head); q / k / v are read where the Linear left them and the output is written token-major [windows, N, heads*D], the
// layout the projection reads -- no permute copy either side; bias and mask come from small packed tables that stay in L2.
//
// History (profiles/r04_wattn_scalar_vs_mfma.txt has both, r04_pmc_wattn.json their counters): the first version kept query row i in lane i and took the wave-uniform K / V rows through
// scalar loads into SGPR operands of packed FMAs -- no LDS, no cross-lane traffic, 17-21 % of HBM at every stage: two or three scalar loads in
// flight per wave cannot feed 16 packed FMAs per 128 bytes.  v_mfma_f32_32x32x2_f32 runs at the same 64 FLOP/clk/SIMD as packed fp32 FMAs
// (exact fp32) and takes its operands from ordinary per-lane registers:
//   S^T = K Q^T  : A = K (lane l: key row l & 31 of tile a), B = Q^T (query l & 31 of tile b); MFMA step s and half-wave hh = l >> 5 stand for
//                  d = 16 hh + s, so a lane's 16 k-values are 64 contiguous bytes of its row (four 16-byte reads of the wave's LDS image);
//   accumulators : lane l holds query column i = 32 b + (l & 31), keys j = 32 a + 8 (e >> 2) + (e & 3) + 4 hh: the softmax over j is 32 registers
//                  in-lane plus one exchange with the other half-wave;
//   O^T = V^T P^T: B = P^T -- for MFMA step (a, e) and half hh the k-index IS key j(a, e, hh), i.e. the lane's own accumulator acc[a][b][e];
//                  A = V^T (lane l: d = l & 31): one 4-byte load per step, two 128-byte rows per instruction;
//   bias / mask  : packed once into the accumulator layout ([head | window][a][b][e >> 2][lane][e & 3], sd_window_attn_pack): a float4 per lane
//                  and accumulator quad; the bias INITIALISES the accumulators; padded keys (j >= 49) carry -inf in it; windows whose mask is
//                  all zero skip the mask read.
// 49 -> 64 padding costs (64 / 49)^2 = 1.7x matrix work: 128 MFMAs = 8192 cycles per (window, head) against ~6300 packed-FMA cycles with
// perfect operand delivery -- in exchange nothing waits on a scalar load: 91 us against 210 at stage 1 of Swin-B (38-40 % of HBM; the matrix
// pipe is ~50 % busy: three waves per SIMD do not fully hide each other's load and softmax phases).
#include <math.h>

#include "sd_common.h"

namespace sd {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// 16 bytes per lane from global memory into LDS without a register in between: M0 = wave-uniform LDS byte address, lane l lands at M0 + 16 l
__device__ __forceinline__ void dma16(const void *gptr, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_byte) : "memory");
}
constexpr int kPackFloats = 2 * 2 * 4 * 64 * 4;     // 64 x 64 per table

template <bool MASK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void window_attn_mfma(const float *__restrict__ qkv, const float *__restrict__ bias_p, const float *__restrict__ mask_p,
                                                         const int *__restrict__ mask_any, float *__restrict__ out, int njobs, int nW, int heads,
                                                         float scale) {
    constexpr int N = 49, D = 32;
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int job = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
    if (job >= njobs) return;                                          // whole waves only
    const int win = job / heads, head = job - win * heads;
    const int C = heads * D;
    const float *base = qkv + (size_t)win * N * 3 * C + (size_t)head * D;

    // K and Q rows come in by LDS-DMA: 64 lanes x 16 bytes = eight whole 128-byte rows per instruction (a lane fetching its own 64-byte fragment
    // straight from global memory touched 32 lines per instruction for 1 KB of use and kept the address unit busy half the kernel).  The wave's
    // private LDS image keeps the 128-byte pitch; chunk c of row p sits at position c ^ ((p >> 1) & 7), applied on the SOURCE side, so that the
    // fragment reads below (16 lanes = 16 rows, one 16-byte chunk each) spread over all 64 banks.  Rows >= 49 are not fetched (lanes off).
    __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * 2 * N * 128];
    unsigned char *mine = lds + (threadIdx.x >> 6) * (2 * N * 128);
    {
        const unsigned m0k = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)mine), m0q = m0k + N * 128;
#pragma unroll
        for (int u = 0; u < (N + 7) / 8; ++u) {
            const int p = 8 * u + (lane >> 3), c = (lane & 7) ^ ((p >> 1) & 7);
            if (p < N) {
                const float *src = base + (size_t)p * 3 * C + 4 * c;
                dma16(src + C, m0k + 1024u * u);
                dma16(src, m0q + 1024u * u);
            }
        }
    }
    // every other load of the job is issued here too, so that one memory latency covers them all: the V values of the second product, and the
    // bias, which INITIALISES the accumulators (S = bias + (q scale) k^T costs no add)
    float vf[2][16];
    const float *vbase = base + 2 * C + r;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int j = min(32 * a + 8 * (e >> 2) + (e & 3) + 4 * hh, N - 1);      // keys >= 49: probability 0
            vf[a][e] = vbase[(size_t)j * 3 * C];
        }
    const float4 *bp = reinterpret_cast<const float4 *>(bias_p + (size_t)head * kPackFloats) + lane;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 v = bp[((a * 2 + b) * 4 + g) * 64];
                acc[a][b][4 * g] = v.x, acc[a][b][4 * g + 1] = v.y, acc[a][b][4 * g + 2] = v.z, acc[a][b][4 * g + 3] = v.w;
            }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the DMA is invisible to the compiler's own counting; the image is wave-private: no barrier
    float kf[2][16], qf[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int p = min(32 * t + r, N - 1);                          // rows 49 .. 63 shadow row 48: finite values, masked / never stored
        const int f = (p >> 1) & 7;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int pos = 128 * p + 16 * ((4 * hh + u) ^ f);
            const float4 kv = *reinterpret_cast<const float4 *>(mine + pos), qv = *reinterpret_cast<const float4 *>(mine + N * 128 + pos);
            kf[t][4 * u] = kv.x, kf[t][4 * u + 1] = kv.y, kf[t][4 * u + 2] = kv.z, kf[t][4 * u + 3] = kv.w;
            qf[t][4 * u] = qv.x * scale, qf[t][4 * u + 1] = qv.y * scale, qf[t][4 * u + 2] = qv.z * scale, qf[t][4 * u + 3] = qv.w * scale;
        }
    }
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[a][s], qf[b][s], acc[a][b], 0, 0, 0);

    if (MASK && mask_any[win % nW] != 0) {                             // wave-uniform; all but the partition's last row / column of windows skip it
        const float4 *mp = reinterpret_cast<const float4 *>(mask_p + (size_t)(win % nW) * kPackFloats) + lane;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 m = mp[((a * 2 + b) * 4 + g) * 64];
                    acc[a][b][4 * g] += m.x, acc[a][b][4 * g + 1] += m.y, acc[a][b][4 * g + 2] += m.z, acc[a][b][4 * g + 3] += m.w;
                }
    }
    float inv[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        float m = -INFINITY;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) m = fmaxf(m, acc[a][b][e]);
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pv = __expf(acc[a][b][e] - m);
                acc[a][b][e] = pv;
                l += pv;
            }
        l += __shfl_xor(l, 32, 64);
        inv[b] = 1.f / l;
    }
    f32x16 o[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[b][e] = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e)
#pragma unroll
            for (int b = 0; b < 2; ++b) o[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[a][e], acc[a][b][e], o[b], 0, 0, 0);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int i = 32 * b + r;
        if (i < N) {
            float *oi = out + ((size_t)win * N + i) * C + (size_t)head * D + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4 *>(oi + 8 * g) =
                    make_float4(o[b][4 * g] * inv[b], o[b][4 * g + 1] * inv[b], o[b][4 * g + 2] * inv[b], o[b][4 * g + 3] * inv[b]);
        }
    }
}

__global__ __launch_bounds__(256) void window_attn_zero_flags(int *__restrict__ flags, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) flags[i] = 0;
}

// tables [n][49][49] in the reference's orientation ([query i][key j]) -> [n][a][b][g][lane][c] as the accumulators hold them; padded positions
// take `pad_key` (keys j >= 49) or 0 (queries i >= 49); flags[n] = 1 if the table has a nonzero entry (NULL: not wanted)
__global__ __launch_bounds__(256) void window_attn_pack(const float *__restrict__ table, float *__restrict__ packed, int *__restrict__ flags, int n,
                                                         float pad_key) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)n * kPackFloats) return;
    const int t = (int)(idx / kPackFloats), rem = (int)(idx - (long)t * kPackFloats);
    const int c = rem & 3, lane = (rem >> 2) & 63, g = (rem >> 8) & 3, b = (rem >> 10) & 1, a = rem >> 11;
    const int j = 32 * a + 8 * g + c + 4 * (lane >> 5), i = 32 * b + (lane & 31);
    float v = 0.f;
    if (j >= 49) v = pad_key;
    else if (i < 49) v = table[((size_t)t * 49 + i) * 49 + j];
    packed[idx] = v;
    if (flags && j < 49 && i < 49 && v != 0.f) flags[t] = 1;           // benign race: every writer stores 1
}

}  // namespace
}  // namespace sd

extern "C" {

int sd_window_attn_supported(int tokens_per_window, int head_dim) { return (tokens_per_window == 49 && head_dim == 32) ? 1 : 0; }

size_t sd_window_attn_packed_floats(void) { return (size_t)sd::kPackFloats; }

int sd_window_attn_pack(const float *tables, float *packed, int32_t *flags, int count, int tokens_per_window, float pad_key_value, void *stream) {
    if (!tables || !packed) return SD_E_NULL;
    if (tokens_per_window != 49) return SD_E_UNSUPPORTED;
    if (count <= 0 || (long)count * sd::kPackFloats > 0x7fffffffL * 256L) return SD_E_SHAPE;
    if (reinterpret_cast<uintptr_t>(packed) & 15) return SD_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the flags are cleared by a KERNEL, not by hipMemsetAsync: a pack that runs inside a hipGraph capture (a cache miss there) must not record
    // a memset node -- memset nodes of one graph are corruptible by a concurrent memset on another stream (engine/trainer.py::_issues_memsets)
    if (flags) hipLaunchKernelGGL(sd::window_attn_zero_flags, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, (int *)flags, count);
    const long total = (long)count * sd::kPackFloats;
    hipLaunchKernelGGL(sd::window_attn_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, tables, packed, (int *)flags, count, pad_key_value);
    return (int)hipGetLastError();
}

int sd_window_attn_fwd_packed(const void *qkv, const float *bias_packed, const float *mask_packed, const int32_t *mask_flags, void *out, int dtype,
                              long windows, int mask_windows, int heads, int tokens_per_window, int head_dim, float scale, void *stream) {
    if (!qkv || !bias_packed || !out) return SD_E_NULL;
    if (dtype != SD_F32) return SD_E_DTYPE;
    if (!sd_window_attn_supported(tokens_per_window, head_dim)) return SD_E_UNSUPPORTED;
    if (windows <= 0 || heads <= 0 || windows * heads > 0x7ffffff0L) return SD_E_SHAPE;
    if (mask_packed && (!mask_flags || mask_windows <= 0 || windows % mask_windows != 0)) return SD_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(bias_packed) |
         reinterpret_cast<uintptr_t>(mask_packed)) & 15)
        return SD_E_ALIGN;
    const int njobs = (int)(windows * heads);
    const dim3 grid((unsigned)((njobs + 3) / 4)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float *q = (const float *)qkv;
    float *o = (float *)out;
    if (mask_packed)
        hipLaunchKernelGGL((sd::window_attn_mfma<true>), grid, block, 0, st, q, bias_packed, mask_packed, (const int *)mask_flags, o, njobs, mask_windows,
                           heads, scale);
    else
        hipLaunchKernelGGL((sd::window_attn_mfma<false>), grid, block, 0, st, q, bias_packed, mask_packed, (const int *)mask_flags, o, njobs, 1, heads,
                           scale);
    return (int)hipGetLastError();
}

}  // extern "C"
