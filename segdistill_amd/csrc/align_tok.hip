// align_tok.hip -- the feature-align projection of a token-major tap FUSED with the channel-group criterion it feeds, bf16 storage, gfx950.
//
// reference: the 1x1 projection of the student feature described at mmseg/models/distillation/opts.py:25-27 and built in the commented generation
// of losses.py:258,332-333,373-374 (`self.ff = nn.Conv2d(**ff_config, kernel_size=1)`), followed by KLDLoss.forward (losses.py:95-113) on its output.
// On a token-major tap [B, P, Cs] (decode_head.linear_c1..4, BASELINE config 5 / SURVEY a-15 + a-16) the projection is the GEMM
//     Y [tokens][Ct] = X [tokens][Cs] . W [Ct][Cs]^T + bias,          Cs = 256, Ct = 768, 174 080 tokens per 8-image step,
// and Y is read by nobody but the criterion.  Rounds 2-4 ran it as a library GEMM that wrote Y (201 MB at stage 1), the criterion forward read Y and T,
// the criterion backward read Y and T again and wrote dY, and three more GEMMs read dY.  Here Y never exists in memory:
//   forward   one launch for all stages: a workgroup owns 256 tokens (4 waves x 64), keeps their X rows as MFMA A-fragments IN REGISTERS for the whole
//             item (K <= 256), streams W in blocks of 32 output channels (LDS-DMA, double-buffered; W is L2-resident) and the matching [256 x 32] block
//             of the teacher tap T; the 32x32x16 bf16 MFMA leaves D[token][channel] with the channel on the lane and 16 tokens in its registers, the
//             T block is read back from LDS in exactly that layout by `ds_read_b64_tr_b16` (a 4-token x 16-channel block delivered channel-major),
//             and every lane folds its 32 (s, t) pairs into one online-softmax record of its channel.  The four waves' records are merged through
//             LDS by one wave while the next block is being multiplied; out goes ONE 20-byte record per (image, 256-token tile, channel) -- the
//             records cgd_tok.hip's finish launch folds into row statistics and the loss.  HBM: X + T once (268 MB at stage 1 instead of 938).
//   backward  the same tile loop recomputes the Y block (bit-identical: same MFMA order), forms dY = k (softmax_row(Y) - softmax_row(T)) in the
//             accumulator registers, transposes it through the wave's own (already consumed) T rows in LDS and stores dY token-major in bf16 with
//             16-byte stores -- the operand of the input-gradient GEMM below and of the weight-gradient GEMM of wgrad_tn.hip; the bias gradient's
//             column sums ride along (one fp32 row per tile).  HBM: X + T + dY (469 MB instead of 603 + 201).
//   plain     the same loop with the projection's output stored (bf16): the stand-alone forward Y = X.W^T + b for taps whose criterion has no token form.
//   bwd-data  dX [tokens][Cs] = dY [tokens][Ct] . W [Ct][Cs]: sd_linear_tok_bf16_nt below (a k-loop GEMM: both operands through an LDS-DMA ring).
// The aligned feature is kept in fp32 between the MFMA and the softmax (it is never rounded to bf16 because it is never stored); dY is rounded once.
// Algorithmic bytes per element of T (e = 2): forward 2 e (T) + X, backward 2 e (T, dY) + X.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cgd_tok_device.h"

namespace sd {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int kBM = 256;            // tokens per item: 4 waves x 64 (two 32-row MFMA blocks per wave)
constexpr int kBN = 32;             // output channels per block
constexpr int kMaxRange = 768;      // channels of one item (the bias / row-constant tables in LDS); wider projections are split over items
constexpr int kTBytes = kBM * kBN * 2;

enum { MODE_FWD = 0, MODE_BWD = 1, MODE_PLAIN = 2 };

// global -> LDS, 16 bytes per lane, no staging registers.  M0 = wave-uniform LDS byte address of lane 0's 16 bytes; lane l lands at M0 + 16 l.
// Issued from inline asm so that hipcc does not put `s_waitcnt vmcnt(0)` in front of every LDS read that follows (wgrad_tn.hip); the waits are
// written by hand below.  M0 is reserved in LLVM and nothing else in these kernels uses it.
// Address = wave-uniform base (SGPR pair) + per-lane unsigned 32-bit byte offset: one VGPR per DMA instruction instead of a 64-bit pointer.
__device__ __forceinline__ void dma16(const void *base, unsigned lane_off, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(base), "s"(lds_byte) : "memory");
}

struct AlignTokTable {
    const bf16_t *X[kTokMaxJobs];
    const bf16_t *W[kTokMaxJobs];
    const float *bias[kTokMaxJobs];
    const bf16_t *T[kTokMaxJobs];
    // forward
    RowPart *part[kTokMaxJobs];
    // backward / plain
    const int32_t *perm[kTokMaxJobs];
    const float *row_lse2[kTokMaxJobs];
    const float *upstream[kTokMaxJobs];
    bf16_t *out[kTokMaxJobs];          // dY (backward) / Y (plain)
    float *db_part[kTokMaxJobs];       // [B * nkb][C] column sums of dY per tile, or NULL
    long P[kTokMaxJobs];
    int C[kTokMaxJobs], g[kTokMaxJobs], G[kTokMaxJobs], nkb[kTokMaxJobs], csplit[kTokMaxJobs];
    float c2[kTokMaxJobs], coef[kTokMaxJobs];
    int blk_begin[kTokMaxJobs + 1];
    int njobs;
};

template <int KS, int MODE>
struct AlignLds {
    static constexpr int K = 16 * KS;
    static constexpr int kWBytes = kBN * K * 2;
    static constexpr int kStage = kWBytes + kTBytes;
    static constexpr int kBias = 2 * kStage;                                       // float[kMaxRange]
    static constexpr int kLse = kBias + kMaxRange * 4;                             // float[kMaxRange][2] (backward)
    static constexpr int kSlots = kLse + (MODE == MODE_BWD ? kMaxRange * 8 : 0);   // forward: float[2][4][5][32]; backward / plain: float[2][4][32]
    static constexpr int kTotal = kSlots + (MODE == MODE_FWD ? 2 * 4 * 5 * 32 * 4 : 2 * 4 * 32 * 4);
};

// One item = (job, image b, 256-token tile kb, channel range cs of csplit).  Workgroups: 256 threads, two per CU (<= 80 KB of LDS, <= 256 registers).
template <int KS, int MODE>
__global__ __launch_bounds__(256, 2) void align_tok_kernel(const AlignTokTable tab, unsigned *__restrict__ counters, int ncounters) {
    typedef AlignLds<KS, MODE> L;
    constexpr int K = L::K;
    constexpr int kChunks = K / 8;                          // 16-byte chunks per W row
    constexpr int SW = kChunks >= 16 ? 15 : kChunks - 1;    // XOR swizzle of the chunk index by the row (conflict-free ds_read_b128 of 32 rows)
    constexpr int kWDma = KS / 4;                           // W-block DMA instructions per wave (1 KB each)
    static_assert(KS == 4 || KS == 8 || KS == 16, "K in {64, 128, 256}");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[L::kTotal];

    if (MODE == MODE_FWD && counters && blockIdx.x == 0 && threadIdx.x < (unsigned)ncounters) counters[threadIdx.x] = 0u;   // the finish launch's tickets

    const int j = tok_find_job(tab, (int)blockIdx.x);
    const int lb = (int)blockIdx.x - tab.blk_begin[j];
    const int C = tab.C[j], nkb = tab.nkb[j], csplit = tab.csplit[j];
    const long P = tab.P[j];
    const int cs = lb % csplit, tile = lb / csplit;          // tile = b * nkb + kb
    const int b = tile / nkb, kb = tile - b * nkb;
    const int crange = C / csplit, c_lo = cs * crange;       // C % (32 csplit) == 0 (the launcher checks)
    const int nblk = crange / kBN;
    const float c2 = tab.c2[j];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int h = lane >> 5, col = lane & 31, g16 = (lane >> 4) & 1, li = lane & 15;

    const long img0 = (long)b * P;                           // first token of the image
    const long tok0 = img0 + (long)kb * kBM;                 // first token of the tile
    const long last_tok = img0 + P - 1;
    const int rows_tile = (int)min((long)kBM, P - (long)kb * kBM);
    const int wrows = min(64, max(0, rows_tile - 64 * wave));    // valid rows of this wave (wave-uniform)

    // ---- A fragments: this wave's 64 token rows, all of K, straight from global in the operand layout (lane l: row l & 31, k = 16 s + 8 (l >> 5) ..+7)
    const bf16_t *X = tab.X[j];
    bf16x8 af[2][KS];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long tok = min(tok0 + 64 * wave + 32 * i + col, last_tok);         // rows past the image: clamped address, masked below
        const bf16_t *px = X + (size_t)tok * K + 8 * h;
#pragma unroll
        for (int s = 0; s < KS; ++s) af[i][s] = *reinterpret_cast<const bf16x8 *>(px + 16 * s);
    }

    // ---- per-channel tables of the item's range
    float *bias_l = reinterpret_cast<float *>(lds + L::kBias);
    {
        const float *bias = tab.bias[j];
        for (int c = t; c < crange; c += 256) bias_l[c] = bias ? bias[c_lo + c] : 0.f;
    }
    float kk = 0.f;
    if constexpr (MODE == MODE_BWD) {
        float *lse_l = reinterpret_cast<float *>(lds + L::kLse);
        const int32_t *perm = tab.perm[j];
        const float *row_lse2 = tab.row_lse2[j];
        const int g = tab.g[j], G = tab.G[j];
        for (int s = t; s < C; s += 256) {                    // slot s holds channel perm[s]; its row is (b, s / g)
            const int c = (perm ? perm[s] : s) - c_lo;
            if (c >= 0 && c < crange) {
                const int row = b * G + s / g;
                lse_l[2 * c] = row_lse2[2 * row];
                lse_l[2 * c + 1] = row_lse2[2 * row + 1];
            }
        }
        const float *up = tab.upstream[j];
        kk = up ? tab.coef[j] * up[0] : tab.coef[j];
    }

    // ---- DMA geometry (lane constants).  W block: 32 rows of 2K bytes, LDS chunk p of row r holds source chunk p ^ (r & SW); wave w moves the
    // block's bytes [w kWDma KB, (w + 1) kWDma KB).  T block: 256 rows of 64 bytes, the wave moves ITS OWN 64 rows (only it reads them).
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    const bf16_t *wbase = tab.W[j] + (size_t)c_lo * K;                  // + nb * 32 rows
    unsigned woff[kWDma];
#pragma unroll
    for (int u = 0; u < kWDma; ++u) {
        const int o = (wave * kWDma + u) * 1024 + 16 * lane;
        const int r = o / (2 * K), p = (o % (2 * K)) >> 4;
        woff[u] = (unsigned)(r * 2 * K + 16 * (p ^ (r & SW)));
    }
    const bf16_t *tbase = MODE != MODE_PLAIN ? tab.T[j] + (size_t)tok0 * C + c_lo : nullptr;     // + nb * 32 channels
    unsigned toff[4];
    if constexpr (MODE != MODE_PLAIN) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long tok = min(tok0 + 64 * wave + 16 * u + (lane >> 2), last_tok);        // >= tok0: the tile has at least one valid row
            toff[u] = (unsigned)((tok - tok0) * C * 2 + 16 * (lane & 3));
        }
    }
    auto issue = [&](int nb) {                                // block nb -> stage nb & 1
        const unsigned st = lds0 + (unsigned)(nb & 1) * (unsigned)L::kStage;
        const bf16_t *wb_ = wbase + (size_t)nb * kBN * K;
#pragma unroll
        for (int u = 0; u < kWDma; ++u) dma16(wb_, woff[u], __builtin_amdgcn_readfirstlane(st + (unsigned)((wave * kWDma + u) * 1024)));
        if constexpr (MODE != MODE_PLAIN) {
            const bf16_t *tb_ = tbase + nb * kBN;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                dma16(tb_, toff[u], __builtin_amdgcn_readfirstlane(st + (unsigned)L::kWBytes + (unsigned)((64 * wave + 16 * u) * 64)));
        }
    };

    const unsigned wbase_lane = (unsigned)(col * 2 * K + ((h ^ (col & SW)) << 4));   // B-fragment address of k-step s: this ^ (s << 5)

    // merge of the four waves' results of block nb (by wave nb & 3, one iteration later: behind the barrier that ends block nb)
    float *slots = reinterpret_cast<float *>(lds + L::kSlots);
    auto merge_block = [&](int nb) {
        if ((nb & 3) != wave || lane >= 32) return;
        const int c = c_lo + nb * kBN + lane;
        if constexpr (MODE == MODE_FWD) {
            const float *q = slots + (size_t)(nb & 1) * 4 * 5 * 32 + lane;
            RowPart acc = {q[0], q[32], q[64], q[96], q[128]};
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float *qw = q + w * 5 * 32;
                merge(acc, RowPart{qw[0], qw[32], qw[64], qw[96], qw[128]}, c2);
            }
            tab.part[j][(size_t)tile * C + c] = acc;
        } else {
            float *dbp = tab.db_part[j];
            if (dbp) {
                const float *q = slots + (size_t)(nb & 1) * 4 * 32 + lane;
                dbp[(size_t)tile * C + c] = ((q[0] + q[32]) + q[64]) + q[96];
            }
        }
    };

    issue(0);
    for (int nb = 0; nb < nblk; ++nb) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // block nb has landed (this wave's part); also the prologue's table stores
        __syncthreads();                                     // ... everyone's part; and nobody reads stage (nb + 1) & 1 any more
        if (nb + 1 < nblk) issue(nb + 1);
        if (nb > 0) merge_block(nb - 1);
        unsigned char *stt = lds + (size_t)(nb & 1) * L::kStage + L::kWBytes + (size_t)wave * 64 * 64;    // this wave's 64 T rows

        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        // B fragments: row `col` of the W block, source chunk 2 s + h, i.e. LDS chunk (2 s + h) ^ (col & SW): byte (wb ^ (s << 5)) of the LDS array.
        // Read four k-steps ahead (two register sets); `wb` is made opaque so that the sixteen addresses are not hoisted out of the block loop.
        unsigned wb = wbase_lane + (unsigned)(nb & 1) * (unsigned)L::kStage;
        asm volatile("" : "+v"(wb));
        auto bfrag = [&](int s) -> bf16x8 { return *reinterpret_cast<const bf16x8 *>(lds + (wb ^ (unsigned)(s << 5))); };
        constexpr int GS = 4, NG = KS / GS;
        bf16x8 bq[2][GS];
#pragma unroll
        for (int u = 0; u < GS; ++u) bq[0][u] = bfrag(u);
        s16x4 tpk[2][4];
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            if (gi + 1 < NG) {
#pragma unroll
                for (int u = 0; u < GS; ++u) bq[(gi + 1) & 1][u] = bfrag(GS * (gi + 1) + u);
            } else if constexpr (MODE != MODE_PLAIN) {
                // the T block in the accumulator layout: block (i, q) = token rows 32 i + 8 q + 4 h .. + 3, lane = channel; element e = row e of the block
                typedef s16x4 __attribute__((address_space(3))) * lds_p;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        tpk[i][q] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(stt + (32 * i + 8 * q + 4 * h + (li >> 2)) * 64 + 32 * g16 + 8 * (li & 3)));
            }
#pragma unroll
            for (int u = 0; u < GS; ++u) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][GS * gi + u], bq[gi & 1][u], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][GS * gi + u], bq[gi & 1][u], acc[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const int cl = nb * kBN + col;                        // channel within the item's range
        const float bias_c = bias_l[cl];
        auto row_of = [&](int i, int e) -> int { return 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h; };      // row within the wave's 64
        if constexpr (MODE != MODE_PLAIN) {
            if (wrows != 64) {                                // ragged tile (wave-uniform): rows >= wrows become -1e30 on both sides and drop out of every sum
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (row_of(i, e) >= wrows) {
                            acc[i][e] = kNegBig;
                            tpk[i][e >> 2][e & 3] = (short)0xF149;           // bf16(-1e30)
                        }
            }
        }
        auto tval = [&](const s16x4 (&tp)[2][4], int i, int e) -> float { return __uint_as_float((unsigned)(unsigned short)tp[i][e >> 2][e & 3] << 16); };

        if constexpr (MODE == MODE_FWD) {
            // pass 1: the maxima.  s = acc + bias, so max(s) = max(acc) + bias and (s - max s) c2 = acc c2 + os
            float mxa = acc[0][0], mxt = tval(tpk, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    mxa = fmaxf(mxa, acc[i][e]);
                    mxt = fmaxf(mxt, tval(tpk, i, e));
                }
            // pass 2 widens the packed T values again instead of keeping 32 floats alive across pass 1 (the copy below is opaque to CSE)
            s16x4 tp2[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    tp2[i][q] = tpk[i][q];
                    asm volatile("" : "+v"(tp2[i][q]));
                }
            const float os = -mxa * c2, ot = -mxt * c2;
            float zs = 0.f, zt = 0.f, a = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float tv = tval(tp2, i, e);
                    zs += ex2(fmaf(acc[i][e], c2, os));
                    const float et = ex2(fmaf(tv, c2, ot));
                    zt += et;
                    a = fmaf(et, tv - acc[i][e], a);          // sum e_t (t - acc); the bias comes off once below
                }
            RowPart p = {mxa + bias_c, zs, mxt, zt, fmaf(-bias_c, zt, a)};
            if (wrows != 64 && 4 * h >= wrows) p = {kNegBig, 0.f, kNegBig, 0.f, 0.f};      // a lane none of whose rows exist: the identity record
            // the other half of the channel's rows sits in lane ^ 32
            RowPart o = {__shfl_xor(p.ms, 32, 64), __shfl_xor(p.zs, 32, 64), __shfl_xor(p.mt, 32, 64), __shfl_xor(p.zt, 32, 64), __shfl_xor(p.a, 32, 64)};
            if (h) {                                          // both halves fold (rows of h = 0, rows of h = 1) in that order
                const RowPart tmp = p;
                p = o;
                o = tmp;
            }
            merge(p, o, c2);
            if (lane < 32) {
                float *q = slots + ((size_t)(nb & 1) * 4 + wave) * 5 * 32 + lane;
                q[0] = p.ms; q[32] = p.zs; q[64] = p.mt; q[96] = p.zt; q[128] = p.a;
            }
        } else {
            // dY (backward) / Y (plain) of the block: 32 values per lane, rounded to bf16 in token quadruples, transposed through the wave's T rows
            float ls = 0.f, lt = 0.f;
            if constexpr (MODE == MODE_BWD) {
                const float *lse_l = reinterpret_cast<const float *>(lds + L::kLse);
                ls = lse_l[2 * cl];
                lt = lse_l[2 * cl + 1];
            }
            const float osb = fmaf(bias_c, c2, -ls);          // s c2 - ls = acc c2 + (bias c2 - ls)
            float dsum = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float d[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * q + e;
                        if constexpr (MODE == MODE_BWD) {
                            d[e] = kk * (ex2(fmaf(acc[i][r], c2, osb)) - ex2(fmaf(tval(tpk, i, r), c2, -lt)));      // masked rows: 0 - 0
                            dsum += d[e];
                        } else {
                            d[e] = acc[i][r] + bias_c;
                        }
                    }
                    const bf16x2 lo = __builtin_convertvector((f32x2){d[0], d[1]}, bf16x2), hi = __builtin_convertvector((f32x2){d[2], d[3]}, bf16x2);
                    const u32x2 v = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
                    // image [32 channels][64 tokens] of 128-byte rows in the wave's own 4 KB; 8-byte unit u = tokens 4u .. 4u+3 stored at unit u ^ (channel & 15)
                    const int u = 8 * i + 2 * q + h;
                    *reinterpret_cast<u32x2 *>(stt + col * 128 + 8 * (u ^ (col & 15))) = v;
                }
            if constexpr (MODE == MODE_BWD) {
                dsum += __shfl_xor(dsum, 32, 64);
                if (lane < 32) slots[((size_t)(nb & 1) * 4 + wave) * 32 + lane] = dsum;
            }
            // read back token-major: lane -> token 16 jj + li, channels 8 gg .. 8 gg + 7 (gg = lane >> 4): two transposed reads, one 16-byte store
            bf16_t *out = tab.out[j];
            const int gg = lane >> 4;
            typedef s16x4 __attribute__((address_space(3))) * lds_p;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int ca = 8 * gg + (li >> 2), cb = ca + 4, u = 4 * jj + (li & 3);
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(stt + ca * 128 + 8 * (u ^ (ca & 15))));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(stt + cb * 128 + 8 * (u ^ (cb & 15))));
                const int row = 16 * jj + li;
                if (row < wrows) {
                    typedef short s16x8 __attribute__((ext_vector_type(8)));
                    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    *reinterpret_cast<s16x8 *>(out + (size_t)(tok0 + 64 * wave + row) * C + c_lo + nb * kBN + 8 * gg) = v;
                }
            }
        }
    }
    if constexpr (MODE != MODE_PLAIN) {
        __syncthreads();
        merge_block(nblk - 1);
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------------------
inline int tiles_per_image(long P) { return (int)((P + kBM - 1) / kBM); }

bool shape_ok(int K, int C) { return (K == 64 || K == 128 || K == 256) && C > 0 && C % kBN == 0; }

// How many channel ranges every item of the SMALL jobs is cut into.  Items are dealt to ~2 workgroup slots per CU; a launch whose last round is mostly
// empty (config 5: 512 + 128 + 32 + 8 tiles on 512 slots) finishes sooner when the tail's items are cut into csplit channel ranges each (the X rows are
// then loaded csplit times: 1/4 of an item's bytes).  Jobs are ordered by size; the cut applies from the first job at which everything that is left fits
// into 60 % of a round.
struct Plan {
    int order[kTokMaxJobs];
    int csplit[kTokMaxJobs];
};

Plan make_plan(const long *ntile, const int *C, int njobs) {
    Plan pl;
    for (int i = 0; i < njobs; ++i) pl.order[i] = i;
    for (int i = 1; i < njobs; ++i)                                   // insertion sort, descending tile count (stable)
        for (int k = i; k > 0 && ntile[pl.order[k]] > ntile[pl.order[k - 1]]; --k) {
            const int tmp = pl.order[k];
            pl.order[k] = pl.order[k - 1];
            pl.order[k - 1] = tmp;
        }
    constexpr long kSlots = 512;
    long left = 0;
    for (int i = 0; i < njobs; ++i) left += ntile[i];
    long cum = 0;
    int d_tail = 0;                                                    // 0: not in the tail yet
    for (int oi = 0; oi < njobs; ++oi) {
        const int i = pl.order[oi];
        if (!d_tail) {
            const long rem = cum % kSlots, free_slots = rem ? kSlots - rem : kSlots;
            if (left * 10 <= free_slots * 6) {
                d_tail = (int)(free_slots / (left > 0 ? left : 1));
                if (d_tail > 4) d_tail = 4;
            }
        }
        int d = d_tail ? d_tail : 1;
        const int nb = C[i] / kBN;
        while (d > 1 && nb % d) --d;
        int dmin = (C[i] + kMaxRange - 1) / kMaxRange;                  // a range must fit the LDS tables
        while (nb % dmin) ++dmin;
        if (d < dmin) d = dmin;
        pl.csplit[i] = d;
        cum += ntile[i] * d;
        left -= ntile[i];
    }
    return pl;
}

template <int MODE>
int launch_tab(const AlignTokTable &tab, int K, unsigned *tickets, hipStream_t st) {
    const unsigned nblk = (unsigned)tab.blk_begin[tab.njobs];
    if (K == 256) hipLaunchKernelGGL((align_tok_kernel<16, MODE>), dim3(nblk), dim3(256), 0, st, tab, tickets, kTokMaxJobs);
    else if (K == 128) hipLaunchKernelGGL((align_tok_kernel<8, MODE>), dim3(nblk), dim3(256), 0, st, tab, tickets, kTokMaxJobs);
    else hipLaunchKernelGGL((align_tok_kernel<4, MODE>), dim3(nblk), dim3(256), 0, st, tab, tickets, kTokMaxJobs);
    return (int)hipGetLastError();
}

int check_jobs(const sd_align_tok_job *jobs, int njobs, int mode) {
    if (!jobs) return SD_E_NULL;
    if (njobs <= 0 || njobs > kTokMaxJobs) return SD_E_SHAPE;
    for (int i = 0; i < njobs; ++i) {
        const sd_align_tok_job &q = jobs[i];
        if (!q.X || !q.W) return SD_E_NULL;
        if (q.B <= 0 || q.P <= 0 || q.C <= 0 || q.K <= 0 || q.B > 65535) return SD_E_SHAPE;
        if ((long)q.B * tiles_per_image(q.P) * 4 > 0x3fffffffL) return SD_E_SHAPE;
        if (!shape_ok(q.K, q.C) || q.K != jobs[0].K) return SD_E_UNSUPPORTED;      // one K per call (the A fragments' register count is compiled in)
        uintptr_t al = reinterpret_cast<uintptr_t>(q.X) | reinterpret_cast<uintptr_t>(q.W);
        if (mode != MODE_PLAIN) {
            if (!q.T || !q.row_lse2 || q.g <= 0) return q.g <= 0 ? SD_E_SHAPE : SD_E_NULL;
            al |= reinterpret_cast<uintptr_t>(q.T);
            if (q.perm && q.C > 2048) return SD_E_UNSUPPORTED;
        }
        if (mode == MODE_FWD && (!q.row_kl || !q.loss || !q.workspace)) return SD_E_NULL;
        if (mode != MODE_FWD) {
            if (!q.out) return SD_E_NULL;
            al |= reinterpret_cast<uintptr_t>(q.out);
        }
        if (al & 15) return SD_E_ALIGN;
    }
    return SD_OK;
}

void fill_common(AlignTokTable &tab, const sd_align_tok_job *jobs, int njobs, int *order = nullptr) {
    long ntile[kTokMaxJobs];
    int Cs[kTokMaxJobs];
    for (int i = 0; i < njobs; ++i) {
        ntile[i] = (long)jobs[i].B * tiles_per_image(jobs[i].P);
        Cs[i] = jobs[i].C;
    }
    const Plan pl = make_plan(ntile, Cs, njobs);
    int nb = 0;
    for (int oi = 0; oi < njobs; ++oi) {
        const int i = pl.order[oi];
        if (order) order[oi] = i;
        const sd_align_tok_job &q = jobs[i];
        tab.X[oi] = static_cast<const bf16_t *>(q.X);
        tab.W[oi] = static_cast<const bf16_t *>(q.W);
        tab.bias[oi] = q.bias;
        tab.T[oi] = static_cast<const bf16_t *>(q.T);
        tab.perm[oi] = q.perm;
        tab.row_lse2[oi] = q.row_lse2;
        tab.upstream[oi] = q.upstream;
        tab.out[oi] = static_cast<bf16_t *>(q.out);
        tab.db_part[oi] = q.db_part;
        tab.P[oi] = q.P;
        tab.C[oi] = q.C;
        tab.g[oi] = q.g > 0 ? q.g : 1;
        tab.G[oi] = (q.C + tab.g[oi] - 1) / tab.g[oi];
        tab.nkb[oi] = tiles_per_image(q.P);
        tab.csplit[oi] = pl.csplit[i];
        tab.c2[oi] = q.inv_tau * 1.44269504088896340736f;
        tab.coef[oi] = q.coef;
        tab.blk_begin[oi] = nb;
        nb += (int)ntile[i] * pl.csplit[i];
        tab.blk_begin[oi + 1] = nb;
    }
    tab.njobs = njobs;
}

}  // namespace
}  // namespace sd

extern "C" {

int sd_align_cgd_tok_supported(int in_channels, int out_channels) { return sd::shape_ok(in_channels, out_channels) ? 1 : 0; }

int sd_align_cgd_tok_tiles(int B, long P) { return B > 0 && P > 0 ? B * sd::tiles_per_image(P) : 0; }

size_t sd_align_cgd_tok_workspace_bytes(int B, int C, long P) {
    if (B <= 0 || C <= 0 || P <= 0) return 0;
    return sd::tok_part_bytes(B, C, sd::tiles_per_image(P)) + sd::kTokTicketBytes;
}

int sd_align_cgd_tok_fwd_multi(const sd_align_tok_job *jobs, int njobs, void *stream) {
    int rc = sd::check_jobs(jobs, njobs, sd::MODE_FWD);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    sd::AlignTokTable tab = {};
    int order[sd::kTokMaxJobs];
    sd::fill_common(tab, jobs, njobs, order);
    sd::TokFinTable fin = {};
    int fb = 0;
    for (int i = 0; i < njobs; ++i) {                        // the finish table keeps the CALLER's job order (job 0 owns the tickets)
        const sd_align_tok_job &q = jobs[i];
        const int nkb = sd::tiles_per_image(q.P);
        const size_t need = sd::tok_part_bytes(q.B, q.C, nkb) + (i == 0 ? sd::kTokTicketBytes : 0);
        if (q.workspace_bytes < need || (reinterpret_cast<uintptr_t>(q.workspace) & 15)) return SD_E_WORKSPACE;
        const int G = (q.C + q.g - 1) / q.g;
        fin.part[i] = static_cast<const sd::RowPart *>(q.workspace);
        fin.perm[i] = q.perm; fin.row_lse2[i] = q.row_lse2; fin.row_kl[i] = q.row_kl; fin.loss[i] = q.loss;
        fin.B[i] = q.B; fin.C[i] = q.C; fin.g[i] = q.g; fin.G[i] = G; fin.nkb[i] = nkb;
        fin.c2[i] = q.inv_tau * 1.44269504088896340736f; fin.inv_tau[i] = q.inv_tau; fin.loss_scale[i] = q.loss_scale;
        fin.blk_begin[i] = fb;
        fb += (q.B * G + 3) / 4;
        fin.blk_begin[i + 1] = fb;
    }
    fin.njobs = njobs;
    for (int oi = 0; oi < njobs; ++oi) tab.part[oi] = static_cast<sd::RowPart *>(jobs[order[oi]].workspace);    // the scan table is in plan order
    const sd_align_tok_job &q0 = jobs[0];
    unsigned *tickets = reinterpret_cast<unsigned *>(static_cast<unsigned char *>(q0.workspace) + sd::tok_part_bytes(q0.B, q0.C, sd::tiles_per_image(q0.P)));
    rc = sd::launch_tab<sd::MODE_FWD>(tab, q0.K, tickets, st);
    if (rc) return rc;
    return sd::tok_finish_launch(fin, tickets, st);
}

int sd_align_cgd_tok_bwd_multi(const sd_align_tok_job *jobs, int njobs, void *stream) {
    int rc = sd::check_jobs(jobs, njobs, sd::MODE_BWD);
    if (rc) return rc;
    sd::AlignTokTable tab = {};
    sd::fill_common(tab, jobs, njobs);
    return sd::launch_tab<sd::MODE_BWD>(tab, jobs[0].K, nullptr, static_cast<hipStream_t>(stream));
}

int sd_linear_tok_bf16_fwd(const void *X, const void *W, const float *bias, void *Y, long tokens, int in_features, int out_features, void *stream) {
    sd_align_tok_job q = {};
    q.X = X; q.W = W; q.bias = bias; q.out = Y; q.P = tokens; q.B = 1; q.K = in_features; q.C = out_features; q.g = 1;
    int rc = sd::check_jobs(&q, 1, sd::MODE_PLAIN);
    if (rc) return rc;
    sd::AlignTokTable tab = {};
    sd::fill_common(tab, &q, 1);
    return sd::launch_tab<sd::MODE_PLAIN>(tab, q.K, nullptr, static_cast<hipStream_t>(stream));
}

}  // extern "C"
