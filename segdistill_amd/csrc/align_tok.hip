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

constexpr int kNW = 8;              // waves per workgroup (512 threads, one workgroup per CU: two waves per SIMD)
constexpr int kBM = 64 * kNW;       // tokens per item: every wave owns 64 (two 32-row MFMA blocks)
constexpr int kBN = 32;             // output channels per block
constexpr int kMaxRange = 768;      // channels of one item (the bias / row-constant tables in LDS); wider projections are split over items
constexpr int kTDepth = 3;          // T blocks in flight per wave (each wave has a ring of its own: nobody else reads its token rows)
constexpr int kTSlot = 64 * kBN * 2;    // one wave's T block: 64 rows of 64 bytes

enum { MODE_FWD = 0, MODE_BWD = 1, MODE_PLAIN = 2 };

// global -> LDS, 16 bytes per lane, no staging registers.  M0 = wave-uniform LDS byte address of lane 0's 16 bytes; lane l lands at M0 + 16 l.
// Issued from inline asm so that hipcc does not put `s_waitcnt vmcnt(0)` in front of every LDS read that follows (wgrad_tn.hip); the waits are
// written by hand below.  M0 is reserved in LLVM and nothing else in these kernels uses it.
// Address = wave-uniform base (SGPR pair) + per-lane unsigned 32-bit byte offset: one VGPR per DMA instruction instead of a 64-bit pointer.
// hazard (gfx9): a VALU instruction that WRITES an SGPR (the v_readlane that restores a spilled base pointer, a v_readfirstlane) followed by a
// vector-memory instruction that READS it needs 5 wait states; hipcc inserts them for its own instructions, not in front of inline asm -- the load
// then goes to a stale address (round 6: a memory fault in head_tail.hip as soon as a spilled pointer was involved).  Every asm load with a
// scalar operand therefore carries its own wait states (tools/asm_sgpr_hazard_scan.py checks the built code).
__device__ __forceinline__ void dma16(const void *base, unsigned lane_off, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(base), "s"(lds_byte) : "memory");
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
    static constexpr int kWBytes = kBN * K * 2;                                    // one W block; two stages
    static constexpr int kTRing = 2 * kWBytes;                                     // [wave][slot][64 rows x 64 B]
    static constexpr int kBias = kTRing + kNW * kTDepth * kTSlot;                  // float[kMaxRange]
    static constexpr int kLse = kBias + kMaxRange * 4;                             // float[kMaxRange][2] (backward)
    static constexpr int kSlots = kLse + (MODE == MODE_BWD ? kMaxRange * 8 : 0);   // forward: float[2][kNW][5][32]; backward / plain: float[2][kNW][32]
    static constexpr int kTotal = kSlots + (MODE == MODE_FWD ? 2 * kNW * 5 * 32 * 4 : 2 * kNW * 32 * 4);
};

// One item = (job, image b, 512-token tile kb, channel range cs of csplit).  One 512-thread workgroup per CU (<= 160 KB of LDS, <= 256 registers).
// Per 32-channel block nb:   wait(W(nb), T(nb)) ; barrier ; [stores of block nb-1] ; issue W(nb+1), T(nb+2) ; MFMA ; epilogue
// The T DMAs are the YOUNGEST vector-memory operations of an iteration, so `s_waitcnt vmcnt(4)` at the next iteration's top retires everything
// older (W(nb+1), T(nb+1), the stores) and leaves exactly T(nb+2) in flight -- independent of how many stores a wave issued.
template <int KS, int MODE>
__global__ __launch_bounds__(64 * kNW, 2) void align_tok_kernel(const AlignTokTable tab, unsigned *__restrict__ counters, int ncounters,
                                                                     unsigned long long *__restrict__ stamps /* diagnostics, normally null */) {
    typedef AlignLds<KS, MODE> L;
    constexpr int K = L::K;
    constexpr int kChunks = K / 8;                          // 16-byte chunks per W row
    constexpr int SW = kChunks >= 16 ? 15 : kChunks - 1;    // XOR swizzle of the chunk index by the row (conflict-free ds_read_b128 of 32 rows)
    constexpr int kWInst = KS;                              // W-block DMA instructions (1 KB each): 32 rows x 2K bytes
    constexpr int kWDma = (kWInst + kNW - 1) / kNW;         // ... per wave (waves beyond the block's end re-load its last KB: same bytes, same place)
    constexpr int NT = 64 * kNW;
    static_assert(KS == 4 || KS == 8 || KS == 16, "K in {64, 128, 256}");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[L::kTotal];
    // -DSD_ALIGN_STAMPS (diagnostic build only, tools/align_tok_stamps.py): cycles per phase of the block loop, summed per wave of workgroup 0
#ifdef SD_ALIGN_STAMPS
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime(), tstart = tprev;
    auto stamp = [&](int k) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        ph[k] += now - tprev;
        tprev = now;
    };
#else
    (void)stamps;
    auto stamp = [](int) {};
#endif

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

    // ---- per-channel tables of the item's range
    float *bias_l = reinterpret_cast<float *>(lds + L::kBias);
    {
        const float *bias = tab.bias[j];
        for (int c = t; c < crange; c += NT) bias_l[c] = bias ? bias[c_lo + c] : 0.f;
    }
    float kk = 0.f;
    if constexpr (MODE == MODE_BWD) {
        float *lse_l = reinterpret_cast<float *>(lds + L::kLse);
        const int32_t *perm = tab.perm[j];
        const float *row_lse2 = tab.row_lse2[j];
        const int g = tab.g[j], G = tab.G[j];
        for (int s = t; s < C; s += NT) {                     // slot s holds channel perm[s]; its row is (b, s / g)
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

    // ---- DMA geometry (lane constants).  W block: 32 rows of 2K bytes, LDS chunk p of row r holds source chunk p ^ (r & SW); DMA instruction q
    // moves the block's bytes [q KB, (q + 1) KB), wave w issues q = w kWDma .. (clamped to the last).  T block: the wave's OWN 64 rows of 64 bytes.
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    const bf16_t *wbase = tab.W[j] + (size_t)c_lo * K;                  // + nb * 32 rows
    unsigned woff[kWDma], wdst[kWDma];
#pragma unroll
    for (int u = 0; u < kWDma; ++u) {
        const int q = min(wave * kWDma + u, kWInst - 1);
        const int o = q * 1024 + 16 * lane;
        const int r = o / (2 * K), p = (o % (2 * K)) >> 4;
        woff[u] = (unsigned)(r * 2 * K + 16 * (p ^ (r & SW)));
        wdst[u] = (unsigned)(q * 1024);
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
    const unsigned tring = lds0 + (unsigned)L::kTRing + (unsigned)wave * (unsigned)(kTDepth * kTSlot);
    auto issue_w = [&](int nb) {                              // W block nb -> stage nb & 1
        const unsigned st = lds0 + (unsigned)(nb & 1) * (unsigned)L::kWBytes;
        const bf16_t *wb_ = wbase + (size_t)nb * kBN * K;
#pragma unroll
        for (int u = 0; u < kWDma; ++u) dma16(wb_, woff[u], __builtin_amdgcn_readfirstlane(st + wdst[u]));
    };
    auto issue_t = [&](int nb) {                              // this wave's T block nb -> its ring slot nb % kTDepth; exactly 4 instructions
        if constexpr (MODE != MODE_PLAIN) {
            const unsigned st = tring + (unsigned)(nb % kTDepth) * (unsigned)kTSlot;
            const bf16_t *tb_ = tbase + nb * kBN;
#pragma unroll
            for (int u = 0; u < 4; ++u) dma16(tb_, toff[u], __builtin_amdgcn_readfirstlane(st + (unsigned)(16 * u * 64)));
        }
    };

    const unsigned wbase_lane = (unsigned)(col * 2 * K + ((h ^ (col & SW)) << 4));   // B-fragment address of k-step s: this ^ (s << 5)

    // merge of the waves' results of block nb (by wave nb % kNW, one iteration later: behind the barrier that ends block nb)
    float *slots = reinterpret_cast<float *>(lds + L::kSlots);
    auto merge_block = [&](int nb) {
        if ((nb % kNW) != wave || lane >= 32) return;
        const int c = c_lo + nb * kBN + lane;
        if constexpr (MODE == MODE_FWD) {
            const float *q = slots + (size_t)(nb & 1) * kNW * 5 * 32 + lane;
            RowPart acc = {q[0], q[32], q[64], q[96], q[128]};
#pragma unroll
            for (int w = 1; w < kNW; ++w) {
                const float *qw = q + w * 5 * 32;
                merge(acc, RowPart{qw[0], qw[32], qw[64], qw[96], qw[128]}, c2);
            }
            tab.part[j][(size_t)tile * C + c] = acc;
        } else {
            float *dbp = tab.db_part[j];
            if (dbp) {
                const float *q = slots + (size_t)(nb & 1) * kNW * 32 + lane;
                float v = q[0];
#pragma unroll
                for (int w = 1; w < kNW; ++w) v += q[w * 32];
                dbp[(size_t)tile * C + c] = v;
            }
        }
    };
    // the output block (dY / Y) a wave left in its T slot, token-major to memory: lane -> token 16 jj + li, channels 8 gg .. 8 gg + 7 (gg = lane >> 4):
    // two transposed reads, one 16-byte store.  Runs one iteration late, so that the store's completion is never waited for right behind it.
    auto store_block = [&](int nb) {
        if constexpr (MODE != MODE_FWD) {
            const unsigned char *stt = lds + L::kTRing + (size_t)wave * (kTDepth * kTSlot) + (size_t)(nb % kTDepth) * kTSlot;
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
    };

    // ---- the block loop.  Per 32-channel block nb:   wait(W(nb), T(nb)) ; barrier ; [stores of block nb-1] ; issue W(nb+1), T(nb+2) ; MFMA ; epilogue.
    // The T DMAs are the YOUNGEST vector-memory operations of an iteration, so `s_waitcnt vmcnt(4)` at the next iteration's top retires everything
    // older (W(nb+1), T(nb+1), the stores) and leaves exactly T(nb+2) in flight -- independent of how many stores a wave issued.
    // (A schedule with the two halves of the workgroup half a block apart -- one half multiplying while the other runs its epilogue, two barriers per
    // block, three W stages -- measured SLOWER: 117-125 us against 109.5 for the forward at config 5's stage 1; the second-dispatched half of the waves
    // runs every phase 1.4x slower than the first whatever its SIMD partner does, with or without `s_setprio`.)
    f32x16 acc[2];
    s16x4 tpk[2][4];
    issue_w(0);
    issue_t(0);
    if (nblk > 1) issue_t(1);
    // The X fragments are requested BEHIND the first blocks' DMAs (and behind the tables' loads): block 0 then waits for W(0) and T(0) only, and its
    // MFMAs wait fragment by fragment (the compiler's own counted waits) instead of for all 2 KS loads -- the prologue was 13 % of an item.
    // A fragments: this wave's 64 token rows, all of K, straight from global in the operand layout (lane l: row l & 31, k = 16 s + 8 (l >> 5) ..+7)
    const bf16_t *X = tab.X[j];
    bf16x8 af[2][KS];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long tok = min(tok0 + 64 * wave + 32 * i + col, last_tok);         // rows past the image: clamped address, masked below
        const bf16_t *px = X + (size_t)tok * K + 8 * h;
#pragma unroll
        for (int s = 0; s < KS; ++s) af[i][s] = *reinterpret_cast<const bf16x8 *>(px + 16 * s);
    }

    stamp(0);                                                 // prologue: tables, first DMAs issued, X fragments requested
    for (int nb = 0; nb < nblk; ++nb) {
        // everything but the youngest T block (T(nb + 1), issued an iteration ago or in the prologue) has landed: W(nb), T(nb)
        if (nb == 0) {                                        // ... and, the first time, but the 2 KS fragment loads issued behind them
            if (MODE != MODE_PLAIN && nblk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * KS + 4) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * KS) : "memory");
        } else if (MODE != MODE_PLAIN && nb + 1 < nblk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(1);                                             // waited for this wave's DMAs
        __syncthreads();                                     // W(nb): everyone's part; and nobody reads W stage (nb + 1) & 1 any more
        stamp(2);                                             // waited for the other waves
        if (nb > 0) {
            store_block(nb - 1);                             // reads the wave's T slot (nb - 1) % 3, which T(nb + 2) overwrites below
            merge_block(nb - 1);
        }
        if (nb + 1 < nblk) issue_w(nb + 1);
        if (nb + 2 < nblk) issue_t(nb + 2);                  // LAST: see the wait above
        stamp(3);                                             // late stores, merge, DMA issue
        unsigned char *stt = lds + L::kTRing + (size_t)wave * (kTDepth * kTSlot) + (size_t)(nb % kTDepth) * kTSlot;    // this wave's 64 T rows of block nb

        const int cl = nb * kBN + col;                        // channel within the item's range
        const float bias_c = bias_l[cl];
        // the bias goes in as the accumulators' start: D[token][channel] = bias[channel] + sum_k
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = bias_c;
        // B fragments: row `col` of the W block, source chunk 2 s + h, i.e. LDS chunk (2 s + h) ^ (col & SW): byte (wb ^ (s << 5)) of the LDS array.
        // Read four k-steps ahead (two register sets); `wb` is made opaque so that the sixteen addresses are not hoisted out of the block loop.
        unsigned wb = wbase_lane + (unsigned)(nb & 1) * (unsigned)L::kWBytes;
        asm volatile("" : "+v"(wb));
        auto bfrag = [&](int s) -> bf16x8 { return *reinterpret_cast<const bf16x8 *>(lds + (wb ^ (unsigned)(s << 5))); };
        constexpr int GS = 4, NG = KS / GS;
        bf16x8 bq[2][GS];
#pragma unroll
        for (int u = 0; u < GS; ++u) bq[0][u] = bfrag(u);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            if (gi + 1 < NG) {
#pragma unroll
                for (int u = 0; u < GS; ++u) bq[(gi + 1) & 1][u] = bfrag(GS * (gi + 1) + u);
            }
            if (gi == NG - 1) {
                if constexpr (MODE != MODE_PLAIN) {
                    // the T block in the accumulator layout: block (i, q) = token rows 32 i + 8 q + 4 h .. + 3, lane = channel; element e = row e of the block
                    typedef s16x4 __attribute__((address_space(3))) * lds_p;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            tpk[i][q] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(stt + (32 * i + 8 * q + 4 * h + (li >> 2)) * 64 + 32 * g16 + 8 * (li & 3)));
                }
            }
#pragma unroll
            for (int u = 0; u < GS; ++u) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][GS * gi + u], bq[gi & 1][u], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][GS * gi + u], bq[gi & 1][u], acc[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef SD_ALIGN_STAMPS
        asm volatile("" ::"v"(acc[0][0]), "v"(acc[1][15]));   // the accumulators are complete before the stamp
#endif
        stamp(4);                                             // B fragments + MFMAs
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
            // pass 1: the maxima
            float mxs = acc[0][0], mxt = tval(tpk, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; e += 2) {             // v_max3_f32: two elements per instruction
                    mxs = __builtin_fmaxf(__builtin_fmaxf(acc[i][e], acc[i][e + 1]), mxs);
                    mxt = __builtin_fmaxf(__builtin_fmaxf(tval(tpk, i, e), tval(tpk, i, e + 1)), mxt);
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
            const float os = -mxs * c2, ot = -mxt * c2;
            float zs = 0.f, zt = 0.f, a = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float tv = tval(tp2, i, e);
                    zs += ex2(fmaf(acc[i][e], c2, os));
                    const float et = ex2(fmaf(tv, c2, ot));
                    zt += et;
                    a = fmaf(et, tv - acc[i][e], a);
                }
            RowPart p = {mxs, zs, mxt, zt, a};
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
                float *q = slots + ((size_t)(nb & 1) * kNW + wave) * 5 * 32 + lane;
                q[0] = p.ms; q[32] = p.zs; q[64] = p.mt; q[96] = p.zt; q[128] = p.a;
            }
        } else {
            // dY (backward) / Y (plain) of the block: 32 values per lane, rounded to bf16 in token quadruples, transposed through the wave's T slot
            float ls = 0.f, lt = 0.f;
            if constexpr (MODE == MODE_BWD) {
                const float *lse_l = reinterpret_cast<const float *>(lds + L::kLse);
                ls = lse_l[2 * cl];
                lt = lse_l[2 * cl + 1];
            }
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
                            d[e] = kk * (ex2(fmaf(acc[i][r], c2, -ls)) - ex2(fmaf(tval(tpk, i, r), c2, -lt)));      // masked rows: 0 - 0
                            dsum += d[e];
                        } else {
                            d[e] = acc[i][r];
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
                if (lane < 32) slots[((size_t)(nb & 1) * kNW + wave) * 32 + lane] = dsum;
            }
        }
        stamp(5);                                             // epilogue
    }
    __syncthreads();
    store_block(nblk - 1);
    merge_block(nblk - 1);
#ifdef SD_ALIGN_STAMPS
    if (stamps && blockIdx.x == 0 && lane == 0) {
        stamp(3);
#pragma unroll
        for (int k = 0; k < 6; ++k) stamps[wave * 8 + k] = ph[k];
        stamps[wave * 8 + 6] = tprev - tstart;
    }
#endif
}

// ---- dX [T][N] = dY [T][C] . W [C][N]: the projection's input gradient -------------------------------------------------------------------
// Same skeleton as above with the roles turned: a wave owns 32 tokens and keeps their WHOLE output row block (N <= 256: NT accumulator tiles of
// 32 x 32) in registers while the reduction index (the C projected channels) streams by in blocks of 32: the W block [32 c][N] is the natural
// row-major weight (the B operand wants 8 consecutive c of one output column: `ds_read_b64_tr_b16` on the row-major image, no transposed copy of W),
// the wave's dY block [32 tokens][32 c] arrives in a ring of its own.  Per block: 16 MFMAs, 2 row reads, 4 NT transposed reads, one barrier.
// DMA groups G(m) = {W(m + 2), dY(m + 3)} are issued once per iteration in that order (and three of them in the prologue, block indices clamped
// into range, so every iteration issues the same count): at the top of iteration m everything up to W(m) must have landed, and what may stay
// in flight is dY(m + 1), W(m + 1), dY(m + 2) = kWDma + 4 instructions: one counted wait, three W stages, four dY slots per wave.
constexpr int kDxTok = 32;          // tokens per wave
constexpr int kDxWDepth = 3, kDxYDepth = 4;
constexpr int kDxYSlot = kDxTok * kBN * 2;     // 2 KB

template <int NT>
__global__ __launch_bounds__(64 * kNW, 2) void tok_dx_kernel(const bf16_t *__restrict__ dY, const bf16_t *__restrict__ W, bf16_t *__restrict__ dX, long T, int C) {
    constexpr int N = 32 * NT;
    constexpr int kWBytes = kBN * N * 2;                    // W block: 32 rows of 2N bytes
    constexpr int kWInst = kWBytes / 1024;                  // 4 (N = 64) .. 16 (N = 256)
    constexpr int kWDma = (kWInst + kNW - 1) / kNW;         // per wave (1 or 2)
    constexpr int kYRing = kDxWDepth * kWBytes;
    constexpr int kLoop = kYRing + kNW * kDxYDepth * kDxYSlot;
    constexpr int kImg = kNW * kDxTok * N * 2;              // the output tile, transposed through LDS once the loop is over
    constexpr int kTotal = kLoop > kImg ? kLoop : kImg;
    constexpr int JM = NT >= 4 ? 3 : NT - 1;                // unit swizzle mask of the W image
    __shared__ __attribute__((aligned(1024))) unsigned char lds[kTotal];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int h = lane >> 5, col = lane & 31, g16 = (lane >> 4) & 1, li = lane & 15;
    const long tok0 = (long)blockIdx.x * (kDxTok * kNW) + (long)wave * kDxTok;       // this wave's first token
    const int wrows = (int)min((long)kDxTok, max(0L, T - tok0));
    const int nblk = C / kBN;

    // W block: LDS chunk p of row r holds source chunk p ^ ((r & JM) << 2): the four rows of a transposed read then sit in different 64-byte units
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    unsigned woff[kWDma], wdst[kWDma];
#pragma unroll
    for (int u = 0; u < kWDma; ++u) {
        const int q = min(wave * kWDma + u, kWInst - 1);
        const int o = q * 1024 + 16 * lane;
        const int r = o / (2 * N), p = (o % (2 * N)) >> 4;
        woff[u] = (unsigned)(r * 2 * N + 16 * (p ^ ((r & JM) << 2)));
        wdst[u] = (unsigned)(q * 1024);
    }
    // dY block: 32 rows of 64 bytes, LDS chunk p of row r holds source chunk p ^ ((r >> 2) & 3) (conflict-free 16-byte row reads of 32 rows)
    const long tlast = T - 1;
    const bf16_t *ybase = dY + (size_t)min(tok0, tlast) * C;
    unsigned yoff[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int r = 16 * u + (lane >> 2), p = lane & 3;
        const long tok = min(tok0 + r, tlast);
        yoff[u] = (unsigned)((tok - min(tok0, tlast)) * C * 2 + 16 * (p ^ ((r >> 2) & 3)));
    }
    const unsigned yring = lds0 + (unsigned)kYRing + (unsigned)wave * (unsigned)(kDxYDepth * kDxYSlot);
    auto issue_group = [&](int m) {                          // G(m) = {W(m + 2), dY(m + 3)}, block indices clamped; kWDma + 2 instructions
        const int bw = min(max(m + 2, 0), nblk - 1), by = min(max(m + 3, 0), nblk - 1);
        const unsigned sw = lds0 + (unsigned)((m + 2 + kDxWDepth) % kDxWDepth) * (unsigned)kWBytes;
        const bf16_t *wb_ = W + (size_t)bw * kBN * N;
#pragma unroll
        for (int u = 0; u < kWDma; ++u) dma16(wb_, woff[u], __builtin_amdgcn_readfirstlane(sw + wdst[u]));
        const unsigned sy = yring + (unsigned)((m + 3 + kDxYDepth) % kDxYDepth) * (unsigned)kDxYSlot;
        const bf16_t *yb_ = ybase + by * kBN;
#pragma unroll
        for (int u = 0; u < 2; ++u) dma16(yb_, yoff[u], __builtin_amdgcn_readfirstlane(sy + (unsigned)(16 * u * 64)));
    };

    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    issue_group(-3);
    issue_group(-2);
    issue_group(-1);
    typedef s16x4 __attribute__((address_space(3))) * lds_p;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    for (int nb = 0; nb < nblk; ++nb) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kWDma + 4) : "memory");              // all but dY(nb + 1) and G(nb - 1): W(nb) and dY(nb) have landed
        __syncthreads();                                     // W(nb): everyone's part; nobody reads W stage (nb + 2) % 3 any more
        issue_group(nb);
        const unsigned char *sw = lds + (size_t)(nb % kDxWDepth) * kWBytes;
        const unsigned char *sy = lds + kYRing + (size_t)wave * (kDxYDepth * kDxYSlot) + (size_t)(nb % kDxYDepth) * kDxYSlot;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8 af = *reinterpret_cast<const bf16x8 *>(sy + col * 64 + 16 * ((2 * s + h) ^ ((col >> 2) & 3)));
            const int q = li >> 2, p = li & 3;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                // B fragment: channels 16 s + 8 h .. + 7 of output column 32 j + (lane & 31): two transposed reads of four rows each
                const int r0 = 16 * s + 8 * h + q;
                const unsigned off = (unsigned)(16 * (4 * (j ^ (q & JM)) + 2 * g16 + (p >> 1)) + 8 * (p & 1));
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(sw + r0 * (2 * N) + off));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(sw + (r0 + 4) * (2 * N) + off));
                const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, v), acc[j], 0, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the tail's placeholder DMAs must not land in what follows
    __syncthreads();                                         // everybody is done with the W stages: the LDS becomes the waves' output images
    // D[token][out]: the out column on the lane, 16 token rows in the registers.  Per 32-column tile j an image [32 outs][32 tokens] of 64-byte rows,
    // 8-byte unit u (tokens 4u .. 4u + 3) stored at unit u ^ ((out >> 1) & 7); read back token-major: lane -> token 16 tb + li, outs 8 gq .. + 7.
    unsigned char *img = lds + (size_t)wave * (kDxTok * N * 2);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bf16x2 lo = __builtin_convertvector((f32x2){acc[j][4 * q], acc[j][4 * q + 1]}, bf16x2);
            const bf16x2 hi = __builtin_convertvector((f32x2){acc[j][4 * q + 2], acc[j][4 * q + 3]}, bf16x2);
            const u32x2 v = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
            *reinterpret_cast<u32x2 *>(img + j * 2048 + col * 64 + 8 * ((2 * q + h) ^ ((col >> 1) & 7))) = v;
        }
    const int gq = lane >> 4;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const int oa = 8 * gq + (li >> 2), ob = oa + 4, u = 4 * tb + (li & 3);
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + j * 2048 + oa * 64 + 8 * (u ^ ((oa >> 1) & 7))));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + j * 2048 + ob * 64 + 8 * (u ^ ((ob >> 1) & 7))));
            const int row = 16 * tb + li;
            if (row < wrows) {
                const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                *reinterpret_cast<s16x8 *>(dX + (size_t)(tok0 + row) * N + 32 * j + 8 * gq) = v;
            }
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
    constexpr long kSlots = 256;
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

unsigned long long *g_align_stamps = nullptr;   // diagnostics (sd_debug_align_stamps): [8 waves][8] cycle sums of workgroup 0

template <int MODE>
int launch_tab(const AlignTokTable &tab, int K, unsigned *tickets, hipStream_t st) {
    const unsigned nblk = (unsigned)tab.blk_begin[tab.njobs];
    if (K == 256) hipLaunchKernelGGL((align_tok_kernel<16, MODE>), dim3(nblk), dim3(64 * kNW), 0, st, tab, tickets, kTokMaxJobs, g_align_stamps);
    else if (K == 128) hipLaunchKernelGGL((align_tok_kernel<8, MODE>), dim3(nblk), dim3(64 * kNW), 0, st, tab, tickets, kTokMaxJobs, g_align_stamps);
    else hipLaunchKernelGGL((align_tok_kernel<4, MODE>), dim3(nblk), dim3(64 * kNW), 0, st, tab, tickets, kTokMaxJobs, g_align_stamps);
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

#ifdef SD_ALIGN_STAMPS
void sd_debug_align_stamps(void *buf) { sd::g_align_stamps = static_cast<unsigned long long *>(buf); }   // diagnostic build only (not in the header)
#endif

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

int sd_linear_tok_bf16_bwd_data(const void *dY, const void *W, void *dX, long tokens, int out_features, int in_features, void *stream) {
    if (!dY || !W || !dX) return SD_E_NULL;
    if (tokens <= 0 || out_features <= 0 || in_features <= 0 || (tokens + 255) / 256 > 0x7fffffffL) return SD_E_SHAPE;
    if (!sd::shape_ok(in_features, out_features)) return SD_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(dX)) & 15) return SD_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((tokens + 255) / 256)), block(64 * sd::kNW);
    const sd::bf16_t *a = static_cast<const sd::bf16_t *>(dY), *w = static_cast<const sd::bf16_t *>(W);
    sd::bf16_t *o = static_cast<sd::bf16_t *>(dX);
    if (in_features == 256) hipLaunchKernelGGL((sd::tok_dx_kernel<8>), grid, block, 0, st, a, w, o, tokens, out_features);
    else if (in_features == 128) hipLaunchKernelGGL((sd::tok_dx_kernel<4>), grid, block, 0, st, a, w, o, tokens, out_features);
    else hipLaunchKernelGGL((sd::tok_dx_kernel<2>), grid, block, 0, st, a, w, o, tokens, out_features);
    return (int)hipGetLastError();
}

}  // extern "C"
