/*
 * segdistill_hip.h -- C ABI of libsegdistill_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the dense distillation-loss hot path of
 * wzpscott/SegDistill.  The reference is 100 % Python and has no FFI of its
 * own; every entry point below replaces a run of ATen calls made by the
 * reference's Python (file:line cited per function, relative to the reference
 * root).  The Python binding a maintainer adds is a ctypes stub, shown in
 * INTEGRATION.md and implemented in segdistill_amd/_lib.py.
 *
 * Contract:
 *  - plain C types only; device pointers are raw addresses owned by the caller
 *    (torch tensors in practice); the library allocates nothing persistent.  Its
 *    only process-wide state are the benchmarking tunables below (sd_set_tunable):
 *    they change launch geometry AND the answers of the *_workspace_bytes()
 *    queries, so a tunable must not be changed between sizing a workspace and
 *    the launch that uses it (the launch then fails cleanly with SD_E_WORKSPACE);
 *    production code never sets them;
 *  - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*;
 *    NULL = the default stream) and never synchronises the host;
 *  - return value: 0 = ok, <0 = argument error (SD_E_*), >0 = a hipError_t;
 *    nothing throws across the boundary.  sd_error_string() names a code.
 *
 * Tensor layout: contiguous NCHW.  `dtype` selects the storage type of the
 * activation operands (SD_F32 | SD_BF16); all arithmetic and all statistics
 * are fp32 (finalisation in fp64).
 *
 * Row geometry of the channel-group criteria ("CGD"; g = group_size):
 *   G = ceil(C/g) groups per image, rows = B*G, row (b,j) = channel slots
 *   j*g .. j*g+g-1 of image b, all H*W pixels.  Slot c' < C holds channel
 *   perm[c'] (or c' when perm == NULL); slots >= C are the reference's -1e9
 *   padding (losses.py:55-58) and are treated as virtual (-inf): they are never
 *   materialised, read or written.
 */
#ifndef SEGDISTILL_HIP_H
#define SEGDISTILL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SD_ABI_VERSION 2

enum { SD_F32 = 0, SD_BF16 = 1 };

enum {
    SD_OK = 0,
    SD_E_NULL = -1,      /* required pointer is NULL */
    SD_E_SHAPE = -2,     /* non-positive or overflowing dimension */
    SD_E_DTYPE = -3,     /* unknown dtype code */
    SD_E_WORKSPACE = -4, /* workspace too small / misaligned */
    SD_E_ALIGN = -5,     /* operand pointer not aligned to its element size */
    SD_E_UNSUPPORTED = -6
};

int sd_abi_version(void);
const char *sd_error_string(int code);

/* Tunables (process-wide, NOT thread-safe, for benchmarking only; defaults are the shipped values; see the contract above:
 * set them before sizing workspaces, never between a *_workspace_bytes() call and its launch).
 * keys: "cgd_fwd_chunk_iters" / "cgd_bwd_chunk_iters" (rounds of 4 x 16-byte loads per
 *       operand per lane per workgroup), "cgd_bwd_nt_store" (0|1), "cgd_bwd_unroll" (2|4|8),
 *       "cgd_up_band_rows" (tap rows per workgroup of the fused-upsample kernels),
 *       "sra_split_bf16" (0|1, default 1: fp32-storage attention products on the bf16 matrix pipe with every operand split exactly
 *       into three bf16 terms -- fp32-grade results; 0 = v_mfma_f32_32x32x2_f32), "sra_bf16_mfma" (0|1, default 1: bf16-storage
 *       attention forward and backward on the bf16 matrix pipe with P / dS rounded to bf16; 0 = the f32-input MFMA kernels), "align_split_bf16"
 *       (0|1, default 1: the three fp32 products of the 1x1 align projection and of sd_linear_nchw_* in split-bf16 arithmetic),
 *       "ce_bwd_multiclass" (0|1, default 1: sd_ce_up_bwd handles 4 (2 at
 *       factor 8) class planes per workgroup with the pixel maps loaded once per group; 0 = one class per workgroup, rounds 1-2).
 *       "wgrad_tn_ring" (0|1|2, default 1: the bf16 weight-gradient kernels of csrc/wgrad_tn.hip -- 0 register-staged tiles everywhere, 1 the LDS-DMA
 *       ring where legal, 2 its 256 x 256 tiles wherever legal), "wgrad_slab_ratio" (per cent, default 40, 0 = no cap: the most split-K slab bytes a
 *       bf16-storage weight gradient may write, as a share of its operand bytes), "wgrad_multi_wgs" (0 = default: workgroups per launch group
 *       that sd_linear_wgrad_tn_multi_plan deals out -- 1536 for bf16, 3072 for fp32 storage; planning only: set it before _plan),
 *       "tok_gemm_bf16_variant" (-1 = by shape; 0..4 force a tile / ring variant of sd_linear_bf16_fwd), "planes_tile" (0 = by shape; 128 / 64: row
 *       tile of the split-bf16 planes GEMM), "align_stream" (0|1, default 1: sd_align1x1_fwd with Cs <= 128 on the streaming kernel of
 *       csrc/align_stream.hip -- W resident in registers -- instead of the generic pipelined GEMM).
 *       The sra_*, align_*, ce_*, wgrad_tn_ring, tok_gemm_bf16_variant and planes_tile keys select arithmetic or tiling, not workspace geometry: no
 *       workspace size depends on them.
 *       "wgrad_slab_ratio" IS workspace geometry: it changes what sd_linear_wgrad_generic_slabs / sd_linear_wgrad_slabs /
 *       sd_linear_wgrad_workspace_bytes answer, so set it before any of those queries (the Python binding reads SEGDISTILL_WGRAD_SLAB_RATIO once,
 *       when the library is loaded). */
int sd_set_tunable(const char *key, int value);
int sd_get_tunable(const char *key);

/* ---------------------------------------------------------------------------
 * CGD / CD criterion, operands ALREADY at softmax resolution ("R1").
 * Replaces losses.py:105-112 (transform pad+view, div tau, log_softmax, softmax,
 * KLDivLoss(sum), normalise, *alpha) and, with `perm`, the gather+copy of
 * losses.py:39-41.  Forward reads S and T once (2*N*e bytes).
 *
 *   loss      = loss_scale * sum_rows KL_r           (loss_scale = alpha/rows)
 *   row_lse2  [rows][2] : base-2 log-partition of S/tau and T/tau per row
 *                         (consumed by the backward)
 *   row_kl    [rows]    : KL_r in nats (diagnostic; also the parity surface)
 *
 * workspace: sd_cgd_kl_workspace_bytes() bytes, 16-byte aligned, contents
 * undefined on entry and exit.
 */
size_t sd_cgd_kl_workspace_bytes(int B, int C, int H, int W, int g);

int sd_cgd_kl_fwd(const void *S, const void *T, int dtype,
                  int B, int C, int H, int W, int g,
                  float inv_tau, float loss_scale,
                  const int32_t *perm /* [C] device, or NULL */,
                  float *row_lse2, float *row_kl, float *loss /* [1] */,
                  void *workspace, size_t workspace_bytes, void *stream);

/* Backward of the above w.r.t. S (the teacher gets no gradient; losses.py runs
 * the teacher under no_grad, SD_structure.py:65-67):
 *   dS_i = upstream * coef * (softmax(S/tau)_i - softmax(T/tau)_i),
 *   coef = alpha/(rows*tau);  `upstream` is a 1-element DEVICE float (the
 *   autograd grad_output) or NULL for 1.  Reads S,T, writes dS (3*N*e bytes).
 * Replaces the autograd chain of losses.py:108-112
 * (_log_softmax_backward_data, kl_div backward, div backward).
 */
int sd_cgd_kl_bwd(const void *S, const void *T, int dtype,
                  int B, int C, int H, int W, int g,
                  float inv_tau, float coef,
                  const int32_t *perm, const float *row_lse2,
                  const float *upstream, void *dS, void *stream);

/* ---------------------------------------------------------------------------
 * CGD / CD criterion with the bilinear up-sampling fused in ("R2").
 * s,t are the TAPPED tensors [B,C,h,w]; the softmax runs at (H,W) = (F*h, F*w),
 * F in {2,4,8}, align_corners=False.  Replaces losses.py:101-102 (two
 * F.interpolate calls, each materialising a [B,C,H,W] tensor) PLUS losses.py:105-112,
 * and in the backward additionally upsample_bilinear2d_backward.  Nothing of size
 * B*C*H*W is ever written to memory.  Same row geometry, outputs and statistics as
 * sd_cgd_kl_fwd/bwd; `ds` is the gradient w.r.t. the tap s ([B,C,h,w]).
 * sd_cgd_kl_up_supported() tells whether a shape pair has a fused kernel (otherwise the
 * caller resizes with ATen and uses the R1 entry points).
 */
int sd_cgd_kl_up_supported(int h, int w, int H, int W);
size_t sd_cgd_kl_up_workspace_bytes(int B, int C, int h, int w, int H, int W, int g);

int sd_cgd_kl_up_fwd(const void *s, const void *t, int dtype,
                     int B, int C, int h, int w, int H, int W, int g,
                     float inv_tau, float loss_scale, const int32_t *perm,
                     float *row_lse2, float *row_kl, float *loss,
                     void *workspace, size_t workspace_bytes, void *stream);

int sd_cgd_kl_up_bwd(const void *s, const void *t, int dtype,
                     int B, int C, int h, int w, int H, int W, int g,
                     float inv_tau, float coef, const int32_t *perm,
                     const float *row_lse2, const float *upstream,
                     void *ds, void *stream);

/* TWO channel criteria on the SAME taps in one pass each way (BASELINE config 3: CGD g = 8, tau = 4 plus the channel-wise logit KL g = 1,
 * tau = 1 on decode_head.linear_pred; the reference's DistillationLoss calls KLDLoss.forward once per entry -- opts.py:100-110 -> losses.py:95-113
 * -- so both taps are resized and read twice and two tap gradients are added).  Forward: every interpolated pair is folded into both
 * criteria's online-softmax states; outputs per criterion as sd_cgd_kl_up_fwd; workspace >= 2 x sd_cgd_kl_up_workspace_bytes.  Backward:
 * ds = coef_a (p_s - p_t)_a + coef_b (p_s - p_t)_b through one transposed interpolation.  `perm` orders the channel slots of BOTH criteria:
 * callers fuse only when that is what the two criteria ask for (criterion b without a shuffle of its own and g_b == 1, where the slot order is
 * immaterial, or neither with a shuffle). */
int sd_cgd_kl_up_fwd2(const void *s, const void *t, int dtype, int B, int C, int h, int w, int H, int W, const int32_t *perm, int g_a,
                      float inv_tau_a, float loss_scale_a, float *row_lse2_a, float *row_kl_a, float *loss_a, int g_b, float inv_tau_b,
                      float loss_scale_b, float *row_lse2_b, float *row_kl_b, float *loss_b, void *workspace, size_t workspace_bytes,
                      void *stream);
int sd_cgd_kl_up_bwd2(const void *s, const void *t, int dtype, int B, int C, int h, int w, int H, int W, const int32_t *perm, int g_a,
                      float inv_tau_a, float coef_a, const float *row_lse2_a, const float *upstream_a, int g_b, float inv_tau_b, float coef_b,
                      const float *row_lse2_b, const float *upstream_b, void *ds, void *stream);

/* ---------------------------------------------------------------------------
 * Pixel-wise criterion (PDLoss; KL term of ATLoss / IFVDLoss): rows = (b, pixel),
 * softmax over the C channels, loss_scale = alpha/(B*H*W), coef = alpha/(B*H*W*tau).
 * Replaces losses.py:47-49 (permute+reshape copy) and :108-112 for loss_type='pixel',
 * and ATLoss/IFVDLoss's log_softmax/softmax/KLDivLoss over dim=1 (losses.py:191-195,
 * :218-220).  pix_lse2: [2][B*H*W] base-2 log-partitions (student plane, teacher plane).
 */
size_t sd_pix_kl_workspace_bytes(int B, int C, int H, int W);

int sd_pix_kl_fwd(const void *S, const void *T, int dtype, int B, int C, int H, int W,
                  float inv_tau, float loss_scale, float *pix_lse2, float *loss,
                  void *workspace, size_t workspace_bytes, void *stream);

int sd_pix_kl_bwd(const void *S, const void *T, int dtype, int B, int C, int H, int W,
                  float inv_tau, float coef, const float *pix_lse2,
                  const float *upstream, void *dS, void *stream);

/* The pixel-wise criterion with the bilinear up-sampling FUSED in (csrc/pix_up.hip): s, t are the TAPPED logits [B,C,h,w]; the class softmax runs at
 * (H, W) = (F*h, F*w), F in {2,4,8}, align_corners=False.  Replaces losses.py:101-102 (both F.interpolate calls) PLUS losses.py:47-49,108-112
 * (`loss_type='pixel'`: the PDLoss preset of :115-128) and, in the backward, upsample_bilinear2d_backward; nothing of size B*C*H*W is written.
 * pix_lse2: [2][B*H*W] as sd_pix_kl_fwd (written by fwd, read by bwd); `ds` is the gradient with respect to the tap s.  A channel shuffle
 * (losses.py:39-41) does not change a softmax over ALL channels of a pixel, so there is no permutation argument. */
int sd_pix_kl_up_supported(int h, int w, int H, int W);
size_t sd_pix_kl_up_workspace_bytes(int B, int h);
int sd_pix_kl_up_fwd(const void *s, const void *t, int dtype, int B, int C, int h, int w, int H, int W,
                     float inv_tau, float loss_scale, float *pix_lse2, float *loss,
                     void *workspace, size_t workspace_bytes, void *stream);
int sd_pix_kl_up_bwd(const void *s, const void *t, int dtype, int B, int C, int h, int w, int H, int W,
                     float inv_tau, float coef, const float *pix_lse2,
                     const float *upstream, void *ds, void *stream);

/* ATLoss (losses.py:175-197: nn.MSELoss between x.mean(dim=1) maps + the class-softmax KL, tau = 1) fused into the same two
 * passes: loss = mean_{b,p}(mean_c S - mean_c T)^2 + 1/(B*H*W) * sum_pixels KL.  planes: [3][B*H*W] fp32 (base-2 lse of S and
 * T, channel-mean difference), written by fwd, read by bwd.  Workspace: sd_pix_kl_workspace_bytes. */
int sd_at_kl_fwd(const void *S, const void *T, int dtype, int B, int C, int H, int W,
                 float *planes, float *loss, void *workspace, size_t workspace_bytes, void *stream);
int sd_at_kl_bwd(const void *S, const void *T, int dtype, int B, int C, int H, int W,
                 const float *planes, const float *upstream /* d loss, device scalar, or NULL = 1 */, void *dS, void *stream);

/* ---------------------------------------------------------------------------
 * IFVDLoss, intra-class feature-variation term (losses.py:221-235): cosine similarity of every pixel's feature to the mean
 * feature of its class (per image), matched between student and teacher: 10 * mean_p (sim_S - sim_T)^2, and its gradient
 * with respect to the student feature INCLUDING the path through the class means (as the reference's autograd does).
 * cls [B][HW] int32 is the class of each pixel (-1 / >= K: no class).  Call order, forward: counts, class_means, cos; backward:
 * coef_sums, bwd.  A class sum is a product with the one-hot label matrix, run on the bf16 matrix pipe with the features split exactly
 * into three bf16 terms (fp32-grade): no sort of the pixels, no gathers, deterministic.
 *   sd_ifvd_counts       counts[b,k] = number of pixels of image b with class k; stepmask (sd_ifvd_stepmask_ints ints): the class blocks
 *                        present in every step of 16 pixels, which the products skip by
 *   sd_ifvd_class_means  mean_x[b,c,k] = sum_{p: cls = k} X[b,c,p] / (n_k + 1e-6) for both networks in one launch (T, mean_t may be NULL);
 *                        tables are [B][C][K]: the per-pixel passes read one channel's K values per wave
 *   sd_ifvd_cos          per pixel cos(X[b,:,p], mean_x[b,:,cls]) (eps 1e-8 per norm, F.cosine_similarity) for both networks,
 *                        loss = 10 * mean (sim_S - sim_T)^2, and coefs [3][B*HW] = alpha, beta, gamma for the backward
 *   sd_ifvd_coef_sums    A[b,c,k] = sum_{p: cls = k} alpha_p S[b,c,p],  Bk[b,k] = sum_{p: cls = k} beta_p
 *   sd_ifvd_bwd          dS = upstream * (alpha*mu_k - gamma*S + (A_k - mu_k*B_k)/(n_k + 1e-6))
 * Workspace (class_means, cos, coef_sums; 16-byte aligned): sd_ifvd_workspace_bytes.  S, T, dS: [B,C,HW] in `dtype`; everything else
 * fp32 / int32.
 */
size_t sd_ifvd_workspace_bytes(int B, int C, int HW, int K);
size_t sd_ifvd_stepmask_ints(int B, int HW, int K);
int sd_ifvd_counts(const int *cls, int B, int HW, int K, int *counts, int *stepmask, void *stream);
int sd_ifvd_class_means(const void *S, const void *T /* or NULL */, int dtype, const int *cls, const int *stepmask, const int *counts,
                        float *mean_s, float *mean_t, void *workspace, size_t workspace_bytes, int B, int C, int HW, int K, void *stream);
int sd_ifvd_cos(const void *S, const void *T, int dtype, const int *cls, const float *mean_s, const float *mean_t, float *coefs, float *loss,
                void *workspace, size_t workspace_bytes, int B, int C, int HW, int K, void *stream);
int sd_ifvd_coef_sums(const void *S, int dtype, const int *cls, const int *stepmask, const int *counts, const float *coefs, float *A, float *Bk,
                      void *workspace, size_t workspace_bytes, int B, int C, int HW, int K, void *stream);
int sd_ifvd_bwd(const void *X, int dtype, const int *cls, const float *mean, const float *coefs, const float *A, const float *Bk,
                const int *counts, const float *upstream /* device scalar or NULL */, void *dS,
                int B, int C, int HW, int K, void *stream);

/* ---------------------------------------------------------------------------
 * 1x1 feature-alignment projection of the student feature: the
 * `channel_nums=(Cs,Ct)` option documented at opts.py:25-27 of the reference (its live code
 * never builds the nn.Conv2d(Cs, Ct, 1) it describes; the commented generation did, at
 * losses.py:258,332-333).  Dense GEMMs on the matrix cores (v_mfma_f32_32x32x2_f32, exact
 * fp32; bf16 storage is widened on load, fp32 accumulation).  W [Ct][Cs] and bias [Ct] are
 * fp32 master parameters; X, Y, dY, dX are [B,C,h,w] in `dtype`.
 *   fwd        Y  = W . X + bias
 *   bwd_data   dX = W^T . dY
 *   bwd_weight dW = sum_b dY_b . X_b^T (deterministic split-K: partial slabs in `workspace`,
 *              then one combine pass), dbias = sum_{b,p} dY   (dbias may be NULL)
 */
size_t sd_align1x1_workspace_bytes(int B, int Cs, int Ct, int h, int w);

int sd_align1x1_fwd(const void *X, const float *W, const float *bias /* or NULL */, void *Y, int dtype,
                    int B, int Cs, int Ct, int h, int w, void *stream);

int sd_align1x1_bwd_data(const void *dY, const float *W, void *dX, int dtype,
                         int B, int Cs, int Ct, int h, int w, void *stream);

int sd_align1x1_bwd_weight(const void *dY, const void *X, float *dW, float *dbias, int dtype,
                           int B, int Cs, int Ct, int h, int w,
                           void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------
 * Weight gradient of nn.Linear on token-major activations: dW [out][in] = dY^T . X with
 * dY [tokens][out], X [tokens][in].  Replaces the aten::mm of every Linear backward in the MiT
 * blocks (q / kv / proj / fc1 / fc2, mix_transformer.py:24-27,75-76,84).  At stage 1 these are
 * tall-skinny GEMMs (tokens = B*128*128 = 131072, out/in = 32..256) that are HBM-bound; the token
 * axis is split over workgroups (f32 MFMA partial slabs in `workspace`, deterministic combine).
 */
size_t sd_linear_wgrad_workspace_bytes(long tokens, int out_features, int in_features);
/* 1 if the kernel chosen for this shape also produces the bias gradient (column sums of dY) for free */
int sd_linear_wgrad_fuses_bias(long tokens, int out_features, int in_features);                 /* fp32 storage */
int sd_linear_wgrad_fuses_bias_dtype(int dtype, long tokens, int out_features, int in_features);  /* the kernel choice depends on the storage type */

int sd_linear_wgrad(const void *dY, const void *X, float *dW, float *dbias /* [out] or NULL */, int dtype,
                    long tokens, int out_features, int in_features,
                    void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------
 * The adaptive average pools of a Pyramid Pooling Module, every pool scale in ONE pass over the map (csrc/ppm_pool.hip).
 * Replaces: nn.AdaptiveAvgPool2d(s) per scale in PPM.forward (reference mmseg/models/decode_heads/psp_head.py:10-58, used by PSPHead :61-101 and
 * UPerHead, uper_head.py:76-126) and their autograd -- ATen's float-atomic `atomic_adaptive_average_gradinput` plus the adds of the branch gradients.
 *   x [planes = B*C][h][w] contiguous; pooled[k] / d_pooled[k]: [planes][s_k][s_k]; bins as ATen's (start = floor(i h / s), end = ceil((i+1) h / s)).
 *   forward reads x once; backward writes dx once as a gather (deterministic, no atomics).
 * _supported(): at most 4 scales of at most 8 bins, one plane + its row sums within 48 KB of LDS (else SD_E_UNSUPPORTED: use the framework's op).
 */
int sd_ppm_pool_supported(int h, int w, const int *scales, int nscales);
int sd_ppm_pool_fwd(const void *x, int dtype, long planes, int h, int w, const int *scales, int nscales, void *const *pooled, void *stream);
int sd_ppm_pool_bwd(void *const *d_pooled, int dtype, long planes, int h, int w, const int *scales, int nscales, void *dx, void *stream);

/* ---------------------------------------------------------------------------
 * Window multi-head self-attention of a Swin block, forward only (the frozen teacher of BASELINE config 4): reference
 * mmseg/models/backbones/swin_transformer.py:119-153 -- per window and head  softmax((q*scale) k^T + bias[head] + mask[window % nW]) v
 * on the matrix pipe (v_mfma_f32_32x32x2_f32, exact fp32).
 *   qkv    [windows, N, 3, heads, D]  the qkv Linear's output as it stands (swin_transformer.py:128-129 before the permute)
 *   out    [windows, N, heads*D]  = (attn @ v).transpose(1, 2).reshape(B_, N, C)   (:148)
 * Bias and mask come PACKED in the accumulator layout of the kernel, sd_window_attn_packed_floats() (= 64 * 64) floats per table:
 * sd_window_attn_pack converts `count` tables [49][49] in the reference's orientation ([query][key]: relative_position_bias[head] of :133-135,
 * mask[window] of :138; window b uses mask b % mask_windows, :140) and fills padded keys with pad_key_value (-INFINITY for the bias, 0 for the
 * mask); flags[count] (optional) receives 1 for a table with a nonzero entry -- the kernel skips the mask read of windows whose mask is all
 * zero (all but 37 of a 19 x 19 shifted partition).  mask_packed NULL: no mask (mask_flags ignored).
 * fp32 storage only (SD_E_DTYPE otherwise); _supported(): N == 49 (7 x 7 windows) and D == 32, else SD_E_UNSUPPORTED -- use the framework's op.
 */
int sd_window_attn_supported(int tokens_per_window, int head_dim);
size_t sd_window_attn_packed_floats(void);
int sd_window_attn_pack(const float *tables, float *packed, int32_t *flags /* or NULL */, int count, int tokens_per_window, float pad_key_value,
                        void *stream);
int sd_window_attn_fwd_packed(const void *qkv, const float *bias_packed, const float *mask_packed, const int32_t *mask_flags, void *out, int dtype,
                              long windows, int mask_windows, int heads, int tokens_per_window, int head_dim, float scale, void *stream);

/* ---------------------------------------------------------------------------
 * CGD / CD criterion on TOKEN-MAJOR operands S, T [B][P][C] (C contiguous; P = h*w pixels): the decoder features of a SegFormer head
 * (decode_head.linear_c1..4 emit [B, h*w, E]; SURVEY a-16, reference opts.py:25-27 / the reshape helper commented out at
 * losses.py:300-318).  Same rows, closed form, row_lse2 / row_kl / loss outputs and `perm` semantics as sd_cgd_kl_fwd / _bwd
 * (losses.py:105-112, gather :39-41, pad :55-58) -- only the memory layout differs, so no transpose copy of either tap or of the
 * gradient is needed.  Requires C % 4 == 0 (fp32) / C % 8 == 0 (bf16) and 16-byte aligned bases (else SD_E_UNSUPPORTED / SD_E_ALIGN:
 * view as NCHW and use the R1 entry points); with `perm`, C <= 2048.
 */
size_t sd_cgd_kl_tok_workspace_bytes(int B, int C, long P);
int sd_cgd_kl_tok_fwd(const void *S, const void *T, int dtype, int B, int C, long P, int g, float inv_tau, float loss_scale,
                      const int32_t *perm, float *row_lse2, float *row_kl, float *loss, void *workspace, size_t workspace_bytes, void *stream);
int sd_cgd_kl_tok_bwd(const void *S, const void *T, int dtype, int B, int C, long P, int g, float inv_tau, float coef, const int32_t *perm,
                      const float *row_lse2, const float *upstream, void *dS, void *stream);

/* Several token-major criteria in ONE call (config 5 taps four decoder stages: reference call site opts.py:100-110 evaluates one entry after the
 * other; each was four chained launches): at most sd_cgd_kl_tok_max_jobs() jobs, all of one dtype; field meanings as the arguments above.
 * Forward = one scan launch per chunk class (long / 16-pixel chunks) + ONE finish launch (row statistics in fp64 and every job's loss, the
 * latter by the workgroup that draws its job's last arrival ticket; the tickets live in job 0's workspace).  Backward = ONE launch. */
typedef struct sd_cgd_tok_fwd_job {
    const void *S, *T;          /* [B][P][C] */
    const int32_t *perm;        /* [C] or NULL */
    float *row_lse2;            /* [rows][2] */
    float *row_kl;              /* [rows] */
    float *loss;                /* [1] */
    void *workspace;            /* >= sd_cgd_kl_tok_workspace_bytes(B, C, P), 16-byte aligned, private to the job */
    size_t workspace_bytes;
    long P;
    int B, C, g;
    float inv_tau, loss_scale;
    int reserved;
} sd_cgd_tok_fwd_job;
typedef struct sd_cgd_tok_bwd_job {
    const void *S, *T;
    const int32_t *perm;
    const float *row_lse2;
    const float *upstream;      /* [1] or NULL */
    void *dS;                   /* [B][P][C], storage dtype */
    long P;
    int B, C, g;
    float inv_tau, coef;
    int reserved;
} sd_cgd_tok_bwd_job;
int sd_cgd_kl_tok_max_jobs(void);
int sd_cgd_kl_tok_fwd_multi(const sd_cgd_tok_fwd_job *jobs, int njobs, int dtype, void *stream);
int sd_cgd_kl_tok_bwd_multi(const sd_cgd_tok_bwd_job *jobs, int njobs, int dtype, void *stream);

/* ---------------------------------------------------------------------------
 * The feature-align projection of a token-major tap FUSED with the channel-group criterion it feeds (csrc/align_tok.hip; bf16 storage).
 * Replaces, for one distillation entry with `channel_nums=(Cs, Ct)` on token-major taps [B, P, Cs] / [B, P, Ct] (BASELINE config 5):
 * the 1x1 projection of the student feature (opts.py:25-27; commented `self.ff = nn.Conv2d(**ff_config, kernel_size=1)` of
 * losses.py:258,332-333,373-374) AND KLDLoss.forward on its output (losses.py:95-113) -- the projected feature Y = X.W^T + bias is consumed
 * in the MFMA accumulators (fp32) and never written.
 *   forward   X [B*P][K] bf16, W [C][K] bf16 (the rounded copy of the fp32 master weight), bias [C] fp32 or NULL, T [B*P][C] bf16
 *             -> row_lse2 / row_kl / loss exactly as sd_cgd_kl_tok_fwd (row geometry of the header's top comment; perm orders the C
 *             projected channels).  Launches: one scan + the finish launch of sd_cgd_kl_tok_fwd_multi for all jobs.
 *   backward  recomputes Y, writes dY [B*P][C] bf16 = upstream * coef * (softmax_row(Y/tau) - softmax_row(T/tau)) -- the operand of the
 *             input- and weight-gradient GEMMs (sd_linear_tok_bf16_bwd_data, sd_linear_wgrad_generic_partials) -- and, when db_part is
 *             not NULL, [sd_align_cgd_tok_tiles(B, P)][C] fp32 partial column sums of dY whose sum over the first axis is the bias gradient.
 * All jobs of a call share K; K in {64, 128, 256}, C % 32 == 0 (sd_align_cgd_tok_supported); operands 16-byte aligned.
 * sd_linear_tok_bf16_fwd is the same tile loop with Y stored (bf16): the stand-alone projection.
 */
typedef struct sd_align_tok_job {
    const void *X, *W;
    const float *bias;          /* [C] or NULL */
    const void *T;
    const int32_t *perm;        /* [C] device, or NULL */
    float *row_lse2;            /* [rows][2]: forward out, backward in */
    float *row_kl, *loss;       /* forward */
    const float *upstream;      /* backward: [1] or NULL */
    void *out;                  /* backward: dY [B*P][C] bf16 */
    float *db_part;             /* backward: [tiles][C] or NULL */
    void *workspace;            /* forward: >= sd_align_cgd_tok_workspace_bytes(B, C, P), 16-byte aligned, private to the job */
    size_t workspace_bytes;
    long P;
    int B, K, C, g;
    float inv_tau, loss_scale, coef;
    int reserved;
} sd_align_tok_job;
int sd_align_cgd_tok_supported(int in_channels, int out_channels);
int sd_align_cgd_tok_tiles(int B, long P);
size_t sd_align_cgd_tok_workspace_bytes(int B, int C, long P);
int sd_align_cgd_tok_fwd_multi(const sd_align_tok_job *jobs, int njobs, void *stream);
int sd_align_cgd_tok_bwd_multi(const sd_align_tok_job *jobs, int njobs, void *stream);
int sd_linear_tok_bf16_fwd(const void *X, const void *W, const float *bias, void *Y, long tokens, int in_features, int out_features, void *stream);
/* dX [tokens][in] = dY [tokens][out] . W [out][in] (bf16 in and out, fp32 accumulation): the projection's input gradient (autograd's
 * `grad_output.mm(weight)` of the same Linear); the same shapes as above (sd_align_cgd_tok_supported(in, out)). */
int sd_linear_tok_bf16_bwd_data(const void *dY, const void *W, void *dX, long tokens, int out_features, int in_features, void *stream);

/* ---------------------------------------------------------------------------
 * nn.Linear on token-major activations under bf16 storage (csrc/tok_gemm_bf16.hip):
 *   forward   Y [tokens][out] = X [tokens][in] . W[out][in]^T + bias[out]        X, W, Y bf16; bias fp32 or bf16 (bias_dtype) or NULL
 * fp32 accumulation, ONE rounding of (accumulator + bias) to bf16 -- what autocast's F.linear computes.  Replaces the F.linear of every q / kv /
 * proj / fc1 / fc2 of the MiT encoders (mix_transformer.py:24-27,48-55,75-84,107-133), the SR convolution as a patch GEMM (:66-70) and the MLP
 * projections of the SegFormer head (segformer_head.py:22-33) in BASELINE config 5.  _supported(): in % 64 == 0, out % 64 == 0, tokens * in and
 * out * in below 2^31 elements; operands 16-byte aligned (SD_E_ALIGN), rows dense.  Tunable (A/B): tok_gemm_bf16_variant (index of a tile /
 * wave-layout / ring-depth instantiation, csrc/tok_gemm_bf16.hip::launch_variant; -1 = by shape).
 */
int sd_linear_bf16_fwd_supported(long tokens, int in_features, int out_features);
int sd_linear_bf16_fwd(const void *X, const void *W, const void *bias, int bias_dtype, void *Y, long tokens, int in_features, int out_features,
                       void *stream);

/* ---------------------------------------------------------------------------
 * nn.Linear on token-major activations, forward and input gradient, as exact-f32 MFMA GEMMs (csrc/token_gemm.hip):
 *   forward   Y [tokens][out] = act( X [tokens][in] . W[out][in]^T + bias[out] ) (+ residual [tokens][out])
 *   bwd-data  dX [tokens][in] = dY [tokens][out] . W[out][in]
 * Replaces the F.linear / autograd mm of every Linear of the MiT encoders (q, kv, proj, fc1, fc2: mix_transformer.py:24-27,48-55,
 * 75-84,107-133) and of the SegFormer head (MLP.proj, the per-branch blocks of linear_fuse, linear_pred: segformer_head.py:22-33,
 * 75-98).  dtype: SD_F32 only (bf16 activations stay on the library's bf16 GEMM).  act: 0 = none, 1 = exact (erf) GELU.
 * bias / residual may be NULL.  X, Y, dY, dX dense row-major; W rows `w_row_stride` elements apart (0 = in_features; a column block of a wider
 * matrix -- the per-branch blocks of linear_fuse -- is passed without a copy).  16-byte loads are used when rows are 16-byte aligned; any shape
 * is accepted.
 * mode: 0 = v_mfma_f32_32x32x2_f32 (bit-equal to an fp32 fmaf chain); 1 = split-bf16 ("bf16x3"): every fp32 operand is split exactly into
 * three bf16 terms and a product is the sum of the six bf16 products of weight >= 2^-16 on v_mfma_f32_32x32x16_bf16 (fp32 accumulation) --
 * the same error level as mode 0 (dropped terms <= 2^-24 relative; tests/test_token_gemm_gpu.py holds both modes to one bound) at 12
 * instead of 32 matrix-pipe cycles per k.  mode 1: act == 0 only.
 */
int sd_linear_fwd(const void *X, const float *W, long w_row_stride, const float *bias, const void *residual, void *Y, int dtype, long tokens,
                  int in_features, int out_features, int act, int mode, void *stream);
int sd_linear_bwd_data(const void *dY, const float *W, long w_row_stride, void *dX, int dtype, long tokens, int in_features, int out_features,
                       int mode, void *stream);

/* ---------------------------------------------------------------------------
 * Depth-wise 3x3 convolution (stride 1, zero pad 1) on TOKEN-MAJOR activations [B, H*W, C]: the
 * DWConv inside every MiT Mix-FFN (mix_transformer.py:376-387: transpose to NCHW ->
 * nn.Conv2d(dim, dim, 3, 1, 1, groups=dim) -> flatten/transpose back; called from Mlp.forward :48-55).
 * Replaces those two transposes plus the grouped conv and its two backward convs.  Weights are passed, and
 * weight gradients returned, in nn.Conv2d's own layout [C][9] fp32 (= [C,1,3,3] contiguous, k = 3*ky + kx):
 * no transpose on either side of the boundary; C % 4 == 0 (fp32) / C % 8 == 0 (bf16), 16-byte aligned pointers;
 * SD_E_SHAPE when (W + 2) * C >= 2^31 (offsets inside an image row are 32-bit).
 * sd_dwconv3x3_bwd_weight with dw == NULL leaves its partials [slabs][9*C + C] (weight gradient in conv layout,
 * then the C bias sums) in the workspace for sd_multi_slab_reduce; slabs = sd_dwconv3x3_wgrad_slabs(...).
 */
size_t sd_dwconv3x3_workspace_bytes(int dtype, int B, int H, int W, int C);
int sd_dwconv3x3_wgrad_slabs(int dtype, int B, int H, int W, int C);
/* the partials of MANY depth-wise convolutions in ONE launch per 24 jobs (round 5: the filter gradients of a backward, deferred to its end):
 * job.partials receives what sd_dwconv3x3_bwd_weight(dw = NULL) leaves in its workspace -- [sd_dwconv3x3_wgrad_slabs()][9*C + C] floats
 * (partials_bytes >= that) -- for sd_multi_slab_reduce. */
typedef struct sd_dw_wgrad_job {
    const void *x, *dy;         /* [B, H*W, C] tokens of the conv's input / of the gradient of its output */
    float *partials;
    size_t partials_bytes;
    int B, H, W, C;
} sd_dw_wgrad_job;
int sd_dwconv3x3_wgrad_multi(const sd_dw_wgrad_job *jobs, int njobs, int dtype, void *stream);

int sd_dwconv3x3_fwd(const void *x, const float *w, const float *bias /* or NULL */, void *y,
                     int dtype, int B, int H, int W, int C, void *stream);

/* inference-only: y = GELU(dwconv(x) + bias) (the erf GELU of nn.GELU(); erfc evaluated by one branch-free degree-6 form, f32 result within
 * 3.9e-7 of the f64 value over [-12, 12], see csrc/dwconv.hip::gelu_erf), i.e. DWConv followed by the Mix-FFN activation
 * (mix_transformer.py:50-51) in one pass; used for the frozen teacher, which never needs the pre-activation */
int sd_dwconv3x3_gelu_fwd(const void *x, const float *w, const float *bias /* or NULL */, void *y,
                          int dtype, int B, int H, int W, int C, void *stream);

/* training form of the same: also keeps the pre-activation y_pre = dwconv(x) + bias, which the GELU backward needs */
int sd_dwconv3x3_gelu_fwd_train(const void *x, const float *w, const float *bias /* or NULL */, void *y_pre, void *y,
                                int dtype, int B, int H, int W, int C, void *stream);

int sd_dwconv3x3_bwd_data(const void *dy, const float *w, void *dx,
                          int dtype, int B, int H, int W, int C, void *stream);

int sd_dwconv3x3_bwd_weight(const void *x, const void *dy, float *dw /* [C][9], or NULL: partials only */, float *dbias /* or NULL */,
                            int dtype, int B, int H, int W, int C,
                            void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------
 * Supervised cross-entropy of the student with the bilinear up-sampling of the logits fused in
 * (SURVEY.md 8f rank 1).  Replaces decode_head.py:217-237 (resize to label size -> F.cross_entropy
 * (reduction='none', ignore_index) of cross_entropy_loss.py:9-32 -> accuracy of accuracy.py:4-49)
 * and the corresponding autograd chain.  logits [B,C,h,w] in `dtype`; label [B,H,W] int32;
 * (H,W) = (F*h, F*w), F in {2,4,8}.
 *   fwd: loss_pix [B,H,W] (0 on ignored pixels), pix_lse2 [B,H,W] (base-2 log-partition, for bwd),
 *        *correct = number of pixels whose arg-max class equals the label (top-1 hits).
 *   bwd: dlogits = g * (softmax - onehot) pulled back through the interpolation; g is either a
 *        [B,H,W] map (upstream_is_map=1) or one device float (0), both scaled by `gscale`.
 */
int sd_ce_up_supported(int h, int w, int H, int W);

int sd_ce_up_fwd(const void *logits, const int32_t *label, float *loss_pix, float *pix_lse2, int *correct,
                 int dtype, int B, int C, int h, int w, int H, int W, int ignore_index, void *stream);

int sd_ce_up_bwd(const void *logits, const int32_t *label, const float *pix_lse2,
                 const float *upstream, int upstream_is_map, float gscale, void *dlogits,
                 int dtype, int B, int C, int h, int w, int H, int W, int ignore_index, void *stream);

/* ---------------------------------------------------------------------------
 * LayerNorm over the channel axis of token-major activations [rows][C] (C % 4 == 0, C <= 1024).
 * Replaces the nn.LayerNorm calls of the MiT encoders (mix_transformer.py:96,139,169-171,214 and the
 * stage norms :336-365) and their autograd.  gamma/beta fp32; mean/rstd [rows] fp32 saved by the
 * forward for the backward; dgamma/dbeta via per-workgroup partials in `workspace` (deterministic).
 */
int sd_layernorm_supported(int C);
size_t sd_layernorm_workspace_bytes(long rows, int C);

int sd_layernorm_fwd(const void *x, const float *gamma, const float *beta, void *y, float *mean, float *rstd,
                     int dtype, long rows, int C, float eps, void *stream);

int sd_layernorm_bwd(const void *x, const void *dy, const float *gamma, const float *mean, const float *rstd,
                     void *dx, float *dgamma, float *dbeta, int dtype, long rows, int C,
                     void *workspace, size_t workspace_bytes, void *stream);

/* Inference form with row maps -- the Swin blocks of a frozen network (reference mmseg/models/backbones/swin_transformer.py:195-262).  Per image
 * (`images` of them) output row r in [0, rows_out) normalises
 *     v = x[xi] (+ res[ri]),   xi = x_map ? x_map[r] : r,   ri = res_map ? res_map[r] : xi        (x: rows_in rows, res: rows_res rows per image)
 * writes y[r] = LayerNorm(v) and, when res != NULL, xsum[xi] = v.  xi == rows_in marks a padding row: y[r] = 0, nothing else is touched.
 *   x_map = window-partition table: pad (:206-210) + cyclic shift (:213-215) + window_partition (:222) of norm1(x [+ pending residual]);
 *   res_map = its inverse: x + un-pad / un-shift / window_reverse (:233-247) of the attention output, then norm2 (:250).
 * Maps are int32, values in [0, rows_in] / [0, rows_res); x_map must hit every row of x at most once when res is given.  No saved statistics. */
int sd_layernorm_map_fwd(const void *x, const void *res /* or NULL */, void *xsum /* or NULL */, const float *gamma, const float *beta, void *y,
                         const int32_t *x_map /* [rows_out] or NULL */, const int32_t *res_map /* [rows_out] or NULL */, int dtype, long images,
                         int rows_out, int rows_in, int rows_res, int C, float eps, void *stream);

/* Residual form: the encoder block's "x = x + drop_path(f(x))" (mix_transformer.py:150-151; timm DropPath: per-sample
 * factor mask/keep_prob) fused with the LayerNorm that consumes the sum (the block's norm2, the next block's norm1 or the
 * stage norm :336-365):
 *   fwd   xsum = x + s * res,  y = LayerNorm(xsum);   s = row_scale[row / rows_per_sample] (row_scale NULL -> 1)
 *   bwd   dx = dLayerNorm(dy) + dres (dres NULL -> 0): gradient of x;  dr = s * dx: gradient of res (dr NULL when row_scale is
 *         NULL -- then dx is also the gradient of res);  dgamma / dbeta as sd_layernorm_bwd.  Same workspace size. */
int sd_add_layernorm_fwd(const void *x, const void *res, const float *row_scale /* [rows/rows_per_sample] or NULL */,
                         long rows_per_sample, void *xsum, const float *gamma, const float *beta, void *y,
                         float *mean, float *rstd, int dtype, long rows, int C, float eps, void *stream);

int sd_add_layernorm_bwd(const void *xsum, const void *dy, const float *gamma, const float *mean, const float *rstd,
                         const void *dres /* or NULL */, const float *row_scale /* or NULL */, long rows_per_sample,
                         void *dx, void *dr /* or NULL */, float *dgamma, float *dbeta, int dtype, long rows, int C,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------
 * MiT spatial-reduction attention (mix_transformer.py:107-133: q Linear :75, kv Linear :76 on the sr-conv reduced map
 * :112-116, softmax(scale * q k^T) v :119-123), forward and backward, all heads in one launch.
 *   q    [B, N, heads*D]      the q Linear output as it is
 *   kv   [B, KV, 2*heads*D]   the kv Linear output as it is (inner order: k|v, head, D -- the reference's reshape :116)
 *   out  [B, N, heads*D]      what the proj Linear consumes (the reference's transpose(1,2).reshape :123)
 *   lse  [B, heads, N] fp32   base-2 log-sum-exp of the scaled scores (saved for the backward)
 * head_dim D in {32, 64}; any N; KV <= 256 (K and V of a head are staged in LDS; KV = 256 at 512x512).  Arithmetic: fp32 storage --
 * split-bf16 products on the bf16 matrix pipe (three exact bf16 terms per operand, six cross products, fp32 accumulation: the error
 * bound of the exact v_mfma_f32_32x32x2_f32 kernels, which remain behind tunable sra_split_bf16 = 0 and serve the head_dim-64
 * backward); bf16 storage -- forward and backward on the bf16 matrix pipe with P / dS rounded to bf16 for the second-stage products
 * (tunable sra_bf16_mfma; 0 = the f32-input MFMA kernels on the widened values).  dkv has kv's layout.
 */
int sd_sra_supported(int head_dim);
size_t sd_sra_workspace_bytes(int B, int N, int KV, int heads, int D);   /* backward only */

int sd_sra_fwd(const void *q, const void *kv, void *out, float *lse, int dtype,
               int B, int N, int KV, int heads, int D, float scale, void *stream);

int sd_sra_bwd(const void *q, const void *kv, const void *out, const void *dout, const float *lse,
               void *dq, void *dkv, int dtype, int B, int N, int KV, int heads, int D, float scale,
               void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------
 * SegFormer head: y = z1 + up(z2) + up(z3) + up(z4) + bias on token-major tensors, one pass.
 * Replaces the three resize() calls, the torch.cat and (together with the per-branch fuse GEMMs done
 * by the binding) the linear_fuse 1x1 conv input path of segformer_head.py:82-91.
 * z1 [B,H*W,E]; z_i [B,(H/f_i)*(W/f_i),E], f_i in {2,4,8}; bias [E] fp32 or NULL; y [B,H*W,E].
 * sd_upsum_bwd: gradient w.r.t. ONE coarse branch, dz [B,h*w,E] from dy [B,(F*h)*(F*w),E]
 * (transposed bilinear interpolation as a gather; dz1 = dy needs no kernel).
 */
int sd_upsum_fwd(const void *z1, const void *z2, const void *z3, const void *z4, const float *bias, void *y,
                 int dtype, int B, int H, int W, int E, int f2, int f3, int f4, void *stream);

/* Inference form for a frozen network: the eval-mode BatchNorm (an affine map per channel: scale = gamma / sqrt(var + eps),
 * shift = beta - mean * scale) and the ReLU that follow the sum in `linear_fuse` (segformer_head.py:66-71, :93-94) applied in
 * the same pass: y = max(0, (z1 + up(z2) + up(z3) + up(z4) + bias) * scale + shift)  (relu = 0: no clamp). */
int sd_upsum_affine_fwd(const void *z1, const void *z2, const void *z3, const void *z4, const float *bias /* or NULL */,
                        const float *scale, const float *shift, int relu, void *y, int dtype,
                        int B, int H, int W, int E, int f2, int f3, int f4, void *stream);

int sd_upsum_bwd(const void *dy, void *dz, int dtype, int B, int h, int w, int E, int F, void *stream);

/* All three coarse branches of the SegFormer geometry (factors 2, 4, 8 of an H x W fine grid) from ONE read of dy, in two separable
 * passes (row reduction into fp32 partials in the workspace, then column reduction): dz2 [B, (H/2)(W/2), E], dz3 [B, (H/4)(W/4), E],
 * dz4 [B, (H/8)(W/8), E].  SD_E_UNSUPPORTED unless H % 8 == W % 8 == 0, E % 64 == 0, W <= 512. */
size_t sd_upsum_bwd3_workspace_bytes(int B, int H, int W, int E);
int sd_upsum_bwd3(const void *dy, void *dz2, void *dz3, void *dz4, int dtype, int B, int H, int W, int E, void *workspace, size_t workspace_bytes,
                  void *stream);

/* ---------------------------------------------------------------------------
 * Training-mode BatchNorm over token-major activations [rows][C] (C % 4 == 0, C <= 1024) with the ReLU and the channel
 * dropout that follow it fused in.  Replaces the `linear_fuse` tail of the SegFormer head -- (Sync)BatchNorm -> ReLU
 * (segformer_head.py:66-71,93-94) -> Dropout2d (decode_head.py:210-215) -- and its autograd; the statistics and the
 * backward sums are separate entry points because with more than one rank a collective sits behind each of them
 * (torch.nn.SyncBatchNorm's all-gather of mean / invstd / count and all-reduce of the dy sums).
 *   sd_bn_stats            mean[C], invstd[C] = 1/sqrt(biased var + eps) of THIS rank's rows; when running_mean / running_var
 *                          are given they are updated in place (momentum; unbiased variance) -- the single-rank case.
 *   sd_bn_act_fwd          y = act((x - mean) * invstd * weight + bias) * drop_scale[row / rows_per_image][c]
 *                          (relu = 0: act = identity; weight / bias / drop_scale may be NULL: 1 / 0 / 1).
 *   sd_bn_act_bwd_reduce   g = dy * drop_scale * [pre-activation > 0];  sum_dy[c] = sum_rows g,
 *                          sum_dy_xmu[c] = sum_rows g * (x - mean)       (grad_bias = sum_dy, grad_weight = sum_dy_xmu * invstd)
 *   sd_bn_act_bwd_elemt    dx = (g - sum_dy * inv_count - (x - mean) * invstd^2 * sum_dy_xmu * inv_count) * invstd * weight,
 *                          inv_count = 1 / (rows summed over all ranks) -- as a host value or, when the total is only known on
 *                          the device (ranks with unequal batches), through inv_count_dev; sum_dy / sum_dy_xmu summed over all ranks.
 * Per-workgroup partials in `workspace` (sd_bn_workspace_bytes), combined in fp64: deterministic, no float atomics.
 */
int sd_bn_supported(int C);
size_t sd_bn_workspace_bytes(long rows, int C);

int sd_bn_stats(const void *x, int dtype, long rows, int C, float eps, float *mean, float *invstd,
                float *running_mean /* or NULL */, float *running_var /* or NULL */, float momentum,
                void *workspace, size_t workspace_bytes, void *stream);

int sd_bn_act_fwd(const void *x, const float *mean, const float *invstd, const float *weight, const float *bias,
                  const float *drop_scale, long rows_per_image, int relu, void *y, int dtype, long rows, int C, void *stream);

int sd_bn_act_bwd_reduce(const void *x, const void *dy, const float *mean, const float *invstd, const float *weight,
                         const float *bias, const float *drop_scale, long rows_per_image, int relu,
                         float *sum_dy, float *sum_dy_xmu, int dtype, long rows, int C,
                         void *workspace, size_t workspace_bytes, void *stream);

int sd_bn_act_bwd_elemt(const void *x, const void *dy, const float *mean, const float *invstd, const float *weight,
                        const float *bias, const float *drop_scale, long rows_per_image, int relu,
                        const float *sum_dy, const float *sum_dy_xmu, float inv_count,
                        const float *inv_count_dev /* or NULL; when given it replaces inv_count */, void *dx,
                        int dtype, long rows, int C, void *stream);

/* ---------------------------------------------------------------------------
 * Deferred combination of partial slabs.  The parameter-gradient kernels (sd_layernorm_bwd / sd_add_layernorm_bwd with
 * dgamma == dbeta == NULL, sd_linear_wgrad_partials) leave their per-workgroup partials in the caller's workspace instead
 * of launching their own combine pass; nothing reads a parameter gradient before the optimizer, so the caller collects the
 * jobs of a whole backward and combines them in ONE launch per 80 jobs:  out[i] = sum_{s < nslabs} partials[s*n + i].
 * `jobs` is a HOST array (its contents travel as kernel arguments: nothing is copied, safe under hipGraph capture).
 *   sd_layernorm_bwd_blocks(rows, C)           slabs the LayerNorm backward leaves: partials [nblk][2][C] (dgamma row, dbeta row)
 *   sd_linear_wgrad_slabs(dtype, T, M, N)      slabs of the tall-skinny weight-gradient plan (0: the shape does not take that plan);
 *                                              a slab is M*N floats, followed by M bias partials when with_bias != 0
 */
typedef struct sd_reduce_job {
    const float *partials;
    float *out;
    long n;
    int nslabs;
    int reserved;
} sd_reduce_job;

int sd_multi_slab_reduce(const sd_reduce_job *jobs, int njobs, void *stream);

/* Column sums of a token-major matrix x [rows][C] (C % 4 == 0): the bias gradient of a Linear / 1x1 conv, db = sum_t dY[t].
 * Leaves partials [sd_colsum_blocks(rows, C)][C] for sd_multi_slab_reduce (n = C). */
typedef struct sd_colsum_job {
    const void *x;        /* [rows][C], all jobs of one call in the same dtype */
    float *partials;      /* [sd_colsum_blocks(rows, C)][C] */
    long rows;
    int C;
    int reserved;
} sd_colsum_job;

int sd_colsum_blocks(long rows, int C);
/* the partials of MANY matrices in one launch per 80 jobs (host array, by-value kernel arguments like sd_multi_slab_reduce) */
int sd_multi_colsum_partials(const sd_colsum_job *jobs, int njobs, int dtype, void *stream);
int sd_colsum_partials(const void *x, int dtype, long rows, int C, float *partials, size_t partials_bytes, void *stream);
int sd_layernorm_bwd_blocks(long rows, int C);
int sd_linear_wgrad_slabs(int dtype, long tokens, int out_features, int in_features);
int sd_linear_wgrad_partials(const void *dY, const void *X, int dtype, long tokens, int out_features, int in_features,
                             int with_bias, void *workspace, size_t workspace_bytes, void *stream);
/* The same split between kernel and combine for sd_linear_wgrad's GENERIC plan (not tall-skinny; bf16 storage mostly): _slabs() = number of
 * [out x in] fp32 slabs (0: un-split or direct plan -- use the calls above), _partials() writes them, the caller sums them. */
/* fp32 token-major Linear weight gradient as split-K slabs in split-bf16 arithmetic on hardware-transposed LDS reads (csrc/wgrad_tn.hip;
 * the tall-skinny products of the SegFormer head: 256 x {256, 160, 64, 32} over 131072 tokens).  _slabs(): number of [out x in] fp32 slabs for
 * this shape, 0 when the shape is not this kernel's (use sd_linear_wgrad / _splitk).  with_bias != 0: every slab is [out x in] followed by `out`
 * column sums of dY over the split's tokens (the bias gradient, fp32 sums of the staged values: no second pass over dY), out * in + out floats. */
int sd_linear_wgrad_tn_slabs(long tokens, int out_features, int in_features);
int sd_linear_wgrad_tn(const float *dY, const float *X, float *slabs, size_t slabs_bytes, long tokens, int out_features, int in_features, int with_bias,
                       void *stream);
/* MANY weight gradients in ONE launch, their k-splits planned together (csrc/wgrad_tn.hip, round 5): the weight gradients of a backward
 * (autograd's `grad_output.t().mm(input)` of every token-major Linear; same reference lines as above), deferred to its end.  dtype = the storage
 * type of dY and X: SD_BF16 -> the LDS-DMA ring kernel; SD_F32 -> split-bf16 arithmetic (wgrad_tn_x3; one launch per tile width 128 / 64 / 32).
 *   _supported(): out % 8 == in % 8 == 0; bf16: tokens % 32 == 0 and tokens >= 96; operands 16-byte aligned.
 *   _plan():  fills job.nsplit (>= 1) for ALL jobs of the coming call: ~1536 (bf16) / ~3072 (fp32) workgroups in total, dealt by tile-k-steps;
 *             bf16 within the "wgrad_slab_ratio" cap.  SD_E_UNSUPPORTED if a job is not _supported (nothing is filled then).
 *   launch:   job.slabs = nsplit slabs of out * in floats (+ out floats when with_bias: the column sums of dY over the slab's tokens = the bias
 *             gradient): slab z = the product over the z-th token range; the caller sums them (sd_multi_slab_reduce, fixed order:
 *             run-to-run identical).  nsplit == 1: `slabs` IS the gradient (point it at the destination, no combine).  One launch per 32 jobs. */
typedef struct sd_wgrad_job {
    const void *dY, *X;         /* [tokens][out], [tokens][in] */
    float *slabs;
    long tokens;
    int out_features, in_features;
    int nsplit;                 /* written by _plan, read by the launch */
    int with_bias;
} sd_wgrad_job;
int sd_linear_wgrad_tn_multi_supported(int dtype, long tokens, int out_features, int in_features);
int sd_linear_wgrad_tn_multi_plan(sd_wgrad_job *jobs, int njobs, int dtype);
int sd_linear_wgrad_tn_multi(const sd_wgrad_job *jobs, int njobs, int dtype, void *stream);
int sd_linear_wgrad_generic_slabs(int dtype, long tokens, int out_features, int in_features);
int sd_linear_wgrad_generic_partials(const void *dY, const void *X, int dtype, long tokens, int out_features, int in_features, void *workspace,
                                     size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------
 * Split-bf16 Linear products with the WEIGHT operand split beforehand (round 3).  The mode-1 kernels above split both operands in
 * registers inside the multiply loop; a weight changes once per optimizer step (never, for the frozen teacher), so its three bf16 planes are
 * written once -- sd_presplit_multi: ONE launch for any number of weights -- in the order the matrix cores consume them (one contiguous
 * 1 KB run per 32-column block, 16-deep k-step and plane) and the GEMM loads them straight into registers: no LDS and no vector
 * arithmetic for that operand.  Same six bf16 products, same planes bit for bit, hence the same results as mode 1
 * (tests/test_token_gemm_gpu.py).  Replaces the same reference ops as sd_linear_fwd / sd_linear_bwd_data (mix_transformer.py:24-27,48-55,
 * 75-84,107-133; segformer_head.py:22-33,75-98).
 *   sd_presplit_bytes(n_cols, k_depth)  bytes of one planes buffer: forward planes (out_features, in_features), bwd planes (in_features,
 *                                       out_features); k_depth % 16 == 0, else 0.  Buffers 16-byte aligned, caller-owned.
 *   job: W [out_features][w_row_stride >= in_features] fp32; fwd_planes and / or bwd_planes (NULL = not wanted)
 *   sd_linear_fwd_planes       Y  [tokens][out] = X [tokens][in] . W^T + bias (+ residual); in_features % 32 == 0, X rows 16-byte aligned
 *   sd_linear_bwd_data_planes  dX [tokens][in]  = dY [tokens][out] . W;                     out_features % 32 == 0
 */
typedef struct sd_presplit_job {
    const float *W;
    long w_row_stride;
    int out_features, in_features;
    void *fwd_planes;
    void *bwd_planes;
    void *row_planes;   /* [3][out_features][in_features] bf16, row-major: the A operand of sd_linear_nchw_fwd_planes (in_features % 8 == 0) */
} sd_presplit_job;
size_t sd_presplit_bytes(int n_cols, int k_depth);
size_t sd_presplit_rows_bytes(int out_features, int in_features);
/* sd_linear_nchw_fwd with the weight's pre-split row planes (fp32 storage, out_features <= 160, in_features % 32 == 0): the class-plane
 * linear_pred of the SegFormer head (segformer_head.py:73,96) whose 160-row tile needs every weight row in every wave -- its A fragments become
 * plain LDS reads of the planes instead of 5/6 of the kernel's split arithmetic. */
int sd_linear_nchw_fwd_planes(const void *X, const void *w_row_planes, const float *bias, void *Y, int dtype, int B, long P, int in_features,
                              int out_features, void *stream);
int sd_presplit_multi(const sd_presplit_job *jobs, int njobs, void *stream);
int sd_linear_fwd_planes(const void *X, const void *fwd_planes, const float *bias, const void *residual, void *Y, int dtype, long tokens,
                         int in_features, int out_features, void *stream);
int sd_linear_bwd_data_planes(const void *dY, const void *bwd_planes, void *dX, int dtype, long tokens, int in_features, int out_features,
                              void *stream);

/* ---------------------------------------------------------------------------
 * Weight gradient of a token-major nn.Linear for the shapes sd_linear_wgrad's tall-skinny plan does not take (fewer than 8192 tokens, or a
 * weight of more than 16 regions of 64 x 64: the stage 3-4 Linears of the MiT encoders, mix_transformer.py:24-27,75-84): split-K over the
 * tokens on the pipelined MFMA kernel (round 3; replaces the library's `dY^T @ X`, which ran these on ~100 workgroups).  fp32 only.
 *   sd_linear_wgrad_splitk_slabs  number of [out x in] slabs the launch writes (0: shape not supported -- out, in must be multiples of 4)
 *   sd_linear_wgrad_splitk        slabs[z] = dY[chunk z]^T . X[chunk z]; the caller sums them (sd_multi_slab_reduce), at once or deferred
 */
int sd_linear_wgrad_splitk_slabs(long tokens, int out_features, int in_features);
int sd_linear_wgrad_splitk(const float *dY, const float *X, float *slabs, size_t slabs_bytes, long tokens, int out_features, int in_features,
                           void *stream);

/* ---------------------------------------------------------------------------
 * A Linear from TOKEN-MAJOR features to contiguous class planes (fp32): the SegFormer head's `linear_pred` 1x1 conv
 * (mmseg/models/decode_heads/segformer_head.py:73,96) on the token-major fused feature map, with the logits landing directly in the
 * [B, out_features, P] planes the loss kernels read (sd_ce_up_*, sd_cgd_kl_up_*) and the gradient read from such planes: no
 * [B, P, out] <-> [B, out, P] transpose copy in either direction (2 x 79 MB per step at config 2).
 *   X  [B, P, in_features]   W [out_features, in_features]   bias [out_features] or NULL   Y / dY [B, out_features, P]
 * X, Y, dY, dX in `dtype` storage; W, bias, dW, dbias always fp32 (master weights).  Arithmetic: fp32 storage -- split-bf16 (tunable
 * align_split_bf16) or exact f32 MFMA; bf16 storage -- bf16 MFMA with fp32 accumulation (the master weight is rounded on its way into LDS).
 * bwd_weight: deterministic slab combine; dbias may be NULL.
 */
size_t sd_linear_nchw_workspace_bytes(int B, long P, int in_features, int out_features);
int sd_linear_nchw_fwd(const void *X, const float *W, const float *bias, void *Y, int dtype, int B, long P, int in_features, int out_features,
                       void *stream);
int sd_linear_nchw_bwd_data(const void *dY, const float *W, void *dX, int dtype, int B, long P, int in_features, int out_features, void *stream);
int sd_linear_nchw_bwd_weight(const void *dY, const void *X, float *dW, float *dbias, int dtype, int B, long P, int in_features, int out_features,
                              void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------
 * Eval-mode BatchNorm (+ residual add) (+ ReLU) of a frozen convolutional network, one in-place pass over a contiguous NCHW fp32 map (round 6):
 *   y[b, c, p] = act(x[b, c, p] * scale[c] + shift[c] (+ residual[b, c, p])),  scale = gamma / sqrt(running_var + eps), shift = beta - mean * scale.
 * Replaces `relu(bn(conv(x)))` and `out += identity; relu(out)` of the reference's BasicBlock / Bottleneck (backbones/resnet.py:18-100) and the
 * norm -> act of ConvModule (psp_head.py:38-44,84-91; uper_head.py:30-75) for a teacher in eval mode: three launches and passes become one.
 * planes = B * C; y may alias x (in place) or residual; residual NULL = none.
 */
int sd_affine_act_nchw(const float *x, const float *residual, float *y, const float *scale, const float *shift, long planes, int C, long HW, int relu,
                       void *stream);

/* ---------------------------------------------------------------------------
 * The second half of a frozen Mix-FFN in one pass (round 6):  y = fc2(GELU(dwconv3x3(h) + dw_bias)) + fc2_bias  on token-major fp32 activations,
 * h [B, H*W, hidden] (fc1's output) -> y [B, H*W, out_features].  Replaces `x = self.dwconv(x, H, W); x = self.act(x); x = self.fc2(x)` of the
 * reference's Mlp (mix_transformer.py:20-55) for a network in eval mode without autograd (the teacher): the activated hidden map -- the block's
 * largest tensor -- is never written.  Convolution and GELU as sd_dwconv3x3_gelu_fwd.  dtype SD_F32: h, y fp32, the product in split-bf16 arithmetic as
 * sd_linear_fwd; SD_BF16 (the network under bf16 autocast): h, y bf16, the activated map and fc2's weight rounded to bf16 as the two-kernel route does,
 * fp32 accumulation + bias, one rounding.  dw_weight [hidden][9] (nn.Conv2d's [hidden, 1, 3, 3]), fc2_weight [out_features][hidden], biases: fp32;
 * everything 16-byte aligned (SD_E_ALIGN).
 * _supported(): H % 8 == 0, W % 16 == 0, hidden % 32 == 0, out_features 64 or 128, H * W * hidden < 2^30; else SD_E_UNSUPPORTED (run the two
 * kernels).
 */
int sd_mixffn_tail_supported(int H, int W, int hidden, int out_features);
int sd_mixffn_tail(const void *h, const float *dw_weight, const float *dw_bias, const float *fc2_weight, const float *fc2_bias, void *y, int dtype, int B,
                   int H, int W, int hidden, int out_features, void *stream);

/* ---------------------------------------------------------------------------
 * The tail of a frozen SegFormer head in one pass (round 6), fp32:
 *   logits [B, classes, H, W] = W_p . relu(scale * (z1 + up2(z2) + up4(z3) + up8(z4) + fuse_bias) + shift) + pred_bias
 * z1 [B, H*W, E], z2 [B, (H/2)(W/2), E], z3 [B, (H/4)(W/4), E], z4 [B, (H/8)(W/8), E]: the per-branch results of the fuse conv, token-major
 * (what sd_upsum_affine_fwd takes); scale / shift [E]: the eval-mode BatchNorm of `linear_fuse` as an affine map; pred_row_planes: `linear_pred`'s
 * weight [classes, E] pre-split by sd_presplit_multi (row_planes, sd_presplit_rows_bytes).  Replaces, for a head in eval mode without autograd
 * (the teacher), `_c = self.linear_fuse(torch.cat([...], dim=1)); x = self.dropout(_c); x = self.linear_pred(x)` (segformer_head.py:93-96): the
 * summed, normalised map [B, H*W, E] is never written.  Interpolation / sum / affine / ReLU as sd_upsum_affine_fwd, the product as
 * sd_linear_nchw_fwd_planes.  _supported(): H % 8 == 0, W % 32 == 0, E % 32 == 0, E <= 1024, classes <= 160, H * W * E < 2^30.
 */
int sd_head_tail_supported(int H, int W, int E, int classes);
int sd_head_tail_f32(const float *z1, const float *z2, const float *z3, const float *z4, const float *fuse_bias /* or NULL */, const float *scale,
                     const float *shift, const void *pred_row_planes, const float *pred_bias /* or NULL */, float *logits, int B, int H, int W, int E,
                     int classes, void *stream);

/* ---------------------------------------------------------------------------
 * Overlapping patch embedding as window gather + token GEMM (round 6).
 * Replaces the nn.Conv2d of OverlapPatchEmbed (mix_transformer.py:185-215: kernel 7 / stride 4 / pad 3 for stage 1, 3 / 2 / 1 for
 * stages 2-4, followed by flatten(2).transpose(1, 2)): MIOpen's filter-gradient kernels for these shapes accumulate with float atomics and its
 * deterministic mode (the reference always launches with --deterministic, tools/dist_train.sh:8) falls back to naive kernels, 11x the step.
 * sd_im2col_tokens gathers the k x k windows of a map x (logical [B, Cin, H, W] with ELEMENT strides sb, sc, sy, sx -- a channels-last view of
 * tokens or a contiguous NCHW image) into col [B * Ho * Wo][Kp], column (ky * k + kx) * Cin + ci (the order of a channels-last filter),
 * Kp = k * k * Cin rounded up to a multiple of 8, the surplus columns and the out-of-image taps zero; Ho = (H + 2 pad - k) / stride + 1.
 * The three products then run on the token-major Linear entry points (sd_linear_fwd* / sd_linear_bwd_data* / sd_linear_wgrad_tn_multi).
 * sd_col2im_tokens is the transposed gather (the convolution's input gradient from dcol): dx [B][H][W][Cin] contiguous, every element the
 * fp32 sum, in fixed order, of the <= ceil(k / stride)^2 windows covering it; needs Cin % (16 / element size) == 0 (SD_E_UNSUPPORTED otherwise).
 * col / dcol / dx 16-byte aligned.  Deterministic, no atomics, no workspace.
 */
int sd_im2col_tokens(const void *x, void *col, int dtype, int B, int H, int W, int Cin, long sb, long sc, long sy, long sx, int k, int stride, int pad,
                     int Ho, int Wo, int Kp, void *stream);
int sd_col2im_tokens(const void *dcol, void *dx, int dtype, int B, int H, int W, int Cin, int k, int stride, int pad, int Ho, int Wo, int Kp, void *stream);

/* ---------------------------------------------------------------------------
 * Bilinear resize of contiguous NCHW maps, forward and backward (round 3).
 * Replaces F.interpolate(x, size, mode='bilinear', align_corners) as the reference's networks call it through
 * mmseg/ops/wrappers.py:6-28 -- PSPHead's pooled branches (psp_head.py:52-58), UPerHead's top-down path and level fusion
 * (uper_head.py:101-121), `resize_concat` (decode_head.py:130-139) -- and KLDLoss.resize (losses.py:25-33) for the sizes the fused
 * up-sample kernels do not take.  Any input / output size, align_corners 0 | 1, ATen's index arithmetic.  planes = B * C.
 * Backward: deterministic separable gather (no atomics): din[y][x] = sum of weight * dout over the outputs whose taps touch (y, x), rows first
 * (fp32 workspace of sd_resize_bilinear_bwd_workspace_bytes(planes, h, W) bytes, 16-byte aligned), then columns.
 */
int sd_resize_bilinear_fwd(const void *in, void *out, int dtype, long planes, int h, int w, int H, int W, int align_corners, void *stream);
size_t sd_resize_bilinear_bwd_workspace_bytes(long planes, int h, int W);
int sd_resize_bilinear_bwd(const void *dout, void *din, int dtype, long planes, int h, int w, int H, int W, int align_corners, void *workspace,
                           size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------
 * LayerNorm / residual-add + LayerNorm (the two entry points above) whose output feeds a spatial-reduction attention (round 3): the
 * normalised tokens are ALSO written in the patch order [B, (H/r)(W/r), r*r*C] that the SR conv -- a Linear over non-overlapping r x r
 * patches, mix_transformer.py:75-84,121-124 -- multiplies, and the backward gathers the gradient that returns in patch order and adds it to
 * the token-order gradient.  Replaces, per SR block, one gather copy forward and one scatter copy + one add backward.  H, W, r powers of
 * two, r >= 2 (sd_layernorm_patch_supported).  res == NULL: plain LayerNorm (xsum, row_scale unused).  dy_patches may be NULL.
 */
int sd_layernorm_patch_supported(int H, int W, int r);
int sd_add_layernorm_patch_fwd(const void *x, const void *res, const float *row_scale, long rows_per_sample, void *xsum, const float *gamma,
                               const float *beta, void *y, void *y_patches, float *mean, float *rstd, int dtype, long rows, int C, float eps, int H,
                               int W, int r, void *stream);
int sd_add_layernorm_patch_bwd(const void *xsum, const void *dy, const void *dy_patches, const float *gamma, const float *mean, const float *rstd,
                               const void *dres, const float *row_scale, long rows_per_sample, void *dx, void *dr, float *dgamma, float *dbeta,
                               int dtype, long rows, int C, int H, int W, int r, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------
 * Optimizer step: AdamW over every trainable tensor in ONE launch.
 * Replaces torch.optim.AdamW.step() as the reference runs it through mmcv's OptimizerHook (mmseg/apis/train.py:89 builds the
 * optimizer; the KD configs use AdamW lr 6e-5, betas (0.9, 0.999), weight_decay 0.01 with paramwise_cfg lr / decay multipliers,
 * e.g. local_configs/exp_tab5/segformer_CGD+WS.py:60-64, psp_CD.py:70-74): fp32 parameters,
 * gradients and moments, arithmetic order of torch/optim/adamw.py::_single_tensor_adamw (decoupled decay first, bias-corrected step).
 *   tensors  device array of 56-byte descriptors {float *p; const float *g; float *m; float *v; uint16 *shadow; float wd;
 *            int32 group_missed; int64 n} (p, g, m, v -- and the optional bf16 shadow of p, NULL if none, rewritten with the new value --
 *            share one dense layout, the update is elementwise over storage; group_missed = parameter-group index in bits
 *            0-7, in bits 8-31 the number of optimizer steps the tensor took no part in: torch counts steps per tensor)
 *   group_lr HOST array of the parameter groups' current learning rates (ngroups <= sd_adamw_max_groups(); passed by value, so the
 *            schedule changes it every step without touching the tables)
 *   blocks   device array of {int32 tensor; int32 first_chunk}: one workgroup per sd_adamw_chunk() consecutive elements of a tensor
 *   step     the optimizer's step count INCLUDING this step (>= 1); bias corrections 1 - beta^(step - missed) are formed in fp64
 */
int sd_adamw_chunk(void);
int sd_adamw_max_groups(void);
int sd_adamw_multi(const void *tensors, const void *blocks, int nblocks, const float *group_lr, int ngroups, double beta1, double beta2, float eps,
                   int step, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SEGDISTILL_HIP_H */
