/*
 * diffews_hip.h -- C ABI of libdiffews_hip.so, the MI355X (gfx950) kernel library behind the
 * DiffewS hot path.
 *
 * The reference (ga1i13o/DiffewS) has no native code: every op below replaces a PyTorch /
 * diffusers-0.25 / xformers library call made from the reference's Python hot path.  Each entry
 * point cites the reference call site(s) it stands in for (paths relative to the reference root;
 * U = diffews/models/unet_2d_condition.py, A = diffews/models/attention_processor.py,
 * P = diffews/marigold_pipeline_rgb_latent_noise.py).
 *
 * Conventions:
 *   - the caller owns every buffer; kernels never allocate, free or synchronise;
 *   - pointers are raw device pointers; activations are NHWC / [rows][channels] row-major with an
 *     explicit element stride per row, images and latents at the pipeline boundary are NCHW fp32;
 *   - launches are asynchronous on the hipStream_t passed in (graph-capturable);
 *   - return value: 0 on success, a negative DFW_E* code for a rejected argument, or a positive
 *     hipError_t from the launch; no C++ exception crosses the boundary;
 *   - no global mutable state and nothing read from the environment, with ONE explicit exception: the process-wide
 *     tuning record of dfw_configure() below (defaults = the measured-best plans; meant for sweeps and A/B runs).
 */
#ifndef DIFFEWS_HIP_H
#define DIFFEWS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dfw_stream_t; /* hipStream_t */

enum { DFW_BF16 = 0, DFW_F16 = 1 };                       /* storage dtype of activations/weights */
enum { DFW_OUT_T = 0, DFW_OUT_F32 = 1, DFW_OUT_NCHW_F32 = 2 };
enum { DFW_ACT_NONE = 0, DFW_ACT_SILU = 1, DFW_ACT_CLAMP1 = 2 /* clamp to [-1, 1]: decode_seg, P:903 */ };
enum {
  DFW_EINVAL = -1,   /* null pointer / non-positive size */
  DFW_ESHAPE = -2,   /* shape not supported by the kernel (alignment / multiple-of constraints) */
  DFW_ERANGE = -3,   /* tensor exceeds the 2 GiB buffer-descriptor range */
  DFW_EWORKSPACE = -4
};

int dfw_version(void);
const char* dfw_error_string(int code);

/* Process-wide kernel-plan switches.  Every field's 0 value is NOT special: pass a record obtained from
 * dfw_get_config() with the fields of interest changed.  dfw_configure(NULL) restores the defaults.  Call it before the
 * launches it should affect; it is not synchronised against concurrent launches from other threads. */
typedef struct {
  int32_t conv_patch;        /* conv_patch_kernel: 0 off, 1 the N = 128 conv3x3 layers only, 2 also the N % 256 == 0 ones; 3: conv_patch8_kernel
                                (64-channel K-tiles on the eight-phase schedule) for the N % 256 == 0 layers, 4: also its 256 x 128 tile for
                                the other N % 64 == 0 layers */
  int32_t big_kernels;       /* 1 (default) gemm_big_kernel where eligible; 0: gemm_kernel tiles only */
  int32_t big_bm, big_bn, big_bk;   /* != 0: force this gemm_big configuration where it fits (sweeps), e.g. 256, 128, 64 */
  int32_t gemm_bm, gemm_bn;  /* != 0: force this gemm_kernel tile (128x128, 128x64, 64x64) instead of the cost model */
  int32_t fsa_key_split;     /* 1 (default) split the bank readers' key range when a workspace is passed; 0 never */
  int32_t fsa_force_splits;  /* != 0: this split count for eligible launches (forward and dQ) */
  int32_t big_min_tiles;     /* gemm_big_kernel only for launches with at least this many tiles (default 192 of the 256 CUs) */
  int32_t k8;                /* gemm8_kernel (64-deep K-tiles, half-tile staging): 0 never; 1 where gemm_big's 256 x 256 tile was planned;
                                2 also its 256 x 128 tiles; 3 also 256 x 128 in place of gemm_big's 512 x 128 tile */
} dfw_config;
int dfw_configure(const dfw_config* cfg);
void dfw_get_config(dfw_config* out);

/* HOST function: number of memset nodes in a captured hipGraph_t (child graphs included), or a negative DFW_E* code;
 * *n_nodes (optional) receives the node count.  The pipeline-owned step graph (pipeline.run_episodes(captured=True))
 * must contain none: on ROCm 7.2 a small memset node replayed next to plain launches on the same stream was observed to
 * receive the next launch's kernel arguments, so the library zeroes scratch with kernels and the host asserts that no
 * torch.zeros / fill_ slipped into the captured step. */
int dfw_graph_memset_nodes(void* hip_graph, int32_t* n_nodes);

/*
 * Implicit-GEMM on MFMA:  C[m][n] = epi( sum_k A(m,k) * W[n][k] ).
 *   taps == 1: A is [M][lda] (a Linear layer / 1x1 conv on NHWC tokens):
 *              to_q/to_k/to_v/to_out (A:237-245, A:276), proj_in/proj_out, GEGLU proj, FF out,
 *              time_embedding / time_emb_proj (U:1015), conv_shortcut, VAE attention projections.
 *   taps == 9: A is the im2col view of an NHWC image [B][Hi][Wi][lda>=Cin], k = (ky*3+kx)*Cin + c:
 *              every conv3x3 of ResnetBlock2D / Downsample2D / Upsample2D / conv_out
 *              (U:1161-1171, U:1191, U:1226-1243, U:1249; VAE encoder/decoder P:852, P:902).
 *              `ups` fuses the nearest-2x upsample of Upsample2D into the load indexing.
 *   W is [N][K] row-major (K = taps*Cin), i.e. nn.Linear.weight, or conv weight permuted to
 *   [Cout][ky][kx][Cin].
 *   epi: + bias[n] + rowbias[m / rows_per_img][n] + residual[m][n], * out_scale, optional SiLU;
 *        geglu != 0: W rows are interleaved in 64-row groups (32 value rows, 32 gate rows) and
 *        C[m][j] = (a + bias_a) * gelu(g + bias_g)  (diffusers GEGLU), C has N/2 columns.
 *   batch > 1: independent GEMMs at strideA/strideW/strideC elements (VAE attention QK^T, PV).
 *   splitk > 1: K is split over `splitk` workgroups per tile; `workspace` must hold
 *               splitk*M*N floats (dfw_gemm_workspace_bytes).
 */
typedef struct {
  const void* A; const void* W; void* C;
  const float* bias; const float* rowbias; const void* residual;
  void* workspace; size_t workspace_bytes;
  int64_t a_elems, w_elems;           /* extent of A / W in elements (for bounds-checked loads) */
  int32_t M, N, K;
  int32_t lda, ldc, ldr;
  int32_t ld_rowbias;                 /* row stride of rowbias in floats (0 => N) */
  int32_t taps, Cin, Hi, Wi, Ho, Wo, stride, pad, ups;
  int32_t rows_per_img;
  float out_scale;
  int32_t act, geglu, out_mode, splitk;
  int32_t batch; int64_t strideA, strideW, strideC;
  int32_t dtype;
  /* Optional fused GroupNorm statistics of the OUTPUT (the next op's GroupNorm input): when
   * gn_partial != NULL the epilogue also writes per-(image, chunk, group) (sum, sum of squares)
   * of the stored values, gn_partial[img][chunk][group][2] floats, chunk < dfw_gemm_gn_chunks().
   * Only kernels/shapes for which dfw_gemm_gn_chunks() returns > 0 support it. */
  float* gn_partial; int32_t gn_groups;
  /* Optional: output columns n < colscale_n (a multiple of 64) are multiplied by `colscale` instead of
   * out_scale, in fp32 before the single rounding to the storage dtype.  The fused [Wq;Wk;Wv] projection
   * (A:237-245) uses it to hand the attention kernel q * (scale * log2 e): the softmax then needs no
   * multiply per score (dfw_fsa_args.q_prescaled). */
  float colscale; int32_t colscale_n;
  /* != 0: `residual` is fp32 [M][ldr floats] instead of the storage dtype.  With out_mode DFW_OUT_F32 this is
   * the fp32 residual stream (residual_dtype=torch.float32 of the Python engines): x + branch(x) of
   * ResnetBlock2D / BasicTransformerBlock / Transformer2DModel is summed and stored in fp32, so the stream
   * is never rounded to 16 bits; only MFMA operands are.  Needs N % 4 == 0 and ldr % 4 == 0. */
  int32_t residual_f32;
  /* Optional second copy of a conv3x3 weight in the BLOCKED layout [tap][Cin / 32][N][32] (element
   * ((tap * Cin/32 + c) * N + n) * 32 + k holds W[n][tap * Cin + 32 c + k]): one (tap, 32-channel chunk) W stage of the
   * patch conv kernel is then ONE contiguous N x 64-byte block, so every LDS-DMA wave-instruction fetches eight whole
   * 128-byte lines instead of sixteen half lines at a row stride of K elements.  Used by the kernels that stage W per
   * (tap, chunk) -- conv_patch_kernel; ignored by the others, which read `W`.  NULL: everything reads `W`. */
  const void* W_blocked;
} dfw_gemm_args;

int dfw_gemm(const dfw_gemm_args* a, dfw_stream_t stream);
size_t dfw_gemm_workspace_bytes(const dfw_gemm_args* a);
/* Which kernel instantiation dfw_gemm would launch for these arguments, e.g.
 * "gemm_kernel<bf16,128,128,conv>" (used by bench.py to attribute time per kernel). */
int dfw_gemm_kernel_name(const dfw_gemm_args* a, char* buf, size_t n);
/* Chunks per image of the fused GroupNorm partial sums this call would emit for gn_groups groups
 * (0: unsupported for this shape / kernel; run dfw_groupnorm's own statistics pass instead). */
int32_t dfw_gemm_gn_chunks(const dfw_gemm_args* a);

/*
 * KV-fusion self-attention (the DiffewS-specific op): out = softmax(q [k_own ; k_bank]^T * scale) [v_own ; v_bank]
 * Replaces xformers.ops.memory_efficient_attention + the bank concat of MyXFormersAttnProcessor
 * (A:247-271).  q/k/v/out are [batch][tokens][heads*64] views with element row strides; the bank
 * holds `nshot` reference images per episode, ref image index = episode*nshot + shot (A:256-257,
 * evaluation_util/main_oss.py:103), each with n_bank tokens; keys are visited in the reference
 * order [own ; shot0 ; shot1 ; ...] without materialising the concat.  nshot == 0: plain
 * self-attention (bank-fill pass, A:251-252).  head_dim must be 64.
 */
typedef struct {
  const void* q; const void* k; const void* v;
  const void* k_bank; const void* v_bank; void* out;
  int32_t batch, heads, n_q, n_kv, n_bank, nshot;
  int32_t ldq, ldk, ldv, ldkb, ldvb, ldo;      /* element strides between tokens */
  int64_t q_bs, k_bs, v_bs, kb_bs, vb_bs, o_bs; /* element strides between batch items */
  float scale;
  int32_t dtype;
  /* Lock-step launch over [support images ; query images] (one trunk pass for both UNet passes): the
   * first n_plain batch entries attend over their own keys only (the bank-fill pass, A:251-252); entry
   * b >= n_plain is episode b - n_plain and reads bank images (b - n_plain)*nshot + shot.  0 = every
   * entry reads the bank (the two-pass form). */
  int32_t n_plain;
  /* != 0: q already carries the factor scale * log2(e) (dfw_gemm_args.colscale on the projection that
   * produced it); `scale` is then ignored and the kernel exponentiates with exp2(q.k - m) directly. */
  int32_t q_prescaled;
  /* Optional (training): lse[batch][heads][n_q] fp32 = log2 of the row's sum of exp2(scaled score), so that
   * exp2(scaled score - lse) is the attention probability -- what dfw_fsa_attention_bwd recomputes P from. */
  float* lse;
  /* Optional scratch of dfw_fsa_workspace_bytes() (16-byte aligned): with it, a lock-step launch whose bank-reading images
   * walk many more keys than the plain ones (nshot >= 2) splits those images' key range over several workgroups and merges
   * the partial softmax results (same mathematics, fp32 partials, fixed order); without it the launch is unsplit. */
  void* workspace; size_t workspace_bytes;
} dfw_fsa_args;

int dfw_fsa_attention(const dfw_fsa_args* a, dfw_stream_t stream);
size_t dfw_fsa_workspace_bytes(const dfw_fsa_args* a);

/*
 * Cross-attention over a short context (attn2 of BasicTransformerBlock; L = 2 prompt tokens at
 * inference P:591-600, 77 in training).  q/out [batch][n_q][heads*64]; k/v [batch][L][heads*64].
 */
typedef struct {
  const void* q; const void* k; const void* v; void* out;
  int32_t batch, heads, n_q, L;
  int32_t ldq, ldk, ldv, ldo;
  int64_t q_bs, k_bs, v_bs, o_bs;
  float scale;
  int32_t dtype;
} dfw_xattn_args;

int dfw_cross_attention(const dfw_xattn_args* a, dfw_stream_t stream);

/*
 * GroupNorm (+ optional SiLU) on NHWC: torch.nn.GroupNorm of ResnetBlock2D.norm1/norm2,
 * Transformer2DModel.norm, conv_norm_out (U:1247-1248), VAE attention group_norm.
 * Two launches: statistics (deterministic two-level reduction, fp32 partials, fp64 combine)
 * then apply.  `stats_ws` must hold dfw_groupnorm_workspace_bytes().
 */
typedef struct {
  const void* x; void* y; const float* gamma; const float* beta;
  void* stats_ws; size_t stats_ws_bytes;
  int32_t B, HW, C, groups, ldx, ldy;
  float eps;
  int32_t silu;
  int32_t dtype;
  /* Optional: partial sums already produced by the conv that wrote x (dfw_gemm_args.gn_partial),
   * [B][pre_chunks][groups][2] floats; the statistics pass over x is then skipped. */
  const float* pre_partial; int32_t pre_chunks;
  /* != 0: x is fp32 [B][HW][ldx floats] (the fp32 residual stream); y stays the storage dtype. */
  int32_t x_f32;
} dfw_groupnorm_args;

int dfw_groupnorm(const dfw_groupnorm_args* a, dfw_stream_t stream);
size_t dfw_groupnorm_workspace_bytes(const dfw_groupnorm_args* a);

/* LayerNorm over the last dim of [rows][C] (BasicTransformerBlock.norm1/2/3). */
typedef struct {
  const void* x; void* y; const float* gamma; const float* beta;
  int32_t rows, C, ldx, ldy;
  float eps;
  int32_t dtype;
  int32_t x_f32;   /* != 0: x is fp32 [rows][ldx floats] (fp32 residual stream); y stays the storage dtype */
} dfw_layernorm_args;

int dfw_layernorm(const dfw_layernorm_args* a, dfw_stream_t stream);

/*
 * Direct convolution for tiny channel counts on the pipeline boundary (NCHW fp32 in):
 * conv_in / conv_in_ref (U:1119-1121), VAE encoder.conv_in, decoder.conv_in, quant_conv,
 * post_quant_conv (P:853, P:901).  Cin <= 8.  W is [Cout][taps][Cin] fp32.
 * out_mode DFW_OUT_T: NHWC storage dtype (Cout % 8 == 0); DFW_OUT_F32: NHWC fp32 (Cout % 8 == 0; the fp32
 * residual stream starts at conv_in); DFW_OUT_NCHW_F32: NCHW fp32.
 * y = (conv(x * in_scale) + bias) * out_scale.
 */
typedef struct {
  const float* x; const float* W; const float* bias; void* y;
  int32_t B, Cin, H, Wd, Cout, taps, ldy;
  float in_scale, out_scale;
  int32_t out_mode;
  int32_t dtype;
  /* Optional fused GroupNorm statistics of the NHWC output (the first resnet's norm1 reads them through
   * dfw_groupnorm_args.pre_partial): gn_partial[B][chunks][gn_groups][2], chunks from
   * dfw_conv_small_gn_chunks() (0: not supported for this shape -- leave gn_partial NULL). */
  float* gn_partial; int32_t gn_groups;
  /* DFW_OUT_NCHW_F32 only: element stride between the images of y (0 => Cout*H*Wd).  Lets the conv write
   * its Cout channels into a channel slice of a wider NCHW tensor, e.g. the latent mean of the support
   * image and of its mask straight into the two halves of `cat([rgb_latent, mask_latent], dim=1)` (P:674). */
  int64_t y_bstride;
  /* Optional: the input batch spread over up to three buffers (each NCHW fp32, contiguous, 16-byte aligned):
   * images [0, b0) from x, [b0, b1) from x1, [b1, B) from x2.  x1 == NULL: all B images from x.  One launch
   * then covers the support images, the support masks and the query images of an episode batch
   * (P:649-651 encodes them in three calls) without first concatenating them. */
  const float* x1; const float* x2; int32_t b0, b1;
} dfw_conv_small_args;

int dfw_conv_small(const dfw_conv_small_args* a, dfw_stream_t stream);
int32_t dfw_conv_small_gn_chunks(const dfw_conv_small_args* a);

/* Row softmax of fp32 scores -> storage dtype probabilities (VAE mid-block attention, 1 head of
 * dim 512: diffusers Attention.get_attention_scores with upcast_softmax).  y = softmax(x*scale). */
int dfw_softmax_rows(const float* x, void* y, int64_t rows, int32_t L, float scale, int32_t dtype,
                     dfw_stream_t stream);

/* Batched 2-D transpose of storage-dtype matrices: y[b][c][r] = x[b][r][c]. */
/* Softmax over L consecutive fp32 scores per (row, group): y[r][g*L + l] = softmax_l(x[r][g*L + l]),
 * g < groups; the remaining columns of the ld-wide row are zeroed.  The folded attn2 (prompt tokens are
 * constants of a checkpoint, P:585-601): scores = LN(x) G, probabilities here, out = P U + residual. */
int dfw_softmax_groups(const float* x, void* y, int64_t rows, int32_t ld, int32_t groups, int32_t L,
                       int32_t dtype, dfw_stream_t stream);
int dfw_transpose(const void* x, void* y, int32_t batch, int32_t R, int32_t C, int32_t dtype,
                  dfw_stream_t stream);

/* Channel concat of two NHWC tensors (torch.cat([h, skip], dim=1) in the up blocks, U:1226). */
int dfw_concat_channels(const void* a, const void* b, void* y, int64_t rows, int32_t Ca, int32_t Cb,
                        int32_t dtype, dfw_stream_t stream);

/* y[i] = (storage dtype) x[i], n % 8 == 0, 16-byte aligned: the 16-bit MFMA-operand copy of an fp32 residual-stream
 * tensor where a conv / Linear consumes the stream itself (conv_shortcut, Downsample2D / Upsample2D convs). */
int dfw_convert_f32(const float* x, void* y, int64_t n, int32_t dtype, dfw_stream_t stream);

/* Zero `bytes` bytes at p (both multiples of 16) with a kernel: buffers zeroed inside a captured step (gradient seeds of
 * the training step, T:1381: the support rows' zero gradient) must not become memset nodes -- see dfw_graph_memset_nodes. */
int dfw_zero(void* p, int64_t bytes, dfw_stream_t stream);

/* hi[i] = (storage dtype) x[i], lo[i] = (storage dtype)(x[i] - hi[i]): the two-operand form of an fp32 tensor (hi + lo = x
 * to ~2^-22 relative in fp16).  Where the fp32 residual stream itself is a conv / Linear operand (ResnetBlock2D.conv_shortcut,
 * Downsample2D / Upsample2D), the GEMM runs on hi, then on lo with the first result as its fp32 residual. */
int dfw_split_f32(const float* x, void* hi, void* lo, int64_t n, int32_t dtype, dfw_stream_t stream);

/* Sinusoidal timestep embedding, flip_sin_to_cos, fp32 math (diffusers Timesteps, U:1008);
 * out [B][dim] in storage dtype. */
int dfw_timestep_embedding(const float* timesteps, void* out, int32_t B, int32_t dim,
                           int32_t flip_sin_to_cos, float freq_shift, int32_t dtype,
                           dfw_stream_t stream);

/*
 * Segmentation post-processing on device (P:790-795, P:534; evaluation_util/main_oss.py:128-137;
 * evaluation_util/common/evaluation.py:24-38): from decoder output x [B][3][H][W] fp32 (already
 * clipped to [-1,1]) produce seg_u8 = uint8(clip((x*0.5+0.5)*255, 0, 255)) [B][3][H][W],
 * and when `gt` is non-null the per-episode 2x2 intersection/union pixel counts of
 * pred = (mean_c(u8/255) > r_threshold * max(u8/255)) against gt [B][H][W] (uint8 0/1, 255 = ignore)
 * into counts [B][4] int64 = {inter0, inter1, union0, union1}.
 * scratch: B uint32 words (per-image max), zeroed by this call.
 */
int dfw_seg_postprocess(const float* x, uint8_t* seg_u8, const uint8_t* gt, int64_t* counts,
                        uint32_t* scratch, int32_t B, int32_t H, int32_t Wd, float r_threshold,
                        dfw_stream_t stream);
/* Same with the launcher's other two choices (evaluation_util/main_oss.py:128-135):
 *   r_threshold > 0: dynamic threshold r_threshold * max; batch_max == 0 takes the max per image (what the
 *     reference computes at --bsz 1, the only batch size its E:128 `to_tensor` accepts; results do not
 *     depend on how episodes are batched), batch_max != 0 takes it over the whole [B,3,H,W] tensor, as
 *     `pred_mask.max()` literally reads for B > 1;
 *   r_threshold <= 0: fixed `--threshold`: pred = mean_c(u8/255) > threshold (> 0 required with gt). */
int dfw_seg_postprocess_ex(const float* x, uint8_t* seg_u8, const uint8_t* gt, int64_t* counts,
                           uint32_t* scratch, int32_t B, int32_t H, int32_t Wd, float r_threshold,
                           float threshold, int32_t batch_max, dfw_stream_t stream);

/* AverageMeter.update on device (evaluation_util/common/logger.py:35-37): for every episode b,
 * inter_buf[k][class_id[b]] += counts[b][k], union_buf[k][class_id[b]] += counts[b][2+k], k = 0, 1.
 * Buffers are int64 [2][nclass] (exact, order-independent sums); class ids outside [0, nclass) are skipped. */
int dfw_meter_update(const int64_t* counts, const int64_t* class_id, int64_t* inter_buf, int64_t* union_buf,
                     int32_t B, int32_t nclass, dfw_stream_t stream);

/*
 * Episode input transform (evaluation_util/data/dataset.py:36-40: Resize((S,S)) on the PIL image,
 * ToTensor, Normalize([0.5],[0.5]); coco.py:36-46: nearest resize of the class mask;
 * main_oss.py:100: mask -> 3 channels, {0,1} -> {-1,+1}).  Bit-exact with Pillow's ImagingResample
 * (BILINEAR, 8 bits per channel: horizontal then vertical pass, uint8 intermediate, 22-bit fixed-point
 * weights) and ATen's `nearest`.  JPEG/PNG decoding stays with the caller.
 *
 * dfw_resample_ksize / dfw_resample_coeffs are HOST functions: Pillow's precompute_coeffs +
 * normalize_coeffs_8bpc for one axis; bounds [out_size][2] = (first input index, tap count),
 * coeffs [out_size][ksize].  The caller copies them to the device next to the image bytes.
 */
int32_t dfw_resample_ksize(int32_t in_size, int32_t out_size);
int dfw_resample_coeffs(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* coeffs);

typedef struct {
  const uint8_t* src;                 /* device, [H][W][3] RGB bytes */
  int32_t H, W, out_h, out_w;
  const int32_t* xbounds; const int32_t* xcoef; int32_t xk;   /* device, from dfw_resample_coeffs(W, out_w) */
  const int32_t* ybounds; const int32_t* ycoef; int32_t yk;   /* device, from dfw_resample_coeffs(H, out_h) */
  uint8_t* tmp;                       /* device scratch, [H][out_w][3] */
  float* dst;                         /* device, [3][out_h][out_w] fp32 */
  const float* lut;                   /* device, 256 floats: byte v after ToTensor + Normalize */
} dfw_image_args;

int dfw_image_to_tensor(const dfw_image_args* a, dfw_stream_t stream);
/* mask: device [H][W] class ids (elem_bytes 1 = uint8 PNG, 4 = int32); on = (id == class_value);
 * dst_pm1 [3][out_h][out_w] fp32 in (-1,+1) and/or dst_bin [out_h][out_w] uint8 in (0,1). */
int dfw_mask_to_tensor(const void* mask, int32_t elem_bytes, int32_t H, int32_t W, int32_t class_value,
                       int32_t out_h, int32_t out_w, float* dst_pm1, uint8_t* dst_bin, dfw_stream_t stream);

/* ======================================================================================================
 * Training step (BASELINE configs[4]; train_tools/train_icl_multitask_nocrop_nearest_nshot_v3.py:1374-1396 =
 * T): backward of the UNet's ops.  Data gradients of Linear / conv3x3 are dfw_gemm calls with transposed /
 * tap-mirrored weights; everything else is below.  All reductions are deterministic (fp32 slabs folded in a
 * fixed order).  `accumulate` != 0 adds into the output gradient (gradient accumulation, T:1323), else
 * overwrites; `scale` / `grad_scale` multiply parameter gradients (1 / loss scale of fp16 training).
 * ====================================================================================================== */

/* Weight gradient on MFMA: out[b][n][tap][k] (+)= scale * sum_m A[m][n] * B[row(m, tap)][k]
 *   taps == 1: dW = dY^T X for nn.Linear (A = dY [M][lda], B = X [M][ldb]; torch autograd of A:237-245, A:276);
 *   taps == 9: dW[Cout][ky][kx][Cin] of a conv3x3 (A = dY NHWC rows, B = the conv's NHWC input gathered with the
 *              forward's stride / pad / fused nearest-2x upsample), the packed layout dfw_gemm consumes.
 * batch / batch2: independent problems at element strides strideA/strideB (and strideA2/strideB2).
 * out strides in floats: ldo_n (0 => taps*Kc), ldo_t (0 => Kc), ldo_b (0 => N*taps*Kc).
 * workspace: dfw_gemm_tn_workspace_bytes() (split-M slabs). N, Kc, lda, ldb multiples of 8. */
typedef struct {
  const void* A; const void* B; float* out;
  void* workspace; size_t workspace_bytes;
  int64_t a_elems, b_elems;
  int32_t M, N, Kc, lda, ldb;
  int32_t taps, Hi, Wi, Ho, Wo, stride, pad, ups;
  int32_t batch, batch2; int64_t strideA, strideB, strideA2, strideB2;
  int64_t ldo_n, ldo_t, ldo_b;
  float scale; int32_t accumulate;
  int32_t dtype;
} dfw_gemm_tn_args;

int dfw_gemm_tn(const dfw_gemm_tn_args* a, dfw_stream_t stream);
size_t dfw_gemm_tn_workspace_bytes(const dfw_gemm_tn_args* a);

/* Column sums: out[seg * ldo + n] (+)= scale * sum_{r < rows_per_seg} x[(seg * rows_per_seg + r) * ldx + n].
 * Bias gradients (segs = 1) and the gradient of the per-image time-embedding projection added in
 * ResnetBlock2D (segment = image).  workspace: dfw_colsum_workspace_bytes(). */
int dfw_colsum(const void* x, float* out, void* workspace, size_t workspace_bytes, int64_t rows_per_seg, int32_t segs,
               int32_t N, int32_t ldx, int64_t ldo, float scale, int32_t accumulate, int32_t dtype, dfw_stream_t stream);
size_t dfw_colsum_workspace_bytes(int64_t rows_per_seg, int32_t segs, int32_t N);

/* Every column sum of a backward walk in two launches.  items: DEVICE array of n_items records of sixteen int64_t
 * { x, out, rows_per_seg, segs, N, ldx, ldo, scale (float bits), accumulate, chunks, rpc, part_off, block_begin1, block_begin2,
 *   0, 0 }: the dfw_colsum arguments of the item (x / out as addresses), its plan from dfw_colsum_plan(), the offset in floats
 * of its segs * chunks * N partial sums in `workspace`, and its first block in the two flattened grids:
 *   block_begin1[i+1] = block_begin1[i] + ceil(N / 256) * chunks * segs,  block_begin2[i+1] = block_begin2[i] + ceil(N / 16) * segs;
 * total_blocks1 / 2 = the sums.  All items share `dtype`; N % 8 == 0, ldx % 8 == 0 (caller-checked).  Deterministic. */
int dfw_colsum_plan(int64_t rows_per_seg, int32_t* chunks, int32_t* rpc);
/* Write n_records (<= 24) records of sixteen int64_t from HOST memory `records` into the DEVICE table at record index
 * first_record.  The records travel as kernel arguments: no pinned memory, no memcpy node -- capture-safe. */
int dfw_table_write(void* table, int64_t first_record, const int64_t* records, int32_t n_records, dfw_stream_t stream);
int dfw_colsum_batch(const void* items, int32_t n_items, int64_t total_blocks1, int64_t total_blocks2, void* workspace,
                     int32_t dtype, dfw_stream_t stream);

/* GroupNorm(+SiLU) backward (torch.nn.GroupNorm + F.silu under autograd; ResnetBlock2D.norm1/norm2,
 * Transformer2DModel.norm, conv_norm_out).  mean_rstd [B][groups][2] are the forward's statistics
 * (the tail of dfw_groupnorm's stats_ws).  dgamma / dbeta may be NULL.
 * dx_add (optional, laid out like dx): a gradient x has already received from another consumer (the residual add that
 * follows the block: pre-norm blocks reach their input's gradient through a norm backward LAST); dx = computed + dx_add,
 * summed in fp32 and rounded once -- the separate add pass over the activation is gone. */
typedef struct {
  const void* x; const void* dy; void* dx; const float* gamma; const float* beta; const float* mean_rstd;
  float* dgamma; float* dbeta;
  void* workspace; size_t workspace_bytes;
  int32_t B, HW, C, groups, ldx, lddy, lddx;
  int32_t silu, accumulate;
  float grad_scale;
  int32_t dtype;
  const void* dx_add;
} dfw_groupnorm_bwd_args;

int dfw_groupnorm_bwd(const dfw_groupnorm_bwd_args* a, dfw_stream_t stream);
size_t dfw_groupnorm_bwd_workspace_bytes(const dfw_groupnorm_bwd_args* a);

/* LayerNorm backward (BasicTransformerBlock.norm1/2/3); statistics are recomputed from x.  dx_add: as above (rows at
 * stride lddx). */
typedef struct {
  const void* x; const void* dy; void* dx; const float* gamma; float* dgamma; float* dbeta;
  void* workspace; size_t workspace_bytes;
  int32_t rows, C, ldx, lddy, lddx;
  float eps;
  int32_t accumulate;
  float grad_scale;
  int32_t dtype;
  const void* dx_add;
} dfw_layernorm_bwd_args;

int dfw_layernorm_bwd(const dfw_layernorm_bwd_args* a, dfw_stream_t stream);
size_t dfw_layernorm_bwd_workspace_bytes(int32_t rows, int32_t C);

/* GEGLU on the packed column order (see dfw_gemm_args.geglu): pre [rows][2H].
 * dout == NULL: forward, out [rows][H] = value * gelu(gate); else backward, out = d(pre) [rows][2H]. */
int dfw_geglu(const void* pre, const void* dout, void* out, int64_t rows, int32_t H, int32_t dtype, dfw_stream_t stream);

/* Data movement of the backward graph, C % 8 == 0:
 *   mode 0  y = a + b                                  [rows][C]    (a tensor consumed twice: residual, skip)
 *   mode 1  y = a[:, c0 : c0 + C], a has lda columns     (backward of torch.cat([h, skip], 1), U:1226)
 *   mode 2  y[B][2H][2W][C]: y[2i][2j] = a[i][j], else 0  (rows = B*2H*2W; stride-2 conv data gradient)
 *   mode 3  y[B][H][W][C] = 2x2 block sums of a[B][2H][2W][C]  (rows = B*H*W; fused nearest-2x upsample) */
int dfw_elementwise(int32_t mode, const void* a, const void* b, void* y, int64_t rows, int32_t C, int32_t lda,
                    int32_t c0, int32_t H, int32_t W, int32_t dtype, dfw_stream_t stream);

/* NCHW fp32 [B][C][HW] -> NHWC storage dtype [B][HW][Cp], channels zero-padded to Cp, times scale
 * (the UNet's latent inputs as the B operand of the conv_in weight gradient). */
int dfw_nchw_to_nhwc(const float* x, void* y, int32_t B, int32_t C, int32_t HW, int32_t Cp, float scale, int32_t dtype,
                     dfw_stream_t stream);

/* loss = mean((pred - target)^2) (F.mse_loss, T:1384; pred/target NCHW fp32 [B][C][HW], C <= 8) and
 * dpred = 2 (pred - target) / numel * loss_scale as NHWC storage dtype [B][HW][8] (zero-initialised by the
 * caller: channels C..7 are not written); dpred_nchw (optional): the same rounded values as NCHW fp32 [B][C][HW], the
 * input of conv_out's data-gradient conv.  workspace: 256 floats. */
int dfw_mse_loss(const float* pred, const float* target, void* dpred, float* dpred_nchw, float* loss, float* workspace,
                 int32_t B, int32_t C, int32_t HW, float loss_scale, int32_t dtype, dfw_stream_t stream);

/* The UNet's backward seeded by an EXTERNAL loss gradient g = d loss / d pred (NCHW fp32 [B][C][HW], C <= 8), e.g. from
 * torch autograd of the launcher's own loss expression (T:1381-1391): dpred / dpred_nchw as dfw_mse_loss writes them,
 * (storage dtype)(g * scale) with scale = the loss scale. */
int dfw_loss_grad(const float* g, void* dpred, float* dpred_nchw, int32_t B, int32_t C, int32_t HW, float scale,
                  int32_t dtype, dfw_stream_t stream);

/* y[i] = (float)x[i] * scale, x in the storage dtype, n % 8 == 0, 16-byte aligned (with dfw_convert_f32: the 16-bit wire
 * format of the gradient all-reduce, T:1226-1228 with grad_comm_dtype=bf16; scale = 1 / world size). */
int dfw_convert_to_f32(const void* x, float* y, int64_t n, float scale, int32_t dtype, dfw_stream_t stream);

/* KV-fusion attention backward for the lock-step batch (dfw_fsa_args.n_plain form; nshot == 0: plain
 * self-attention).  qkv [batch][n][ld >= 3C]: the fused projection output with q PRE-SCALED
 * (dfw_gemm_args.colscale); out / dout [batch][n][ldo]; lse from the forward; delta is scratch of
 * 2 * batch * heads * n floats (the kernels keep -rowsum(dO o O) in the first half and -lse in the second so the dK/dV
 * kernel fetches both row constants of a query tile with one LDS-DMA descriptor).  dqkv [batch][n][ldd >= 3C] receives (dq, dk, dv) with dq taken with respect to the UNSCALED
 * projection output, so it is the dY of the QKV Linear as is.  scale = attn.scale (A:269-271). */
typedef struct {
  const void* qkv; const void* out; const void* dout; const float* lse; float* delta; void* dqkv;
  int32_t batch, heads, n, nshot, n_plain;
  int32_t ld, ldo, ldd;
  float scale;
  int32_t dtype;
  /* Optional scratch of dfw_fsa_attention_bwd_workspace_bytes() (16-byte aligned): the dQ kernel then splits the key range
   * of the bank-reading images over several workgroups (partial dQ in fp32, summed in order), as the forward does. */
  void* workspace; size_t workspace_bytes;
  /* Bytes behind `delta` (version >= 103): must be >= 2 * batch * heads * n * sizeof(float), else DFW_EWORKSPACE -- the
   * scratch doubled in version 102, and a caller still sized for the single array must get an error, not a device write
   * past its buffer. */
  size_t delta_bytes;
} dfw_fsa_bwd_args;

int dfw_fsa_attention_bwd(const dfw_fsa_bwd_args* a, dfw_stream_t stream);
size_t dfw_fsa_attention_bwd_workspace_bytes(const dfw_fsa_bwd_args* a);

/* The same backward with queries and keys / values in their own tensors (attn2 of the training step, T:1368-1375, on
 * the MFMA path: forward = dfw_fsa_attention with n_kv = 77 prompt tokens and lse requested).  q [batch][n_q][heads*64]
 * PRE-SCALED as above; k / v [batch][n_kv][heads*64] column slices of ONE buffer (v >= k, same strides); out / dout share
 * strides; lse fp32 [batch][heads][n_q]; delta: scratch of 2 * batch * heads * n_q floats, as above.  dq with respect to the unscaled projection output;
 * dk / dv share strides.  Strides in elements.  Deterministic: no cross-workgroup sums. */
typedef struct {
  const void* q; const void* k; const void* v; const void* out; const void* dout; const float* lse; float* delta;
  void* dq; void* dk; void* dv;
  int32_t batch, heads, n_q, n_kv;
  int32_t ldq, ldkv, ldo, lddq, lddkv;
  int64_t q_bs, kv_bs, o_bs, dq_bs, dkv_bs;
  float scale;
  int32_t dtype;
  /* Optional scratch of dfw_attention_bwd_workspace_bytes() (16-byte aligned; 0 bytes when the launch fills the chip by
   * itself): with a short key axis (77 prompt tokens = one key block per image and head) the dK/dV kernel then splits the
   * query rows over several workgroups per key block (fp32 partials, summed in order: still deterministic). */
  void* workspace; size_t workspace_bytes;
  size_t delta_bytes;   /* bytes behind `delta`: >= 2 * batch * heads * n_q * sizeof(float), else DFW_EWORKSPACE (version >= 103) */
} dfw_attn_bwd_args;

int dfw_attention_bwd(const dfw_attn_bwd_args* a, dfw_stream_t stream);
size_t dfw_attention_bwd_workspace_bytes(const dfw_attn_bwd_args* a);

/* Cross-attention (attn2) backward over a short context; same tensor conventions as dfw_cross_attention.
 * dq [batch][n_q][heads*64] (strides lddq / dq_bs); dk / dv [batch][L][heads*64] at row stride lddkv, image stride
 * dkv_bs (column slices of the fused prompt K/V gradient buffer).  workspace: dfw_cross_attention_bwd_workspace_bytes. */
typedef struct {
  const void* q; const void* k; const void* v; const void* dout; void* dq; void* dk; void* dv;
  void* workspace; size_t workspace_bytes;
  int32_t batch, heads, n_q, L;
  int32_t ldq, ldk, ldv, ldo, lddq, lddkv;
  int64_t q_bs, k_bs, v_bs, o_bs, dq_bs, dkv_bs;
  float scale;
  int32_t dtype;
} dfw_xattn_bwd_args;

int dfw_cross_attention_bwd(const dfw_xattn_bwd_args* a, dfw_stream_t stream);
size_t dfw_cross_attention_bwd_workspace_bytes(int32_t batch, int32_t heads, int32_t n_q, int32_t L);

/* y = silu(a) (dy == NULL) or y = dy * silu'(a), n storage-dtype elements (the timestep-embedding MLP). */
int dfw_silu(const void* a, const void* dy, void* y, int64_t n, int32_t dtype, dfw_stream_t stream);

/* Memory-efficient attention of the VAE mid-block (diffusers AutoencoderKL `Attention`: ONE head of dim 512 over N = h * w
 * tokens; the reference routes it through xformers memory_efficient_attention, evaluation_util/main_oss.py:374-376, so the
 * N x N scores are never materialised): out[b][i] = softmax_j(q[b][i] . k[b][j]) v[b][j], flash-style, fp32 statistics.
 * q / k / v / out: [batch][n][ld* >= 512] views (16-byte aligned, ld % 8 == 0; column slices of one fused QKV buffer are
 * fine), batch strides in elements.  q must arrive multiplied by head_dim^-0.5 * log2(e) (dfw_gemm_args.colscale of the
 * projection that produced it): q_prescaled != 0 is required, head_dim must be 512.  Any n (keys beyond n are masked). */
typedef struct {
  const void* q; const void* k; const void* v; void* out;
  int32_t batch, n, head_dim, q_prescaled;
  int32_t ldq, ldk, ldv, ldo;
  int64_t q_bs, k_bs, v_bs, o_bs;
  int32_t dtype;
} dfw_vattn_args;
int dfw_vae_attention(const dfw_vattn_args* a, dfw_stream_t stream);

/* Sum of squares of an fp32 vector (the global gradient norm of clip_grad_norm_, T:1393); workspace: 1024 floats. */
int dfw_sumsq(const float* x, float* out, float* workspace, int64_t n, dfw_stream_t stream);

/* AdamW step on flat fp32 tensors (torch.optim.AdamW, T:1186-1194), with the clip factor
 * min(1, max_grad_norm / (sqrt(*grad_sumsq) + 1e-6)) applied to the gradient when grad_sumsq != NULL. */
typedef struct {
  float* param; const float* grad; float* exp_avg; float* exp_avg_sq; const float* grad_sumsq;
  int64_t n;
  float lr, beta1, beta2, eps, weight_decay, max_grad_norm;
  int32_t step;
  /* Optional: storage-dtype copy of the updated parameters (what the MFMA kernels read), written in the same pass. */
  void* shadow; int32_t shadow_dtype;
  /* Overflow guard: when grad_sumsq != NULL and *grad_sumsq is inf / NaN (one non-finite gradient), the step is SKIPPED --
   * param, both moments and the shadow are left untouched.  *found_inf (optional, device int32) is written by every call
   * that has grad_sumsq: 1 when the step was skipped, else 0.  The
   * behaviour of torch.cuda.amp.GradScaler.step under accelerate mixed_precision='fp16' (T:1017, T:1239, T:1394); the
   * host halves its loss scale on the flag (diffews_amd.train.UNetTrainer.optimizer_step). */
  int32_t* found_inf;
} dfw_adamw_args;

int dfw_adamw(const dfw_adamw_args* a, dfw_stream_t stream);

/* 16-bit weight re-layout for the data-gradient GEMMs: y[(flip ? nb-1-b : b)][c][r] = x[b][r][c], matrices of R x C
 * elements with row strides ldx / ldy and matrix strides x_bs / y_bs (all multiples of 8 elements).
 *   Linear:  nb = 1            -> W^T ([K][N] read by dfw_gemm as the weight of dX = dY W)
 *   conv3x3: nb = 9, flip != 0 -> W'[Cin][8 - tap][Cout] = W[Cout][tap][Cin] (x_bs = Cin, ldx = 9*Cin, y_bs = Cout, ldy = 9*Cout) */
int dfw_weight_relayout(const void* x, void* y, int32_t R, int32_t C, int64_t ldx, int64_t ldy, int32_t nb, int64_t x_bs,
                        int64_t y_bs, int32_t flip, dfw_stream_t stream);

/* Every re-layout of a training step in one launch.  items: DEVICE array of n_items records of twelve int64_t
 * { x, y, R, C, ldx, ldy, x_bs, y_bs, nb, flip, block_begin, 0 } -- the dfw_weight_relayout arguments of the item (x / y
 * as addresses) and the index of its first block in the flattened grid: block_begin[0] = 0, block_begin[i+1] =
 * block_begin[i] + ceil(C/64) * ceil(R/64) * nb; total_blocks = the sum.  Same alignment rules as above (caller-checked). */
int dfw_weight_relayout_batch(const void* items, int32_t n_items, int64_t total_blocks, dfw_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFEWS_HIP_H */
