/*
 * dinounet_hip.h -- C ABI of libdinounet_hip.so: the MI355X (gfx950) kernels behind the Dino U-Net
 * forward/backward hot path.  Plain pointers and sizes only (no torch types); every entry point
 * enqueues on the caller's HIP stream, does not synchronise, keeps no global state and returns 0 on
 * success or a negative DU_ERR_* code (the Python host turns that into RuntimeError, mirroring the
 * AT_ASSERTM behaviour of the reference extension, ops/src/cuda/ms_deform_attn_cuda.cu:33-57).
 *
 * All pointers are DEVICE pointers unless stated otherwise.  Conv-side tensors are NHWC ("pixel-major",
 * channel-contiguous); token tensors are (B, N, D) row-major -- the same memory as NHWC.  `ld*` are row
 * (pixel) strides in ELEMENTS, so a channel slice of a wider NHWC tensor can be read/written in place.
 *
 * Reference interfaces replaced (paths relative to the reference root):
 *   du_msda_forward / du_msda_backward
 *       <- ms_deform_attn_forward / ms_deform_attn_backward, the two pybind names of the extension
 *          `MultiScaleDeformableAttention` (ops/src/vision.cpp:18-21, ops/src/ms_deform_attn.h:25-66,
 *          kernels ops/src/cuda/ms_deform_im2col_cuda.cuh:242-304 and :961-1332)
 *   du_gemm (linear / 1x1 conv / implicit 3x3 conv / transposed conv, fwd + dgrad + wgrad)
 *       <- torch F.linear / F.conv2d / F.conv_transpose2d call sites: layers/attention.py:88-90,
 *          layers/ffn_layers.py:43-49, layers/patch_embed.py:70, ms_deform_attn.py:183-215,
 *          dinov3_adapter.py:85-89,239-277,360,467, dinounet_training.py:423-438,253,613-619
 *   du_attention_fwd, du_qkv_rope_split
 *       <- SelfAttention.compute_attention / apply_rope (layers/attention.py:66-85,106-118)
 *   du_layernorm_fwd / _bwd      <- nn.LayerNorm (layers/block.py:43,56; dinov3_adapter.py:128-137)
 *   du_chan_stats / du_norm_act_* <- InstanceNorm2d+LeakyReLU (dinounet_training.py:401,238; decoder
 *          StackedConvBlocks :581-592) and SyncBatchNorm(+ReLU) (dinov3_adapter.py:242-270,361-364)
 *   du_dwconv3x3_*               <- DWConv (dinov3_adapter.py:94-109), DepthwiseSeparableConv.depthwise
 *          (dinounet_training.py:235)
 *   du_maxpool3x3s2_*            <- nn.MaxPool2d(3,2,1) (dinov3_adapter.py:250)
 *   du_bilinear_add_*            <- F.interpolate(bilinear, align_corners=False)+add (dinov3_adapter.py:472-476)
 *   du_se_gate_* / du_se_scale_*  <- SqueezeExcitation + residual (dinounet_training.py:210-225,438)
 *   du_msda_prep / du_msda_prep_bwd <- ms_deform_attn.py:188-197
 *   du_dice_ce_*                  <- DC_and_CE_loss (training/loss/compound_losses.py:8-56, dice.py:58-119), the loss nnUNetTrainer.train_step
 *          applies to the logits (nnUNetTrainer.py:917); "next" row of the scope table
 */
#ifndef DINOUNET_HIP_H
#define DINOUNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DU_OK 0
#define DU_ERR_BAD_ARG (-1)
#define DU_ERR_UNSUPPORTED (-2)
#define DU_ERR_LAUNCH (-3)

/* element types */
#define DU_F32 0
#define DU_BF16 1

/* du_gemm operand addressing modes.  The logical product is C[m][n] = sum_k A(m,k) * B(n,k). */
#define DU_PLAIN_ROW 0   /* element (outer o, contraction k) at p[o*ld + k]          (k contiguous) */
#define DU_PLAIN_COL 1   /* element (outer o, contraction k) at p[k*ld + o]          (o contiguous) */
#define DU_IM2COL_ROW 2  /* outer = output pixel, contraction = (tap, channel) gathered from an NHWC tensor */
#define DU_IM2COL_COL 3  /* outer = (tap, channel), contraction = pixel (used by conv weight gradients) */

/* du_gemm epilogue activations */
#define DU_ACT_NONE 0
#define DU_ACT_GELU 1    /* exact erf GELU (nn.GELU default) */
#define DU_ACT_RELU 2
#define DU_ACT_LEAKY 3   /* LeakyReLU(0.01) */
#define DU_ACT_SWIGLU 4  /* gate of SwiGLUFFN (layers/ffn_layers.py:73-77) on an INTERLEAVED projection: B rows (2j, 2j+1) = (w1[j], w2[j]),
                            bias likewise; the result has N/2 columns, C[m][j] = silu(v[2j]) * v[2j+1], v = alpha*acc + bias; ldc / the
                            stored row length refer to those N/2 columns.  Served by the multi-phase bf16 NT kernels only (du_gemm
                            returns DU_ERR_UNSUPPORTED otherwise: run the product without activation + du_swiglu_pairs) */

/* du_gemm store modes */
#define DU_STORE_PLAIN 0
#define DU_STORE_SLABS 4 /* split_k > 1, fp32 result, generic bf16 engine only: split s writes its partial product with plain stores to
                            C + s * M * ldc (split_k slabs of M x ldc floats, no zero fill needed) instead of adding into C atomically --
                            the caller reduces the slabs in a fixed order (du_splitk_reduce_bf16): bit-reproducible results */
#define DU_STORE_QKV_HEADS 5 /* as DU_STORE_QKV_ROPE without the rotation and the q scale: the projection (+ bias) stored head-major in the
                               three planes (ps_H / ps_W / ps_C as there), rotated afterwards where it lies by du_qkv_rope_inplace.  Served by the
                               persistent multi-phase kernel only (M % 256 == 0, K >= 384; DU_ERR_UNSUPPORTED otherwise) */
#define DU_STORE_MSDA_PREP 6 /* MSDeformAttn's sampling_offsets | attention_weights product (ms_deform_attn.py:188-197) with the reference-point /
                               softmax step in the epilogue: N = heads * 12 columns = [heads x 4 points x (x, y) offsets | heads x 4 logits] (+ bias),
                               fp32; instead of the matrix, C receives the sampling locations (M, heads, 4, 2) = ref[m % ps_C] + offset / (ps_W,
                               ps_H) and C2 the softmax over the 4 points (M, heads, 4); rope_sin = ref (ps_C, 2) fp32.  Served by the 256 x 256
                               multi-phase kernel only (N <= 256, K % 128 == 0; DU_ERR_UNSUPPORTED otherwise: du_msda_prep on the plain result) */
#define DU_STORE_TAPS 3 /* grouped convolution weight gradients only (du_gemm_tn_group: du_tn_job.taps): column n = (tap, c) of the product is
                           element (m, c, tap) of a torch-layout weight gradient */
#define DU_STORE_QKV_ROPE 2 /* the ViT's qkv projection stored head-major with RoPE: row m = (b, token t) of M = B * ps_H tokens, column
                              n = (which, h, d) of N = 3 * ps_C * 64 -> out[which][b][h][t][d] in three (B, ps_C, ps_W = Npad, 64) planes ldc
                              elements apart; q and k of tokens >= rope_prefix are rotated (rotate-half, fp32) with rope_sin / rope_cos
                              ((ps_H - prefix, 64) fp32), q is multiplied by rope_qscale.  bf16, d_head 64, bias-only epilogue; served
                              by the 256 x 128 multi-phase kernel only (DU_ERR_UNSUPPORTED otherwise: du_qkv_rope_split then) */
#define DU_STORE_PIXEL_SHUFFLE2 1 /* ConvTranspose2d k2 s2: column n=(dy*2+dx)*Cout+co of input pixel (b,y,x)
                                     goes to output pixel (b,2y+dy,2x+dx), channel co */

typedef struct {
  /* source NHWC tensor(s): channels [0,C1) come from p (pixel stride ld), channels [C1,C) from p2 (ld2).
     This is how the decoder's torch.cat (dinounet_training.py:614) is consumed without materialising it. */
  const void* p2;
  int64_t ld2;
  int32_t C1;
  int32_t Hi, Wi;      /* source spatial size */
  int32_t C;           /* total gathered channels (C1 + C2) */
  int32_t KH, KW, stride, pad;
  int32_t Ho, Wo;      /* grid the "pixel" index runs over */
  int32_t transposed;  /* 0: yi = yo*stride - pad + dy.  1: yi = (yo + pad - dy)/stride when divisible (dgrad) */
} du_conv_geom;

typedef struct {
  int32_t dtype;      /* DU_F32 / DU_BF16: element type of A and B */
  int32_t out_dtype;  /* element type of C and residual */
  int32_t a_mode, b_mode;
  int32_t M, N, K;
  const void* A; int64_t lda; int64_t a_batch_stride;
  const void* B; int64_t ldb; int64_t b_batch_stride;
  void* C; int64_t ldc; int64_t c_batch_stride;
  int32_t batch;      /* >= 1 */
  int32_t split_k;    /* >= 1; > 1 requires out_dtype F32, a zero-initialised C and no epilogue ops */
  float alpha;
  const float* bias;  /* [N] fp32 or NULL */
  int32_t act;
  const float* gamma; /* [N] fp32 or NULL: per-column scale applied after act (LayerScale) */
  const float* row_scale; int32_t rs_rows; /* optional per-row-block scale v *= row_scale[m / rs_rows] (DropPath) */
  const void* residual; int64_t ldr; /* added last, NULL for none; may alias C.  Indexed like C: with DU_STORE_PIXEL_SHUFFLE2 it is a
                                        tensor of the OUTPUT shape (pixel stride ldr), e.g. the skip added to a ConvTranspose result */
  int32_t store_mode;
  int32_t ps_H, ps_W, ps_C; /* pixel-shuffle geometry: input grid H x W, Cout */
  du_conv_geom geom;  /* used by the IM2COL operand (at most one operand is IM2COL) */
  float* ws; int64_t ws_elems; /* optional scratch, du_gemm_ws_elems(args) floats (0 = none wanted for this product); without it the
                                  product still runs, on the plain tile kernels */
  const float* rope_sin; const float* rope_cos; int32_t rope_prefix; float rope_qscale;   /* DU_STORE_QKV_ROPE only */
  float* a_colsum;    /* optional, weight-gradient products only (a_mode PLAIN_COL, bf16): a_colsum[m] += sum_k A(m, k), fp32 atomics
                         into a buffer the caller zero-initialised -- the bias gradient sum_rows dY for free while dY^T X streams dY
                         anyway (nn.Linear backward).  du_gemm returns DU_ERR_UNSUPPORTED if the kernel family serving the product
                         cannot do it (du_gemm_route != 1 and != 5): call du_colsum then.  The side sum carries the product's `alpha`
                         (a_colsum[m] += alpha * sum_k A(m, k): the per-sample DropPath scale of the grouped jobs needs it); pass alpha = 1
                         for a plain bias gradient */
  float* b_colsum;    /* optional, ConvTranspose2d k2 s2 weight-gradient products only (a_mode PLAIN_COL, b_mode IM2COL_COL, bf16):
                         b_colsum[n % geom.C] += sum_k B(n, k) -- every dY pixel appears exactly once among the (input pixel, tap) pairs, so
                         this is the bias gradient sum_pixels dY[.., co]; same contract as a_colsum (zero-initialised, atomics,
                         DU_ERR_UNSUPPORTED unless du_gemm_route is 1 or 5) */
  float* C2;          /* DU_STORE_MSDA_PREP only: the second result (attention weights) */
  void* ks_ws;        /* optional scratch for workgroups that meet INSIDE a launch: the K-sliced ragged-row units of a tall product with a long
                         contraction (the ViT's fc2: M = 8 x 1029, K = 4096) and the K-split pair kernel (option key 16):
                         du_gemm_ks_ws_bytes(args) bytes that PERSIST between calls on one stream; the first 131072 bytes zeroed ONCE by
                         the caller (every launch leaves them zero again), the rest uninitialised.  One buffer per stream that may run
                         such products concurrently.  NULL / too small: one unit per 32 columns, one workgroup per tile */
  int64_t ks_ws_bytes;
} du_gemm_args;

int du_gemm(const du_gemm_args* args, void* stream);
/* Scratch du_gemm can use for `args` (ws / ws_elems are ignored here).  Non-zero for tall bf16 "NT" products whose row count
   leaves a short ragged last tile row on an otherwise exactly filled GPU (the ViT's M = 8 * 1029): those rows then go through a
   K-parallel skinny kernel pair instead of opening a nearly empty extra round of 128 x 128 tiles. */
int64_t du_gemm_ws_elems(const du_gemm_args* args);
/* Bytes of ks_ws du_gemm would use for `args` (0: nothing in this product meets inside the launch: shape, epilogue, du_set_option keys
   16 / 17). */
int64_t du_gemm_ks_ws_bytes(const du_gemm_args* args);
/* Kernel family du_gemm runs for the bulk of `args` (for profilers: names the kernel without mirroring the dispatch): 0 generic,
   1 bf16 tile engine (gemm_bf16.hip), 2 128 x 128 direct-to-LDS NT kernel (gemm_glds.hip), 3 / 4 the 256 x 256 / 256 x 128
   multi-phase NT kernels (gemm_p8.hip). */
int du_gemm_route(const du_gemm_args* args);

/* ---- grouped weight gradients: several dW = dY^T X products in ONE launch ----------------------------------------------------------
   The reference computes every weight gradient inside its layer's backward (torch autograd: F.linear / conv backward at
   dinov3_adapter.py:85-89,140-156, ms_deform_attn.py:158-216); nothing reads one before clip_grad_norm_ / optimizer.step()
   (nnUNetTrainer.py:919-928), so the host may queue them and launch them together.  Each job: C[m][n] (+)= sum_k A[k][m] * B[k][n] with
   A = dY (K rows, M columns, row stride lda), B = X (K rows, N columns, row stride ldb), bf16; C fp32 (M, N), row stride ldc, ZEROED by
   the caller (jobs that get more than one K split add into it atomically); a_colsum (nullable, zeroed, M floats): += sum_k A[k][m], the
   bias gradient.  The 256 workgroups of a launch are dealt out over the jobs in proportion to their contraction length.
   du_gemm_tn_group_legal: 1 if a job can be queued (K % 128 == 0, K >= 512, lda / ldb % 8 == 0, 16-byte aligned operands, < 2 GB
   operands); du_gemm_tn_group returns DU_ERR_UNSUPPORTED if any job is not.
   Convolution weight gradients (gather != 0) read B in place from an NHWC tensor, K = B * Hs * Ws contraction pixels (Ws -- and for
   gather 3 also Hs -- a power of two; gather 3: Ws % 64 == 0, a 64-pixel K-tile lies inside one image row), Cb channels per tap
   (Cb % 8 == 0), N = taps * Cb:
     gather 2: nn.ConvTranspose2d(k 2, s 2) (dinounet_training.py:255-264,558; dinov3_adapter.py:360): A = x (K, M = Cin), B(k, (tap, co)) =
               dy[pixel (2y + tap / 2, 2x + tap % 2)][co] with dy (B, 2 Hs, 2 Ws, Cb), pixel stride ldb; b_colsum (nullable, Cb floats,
               zeroed) += the bias gradient sum_pixels dy;
     gather 3: 3 x 3 / stride 1 / pad 1 convolution (the U-Net decoder blocks, dinounet_training.py:581-592): A = dy (K, M = Cout),
               B(k, (tap, ci)) = x[pixel + (tap / 3 - 1, tap % 3 - 1)][ci], zero outside the image; x (B, Hs, Ws, Cb), pixel stride ldb.
               A convolution over a fused channel concat (dinounet_training.py:614) is queued as one job per source.
   taps > 1: C is written in torch's weight layout, element (m, c_off + c, tap) at (m * inner_total + c_off + c) * taps + tap (inner = Cb
   channels in this job, inner_total in the parameter): the gradient of a (Cout, Cin, 3, 3) or (Cin, Cout, 2, 2) parameter without a
   permute afterwards; ldc is ignored. */
typedef struct du_tn_job {
  const void* A; int64_t lda;
  const void* B; int64_t ldb;
  float* C; int64_t ldc;
  float* a_colsum;
  const float* alpha;        /* nullable: scale of the whole product, read from device memory at run time (DropPath's per-sample scale in
                                backward, dinov3_adapter.py:18-37,148: the caller queues one job per sample); *alpha == 0 costs nothing */
  int32_t M, N, K;
  int32_t accumulate;        /* != 0: other jobs add into the same C / a_colsum (per-sample jobs, a parameter shared by several layers) */
  float* b_colsum;
  int32_t gather, Hs, Ws, Cb, taps, inner, inner_total, c_off;
} du_tn_job;
int du_gemm_tn_group_legal(const du_tn_job* job);
int du_gemm_tn_group(const du_tn_job* jobs, int njobs, void* stream);

/* ---- LDS-tiled direct 3x3 convolution (stride 1, pad 1), bf16 NHWC: decoder / FAPM / SPM-stem convs (dinounet_training.py:581-592,
        dinov3_adapter.py:243-249) and, with flipped + transposed weights, their data gradients --------------------------------------- */
/* x (B,H,W,C1) [+ x2 (B,H,W,Cin-C1): fused concat, nullable]; w bf16 [Cout][9*Cin] in (tap, ci) column order; y (B,H,W,Cout).
   stats_part (nullable): (du_conv3x3_halo_parts(...), Cout, 2) fp32 partial (sum, sum of squares) of the outputs, the partials of
   one image contiguous -- feed to du_strip_finalize(G = B).  Returns DU_ERR_UNSUPPORTED for shapes it does not serve (H%8, W%16, Cout not in
   {32,64,128}, channel counts not multiples of 32): the caller then uses du_gemm's implicit-GEMM path. */
int du_conv3x3_halo(const void* x, int64_t ldx, const void* x2, int64_t ldx2, int C1, int Cin, int Cout, int B, int H, int W,
                    const void* w, const float* bias, void* y, int64_t ldy, float* stats_part, void* stream);
/* rows of the partial-statistics array du_conv3x3_halo writes for this shape (depends on the kernel that serves it; 0 = not served) */
int du_conv3x3_halo_parts(int C1, int Cin, int Cout, int B, int H, int W);
/* The streaming kernel behind du_conv3x3_halo for Cin in {32, 64 (one tensor, or a 32 + 32 concat)}, Cout in {32, 64}, W % 128 == 0,
   H % 8 == 0 (one wave per 32-column strip, weights in registers / an LDS image, rows by LDS-DMA; csrc/conv_strip.hip).  Arguments as
   du_conv3x3_halo.  DU_ERR_UNSUPPORTED for any other shape. */
int du_conv3x3_strip(const void* x, int64_t ldx, const void* x2, int64_t ldx2, int C1, int Cin, int Cout, int B, int H, int W,
                     const void* w, const float* bias, void* y, int64_t ldy, float* stats_part, void* stream);
/* Weight gradient of the same convolution, dw (Cout, 9*Cin) fp32 in (tap, ci) column order (OVERWRITTEN).  part: scratch of
   du_conv3x3_wgrad_halo_blocks(...) * Cout * 9*Cin floats (0 blocks = shape not served -> use du_gemm's IM2COL_COL path).
   with_db != 0: the bias gradient db[co] = sum_pixels dy rides along -- dw then has Cout * 9*Cin + Cout elements (db behind the weight
   gradient) and part blocks * (Cout * 9*Cin + Cout). */
int du_conv3x3_wgrad_halo_blocks(int C1, int Cin, int Cout, int B, int H, int W);
int du_conv3x3_wgrad_halo(const void* x, int64_t ldx, const void* x2, int64_t ldx2, int C1, int Cin, int Cout, int B, int H, int W,
                          const void* dy, int64_t lddy, float* part, float* dw, int with_db, void* stream);
/* out[g][c][j] = sum_s part[g*strips + s][c][j]: second stage of the column reductions, exposed for producers that emit partials */
int du_strip_finalize(const float* part, float* out, int G, int strips, int C, void* stream);

/* ---- ViT attention -------------------------------------------------------------------------- */
/* qkv: (B, N, 3, H, Dh) as produced by the fused QKV GEMM.  Writes q (scaled by `qscale`), k, v as
   (B, H, Npad, Dh) (rows >= N untouched: the caller zero-fills once); RoPE (rotate-half form, fp32 math) is
   applied to q and k for tokens >= prefix using sin/cos tables (N - prefix, Dh) fp32. */
int du_qkv_rope_split(int dtype, const void* qkv, void* q, void* k, void* v, const float* sin_t, const float* cos_t,
                      int B, int N, int Npad, int H, int Dh, int prefix, float qscale, void* stream);
/* The same for the token rows [m_begin, m_begin + m_count) of the (B * N, 3 * H * Dh) matrix only; qkv_rows points at row m_begin (the
   rows a tile kernel left to the skinny tail when it wrote the rest through DU_STORE_QKV_ROPE). */
int du_qkv_rope_split_rows(int dtype, const void* qkv_rows, void* q, void* k, void* v, const float* sin_t, const float* cos_t, int B, int N,
                           int Npad, int H, int Dh, int prefix, float qscale, int64_t m_begin, int m_count, void* stream);
/* RoPE + q scale IN PLACE on head-major q / k planes (B, H, Npad, Dh) that du_gemm stored unrotated (DU_STORE_QKV_HEADS): the arithmetic of
   du_qkv_rope_split on rows b * N + n < m_limit (rows at and behind m_limit were written, rotated, by du_qkv_rope_split_rows); bf16.
   Replaces layers/attention.py:66-85 like du_qkv_rope_split. */
int du_qkv_rope_inplace(int dtype, void* q, void* k, const float* sin_t, const float* cos_t, int B, int N, int Npad, int H, int Dh,
                        int prefix, float qscale, int64_t m_limit, void* stream);
/* Non-causal softmax2(q k^T) v per (b, head) with base-2 exponentials: q must be pre-scaled by
   Dh^-0.5 * log2(e).  q,k,v: (B, H, Npad, Dh) bf16; out: (B, N, H*Dh) bf16.  Dh in {64, 128}. */
int du_attention_fwd(const void* q, const void* k, const void* v, void* out, int B, int H, int N, int Npad, int Dh,
                     void* stream);
/* row softmax over the first `cols` entries of each row of a (rows, ld) fp32 matrix, in place; entries
   [cols, ld) are zeroed (parity-mode attention with materialised scores). */
int du_softmax_rows_f32(float* x, int64_t rows, int cols, int64_t ld, void* stream);

/* ---- LayerNorm -------------------------------------------------------------------------------- */
int du_layernorm_fwd(int in_dtype, int out_dtype, const void* x, int64_t ldx, const float* w, const float* b, void* y,
                     int64_t ldy, float* mean_out, float* rstd_out, int64_t rows, int D, float eps, void* stream);
/* Column reductions (statistics, bias / norm-parameter gradients) run in two stages when the caller lends a scratch buffer
   `ws` of at least du_reduce_ws_elems(...) floats: every workgroup writes its strip's partial sums, a second kernel adds them
   up and OVERWRITES the result buffer.  With ws == NULL the partials are accumulated with fp32 atomics into a result buffer the
   caller must have zero-filled (same-cache-line atomics serialise: ~0.1 us each, measured, so this is the slow fallback). */
int64_t du_reduce_ws_elems(int dtype, int G, int64_t pix_per_group, int C);
/* dx (same dtype as x); dwdb fp32: with scratch (du_reduce_ws_elems(dtype, 1, rows, D) floats) PLANAR dw[D] then db[D]; with ws == NULL the
   atomics fallback accumulates (dw[c], db[c]) interleaved into a zero-filled (D, 2) buffer.
   dres (nullable, same dtype/shape as x): gradient of a residual branch that by-passes the norm, added into dx. */
int du_layernorm_bwd(int dtype, const void* x, const void* dy, const float* w, const float* mean, const float* rstd,
                     void* dx, float* dwdb, int64_t rows, int D, float* ws, int64_t ws_elems, const void* dres, void* stream);

/* ---- channel-statistics norms (InstanceNorm2d / BatchNorm2d over NHWC) ---------------------------- */
/* sums[g][c][0..1] += (sum x, sum x^2) over the pixels of group g (G groups of `pix_per_group` pixels). */
int du_chan_stats(int dtype, const void* x, int64_t ldx, float* sums, int G, int64_t pix_per_group, int C, float* ws,
                  int64_t ws_elems, void* stream);
/* out[c] = sum over rows of x[row][c] (bias gradients); scratch of du_reduce_ws_elems(dtype, 1, rows, C) floats is required */
int du_colsum(int dtype, const void* x, int64_t ldx, float* out, int64_t rows, int C, float* ws, int64_t ws_elems, void* stream);
/* sums[g][c][0..1] += (sum a*b, sum a) over the pixels of group g (squeeze-excitation gate gradient, a = dy, b = x). */
int du_chan_dot(int dtype, const void* a, int64_t lda, const void* b, int64_t ldb, float* sums, int G, int64_t pix_per_group, int C,
                float* ws, int64_t ws_elems, void* stream);
/* mean = sum/count, rstd = rsqrt(max(sumsq/count - mean^2, 0) + eps) for sums (G,C,2) -> (G,C); with run_mean/run_var (nullable, G = 1):
   BatchNorm running-statistics update, momentum m, unbiased variance (torch semantics). */
int du_norm_stats_finalize(const float* sums, float count, float eps, float* mean, float* rstd, int G, int C, float* run_mean,
                           float* run_var, float momentum, void* stream);
/* affine-parameter gradients from the backward sums: dw[c] = sum_g bsums[g][c][1], db[c] = sum_g bsums[g][c][0] */
int du_norm_param_grads(const float* bsums, float* dw, float* db, int G, int C, void* stream);
/* Fused forms (one launch less each: a launch of this size costs ~5 us inside the replayed graph).  All three need the scratch of
   du_reduce_ws_elems (no atomics fallback):
   du_chan_stats_norm       = du_chan_stats + du_norm_stats_finalize (sums nullable);
   du_strip_finalize_norm   = du_strip_finalize + du_norm_stats_finalize, for statistics partials from a convolution epilogue;
   du_norm_act_bwd_stats_grads = du_norm_act_bwd_stats + du_norm_param_grads (bsums, dw, db all written). */
int du_chan_stats_norm(int dtype, const void* x, int64_t ldx, float* sums, int G, int64_t pix_per_group, int C, float* ws, int64_t ws_elems,
                       float count, float eps, float* mean, float* rstd, float* run_mean, float* run_var, float momentum, void* stream);
int du_strip_finalize_norm(const float* part, float* sums, int G, int strips, int C, float count, float eps, float* mean, float* rstd,
                           float* run_mean, float* run_var, float momentum, void* stream);
int du_norm_act_bwd_stats_grads(int dtype, const void* x, int64_t ldx, const void* dy, int64_t lddy, const float* mean, const float* rstd,
                                const float* w, const float* b, float* bsums, float* dw, float* db, int G, int64_t pix_per_group, int C,
                                int act, float* ws, int64_t ws_elems, void* stream);
/* y = act((x - mean[g,c]) * rstd[g,c] * w[c] + b[c]); mean/rstd: (G, C) fp32. */
int du_norm_act_fwd(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, const float* mean, const float* rstd,
                    const float* w, const float* b, int G, int64_t pix_per_group, int C, int act, void* stream);
/* backward pass 1: through the activation, accumulate per-(g,c) sum(dz) and sum(dz*xhat) into bsums (G,C,2). */
int du_norm_act_bwd_stats(int dtype, const void* x, int64_t ldx, const void* dy, int64_t lddy, const float* mean,
                          const float* rstd, const float* w, const float* b, float* bsums, int G, int64_t pix_per_group,
                          int C, int act, float* ws, int64_t ws_elems, void* stream);
/* backward pass 2: dx = w*rstd*(dz - s1/n - xhat*s2/n) with (s1,s2) from bsums (G,C,2) and n = `count` (pixels the
   statistics were taken over: pix_per_group for IN, all pixels x world for BN).  use_batch_stats=0 => dx = dz*w*rstd */
int du_norm_act_bwd_dx(int dtype, const void* x, int64_t ldx, const void* dy, int64_t lddy, void* dx, int64_t lddx,
                       const float* mean, const float* rstd, const float* w, const float* b, const float* bsums, int G,
                       int64_t pix_per_group, int C, int act, float count, int use_batch_stats, void* stream);

/* ---- multi-scale deformable attention ---------------------------------------------------------------- */
/* value (N, S, M, D); spatial_shapes (L,2) int64 (H,W); level_start_index (L) int64; sampling_loc (N,Lq,M,L,P,2)
   as (x,y) in [0,1]; attn_weight (N,Lq,M,L,P); out (N,Lq,M*D).  value/out dtype = `dtype`; loc/weights fp32 when
   dtype is BF16, same as value when F32.  Semantics: ms_deform_im2col_cuda.cuh:242-304. */
int du_msda_forward(int dtype, const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                    const float* sampling_loc, const float* attn_weight, void* out, int N, int S, int M, int D, int L,
                    int Lq, int P, void* stream);
/* grad_value (fp32, N,S,M,D), grad_sampling_loc, grad_attn_weight (fp32): OVERWRITTEN -- the caller need not zero them (the
   reference allocates them with at::zeros, ms_deform_attn_cuda.cu:126-128; here the library clears grad_value itself on the
   paths that scatter into it atomically and writes every element of the other two). */
/* Optional scratch `ws` (du_msda_bwd_ws_elems floats, 0 = not needed for this shape): the LDS-resident kernel then writes one
   partial grad_value plane per query chunk with plain stores and a second kernel sums them into grad_value (OVERWRITING it)
   instead of flushing every chunk with global fp32 atomics. */
int64_t du_msda_bwd_ws_elems(int N, int S, int M, int D, int L, int Lq, int P);
int du_msda_backward(int dtype, const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                     const float* sampling_loc, const float* attn_weight, const void* grad_out, float* grad_value,
                     float* grad_sampling_loc, float* grad_attn_weight, int N, int S, int M, int D, int L, int Lq, int P,
                     float* ws, int64_t ws_elems, void* stream);

/* fp64 forms: the reference extension dispatches AT_DISPATCH_FLOATING_TYPES (fp32 and fp64, ops/src/cuda/ms_deform_attn_cuda.cu:69,139) and
   its acceptance script feeds .double() tensors through MSDeformAttnFunction and torch.autograd.gradcheck (ops/test.py:40-58,101-121).
   Every tensor double (value, sampling_loc, attn_weight, out / the three gradients); same semantics; grad_value is cleared by the library. */
int du_msda_forward_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start_index, const double* sampling_loc,
                        const double* attn_weight, double* out, int N, int S, int M, int D, int L, int Lq, int P, void* stream);
int du_msda_backward_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start_index, const double* sampling_loc,
                         const double* attn_weight, const double* grad_out, double* grad_value, double* grad_sampling_loc,
                         double* grad_attn_weight, int N, int S, int M, int D, int L, int Lq, int P, void* stream);

/* The same with grad_value written in bf16 (value's dtype: what the value projection's backward reads) -- no cast pass afterwards.
   bf16, one level, 4 points, D <= 32 only (the MFMA grad_value path); DU_ERR_UNSUPPORTED otherwise: use du_msda_backward and cast. */
int du_msda_backward_bf16gv(const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index, const float* sampling_loc,
                            const float* attn_weight, const void* grad_out, void* grad_value_bf16, float* grad_sampling_loc,
                            float* grad_attn_weight, int N, int S, int M, int D, int L, int Lq, int P, float* ws, int64_t ws_elems,
                            void* stream);

/* MSDeformAttn glue (ms_deform_attn.py:188-197, single level): raw (rows, M*P*2 + M*P) = [offsets | logits] from the
   fused sampling_offsets/attention_weights GEMM; ref (Lq, 2) reference points (x, y);
   loc (rows, M, P, 2) = ref + off / (Ws, Hs); attn (rows, M, P) = softmax over P. */
int du_msda_prep(int dtype, const void* raw, int64_t ldr, const float* ref, float* loc, float* attn, int64_t rows, int Lq,
                 int M, int P, int Hs, int Ws, void* stream);
int du_msda_prep_bwd(int dtype, const float* attn, const float* gloc, const float* gattn, void* graw, int64_t ldr,
                     int64_t rows, int M, int P, int Hs, int Ws, void* stream);

/* ---- small conv-side ops (NHWC; `*bs` = per-image stride in elements so token ranges can be viewed as images) ---- */
/* y = act(dwconv3x3(x) + bias); z (nullable) receives the pre-activation. w: (C,3,3) fp32. */
int du_dwconv3x3_fwd(int dtype, const void* x, int64_t ldx, int64_t xbs, const float* w, const float* bias, void* y,
                     int64_t ldy, int64_t ybs, void* z, int B, int H, int W, int C, int act, void* stream);
int du_dwconv3x3_bwd_data(int dtype, const void* dy, int64_t lddy, int64_t dybs, const float* w, void* dx, int64_t lddx,
                          int64_t dxbs, int B, int H, int W, int C, void* stream);
/* dw (C,9) and db (C, nullable).  With scratch `ws` (>= du_dwconv_wgrad_ws_elems floats) the result is written (accumulate = 0)
   or added to dw/db (accumulate = 1: further image segments of the same kernel); with ws == NULL partials are added with atomics
   into buffers the caller zero-filled. */
int64_t du_dwconv_wgrad_ws_elems(int dtype, int B, int H, int W, int C);
int du_dwconv3x3_bwd_weight(int dtype, const void* x, int64_t ldx, int64_t xbs, const void* dy, int64_t lddy, int64_t dybs,
                            float* dw, float* db, int B, int H, int W, int C, float* ws, int64_t ws_elems, int accumulate,
                            void* stream);
/* The same depthwise kernel over ConvFFN's token pyramid (DWConv.forward, dinov3_adapter.py:99-109): x, y, z, dy, dx are contiguous
   (B, 21 n, C) token tensors, n = H*W/4, each image holding a (2H x 2W), an (H x W) and an (H/2 x W/2) grid back to back; one launch
   instead of one per grid.  z (nullable): pre-activation copy for the backward of `act`.  bwd_weight OVERWRITES dw (C,9) / db (C) and
   needs du_dwconv_wgrad_ws_elems(dtype, B, 21 n, 1, C) floats of scratch. */
int du_dwconv3x3_tokens_fwd(int dtype, const void* x, const float* w, const float* bias, void* y, void* z, int B, int H, int W, int C,
                            int act, void* stream);
int du_dwconv3x3_tokens_bwd_data(int dtype, const void* dy, const float* w, void* dx, int B, int H, int W, int C, void* stream);
int du_dwconv3x3_tokens_bwd_weight(int dtype, const void* x, const void* dy, float* dw, float* db, int B, int H, int W, int C, float* ws,
                                   int64_t ws_elems, void* stream);
/* MaxPool2d(3, 2, 1) on contiguous NHWC; idx (nullable, same shape as y, uint8) records the winning tap */
int du_maxpool3x3s2_fwd(int dtype, const void* x, void* y, uint8_t* idx, int B, int H, int W, int C, void* stream);
int du_maxpool3x3s2_bwd(int dtype, const uint8_t* idx, const void* dy, void* dx, int B, int H, int W, int C, void* stream);
/* out = base + bilinear_upsample(src) (align_corners=False); src (B,Hs,Ws,C) of src_dtype, base/out (B,Ho,Wo,C).  base == NULL: the
   plain resize F.interpolate(src, size=(Ho,Wo), mode='bilinear', align_corners=False) (tail of LearnableUpsampleBlock,
   dinounet_training.py:262-263) */
int du_bilinear_add_fwd(int src_dtype, int dtype, const void* src, int64_t lds_, const void* base, int64_t ldb, void* out,
                        int64_t ldo, int B, int Hs, int Ws, int Ho, int Wo, int C, void* stream);
/* data gradient of that resize: dx (B,Hs,Ws,C) from dy (B,Ho,Wo,C); gather form (deterministic, no atomics) */
int du_bilinear_resize_bwd(int dtype, const void* dy, int64_t lddy, void* dx, int64_t lddx, int B, int Hs, int Ws, int Ho, int Wo,
                           int C, void* stream);
/* dz = dy * act'(z) */
int du_act_bwd(int dtype, const void* z, const void* dy, void* dz, int64_t n, int act, void* stream);

/* ---- squeeze-excitation (SqueezeExcitation, dinounet_training.py:210-225; residual add of :438 fused) -------------- */
/* sums (B,C,2) from du_chan_stats(G=B); W1 (R,C), b1 (R), W2 (C,R), b2 (C) fp32; hidden (B,R) post-ReLU, gate (B,C) = sigmoid. */
int du_se_gate_fwd(const float* sums, float inv_count, const float* W1, const float* b1, const float* W2, const float* b2,
                   float* hidden, float* gate, int B, int C, int R, void* stream);
/* y = x * gate[b][c] (+ shortcut, nullable) over (B, P pixels, C) NHWC */
int du_se_scale_fwd(int dtype, const void* x, int64_t ldx, const float* gate, const void* shortcut, int64_t ldsc, void* y, int64_t ldy,
                    int B, int64_t P, int C, void* stream);
/* dsum (B,C,2) from du_chan_dot(dy, x).  Writes dpool (B,C) (gradient reaching every pixel through the pooling path, already
   divided by P) and the gate-MLP parameter gradients dW1 (R,C), db1 (R), dW2 (C,R), db2 (C). */
int du_se_gate_bwd(const float* dsum, const float* sums, float inv_count, const float* gate, const float* hidden, const float* W1,
                   const float* W2, float* dpool, float* dW1, float* db1, float* dW2, float* db2, int B, int C, int R, void* stream);
/* dx = dy * gate[b][c] + dpool[b][c] */
int du_se_scale_bwd(int dtype, const void* dy, int64_t lddy, const float* gate, const float* dpool, void* dx, int64_t lddx, int B,
                    int64_t P, int C, void* stream);

/* ---- trainer loss: DC_and_CE_loss (training/loss/compound_losses.py:8-56, dice.py:58-119: batch dice, no background, smooth 1e-5) ---- */
/* logits (B,K,H,W) fp32 NCHW, target (B,H*W) int64 labels, K in [2,8].  sums: 1 + 3(K-1) floats = [sum -log p_t, (I_c, P_c, G_c) c>=1].
   scratch: du_dice_ce_ws_elems floats.  Under data parallelism the caller all-reduces sums[1:] between _sums and _finish. */
int64_t du_dice_ce_ws_elems(int B, int K, int64_t HW);
int du_dice_ce_sums(const float* logits, const int64_t* target, float* sums, int B, int K, int64_t HW, float* ws, int64_t ws_elems,
                    void* stream);
/* loss (1 float) = CE - mean_c dice_c; coef (2(K-1)) = backward coefficients; npix = this rank's pixel count; grad_mult = world size
   when the dice sums were all-reduced, else 1 */
int du_dice_ce_finish(const float* sums, float* loss, float* coef, int K, int64_t npix, float smooth, float grad_mult, void* stream);
/* dlogits (B,K,H,W) fp32 = grad_out[0] * d loss / d logits (grad_out: device scalar or NULL for 1) */
int du_dice_ce_bwd(const float* logits, const int64_t* target, const float* coef, const float* grad_out, float* dlogits, int B, int K,
                   int64_t HW, void* stream);

/* ---- FAPM FiLM modulation (dinounet_training.py:427-429): z = gamma * z_specific + beta; gb (rows,2R) = [gamma|beta], z2 (rows,2R) =
        [z_shared|z_specific], z (rows,R).  Backward writes all of dgb and the z_specific half of dz2. ---- */
int du_film_fwd(int dtype, const void* gb, const void* z2, void* z, int64_t rows, int R, void* stream);
int du_film_bwd(int dtype, const void* dz, const void* gb, const void* z2, void* dgb, void* dz2, int64_t rows, int R, void* stream);

/* ---- elementwise helpers --------------------------------------------------------------------------- */
/* SwiGLU gate of an interleaved projection u (rows, 2h): out (rows, h) = silu(u[:, 2j]) * u[:, 2j+1]   (layers/ffn_layers.py:73-77) */
int du_swiglu_pairs(int dtype, const void* u, void* out, int64_t rows, int64_t h, void* stream);
/* whole samples of an fp32 (B, n_per_sample) tensor by int64 index: scatter = 0: dst[j] = src[idx[j]], 1: dst[idx[j]] = src[j]
   (batch-subset stochastic depth of the ViT-7B blocks in train mode, layers/block.py:126-187) */
int du_sample_copy(const float* src, float* dst, const int64_t* idx, int k, int64_t n_per_sample, int scatter, void* stream);
int du_cast(int src_dtype, int dst_dtype, const void* src, void* dst, int64_t n, void* stream);
/* number of slabs a DU_STORE_SLABS product with contraction K asked to run `split_k` ranges actually writes (each range is rounded up
   to whole K tiles of the engine; empty ranges are dropped): what the caller passes to du_splitk_reduce_bf16 as `splits` */
int du_gemm_slab_count(int K, int split_k);
/* out[i] = bf16( sum_{s < splits, in order} slabs[s * n + i] ): the second half of a DU_STORE_SLABS split-K product (n % 4 == 0) */
int du_splitk_reduce_bf16(const float* slabs, void* out, int splits, int64_t n, void* stream);
/* NCHW fp32 image -> NHWC `dtype` with channels zero-padded to Cpad */
int du_nchw_to_nhwc_pad(int dst_dtype, const float* src, void* dst, int B, int C, int H, int W, int Cpad, void* stream);
/* NHWC `dtype` (pixel stride ld) -> NCHW fp32 */
int du_nhwc_to_nchw_f32(int src_dtype, const void* src, int64_t ld, float* dst, int B, int C, int H, int W, void* stream);
/* 16x16/s16 patch gather: NCHW fp32 image -> (B*h*w, C*256) rows ordered (c, dy, dx) like Conv2d weight.flatten(1) */
int du_patchify16(int dst_dtype, const float* src, void* dst, int B, int C, int H, int W, void* stream);

/* One launch that turns every trainable fp32 weight into its kernel-ready form (bf16 cast / im2col column order / flipped data-gradient
   order / ConvTranspose layouts / concatenations).  table: n rows of 8 int64 on the DEVICE [src, dst, kind | dst_is_f32 << 8, A, B, T,
   Cp, n_out] (kinds: see elementwise.hip), bprefix: n+1 exclusive prefix sums of ceil(n_out / 4096) = workgroups per row. */
int du_pack_weights(const int64_t* table, const int64_t* bprefix, int n, int64_t nblocks, void* stream);

/* ---- GPU-side training augmentation for 2D slices: the transforms of nnUNetTrainer.get_training_transforms
   (dinounet/training/nnUNetTrainer/nnUNetTrainer.py:684-776; the reference delegates them to the un-vendored `batchgenerators` package:
   parity unpinned, algorithms restated from the package, see csrc/augment.hip).  All tensors NCHW fp32 on the device; a "plane" is one
   (sample, channel) image of n = H * W elements; per-plane parameter arrays live on the device. -------------------------------------- */
/* SpatialTransform (rotation + isotropic scale + centre crop) and MirrorTransform in one resampling pass.  params (B, 6) =
   [m00, m01, m10, m11, flip_y, flip_x]: input coordinate = input centre + M (output coordinate - output centre), M = scale * R(angle).
   data (B,C,Hi,Wi) -> data_out (B,C,Ho,Wo) bicubic, zero outside; seg (B,1,Hi,Wi) (nullable) -> seg_out by the order-1 label rule. */
int du_aug_spatial(const float* data, const float* seg, const float* params, float* data_out, float* seg_out, int B, int C, int Hi, int Wi,
                   int Ho, int Wo, void* stream);
/* stats (planes, 4) = mean, population std, min, max of every plane (n % 4 == 0) */
int du_aug_plane_stats(const float* x, float* stats, int planes, int64_t n, void* stream);
/* GaussianNoiseTransform + BrightnessMultiplicativeTransform: x = (x + sigma[p] * N(0,1)) * mult[p]   (sigma 0 / mult 1 = off) */
int du_aug_noise_mult(float* x, const float* sigma, const float* mult, int planes, int64_t n, int seed, void* stream);
/* ContrastAugmentationTransform (preserve_range): x = clip((x - mean) * factor[p] + mean, min, max), stats of the current contents */
int du_aug_contrast(float* x, const float* factor, const float* stats, int planes, int64_t n, void* stream);
/* GammaTransform body on x (or on -x where invert[p] != 0): ((v - min) / (range + 1e-7))^gamma[p] * range + min; gamma <= 0 = off.
   retain_stats and the final negation are a du_aug_plane_stats + du_aug_affine pair */
int du_aug_gamma(float* x, const float* gamma, const float* invert, const float* stats, int planes, int64_t n, void* stream);
int du_aug_affine(float* x, const float* a, const float* b, int planes, int64_t n, void* stream);
/* GaussianBlurTransform: one separable pass along axis 0 (y) / 1 (x), radius int(4 sigma + 0.5), reflect borders; sigma[p] <= 0 copies */
int du_aug_blur(const float* x, float* y, const float* sigma, int planes, int H, int W, int axis, void* stream);
/* SimulateLowResolutionTransform: nearest-neighbour down to round(size * zoom[p]), cubic back up; zoom outside (0, 1) copies */
int du_aug_lowres(const float* x, float* y, const float* zoom, int planes, int H, int W, void* stream);

/* Segmentation head (the decoder's last 1x1 convolution, 32 channels -> K <= 4 classes; dinounet_training.py:603-629, nnU-Net UNetDecoder
   seg_layers): one streaming pass instead of a padded GEMM + layout passes.  bf16 activations, weights rounded to bf16 as autocast does.
   du_seg_head_fwd: x (B, HW, 32) NHWC bf16 (pixel stride ldx), w (K, 32) fp32, bias (K) fp32 or null -> out (B, K, HW) fp32 (NCHW logits).
   du_seg_head_bwd: dl (B, K, HW) fp32 -> dx (B, HW, 32) bf16 (nullable), dwb = [dw (K, 32), db (K)] fp32 (K * 33 floats rounded up to an
   even count, OVERWRITTEN); part = du_seg_head_bwd_blocks(B, HW) rows of that length, scratch.  DU_ERR_UNSUPPORTED for other C / K. */
int du_seg_head_fwd(const void* x, int64_t ldx, const float* w, const float* bias, float* out, int B, int64_t HW, int C, int K, void* stream);
int du_seg_head_bwd_blocks(int B, int64_t HW);
int du_seg_head_bwd(const void* x, int64_t ldx, const float* w, const float* dl, void* dx, int64_t lddx, float* part, float* dwb, int B,
                    int64_t HW, int C, int K, void* stream);

/* library self-description */
const char* du_version(void);
int du_device_ok(void); /* 1 if the current device is gfx950 */
/* Kernel-selection knobs for measurement tools (within-process A/B runs, tools/gemm_p8_bench.py); results never depend on them.
   key 0: 256 x 256 multi-phase NT GEMM (gemm_p8.hip): -1 heuristic (default), 0 never, 1 wherever legal;
   key 1: its pinned issue order on (1, default) / off (0);  key 2: its tile band height (default 4);
   key 3: epilogue ablations of those kernels (timing only);  key 4: attention tile-program ablation bits (0 = the product kernel; tools/attn_ablate.py);
   key 5: weight gradients (contraction-major operands, split-K) on the multi-phase kernel: 1 where it pays (default), 2 wherever legal,
          0 never (the 128 x 128 kernel);
   key 10: the persistent 256 x 128 NT kernel: 1 where its cost model wins (default), 0 never, 2 wherever legal;
   key 12: the resident-weights streaming kernel for K <= 256 (csrc/gemm_rk.hip): 1 = K <= 192 (default), 2 = K = 256 too, 0 = never;
   key 13: 3 x 3 weight gradients (du_conv3x3_wgrad_halo): 1 = the round-5 kernel for 32 / 64 output channels (default), 2 = for 128 too,
           0 = the round-3 one;
   key 15: the <= 64 ragged rows behind the last full 256-row tile of a tall NT product (the ViT: M = 8 x 1029): 0 = by extra workgroups
           behind the tile grid (default), 1 = by the tile workgroups themselves once their tiles are done;
   key 16: fp32-result NT products with 64-128 tiles of 256 x 256 and K >= 1024 (the ViT's proj / fc2) as K-split pairs of workgroups that
           exchange fp32 halves inside the launch (needs du_gemm_args.ks_ws): 0 = off (default), 1 = on;
   key 17: the ragged-row units of a product with K >= 2048 on the one-shot tile kernels (the ViT's fc2) as (32 columns, K slice) pairs that
           meet through du_gemm_args.ks_ws: 1 (default), 0 = one unit per 32 columns walks the whole contraction;
   key 18: du_layernorm_fwd with two rows per wave: 1 = where the rows overflow one round of waves (32 per CU) by less than 2x -- the ViT's
           8232 rows (default), 0 = never, 2 = always;
   key 14: bf16 products with a bf16 residual on the persistent kernel (the residual as two more K-steps): 1 (default), 0 = one-shot kernels;
   key 9: number of independent products the caller keeps in flight on DIFFERENT streams (default 1; dinounet_amd runs the frozen ViT as
          two half-batch chains): du_gemm's tile choice then counts workgroup rounds on 256 / value CUs. */
int du_set_option(int key, int value);
/* tuning aid: the 8 per-segment cycle sums of the last probed attention launch (du_set_option(4, bits | 64), tools/attn_ablate.py) */
int du_debug_attn_probe(uint64_t* host8);
/* tuning aid: workgroups per CU the runtime's occupancy query admits for an attention kernel (0: 64 queries per wave, d_head 64;
   1: d_head 128; 2: the round-3 kernel) */
int du_debug_attn_occupancy(int which);
/* tuning aid: 8 words (loop begin tick, loop end tick, HW_ID, XCC_ID, kernel entry tick, exit tick, 2 spare) of the first n (<= 1024) workgroups
   of the last probed launch of the 64-queries-per-wave kernel */
int du_debug_attn_census(uint64_t* host, int n);

/* ---- sliding-window inference: predicted_logits[sl] += prediction * gaussian; n_predictions[sl] += gaussian
   (dinounet/inference/predict_from_raw_data.py:607-608) for a batch of nb windows, then predicted_logits /= n_predictions (:610).
   logits (nb, K, ph, pw) fp32; gauss (ph, pw) fp32; coords (nb, 3) int32 = (slice d, row y0, column x0) of each window;
   pred (K, D, H, W) and npred (D, H, W) fp32 accumulators, zero-initialised by the caller before the first window. */
int du_window_accumulate(const float* logits, const float* gauss, const int32_t* coords, float* pred, float* npred, int nb, int K,
                         int ph, int pw, int D, int H, int W, void* stream);
int du_window_normalize(float* pred, const float* npred, int K, int64_t n, void* stream);

/* ---- fused clip_grad_norm_ + Nesterov SGD over all trainable tensors (replaces
   torch.nn.utils.clip_grad_norm_(params, 12) + torch.optim.SGD.step(), dinounet/training/nnUNetTrainer/nnUNetTrainer.py:486,922-924).
   table: n_tensors rows of 4 x int64 [param ptr, grad ptr, momentum-buffer ptr, numel] (fp32 tensors, contiguous);
   bprefix[i] = first workgroup of row i, bprefix[n_tensors] = nblocks, 4096 elements per workgroup;
   hyper (DEVICE, fp32): [lr, momentum, weight_decay, max_norm, nesterov (0/1)] -- read at run time, so a captured graph follows a
   learning-rate schedule; ws: du_clip_sgd_ws_elems(nblocks) floats, ws[nblocks] = total gradient norm, ws[nblocks+1] = clip coefficient
   on return.  Gradients are scaled in place by the coefficient (as clip_grad_norm_ does); zero-filled momentum buffers on the first
   step reproduce torch's "buf = d_p". */
int64_t du_clip_sgd_ws_elems(int nblocks);
int du_clip_sgd(const int64_t* table, const int64_t* bprefix, int n_tensors, int nblocks, const float* hyper, float* ws,
                int64_t ws_elems, void* stream);

#ifdef __cplusplus
}
#endif
#endif
