/* pixart_hip.h — C ABI of libpixart_hip.so: hand-written gfx950 (MI355X, CDNA4) kernels for the PixArt-Sigma
 * denoiser hot path (PixArtMS / PixArtMSBlock forward + backward + AdamW).
 *
 * The reference (PixArt-alpha/PixArt-sigma) has NO native/FFI boundary: its kernels live in third-party wheels
 * (xformers FMHA, cuBLAS/cuDNN through torch.nn).  The seam this library replaces is therefore the reference's
 * *operator call sites*; each entry point below cites the reference lines whose math it computes
 * (paths relative to the reference repo root).  The Python binding a maintainer adds is a ctypes stub, shown in
 * INTEGRATION.md and implemented in pixart_sigma_amd/lib.py.
 *
 * Conventions
 *   - plain C: device pointers + sizes + a hipStream_t; no torch types.  The caller owns every buffer (including
 *     saved activations and workspaces); kernels never allocate, never synchronise, and launch only on `stream`.
 *   - return 0 on success, negative on error; pxa_last_error() returns a thread-local message.  No global state.
 *   - "bf16" pointers are `void*` to 16-bit bfloat16; everything else is float32 / int32.
 *   - row-major tensors exactly as the reference lays them out: tokens (B*N, C), qkv (B, N, 3, H, 72).
 */
#ifndef PIXART_HIP_H
#define PIXART_HIP_H

#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PXA_ABI_VERSION 9
/* Kernels that fuse a bias-gradient column sum add into one of PXA_COLSUM_SLOTS partial rows ([slot][stride] fp32, caller-zeroed),
 * chosen per sample / row tile, so no address sees thousands of atomics; pxa_colsum_reduce folds the partials into the gradient. */
#define PXA_COLSUM_SLOTS 16

const char* pxa_last_error(void);
int pxa_abi_version(void);
/* 16-bit operand type of this build of the library: 0 = bfloat16 (libpixart_hip.so), 1 = IEEE half (libpixart_hip_f16.so).  Every
 * "bf16" pointer of this header means "the operand type"; fp32 pointers are fp32 in both builds. */
int pxa_operand_dtype(void);
int pxa_device_info(int* cu_count, int* is_gfx950);

/* ---------------------------------------------------------------------------------------------- GEMM family
 * Replaces torch.nn.Linear forward/backward at: attn.qkv / attn.proj (PixArt_blocks.py:130,155-156),
 * cross_attn.q_linear / kv_linear / proj (PixArt_blocks.py:47-48,54-55), mlp.fc1 / fc2 (timm Mlp, PixArtMS.py:66-67,77),
 * y_embedder.y_proj (PixArt_blocks.py:385,406), final_layer.linear (PixArt_blocks.py:220).
 * layout 0 (NT): C[m][n] = sum_k A[m][k]*B[n][k]   y = x W^T          A:(M,K) B:(N,K)
 * layout 1 (NN): C[m][n] = sum_k A[m][k]*B[k][n]   dX = dY W          A:(M,K) B:(K,N)
 * layout 2 (TN): C[m][n] = sum_k A[k][m]*B[k][n]   dW = dY^T X        A:(K,M) B:(K,N)
 * act 0: none; 1: GELU(tanh) applied after bias (pre-activation optionally stored to out2_bf16);
 * act 2: multiply by GELU'(aux[m][n]) (aux = saved pre-activation) — the fc1 backward input gradient;
 * act 3: GELU(tanh) after bias with GELU'(pre-activation) stored to out2_bf16 (required) — the training forward of fc1:
 *        the derivative shares the sigmoid of the activation, so saving it costs 5 VALU ops where recomputing it in the
 *        backward epilogue costs 12 with every accumulator live;
 * act 4: multiply by aux[m][n] (aux = the derivative saved by act 3) — the fc1 backward input gradient of that path;
 * act 5: add aux[m][n] after bias — the residual connection of a ResnetBlock2D folded into its second convolution (VAE). */
typedef struct {
  const void* A; const void* B;  /* bf16 */
  int lda, ldb;
  int M, N, K;
  int layout;
  const float* bias;             /* [N] or NULL */
  int act;
  const void* aux; int ldaux;    /* bf16 [M][N], act == 2, 4 or 5 */
  void* out_bf16; void* out2_bf16; int ld_out;
  float* out_f32; int ld_f32;
  int accumulate;                /* out_f32: 0 = store, 1 = atomicAdd (gradient accumulation / split-K) */
  int split_k;                   /* >1 only with accumulate; 0 = library picks tile shape and split for the dW case */
  float* splitk_ws;              /* optional caller-owned workspace for split-K partial slabs (>= split*M*N floats, see    */
  long splitk_ws_elems;          /*  pxa_gemm_splitk_ws_elems); NULL -> partials are combined with fp32 atomics (slow)       */
  float* colsum;                 /* optional slotted partials: += column sums of the bf16 output (bias gradient)              */
  long colsum_stride;
  int k_seg;                     /* segmented-K A operand (layout NT only, 0 = plain): element k of row m is read from          */
  long a_seg_stride;             /*  A[m*lda + (k / k_seg)*a_seg_stride + k % k_seg]; k_seg a multiple of 64 dividing K.       */
                                 /*  The implicit 3x3 convolution of the VAE kernel set: see pxa_vae_* below.                  */
  int k_tap;                     /* 0: K runs segment-major as above.  > 0 (= channels C of a 3x3 convolution; k_seg = 3*k_tap,  */
                                 /*  K = 9*k_tap): K is ordered [C/64 chunks][3 kernel rows][3 taps][64] and element k is read   */
                                 /*  from A[m*lda + row*a_seg_stride + tap*k_tap + chunk*64 + k%64] - the same patch, visited    */
                                 /*  so that all nine reads of a pixel's 64-channel chunk happen within 18 k-units (L2-resident) */
                                 /*  instead of up to K/3 apart; B's rows follow the same K order.                               */
  float* gn_part;                /* optional, implicit convolutions (k_seg) with a bf16 output only: GroupNorm statistics of the  */
  int gn_img_rows;               /*  output folded into the epilogue.  Output row m is the padded-grid pixel pix = m % gn_img_rows */
  int gn_row_pitch, gn_h, gn_w;  /*  of image m / gn_img_rows (gn_img_rows a multiple of 256, >= (gn_h+2)*gn_row_pitch,            */
                                 /*  gn_row_pitch = gn_w+2); the interior pixels (1 <= pix / pitch <= gn_h, 1 <= pix % pitch       */
                                 /*  <= gn_w) add their stored (rounded) outputs and squares, per quad of adjacent channels, into   */
                                 /*  gn_part[slot][image][N/4][2] fp32 (PXA_COLSUM_SLOTS slots, caller-zeroed);                    */
                                 /*  pxa_vae_gn_finalize turns the partials into mean / rstd (groups are multiples of 4 channels).  */
                                 /*  Replaces the separate statistics pass over the output.                                        */
  int items_descending;          /* (ABI 8, round 5) persistent NT / NN kernels: 1 = every XCD walks its range of output tiles from the  */
                                 /*  LAST row tile to the first.  For a launch whose A operand was written by a kernel that swept the   */
                                 /*  token rows in ascending order: the rows written last are the ones still in the 256 MB Infinity     */
                                 /*  Cache, and this launch's own output then ends with the FIRST rows - fresh for an ascending         */
                                 /*  consumer.  Results are bit-identical either way.  0 = ascending.                                   */
  int up_row_pitch;              /* (ABI 9, round 6) > 0: ONE PHASE of a 3x3 convolution over a 2x nearest-upsampled input, computed on the LOW-RES grid */
  int up_img_rows;               /*  (diffusers Upsample2D: F.interpolate(scale 2, nearest) then Conv2d(3, padding 1); decoder call site reference        */
  int up_dy, up_dx;              /*  scripts/inference.py:136).  Output pixel (2y+dy, 2x+dx) of the upsampled convolution sees only a 2 x 2 patch of       */
                                 /*  low-res pixels, with the 3 x 3 taps that fall on the same low-res pixel summed: 4 phases x 4 taps = 16 / 36 of the     */
                                 /*  products.  The launch is an implicit convolution over the low-res padded grid (gn_* describe THAT grid; k_seg = 2 C,   */
                                 /*  K = 4 C, gn_part required); the row of interior low-res pixel (py, px) is stored at row                               */
                                 /*  image * up_img_rows + (2 py - 1 + up_dy) * up_row_pitch + (2 px - 1 + up_dx) of the output (the high-res padded grid,  */
                                 /*  row pitch up_row_pitch = 2 gn_w + 2), border rows are not stored; statistics as for gn_part.  0 = off.                */
} pxa_gemm_args;
/* Upper bound of the split-K workspace (in floats) pxa_gemm may use for an (M, N) fp32-accumulate output. */
long pxa_gemm_splitk_ws_elems(int M, int N);
int pxa_gemm(const pxa_gemm_args* args, hipStream_t stream);
/* How the persistent 256 x 256 kernels of the token GEMMs (NT / NN, 16-bit output) hand their items to the workgroups:
 *   0 (default)  static split - workgroup b takes items b, b + #CUs, ... : the fastest when the GEMM owns the GPU (training step on one GPU: -4.7 ms of 434,
 *                profiles/r4_38_step_ab_gemm_sched.txt);
 *   1            dynamic per-XCD cursors - a workgroup that starts late (a collective of the data-parallel all-reduce holds its CU: the reference overlaps
 *                DDP's bucket all-reduce with the backward, train_scripts/train.py:128-133 through accelerate) leaves its items to the others instead of
 *                doubling the kernel's time (profiles/r03*_contention*).  The data-parallel runtime (pixart_sigma_amd/dp.py) switches it on for world size > 1.
 * Process-wide; returns the previous setting.  Environment overrides for A/B runs: PXA_GEMM_STATIC=1 / PXA_GEMM_DYNAMIC=1 (read at the first GEMM). */
int pxa_gemm_set_dynamic_items(int on);

/* ---------------------------------------------------------------------------------------------- adaLN-single rows
 * x' = x + gate*u (u bf16, gate per sample; either may be NULL), optional bf16 copy of x' (cross-attn input),
 * xn = LayerNorm(x', no affine, eps) * (1 + scale) + shift  -> bf16.
 * Replaces nn.LayerNorm(elementwise_affine=False, eps=1e-6) + t2i_modulate + the gated residual adds
 * (PixArtMS.py:58,64,74-77; PixArt_blocks.py:24-25,217-219).  shift/scale (gate) point at sample 0; sample b is at
 * ptr + b*mod_stride (b*gate_stride).  rows_per_batch = tokens per sample. x_out may alias x. */
int pxa_ln_mod_fwd(const float* x, const void* u_bf16, const float* gate, int gate_stride, const float* shift, const float* scale, int mod_stride,
                   float* x_out, void* xn_bf16, void* xb_bf16, float* mean, float* rstd,
                   int R, int D, int rows_per_batch, float eps, hipStream_t stream);
/* dx_out = dx_in + dLN(dy*(1+scale)) (optionally also as bf16);  dshift[b] += sum dy;  dscale[b] += sum dy*xhat  (atomic; caller zeroes).
 * dbias (optional, round 5): slotted partials (PXA_COLSUM_SLOTS x dbias_stride fp32, caller-zeroed) += column sums of dx_out - the bias gradient of the Linear
 * that dx_bf16 is the output gradient of (x2 = x1 + cross_attn.proj(...), PixArtMS.py:76): replaces a column-sum pass over dx_bf16. */
int pxa_ln_mod_bwd(const void* dy_bf16, const float* x, const float* mean, const float* rstd, const float* scale, int mod_stride,
                   const float* dx_in, float* dx_out, void* dx_bf16, float* dshift, float* dscale, int dmod_stride,
                   float* dbias, long dbias_stride, int R, int D, int rows_per_batch, hipStream_t stream);

/* q / k LayerNorm of AttentionKVCompress(qk_norm=True) (reference PixArt_blocks.py:90-92,133-134): nn.LayerNorm(D), affine, over bf16 rows
 * with an element stride (the q / k column blocks of the qkv buffer, in place when y == x).  fwd also copies the un-normalised rows to
 * xsave [R][D] (may be NULL) and writes fp32 mean / rstd [R]; bwd overwrites dy's rows with dx (dx may alias dy) and ADDS the weight / bias
 * gradients into dw / db [D]. */
int pxa_ln_affine_fwd(const void* x_bf16, long x_stride, const float* w, const float* b, void* y_bf16, long y_stride, void* xsave_bf16,
                      float* mean, float* rstd, int R, int D, float eps, hipStream_t stream);
int pxa_ln_affine_bwd(const void* dy_bf16, long dy_stride, const void* xsave_bf16, const float* mean, const float* rstd, const float* w,
                      void* dx_bf16, long dx_stride, float* dw, float* db, int R, int D, hipStream_t stream);
/* g = dx (+ add_bf16);  dx_out = g (optional);  du = gate*g (bf16; plain cast if gate NULL);  dgate[b] += sum g*u;
 * dbias[b % PXA_COLSUM_SLOTS][d] += sum_rows du (optional slotted partials: bias gradient of the Linear whose output gradient du is). */
int pxa_gate_bwd(const float* dx, const void* add_bf16, const void* u_bf16, const float* gate, int mod_stride,
                 float* dx_out, void* du_bf16, float* dgate, int dmod_stride, float* dbias, long dbias_stride, int R, int D,
                 int rows_per_batch, hipStream_t stream);
/* out[i] += sum_p part[p*stride + i], p < PXA_COLSUM_SLOTS, i < n. */
int pxa_colsum_reduce(const float* part, long stride, float* out, long n, hipStream_t stream);
/* out[n] += sum_r dY[r][n]  — nn.Linear bias gradients. */
int pxa_colsum_bf16(const void* dy_bf16, int ld, float* out, int R, int N, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------- attention
 * softmax(q k^T * scale) v with head_dim 72; replaces xformers.ops.memory_efficient_attention at
 * PixArt_blocks.py:153 (self-attention, optionally against KV-compressed tokens: Nk != Nq) and at
 * PixArt_blocks.py:52-53 (cross-attention under BlockDiagonalMask.from_seqlens([N]*B, y_lens): pass kv_start/kv_len).
 * All strides are in elements: *_bs batch, *_ts token, *_hs head.  With kv_start != NULL the K/V (and dK/dV) batch
 * strides are ignored and sample b uses rows kv_start[b] .. kv_start[b]+kv_len[b] of the packed K/V.
 * lse: [B][H][Nq] float, log2 domain (written by fwd, read by bwd).  delta: [B][H][Nq] float workspace (bwd).
 * d_o has the layout of o. */
typedef struct {
  const void* q; const void* k; const void* v; void* o;
  const void* d_o; void* dq; void* dk; void* dv;
  float* lse; float* delta;
  long q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts;
  int q_hs, k_hs, v_hs, o_hs;
  long dq_bs, dq_ts, dk_bs, dk_ts, dv_bs, dv_ts;
  int dq_hs, dk_hs, dv_hs;
  int B, H, Nq, Nk, head_dim;
  const int* kv_start; const int* kv_len;  /* device int32 [B] or NULL */
  int max_kv_len;                           /* host-known max(kv_len) for the varlen backward grid (0: use Nk) */
  float scale;
  float* dq_colsum; float* dk_colsum; float* dv_colsum;  /* optional (bwd) slotted partials: += column sums of dq / dk / dv (bias gradients) */
  long colsum_stride;
  void* bwd_stats;   /* optional (bwd) workspace of pxa_attn_bwd_stats_bytes(B, H, Nq) bytes, 16-byte aligned: lse / delta as operand-type rows, which the
                        dK/dV kernel takes through its matrix products (round 3).  NULL: the round-2 dK/dV kernel runs. */
  int q_prescaled;   /* 0: q holds the queries as the reference's qkv linear produces them (PixArt_blocks.py:130-131).  1 (round 5): q holds
                        (scale * log2 e) * queries - the caller folded the softmax scale into the projection that produced q (engine.py: a copy of the
                        qkv weight's q rows times that constant, ONE rounding of the scaled query) - so the kernels exponentiate q k^T as it comes out of
                        the matrix product; lse keeps its meaning, dq is still the gradient with respect to the UNSCALED queries (what the projection's
                        backward needs), dk / dv are unchanged.  Same results up to the rounding point of q. */
} pxa_attn_args;
int pxa_attn_fwd(const pxa_attn_args* args, hipStream_t stream);
/* backward: delta pre-pass, then the dQ kernel (skipped when dq == NULL) and the dK/dV kernel (skipped when dk == dv == NULL) */
int pxa_attn_bwd(const pxa_attn_args* args, hipStream_t stream);
long pxa_attn_bwd_stats_bytes(int B, int H, int Nq);

/* ---------------------------------------------------------------------------------------------- token boundary
 * PatchEmbed conv (k=2,s=2) + bias + pos_embed -> fp32 tokens (PixArtMS.py:38-44,184); its weight/bias gradient;
 * unpatchify 'nhwpqc->nchpwq' (PixArtMS.py:236-248) and the inverse permutation of the output gradient (bf16);
 * packed caption-row gather = masked_select + CaptionEmbedder.token_drop (PixArtMS.py:196-204, PixArt_blocks.py:389-398). */
int pxa_patch_embed_fwd(const float* x, const float* w, const float* bias, const float* pos, float* out,
                        int B, int C, int Hl, int Wl, int D, hipStream_t stream);
int pxa_patch_embed_bwd(const float* x, const float* dtok, float* dw, float* dbias, int B, int C, int Hl, int Wl, int D, hipStream_t stream);
int pxa_unpatchify_fwd(const float* lin, float* img, int B, int h, int w, int Co, hipStream_t stream);
int pxa_patchify_bwd(const float* dimg, void* dlin_bf16, int B, int h, int w, int Co, hipStream_t stream);
int pxa_gather_rows_bf16(const float* src, const float* alt, const int* row_idx, const int* drop, void* out_bf16,
                         int rows, int L, int Cw, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------- KV token compression
 * AttentionKVCompress.downsample_2d 'conv' mode (PixArt_blocks.py:84-89,97-121): depthwise Conv2d(C,C,k=sr,s=sr) over
 * the (H,W) token grid + affine LayerNorm(C, eps 1e-5); the same parameters process K and V.
 * in: bf16 rows with token stride in_ts (e.g. the k / v slice of qkv), out: bf16 (B, (H/sr)*(W/sr), C). */
int pxa_kv_compress_fwd(const void* in_bf16, long in_bs, long in_ts, const float* conv_w, const float* conv_b,
                        const float* ln_w, const float* ln_b, void* out_bf16, int B, int H, int W, int C, int sr, float eps,
                        hipStream_t stream);
/* Backward of the above for one of K / V: din (same strided layout as in) <- gradient w.r.t. the sr*sr source tokens of every
 * compressed token; d_conv_w / d_conv_b / d_ln_w / d_ln_b += parameter gradients (atomic; the parameters are shared by K and V). */
int pxa_kv_compress_bwd(const void* dyc_bf16, const void* in_bf16, long in_bs, long in_ts, const float* conv_w, const float* conv_b,
                        const float* ln_w, void* din_bf16, long din_bs, long din_ts, float* d_conv_w, float* d_conv_b, float* d_ln_w,
                        float* d_ln_b, int B, int H, int W, int C, int sr, float eps, hipStream_t stream);
/* 'uniform' and 'ave' sampling (PixArt_blocks.py:110-115; nearest interpolation == strided pick): token (r*sr, c*sr) of the
 * (H,W) grid.  backward=0: dst (B, nH*nW, C) <- picked rows of the strided src; backward=1: strided dst <- rows of src. */
int pxa_kv_pick(int backward, const void* src_bf16, void* dst_bf16, long full_bs, long full_ts, int B, int H, int W, int C, int sr,
                hipStream_t stream);

/* ---------------------------------------------------------------------------------------------- optimizer
 * Global grad norm + clip coefficient (accelerator.clip_grad_norm_, train.py:182) and torch.optim.AdamW
 * (configs/PixArt_xl2_internal.py:48) over flat fp32 buffers, refreshing the bf16 shadow weights in the same pass. */
int pxa_sumsq_f32(const float* x, long n, float* out, hipStream_t stream);
int pxa_clip_coef(const float* sumsq, float* out2, float max_norm, float inv_world, hipStream_t stream);
int pxa_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, long n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int step, const float* gscale, hipStream_t stream);
int pxa_cast_f32_bf16(const float* x, void* y_bf16, long n, hipStream_t stream);
/* nblocks strided copies of n_total fp32 values each (block b at src + b * src_stride -> dst + b * dst_stride), the first n_scaled of every block times
 * `scale`, rounded ONCE to the operand type (y_bf16) and / or kept fp32 (y_f32); either output may be NULL.  The forward-only copy of every block's
 * attn.qkv weight / bias whose q rows carry the softmax scale (pxa_attn_args.q_prescaled): one launch for all blocks after each optimizer step. */
int pxa_scale_copy_f32(const float* src, long src_stride, void* y_bf16, float* y_f32, long dst_stride, int nblocks, long n_scaled, long n_total,
                       float scale, hipStream_t stream);
/* Loss-scaled (fp16-operand) training: the reference's mixed_precision='fp16' (configs/PixArt_xl2_internal.py:57) runs accelerate's
 * torch.cuda.amp.GradScaler around loss.backward() / clip_grad_norm_ / optimizer.step() (train_scripts/train.py:180-184).  Here the
 * same protocol lives on the device in a 5-float record `scaler`:
 *   [0] loss scale  [1] clean steps since the last scale change  [2] found_inf of the current step  [3] optimizer steps applied
 *   [4] steps skipped.
 * pxa_clip_coef_scaled: norm = sqrt(sumsq) * inv_world / scale.  Finite: out2 = {clip coefficient * inv_world / scale, norm}, the
 * tracker advances and the scale grows by growth_factor every growth_interval clean steps.  Inf / nan: out2[0] = 0, found_inf = 1,
 * scale *= backoff_factor (GradScaler.update semantics).  pxa_adamw_step_scaled / pxa_came_step (args->scaler) do nothing when
 * found_inf is set; AdamW takes its bias-correction step count from scaler[3].                                                */
int pxa_clip_coef_scaled(const float* sumsq, float* out2, float max_norm, float inv_world, float* scaler, float growth_factor,
                         float backoff_factor, int growth_interval, hipStream_t stream);
int pxa_adamw_step_scaled(float* p, const float* g, float* m, float* v, void* p_bf16, long n, float lr, float beta1, float beta2,
                          float eps, float weight_decay, const float* gscale, const float* scaler, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------- fused diffusion loss
 * GaussianDiffusion.training_losses for IDDPM(learn_sigma=True, pred_sigma=True) = MSE + VB with the learned-range variance
 * (diffusion/model/gaussian_diffusion.py:744-855, :711-742, :280-361; diffusion_utils.py:10-88).  model_out (B, 2C, H, W), x0 / noise
 * (B, C, H, W), all fp32 contiguous; coef8: per sample the 8 schedule entries at its timestep { sqrt(abar), sqrt(1-abar), posterior_mean_coef1,
 * posterior_mean_coef2, posterior_log_variance_clipped, log(beta), sqrt(1/abar), sqrt(1/abar - 1) }; t_is_zero: 1 where t == 0 (decoder NLL
 * instead of the KL).  fwd: mse[b], vb[b] (bits per dimension; loss = mse + vb).  bwd: d_model_out = g_mse[b] d mse_b + g_vb[b] d vb_b with respect to
 * model_out (eps half from the MSE only - the VB term sees a detached eps -, variance half from the VB only).  H*W % 4 == 0.            */
int pxa_iddpm_loss_fwd(const float* model_out, const float* x0, const float* noise, const float* coef8, const int* t_is_zero, int B, int C,
                       int HW, float* mse, float* vb, hipStream_t stream);
int pxa_iddpm_loss_bwd(const float* model_out, const float* x0, const float* noise, const float* coef8, const int* t_is_zero, int B, int C,
                       int HW, const float* g_mse, const float* g_vb, float* d_model_out, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------- CAME optimizer
 * came_pytorch.CAME.step() (un-vendored dependency; the reference's CAMEWrapper subclasses it unchanged:
 * diffusion/utils/optimizer.py:15,242-246; used by every PixArt-Sigma config, e.g.
 * configs/pixart_sigma_config/PixArt_sigma_xl2_img1024_internalms.py:29) over the flat parameter store, all tensors per launch.
 * A tensor with >= 2 dims is factored: viewed as [batch][R][C] (its last two dims), state = row means [batch*R] and column means
 * [batch*C] of the second moment and of the confidence residual; 1-D tensors keep a full second moment.  Tables are built once by
 * the caller (pixart_sigma_amd/dp.py: FusedCAME) and live in device memory:
 *   tensors[t]: off = first element in the flat buffers; factored: batch, R, C and the offsets of its row state (row_off, length
 *               batch*R), column state (col_off, batch*C) and row-state means (rm_off, batch); 1-D: factored = 0, C = numel,
 *               nf_off = offset of its full second moment in nf_sq.
 *   tiles[i]:   factored: global rows [first, first+count) of tensor `tensor` (whole rows only); 1-D: elements [first, first+count).
 *   col_inv_r:  [n_cols_total] 1/R of the tensor owning each column-state entry.
 * scratch: pxa_came_scratch_elems() floats, zeroed by the call.  gscale: optional device scalar multiplying every gradient (the
 * clip coefficient of pxa_clip_coef).  Also refreshes the bf16 shadow weights.                                                   */
typedef struct { long off; int batch, R, C, factored; long row_off, col_off, rm_off, nf_off; } pxa_came_tensor;
typedef struct { int tensor, first, count, pad; } pxa_came_tile;
typedef struct {
  float* p; const float* g; float* exp_avg; void* p_bf16;
  float* sq_row; float* sq_col; float* res_row; float* res_col; float* nf_sq;
  float* scratch;
  const pxa_came_tensor* tensors; int n_tensors;
  const pxa_came_tile* tiles; int n_tiles;
  const float* col_inv_r;
  long n_cols_total, n_rm_total;
  double lr, beta1, beta2, beta3, eps0, eps1, clip_threshold, weight_decay;   /* doubles: 1 - beta is formed in fp64 like torch's alpha */
  const float* gscale;
  const float* scaler;   /* optional loss-scaler record of pxa_clip_coef_scaled: the step is skipped when scaler[2] != 0 */
} pxa_came_args;
long pxa_came_scratch_elems(long n_cols_total, long n_rm_total, int n_tensors);
int pxa_came_step(const pxa_came_args* args, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------- VAE conv stack
 * The SDXL-VAE / SD-VAE (diffusers AutoencoderKL) encode / decode path: vae.encode(...).latent_dist (train_scripts/train.py:149-153),
 * vae.decode(latent / scaling_factor).sample (scripts/inference.py:136, train_scripts/train.py:88).  Forward only (the VAE is frozen).
 * Activations are bf16 NHWC pixel grids; pixel (b, y, x) is the C-vector at  ptr + ((b*img_pitch + y*row_pitch + x + origin) * C).
 *   compact grid:      row_pitch = W,   img_pitch = H*W,         origin = 0
 *   padded-grid view:  row_pitch = W+2, img_pitch = (H+2)*(W+2), origin = W+3   (pixel (0,0) of the image inside its zero border)
 * A 3x3 stride-1 pad-1 convolution is pxa_gemm over the PADDED pixels of a zero-bordered input (layout NT, A = first padded pixel
 * minus (W+3) pixels, lda = C, M = B*(H+2)*(W+2), K = 9*C, k_seg = 3*C, a_seg_stride = (W+2)*C, B = weight as [Cout][ky][kx][Cin]):
 * output row m is the convolution centred on padded pixel m, i.e. the result is itself a padded-grid view whose border rows hold
 * garbage nobody reads.  The buffer needs W+3 readable pixels in front of and behind the padded images.
 * 1x1 convolutions and the attention projections are plain pxa_gemm calls over the rows of a grid.                              */
typedef struct { void* ptr; int B, H, W, C; int row_pitch; long img_pitch; long origin; } pxa_grid;
/* GroupNorm statistics over the interior pixels: mean / rstd [B*groups] (torch.nn.GroupNorm(groups, C, eps): biased variance).
 * ws: B*groups*2 doubles of scratch (zeroed by the call).  C/groups must be a multiple of 4, C = 8 * 2^n. */
int pxa_vae_gn_stats(const pxa_grid* x, int groups, float eps, double* ws, float* mean, float* rstd, hipStream_t stream);
/* The same (mean, rstd) from statistics a convolution's epilogue already accumulated (pxa_gemm_args.gn_part [slot][B][C/4][2]); pixels = H*W. */
int pxa_vae_gn_finalize(const float* part, int B, int C, int groups, long pixels, float eps, float* mean, float* rstd, hipStream_t stream);
/* y[b, yo, xo] = act(norm(x[b, yo/upsample, xo/upsample])): GroupNorm affine when mean != NULL, SiLU when silu, nearest-neighbour
 * 2x upsampling when upsample == 2 (diffusers Upsample2D).  Writes the interior of y only. */
int pxa_vae_gn_apply(const pxa_grid* x, const float* mean, const float* rstd, const float* gamma, const float* beta, int groups,
                     int silu, int upsample, const pxa_grid* y, hipStream_t stream);
/* Explicit patch matrix for the stride-2 / few-channel convolutions: col[(b, yo, xo)][(ky*3+kx)*C + c] (bf16, row length 9*C) =
 * act(norm(x[b, yo*stride + ky - pad, xo*stride + kx - pad])), zero outside the image.  pad 1: Conv2d(padding=1); pad 0 with
 * Ho = H/2: diffusers Downsample2D (F.pad (0,1,0,1) then stride 2, padding 0). */
int pxa_vae_im2col3x3(const pxa_grid* x, const float* mean, const float* rstd, const float* gamma, const float* beta, int groups,
                      int silu, int stride, int pad, int Ho, int Wo, void* col_bf16, hipStream_t stream);
/* out = a + b over the interior pixels (residual connections; the three grids may have different pitches). */
int pxa_vae_add(const pxa_grid* a, const pxa_grid* b, const pxa_grid* out, hipStream_t stream);
/* P[r][:] = softmax(scale * S[r][:]) with fp32 scores in, bf16 probabilities out (mid-block attention: one 512-wide head). */
int pxa_vae_softmax_rows(const float* s, long ld, void* p_bf16, long ldp, int rows, int cols, float scale, hipStream_t stream);
/* fp32 NCHW (B, C, H, W) image / latent -> bf16 grid scaled by mul, channels C..grid.C-1 zero; and back (first C channels). */
int pxa_vae_nchw_to_grid(const float* img, int C, float mul, const pxa_grid* y, hipStream_t stream);
int pxa_vae_grid_to_nchw(const pxa_grid* x, int C, float* img, hipStream_t stream);
/* Direct 3x3 stride-1 pad-1 convolution of act(norm(x)) to Cout <= 4 channels, written as an fp32 NCHW image (B, Cout, H, W): the decoder's conv_out
 * (diffusers AutoencoderKL: decoder.conv_norm_out -> SiLU -> decoder.conv_out, 128 -> 3; call site of the whole decode: reference scripts/inference.py:136).
 * One pass over x: no padded copy, no patch matrix, no crop / permute.  w_taps: [9][Cout][C] in the library's operand type (tap = ky * 3 + kx);
 * norm arguments as pxa_vae_gn_apply (mean == NULL: none); x.C a multiple of 64. */
int pxa_vae_conv3x3_small_out(const pxa_grid* x, const float* mean, const float* rstd, const float* gamma, const float* beta, int groups, int silu,
                              const void* w_taps, const float* bias, int Cout, float* img, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------- conditioning linears (fp32)
 * y = x W^T + b and its backward for the O(batch) conditioning path: TimestepEmbedder / SizeEmbedder MLPs and t_block (PixArt_blocks.py:267-344,
 * PixArtMS.py:134-137,193).  fp32 in, fp32 out (their outputs modulate every token of every block); x (M, K), W (N, K), y / dy (M, N), K a multiple of 4.
 * bwd: dx_zeroed (M, K) += dy W (caller-zeroed; NULL skips it), dw (N, K) = dy^T x and db (N) = column sums of dy (plain stores; NULL skips both). */
int pxa_linear_f32_fwd(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, hipStream_t stream);
int pxa_linear_f32_bwd(const float* dy, const float* x, const float* w, float* dx_zeroed, float* dw, float* db, int M, int N, int K, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------- measurement
 * The part's matrix rate under its power limit, for the `roofline` object of bench.py (no reference counterpart: the reference reports no roofline).
 * One launch = `iters` x 32 v_mfma_f32_32x32x16 (shape 32) or 64 v_mfma_f32_16x16x32 (shape 16) per wave on register-resident operand data, one wave per
 * SIMD on every CU, nothing else in the loop.  `operands`: pxa_mfma_rate_probe_bytes() bytes of operand-type values (the caller fills them, e.g. N(0,1):
 * the multiplier inputs' toggle rate sets the power draw and with it the clock); `sink`: one float, never written on such data; *flops_per_launch (optional)
 * receives the FLOPs the launch issues.  The caller times the launches with events on `stream`. */
long pxa_mfma_rate_probe_bytes(void);
int pxa_mfma_rate_probe(const void* operands, int shape, int iters, float* sink, double* flops_per_launch, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PIXART_HIP_H */
