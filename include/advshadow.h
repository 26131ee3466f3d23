/*
 * advshadow.h — C ABI of libadvshadow_hip.so: the MI355X (gfx950) kernels behind the
 * AdvShadow hot path (DDIM reverse sampling through the UNet eps-predictor, shadow
 * composite, victim forward, ASR/PSNR/SSIM reduction).
 *
 * The reference is pure Python/PyTorch; the "FFI" a maintainer would bind is the set of
 * torch ops its modules call.  Every entry point below names the reference call site it
 * replaces (paths relative to the reference tree).  Conventions:
 *   - plain pointers and sizes only; all tensor pointers are DEVICE pointers owned by the
 *     caller (e.g. torch tensors' data_ptr()); the library owns nothing but a 4 KiB zero page;
 *   - activations are NHWC ("channels last": [B][H][W][C]); dtype ADVS_F32, ADVS_BF16 or ADVS_F16;
 *   - every call enqueues on the given hipStream_t (passed as void*), never synchronises,
 *     allocates nothing, and is therefore legal inside stream capture;
 *   - the ONE exception to the three lines above is the handle-level group at the end (advs_unet_*, advs_ddim_run): for a
 *     host without a plan builder of its own the library keeps the network, its activations (hipMalloc) and the captured
 *     launch list behind an opaque handle, and synchronises where it says so;
 *   - return 0 on success, a negative ADVS_ERR_* otherwise; advs_last_error() has the text.
 */
#ifndef ADVSHADOW_H
#define ADVSHADOW_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ADVS_OK = 0, ADVS_ERR_ARG = -1, ADVS_ERR_HIP = -2, ADVS_ERR_STATE = -3 };
enum { ADVS_F32 = 0, ADVS_BF16 = 1, ADVS_F16 = 2 };   /* 16-bit modes: 16-bit storage + MFMA, f32 accumulation */
enum { ADVS_ACT_NONE = 0, ADVS_ACT_RELU = 1, ADVS_ACT_SILU = 2, ADVS_ACT_GELU = 3,
       ADVS_ACT_RELU6 = 4, ADVS_ACT_LRELU01 = 5, ADVS_ACT_LRELU001 = 6, ADVS_ACT_SIGMOID = 7 };
/* OR-ed into advs_groupnorm*'s `act`: y = act(norm(x)) + chan_add + residual instead of
 * act(norm(x) + residual) + chan_add -- Bottleneck's `y = conv2(conv1(x)); y = y + x`
 * (model/modules/module.py:42-47) where conv2 is Conv -> GroupNorm -> act (conv.py:96-97).   */
#define ADVS_GN_RESIDUAL_AFTER_ACT 0x100

/* ---- library ---------------------------------------------------------------------- */
int advs_init(void);                    /* allocate the zero page on the current device   */
const char* advs_last_error(void);
int advs_abi_version(void);

/* ---- weight packing ------------------------------------------------------------------
 * nn.Conv2d weight [Cout][Cin][R][S] f32 (torch layout) -> GEMM operand [Cout][R][S][Cin]
 * in `dtype`.  nn.Linear weight [N][K] is the R=S=1 case.                                */
int advs_pack_conv_weight(const float* w_oihw, void* w_packed, int cout, int cin, int r, int s,
                          int dtype, void* stream);

/* ---- layout / dtype conversion ------------------------------------------------------- */
int advs_nchw_f32_to_nhwc(const float* x, void* y, int b, int c, int h, int w, int dtype, void* stream);
int advs_nhwc_to_nchw_f32(const void* x, float* y, int b, int c, int h, int w, int dtype, void* stream);

/* ---- Conv2d as implicit GEMM on MFMA ---------------------------------------------------
 * Replaces nn.Conv2d (3x3 pad 1 stride 1|2, 1x1) at diff_model.py:73,86,90,114,115,134,148,
 * model/modules/conv.py:38,41, victim convs.  Fusions selected by the descriptor:
 *   - x2/c2: second source concatenated after x1 on the channel axis (torch.cat,
 *     diff_model.py:265; model/modules/block.py:87);
 *   - upsample: nearest x2 applied to the sources on load (F.interpolate, diff_model.py:137);
 *   - bias[n], temb[b][n] (h += time_emb(t)[:, :, None, None], diff_model.py:101),
 *     residual[m][n] (h + shortcut(x), diff_model.py:103,127), activation.
 * Requirements: c1 and c2 multiples of 64 (bf16) / 32 (f32).                               */
#define ADVS_UPSAMPLE_SUBPIXEL 2
typedef struct advs_conv_args {
    const void* x1; const void* x2;     /* NHWC sources [b][h][w][c1], [b][h][w][c2]        */
    const void* w;                      /* packed [cout][r][r][c1+c2]                       */
    const float* bias;                  /* [cout] or NULL                                   */
    const float* temb;                  /* row b at temb + b*temb_stride, or NULL           */
    const void* residual;               /* NHWC [b][ho][wo][cout] or NULL                   */
    void* y;                            /* NHWC [b][ho][wo][cout]                           */
    int b, h, w_, c1, c2, cout;         /* h,w_: stored source size (before upsample)       */
    int ksize, stride, pad, upsample;   /* ksize 1|3; upsample 0 | 1 (nearest x2 on load) | ADVS_UPSAMPLE_SUBPIXEL:
                                           the same function computed as four 2x2 convolutions of the low-res input,
                                           one per output parity (2.25x fewer MACs); w is then [4][cout][2][2][c1+c2],
                                           parity (a,b) = 2a+b, with the coinciding taps of the 3x3 kernel summed:
                                           rows {w0, w1+w2} for a = 0, {w0+w1, w2} for a = 1, columns likewise.
                                           3x3 stride 1 pad 1, h and w multiples of 16.                          */
    int act, dtype;                     /* act may carry ADVS_GN_RESIDUAL_AFTER_ACT: y = act(conv + bias) + residual */
    int temb_stride;                    /* floats between consecutive samples' temb rows (0: cout); < 0: ONE row shared by
                                           every sample -- a sampler step, where the whole batch sits at one timestep     */
    int tile;                           /* 0 = choose; 1: 128x128, 2|3: 256x128, 4: 256x256,
                                           10: 16x16-pixel halo tile (3x3 stride 1 only); 12 (implied by
                                           ADVS_UPSAMPLE_SUBPIXEL): its 4-tap sub-pixel form;
                                           15: 64x128, 16: 64x64 (chosen for maps of <= 56x56 / 14x14 pixels);
                                           17 | 18 | 19: second-generation halo tiles (16-bit, 3x3 stride 1, plain /
                                           sub-pixel / fused 1x1): 16x32 pixels x 8 waves, 16x16 x 8 waves, 16x16 x 4
                                           waves with two workgroups per CU (chosen from 64x64 maps up)          */
    float* stats;                       /* NULL, or [ceil(M/rows)][cout][2]: per row block (rows =
                                           advs_conv_tile_rows(tile), must divide ho*wo) and channel the
                                           (sum, sum of squares) of y as stored -> advs_groupnorm_stats */
    int stats_rows;                     /* the row-block height `stats` was sized for (checked)          */
    const void* e1; const void* e2;     /* extra 1x1 stride-1 operand [b][ho][wo][ce1],[...][ce2] or NULL:
                                           y = conv(x) + conv1x1(cat(e1,e2)); w rows are then
                                           [ksize*ksize*(c1+c2) | ce1+ce2] (h + shortcut(x), diff_model.py:103) */
    int ce1, ce2;
    int ld1, ld2;                       /* pixel stride of x1 / x2 in elements, 0 = c1 / c2.  ld < c lets a source
                                           with ld channels (a multiple of 16 bytes) be read as whole 128-byte slabs:
                                           the weights of channels [ld, c) must be zero (they meet the next pixel's
                                           data, or zeros past the end of the buffer) -- 32-channel bf16 layers of
                                           CSPDarkUnet (model/networks/cspdarkunet.py:24-29)                        */
    const void* relu_mask;              /* NULL, or NHWC [b][ho][wo][cout]: y is zeroed wherever relu_mask <= 0, after bias /
                                           residual / act -- ReLU backward fused into a data-gradient conv (the mask is the
                                           forward activation behind the ReLU; train_shadow.py:209 loss.backward()).
                                           Per-tap tiles only (tile 0 | 1 | 4), no stats.                                */
    const float* norm;                  /* NULL, or [b][c1+c2][2] f32 (scale, shift; times log2 e) from advs_groupnorm_affine_stats: the conv reads
                                           SiLU(GroupNorm(x)) in place of x -- GroupNorm + SiLU of the conv's input applied while
                                           the halo is staged, zero padding staying zero (norm_layer + SiLU + Conv2d, diff_model.py:70-73,
                                           83-86): no normalised tensor in HBM.  Values are rounded to the storage dtype exactly as
                                           advs_groupnorm_stats would have stored them, so the result is bit-identical to the two-pass
                                           form.  16-bit dtypes, 3x3 stride 1 pad 1, tile 19's shapes, c1 + c2 <= 384.          */
} advs_conv_args;
int advs_conv2d(const advs_conv_args* a, void* stream);
int advs_conv_set_tile(int tile);       /* tuning hook: non-zero overrides every call's tile  */
int advs_conv_resolve_tile(const advs_conv_args* a);   /* the tile id advs_conv2d will use for this descriptor */
int advs_conv_tile_rows(int tile);      /* row-block height (rows per stats entry) of a tile id: 64 per wave for the older
                                           tiles, one entry per WORKGROUP for 17 (512), 18 and 19 (256) */

/* First conv: NCHW f32 image (cin <= 4) -> NHWC `dtype`, 3x3 pad 1 (diff_model.py:192;
 * model/modules/conv.py:38 for inc).  w is the torch OIHW f32 weight.                      */
int advs_conv3x3_first(const float* x_nchw, const float* w_oihw, const float* bias, void* y,
                       int b, int cin, int h, int w, int cout, int dtype, void* stream);
/* Same, optionally leaving per-channel (sum, sum of squares) of the rounded outputs per block of
 * advs_conv_first_stats_rows() pixels -- stats[b * (h*w/rows) + block][cout][2], the layout advs_groupnorm_stats
 * folds -- so the first GroupNorm (diff_model.py:70) needs no statistics pass.  stats may be NULL.
 * advs_conv_first_stats_rows returns 0 when the shape has no statistics path (f32, cin > 3, odd sizes).      */
int advs_conv_first_stats_rows(int cin, int h, int w, int cout, int dtype);
int advs_conv3x3_first_stats(const float* x, const float* w, const float* bias, void* y, float* stats,
                             int b, int cin, int h, int wd, int cout, int dtype, void* stream);
/* Last conv: NHWC `dtype` (cin multiple of 8) -> NCHW f32 with cout <= 4 (diff_model.py:242,
 * ksize 3; model/networks/unet.py:92, ksize 1).  w is the torch OIHW f32 weight.           */
int advs_conv_last(const void* x, const float* w_oihw, const float* bias, float* y_nchw,
                   int b, int cin, int h, int w, int cout, int ksize, int dtype, void* stream);

/* ---- GroupNorm (+activation) ---------------------------------------------------------
 * nn.GroupNorm(G, C) followed by an optional activation (norm_layer+SiLU, diff_model.py:62-63,
 * 71-72,83-84,113,240-241; GroupNorm(1, C)+act, model/modules/conv.py:39-42).  eps = 1e-5.
 * `partials` is caller scratch of advs_groupnorm_scratch_bytes(b, groups) bytes.
 * Optional residual_in: y = act(residual_in + GN(x)) (DoubleConv residual, conv.py:53).     */
size_t advs_groupnorm_scratch_bytes(int b, int groups);
/* x2/c2: optional second source concatenated after x on the channel axis (the norm of
 * torch.cat([h, skip]), diff_model.py:265 -> :71); y is the concatenated [b][hw][c+c2].     */
/* chan_add (or NULL): per-sample per-channel vector added AFTER the activation,
 * y += chan_add[b*chan_add_stride + c]  (x + emb_layer(t)[:, :, None, None], block.py:47-49). */
int advs_groupnorm(const void* x, const void* x2, const float* gamma, const float* beta,
                   const void* residual_in, const float* chan_add, int chan_add_stride, void* y,
                   void* partials, int b, int hw, int c, int c2, int groups, int act, int dtype,
                   void* stream);

/* GroupNorm whose statistics come from the producing convs' epilogues (advs_conv_args.stats) instead
 * of a pass over x: stats1/stats2 are those arrays, row_blocks_per_image = ho*wo / tile rows.
 * scratch: b*groups*2 floats.                                                                    */
int advs_groupnorm_stats(const void* x, const void* x2, const float* stats1, int row_blocks_per_image1,
                         const float* stats2, int row_blocks_per_image2, const float* gamma,
                         const float* beta, const void* residual_in, const float* chan_add,
                         int chan_add_stride, void* y, void* scratch, int b, int hw, int c, int c2,
                         int groups, int act, int dtype, void* stream);

/* The (scale, shift) table of GroupNorm(groups) + affine for advs_conv_args.norm, from the same epilogue statistics
 * advs_groupnorm_stats folds: table[b][c][0] = log2(e) * rstd_g * gamma_c, [1] = log2(e) * (beta_c - mean_g * rstd_g * gamma_c)
 * (pre-scaled for the consumer's exp2-based SiLU; the table is an opaque hand-off between these two entry points).
 * scratch as for advs_groupnorm_stats.  table: b * (c + c2) * 2 floats.                                         */
int advs_groupnorm_affine_stats(const float* stats1, int row_blocks_per_image1, const float* stats2,
                                int row_blocks_per_image2, const float* gamma, const float* beta, void* scratch,
                                float* table, int b, int hw, int c, int c2, int groups, void* stream);

/* ---- resampling / token norm of the class-conditional UNet -----------------------------
 * MaxPool2d(2) (model/modules/block.py:27); y is [b][h/2][w/2][c].                          */
int advs_maxpool2(const void* x, void* y, int b, int h, int w, int c, int dtype, void* stream);
/* y = cat([skip, Upsample(scale 2, bilinear, align_corners=True)(x)], channel axis)
 * (block.py:66,86-87): skip [b][2h][2w][c1], x [b][h][w][c2] -> y [b][2h][2w][c1+c2].          */
int advs_concat_upsample2x(const void* skip, const void* x, void* y, int b, int h, int w, int c1, int c2,
                           int dtype, void* stream);
/* Same with Upsample(scale 2, mode="nearest") -- CSPDarkUpBlock (block.py:116,127-128).           */
int advs_concat_nearest2x(const void* skip, const void* x, void* y, int b, int h, int w, int c1, int c2,
                          int dtype, void* stream);
/* nn.LayerNorm([c]) over the last axis of [rows][c] (model/modules/attention.py:25,27: eps 1e-5; HF ViT: 1e-12). */
int advs_layernorm(const void* x, const float* gamma, const float* beta, void* y, long long rows, int c,
                   float eps, int dtype, void* stream);

/* ---- self-attention, flash style -------------------------------------------------------
 * softmax(q k^T / sqrt(d)) v per (batch, head) without materialising the N x N scores
 * (AttentionBlock.forward, diff_model.py:120-125; nn.MultiheadAttention core,
 * model/modules/attention.py:50).  qkv is [b][n][ld] (row = token); head hd takes
 * q/k/v at column q_off/k_off/v_off + hd*head_stride, d columns each.  out is [b][n][heads*d].
 * Any n >= 1; d a multiple of 8 (bf16) or 4 (f32), d <= 128.                             */
int advs_attention(const void* qkv, void* out, int b, int n, int heads, int d, int ld,
                   int q_off, int k_off, int v_off, int head_stride, int dtype, void* stream);

/* Token axis padded (ViT: 197 tokens in rows of 208): keys >= n_valid are masked.  */
int advs_attention_masked(const void* qkv, void* out, int b, int n, int n_valid, int heads, int d, int ld,
                          int q_off, int k_off, int v_off, int head_stride, int dtype, void* stream);

/* softmax(q k^T / sqrt(d) + bias) v: bias is f32 [bias_mod][heads][n][n] ALREADY MULTIPLIED BY log2(e); sequence i
 * uses block i % bias_mod.  Swin's window attention: relative position bias (+ the shifted-window mask of window
 * position i % nW) -- timm swin_base_patch4_window7_224 (ASR_fast.py:27-32).                                    */
int advs_attention_bias(const void* qkv, void* out, const float* bias_log2e, int bias_mod, int b, int n,
                        int heads, int d, int ld, int q_off, int k_off, int v_off, int head_stride, int dtype,
                        void* stream);
/* Swin's cyclic shift + window partition as one gather, and its inverse (+ the block's residual):
 * image side [b][h][w][c], window side [b*(h/window)*(w/window)][window*window][c]; token (ty,tx) of window (wy,wx)
 * is pixel ((wy*window+ty+shift) % h, (wx*window+tx+shift) % w).  inverse: y[pixel] = x[token] + residual[pixel].  */
int advs_window_shift(const void* x, const void* residual, void* y, int b, int h, int w, int c, int window,
                      int shift, int inverse, int dtype, void* stream);

/* ---- small dense layers of the time-embedding path (always f32) ------------------------
 * y[b][n] = bias[n] + sum_k act_in(x[b][k]) * w[n][k]   (nn.Linear after optional SiLU:
 * diff_model.py:184-188,77-80; model/modules/block.py:33-36).  w is torch layout [n][k].     */
int advs_linear_f32(const float* x, const float* w, const float* bias, float* y, int b, int k, int n,
                    int act_in, int act_out, void* stream);
/* Sinusoidal embedding of t[b] (int64, device) with a host-computed frequency table
 * freqs[half] (device): out[b] = [cos|sin] (cos_first=1: diff_model.py:16-33) or [sin|cos]
 * (cos_first=0: model/networks/base.py:56-68).  label_emb rows (or NULL) are added:
 * out[b] += emb_table[labels[b]] (model/networks/unet.py:106-107).                           */
int advs_timestep_embedding(const int64_t* t, const float* freqs, int half, int cos_first,
                            const float* emb_table, const int64_t* labels, float* out, int b,
                            void* stream);

/* ---- DDIM update -------------------------------------------------------------------------
 * One reverse step on NCHW f32 x (in place), coefficients read from a device table so the
 * same captured launch serves every step:
 *   coef[step] = { a_t, a_prev, sigma }  (f32, host-computed exactly as the reference does)
 *   eps' = eps_u + cfg*(eps - eps_u) if eps_uncond (torch.lerp, model/samples/ddim.py:89)
 *   x0 = clamp((x - sqrt(1-a_t) eps')/sqrt(a_t), -1, 1); x = sqrt(a_prev) x0 + sqrt(1-a_prev-sigma^2) eps' + sigma*noise
 * (diff_model.py:457-470, model/samples/ddim.py:91-94).  *step_counter (device int32) selects the
 * row, is written to t_out[b] as the NEXT step's timestep from tseq, and is incremented.     */
int advs_ddim_step(float* x, const float* eps, const float* eps_uncond, float cfg_scale,
                   const float* noise, const float* coef, const int64_t* tseq, int nsteps,
                   int32_t* step_counter, int64_t* t_out, int b, size_t per_sample, int clip,
                   void* stream);
/* GaussianDiffusion.p_sample (diff_model.py:361-396): the posterior step of the full-length ancestral sampler
 * behind GaussianDiffusion.sample / p_sample_loop (what main.py:124 and gen.py:562 call).  coef is [nsteps][5] =
 * {sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod, posterior_mean_coef1, posterior_mean_coef2,
 *  (t != 0) * exp(0.5 * posterior_log_variance_clipped)} per step in loop order; the step counter, tseq and t_out
 * work as in advs_ddim_step.  Bit-exact with the reference's f32 op chain.                                   */
int advs_ddpm_posterior_step(float* x, const float* eps, const float* noise, const float* coef,
                             const int64_t* tseq, int nsteps, int32_t* step_counter, int64_t* t_out,
                             int b, size_t per_sample, int clip, void* stream);

/* DDPM ancestral step (model/samples/ddpm.py:74-88): coef[step] = {alpha, alpha_hat, beta};
 *   x = 1/sqrt(alpha) * (x - ((1-alpha)/sqrt(1-alpha_hat)) * eps') + sqrt(beta) * noise (noise NULL = zeros). */
int advs_ddpm_step(float* x, const float* eps, const float* eps_uncond, float cfg_scale, const float* noise,
                   const float* coef, const int64_t* tseq, int nsteps, int32_t* step_counter,
                   int64_t* t_out, int b, size_t per_sample, void* stream);
/* PLMS linear-multistep combination of eps predictions (model/samples/plms.py:93-107); `guided` (optional)
 * receives the CFG-lerped prediction of this step, `out` the combined one.                            */
int advs_plms_combine(const float* eps, const float* eps_uncond, float cfg_scale, const float* eps_next,
                      const float* old1, const float* old2, const float* old3, int order, float* guided,
                      float* out, size_t n, void* stream);
/* uint8 image = trunc((x+1)*0.5*255) wrapped mod 256 (no clamp: model/samples/ddim.py:97-99),
 * or clamped when clamp != 0.                                                                */
int advs_to_uint8(const float* x, uint8_t* y, size_t n, int clamp, void* stream);
/* [0,1] float image -> uint8 by clamp(x*255) truncated (ToPILImage's pic.mul(255).byte()).          */
int advs_unit_to_uint8(const float* x, uint8_t* y, size_t n, void* stream);

/* ---- ViT victim backwards (SURVEY 8f rank 4 for the HF ViT of ASR_fast.py:47-58; csrc/vit_grad.hip) ------------------------
 * The data-gradient pieces `loss.backward(); image.grad` (tools/train_shadow.py:204-212) needs beyond the Linear layers (whose
 * gradients are advs_conv2d on transposed weights).  Token tensors [rows][c] in the compute dtype; f32 arithmetic inside.
 * layernorm_bwd: dx = d LayerNorm(x; gamma, eps) applied to dy, + `add` (the residual stream's gradient, may be null).          */
int advs_layernorm_bwd(const void* dy, const void* x, const float* gamma, const void* add, void* dx, long long rows,
                       int c, float eps, int dtype, void* stream);
/* exact (erf) GELU as its own pass, and its gradient dx = dy * (Phi(x) + x phi(x)) from the PRE-activation x.                    */
int advs_gelu(const void* x, void* y, long long n, int dtype, void* stream);
int advs_gelu_bwd(const void* x, const void* dy, void* dx, long long n, int dtype, void* stream);
/* Gradient of advs_attention_masked: qkv / d_qkv laid out as there, out = the forward's output, d_out its gradient ([b][n][heads*d]);
 * d <= 64.  16-bit dtypes with d and every offset a multiple of 8: two MFMA kernels that recompute the scores (csrc/attention_bwd.hip,
 * any n); otherwise (f32: the parity path) f32 VALU kernels that need 2 * n_valid * d f32 in LDS (ViT-B/16: 197 x 64).  scratch:
 * advs_attention_bwd_scratch_bytes(b, n, heads) bytes (f32 path: [2][b][heads][n][n] P and dS, transposed; MFMA path: the first
 * [2][b][heads][n] floats, row log-sum-exp and D = dO . O).                                                                       */
size_t advs_attention_bwd_scratch_bytes(int b, int n, int heads);
int advs_attention_bwd(const void* qkv, const void* out, const void* d_out, void* d_qkv, void* scratch, int b, int n,
                       int n_valid, int heads, int d, int ld, int q_off, int k_off, int v_off, int head_stride, int dtype,
                       void* stream);
/* Gradient of advs_attention_bias (Swin's windows, swin_transformer.py WindowAttention.forward): as advs_attention_bwd with the same
 * additive score bias (units of log2 e, block i % bias_mod for sequence i) inside the recomputed softmax; every token is a key.    */
int advs_attention_bias_bwd(const void* qkv, const void* out, const void* d_out, void* d_qkv, void* scratch, const float* bias_log2e,
                            int bias_mod, int b, int n, int heads, int d, int ld, int q_off, int k_off, int v_off, int head_stride,
                            int dtype, void* stream);
/* dst[b * row_stride][0..c) = src[b][0..c) (f32 -> compute dtype): the classifier head's gradient enters the CLS row.          */
int advs_scatter_row0(const float* src, void* dst, int b, long long row_stride, int c, int dtype, void* stream);
/* ---- ConvNeXt victim backwards (timm convnext_base of ASR_fast.py:21-26; csrc/convnext_grad.hip) ------------------------------
 * Data gradient of advs_dwconv2d (stride 1) plus the residual stream's gradient `add` (may be NULL): the forward gather with the
 * taps mirrored; w_taps_c is the FORWARD weight layout [k*k][c] f32.                                                              */
int advs_dwconv2d_bwd(const void* dy, const float* w_taps_c, const void* add, void* dx, int b, int h, int w, int c, int ksize,
                      int dtype, void* stream);
/* Inverse of advs_space_to_depth2: x [b][h/2][w/2][4c] -> y [b][h][w][c].                                                          */
int advs_depth_to_space2(const void* x, void* y, int b, int h, int w, int c, int dtype, void* stream);
/* Gradient of advs_global_avgpool: out[b][p][:] = g[b][:] / hw (g f32).                                                            */
int advs_avgpool_bwd(const float* g, void* out, int b, int hw, int c, int dtype, void* stream);
/* ---- EfficientNetV2-S victim backwards (torchvision efficientnet_v2_s of ASR_fast.py:59-65; csrc/effnet_grad.hip) ---------------
 * SiLU as its own pass (y = silu(x) + add, add may be NULL) and its gradient from the PRE-activation: dx = dy * silu'(x_pre).          */
int advs_silu(const void* x, const void* add, void* y, long long n, int dtype, void* stream);
int advs_silu_bwd(const void* x_pre, const void* dy, void* dx, long long n, int dtype, void* stream);
/* Data gradient of advs_dwconv2d for stride 1 or 2: dx [b][h][w][c] (the conv's input size) from dy at the output size.             */
int advs_dwconv2d_bwd_strided(const void* dy, const float* w_taps_c, void* dx, int b, int h, int w, int c, int ksize, int stride,
                              int dtype, void* stream);
/* out[b][c] = sum_p a[b][p][c] * bb[b][p][c] (f32): the gradient reaching a squeeze-excitation scale.                               */
int advs_channel_dot(const void* a, const void* bb, float* out, int b, int hw, int c, int dtype, void* stream);
/* Gradient through the squeeze-excitation gate s = sigmoid(z2): out = gs * s * (1 - s), f32 vectors of n elements; the block's two
 * Linear layers run backwards as advs_linear_f32 on transposed weights with advs_silu_bwd (f32) between them.                      */
int advs_sigmoid_gate_bwd(const float* gs, const float* s, float* out, long long n, void* stream);
/* out = (dsc * s[b][c] + dpooled[b][c] / hw) * silu'(pre): the gradient at the depthwise conv's pre-activation through y = silu(pre) * s. */
int advs_se_scale_bwd(const void* dsc, const float* s, const float* dpooled, const void* pre, void* out, int b, int hw, int c,
                      int dtype, void* stream);
/* Gradient of advs_cls_mean_rows_f32 (the DINOv2 head input, ASR_fast.py:47-58 with a Dinov2 checkpoint): dst[b][0][:] = src[b][0..c),
 * dst[b][1..np][:] = src[b][c..2c) / np; src f32 [b][2c], dst [b][n_pad][c] in the compute dtype, other rows untouched.            */
int advs_scatter_cls_mean(const float* src, void* dst, int b, int n_pad, int np, int c, int dtype, void* stream);
/* Inverse of advs_patchify_padded for gradients: image gradient (NCHW f32) from the patch-column gradients, which sit in rows
 * row_off + patch index of a [b][rows_per_image][kpad] matrix (the token layout: row_off = 1 skips the CLS row).               */
int advs_unpatchify_padded(const void* dcols, float* dx_nchw, int b, int cin, int h, int w, int patch, int kpad,
                           int rows_per_image, int row_off, int dtype, void* stream);

/* ---- shadow composite, image hand-off, metrics ------------------------------------------------
 * Closed form of apply_shadow (tools/train_shadow.py:242-256,262-266; ddim2/test.py:830-871; with
 * ntaps = 1, taps = {1}: ddim2/diff_model2.py:615-654): circular mask at centers[b] = (cx, cy) with
 * radii[b], blurred by the separable `taps` (cv2.GaussianBlur(k, sigma 0), BORDER_REFLECT_101),
 * times feature_mask; img/out NCHW f32 [b][c][h][w], feature_mask [b][mask_channels][h][w].
 * `taps` is a HOST array of ntaps (odd, <= 7) floats: it travels in the launch arguments.        */
int advs_apply_shadow(const float* img, const float* feature_mask, const float* centers, const float* radii,
                      float* out, int b, int c, int h, int w, int mask_channels, float intensity,
                      const float* taps, int ntaps, void* stream);
/* The same composite stopped half way, for the gradient attack of train_shadow.py:250-260: the shadowed image (NOT
 * clamped) and the combined mask cm = blur(shadow mask) * feature mask, both [b][c][h][w] f32.  `taps`: HOST array, as above. */
int advs_apply_shadow_parts(const float* img, const float* feature_mask, const float* centers, const float* radii,
                            float* shadowed, float* cmask, int b, int c, int h, int w, int mask_channels,
                            float intensity, const float* taps, int ntaps, void* stream);
/* out = clamp(img * (1 - cm) + adv * cm, 0, 1)  (train_shadow.py:262-265), n f32 elements each.                      */
int advs_blend_mask_clamp01(const float* img, const float* adv, const float* cmask, float* out, long long n, void* stream);

/* Pillow-exact uint8 composites on [npix] RGB pixels with an RGBA layer and an L mask.
 * mode 0: Image.composite(Image.alpha_composite(img, layer), img, mask)      (add_shadow.py:57-58)
 * mode 1: add_shadow_to_mask_area + adjust_shadow_brightness(factor)  (shadow_for_attack.py:50-93) */
int advs_composite_u8(const uint8_t* img_hwc, const uint8_t* layer_rgba, const uint8_t* mask,
                      uint8_t* out_hwc, size_t npix, int mode, float factor, void* stream);
/* The same with the two roles of the mask apart (shadow_for_attack.py:76-93 when mask.size != image.size): paste_mask [npix]
 * gates the layer (L(layer) & paste_mask, or the composite's blend in mode 0), dark_mask [npix][dark_channels] (1 or 3) selects
 * the elements scaled by `factor` -- np.array(mask) after cv2.resize(INTER_NEAREST), per channel for a 3-channel mask.      */
int advs_composite_u8_masks(const uint8_t* img_hwc, const uint8_t* layer_rgba, const uint8_t* paste_mask,
                            const uint8_t* dark_mask, int dark_channels, uint8_t* out_hwc, size_t npix, int mode,
                            float factor, void* stream);
/* cv2.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) + boundingRect + contourArea for n masks [n][h][w] (nonzero =
 * foreground; add_shadow.py:40-47, shadow_for_attack.py:30-35), without border following: out[i][k][8] = {first pixel in
 * raster order, min x, min y, max x, max y, pixels of the hole-filled component, 2 * contourArea, 0} for the k-th external
 * contour found (any order; sort by the first pixel), count[i] = how many (may exceed max_components: then only that many rows
 * were written).  work: advs_mask_contours_work_bytes(n, h, w) bytes of scratch.                                              */
size_t advs_mask_contours_work_bytes(int n, int h, int w);
int advs_mask_contours(const uint8_t* mask, int n, int h, int w, void* work, int* out, int* count, int max_components,
                       void* stream);
/* One pass of Pillow's 8-bit ImagingResample (Image.resize behind transforms.Resize, ASR_fast.py:93,
 * PSNR_SSIM_fast.py:11): bounds[o] = {first input index, tap count}, coefs[o][ksize] int32 fixed
 * point (22 bits), computed by the host as Pillow does.  Images [n][h][w][channels] uint8.        */
int advs_resample_u8(const uint8_t* in, uint8_t* out, const int* bounds, const int* coefs, int ksize,
                     int n, int in_h, int in_w, int out_h, int out_w, int channels, int horizontal,
                     void* stream);
/* transforms.ToTensor (+ optional Normalize): uint8 [n][h][w][ch] -> f32 [n][ch][h][w] / 255.        */
int advs_u8hwc_to_f32nchw(const uint8_t* in, float* out, int n, int h, int w, int channels,
                          const float* mean, const float* stdv, void* stream);
/* uint8 [n][ch][h][w] -> [n][h][w][ch]: the permute of save_images (utils/utils.py:59-60).           */
int advs_u8_nchw_to_hwc(const uint8_t* in, uint8_t* out, int n, int channels, int h, int w, void* stream);
/* ---- the JPEG file hop, on the device ---------------------------------------------------
 * What saving an image as .jpg and reading it back does to its pixels (utils/utils.py:51-91 Image.save
 * with Pillow's defaults -> ASR_fast.py:90-92 / PSNR_SSIM_fast.py:21-24 Image.open): baseline JPEG, 4:2:0,
 * integer colour conversion, "islow" DCTs, quantisation at `quality` (Pillow's default 75), triangle chroma
 * upsampling.  Bit-exact with Pillow's round trip.  src/dst: uint8 [n][h][w][3]; h, w multiples of 16;
 * scratch: advs_jpeg_scratch_bytes(n, h, w) bytes.                                                    */
size_t advs_jpeg_scratch_bytes(int n, int h, int w);
int advs_jpeg_roundtrip_u8(const unsigned char* src, unsigned char* dst, void* scratch, int n, int h, int w,
                           int quality, void* stream);

/* calculate_ssim_psnr (PSNR_SSIM_fast.py:21-26, skimage semantics) for b image pairs, NCHW f32,
 * planes <= 64x64: out[b] = {ssim, psnr} as f64.                                                 */
int advs_psnr_ssim(const float* img1, const float* img2, double* out_ssim_psnr, int b, int c, int h, int w,
                   int win_size, void* stream);
/* torch.max(outputs, 1) indices (ASR_fast.py:115): first maximum of each row.                      */
int advs_argmax_rows(const float* x, int* out, int rows, int n, void* stream);

/* ---- victim classifier pieces that are not GEMMs ---------------------------------------------
 * (timm/torchvision ResNet-50: ASR_fast.py:16-20, ddim2/diff_model2.py:19-44; BatchNorm is folded
 * into weights+bias by the host, eval mode).  Stem: NCHW f32 image -> NHWC `dtype`,
 * y = act(conv_{k x k, stride, pad}(x) + bias), cin <= 4, w in torch OIHW f32.                    */
int advs_conv_stem(const float* x_nchw, const float* w_oihw, const float* bias, void* y,
                   int b, int cin, int h, int w, int cout, int ksize, int stride, int pad, int act,
                   int dtype, void* stream);
/* The stem as a GEMM: y[b][oy][ox][k] = x[b][c][oy*stride+r-pad][ox*stride+s-pad] (0 outside the image), k = (c*ksize + r)*ksize + s,
 * zero for k in [cin*ksize*ksize, kp).  The conv is then advs_conv2d 1x1 over kp channels with the OIHW weight viewed as
 * [cout][cin*ksize*ksize] and zero-padded to kp (ResNet-50's 7x7 stride-2 conv1, ASR_fast.py:16-20, on MFMA).           */
int advs_im2col_nchw(const float* x_nchw, void* y, int b, int cin, int h, int w, int ksize, int stride, int pad,
                     int kp, int dtype, void* stream);
int advs_maxpool3x3s2(const void* x, void* y, int b, int h, int w, int c, int dtype, void* stream); /* MaxPool2d(3,2,1) */
int advs_global_avgpool(const void* x, float* y, int b, int hw, int c, int dtype, void* stream);    /* -> f32 [b][c] */

/* ---- input gradient of the victim (gradient-based perturbation inside apply_shadow: tools/train_shadow.py:177-221
 * apply_adversarial_perturbation, 20 x { loss = cross_entropy(model(image + perturbation), label); loss.backward();
 * perturbation -= alpha * sign(image.grad * mask) }, and its integrated-gradient variant ddim2/test.py:647-681).
 * d loss / d image of the eval-mode ResNet-50 runs the network backwards: each conv's data gradient is advs_conv2d on
 * transposed / flipped weights (stride 2: advs_zero_insert2x first); the rest are the entry points below.  NHWC T
 * activations as in the forward, f32 NCHW at the image end.                                                        */
/* out[b][k] = scale * (softmax(logits[b])[k] - [k == labels[b]])  -- d cross_entropy / d logits (labels int64)     */
int advs_softmax_ce_grad(const float* logits, const long long* labels, float* out, int b, int k, float scale, void* stream);
/* out = y > 0 ? g + add : 0; y = forward activation after its ReLU; add may be NULL; out may alias g.  n elements.   */
int advs_relu_bwd(const void* g, const void* add, const void* y, void* out, long long n, int dtype, void* stream);
/* out[b][2i][2j] = in[b][i][j], zero elsewhere; out is [b][ho][wo][c] with ho in {2h-1, 2h} (likewise wo).           */
int advs_zero_insert2x(const void* in, void* out, int b, int h, int w, int c, int ho, int wo, int dtype, void* stream);
/* The stride-2 3x3 data gradient without zero insertion (even h, w): out[b][i][j] = [in[i][j] | in[i][j+1] | in[i+1][j] | in[i+1][j+1]]
 * (zeros beyond the border), then advs_conv2d 1x1 on the host-packed [4c'] -> [4c] parity matrix, then the interleave with the ReLU
 * mask: out[b][2i + (q>>1)][2j + (q&1)][:] = y > 0 ? x[b][i][j][q*c .. +c) : 0.                                                   */
int advs_gather2x2(const void* in, void* out, int b, int h, int w, int c, int dtype, void* stream);
int advs_depth_to_space2_relu(const void* x, const void* y, void* out, int b, int h, int w, int c, int dtype, void* stream);
/* out[b][p][c] = y[b][p][c] > 0 ? gp[b][c] / hw : 0  (AdaptiveAvgPool2d(1) backward + the last block's ReLU)         */
int advs_avgpool_bwd_relu(const float* gp, const void* y, void* out, int b, int hw, int c, int dtype, void* stream);
/* MaxPool2d(3,2,1) backward (first maximum of a window gets its gradient, as torch) times [x > 0]; x [b][h][w][c] is
 * the pool's input (the stem output after ReLU), g the gradient at the pooled resolution.                            */
int advs_maxpool3x3s2_bwd_relu(const void* g, const void* x, void* out, int b, int h, int w, int c, int dtype, void* stream);
/* MaxPool2d(2) backward (first maximum of each 2x2 window, as torch) times [x > 0]; x [b][h][w][c] is the pool's input
 * (a conv output after ReLU: VGG of ASR_fast.py:33-46), g the gradient at [b][h/2][w/2][c].                              */
int advs_maxpool2_bwd_relu(const void* g, const void* x, void* out, int b, int h, int w, int c, int dtype, void* stream);
/* data gradient of advs_conv_stem: g NHWC T [b][ho][wo][cout], w the same f32 OIHW weight -> dx NCHW f32 [b][cin][h][w] */
int advs_conv_stem_bwd(const void* g, const float* w_oihw, float* dx_nchw, int b, int cin, int h, int w, int cout,
                       int ksize, int stride, int pad, int dtype, void* stream);
/* Adjoint of advs_im2col_nchw: dx[b][c][iy][ix] = sum of gcol[b][oy][ox][(c*ksize + r)*ksize + s] over the taps that read
 * that pixel; gcol [b][ho][wo][kp] T is advs_conv2d 1x1 of the stem-output gradient with the transposed stem weight.       */
int advs_col2im_nchw(const void* gcol, float* dx_nchw, int b, int cin, int h, int w, int ksize, int stride, int pad,
                     int kp, int dtype, void* stream);
/* pert = clamp(pert - alpha * sign(sum_k grad[b][k] * mask), -eps, eps); x_in = x0 + pert (x_in may be NULL).
 * x0, pert, x_in NCHW f32 [b][c][hw]; grad [b][nsum][c][hw]; mask [b][mask_channels (1 | c)][hw].                   */
int advs_iga_step(const float* x0, const float* grad, const float* mask, float* pert, float* x_in,
                  int b, int c, int hw, int mask_channels, int nsum, float alpha, float eps, void* stream);
/* out = clamp(x0 + pert, 0, 1)   (train_shadow.py:219-220)                                                          */
int advs_perturb_clamp01(const float* x0, const float* pert, float* out, long long n, void* stream);
/* out[k] = base + (k / steps) * (x - base), k = 0..steps, stacked [steps+1][n]  (ddim2/test.py:659-660)            */
int advs_lerp_stack(const float* base, const float* x, float* out, int steps, long long n, void* stream);

/* DINOv2's classifier input (Dinov2ForImageClassification.forward; HF checkpoints of ASR_fast.py:47-58):
 * y[b] = [ tokens[b][0] | mean(tokens[b][1..np]) ], tokens [b][n_pad][c] -> y [b][2c] f32.                        */
int advs_cls_mean_rows_f32(const void* tokens, float* y, int b, int n_pad, int np, int c, int dtype, void* stream);

/* ---- ConvNeXt victim pieces (timm convnext_base.fb_in1k, ASR_fast.py:21-26) ----------------
 * Depthwise k x k conv (groups = c), stride 1|2, padding k/2, NHWC; w_taps_c is the [c][1][k][k] weight transposed to
 * [k*k][c] f32.  y is [b][ho][wo][c].                                                                          */
int advs_dwconv2d(const void* x, const float* w_taps_c, const float* bias, void* y, int b, int h, int w, int c,
                  int ksize, int stride, int dtype, void* stream);
/* ... followed by an activation (EfficientNetV2's depthwise Conv-BN-SiLU, BN folded by the host).                 */
int advs_dwconv2d_act(const void* x, const float* w_taps_c, const float* bias, void* y, int b, int h, int w, int c,
                      int ksize, int stride, int act, int dtype, void* stream);
/* Squeeze-and-excitation scale: y[b][p][c] = x[b][p][c] * s[b][c], s f32 [b][c] (torchvision SqueezeExcitation).    */
int advs_scale_channels(const void* x, const float* s, void* y, int b, int hw, int c, int dtype, void* stream);
/* [b][h][w][c] -> [b][h/2][w/2][4c], channel (dy*2+dx)*c + ch: a Conv2d(c, cout, 2, stride 2) becomes a 1x1 conv. */
int advs_space_to_depth2(const void* x, void* y, int b, int h, int w, int c, int dtype, void* stream);

/* HF ViTForImageClassification victim (ASR_fast.py:47-51) token plumbing: image -> patch rows whose K order is
 * the patch-embedding conv weight's ([hidden][cin*ps*ps]); tokens = [cls | patches] + position embeddings, rows
 * padded with zeros to n_pad; CLS rows gathered to f32 for the classifier head.                                */
int advs_patchify(const float* x_nchw, void* y, int b, int cin, int h, int w, int patch, int dtype, void* stream);
/* Same with rows of kpad >= cin*patch*patch elements, zero beyond the patch: a row length that is a whole number of
 * 128-byte slabs for any patch size (DINOv2's 14x14 patches are 588 elements = 1176 bytes in the 16-bit modes).   */
int advs_patchify_padded(const float* x_nchw, void* y, int b, int cin, int h, int w, int patch, int kpad, int dtype,
                         void* stream);
int advs_vit_assemble(const void* patches, const float* cls, const float* pos, void* tokens, int b, int np,
                      int n_pad, int c, int dtype, void* stream);
int advs_gather_rows_f32(const void* x, float* y, int b, long long row_stride, int c, int dtype, void* stream);

/* ---- handle-level entry points: a host without Python drives the eps-predictor and the DDIM loop -------------------------------
 * The op-level calls above are what the Python plan builder binds (engine.py / diff_model.py of the package); these restate that
 * plan in C++ (csrc/unet_handle.hip) for a caller that has only the C ABI: same kernels, same order, same bits.
 *   advs_unet_create      diff_model.UNetModel's constructor arguments (diff_model.py:163-175).
 *   advs_unet_param_*     the state_dict keys and element counts the network expects, in the reference's construction order.
 *   advs_unet_set_param   one state_dict tensor (host f32, torch layout).  The optional name "freqs" overrides the sinusoidal
 *                         embedding's frequency table (model_channels / 2 floats, diff_model.py:26-28) with the caller's bits.
 *   advs_unet_plan        packs the weights (once), builds the launch list of one forward for (batch, size) on `stream` and captures
 *                         it into a hipGraph.  uniform_t = 1: the sampler's plan, one timestep (t[0]) for the whole batch.
 *   advs_unet_forward     eps = model(x, t) (diff_model.py:245-267); x, eps NCHW f32 and t int64, all DEVICE pointers; stream-ordered.
 *   advs_ddim_tables      per-step (alpha_t, alpha_prev, sigma) and timesteps in loop order (diff_model.py:269-285, 304, 428-464);
 *                         coef_out / tseq_out NULL = size query (*nsteps_out only).  Host arrays.
 *   advs_ddim_run         the reverse loop (diff_model.py:442-474) with eta = 0: x holds x_T on entry, the sample on return.       */
typedef struct advs_unet advs_unet;
typedef struct advs_unet_config {
    int in_channels, model_channels, out_channels, num_res_blocks;
    int n_attention_resolutions, attention_resolutions[8];
    int n_channel_mult, channel_mult[8];
    int num_heads;
    int dtype;                          /* ADVS_F32 | ADVS_BF16 | ADVS_F16 */
} advs_unet_config;
int advs_unet_create(const advs_unet_config* cfg, advs_unet** out);
int advs_unet_param_count(const advs_unet* u);
int advs_unet_param_name(const advs_unet* u, int i, char* name, int name_len, long long* numel);
int advs_unet_set_param(advs_unet* u, const char* name, const float* host_data, long long numel);
int advs_unet_plan(advs_unet* u, int batch, int size, int uniform_t, void* stream);
int advs_unet_forward(advs_unet* u, const float* x_nchw, const int64_t* t, float* eps_nchw);
int advs_ddim_tables(int cosine_schedule, int timesteps, int ddim_timesteps, int quad, float eta, float* coef_out,
                     int64_t* tseq_out, int* nsteps_out);
int advs_ddim_run(advs_unet* u, float* x, const float* coef, const int64_t* tseq, int nsteps, int clip_denoised);
void advs_unet_destroy(advs_unet* u);

/* The victim side of the attack loop for the same kind of host (csrc/unet_handle.hip, second half): ResNet-50 in the timm / torchvision
 * layout (ASR_fast.py:16-20, ddim2/diff_model2.py:19-44; state_dict keys without BatchNorm's num_batches_tracked; BatchNorm is folded
 * into the convs in f32) and the evaluation chain in front of it.
 *   advs_resize_tables     Pillow's precompute_coeffs + normalize_coeffs_8bpc for BILINEAR (what transforms.Resize does to a PIL image,
 *                          ASR_fast.py:94, PSNR_SSIM_fast.py:10-13): the bounds / coefs advs_resample_u8 takes.  NULL tables = size query.
 *   advs_resnet50_plan     batch images of size x size; src_size > 0 also prepares advs_resnet50_eval_u8 for src_size x src_size uint8 input.
 *   advs_resnet50_forward  logits = model(x) (ASR_fast.py:113-115); device pointers, stream-ordered.
 *   advs_resnet50_eval_u8  asr.evaluate_batch: uint8 [b][3][src][src] sampler output -> HWC -> Resize((size, size)) -> ToTensor -> victim ->
 *                          argmax (ASR_fast.py:90-97, 113-117); pred int32 [b] on the device.                                        */
typedef struct advs_resnet50 advs_resnet50;
int advs_resize_tables(int in_size, int out_size, int* bounds, int* coefs, int* ksize_out);
int advs_resnet50_create(int num_classes, int dtype, advs_resnet50** out);
int advs_resnet50_param_count(const advs_resnet50* r);
int advs_resnet50_param_name(const advs_resnet50* r, int i, char* name, int name_len, long long* numel);
int advs_resnet50_set_param(advs_resnet50* r, const char* name, const float* host_data, long long numel);
int advs_resnet50_plan(advs_resnet50* r, int batch, int size, int src_size, void* stream);
int advs_resnet50_forward(advs_resnet50* r, const float* x_nchw, float* logits);
int advs_resnet50_eval_u8(advs_resnet50* r, const uint8_t* images_nchw, int* pred);
void advs_resnet50_destroy(advs_resnet50* r);

/* ---- stream capture (hipGraph) -------------------------------------------------------- */
int advs_graph_begin(void* stream);
int advs_graph_end(void* stream, void** graph_exec_out);
int advs_graph_launch(void* graph_exec, void* stream);
int advs_graph_destroy(void* graph_exec);

/* ---- timing helpers for bench.py (HIP events on the engine's own stream) ----------------- */
int advs_event_create(void** ev);
int advs_event_record(void* ev, void* stream);
int advs_event_elapsed_ms(void* start, void* stop, float* ms);   /* synchronises on stop */
int advs_event_destroy(void* ev);
int advs_stream_sync(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ADVSHADOW_H */
