/*
 * svg_hip.h — C ABI of libsvg_hip.so: the MI355X (gfx950) implementation of the
 * sd-video-gen sampling path.  Plain pointers and sizes only; no torch types.
 *
 * The reference has no FFI: this boundary replaces the third-party numerics its Python
 * calls into.  Each entry point names the reference call site it stands in for
 * (paths relative to the reference repo root):
 *
 *   svg_transformer_forward   models/transformer.py:47-68 (Transformer.forward: embedding*sqrt(d),
 *                             PositionalEncoding by batch index positional_encoding.py:33-35,
 *                             nn.Transformer, out Linear); called from prediction/predict.py:16-42
 *   svg_vae_encode            utils/sd_utils.py:128-145 (encode_img: /255, NHWC->NCHW, 2(x-.5),
 *                             vae.encode(...).sample(), *0.18215) incl. the uint8 nearest resize
 *                             of prediction/predict.py:158,178
 *   svg_vae_decode            utils/sd_utils.py:156-169 (decode_img_latents: /0.18215, vae.decode,
 *                             (x/2+.5).clamp(0,1), *255 round -> uint8 NHWC)
 *   svg_unet_forward          utils/sd_utils.py:253 (self.unet(latent_model_input, t,
 *                             encoder_hidden_states=...)['sample'])
 *   svg_ddim_loop             utils/sd_utils.py:222-267 (gen_i2i_latents: DDIMScheduler(0.00085,
 *                             0.012,'scaled_linear',1000), set_timesteps, add_noise, CFG combine,
 *                             scheduler.step) — the hot loop
 *   svg_clip_text_forward     utils/sd_utils.py:84,91 (self.text_encoder(input_ids)[0]: transformers CLIPTextModel of
 *                             'openai/clip-vit-large-patch14', last_hidden_state) — tokenisation stays on the host
 *   svg_resize_nearest_u8     prediction/predict.py:158,178 (F.interpolate on uint8, mode nearest)
 *   svg_resize_bilinear_f32   evaluation/predict_fvd.py:165 (F.interpolate of the predicted latent, mode bilinear)
 *   svg_load_weight/finalize  utils/sd_utils.py:52-66 + prediction/predict.py:50-51 (from_pretrained /
 *                             load_state_dict: tensors are handed over by their state_dict names)
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (SVG_ERR_INVALID: the arguments / shapes / call order are
 *     not acceptable — the Python facade raises ValueError; SVG_ERR_RUNTIME: a HIP call or kernel launch failed —
 *     RuntimeError); svg_last_error() gives the message.
 *   - all device pointers are caller-owned HBM (e.g. torch tensors); the library owns packed
 *     weights and one workspace arena per context, sized at svg_finalize()/first call; no
 *     allocation in steady state.  The library never frees caller memory.
 *   - `stream` is a hipStream_t (NULL = the null stream); every launch is asynchronous on it;
 *     no entry point synchronises the device except svg_load_weight/svg_finalize/svg_prof_*.
 *   - one context per (process, GPU); calls on a context are serialised by the caller.
 *   - boundary dtypes: f32 latents / embeddings / masks / noise, u8 images (NHWC).
 *     Storage type of the SD networks (activations + packed weights, f32 accumulation everywhere): bf16 by default,
 *     IEEE fp16 when the model is configured with f16=1 — the reference runs the UNet under fp16 autocast
 *     (utils/sd_utils.py:246); both sets of kernels are in the library (same sources compiled per type).
 *     The latent Transformer and the CLIP text tower compute in f32 (f32-input MFMA).
 */
#ifndef SVG_HIP_H
#define SVG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct svg_ctx svg_ctx;

enum svg_model { SVG_TRANSFORMER = 0, SVG_VAE = 1, SVG_UNET = 2, SVG_CLIP_TEXT = 3, SVG_MINILM = 4, SVG_I3D = 5 };
enum svg_status { SVG_OK = 0, SVG_ERR_RUNTIME = -1, SVG_ERR_INVALID = -2 };

/* ---- context ------------------------------------------------------------------------------ */
int svg_create(int device_id, svg_ctx** out);
void svg_destroy(svg_ctx* ctx);
const char* svg_last_error(svg_ctx* ctx);            /* ctx may be NULL (creation errors) */
const char* svg_version(void);
/* The library reads its $SVG_* tuning / debugging knobs once per name and caches them (no getenv on a launch path); a process
 * that changes such a variable after the first call (the tests do) calls this to have the next call look it up again. */
void svg_env_refresh(void);

/* ---- models: configure -> load every tensor by state_dict name -> finalize ------------------ */
/* `kv`: "key=v[,v...];key=v" e.g. "block_out=320,640,1280,1280;layers=2;heads=8;ctx_dim=768".
 * Transformer keys: d_lat, d_model, heads, enc_layers, dec_layers, ffn (default 2048), text_dim (0; 384 for the
 *   text-conditioned variant, whose first layer is named project_image_embedding instead of embedding).
 * VAE keys: block_out (128,256,512,512), layers (2), groups (32), latent (4), f16 (0; 1 = fp16 storage instead of bf16).
 * UNet keys: block_out (320,640,1280,1280), layers (2), heads (8), ctx_dim (768), groups (32),
 *            in_ch (4), out_ch (4), attn (1,1,1,0: cross-attention per down block), fp8 (0; 1 = BASELINE configs[4]: the
 *            dense projections with K % 128 == 0 that carry no folded LayerNorm / GEGLU run in MX block-scaled fp8 —
 *            OCP e4m3 + E8M0 per 32 — with activations quantised on the way in; everything else stays 16-bit),
 *            f16 (0; 1 = fp16 storage instead of bf16: utils/sd_utils.py:246 autocast).
 * CLIP text keys: vocab (49408), d_model (768), heads (12), layers (12), ffn (3072), max_pos (77); tensors by their
 *   transformers names without the "text_model." prefix (embeddings.token_embedding.weight, encoder.layers.N.*, ...).
 * MiniLM keys: vocab (30522), d_model (384), heads (12), layers (6), ffn (1536), max_pos (512); tensors by their transformers
 *   BertModel names (embeddings.word_embeddings.weight, encoder.layer.N.attention.self.query.weight, ...; the reference's text
 *   checkpoints carry them as sent_transformer.0.auto_model.<name>).
 * I3D keys: num_classes (400); tensors by the names of evaluation/pytorch_i3d.py's state_dict (Conv3d_1a_7x7.conv3d.weight,
 *   Mixed_3b.b1b.bn.running_var, logits.conv3d.bias, ...; num_batches_tracked entries are accepted and ignored). */
int svg_model_configure(svg_ctx* ctx, int model, const char* kv);
/* data: f32, host or device memory (hipMemcpyDefault); shape/ndim as in the state_dict. */
int svg_load_weight(svg_ctx* ctx, int model, const char* name, const float* data,
                    const int64_t* shape, int ndim);
/* packs fused layouts, checks that every expected tensor arrived (error names the first
 * missing key), returns the model's parameter count through *n_params if non-NULL. */
int svg_finalize(svg_ctx* ctx, int model, int64_t* n_params);
/* storage type of a configured model: "bf16" / "fp16" (SVG_VAE, SVG_UNET), "f32" (SVG_TRANSFORMER, SVG_CLIP_TEXT, SVG_MINILM, SVG_I3D); NULL if absent */
const char* svg_model_dtype(svg_ctx* ctx, int model);

/* ---- latent Transformer -------------------------------------------------------------------- */
/* src (B,Ts,D_lat), tgt (B,Tt,D_lat) batch-first f32; mask (Tt,Tt) additive f32 or NULL;
 * out (Tt,B,D_lat) sequence-first like the reference.  pe_row: NULL -> reference quirk (row b of
 * the batch gets PE(b)); else int32[B] giving the PE row used for each batch row (clip-batched
 * sampling passes zeros so every clip sees PE(0) exactly as at batch 1). */
int svg_transformer_forward(svg_ctx* ctx, const float* src, const float* tgt, int B, int Ts, int Tt,
                            const float* mask, const int32_t* pe_row, float* out, void* stream);

/* text-conditioned variant (models/transformer_text.py:71-111; configure text_dim=384): `text` (B,text_dim) f32 is the
 * class-name embedding (an INPUT: SentenceTransformer('all-MiniLM-L6-v2').encode(cls_list), transformer_text.py:82-83);
 * token = cat(project_image_embedding(x), text[b]) * sqrt(d_model) + PE, d_model = DIM_MODEL + text_dim. */
int svg_transformer_forward_text(svg_ctx* ctx, const float* src, const float* tgt, const float* text, int B, int Ts,
                                 int Tt, const float* mask, const int32_t* pe_row, float* out, void* stream);

/* The same forward with nn.Transformer's key-padding masks (models/transformer.py:64: src_key_padding_mask = src_pad_mask,
 * tgt_key_padding_mask = tgt_pad_mask): src_pad (B,Ts) / tgt_pad (B,Tt) are ADDITIVE f32 biases on the scores of every query and
 * head of batch row b (a bool mask's True is -inf, as torch canonicalises it), applied to the encoder / decoder SELF-attention keys;
 * the cross-attention gets none (the reference passes no memory_key_padding_mask).  Either may be NULL; text as above or NULL. */
int svg_transformer_forward_padded(svg_ctx* ctx, const float* src, const float* tgt, const float* text, int B, int Ts, int Tt,
                                   const float* mask, const float* src_pad, const float* tgt_pad, const int32_t* pe_row,
                                   float* out, void* stream);

/* ---- latent Transformer: training step ------------------------------------------------------- */
/* Replaces the body of trainers/trainer.py:111-190 (train_loop: forward in train mode, criterion, loss.backward(),
 * opt.step()) and :192-260 (validation_loop: the same loss in eval mode) for the latent Transformer; the Stable
 * Diffusion side is frozen there too (encode_batch = svg_vae_encode).  f32 throughout like the reference trainer.
 * criterion() of trainers/trainer.py:65-109 is  w_mse*MSE + w_l1*L1 + w_gdl*GDL(alpha) + w_contrastive*BiPatchNCE(temperature)
 * (models/contrastive_loss.py:7-60) on the last frames_to_predict positions; w_* = use_* x lambda_*. */
typedef struct svg_train_cfg {
  int frames_to_predict;          /* F: loss on pred[-F:] (trainer.py:145) */
  int feat_h, feat_w;             /* FRAME_SIZE / 8: latent (4, feat_h, feat_w) */
  float w_mse, w_l1, w_gdl, gdl_alpha, w_contrastive, temperature;
  float dropout_p;                /* nn.Transformer / PositionalEncoding dropout (train mode only) */
  uint64_t seed;                  /* dropout masks are a pure function of (seed, site, element): pass a new seed every step */
} svg_train_cfg;
/* src (B,Ts,D_lat), tgt (B,Tt,D_lat), expected (B,Tt,D_lat) batch-first f32 (trainer.py:124-131: new_batch, new_batch[:,:-1],
 * new_batch[:,1:]); text (B,text_dim) or NULL; mask (Tt,Tt) additive or NULL.  backward = 0: eval-mode forward + criterion only;
 * 1: train mode (dropout) and the gradient of every parameter (overwriting the previous step's: zero_grad + backward).
 * losses: host float[5] = {total, mse, l1, gdl, contrastive} (synchronises the stream), or NULL.  B <= 64, Ts, Tt <= 32. */
int svg_transformer_loss(svg_ctx* ctx, const svg_train_cfg* cfg, const float* src, const float* tgt, const float* expected,
                         const float* text, int B, int Ts, int Tt, const float* mask, int backward, float* losses, void* stream);
/* The train-mode forward on its own (models/transformer.py:47-68 with model.train(): dropout active, same sites and masks as
 * svg_transformer_loss with this seed); out (Tt,B,D_lat).  text: (B,text_dim) for the text-conditioned variant, else NULL. */
int svg_transformer_forward_train(svg_ctx* ctx, const float* src, const float* tgt, const float* text, int B, int Ts, int Tt,
                                  const float* mask, float dropout_p, uint64_t seed, float* out, void* stream);
/* torch.optim.Adam(lr, betas=(beta1, beta2), eps) step on the gradients of the last svg_transformer_loss(backward=1)
 * (trainer.py:365: optim.Adam(model.parameters(), lr=lr) -> betas (0.9, 0.999), eps 1e-8, no weight decay). */
int svg_transformer_adam_step(svg_ctx* ctx, float lr, float beta1, float beta2, float eps, void* stream);
/* copies a parameter / its gradient / its Adam moments out (host or device `out`, numel floats; state_dict key names):
 * what torch.save(model.state_dict()) at trainer.py:469-480 needs after steps taken in the library. */
enum svg_tensor_kind { SVG_TENSOR_PARAM = 0, SVG_TENSOR_GRAD = 1, SVG_TENSOR_EXP_AVG = 2, SVG_TENSOR_EXP_AVG_SQ = 3 };
int svg_transformer_tensor(svg_ctx* ctx, int kind, const char* name, float* out, int64_t numel, void* stream);

/* ---- CLIP text encoder ---------------------------------------------------------------------- */
/* input_ids (B,T) int32 token ids (T <= max_pos; the reference pads to 77); out (B,T,d_model) f32 = last_hidden_state.
 * Causal mask only (the reference passes no attention mask); f32 arithmetic like the reference. */
int svg_clip_text_forward(svg_ctx* ctx, const int32_t* input_ids, int B, int T, float* out, void* stream);

/* ---- MiniLM sentence encoder ------------------------------------------------------------------ */
/* models/transformer_text.py:82-83: txt = self.sent_transformer.encode(cls_list) with SentenceTransformer('all-MiniLM-L6-v2')
 * (:12): BertModel -> attention-mask-weighted mean pooling -> L2 normalisation.  input_ids (B,T) int32: [CLS] tokens [SEP],
 * then padding; lengths (B) int32: tokens of each row that are not padding; out (B,d_model) f32 unit-norm embeddings;
 * hidden (optional, may be NULL): (B,T,d_model) last_hidden_state.  T <= 128.  Tokenisation (WordPiece) stays on the host. */
int svg_minilm_encode(svg_ctx* ctx, const int32_t* input_ids, const int32_t* lengths, int B, int T, float* out, float* hidden,
                      void* stream);

/* ---- FVD evaluation (evaluation/pytorch_i3d.py, evaluation/fvd_2.py; called at the end of prediction/predict_text.py) ---------- */
/* InceptionI3d.forward (pytorch_i3d.py:303-312): x (B,3,T,224,224) f32 in [-1,1] -> logits (B,num_classes) f32 (mean over time). */
int svg_i3d_forward(svg_ctx* ctx, const float* x, int B, int T, int H, int W, float* logits, void* stream);
/* fvd_2.get_fvd_logits (fvd_2.py:16-19): videos (B,T,H,W,3) uint8 -> preprocess (fvd_2.py:7-14,109-136: /255, bilinear resize of the
 * shorter side to 224, centre crop, [-1,1]) -> I3D logits (B,num_classes). */
int svg_fvd_logits(svg_ctx* ctx, const uint8_t* videos, int B, int T, int H, int W, float* logits, void* stream);
/* the preprocessing alone: out (B,3,T,224,224) f32 */
int svg_fvd_preprocess(svg_ctx* ctx, const uint8_t* videos, int B, int T, int H, int W, float* out, void* stream);
/* fvd_2.frechet_distance (fvd_2.py:66-78): x1 (n1,d), x2 (n2,d) f32 device embeddings -> *out (HOST double; synchronises the stream).
 * Means / unbiased covariances in f64, the two symmetric square roots (fvd_2.py:22-33, SVD there) by a Jacobi eigen-decomposition. */
int svg_frechet_distance(svg_ctx* ctx, const float* x1, int n1, const float* x2, int n2, int d, double* out, void* stream);

/* ---- VAE ------------------------------------------------------------------------------------ */
/* img: u8 NHWC (N,srcH,srcW,3); nearest-resized to (H,W) on the fly when they differ.
 * eps: f32 (N,4,H/8,W/8) standard-normal draws for .sample(), or NULL for the distribution mean.
 * z_out: f32 (N,4,H/8,W/8), already multiplied by 0.18215.  moments_out (optional, may be NULL):
 * f32 (N,8,H/8,W/8) = [mean; logvar] before sampling. */
int svg_vae_encode(svg_ctx* ctx, const uint8_t* img, int N, int srcH, int srcW, int H, int W,
                   const float* eps, float* z_out, float* moments_out, void* stream);
/* z: f32 (N,4,h,w) scaled latents (divided by 0.18215 inside).  img_out: u8 NHWC (N,outH,outW,3),
 * nearest-resized from (8h,8w) when they differ.  float_out (optional): f32 NCHW (N,3,8h,8w),
 * the decoder output before the clamp/quantise. */
int svg_vae_decode(svg_ctx* ctx, const float* z, int N, int h, int w, uint8_t* img_out, int outH,
                   int outW, float* float_out, void* stream);

/* ---- UNet / DDIM ---------------------------------------------------------------------------- */
/* x (N,4,h,w) f32; timesteps f32[N]; ctx_emb (N,ctx_len,ctx_dim) f32; eps_out (N,4,h,w) f32. */
int svg_unet_forward(svg_ctx* ctx, const float* x, int N, int h, int w, const float* timesteps,
                     const float* ctx_emb, int ctx_len, float* eps_out, void* stream);
/* DDIM img2img over timesteps[start_step:] of a `num_steps` schedule (1000 train steps,
 * scaled_linear 0.00085..0.012, clip_sample, set_alpha_to_one, eta 0).
 * z (N,4,h,w) f32 in/out.  text_emb (2N,ctx_len,ctx_dim) = [uncond; cond] like encode_text().
 * noise: f32 (N,4,h,w) for add_noise when start_step>0 (required then), else ignored.
 * guidance==0 runs the UNet on the uncond half only (exact: u + 0*(c-u) == u).
 * hist (optional): f32 ((num_steps-start_step+1)*N,4,h,w) latent history incl. the start. */
int svg_ddim_loop(svg_ctx* ctx, float* z, int N, int h, int w, const float* text_emb, int ctx_len,
                  int num_steps, int start_step, float guidance, const float* noise, float* hist,
                  void* stream);
/* one scheduler step on caller data (x, eps -> prev); t = timestep value, t_prev = t - 1000/num_steps */
int svg_ddim_step(svg_ctx* ctx, const float* x, const float* eps, float* prev, int64_t n, int t,
                  int t_prev, void* stream);

int svg_resize_nearest_u8(svg_ctx* ctx, const uint8_t* src, int N, int sh, int sw, int C,
                          uint8_t* dst, int dh, int dw, void* stream);
/* evaluation/predict_fvd.py:165 (nn.functional.interpolate(latent, (64, 64), mode='bilinear'), align_corners=False):
 * `planes` f32 images of h x w -> oh x ow (NCHW latents: planes = N * 4). */
int svg_resize_bilinear_f32(svg_ctx* ctx, const float* src, int planes, int h, int w, float* dst, int oh, int ow, void* stream);

/* ---- operator level (the kernels the graphs are made of; used by the parity tests) ---------- */
/* 16-bit buffers are passed as uint16_t bit patterns: bf16 for svg_op_<name>, IEEE fp16 for the svg_op_<name>_f16 twin
 * declared at the end of this section (same arguments, the fp16 build of the same kernel).  NHWC activations, weights
 * [N][K] K-contiguous. */
/* C[M,N] = act(A[M,K] * W[N,K]^T + bias[N] + residual[M,N]);  out_f32: C is f32 instead of bf16.
 * act: 0 none, 1 SiLU, 2 GELU(erf), 3 GEGLU (W rows = [h;gate] halves of 2*N_out, C is [M,N/2]). */
int svg_op_gemm(svg_ctx* ctx, const uint16_t* A, const uint16_t* W, const float* bias,
                const uint16_t* residual, void* C, int M, int N, int K, int act, int out_f32,
                void* stream);
/* 3x3 convolution on NHWC bf16: x (B,H,W,Cin), w f32 OIHW (packed inside, cached by pointer is NOT
 * done: the packed copy is rebuilt per call — test hook).  mode: 0 stride1 pad1, 1 stride2 pad1,
 * 2 stride2 pad (0,1,0,1), 3 nearest-2x upsample then stride1 pad1.  out (B,Ho,Wo,Cout) bf16. */
int svg_op_conv3x3(svg_ctx* ctx, const uint16_t* x, const float* w_oihw, const float* bias,
                   uint16_t* out, int B, int H, int W, int Cin, int Cout, int mode, void* stream);
/* stride-1 3x3 conv whose tile epilogue leaves the GroupNorm column sums of its output, then the GroupNorm (+SiLU) that
 * consumes them (no statistics pass over the tensor).  *used_epilogue_stats: 1 when that path ran (images >= 32 x 32 without
 * split-K), 0 when the GroupNorm fell back to its own statistics pass.  Cin % 64 == 0, Cout % 4 == 0. */
int svg_op_conv3x3_gn(svg_ctx* ctx, const uint16_t* x, const float* w_oihw, const float* bias, const float* gamma,
                      const float* beta, uint16_t* conv_out, uint16_t* gn_out, int B, int H, int W, int Cin, int Cout,
                      int groups, float eps, int silu, int* used_epilogue_stats, void* stream);
/* stride-1 3x3 conv in OCP MX block-scaled fp8 (BASELINE configs[4], conv_halo_fp8.hip): x (B,H,W,Cin) 16-bit and the f32 OIHW weights
 * are quantised on the device (e4m3 + one E8M0 scale per 32 input channels), the products run on v_mfma_scale_f32_16x16x128_f8f6f4
 * with f32 accumulation; out (B,H,W,Cout) 16-bit = conv + bias (+ residual).  q_out ((B*H*W, Cp) bytes, Cp = Cin rounded up to 128)
 * and s_out ((B*H*W, Cp/32) bytes) optionally receive the quantised activations.  Cin % 64 == 0, Cout % 4 == 0 and >= 128,
 * H % 16 == W % 16 == 0 (of the output), at least 192 (16 x 16 pixel block, channel tile) pairs.  mode 0: stride 1 pad 1; mode 3: nearest-2x
 * upsample fused in front (out (B,2H,2W,Cout)), the UNet's Upsample2D.  Replaces F.conv2d of the resnets' convs under
 * the fp8 = 1 model key (reference call sites: the conv1 / conv2 of diffusers' ResnetBlock2D behind utils/sd_utils.py:253). */
int svg_op_conv3x3_mx(svg_ctx* ctx, const uint16_t* x, const float* w_oihw, const float* bias, const uint16_t* residual,
                      uint16_t* out, uint8_t* q_out, uint8_t* s_out, int B, int H, int W, int Cin, int Cout, int mode, void* stream);
/* C[M,N] = [A | A2] * W[N,K]^T + bias with the A operand given as two tensors (A: M x k_split, A2: M x (K - k_split)): the
 * torch.cat([hidden, skip], dim=1) in front of a resnet's 1x1 shortcut, never materialised.  k_split % 64 == 0. */
/* C[batch*M,N] = A W^T + bias + residual, and the LayerNorm statistics of its rows (rs = rstd, rm = rstd * mean, eps 1e-5) as the
 * transformer blocks of the UNet get them: from row partials the GEMM epilogue emits (*used = column tiles that emitted, 0 = fallback pass) */
int svg_op_gemm_lnstats(svg_ctx* ctx, const uint16_t* A, const uint16_t* W, const float* bias, const uint16_t* residual, uint16_t* C, int M,
                        int N, int K, int batch, float* rs, float* rm, int* used, void* stream);
int svg_op_gemm_cat(svg_ctx* ctx, const uint16_t* A, const uint16_t* A2, const uint16_t* W, const float* bias, uint16_t* C,
                    int M, int N, int K, int k_split, void* stream);
/* Fused GEGLU feed-forward of a BasicTransformerBlock (C = 320): out = ff.net.2(GEGLU(ff.net.0(LayerNorm(x)))) + residual in ONE
 * kernel (the M x 4C intermediate never reaches HBM).  x, residual, out: (M,C) bf16; w1 (8C,C) = [h; gate], b1 (8C), w2 (C,4C),
 * b2 (C), LayerNorm gamma / beta (C): f32 in the state_dict layout (folded and packed inside: test hook). */
int svg_op_ff_fused(svg_ctx* ctx, const uint16_t* x, const float* ln_gamma, const float* ln_beta, const float* w1,
                    const float* b1, const float* w2, const float* b2, const uint16_t* residual, uint16_t* out, int M,
                    int C, void* stream);
/* Cross-attention of a BasicTransformerBlock (C = 320: 8 heads of 40, context of L <= 80 tokens) in ONE kernel:
 * out = x + to_out(softmax(to_q(LayerNorm(x)) K^T / sqrt(40)) V) + bo.  x, out (M,320) 16-bit; k (N,L,320) and vt (N,320,Lp) 16-bit: the
 * projected context of each of the N = M / rows_per_sample samples; wq, wo (320,320), bo, LayerNorm gamma / beta (320): f32 in the
 * state_dict layout (folded, head-padded and packed inside: test hook).  Chained form: x == NULL and a (M,320) = the self-attention's
 * output, r (M,320) its residual, wp (320,320) / bp (320) its output projection: x = r + a wp^T + bp is formed inside the kernel. */
int svg_op_xattn_fused(svg_ctx* ctx, const uint16_t* x, const uint16_t* a, const uint16_t* r, const float* wp, const float* bp,
                       const float* ln_gamma, const float* ln_beta, const float* wq, const uint16_t* k, const uint16_t* vt, int Lp,
                       const float* wo, const float* bo, uint16_t* out, int M, int rows_per_sample, int L, void* stream);
/* dropout mask of one site of the training step: out[i] = 1/(1-p) (kept) or 0, i < n (tests regenerate the masks with it) */
int svg_op_dropout_mask(svg_ctx* ctx, uint64_t seed, int site, float p, float* out, int64_t n, void* stream);
/* MX block-scaled fp8 (BASELINE configs[4]): OCP e4m3 elements with one E8M0 scale per 32 consecutive K elements.
 * svg_op_quant_mx: x (rows,K) bf16 -> q (rows,K) e4m3 bytes, scales (rows,K/32) bytes; shared exponent floor(log2(amax)) - 8,
 * round to nearest even, saturating at +-448 (OCP MX v1.0).  K % 32 == 0. */
int svg_op_quant_mx(svg_ctx* ctx, const uint16_t* x, uint8_t* q, uint8_t* scales, int64_t rows, int K, void* stream);
/* C[M,N] = act(Q(A)[M,K] Q(W)[N,K]^T + bias + residual) on v_mfma_scale_f32_16x16x128_f8f6f4 (f32 accumulate), both operands
 * quantised on the fly by the routine above; act 0 none, 1 SiLU, 2 GELU.  K % 128 == 0, N % 4 == 0. */
int svg_op_gemm_fp8(svg_ctx* ctx, const uint16_t* A, const uint16_t* W, const float* bias, const uint16_t* residual, void* C,
                    int M, int N, int K, int act, int out_f32, void* stream);
/* GroupNorm (+SiLU) on NHWC bf16. */
int svg_op_groupnorm(svg_ctx* ctx, const uint16_t* x, const float* gamma, const float* beta,
                     uint16_t* out, int B, int HW, int C, int groups, float eps, int silu,
                     void* stream);
/* LayerNorm over the last dim of (M,C) bf16. */
int svg_op_layernorm(svg_ctx* ctx, const uint16_t* x, const float* gamma, const float* beta,
                     uint16_t* out, int M, int C, float eps, void* stream);
/* softmax(Q K^T * scale) V per (batch, head).  q (B,Sq,heads*d) bf16 row stride ldq; k (B,Skv,·)
 * row stride ldk; vt (B,heads*d,SkvPad) = V transposed, row stride ldvt (kv contiguous);
 * out (B,Sq,heads*d) row stride ldo.  Skv valid keys (columns >= Skv are masked). */
int svg_op_attention(svg_ctx* ctx, const uint16_t* q, const uint16_t* k, const uint16_t* vt,
                     uint16_t* out, int B, int heads, int Sq, int Skv, int d, int ldq, int ldk,
                     int ldvt, int ldo, int64_t q_bstride, int64_t k_bstride, int64_t vt_bstride,
                     int64_t o_bstride, float scale, void* stream);
/* fp16-storage twins of the hooks above (identical contracts; 16-bit buffers hold IEEE half) */
int svg_op_gemm_f16(svg_ctx* ctx, const uint16_t* A, const uint16_t* W, const float* bias, const uint16_t* residual, void* C, int M,
                    int N, int K, int act, int out_f32, void* stream);
int svg_op_conv3x3_f16(svg_ctx* ctx, const uint16_t* x, const float* w_oihw, const float* bias, uint16_t* out, int B, int H, int W,
                       int Cin, int Cout, int mode, void* stream);
int svg_op_conv3x3_gn_f16(svg_ctx* ctx, const uint16_t* x, const float* w_oihw, const float* bias, const float* gamma,
                          const float* beta, uint16_t* conv_out, uint16_t* gn_out, int B, int H, int W, int Cin, int Cout,
                          int groups, float eps, int silu, int* used_epilogue_stats, void* stream);
int svg_op_gemm_lnstats_f16(svg_ctx* ctx, const uint16_t* A, const uint16_t* W, const float* bias, const uint16_t* residual,
                            uint16_t* C, int M, int N, int K, int batch, float* rs, float* rm, int* used, void* stream);
int svg_op_gemm_cat_f16(svg_ctx* ctx, const uint16_t* A, const uint16_t* A2, const uint16_t* W, const float* bias, uint16_t* C,
                        int M, int N, int K, int k_split, void* stream);
int svg_op_ff_fused_f16(svg_ctx* ctx, const uint16_t* x, const float* ln_gamma, const float* ln_beta, const float* w1,
                        const float* b1, const float* w2, const float* b2, const uint16_t* residual, uint16_t* out, int M, int C,
                        void* stream);
int svg_op_xattn_fused_f16(svg_ctx* ctx, const uint16_t* x, const uint16_t* a, const uint16_t* r, const float* wp, const float* bp,
                           const float* ln_gamma, const float* ln_beta, const float* wq, const uint16_t* k, const uint16_t* vt, int Lp,
                           const float* wo, const float* bo, uint16_t* out, int M, int rows_per_sample, int L, void* stream);
int svg_op_conv3x3_mx_f16(svg_ctx* ctx, const uint16_t* x, const float* w_oihw, const float* bias, const uint16_t* residual,
                          uint16_t* out, uint8_t* q_out, uint8_t* s_out, int B, int H, int W, int Cin, int Cout, int mode, void* stream);
int svg_op_quant_mx_f16(svg_ctx* ctx, const uint16_t* x, uint8_t* q, uint8_t* scales, int64_t rows, int K, void* stream);
int svg_op_gemm_fp8_f16(svg_ctx* ctx, const uint16_t* A, const uint16_t* W, const float* bias, const uint16_t* residual, void* C,
                        int M, int N, int K, int act, int out_f32, void* stream);
int svg_op_groupnorm_f16(svg_ctx* ctx, const uint16_t* x, const float* gamma, const float* beta, uint16_t* out, int B, int HW,
                         int C, int groups, float eps, int silu, void* stream);
int svg_op_layernorm_f16(svg_ctx* ctx, const uint16_t* x, const float* gamma, const float* beta, uint16_t* out, int M, int C,
                         float eps, void* stream);
int svg_op_attention_f16(svg_ctx* ctx, const uint16_t* q, const uint16_t* k, const uint16_t* vt, uint16_t* out, int B, int heads,
                         int Sq, int Skv, int d, int ldq, int ldk, int ldvt, int ldo, int64_t q_bstride, int64_t k_bstride,
                         int64_t vt_bstride, int64_t o_bstride, float scale, void* stream);
/* f32 skinny GEMM of the latent Transformer: Y[M,N] = X[M,K] * W[N,K]^T + bias (relu_in: X:=max(X,0)). */
int svg_op_xf_gemm(svg_ctx* ctx, const float* X, const float* W, const float* bias, float* Y,
                   int M, int N, int K, int relu_in, void* stream);

/* ---- measurement ---------------------------------------------------------------------------- */
/* When enabled (on = 1), every launch of each kernel family is bracketed by hipEvents on its stream;
 * svg_prof_report waits for the context's own brackets (no device-wide sync) and writes "name calls total_ms flops bytes\n" lines into buf.
 * on = 2 additionally keeps one entry per call-site signature, reported as "@family|shape ..." lines. */
int svg_prof_enable(svg_ctx* ctx, int on);
int svg_prof_reset(svg_ctx* ctx);
int svg_prof_report(svg_ctx* ctx, char* buf, int buflen);
/* Has a layer-walking forward of the latent Transformer (models/transformer.py:47-68 in one launch, xf_walk.hip) on this device given up at a
 * device-wide barrier since the last Transformer call / status query?  0 = no; SVG_ERR_RUNTIME = yes: that forward's output is NaN-filled,
 * the walk is now off for the device (later forwards run the per-GEMM kernels) and the forward must be re-issued.  Call it where the
 * forward's result is consumed (after synchronising its stream).  The event is also raised by the next svg_transformer_* call; the VAE /
 * UNet / DDIM entry points log it and keep running. */
int svg_transformer_status(svg_ctx* ctx);
/* workspace bytes currently reserved by the context */
int64_t svg_workspace_bytes(svg_ctx* ctx);

/* ---- workspace ownership ("workspace arena sized at svg_create / first call; no allocation in steady state") ----
 * The reference has no counterpart: torch's caching allocator serves its intermediates (utils/sd_utils.py:247-261 allocates per
 * step).  Here every model call plans its intermediates into one arena per context.  A caller sizes it once:
 *   svg_reserve_workspace   at least `bytes` of workspace from now on;
 *   svg_plan_begin .. svg_plan_end   every model call in between runs its planning pass ONLY (nothing is launched, outputs are
 *                           not written) and records its need; svg_plan_end reserves the largest and returns it in *bytes
 *                           (the cross-attention K / V cache of svg_ddim_loop is sized too).
 * A call that still needs more grows the arena without freeing or synchronising (the outgrown block is released by the next
 * reserve / plan end / svg_destroy): safe while another thread of the process captures a stream.
 * svg_workspace_growths: (re)allocations since svg_create — constant once the workload has been planned.
 * Threads: one context per thread; svg_destroy, svg_model_configure, svg_finalize and svg_reserve_workspace synchronise the device
 * and therefore wait for any other thread's capture window inside the library (the DDIM loop's, the training step's) to close. */
int svg_reserve_workspace(svg_ctx* ctx, int64_t bytes);
int svg_plan_begin(svg_ctx* ctx);
int svg_plan_end(svg_ctx* ctx, int64_t* bytes);
int64_t svg_workspace_growths(svg_ctx* ctx);
/* capture windows open inside the library right now, over all contexts of the process (tests: drive a call into another
 * thread's window, $SVG_TEST_CAPTURE_HOLD_MS keeps svg_ddim_loop's open) */
int svg_debug_captures_active(void);
#ifdef __cplusplus
}
#endif
#endif /* SVG_HIP_H */
