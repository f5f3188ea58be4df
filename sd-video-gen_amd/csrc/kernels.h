// Host-callable launchers of every HIP kernel family (gfx950).
#pragma once
#include "common.h"

// Everything down to the latent-Transformer section belongs to the Stable-Diffusion side and lives in namespace SDNS
// (sd_bf16 or sd_f16, see common.h): these sources are compiled once per storage type.
namespace SDNS {

// ------------------------------------------------------------------------------------------------
// MFMA implicit GEMM:  C[M,N] = epi( A[M,K] * Wt[N,K]^T )
//   A is either a dense row-major bf16 matrix or a gathered 3x3-conv window of an NHWC tensor.
// ------------------------------------------------------------------------------------------------
enum AMode { A_DENSE = 0, A_CONV_S1 = 1, A_CONV_S2P1 = 2, A_CONV_S2ASYM = 3, A_CONV_UP2 = 4, A_CONV_SMALLC = 5 };
enum Act { ACT_NONE = 0, ACT_SILU = 1, ACT_GELU = 2, ACT_GEGLU = 3 };

struct GemmArgs {
  // A operand
  const h16* A = nullptr;
  int amode = A_DENSE;
  int lda = 0;                   // dense: row stride (elements)
  // dense, two-source A = channel concat [A | A2] (torch.cat([hidden, skip], dim=1) feeding a 1x1 shortcut): K columns
  // >= k_split (a multiple of 64) come from A2 (row stride lda2); igemm only
  const h16* A2 = nullptr;
  int lda2 = 0, k_split = 0;
  int H = 0, W = 0, Cin = 0;     // conv: input dims (NHWC)
  int Ho = 0, Wo = 0;            // conv: output dims
  // B operand (weights [N][K], K contiguous)
  const h16* Wt = nullptr;
  int ldb = 0;
  int n_valid = 0;               // rows of Wt that exist (<= N); rows beyond read as zero
  // C
  void* C = nullptr;
  int ldc = 0;
  int M = 0, N = 0, K = 0;       // N multiple of 4, K multiple of 8
  // batch (grid.y)
  int batch = 1;
  int64_t sA = 0, sB = 0, sC = 0;
  // epilogue
  float alpha = 1.f;
  const float* bias = nullptr;       // [N] (or [M] when bias_row)
  int bias_row = 0;
  const float* bias_bn = nullptr;    // [M/rows_per_batch][N] per-sample column bias (time embedding)
  int rows_per_batch = 1;
  int bias_bn_ld = 0;                // row stride of bias_bn (0 = N)
  const h16* residual = nullptr;    // [M][ldr]
  int ldr = 0;
  int act = ACT_NONE;
  int out_f32 = 0;
  // LayerNorm folded into the GEMM (W' = W * gamma, b' = b + W beta packed at load): the epilogue turns acc = x W'^T into
  // rstd[m] * acc - (rstd[m] * mean[m]) * s[n] with s[n] = sum_k W'[n][k]; applied before bias / residual / activation.
  // ln_swapped: the normalised operand is the B side (V^T = Wv * x^T): statistics per column (token z*ln_zstride + n), s per row.
  const float* ln_rs = nullptr;      // rstd per token
  const float* ln_rm = nullptr;      // rstd * mean per token
  const float* ln_s = nullptr;       // row sums of the packed bf16 weights
  int ln_swapped = 0;
  int64_t ln_zstride = 0;
  // split-K (0/1 = off). slabs: f32 [splitk][M][N] workspace
  int splitk = 1;
  float* slabs = nullptr;
  // GroupNorm statistics of the OUTPUT emitted by the tile epilogue (the consumer is a GroupNorm): per row tile and column
  // the sum and the sum of squares of the stored (bf16-rounded) values, gn_part[(tile_m * N + n) * 2 + {0,1}], written once per
  // tile in a fixed order (deterministic).  Row tile = 128 rows (igemm), 256 rows (gemm_pp) or a 16 x 16 pixel block (halo conv);
  // only without split-K.  gemm_emits_gn() says whether a problem qualifies and how many rows a tile has.
  float* gn_part = nullptr;
  // LayerNorm row sums of the stored (bf16-rounded) values, one partial per column tile: ln_part[(m * ln_tiles + tile_n) * 2 + {0,1}]
  // = sum, sum of squares of row m over the tile's columns (the consumer's ln_finish turns them into rstd, rstd * mean)
  float* ln_part = nullptr;
  int ln_tiles = 0;
  int64_t bias_zs = 0;               // batched problems: element stride of `bias` per batch (0 = shared)
  // fused q | k | V^T projection (gemm_ws only): output columns >= vt_n0 are the V projection and go, TRANSPOSED, to
  // vt_out[sample][n - vt_n0][token] (row m = sample * vt_rows + token; row stride vt_ld, sample stride vt_bs elements): the layout
  // the attention kernel reads V in.  ldc / C cover the columns below vt_n0 only.
  h16* vt_out = nullptr;
  int vt_n0 = 0, vt_rows = 0, vt_ld = 0;
  int64_t vt_bs = 0;
  int tn_major = 0;                  // tile order inside an XCD's run: 0 = tiles sharing the A rows adjacent, 1 = tiles sharing the weights adjacent
  int group_m = 0;                   // gemm_pp: row tiles per group of the grouped tile order (<= 1: row-major)
  int pp_merge = 0;                  // gemm_pp: one phase per slab (both k halves between two barriers) instead of two
  int pp_dma_m = 0;                  // merged ping-pong loops (gemm_pp, conv_halo): DMA instructions of a step issued among its MFMAs instead of in the load segment
  int dbg = 0;                       // ablation switch (SVG_GEMM_DBG): 1 no stores, 2 no MFMA, 3 no DMA, 5 LDS-staged epilogue
};

void launch_gemm(svg_ctx* ctx, const GemmArgs& g, hipStream_t s, int prof_kind);
// rows per row tile of the kernel gemm_auto() would launch for g, or 0 when that launch cannot emit GroupNorm statistics
// (split-K, batched, f32 / GEGLU output)
int gemm_emits_gn(const GemmArgs& g);
// per-device kernel attributes (dynamic LDS limits) of every instantiation; called by svg_create after hipSetDevice
void gemm_init_device();
void gemm_pp_init_device();
void gemm_ws_init_device();
void conv_halo_init_device();
void ff_fused_init_device();
void xattn_fused_init_device();
void gemm_fp8_init_device();

// ---- MX block-scaled fp8 (OCP e4m3 elements, one E8M0 scale per 32 K elements): quantiser and GEMM (gemm_fp8.hip)
void quant_mx_h16(svg_ctx* ctx, const h16* x, int ldx, uint8_t* q, uint8_t* sc, int64_t rows, int K, hipStream_t s);
void quant_mx_f32(const float* x, int ldx, uint8_t* q, uint8_t* sc, int64_t rows, int K, hipStream_t s);
bool gemm_fp8_supported(int M, int N, int K);
void gemm_fp8(svg_ctx* ctx, const uint8_t* A, const uint8_t* As, const uint8_t* W, const uint8_t* Ws, const GemmArgs& g, hipStream_t s);

// ---- MX fp8 3x3 convolution (conv_halo_fp8.hip): activations e4m3 [P][Cp] + E8M0 [P][Cp/32] (Cp = C rounded up to 128, padding
// zero), weights e4m3 [Npad][9][Cp] + E8M0 [9][Cp/128][Npad][4]
bool conv_halo_fp8_supported(int B, int H, int W, int Cin, int N);
void conv_halo_fp8_init_device();
void quant_act_mx(svg_ctx* ctx, const h16* x, int C, uint8_t* q, uint8_t* sc, int64_t P, hipStream_t s);
void pack_conv3x3_mx(const float* w_oihw, uint8_t* q, uint8_t* sc, int O, int I, int Npad, int Cp, hipStream_t s);
void conv_halo_fp8(svg_ctx* ctx, const uint8_t* A8, const uint8_t* As, const uint8_t* W8, const uint8_t* Ws, int Npad, const GemmArgs& g, hipStream_t s);

// fused GEGLU feed-forward (C = 320): out = (GEGLU(LN(x) W1^T + b1)) W2^T + b2 + residual; the M x 4C intermediate stays on chip
bool ff_fused_supported(int C, int M);
// cross-attention of a C = 320 block in one launch (xattn_fused.hip): to_q + attention over <= 80 context keys + to_out + residual
bool xattn_fused_supported(int C, int heads, int M, int rows_per_sample, int L);
int64_t xattn_kv_pack_elems(int N);
void xattn_pack_q(const float* w, const float* gamma, const float* beta, h16* Wq, float* sq, float* bq, int perm, hipStream_t s);
bool xattn_chain_enabled();
void xattn_pack_o(const float* w, h16* Wo, hipStream_t s);
void xattn_pack_kv(const h16* K, int ldk, int64_t k_bs, const h16* Vt, int ldv, int64_t v_bs, h16* Kp, h16* Vp, int N, int L, hipStream_t s);
void xattn_fused(svg_ctx* ctx, const h16* X, int ldx, const h16* R, int ldr, const h16* Wp, const float* bp, const float* rs, const float* rm,
                 const h16* Wq, const float* sq, const float* bq, const h16* Kp, const h16* Vp, const h16* Wo, const float* bo, h16* out, int ldo,
                 int M, int rows_per_sample, int L, hipStream_t s);
void pack_ff2_perm(const float* w, h16* out, int N, int K, hipStream_t s);
void ff_fused(svg_ctx* ctx, const h16* X, int ldx, const h16* W1, const float* b1, const float* s1, const float* rs, const float* rm,
              const h16* W2p, const float* b2, const h16* residual, int ldr, h16* out, int ldo, int M, hipStream_t s);
// picks split-K from the shape, allocates slabs from the arena, launches
void gemm_auto(svg_ctx* ctx, GemmArgs g, hipStream_t s, int prof_kind);
// a GemmArgs with vt_out set can only be served by the weight-stationary kernel: ask before launching
bool gemm_fused_qkv_supported(const GemmArgs& g);

// weight packing (device): f32 OIHW -> bf16 [Opad][ky][kx][Ipad]; f32 [N][K] -> bf16 [Npad][K]
void pack_conv3x3(const float* w_oihw, h16* out, int O, int I, int Opad, int Ipad, hipStream_t s);
void pack_linear(const float* w, h16* out, int N, int K, int Npad, hipStream_t s);
// LayerNorm folding (load time): bias_out[n] = bias_in[n] + sum_k W[n][k] beta[k]; then W[n][k] *= gamma[k] in place
void fold_ln_weights(float* w, const float* bias_in, const float* gamma, const float* beta, float* bias_out, int N, int K, hipStream_t s);
void rowsum_h16(const h16* w, float* out, int N, int K, hipStream_t s);
// per-row LayerNorm statistics of x[M][C]: rs = rstd, rm = rstd * mean
void ln_stats(svg_ctx* ctx, const h16* x, float* rs, float* rm, int M, int C, float eps, hipStream_t s);
// GEGLU: rows [h(0..F-1); gate(0..F-1)] -> 16-row tiles alternating h / gate; bias likewise
void pack_geglu(const float* w, const float* b, h16* wout, float* bout, int F, int K, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// normalisation / softmax
// ------------------------------------------------------------------------------------------------
// column tiles a dense GEMM launch of g will use when it can emit LayerNorm row partials (0: it cannot — split-K, f32 / GEGLU output ...)
int gemm_ln_tiles(const GemmArgs& g);
// rs[m] = rstd, rm[m] = rstd * mean of row m from `tiles` partials per row (GemmArgs::ln_part)
void ln_finish(svg_ctx* ctx, const float* part, int tiles, float* rs, float* rm, int M, int C, float eps, hipStream_t s);
// per-row-tile column sums of a tensor, emitted by the epilogue that produced it (GemmArgs::gn_part)
struct GnStats {
  const float* part = nullptr;   // [B * tiles_per_sample][C][2]
  int tiles_per_sample = 0;
  bool valid() const { return part != nullptr && tiles_per_sample > 0; }
};
// x (B,HW,C) bf16 NHWC; optional second source for channel concat [x | x2] (C = C1 + C2).  With the producers' column sums
// (st1 for x, st2 for x2) the statistics pass over the tensor is skipped: a small kernel finishes (mean, rstd) per (sample, group).
void groupnorm(svg_ctx* ctx, const h16* x, int C1, const h16* x2, int C2, const float* gamma,
               const float* beta, h16* out, int B, int HW, int groups, float eps, int silu,
               hipStream_t s, const GnStats* st1 = nullptr, const GnStats* st2 = nullptr);
// the same with an MX fp8 output (norm.hip: gn_apply_mx_kernel) for conv_halo_fp8; false = the caller takes groupnorm() + quant_act_mx()
bool groupnorm_mx(svg_ctx* ctx, const h16* x, int C1, const h16* x2, int C2, const float* gamma, const float* beta, uint8_t* q, uint8_t* sc,
                  int B, int HW, int groups, float eps, int silu, hipStream_t s, const GnStats* st1, const GnStats* st2);
// (mean, rstd) per (sample, group) from producer column sums -> stats[B][groups][2]
void gn_finish(svg_ctx* ctx, const GnStats& st1, int C1, const GnStats* st2, int C2, float* stats, int B, int HW, int groups, float eps,
               hipStream_t s);
// GroupNorm folded into a following 1x1 projection (no activation in between): per-sample weights
// Wb[b][n][c] = bf16(W[n][c] gamma[c] rstd[b][g(c)]) and bias bb[b][n] = bias[n] + sum_c W[n][c] (beta[c] - mean rstd gamma[c])
void gn_fold_weights(const float* W, const float* bias, const float* gamma, const float* beta, const float* stats, h16* Wb,
                     float* bb, int B, int N, int C, int groups, hipStream_t s);
void layernorm(svg_ctx* ctx, const h16* x, const float* gamma, const float* beta, h16* out, int M,
               int C, float eps, hipStream_t s);
// rows of f32 scores -> bf16 probabilities; cols valid < n_valid, row stride ld (elements)
void softmax_rows(svg_ctx* ctx, const float* s_in, h16* p_out, int64_t rows, int cols, int ld_in,
                  int ld_out, float scale, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// fused attention
// ------------------------------------------------------------------------------------------------
struct AttnArgs {
  const h16 *q, *k, *vt;
  h16* out;
  int B, heads, Sq, Skv, d;
  int ldq, ldk, ldvt, ldo;
  int64_t qb, kb, vtb, ob;       // batch strides (elements)
  float scale;
};
void attention(svg_ctx* ctx, const AttnArgs& a, hipStream_t s);
// single-head d = 512 attention of the VAE mid blocks, fused (attn_vae.hip): q, k rows of stride ldqk, vt = V transposed (B, C, ldvt)
bool vae_attention_supported(int S, int C, int ldqk, int ldvt, int ldo);
void vae_attention(svg_ctx* ctx, const h16* q, const h16* k, int ldqk, int64_t qkb, const h16* vt, int ldvt, int64_t vtb, h16* out, int ldo,
                   int64_t ob, int B, int S, int C, hipStream_t s);
void vae_attn_init_device();

// ------------------------------------------------------------------------------------------------
// element-wise / layout
// ------------------------------------------------------------------------------------------------
// u8 NHWC (N,sh,sw,3) -> bf16 NHWC (N,H,W,8) with nearest resize, x/255*2-1, channels 3..7 zero
void img_to_act(const uint8_t* img, h16* out, int N, int sh, int sw, int H, int W, hipStream_t s);
// f32 NHWC (N,h,w,ldc) first 3 channels -> optional f32 NCHW + u8 NHWC (N,oh,ow,3) nearest resized:
// (x/2+.5).clamp(0,1)*255 round-half-even
void act_to_img(const float* x, int ldc, uint8_t* img, float* fout, int N, int h, int w, int oh, int ow,
                hipStream_t s);
// f32 NCHW (N,C,h,w) * scale -> bf16 NHWC (N,h,w,Cpad)
void nchw_to_act(const float* x, h16* out, int N, int C, int h, int w, int Cpad, float scale, hipStream_t s);
// f32 NCHW -> f32 NHWC (scaled); f32 NHWC (P,C) -> bf16 (P,Cpad) zero padded
void nchw_to_actf32(const float* x, float* out, int N, int C, int h, int w, float scale, hipStream_t s);
void actf32_pad_h16(const float* x, int C, h16* out, int Cpad, int64_t P, hipStream_t s);
// f32 NHWC-ish source (N,h,w,ld) f32 -> f32 NCHW (N,C,h,w)
void actf32_to_nchw(const float* x, int ld, float* out, int N, int C, int h, int w, hipStream_t s);
// per-pixel 1x1 conv with tiny channel counts, f32: y[p][o] = sum_i w[o][i] x[p][i] + b[o]
void pixel_linear_f32(const float* x, int ldx, const float* w, const float* b, float* y, int ldy,
                      int64_t P, int Cin, int Cout, hipStream_t s);
// VAE posterior sample: moments f32 (P,8) [mean(4); logvar(4)] -> z NCHW f32 = (mean + exp(.5*clamp(lv))*eps)*0.18215
void vae_sample(const float* mom, int ldm, const float* eps_nchw, float* z_nchw, float* mom_nchw, int N,
                int h, int w, hipStream_t s);
void concat_channels(const h16* a, int Ca, const h16* b, int Cb, h16* out, int64_t P, hipStream_t s);
void resize_bilinear_f32(const float* src, float* dst, int P, int h, int w, int oh, int ow, hipStream_t s);
void resize_nearest_u8(const uint8_t* src, uint8_t* dst, int N, int sh, int sw, int C, int dh, int dw,
                       hipStream_t s);
void f32_to_h16(const float* x, h16* y, int64_t n, hipStream_t s);
void h16_to_f32(const h16* x, float* y, int64_t n, hipStream_t s);
// timestep sinusoid (flip_sin_to_cos, shift 0): t f32[N] -> bf16 (N,dim) = [cos | sin]
void timestep_embed(const float* t, h16* out, int N, int dim, hipStream_t s);
void silu_h16(const h16* x, h16* y, int64_t n, hipStream_t s);
// DDIM: z <- step(z, eps) with clip_sample; coefficients from the device table `coef` row `*step_idx`
void ddim_step(const float* z, const float* eps_u, const float* eps_c, float guidance, float* z_out,
               int64_t n, float sqrt_at, float sqrt_1mat, float sqrt_ap, float sqrt_1map, hipStream_t s);
// the step with (t, coefficients) read from row idx[0] of a device table of 5-float rows; tvec[i] = tab[idx][0]; idx[0] += 1
void ddim_step_tab(const float* z, const float* eps_u, const float* eps_c, float guidance, float* z_out, int64_t n, const float* tab,
                   const int* idx, hipStream_t s);
void ddim_tvec(float* tvec, int nb, const float* tab, const int* idx, hipStream_t s);
void ddim_bump(int* idx, hipStream_t s);
void add_noise(const float* x0, const float* noise, float* out, int64_t n, float sa, float s1a, hipStream_t s);
void fill_f32(float* p, int64_t n, float v, hipStream_t s);

}  // namespace SDNS

void xf_train_init_device();
void xformer_init_device();

// ------------------------------------------------------------------------------------------------
// latent Transformer (f32, f32-input MFMA)
// ------------------------------------------------------------------------------------------------
// act_in on X: 0 none, 1 ReLU, 2 quick-GELU, 3 exact GELU (erf); Y = act_in(X) W^T + bias (+ residual[M][N]); M <= 336
void xf_gemm(svg_ctx* ctx, const float* X, const float* W, const float* bias, float* Y, int M, int N,
             int K, int act_in, hipStream_t s, const float* residual = nullptr);
// CLIP text tower pieces (f32): token + position embedding lookup; causal self-attention on packed [q|k|v] rows
void xf_embed_tokens(const int32_t* ids, const float* tok, const float* pos, float* y, int rows, int T, int d, int vocab, hipStream_t s);
void xf_attention_causal(const float* qkv, float* o, int B, int T, int heads, int hd, hipStream_t s);
// BERT (MiniLM sentence encoder): bidirectional attention over the first lens[b] keys of packed [q|k|v] rows; word + position +
// token-type-0 embedding lookup; masked mean pooling + L2 normalisation (sentence-transformers Pooling + Normalize)
void xf_attention_padded(const float* qkv, const int32_t* lens, float* o, int B, int T, int heads, int hd, hipStream_t s);
void xf_embed_bert(const int32_t* ids, const float* word, const float* pos, const float* type0, float* y, int rows, int T, int d, int vocab,
                   hipStream_t s);
void xf_mean_pool_norm(const float* x, const int32_t* lens, float* out, int B, int T, int d, hipStream_t s);
// y = LayerNorm(x + r) rows of d
void xf_add_ln(const float* x, const float* r, const float* g, const float* b, float* y, int M, int d,
               float eps, hipStream_t s);
// emb (B*T,d) -> (T,B,d): y[t][b] = emb[b][t]*sqrt(d) + pe[pe_row[b]]
// text (B,d_txt) or null: channels >= d - d_txt of every token come from text[b] instead of emb (emb rows are d - d_txt wide)
void xf_embed_post(const float* emb, const float* pe, const int32_t* pe_row, const float* text, int d_txt, float* y, int B, int T,
                   int d, float scale, hipStream_t s);
// ---- training step of the latent Transformer (xf_train.hip) --------------------------------------------------------------
// dropout site: mask element i of site `site` is a pure function of (seed, site, i); p = 0 disables it.  The seed is read from
// device memory, so a captured graph of the step replays with a new seed without new kernel arguments.
struct XfDrop { const uint64_t* seed; uint32_t site; float p; };
struct XfAdamTensor { float* p; const float* g; float* m; float* v; };
struct XfAdamChunk { int32_t ten; int32_t n; int64_t off; };
void xf_drop_mask(const XfDrop d, float* out, int64_t n, hipStream_t s);
// dW[N][K] (+)= dY[M][N]^T X[M][K];  db[N] (+)= column sums of dY (db may be null)
void xf_gemm_tn(const float* dY, int ldy, const float* X, int ldx, float* dW, float* db, int M, int N, int K, int accumulate, hipStream_t s);
// out[M][K] = gate(dY[M][N] W[N][K]) + add;  gate (or null): x * (gate[i] > 0 ? gate_scale : 0);  slabs: xf_gemm_nn_slab_floats() floats
int64_t xf_gemm_nn_slab_floats(int M, int N, int K);
void xf_gemm_nn(const float* dY, int ldy, const float* W, float* slabs, float* out, int M, int N, int K, const float* gate, float gate_scale,
                const float* add, hipStream_t s);
void xf_relu_drop(const float* h, float* r, int64_t n, const XfDrop d, hipStream_t s);
void xf_add_ln_train(const float* x, const float* r, const XfDrop dr, const float* g, const float* b, float* y, float* xhat, float* rstd, int M,
                     int d, float eps, hipStream_t s);
void xf_ln_bwd(const float* dy, const float* xhat, const float* rstd, const float* g, float* dz, float* dz_drop, const XfDrop dr, float* dgamma,
               float* dbeta, int M, int d, hipStream_t s);
void xf_embed_post_train(const float* emb, const float* pe, const int32_t* pe_row, const float* text, int d_txt, float* y, int B, int T, int d,
                         float scale, const XfDrop dr, hipStream_t s);
void xf_embed_post_bwd(const float* dy, float* de, int B, int T, int d, int d_img, float scale, const XfDrop dr, hipStream_t s);
void xf_attention_train(const float* q, int ldq, const float* k, const float* v, int ldk, const float* mask, float* o, float* P, int Tq, int Tk,
                        int B, int heads, int hd, const XfDrop dr, hipStream_t s);
void xf_attention_bwd(const float* dout, const float* q, int ldq, const float* k, const float* v, int ldk, const float* P, float* dq, int lddq,
                      float* dk, float* dv, int lddk, int Tq, int Tk, int B, int heads, int hd, const XfDrop dr, hipStream_t s);
// criterion of trainers/trainer.py:65-109 on pred (Tt,B,D) rows t >= t0 vs expected (B,Tt,D): losses[5] = {total, mse, l1, gdl, nce}, dpred
void xf_criterion(const float* pred, const float* expected, float* dpred, float* part, float* part2, float* losses, int Tt, int B, int D, int t0,
                  int fh, int fw, float w_mse, float w_l1, float w_gdl, float alpha, float w_nce, float temperature, hipStream_t s);
void xf_adam(const XfAdamTensor* tens, const XfAdamChunk* chunks, int n_chunks, float lr, float beta1, float beta2, float eps, int step,
             hipStream_t s);
// seq-first MHA core on packed projections: q (Tq,B,ldq) k,v (Tk,B,ldk) -> o (Tq,B,d); mask (Tq,Tk) or null;
// kpad (B,Tk) or null: additive key-padding bias per batch row
void xf_attention(const float* q, int ldq, const float* k, const float* v, int ldk, const float* mask,
                  float* o, int Tq, int Tk, int B, int heads, int hd, hipStream_t s, const float* kpad = nullptr);
