// Model graphs executed by the library: latent Transformer, SD VAE, SD UNet, DDIM loop.
#pragma once
#include "kernels.h"
#include <algorithm>
#include <initializer_list>
#include <memory>

struct WeightStore {
  std::unordered_map<std::string, Weight> map;
  void put(svg_ctx* ctx, const std::string& name, const float* data, const int64_t* shape, int ndim);
  const Weight& get(const std::string& name) const;
  const Weight& get(const std::string& name, std::initializer_list<int64_t> shape) const;
  bool has(const std::string& name) const { return map.count(name) != 0; }
  void release(const std::string& name);   // frees the f32 copy (after packing)
  void clear();
  int64_t total_params() const;
};

std::unordered_map<std::string, std::vector<int64_t>> parse_kv(const char* kv);

// Plans a call twice: a dry pass measures the arena high-water mark, then the real pass launches (in plan mode: the dry pass only).
template <typename F>
inline void run_planned(svg_ctx* ctx, F&& body) {
  ctx->arena.reset();
  ctx->arena.dry = true;
  ctx->arena.high = 0;
  try { body(); } catch (...) { ctx->arena.dry = false; throw; }
  ctx->arena.dry = false;
  if (ctx->plan_only) {   // svg_plan_begin .. svg_plan_end: only the need is recorded
    ctx->plan_high = std::max(ctx->plan_high, ctx->arena.high);
    ctx->arena.reset();
    return;
  }
  ctx->ensure_arena(ctx->arena.high);
  ctx->arena.reset();
  body();
}

// ---- latent Transformer ------------------------------------------------------------------------------
struct XfModel {
  WeightStore ws;
  int d_lat = 0, d_model = 0, heads = 8, enc_layers = 0, dec_layers = 0, ffn = 2048;
  int text_dim = 0;   // > 0: text-conditioned variant (models/transformer_text.py): d_model = DIM_MODEL + text_dim
  bool ready = false;
  float* pe = nullptr;   // (64, d_model)
  int pe_d = 0;
  int32_t* iota = nullptr;
  struct XfTrain* train = nullptr;   // gradients + Adam moments (xf_trainer.cpp); dropped whenever weights are (re)loaded
  // weight pointers by role, resolved once in finalize() (the layer-walking forward builds its stage table from them)
  struct LayerW {
    const float *in_w, *in_b, *out_w, *out_b;          // self-attention
    const float *cin_w, *cin_b, *cout_w, *cout_b;      // decoder: attention over the encoder memory
    const float *l1_w, *l1_b, *l2_w, *l2_b;
    const float *n_w[3], *n_b[3];
  };
  std::vector<LayerW> enc_w, dec_w;
  const float *emb_w = nullptr, *emb_b = nullptr, *out_w = nullptr, *out_b = nullptr, *encn_w = nullptr, *encn_b = nullptr, *decn_w = nullptr,
              *decn_b = nullptr;
  void configure(const char* kv);
  void finalize(svg_ctx* ctx, int64_t* n_params);
  // src_pad (B,Ts) / tgt_pad (B,Tt): additive key-padding biases (models/transformer.py:64) or null
  void forward(svg_ctx* ctx, const float* src, const float* tgt, int B, int Ts, int Tt, const float* mask,
               const int32_t* pe_row, float* out, hipStream_t s, const float* text = nullptr, const float* src_pad = nullptr,
               const float* tgt_pad = nullptr);
};

// ---- CLIP text tower (transformers CLIPTextModel; reference call site utils/sd_utils.py:60,84,91) ---------------------
// f32 like the reference (the text encoder runs outside autocast), on the latent Transformer's weight-streaming kernels.
struct ClipTextModel {
  WeightStore ws;
  int vocab = 49408, d_model = 768, heads = 12, layers = 12, ffn = 3072, max_pos = 77;
  bool ready = false;
  std::vector<float*> qkv_w, qkv_b;        // per layer: [Wq; Wk; Wv] stacked (3d x d) so the three projections are one stream
  void configure(const char* kv);
  void finalize(svg_ctx* ctx, int64_t* n_params);
  // ids (B,T) int32 -> out (B,T,d_model) f32 = last_hidden_state (after final_layer_norm)
  void forward(svg_ctx* ctx, const int32_t* ids, int B, int T, float* out, hipStream_t s);
};

// ---- MiniLM sentence encoder (SentenceTransformer('all-MiniLM-L6-v2').encode; reference call site models/transformer_text.py:12,82-83)
// BertModel (6 layers, d 384, 12 heads, ffn 1536) + masked mean pooling + L2 normalisation, f32 on the weight-streaming kernels.
struct MiniLmModel {
  WeightStore ws;
  int vocab = 30522, d_model = 384, heads = 12, layers = 6, ffn = 1536, max_pos = 512;
  bool ready = false;
  std::vector<float*> qkv_w, qkv_b;        // per layer: [Wq; Wk; Wv] stacked
  void configure(const char* kv);
  void finalize(svg_ctx* ctx, int64_t* n_params);
  // ids (B,T) int32 ([CLS] ... [SEP] then padding), lens (B) valid tokens per row -> out (B,d_model) unit-norm sentence embeddings;
  // hidden (optional): (B,T,d_model) last_hidden_state
  void encode(svg_ctx* ctx, const int32_t* ids, const int32_t* lens, int B, int T, float* out, float* hidden, hipStream_t s);
};

// ---- I3D (FVD evaluation: evaluation/pytorch_i3d.py, evaluation/fvd_2.py; f32, channels-last, BatchNorm folded at finalize) ----
struct I3dModel {
  WeightStore ws;
  int num_classes = 400;
  bool ready = false;
  struct Unit { float* w = nullptr; float* b = nullptr; int cin = 0, cin_pad = 0, cout = 0, k = 1; };
  struct Mixed { Unit b0, b1a, b1b, b2a, b2b, b3b; };
  Unit conv1a, conv2b, conv2c, logits;
  std::vector<Mixed> mixed;
  void configure(const char* kv);
  void finalize(svg_ctx* ctx, int64_t* n_params);
  Unit load_unit(svg_ctx* ctx, const std::string& prefix, int cin, int cout, int k, bool bn);
  // x (B,3,T,H,W) f32 in [-1,1] (H = W = 224), or video_u8 (B,T,H,W,3) uint8 (preprocessed inside) -> logits (B,num_classes)
  void forward(svg_ctx* ctx, const float* x_ncthw, const uint8_t* video_u8, int B, int T, int H, int W, float* logits, hipStream_t s);
};

void xf_train_free(XfModel* m);
void destroy_models(svg_ctx* ctx);

// ---- Stable-Diffusion networks: one implementation per storage type (namespace sd_bf16 / sd_f16), chosen at configure time ----
struct VaeIface {
  WeightStore ws;
  bool ready = false;
  virtual ~VaeIface() {}
  virtual const char* dtype() const = 0;
  virtual void configure(const char* kv) = 0;
  virtual void finalize(svg_ctx* ctx, int64_t* n_params) = 0;
  virtual void encode(svg_ctx* ctx, const uint8_t* img, int N, int srcH, int srcW, int H, int W, const float* eps, float* z_out,
                      float* moments_out, hipStream_t s) = 0;
  virtual void decode(svg_ctx* ctx, const float* z, int N, int h, int w, uint8_t* img_out, int outH, int outW, float* float_out,
                      hipStream_t s) = 0;
};
struct UnetIface {
  WeightStore ws;
  bool ready = false;
  virtual ~UnetIface() {}
  virtual const char* dtype() const = 0;
  virtual void configure(const char* kv) = 0;
  virtual void finalize(svg_ctx* ctx, int64_t* n_params) = 0;
  virtual void forward(svg_ctx* ctx, const float* x, int N, int h, int w, const float* timesteps, const float* ctx_emb, int ctx_len,
                       float* eps_out, hipStream_t s) = 0;
  virtual void ddim_loop(svg_ctx* ctx, float* z, int N, int h, int w, const float* text_emb, int ctx_len, int num_steps,
                         int start_step, float guidance, const float* noise, float* hist, hipStream_t s) = 0;
  virtual void ddim_coefs(int t, int t_prev, float* sa, float* s1a, float* sap, float* s1ap) const = 0;
};
VaeIface* new_vae_bf16();
VaeIface* new_vae_f16();
UnetIface* new_unet_bf16();
UnetIface* new_unet_f16();
namespace sd_bf16 { void sd_init_device(); }
namespace sd_f16 { void sd_init_device(); }

namespace SDNS {

// packed h16 weight matrix [N][K] (+ f32 bias)
struct PackedLinear {
  // MX fp8 copy of w (gemm_fp8.hip): e4m3 elements + one E8M0 scale per 32 K elements; present when the owning model was
  // configured with fp8=1 and the layer qualifies (K % 128 == 0, no folded LayerNorm, no GEGLU)
  uint8_t* w8 = nullptr;
  uint8_t* w8s = nullptr;
  h16* w = nullptr;
  float* b = nullptr;
  float* ln_s = nullptr;     // row sums of w when a LayerNorm (gamma, beta) has been folded into w / b at load
  int N = 0, K = 0, n_valid = 0;
};

// ---- SD VAE -------------------------------------------------------------------------------------------
struct ConvW {
  h16* w = nullptr; float* b = nullptr; int Cin = 0, Cout = 0, Opad = 0;
  // MX fp8 copy (conv_halo_fp8.hip; models configured with fp8=1, stride-1 convs with Cin % 64 == 0): e4m3 [Opad][9][Cp] and
  // E8M0 [9][Cp/128][Opad][4], quantised from the f32 weights at load
  uint8_t* w8 = nullptr; uint8_t* w8s = nullptr; int Cp = 0;
};
struct NormW { float* g = nullptr; float* b = nullptr; int C = 0; };
struct ResW { NormW n1, n2; ConvW c1, c2; PackedLinear sc; bool has_sc = false; int temb_off = -1; int site = 0; /* UNet: bit of its block in the fp8 placement masks */ };
struct VaeAttnW { NormW gn; PackedLinear qk, v, proj; int C = 0; };

struct VaeModel : VaeIface {
  std::vector<int> block_out{128, 256, 512, 512};
  int layers = 2, groups = 32, latent = 4;
  const char* dtype() const override { return SD_F16 ? "fp16" : "bf16"; }
  // encoder
  ConvW e_conv_in, e_conv_out;
  std::vector<std::vector<ResW>> e_down; std::vector<ConvW> e_downs;
  ResW e_mid0, e_mid1; VaeAttnW e_attn; NormW e_norm_out;
  float *quant_w = nullptr, *quant_b = nullptr, *pquant_w = nullptr, *pquant_b = nullptr;
  // decoder
  ConvW d_conv_in, d_conv_out;
  std::vector<std::vector<ResW>> d_up; std::vector<ConvW> d_ups;
  ResW d_mid0, d_mid1; VaeAttnW d_attn; NormW d_norm_out;
  void configure(const char* kv) override;
  void finalize(svg_ctx* ctx, int64_t* n_params) override;
  void encode(svg_ctx* ctx, const uint8_t* img, int N, int srcH, int srcW, int H, int W, const float* eps, float* z_out,
              float* moments_out, hipStream_t s) override;
  void decode(svg_ctx* ctx, const float* z, int N, int h, int w, uint8_t* img_out, int outH, int outW, float* float_out,
              hipStream_t s) override;
};

// ---- SD UNet --------------------------------------------------------------------------------------------
struct XfBlockW {   // SpatialTransformer with one BasicTransformerBlock
  NormW gn, ln1, ln2, ln3;
  PackedLinear proj_in, proj_out, qk1, v1, o1, q2, k2, v2, o2, ff1, ff2;
  PackedLinear qkv1;               // [Wq; Wk; Wv] of the self-attention stacked (C = 320 only): the fused q | k | V^T projection
  float* proj_in_f32 = nullptr;   // [C][C] f32: source of the per-sample GroupNorm-folded weights
  h16* ff2p = nullptr;           // ff.net.2 weights with the k order of the fused GEGLU feed-forward (ff_fused.hip), C = 320 only
  h16* xq = nullptr; float* xq_s = nullptr; float* xq_b = nullptr;   // attn2.to_q for the fused cross-attention (xattn_fused.hip): [384][320] head-padded, LN2 folded
  h16* xqp = nullptr;            // the same with the contraction index in accumulator order (CHAIN form: its B operand comes from registers)
  h16* xo = nullptr;             // attn2.to_out.0 [320][384] with the padded / permuted contraction order of that kernel
  int C = 0;
};
// cross-attention K / V^T of a constant context, computed on the first DDIM step and reused by the others
struct KvCache {
  std::vector<h16*> k, vt;
  std::vector<int64_t> k_cap, vt_cap;
  std::vector<h16*> kp, vp;            // per-head packed copies for the fused cross-attention kernel
  std::vector<int64_t> kvp_cap;
  bool valid = false;
};

struct UnetModel : UnetIface {
  KvCache kv;
  std::vector<int> block_out{320, 640, 1280, 1280};
  std::vector<int> attn{1, 1, 1, 0};
  int layers = 2, heads = 8, ctx_dim = 768, groups = 32, in_ch = 4, out_ch = 4;
  int fp8 = 0;                       // configure key fp8=1: the resnets' / upsamplers' 3x3 convs run on MX block-scaled fp8 operands
  const char* dtype() const override { return SD_F16 ? "fp16" : "bf16"; }
  int temb_dim = 0;
  PackedLinear time1, time2, temb_all;     // temb_all: every resnet's time_emb_proj stacked [sum Cout][temb_dim]
  ConvW conv_in, conv_out; NormW norm_out;
  std::vector<std::vector<ResW>> down_res; std::vector<std::vector<XfBlockW>> down_attn; std::vector<ConvW> down_s;
  ResW mid0, mid1; XfBlockW mid_attn;
  std::vector<std::vector<ResW>> up_res; std::vector<std::vector<XfBlockW>> up_attn; std::vector<ConvW> up_s;
  // DDIM tables (scaled_linear 0.00085..0.012, 1000 train steps)
  std::vector<float> alphas_cumprod;
  void configure(const char* kv) override;
  void finalize(svg_ctx* ctx, int64_t* n_params) override;
  // one UNet call (planned by the caller); cache: cross-attention K / V^T reuse across the steps of a DDIM loop
  void run(svg_ctx* ctx, const float* x, int N, int h, int w, const float* timesteps, const float* ctx_emb, int ctx_len,
           float* eps_out, hipStream_t s, KvCache* cache = nullptr, int64_t fp8_sites = -1);   // fp8_sites: MX-fp8 placement mask of this call (-1: all)
  void forward(svg_ctx* ctx, const float* x, int N, int h, int w, const float* timesteps, const float* ctx_emb, int ctx_len,
               float* eps_out, hipStream_t s) override;
  void ddim_loop(svg_ctx* ctx, float* z, int N, int h, int w, const float* text_emb, int ctx_len, int num_steps,
                 int start_step, float guidance, const float* noise, float* hist, hipStream_t s) override;
  void ddim_coefs(int t, int t_prev, float* sa, float* s1a, float* sap, float* s1ap) const override;
};

// Request for the GroupNorm column sums of an output (GemmArgs::gn_part): the caller provides `buf` (gn_part_floats() floats, at
// the arena scope of the output tensor); the wrapper fills `st` when the launch it chose can emit them (else st stays invalid
// and the consumer runs its own statistics pass).
struct GnEmit {
  float* buf = nullptr;
  GnStats st;
};
// Request for the LayerNorm row partials of a linear()'s output (GemmArgs::ln_part): the caller provides `buf` (M * 16 floats covers
// every tile width here); `tiles` > 0 afterwards when the launch emitted them (else the consumer runs ln_stats).
struct LnEmit {
  float* buf = nullptr;
  int tiles = 0;
};
static inline int64_t gn_part_floats(int64_t B, int64_t HW, int64_t N) { return B * (HW / 128 + 1) * N * 2; }

// shared graph pieces (sdnet.cpp)
// fp8: additionally keep the MX fp8 copy of the weights (ConvW::w8) when the conv qualifies (Cin % 64 == 0)
ConvW load_conv3x3(svg_ctx* ctx, WeightStore& ws, const std::string& prefix, int Cin, int Cout, hipStream_t s, bool fp8 = false);
// fold != nullptr: the LayerNorm (gamma, beta) in front of this projection is folded into the packed weights (GemmArgs::ln_*)
PackedLinear load_linear(svg_ctx* ctx, WeightStore& ws, const std::string& prefix, int N, int K, bool bias, hipStream_t s,
                         const NormW* fold = nullptr);
NormW load_norm(svg_ctx* ctx, WeightStore& ws, const std::string& prefix, int C);
// adds the MX fp8 copy of a packed linear when it qualifies (see PackedLinear::w8)
void add_fp8_copy(svg_ctx* ctx, PackedLinear& pl, hipStream_t s);
float* keep_f32(svg_ctx* ctx, WeightStore& ws, const std::string& name, int64_t numel);
// out (B,Ho,Wo,Cout) = conv3x3(x) + bias [+ per-sample bias] [+ residual]
void conv3x3(svg_ctx* ctx, const h16* x, const ConvW& cw, void* out, int B, int H, int W, int amode, const float* bias_bn,
             int bias_bn_ld, const h16* residual, int out_f32, hipStream_t s, GnEmit* emit = nullptr);
// the same stride-1 conv on MX fp8 operands (conv_halo_fp8.hip): x8 / xs = the quantised input of quant_act_mx (or of the quantising
// GroupNorm apply pass); the caller asks conv3x3_fp8_ok() first
// up2: nearest-2x upsample fused in front (the UNet's upsamplers): x8 is the H x W source, the output is 2H x 2W
bool conv3x3_fp8_ok(const ConvW& cw, int B, int H, int W, bool up2 = false);
void conv3x3_fp8(svg_ctx* ctx, const uint8_t* x8, const uint8_t* xs, const ConvW& cw, h16* out, int B, int H, int W, const float* bias_bn,
                 int bias_bn_ld, const h16* residual, hipStream_t s, GnEmit* emit = nullptr, bool up2 = false);
// C[M,N] = act(A[M,K] W^T + b) [+ residual]
// emit / rows_per_sample: GroupNorm column sums of the output (M = samples x rows_per_sample).  A2 / k_split: the A operand is the
// channel concat [A | A2] of two tensors (columns >= k_split come from A2, row stride lda2) without materialising it.
void linear(svg_ctx* ctx, const h16* A, int lda, const PackedLinear& pl, void* C, int ldc, int M, int act, const h16* residual,
            int ldr, int out_f32, hipStream_t s, const float* ln_rs = nullptr, const float* ln_rm = nullptr, GnEmit* emit = nullptr,
            int rows_per_sample = 0, const h16* A2 = nullptr, int lda2 = 0, int k_split = 0, LnEmit* ln = nullptr);

}  // namespace SDNS
