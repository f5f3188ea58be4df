// SD AutoencoderKL graph (diffusers 0.2.x layout) on NHWC bf16 activations.
// Reference call sites: utils/sd_utils.py:128-145 (encode_img) and :156-169 (decode_img_latents).
#include "models.h"

namespace SDNS {

void VaeModel::configure(const char* kv) {
  auto m = parse_kv(kv);
  if (m.count("block_out")) { block_out.clear(); for (auto v : m["block_out"]) block_out.push_back((int)v); }
  if (m.count("layers")) layers = (int)m["layers"][0];
  if (m.count("groups")) groups = (int)m["groups"][0];
  if (m.count("latent")) latent = (int)m["latent"][0];
  ready = false;
}

static ResW load_res(svg_ctx* ctx, WeightStore& ws, const std::string& p, int cin, int cout, hipStream_t s) {
  ResW r;
  r.n1 = load_norm(ctx, ws, p + ".norm1", cin);
  r.c1 = load_conv3x3(ctx, ws, p + ".conv1", cin, cout, s);
  r.n2 = load_norm(ctx, ws, p + ".norm2", cout);
  r.c2 = load_conv3x3(ctx, ws, p + ".conv2", cout, cout, s);
  r.has_sc = cin != cout;
  if (r.has_sc) r.sc = load_linear(ctx, ws, p + ".conv_shortcut", cout, cin, true, s);
  return r;
}

static VaeAttnW load_vae_attn(svg_ctx* ctx, WeightStore& ws, const std::string& p, int C, hipStream_t s) {
  VaeAttnW a;
  a.C = C;
  a.gn = load_norm(ctx, ws, p + ".group_norm", C);
  // q and k share one GEMM: rows [Wq; Wk]
  {
    const Weight& wq = ws.get(p + ".query.weight", {C, C});
    const Weight& wk = ws.get(p + ".key.weight", {C, C});
    a.qk.N = 2 * C; a.qk.K = C; a.qk.n_valid = 2 * C;
    a.qk.w = (h16*)ctx->dalloc((int64_t)2 * C * C * sizeof(h16));
    pack_linear(wq.f32, a.qk.w, C, C, C, s);
    pack_linear(wk.f32, a.qk.w + (int64_t)C * C, C, C, C, s);
    a.qk.b = (float*)ctx->dalloc(2 * C * sizeof(float));
    HIP_OK(hipMemcpyAsync(a.qk.b, keep_f32(ctx, ws, p + ".query.bias", C), C * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIP_OK(hipMemcpyAsync(a.qk.b + C, keep_f32(ctx, ws, p + ".key.bias", C), C * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIP_OK(hipStreamSynchronize(s));
    ws.release(p + ".query.weight"); ws.release(p + ".key.weight");
  }
  a.v = load_linear(ctx, ws, p + ".value", C, C, true, s);
  a.proj = load_linear(ctx, ws, p + ".proj_attn", C, C, true, s);
  return a;
}

void VaeModel::finalize(svg_ctx* ctx, int64_t* n_params) {
  hipStream_t s = nullptr;
  SVG_CHECK(latent == 4, "vae: latent channels must be 4");
  const int nb = (int)block_out.size();
  const int64_t total = ws.total_params();
  // ---- encoder
  e_conv_in = load_conv3x3(ctx, ws, "encoder.conv_in", 3, block_out[0], s);
  e_down.clear(); e_downs.clear();
  int cin = block_out[0];
  for (int i = 0; i < nb; ++i) {
    std::vector<ResW> rs;
    for (int j = 0; j < layers; ++j) {
      rs.push_back(load_res(ctx, ws, "encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), cin, block_out[i], s));
      cin = block_out[i];
    }
    e_down.push_back(rs);
    if (i < nb - 1) e_downs.push_back(load_conv3x3(ctx, ws, "encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv", cin, cin, s));
  }
  const int cm = block_out[nb - 1];
  e_mid0 = load_res(ctx, ws, "encoder.mid_block.resnets.0", cm, cm, s);
  e_attn = load_vae_attn(ctx, ws, "encoder.mid_block.attentions.0", cm, s);
  e_mid1 = load_res(ctx, ws, "encoder.mid_block.resnets.1", cm, cm, s);
  e_norm_out = load_norm(ctx, ws, "encoder.conv_norm_out", cm);
  e_conv_out = load_conv3x3(ctx, ws, "encoder.conv_out", cm, 2 * latent, s);
  ws.get("quant_conv.weight", {8, 8, 1, 1}); ws.get("post_quant_conv.weight", {4, 4, 1, 1});
  quant_w = keep_f32(ctx, ws, "quant_conv.weight", 64); quant_b = keep_f32(ctx, ws, "quant_conv.bias", 8);
  pquant_w = keep_f32(ctx, ws, "post_quant_conv.weight", 16); pquant_b = keep_f32(ctx, ws, "post_quant_conv.bias", 4);
  // ---- decoder (block_out reversed; layers+1 resnets per up block)
  d_conv_in = load_conv3x3(ctx, ws, "decoder.conv_in", latent, cm, s);
  d_mid0 = load_res(ctx, ws, "decoder.mid_block.resnets.0", cm, cm, s);
  d_attn = load_vae_attn(ctx, ws, "decoder.mid_block.attentions.0", cm, s);
  d_mid1 = load_res(ctx, ws, "decoder.mid_block.resnets.1", cm, cm, s);
  d_up.clear(); d_ups.clear();
  cin = cm;
  for (int i = 0; i < nb; ++i) {
    const int cout = block_out[nb - 1 - i];
    std::vector<ResW> rs;
    for (int j = 0; j < layers + 1; ++j) {
      rs.push_back(load_res(ctx, ws, "decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), cin, cout, s));
      cin = cout;
    }
    d_up.push_back(rs);
    if (i < nb - 1) d_ups.push_back(load_conv3x3(ctx, ws, "decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv", cin, cin, s));
  }
  d_norm_out = load_norm(ctx, ws, "decoder.conv_norm_out", block_out[0]);
  d_conv_out = load_conv3x3(ctx, ws, "decoder.conv_out", block_out[0], 3, s);
  if (n_params) *n_params = total;
  ready = true;
}

namespace {
struct VaeRun {
  svg_ctx* ctx; VaeModel* m; hipStream_t s; int N;
  static constexpr float EPS = 1e-6f;

  // an activation tensor with, when its producer's epilogue left them, the GroupNorm column sums of its row tiles
  struct Act {
    h16* p = nullptr;
    GnStats st;
  };
  GnEmit emit_for(int64_t hw, int Cout) {
    GnEmit e;
    if (hw >= 1024) e.buf = ctx->arena.get<float>(gn_part_floats(N, hw, Cout));
    return e;
  }
  Act conv(const h16* x, const ConvW& cw, int H, int W, int amode) {
    const int Ho = amode == A_CONV_UP2 ? 2 * H : (amode == A_CONV_S2ASYM ? H / 2 : H), Wo = amode == A_CONV_UP2 ? 2 * W : (amode == A_CONV_S2ASYM ? W / 2 : W);
    Act y;
    y.p = ctx->arena.get<h16>((int64_t)N * Ho * Wo * cw.Opad);
    GnEmit e = emit_for((int64_t)Ho * Wo, cw.Opad);
    conv3x3(ctx, x, cw, y.p, N, H, W, amode, nullptr, 0, nullptr, 0, s, &e);
    y.st = e.st;
    return y;
  }

  // out = conv2(silu(gn2(conv1(silu(gn1(x)))))) + shortcut(x)
  Act resnet(const Act& x, const ResW& r, int H, int W) {
    const int64_t P = (int64_t)N * H * W;
    Act out;
    out.p = ctx->arena.get<h16>(P * r.c2.Opad);
    GnEmit eo = emit_for((int64_t)H * W, r.c2.Opad);
    ctx->arena.push();
    h16* t0 = ctx->arena.get<h16>(P * r.n1.C);
    groupnorm(ctx, x.p, r.n1.C, nullptr, 0, r.n1.g, r.n1.b, t0, N, H * W, m->groups, EPS, 1, s, &x.st, nullptr);
    h16* t1 = ctx->arena.get<h16>(P * r.c1.Opad);
    GnEmit e1 = emit_for((int64_t)H * W, r.c1.Opad);
    conv3x3(ctx, t0, r.c1, t1, N, H, W, A_CONV_S1, nullptr, 0, nullptr, 0, s, &e1);
    h16* t2 = ctx->arena.get<h16>(P * r.n2.C);
    groupnorm(ctx, t1, r.n2.C, nullptr, 0, r.n2.g, r.n2.b, t2, N, H * W, m->groups, EPS, 1, s, &e1.st, nullptr);
    const h16* res = x.p;
    if (r.has_sc) {
      h16* sc = ctx->arena.get<h16>(P * r.sc.N);
      linear(ctx, x.p, r.n1.C, r.sc, sc, r.sc.N, (int)P, ACT_NONE, nullptr, 0, 0, s);
      res = sc;
    }
    conv3x3(ctx, t2, r.c2, out.p, N, H, W, A_CONV_S1, nullptr, 0, res, 0, s, &eo);
    ctx->arena.pop();
    out.st = eo.st;
    return out;
  }

  // single-head attention over HW tokens (d = C = 512): fused (attn_vae.hip) when HW is a multiple of 64, else three GEMMs + row softmax
  Act attn(const Act& xa, const VaeAttnW& a, int H, int W) {
    const h16* x = xa.p;
    const int HW = H * W, C = a.C;
    const int64_t P = (int64_t)N * HW;
    const int HWp = (int)align_up(HW, 8);
    h16* out = ctx->arena.get<h16>(P * C);
    GnEmit eo = emit_for(HW, C);
    ctx->arena.push();
    h16* n = ctx->arena.get<h16>(P * C);
    groupnorm(ctx, x, C, nullptr, 0, a.gn.g, a.gn.b, n, N, HW, m->groups, EPS, 0, s, &xa.st, nullptr);
    h16* qk = ctx->arena.get<h16>(P * 2 * C);
    linear(ctx, n, C, a.qk, qk, 2 * C, (int)P, ACT_NONE, nullptr, 0, 0, s);
    // V^T[b] = Wv * n_b^T + bv (per row)
    h16* vt = ctx->arena.get<h16>((int64_t)N * C * HWp);
    {
      GemmArgs g;
      g.A = a.v.w; g.lda = C; g.Wt = n; g.ldb = C; g.M = C; g.N = HWp; g.n_valid = HW; g.K = C;
      g.batch = N; g.sA = 0; g.sB = (int64_t)HW * C; g.sC = (int64_t)C * HWp;
      g.bias = a.v.b; g.bias_row = 1; g.C = vt; g.ldc = HWp;
      gemm_auto(ctx, g, s, PK_GEMM);
    }
    h16* o = ctx->arena.get<h16>(P * C);
    if (vae_attention_supported(HW, C, 2 * C, HWp, C)) {
      // flash-style, the head dimension split over the waves of a workgroup (attn_vae.hip): no S x S matrix in HBM
      vae_attention(ctx, qk, qk + C, 2 * C, (int64_t)HW * 2 * C, vt, HWp, (int64_t)C * HWp, o, C, (int64_t)HW * C, N, HW, C, s);
    } else {
      float* S = ctx->arena.get<float>((int64_t)N * HW * HWp);
      {
        GemmArgs g;
        g.A = qk; g.lda = 2 * C; g.Wt = qk + C; g.ldb = 2 * C; g.M = HW; g.N = HWp; g.n_valid = HW; g.K = C;
        g.batch = N; g.sA = (int64_t)HW * 2 * C; g.sB = (int64_t)HW * 2 * C; g.sC = (int64_t)HW * HWp;
        g.C = S; g.ldc = HWp; g.out_f32 = 1;
        gemm_auto(ctx, g, s, PK_GEMM);
      }
      h16* Pm = ctx->arena.get<h16>((int64_t)N * HW * HWp);
      softmax_rows(ctx, S, Pm, (int64_t)N * HW, HW, HWp, HWp, 1.f / sqrtf((float)C), s);
      {
        GemmArgs g;
        g.A = Pm; g.lda = HWp; g.Wt = vt; g.ldb = HWp; g.M = HW; g.N = C; g.n_valid = C; g.K = HWp;
        g.batch = N; g.sA = (int64_t)HW * HWp; g.sB = (int64_t)C * HWp; g.sC = (int64_t)HW * C;
        g.C = o; g.ldc = C;
        gemm_auto(ctx, g, s, PK_GEMM);
      }
    }
    linear(ctx, o, C, a.proj, out, C, (int)P, ACT_NONE, x, C, 0, s, nullptr, nullptr, &eo, HW);
    ctx->arena.pop();
    Act y;
    y.p = out; y.st = eo.st;
    return y;
  }

  h16* norm_act(const Act& x, const NormW& nw, int H, int W) {
    h16* t = ctx->arena.get<h16>((int64_t)N * H * W * nw.C);
    groupnorm(ctx, x.p, nw.C, nullptr, 0, nw.g, nw.b, t, N, H * W, m->groups, EPS, 1, s, &x.st, nullptr);
    return t;
  }
};
}  // namespace

void VaeModel::encode(svg_ctx* ctx, const uint8_t* img, int N, int srcH, int srcW, int H, int W, const float* eps, float* z_out,
                      float* moments_out, hipStream_t s) {
  SVG_CHECK(ready, "vae: svg_finalize has not been called");
  const int nb = (int)block_out.size();
  const int down = 1 << (nb - 1);
  SVG_CHECK(N >= 1 && H % down == 0 && W % down == 0 && H >= down && W >= down, "vae encode: bad size %dx%d (batch %d)", H, W, N);
  run_planned(ctx, [&]() {
    VaeRun r{ctx, this, s, N};
    h16* x0 = ctx->arena.get<h16>((int64_t)N * H * W * 8);
    if (SVG_LAUNCHING(ctx)) { ProfScope ps(ctx, PK_ELT, s, 0, 0); img_to_act(img, x0, N, srcH, srcW, H, W, s); }
    int h = H, w = W;
    VaeRun::Act x = r.conv(x0, e_conv_in, h, w, A_CONV_S1);
    for (int i = 0; i < nb; ++i) {
      for (auto& rw : e_down[i]) x = r.resnet(x, rw, h, w);
      if (i < nb - 1) {
        x = r.conv(x.p, e_downs[i], h, w, A_CONV_S2ASYM);
        h /= 2; w /= 2;
      }
    }
    x = r.resnet(x, e_mid0, h, w);
    x = r.attn(x, e_attn, h, w);
    x = r.resnet(x, e_mid1, h, w);
    h16* t = r.norm_act(x, e_norm_out, h, w);
    const int64_t P = (int64_t)N * h * w;
    float* mom0 = ctx->arena.get<float>(P * 8);
    conv3x3(ctx, t, e_conv_out, mom0, N, h, w, A_CONV_S1, nullptr, 0, nullptr, 1, s);
    float* mom = ctx->arena.get<float>(P * 8);
    if (SVG_LAUNCHING(ctx)) {
      ProfScope ps(ctx, PK_ELT, s, 0, 0);
      pixel_linear_f32(mom0, 8, quant_w, quant_b, mom, 8, P, 8, 8, s);
      vae_sample(mom, 8, eps, z_out, moments_out, N, h, w, s);
    }
  });
}

void VaeModel::decode(svg_ctx* ctx, const float* z, int N, int h, int w, uint8_t* img_out, int outH, int outW, float* float_out,
                      hipStream_t s) {
  SVG_CHECK(ready, "vae: svg_finalize has not been called");
  SVG_CHECK(N >= 1 && h >= 1 && w >= 1, "vae decode: bad size");
  const int nb = (int)block_out.size();
  run_planned(ctx, [&]() {
    VaeRun r{ctx, this, s, N};
    const int64_t P0 = (int64_t)N * h * w;
    // sd_utils.py:159: latents / 0.18215, then post_quant_conv (1x1, 4->4) in f32
    float* zl = ctx->arena.get<float>(P0 * 4);
    float* zq = ctx->arena.get<float>(P0 * 4);
    h16* x0 = ctx->arena.get<h16>(P0 * 8);
    if (SVG_LAUNCHING(ctx)) {
      ProfScope ps(ctx, PK_ELT, s, 0, 0);
      nchw_to_actf32(z, zl, N, 4, h, w, 1.f / 0.18215f, s);
      pixel_linear_f32(zl, 4, pquant_w, pquant_b, zq, 4, P0, 4, 4, s);
      actf32_pad_h16(zq, 4, x0, 8, P0, s);
    }
    int H = h, W = w;
    VaeRun::Act x = r.conv(x0, d_conv_in, H, W, A_CONV_S1);
    x = r.resnet(x, d_mid0, H, W);
    x = r.attn(x, d_attn, H, W);
    x = r.resnet(x, d_mid1, H, W);
    for (int i = 0; i < nb; ++i) {
      for (auto& rw : d_up[i]) x = r.resnet(x, rw, H, W);
      if (i < nb - 1) {
        x = r.conv(x.p, d_ups[i], H, W, A_CONV_UP2);
        H *= 2; W *= 2;
      }
    }
    h16* t = r.norm_act(x, d_norm_out, H, W);
    float* o = ctx->arena.get<float>((int64_t)N * H * W * 4);
    conv3x3(ctx, t, d_conv_out, o, N, H, W, A_CONV_S1, nullptr, 0, nullptr, 1, s);
    if (SVG_LAUNCHING(ctx)) {
      ProfScope ps(ctx, PK_ELT, s, 0, 0);
      act_to_img(o, 4, img_out, float_out, N, H, W, img_out ? outH : H, img_out ? outW : W, s);
    }
  });
}

}  // namespace SDNS

#if SD_F16
VaeIface* new_vae_f16() { return new sd_f16::VaeModel(); }
#else
VaeIface* new_vae_bf16() { return new sd_bf16::VaeModel(); }
#endif
