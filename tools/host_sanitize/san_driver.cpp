// Drives the library's host code through the C ABI on the stub runtime (hip_stub.cpp) under ASan / UBSan: every model slot at a
// reduced size (the graphs, planners and packers do not depend on the widths), the chunked / split paths, both stream kinds (the
// null stream: direct launches; a non-null token: the hipGraph capture branches), the error paths, reconfiguration and destroy.
#include "../../include/svg_hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define OK(call)                                                                                       \
  do {                                                                                                 \
    int rc_ = (call);                                                                                  \
    if (rc_ != 0) { fprintf(stderr, "FAILED %s -> %d: %s\n", #call, rc_, svg_last_error(ctx)); exit(2); } \
  } while (0)
#define EXPECT_ERR(call)                                                                     \
  do {                                                                                       \
    int rc_ = (call);                                                                        \
    if (rc_ == 0) { fprintf(stderr, "expected an error from %s\n", #call); exit(3); }        \
  } while (0)

static svg_ctx* ctx;
static std::vector<float> buf(size_t n, float v = 0.01f) { return std::vector<float>(n, v); }

static void load(int model, const std::string& name, std::vector<int64_t> shape, float v = 0.01f) {
  size_t n = 1;
  for (auto s : shape) n *= (size_t)s;
  auto d = buf(n, v);
  OK(svg_load_weight(ctx, model, name.c_str(), d.data(), shape.data(), (int)shape.size()));
}
static void load_lin(int model, const std::string& p, int n, int k, bool bias = true) {
  load(model, p + ".weight", {n, k});
  if (bias) load(model, p + ".bias", {n});
}
static void load_norm(int model, const std::string& p, int c) { load(model, p + ".weight", {c}, 1.f); load(model, p + ".bias", {c}, 0.f); }
static void load_conv(int model, const std::string& p, int cin, int cout) { load(model, p + ".weight", {cout, cin, 3, 3}); load(model, p + ".bias", {cout}); }
static void load_res(int model, const std::string& p, int cin, int cout, int temb) {
  load_norm(model, p + ".norm1", cin); load_conv(model, p + ".conv1", cin, cout);
  load_norm(model, p + ".norm2", cout); load_conv(model, p + ".conv2", cout, cout);
  if (cin != cout) { load(model, p + ".conv_shortcut.weight", {cout, cin, 1, 1}); load(model, p + ".conv_shortcut.bias", {cout}); }
  if (temb) load_lin(model, p + ".time_emb_proj", cout, temb);
}

// dm = 128, d_lat = 128, ffn = 256: shapes the layer-walking launch takes (csrc/xf_walk.hip: its stage table is built on the host)
static void transformer(int text_dim, int dm = 32, int d_lat = 64, int ffn = 2048) {
  const int d = dm + text_dim;
  char kv[256];
  snprintf(kv, sizeof(kv), "d_lat=%d;d_model=%d;heads=4;enc_layers=1;dec_layers=1;ffn=%d;text_dim=%d", d_lat, d, ffn, text_dim);
  OK(svg_model_configure(ctx, SVG_TRANSFORMER, kv));
  const int T = SVG_TRANSFORMER;
  load_lin(T, text_dim ? "project_image_embedding" : "embedding", dm, d_lat);
  load(T, "positional_encoder.pos_encoding", {64, 1, d});
  auto attn = [&](const std::string& p) { load(T, p + "in_proj_weight", {3 * d, d}); load(T, p + "in_proj_bias", {3 * d}); load_lin(T, p + "out_proj", d, d); };
  std::string e = "transformer.encoder.layers.0.", dd = "transformer.decoder.layers.0.";
  attn(e + "self_attn."); load_lin(T, e + "linear1", ffn, d); load_lin(T, e + "linear2", d, ffn); load_norm(T, e + "norm1", d); load_norm(T, e + "norm2", d);
  attn(dd + "self_attn."); attn(dd + "multihead_attn."); load_lin(T, dd + "linear1", ffn, d); load_lin(T, dd + "linear2", d, ffn);
  load_norm(T, dd + "norm1", d); load_norm(T, dd + "norm2", d); load_norm(T, dd + "norm3", d);
  load_norm(T, "transformer.encoder.norm", d); load_norm(T, "transformer.decoder.norm", d);
  load_lin(T, "out", d_lat, d);
  int64_t n = 0;
  OK(svg_finalize(ctx, T, &n));
  for (int B : {1, 3, 70}) {                       // 70 x 6 rows: two passes of the 336-row weight stream
    auto x = buf((size_t)B * 6 * d_lat), y = buf((size_t)6 * B * d_lat), txt = buf((size_t)B * 384), pad = buf((size_t)B * 6, 0.f);
    std::vector<int32_t> pe(B, 0);
    if (text_dim) OK(svg_transformer_forward_text(ctx, x.data(), x.data(), txt.data(), B, 6, 6, nullptr, pe.data(), y.data(), nullptr));
    else if (B <= 64) OK(svg_transformer_forward(ctx, x.data(), x.data(), B, 6, 6, nullptr, nullptr, y.data(), nullptr));
    OK(svg_transformer_forward_padded(ctx, x.data(), x.data(), text_dim ? txt.data() : nullptr, B, 6, 6, nullptr, pad.data(), pad.data(), pe.data(), y.data(), nullptr));
  }
  EXPECT_ERR(svg_transformer_forward(ctx, nullptr, nullptr, 65, 6, 6, nullptr, nullptr, nullptr, nullptr));     // batch > 64 without pe_row
  // training step: direct launches (null stream), then the captured-graph branch (any non-null stream token), then Adam
  svg_train_cfg tc;
  memset(&tc, 0, sizeof(tc));
  tc.frames_to_predict = 2; tc.feat_h = 4; tc.feat_w = d_lat / 16;      // d_lat = 4 channels x feat_h x feat_w
  tc.w_mse = 1.f; tc.w_gdl = 1.f; tc.gdl_alpha = 2.f; tc.w_contrastive = 0.1f; tc.temperature = 0.07f;
  tc.dropout_p = 0.1f; tc.seed = 3;
  const int B = 4, Ts = 6, Tt = 5;
  auto src = buf((size_t)B * Ts * d_lat), tgt = buf((size_t)B * Tt * d_lat), txt = buf((size_t)B * 384);
  float losses[5];
  for (void* stream : {(void*)nullptr, (void*)0x10}) {
    for (int rep = 0; rep < 2; ++rep) {
      OK(svg_transformer_loss(ctx, &tc, src.data(), tgt.data(), tgt.data(), text_dim ? txt.data() : nullptr, B, Ts, Tt, nullptr, 1, losses, stream));
      OK(svg_transformer_adam_step(ctx, 1e-3f, 0.9f, 0.999f, 1e-8f, stream));
    }
  }
  OK(svg_transformer_loss(ctx, &tc, src.data(), tgt.data(), tgt.data(), text_dim ? txt.data() : nullptr, B, Ts, Tt, nullptr, 0, losses, nullptr));
  auto w = buf((size_t)d_lat * d);
  OK(svg_transformer_tensor(ctx, SVG_TENSOR_GRAD, "out.weight", w.data(), (int64_t)w.size(), nullptr));
  EXPECT_ERR(svg_transformer_tensor(ctx, SVG_TENSOR_PARAM, "no.such.weight", w.data(), 1, nullptr));
}

static void vae(int f16) {
  const int V = SVG_VAE;
  OK(svg_model_configure(ctx, V, f16 ? "block_out=64,128;layers=1;groups=32;latent=4;f16=1" : "block_out=64,128;layers=1;groups=32;latent=4"));
  auto attn = [&](const std::string& p, int c) {
    load_norm(V, p + ".group_norm", c);
    for (const char* n : {"query", "key", "value", "proj_attn"}) load_lin(V, p + "." + n, c, c);
  };
  load_conv(V, "encoder.conv_in", 3, 64);
  load_res(V, "encoder.down_blocks.0.resnets.0", 64, 64, 0); load_conv(V, "encoder.down_blocks.0.downsamplers.0.conv", 64, 64);
  load_res(V, "encoder.down_blocks.1.resnets.0", 64, 128, 0);
  load_res(V, "encoder.mid_block.resnets.0", 128, 128, 0); attn("encoder.mid_block.attentions.0", 128); load_res(V, "encoder.mid_block.resnets.1", 128, 128, 0);
  load_norm(V, "encoder.conv_norm_out", 128); load_conv(V, "encoder.conv_out", 128, 8);
  load(V, "quant_conv.weight", {8, 8, 1, 1}); load(V, "quant_conv.bias", {8});
  load(V, "post_quant_conv.weight", {4, 4, 1, 1}); load(V, "post_quant_conv.bias", {4});
  load_conv(V, "decoder.conv_in", 4, 128);
  load_res(V, "decoder.mid_block.resnets.0", 128, 128, 0); attn("decoder.mid_block.attentions.0", 128); load_res(V, "decoder.mid_block.resnets.1", 128, 128, 0);
  for (int j = 0; j < 2; ++j) load_res(V, "decoder.up_blocks.0.resnets." + std::to_string(j), 128, 128, 0);
  load_conv(V, "decoder.up_blocks.0.upsamplers.0.conv", 128, 128);
  load_res(V, "decoder.up_blocks.1.resnets.0", 128, 64, 0); load_res(V, "decoder.up_blocks.1.resnets.1", 64, 64, 0);
  load_norm(V, "decoder.conv_norm_out", 64); load_conv(V, "decoder.conv_out", 64, 3);
  OK(svg_finalize(ctx, V, nullptr));
  const int N = 3, H = 32;
  std::vector<uint8_t> img((size_t)N * 16 * 16 * 3, 100), out((size_t)N * 24 * 24 * 3);
  auto eps = buf((size_t)N * 4 * 16 * 16), z = buf((size_t)N * 4 * 16 * 16), mom = buf((size_t)N * 8 * 16 * 16), fl = buf((size_t)N * 3 * H * H);
  OK(svg_vae_encode(ctx, img.data(), N, 16, 16, H, H, eps.data(), z.data(), mom.data(), nullptr));         // fused 16 -> 32 resize
  setenv("SVG_CHUNK_LIMIT", "400000", 1);                                                                    // the batch / row chunking of conv3x3() and linear()
  OK(svg_vae_encode(ctx, img.data(), N, 16, 16, H, H, nullptr, z.data(), nullptr, nullptr));
  OK(svg_vae_decode(ctx, z.data(), N, H / 2, H / 2, out.data(), 24, 24, fl.data(), nullptr));
  unsetenv("SVG_CHUNK_LIMIT");
  OK(svg_vae_decode(ctx, z.data(), N, H / 2, H / 2, out.data(), 24, 24, nullptr, nullptr));
  EXPECT_ERR(svg_vae_decode(ctx, z.data(), 0, 4, 4, out.data(), 8, 8, nullptr, nullptr));
}

static void unet(int f16, int fp8) {
  const int U = SVG_UNET, c0 = 64, c1 = 128, temb = 256, cd = 64;
  char kv[256];
  snprintf(kv, sizeof(kv), "block_out=64,128;layers=1;heads=4;ctx_dim=%d;groups=32;in_ch=4;out_ch=4;attn=1,0;f16=%d;fp8=%d", cd, f16, fp8);
  OK(svg_model_configure(ctx, U, kv));
  auto xf = [&](const std::string& p, int c) {
    load_norm(U, p + ".norm", c); load(U, p + ".proj_in.weight", {c, c, 1, 1}); load(U, p + ".proj_in.bias", {c});
    load(U, p + ".proj_out.weight", {c, c, 1, 1}); load(U, p + ".proj_out.bias", {c});
    const std::string t = p + ".transformer_blocks.0";
    for (const char* n : {".norm1", ".norm2", ".norm3"}) load_norm(U, t + n, c);
    for (const char* n : {".attn1.to_q", ".attn1.to_k", ".attn1.to_v", ".attn2.to_q"}) load_lin(U, t + n, c, c, false);
    load_lin(U, t + ".attn2.to_k", c, cd, false); load_lin(U, t + ".attn2.to_v", c, cd, false);
    load_lin(U, t + ".attn1.to_out.0", c, c); load_lin(U, t + ".attn2.to_out.0", c, c);
    load_lin(U, t + ".ff.net.0.proj", 8 * c, c); load_lin(U, t + ".ff.net.2", c, 4 * c);
  };
  load_lin(U, "time_embedding.linear_1", temb, c0); load_lin(U, "time_embedding.linear_2", temb, temb);
  load_conv(U, "conv_in", 4, c0);
  load_res(U, "down_blocks.0.resnets.0", c0, c0, temb); xf("down_blocks.0.attentions.0", c0); load_conv(U, "down_blocks.0.downsamplers.0.conv", c0, c0);
  load_res(U, "down_blocks.1.resnets.0", c0, c1, temb);
  load_res(U, "mid_block.resnets.0", c1, c1, temb); xf("mid_block.attentions.0", c1); load_res(U, "mid_block.resnets.1", c1, c1, temb);
  load_res(U, "up_blocks.0.resnets.0", c1 + c1, c1, temb); load_res(U, "up_blocks.0.resnets.1", c1 + c0, c1, temb);
  load_conv(U, "up_blocks.0.upsamplers.0.conv", c1, c1);
  load_res(U, "up_blocks.1.resnets.0", c1 + c0, c0, temb); xf("up_blocks.1.attentions.0", c0);
  load_res(U, "up_blocks.1.resnets.1", c0 + c0, c0, temb); xf("up_blocks.1.attentions.1", c0);
  load_norm(U, "conv_norm_out", c0); load_conv(U, "conv_out", c0, 4);
  OK(svg_finalize(ctx, U, nullptr));
  const int N = 2, h = 32, L = 7;
  auto x = buf((size_t)N * 4 * h * h), t = buf(N, 500.f), emb = buf((size_t)2 * N * L * cd), e = buf((size_t)N * 4 * h * h), noise = buf((size_t)N * 4 * h * h);
  auto hist = buf((size_t)5 * N * 4 * h * h);
  OK(svg_unet_forward(ctx, x.data(), N, h, h, t.data(), emb.data(), L, e.data(), nullptr));
  OK(svg_ddim_loop(ctx, x.data(), N, h, h, emb.data(), L, 50, 46, 7.5f, noise.data(), hist.data(), nullptr));      // CFG, history, direct launches
  OK(svg_ddim_loop(ctx, x.data(), N, h, h, emb.data(), L, 50, 40, 0.f, noise.data(), nullptr, (void*)0x10));       // the hipGraph branch
  OK(svg_ddim_loop(ctx, x.data(), N, h, h, emb.data(), L, 50, 50, 0.f, noise.data(), nullptr, nullptr));           // zero steps
  EXPECT_ERR(svg_ddim_loop(ctx, x.data(), N, h, h, emb.data(), L, 50, 10, 0.f, nullptr, nullptr, nullptr));         // start_step > 0 without noise
  OK(svg_ddim_step(ctx, x.data(), e.data(), x.data(), (int64_t)x.size(), 980, 960, nullptr));
  // a larger batch at 64 x 64: the weight-stationary / ping-pong / fused feed-forward selections (C = 320 needs the SD widths: not here)
  auto x2 = buf((size_t)6 * 4 * 64 * 64), t2 = buf(6, 20.f), emb2 = buf((size_t)6 * L * cd), e2 = buf((size_t)6 * 4 * 64 * 64);
  OK(svg_unet_forward(ctx, x2.data(), 6, 64, 64, t2.data(), emb2.data(), L, e2.data(), nullptr));
}

static void text_towers() {
  OK(svg_model_configure(ctx, SVG_CLIP_TEXT, "vocab=100;d_model=64;heads=4;layers=2;ffn=128;max_pos=77"));
  const int C = SVG_CLIP_TEXT;
  load(C, "embeddings.token_embedding.weight", {100, 64}); load(C, "embeddings.position_embedding.weight", {77, 64}); load_norm(C, "final_layer_norm", 64);
  for (int i = 0; i < 2; ++i) {
    const std::string p = "encoder.layers." + std::to_string(i) + ".";
    for (const char* n : {"self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.out_proj"}) load_lin(C, p + n, 64, 64);
    load_lin(C, p + "mlp.fc1", 128, 64); load_lin(C, p + "mlp.fc2", 64, 128); load_norm(C, p + "layer_norm1", 64); load_norm(C, p + "layer_norm2", 64);
  }
  OK(svg_finalize(ctx, C, nullptr));
  std::vector<int32_t> ids(6 * 77, 5);
  auto out = buf((size_t)6 * 77 * 64);
  OK(svg_clip_text_forward(ctx, ids.data(), 6, 77, out.data(), nullptr));          // 6 x 77 rows: two passes
  EXPECT_ERR(svg_clip_text_forward(ctx, ids.data(), 1, 78, out.data(), nullptr));
  OK(svg_model_configure(ctx, SVG_MINILM, "vocab=200;d_model=64;heads=4;layers=2;ffn=128;max_pos=64"));
  const int M = SVG_MINILM;
  load(M, "embeddings.word_embeddings.weight", {200, 64}); load(M, "embeddings.position_embeddings.weight", {64, 64});
  load(M, "embeddings.token_type_embeddings.weight", {2, 64}); load_norm(M, "embeddings.LayerNorm", 64);
  for (int i = 0; i < 2; ++i) {
    const std::string p = "encoder.layer." + std::to_string(i) + ".";
    for (const char* n : {"attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"}) load_lin(M, p + n, 64, 64);
    load_lin(M, p + "intermediate.dense", 128, 64); load_lin(M, p + "output.dense", 64, 128);
    load_norm(M, p + "attention.output.LayerNorm", 64); load_norm(M, p + "output.LayerNorm", 64);
  }
  OK(svg_finalize(ctx, M, nullptr));
  std::vector<int32_t> ids2(40 * 12, 7), lens(40, 5);
  auto emb = buf((size_t)40 * 64), hid = buf((size_t)40 * 12 * 64);
  OK(svg_minilm_encode(ctx, ids2.data(), lens.data(), 40, 12, emb.data(), hid.data(), nullptr));       // 480 rows: two passes
  EXPECT_ERR(svg_minilm_encode(ctx, ids2.data(), lens.data(), 1, 200, emb.data(), nullptr, nullptr));
}

static void i3d_and_fvd() {
  struct M { const char* n; int cin; int oc[6]; };
  const M mixed[] = {{"Mixed_3b", 192, {64, 96, 128, 16, 32, 32}},   {"Mixed_3c", 256, {128, 128, 192, 32, 96, 64}},   {"Mixed_4b", 480, {192, 96, 208, 16, 48, 64}},
                     {"Mixed_4c", 512, {160, 112, 224, 24, 64, 64}}, {"Mixed_4d", 512, {128, 128, 256, 24, 64, 64}},   {"Mixed_4e", 512, {112, 144, 288, 32, 64, 64}},
                     {"Mixed_4f", 528, {256, 160, 320, 32, 128, 128}}, {"Mixed_5b", 832, {256, 160, 320, 32, 128, 128}}, {"Mixed_5c", 832, {384, 192, 384, 48, 128, 128}}};
  OK(svg_model_configure(ctx, SVG_I3D, "num_classes=400"));
  const int I = SVG_I3D;
  auto unit = [&](const std::string& p, int cin, int cout, int k, bool bn = true) {
    load(I, p + ".conv3d.weight", {cout, cin, k, k, k});
    if (bn) for (const char* n : {"weight", "bias", "running_mean", "running_var"}) load(I, p + ".bn." + n, {cout}, 1.f);
    else load(I, p + ".conv3d.bias", {cout});
  };
  unit("Conv3d_1a_7x7", 3, 64, 7); unit("Conv3d_2b_1x1", 64, 64, 1); unit("Conv3d_2c_3x3", 64, 192, 3);
  for (const M& m : mixed) {
    const std::string p = m.n;
    unit(p + ".b0", m.cin, m.oc[0], 1); unit(p + ".b1a", m.cin, m.oc[1], 1); unit(p + ".b1b", m.oc[1], m.oc[2], 3);
    unit(p + ".b2a", m.cin, m.oc[3], 1); unit(p + ".b2b", m.oc[3], m.oc[4], 3); unit(p + ".b3b", m.cin, m.oc[5], 1);
  }
  unit("logits", 1024, 400, 1, false);
  OK(svg_finalize(ctx, I, nullptr));
  std::vector<uint8_t> vid((size_t)2 * 16 * 30 * 40 * 3, 128);
  auto lg = buf(2 * 400);
  OK(svg_fvd_logits(ctx, vid.data(), 2, 16, 30, 40, lg.data(), nullptr));
  std::vector<uint8_t> vid8((size_t)1 * 8 * 30 * 40 * 3, 128);
  EXPECT_ERR(svg_fvd_logits(ctx, vid8.data(), 1, 8, 30, 40, lg.data(), nullptr));            // 8 frames: no [2,7,7] window
  auto a = buf(20 * 16), b = buf(30 * 16);
  double fd = 0;
  OK(svg_frechet_distance(ctx, a.data(), 20, b.data(), 30, 16, &fd, nullptr));
  EXPECT_ERR(svg_frechet_distance(ctx, a.data(), 1, b.data(), 30, 16, &fd, nullptr));
}

static void ops() {
  const int M = 300, N = 64, K = 128;
  std::vector<uint16_t> A((size_t)M * K, 0x3c00), W((size_t)N * K, 0x3c00), R((size_t)M * N, 0), Cc((size_t)M * N);
  auto bias = buf(N);
  OK(svg_op_gemm(ctx, A.data(), W.data(), bias.data(), R.data(), Cc.data(), M, N, K, 0, 0, nullptr));
  OK(svg_op_gemm_f16(ctx, A.data(), W.data(), bias.data(), R.data(), Cc.data(), M, N, K, 1, 0, nullptr));
  std::vector<uint16_t> big((size_t)20000 * 320, 0), wb((size_t)320 * 320, 0), ob((size_t)20000 * 320);
  auto b2 = buf(320);
  OK(svg_op_gemm(ctx, big.data(), wb.data(), b2.data(), big.data(), ob.data(), 20000, 320, 320, 0, 0, nullptr));    // the weight-stationary selection
  EXPECT_ERR(svg_op_gemm(ctx, A.data(), W.data(), nullptr, nullptr, Cc.data(), M, 3, K, 0, 0, nullptr));           // N % 4
  std::vector<uint16_t> x((size_t)2 * 32 * 32 * 64, 0), y((size_t)2 * 32 * 32 * 64), y2 = y;
  auto w = buf((size_t)64 * 64 * 9), g = buf(64, 1.f), be = buf(64, 0.f);
  int used = 0;
  OK(svg_op_conv3x3_gn(ctx, x.data(), w.data(), b2.data(), g.data(), be.data(), y.data(), y2.data(), 2, 32, 32, 64, 64, 32, 1e-5f, 1, &used, nullptr));
  OK(svg_op_groupnorm_f16(ctx, x.data(), g.data(), be.data(), y.data(), 2, 1024, 64, 32, 1e-5f, 1, nullptr));
}

int main() {
  if (svg_create(0, &ctx) != 0) { fprintf(stderr, "svg_create: %s\n", svg_last_error(nullptr)); return 1; }
  EXPECT_ERR(svg_finalize(ctx, SVG_UNET, nullptr));                       // nothing loaded yet
  transformer(0);
  transformer(384);
  transformer(0, 128, 128, 256);
  transformer(384, 128, 128, 256);
  transformer(0, 256, 256, 256);                   // d = K = 256: one clip (6 rows) takes the small-row walk (xf_forward_walk_small: whole-K stage table)
  vae(0); vae(1);
  unet(0, 0); unet(1, 0); unet(0, 1);
  text_towers();
  i3d_and_fvd();
  ops();
  // a missing tensor is reported by name, and the slot can be configured again afterwards
  OK(svg_model_configure(ctx, SVG_VAE, "block_out=64,128;layers=1;groups=32;latent=4"));
  load_conv(SVG_VAE, "encoder.conv_in", 3, 64);
  EXPECT_ERR(svg_finalize(ctx, SVG_VAE, nullptr));
  if (!strstr(svg_last_error(ctx), "missing weight")) { fprintf(stderr, "unexpected message: %s\n", svg_last_error(ctx)); return 4; }
  vae(0);
  char rep[1 << 16];
  OK(svg_prof_enable(ctx, 2)); OK(svg_prof_reset(ctx));
  unet(0, 0);
  OK(svg_prof_report(ctx, rep, sizeof(rep)));
  svg_prof_enable(ctx, 0);
  printf("workspace %lld bytes, %s, %s / %s\n", (long long)svg_workspace_bytes(ctx), svg_version(), svg_model_dtype(ctx, SVG_UNET), svg_model_dtype(ctx, SVG_VAE));
  svg_destroy(ctx);
  puts("host-sanitize: OK");
  return 0;
}
