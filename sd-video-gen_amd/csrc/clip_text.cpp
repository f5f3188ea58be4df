// CLIP text tower (openai/clip-vit-large-patch14 text model; the reference calls transformers.CLIPTextModel at
// utils/sd_utils.py:60 and runs it at :84 / :91 in fp32, outside the autocast block).  transformers is a third-party
// dependency (pinned 4.21.0, environment.yml:157) whose source is not under /root/reference; the algorithm restated here is
// its published modeling_clip.CLIPTextTransformer:
//   x = token_embedding[ids] + position_embedding[0..T)
//   per layer:  h = LN1(x); q,k,v = Linear(h); attn = softmax(q k^T / sqrt(hd) + causal mask) v; x = x + out_proj(attn)
//               h = LN2(x); x = x + fc2(quick_gelu(fc1(h)))           quick_gelu(u) = u * sigmoid(1.702 u)
//   last_hidden_state = final_layer_norm(x)                           (no padding mask: the reference passes input_ids only)
// Rows M = B*T <= 336 per pass (the f32 weight-streaming GEMM's limit): prompts are processed four at a time.
#include "models.h"
#include "../../include/svg_hip.h"

void ClipTextModel::configure(const char* kv) {
  auto m = parse_kv(kv);
  auto geti = [&](const char* k, int& dst) { if (m.count(k)) dst = (int)m[k][0]; };
  vocab = 49408; d_model = 768; heads = 12; layers = 12; ffn = 3072; max_pos = 77;
  geti("vocab", vocab); geti("d_model", d_model); geti("heads", heads); geti("layers", layers); geti("ffn", ffn); geti("max_pos", max_pos);
  ready = false;
}

static std::string L(int i, const char* rest) { return "encoder.layers." + std::to_string(i) + "." + rest; }

void ClipTextModel::finalize(svg_ctx* ctx, int64_t* n_params) {
  const int64_t d = d_model;
  SVG_CHECK(d % heads == 0 && d / heads <= 64 && d % 8 == 0 && ffn % 8 == 0, "clip: d_model %d / heads %d / ffn %d unsupported (head dim <= 64)", d_model, heads, ffn);
  SVG_CHECK(max_pos >= 1 && max_pos <= 128, "clip: max_pos %d unsupported", max_pos);
  ws.get("embeddings.token_embedding.weight", {vocab, d});
  ws.get("embeddings.position_embedding.weight", {max_pos, d});
  ws.get("final_layer_norm.weight", {d}); ws.get("final_layer_norm.bias", {d});
  qkv_w.assign(layers, nullptr); qkv_b.assign(layers, nullptr);
  for (int i = 0; i < layers; ++i) {
    for (const char* p : {"self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.out_proj"}) {
      ws.get(L(i, p) + ".weight", {d, d}); ws.get(L(i, p) + ".bias", {d});
    }
    ws.get(L(i, "mlp.fc1.weight"), {ffn, d}); ws.get(L(i, "mlp.fc1.bias"), {ffn});
    ws.get(L(i, "mlp.fc2.weight"), {d, ffn}); ws.get(L(i, "mlp.fc2.bias"), {d});
    for (const char* p : {"layer_norm1", "layer_norm2"}) { ws.get(L(i, p) + ".weight", {d}); ws.get(L(i, p) + ".bias", {d}); }
    qkv_w[i] = (float*)ctx->dalloc(3 * d * d * sizeof(float));
    qkv_b[i] = (float*)ctx->dalloc(3 * d * sizeof(float));
    int j = 0;
    for (const char* p : {"self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"}) {
      HIP_OK(hipMemcpy(qkv_w[i] + (int64_t)j * d * d, ws.get(L(i, p) + ".weight").f32, d * d * sizeof(float), hipMemcpyDeviceToDevice));
      HIP_OK(hipMemcpy(qkv_b[i] + (int64_t)j * d, ws.get(L(i, p) + ".bias").f32, d * sizeof(float), hipMemcpyDeviceToDevice));
      ++j;
    }
    for (const char* p : {"self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"}) ws.release(L(i, p) + ".weight");
  }
  int64_t n = 0;
  for (auto& kv : ws.map)
    if (kv.first.find("position_ids") == std::string::npos) n += kv.second.numel;
  if (n_params) *n_params = n;
  ready = true;
}

void ClipTextModel::forward(svg_ctx* ctx, const int32_t* ids, int B, int T, float* out, hipStream_t s) {
  SVG_CHECK(ready, "clip: svg_finalize has not been called");
  SVG_CHECK(B >= 1 && T >= 1 && T <= max_pos, "clip: B=%d T=%d (max_position_embeddings %d)", B, T, max_pos);
  const int d = d_model, hd = d / heads;
  const int Bc = std::max(1, 336 / T);
  auto W = [&](const std::string& n) { return ws.get(n).f32; };
  run_planned(ctx, [&]() {
    for (int b0 = 0; b0 < B; b0 += Bc) {
      const int nb = std::min(Bc, B - b0), M = nb * T;
      ctx->arena.push();
      float* x = ctx->arena.get<float>((int64_t)M * d);
      float* h = ctx->arena.get<float>((int64_t)M * d);
      float* qkv = ctx->arena.get<float>((int64_t)M * 3 * d);
      float* att = ctx->arena.get<float>((int64_t)M * d);
      float* x2 = ctx->arena.get<float>((int64_t)M * d);
      float* f = ctx->arena.get<float>((int64_t)M * ffn);
      if (SVG_LAUNCHING(ctx)) {
        ProfScope ps(ctx, PK_XF_MISC, s, 0, 0);
        xf_embed_tokens(ids + (int64_t)b0 * T, W("embeddings.token_embedding.weight"), W("embeddings.position_embedding.weight"), x, M, T, d, vocab, s);
      }
      for (int i = 0; i < layers; ++i) {
        if (SVG_LAUNCHING(ctx)) { ProfScope ps(ctx, PK_XF_MISC, s, 0, 0); xf_add_ln(x, nullptr, W(L(i, "layer_norm1.weight")), W(L(i, "layer_norm1.bias")), h, M, d, 1e-5f, s); }
        xf_gemm(ctx, h, qkv_w[i], qkv_b[i], qkv, M, 3 * d, d, 0, s);
        if (SVG_LAUNCHING(ctx)) { ProfScope ps(ctx, PK_XF_MISC, s, 0, 0); xf_attention_causal(qkv, att, nb, T, heads, hd, s); }
        xf_gemm(ctx, att, W(L(i, "self_attn.out_proj.weight")), W(L(i, "self_attn.out_proj.bias")), x2, M, d, d, 0, s, x);          // x2 = x + out_proj(attn)
        if (SVG_LAUNCHING(ctx)) { ProfScope ps(ctx, PK_XF_MISC, s, 0, 0); xf_add_ln(x2, nullptr, W(L(i, "layer_norm2.weight")), W(L(i, "layer_norm2.bias")), h, M, d, 1e-5f, s); }
        xf_gemm(ctx, h, W(L(i, "mlp.fc1.weight")), W(L(i, "mlp.fc1.bias")), f, M, ffn, d, 0, s);
        xf_gemm(ctx, f, W(L(i, "mlp.fc2.weight")), W(L(i, "mlp.fc2.bias")), x, M, d, ffn, /*quick_gelu on the input*/ 2, s, x2);     // x = x2 + fc2(quick_gelu(fc1 h))
      }
      if (SVG_LAUNCHING(ctx)) {
        ProfScope ps(ctx, PK_XF_MISC, s, 0, 0);
        xf_add_ln(x, nullptr, W("final_layer_norm.weight"), W("final_layer_norm.bias"), out + (int64_t)b0 * T * d, M, d, 1e-5f, s);
      }
      ctx->arena.pop();
    }
  });
}

extern "C" int svg_clip_text_forward(svg_ctx* ctx, const int32_t* input_ids, int B, int T, float* out, void* stream) {
  try {
    SVG_CHECK(ctx && ctx->clip, "clip: model not configured");
    ctx->clip->forward(ctx, input_ids, B, T, out, (hipStream_t)stream);
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}
