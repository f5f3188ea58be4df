// MiniLM sentence encoder: what SentenceTransformer('sentence-transformers/all-MiniLM-L6-v2').encode(cls_list) computes for the
// text-conditioned latent Transformer (reference: models/transformer_text.py:12,82-83).  sentence-transformers and transformers
// are third-party dependencies whose sources are not under /root/reference; the algorithm restated here is the published one:
//   transformers modeling_bert.BertModel (post-LayerNorm encoder):
//     x = LayerNorm_1e-12(word_embeddings[ids] + position_embeddings[0..T) + token_type_embeddings[0])
//     per layer:  q,k,v = Linear(x); a = softmax(q k^T / sqrt(hd) + padding mask) v
//                 x = LayerNorm(x + attention.output.dense(a));  x = LayerNorm(x + output.dense(gelu_erf(intermediate.dense(x))))
//   sentence-transformers Pooling(mean, attention-mask weighted) and Normalize (L2).
// Rows M = B*T <= 336 per pass of the f32 weight-streaming GEMM: sentences are processed in groups.
#include "models.h"
#include "../../include/svg_hip.h"

void MiniLmModel::configure(const char* kv) {
  auto m = parse_kv(kv);
  auto geti = [&](const char* k, int& dst) { if (m.count(k)) dst = (int)m[k][0]; };
  vocab = 30522; d_model = 384; heads = 12; layers = 6; ffn = 1536; max_pos = 512;
  geti("vocab", vocab); geti("d_model", d_model); geti("heads", heads); geti("layers", layers); geti("ffn", ffn); geti("max_pos", max_pos);
  ready = false;
}

static std::string LN(int i, const char* rest) { return "encoder.layer." + std::to_string(i) + "." + rest; }

void MiniLmModel::finalize(svg_ctx* ctx, int64_t* n_params) {
  const int64_t d = d_model;
  SVG_CHECK(d % heads == 0 && d / heads <= 64 && d % 8 == 0 && ffn % 8 == 0, "minilm: d_model %d / heads %d / ffn %d unsupported (head dim <= 64)", d_model, heads, ffn);
  ws.get("embeddings.word_embeddings.weight", {vocab, d});
  ws.get("embeddings.position_embeddings.weight", {max_pos, d});
  SVG_CHECK(ws.get("embeddings.token_type_embeddings.weight").numel % d == 0, "minilm: token_type_embeddings width");
  ws.get("embeddings.LayerNorm.weight", {d}); ws.get("embeddings.LayerNorm.bias", {d});
  qkv_w.assign(layers, nullptr); qkv_b.assign(layers, nullptr);
  for (int i = 0; i < layers; ++i) {
    for (const char* p : {"attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"}) {
      ws.get(LN(i, p) + ".weight", {d, d}); ws.get(LN(i, p) + ".bias", {d});
    }
    ws.get(LN(i, "intermediate.dense.weight"), {ffn, d}); ws.get(LN(i, "intermediate.dense.bias"), {ffn});
    ws.get(LN(i, "output.dense.weight"), {d, ffn}); ws.get(LN(i, "output.dense.bias"), {d});
    for (const char* p : {"attention.output.LayerNorm", "output.LayerNorm"}) { ws.get(LN(i, p) + ".weight", {d}); ws.get(LN(i, p) + ".bias", {d}); }
    qkv_w[i] = (float*)ctx->dalloc(3 * d * d * sizeof(float));
    qkv_b[i] = (float*)ctx->dalloc(3 * d * sizeof(float));
    int j = 0;
    for (const char* p : {"attention.self.query", "attention.self.key", "attention.self.value"}) {
      HIP_OK(hipMemcpy(qkv_w[i] + (int64_t)j * d * d, ws.get(LN(i, p) + ".weight").f32, d * d * sizeof(float), hipMemcpyDeviceToDevice));
      HIP_OK(hipMemcpy(qkv_b[i] + (int64_t)j * d, ws.get(LN(i, p) + ".bias").f32, d * sizeof(float), hipMemcpyDeviceToDevice));
      ++j;
    }
    for (const char* p : {"attention.self.query", "attention.self.key", "attention.self.value"}) ws.release(LN(i, p) + ".weight");
  }
  int64_t n = 0;
  for (auto& kv : ws.map) n += kv.second.numel;      // the pooler (unused by mean pooling) counts when it was handed over, as in the checkpoint
  if (n_params) *n_params = n;
  ready = true;
}

void MiniLmModel::encode(svg_ctx* ctx, const int32_t* ids, const int32_t* lens, int B, int T, float* out, float* hidden, hipStream_t s) {
  SVG_CHECK(ready, "minilm: svg_finalize has not been called");
  SVG_CHECK(B >= 1 && T >= 1 && T <= max_pos && T <= 128, "minilm: B=%d T=%d (sequence length <= min(128, max_position_embeddings %d))", B, T, max_pos);
  const int d = d_model, hd = d / heads;
  const int Bc = std::max(1, 336 / T);
  auto W = [&](const std::string& n) { return ws.get(n).f32; };
  run_planned(ctx, [&]() {
    for (int b0 = 0; b0 < B; b0 += Bc) {
      const int nb = std::min(Bc, B - b0), M = nb * T;
      ctx->arena.push();
      float* e = ctx->arena.get<float>((int64_t)M * d);
      float* x = ctx->arena.get<float>((int64_t)M * d);
      float* x2 = ctx->arena.get<float>((int64_t)M * d);
      float* qkv = ctx->arena.get<float>((int64_t)M * 3 * d);
      float* att = ctx->arena.get<float>((int64_t)M * d);
      float* t = ctx->arena.get<float>((int64_t)M * d);
      float* f = ctx->arena.get<float>((int64_t)M * ffn);
      if (SVG_LAUNCHING(ctx)) {
        ProfScope ps(ctx, PK_XF_MISC, s, 0, 0);
        xf_embed_bert(ids + (int64_t)b0 * T, W("embeddings.word_embeddings.weight"), W("embeddings.position_embeddings.weight"),
                      W("embeddings.token_type_embeddings.weight"), e, M, T, d, vocab, s);
        xf_add_ln(e, nullptr, W("embeddings.LayerNorm.weight"), W("embeddings.LayerNorm.bias"), x, M, d, 1e-12f, s);
      }
      for (int i = 0; i < layers; ++i) {
        xf_gemm(ctx, x, qkv_w[i], qkv_b[i], qkv, M, 3 * d, d, 0, s);
        if (SVG_LAUNCHING(ctx)) { ProfScope ps(ctx, PK_XF_MISC, s, 0, 0); xf_attention_padded(qkv, lens + b0, att, nb, T, heads, hd, s); }
        xf_gemm(ctx, att, W(LN(i, "attention.output.dense.weight")), W(LN(i, "attention.output.dense.bias")), t, M, d, d, 0, s);
        if (SVG_LAUNCHING(ctx)) {
          ProfScope ps(ctx, PK_XF_MISC, s, 0, 0);
          xf_add_ln(x, t, W(LN(i, "attention.output.LayerNorm.weight")), W(LN(i, "attention.output.LayerNorm.bias")), x2, M, d, 1e-12f, s);
        }
        xf_gemm(ctx, x2, W(LN(i, "intermediate.dense.weight")), W(LN(i, "intermediate.dense.bias")), f, M, ffn, d, 0, s);
        xf_gemm(ctx, f, W(LN(i, "output.dense.weight")), W(LN(i, "output.dense.bias")), t, M, d, ffn, /*exact GELU on the input*/ 3, s);
        if (SVG_LAUNCHING(ctx)) {
          ProfScope ps(ctx, PK_XF_MISC, s, 0, 0);
          xf_add_ln(x2, t, W(LN(i, "output.LayerNorm.weight")), W(LN(i, "output.LayerNorm.bias")), x, M, d, 1e-12f, s);
        }
      }
      if (SVG_LAUNCHING(ctx)) {
        ProfScope ps(ctx, PK_XF_MISC, s, 0, 0);
        xf_mean_pool_norm(x, lens + b0, out + (int64_t)b0 * d, nb, T, d, s);
        if (hidden) HIP_OK(hipMemcpyAsync(hidden + (int64_t)b0 * T * d, x, (size_t)M * d * sizeof(float), hipMemcpyDeviceToDevice, s));
      }
      ctx->arena.pop();
    }
  });
}

extern "C" int svg_minilm_encode(svg_ctx* ctx, const int32_t* input_ids, const int32_t* lengths, int B, int T, float* out, float* hidden,
                                 void* stream) {
  try {
    SVG_CHECK(ctx && ctx->minilm, "minilm: model not configured");
    SVG_CHECK(input_ids && lengths && out, "minilm: null argument");
    ctx->minilm->encode(ctx, input_ids, lengths, B, T, out, hidden, (hipStream_t)stream);
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}
