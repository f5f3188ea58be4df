// Latent-sequence Transformer graph (reference: models/transformer.py:47-68 over torch.nn.Transformer
// defaults — post-norm, ReLU, final encoder/decoder LayerNorm, sequence-first).  f32 throughout.
#include "models.h"
#include "xf_walk.h"
#include "../../include/svg_hip.h"

void XfModel::configure(const char* kv) {
  auto m = parse_kv(kv);
  auto geti = [&](const char* k, int& dst) { if (m.count(k)) dst = (int)m[k][0]; };
  // a configure call describes the whole model: keys it leaves out go back to their defaults (a context that held a
  // text-conditioned model must not keep its text_dim for the next, plain one)
  d_lat = 0; d_model = 0; heads = 8; enc_layers = 0; dec_layers = 0; ffn = 2048; text_dim = 0;
  geti("d_lat", d_lat); geti("d_model", d_model); geti("heads", heads);
  geti("enc_layers", enc_layers); geti("dec_layers", dec_layers); geti("ffn", ffn); geti("text_dim", text_dim);
  ready = false;
}

void XfModel::finalize(svg_ctx* ctx, int64_t* n_params) {
  SVG_CHECK(d_lat > 0 && d_model > 0 && heads > 0, "transformer: configure d_lat/d_model/heads first");
  SVG_CHECK(d_model % heads == 0 && d_model % 16 == 0 && d_lat % 16 == 0 && ffn % 16 == 0,
            "transformer: d_model %d / d_lat %d / ffn %d must be multiples of 16 (and d_model of heads)", d_model, d_lat, ffn);
  const int64_t d = d_model;
  SVG_CHECK(text_dim >= 0 && text_dim < d_model, "transformer: text_dim %d out of range", text_dim);
  const char* emb_name = text_dim ? "project_image_embedding" : "embedding";     // transformer_text.py:60 vs transformer.py:37
  ws.get(std::string(emb_name) + ".weight", {d - text_dim, d_lat}); ws.get(std::string(emb_name) + ".bias", {d - text_dim});
  ws.get("out.weight", {d_lat, d}); ws.get("out.bias", {d_lat});
  auto check_mha = [&](const std::string& p) {
    ws.get(p + "in_proj_weight", {3 * d, d}); ws.get(p + "in_proj_bias", {3 * d});
    ws.get(p + "out_proj.weight", {d, d}); ws.get(p + "out_proj.bias", {d});
  };
  auto check_ffn_norms = [&](const std::string& p, int nnorm) {
    ws.get(p + "linear1.weight", {ffn, d}); ws.get(p + "linear1.bias", {ffn});
    ws.get(p + "linear2.weight", {d, ffn}); ws.get(p + "linear2.bias", {d});
    for (int i = 1; i <= nnorm; ++i) {
      ws.get(p + "norm" + std::to_string(i) + ".weight", {d});
      ws.get(p + "norm" + std::to_string(i) + ".bias", {d});
    }
  };
  for (int i = 0; i < enc_layers; ++i) {
    std::string p = "transformer.encoder.layers." + std::to_string(i) + ".";
    check_mha(p + "self_attn."); check_ffn_norms(p, 2);
  }
  for (int i = 0; i < dec_layers; ++i) {
    std::string p = "transformer.decoder.layers." + std::to_string(i) + ".";
    check_mha(p + "self_attn."); check_mha(p + "multihead_attn."); check_ffn_norms(p, 3);
  }
  ws.get("transformer.encoder.norm.weight", {d}); ws.get("transformer.encoder.norm.bias", {d});
  ws.get("transformer.decoder.norm.weight", {d}); ws.get("transformer.decoder.norm.bias", {d});
  // positional table (models/positional_encoding.py:16-30): use the state_dict buffer when it was handed
  // over, else build it — sin/cos in double, rounded to f32.
  if (!pe || pe_d != d) { pe = (float*)ctx->dalloc(64 * d * sizeof(float)); pe_d = (int)d; }
  if (ws.has("positional_encoder.pos_encoding")) {
    const Weight& w = ws.get("positional_encoder.pos_encoding");
    SVG_CHECK(w.numel == 64 * d, "positional_encoder.pos_encoding has %lld elements", (long long)w.numel);
    HIP_OK(hipMemcpy(pe, w.f32, 64 * d * sizeof(float), hipMemcpyDeviceToDevice));
  } else {
    std::vector<float> h(64 * d);
    for (int pos = 0; pos < 64; ++pos)
      for (int i = 0; i < d; i += 2) {
        float div = expf((float)i * (-logf(10000.0f)) / (float)d);
        h[pos * d + i] = sinf((float)pos * div);
        if (i + 1 < d) h[pos * d + i + 1] = cosf((float)pos * div);
      }
    HIP_OK(hipMemcpy(pe, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  if (!iota) {
    iota = (int32_t*)ctx->dalloc(64 * sizeof(int32_t));
    int32_t h[64];
    for (int i = 0; i < 64; ++i) h[i] = i;
    HIP_OK(hipMemcpy(iota, h, sizeof(h), hipMemcpyHostToDevice));
  }
  int64_t n = 0;
  for (auto& kv : ws.map)
    if (kv.first != "positional_encoder.pos_encoding") n += kv.second.numel;
  if (n_params) *n_params = n;
  auto P = [&](const std::string& name) { return (const float*)ws.get(name).f32; };
  auto layer = [&](const std::string& p, bool dec) {
    LayerW w{};
    w.in_w = P(p + "self_attn.in_proj_weight"); w.in_b = P(p + "self_attn.in_proj_bias");
    w.out_w = P(p + "self_attn.out_proj.weight"); w.out_b = P(p + "self_attn.out_proj.bias");
    if (dec) {
      w.cin_w = P(p + "multihead_attn.in_proj_weight"); w.cin_b = P(p + "multihead_attn.in_proj_bias");
      w.cout_w = P(p + "multihead_attn.out_proj.weight"); w.cout_b = P(p + "multihead_attn.out_proj.bias");
    }
    w.l1_w = P(p + "linear1.weight"); w.l1_b = P(p + "linear1.bias"); w.l2_w = P(p + "linear2.weight"); w.l2_b = P(p + "linear2.bias");
    for (int i = 0; i < (dec ? 3 : 2); ++i) {
      w.n_w[i] = P(p + "norm" + std::to_string(i + 1) + ".weight"); w.n_b[i] = P(p + "norm" + std::to_string(i + 1) + ".bias");
    }
    return w;
  };
  enc_w.clear(); dec_w.clear();
  for (int i = 0; i < enc_layers; ++i) enc_w.push_back(layer("transformer.encoder.layers." + std::to_string(i) + ".", false));
  for (int i = 0; i < dec_layers; ++i) dec_w.push_back(layer("transformer.decoder.layers." + std::to_string(i) + ".", true));
  emb_w = P(std::string(emb_name) + ".weight"); emb_b = P(std::string(emb_name) + ".bias");
  out_w = P("out.weight"); out_b = P("out.bias");
  encn_w = P("transformer.encoder.norm.weight"); encn_b = P("transformer.encoder.norm.bias");
  decn_w = P("transformer.decoder.norm.weight"); decn_b = P("transformer.decoder.norm.bias");
  ready = true;
}

namespace {
struct XfRun {
  svg_ctx* ctx; XfModel* m; hipStream_t s; int B;
  const float* text = nullptr;   // (B, text_dim) for the text-conditioned variant
  const float* W(const std::string& n) { return m->ws.get(n).f32; }
  float* gemm(const float* X, const std::string& w, const std::string& b, int M, int N, int K, int relu_in = 0,
              int64_t woff = 0, int64_t boff = 0) {
    float* Y = ctx->arena.get<float>((int64_t)M * N);
    xf_gemm(ctx, X, W(w) + woff, W(b) + boff, Y, M, N, K, relu_in, s);
    return Y;
  }
  float* add_ln(const float* x, const float* r, const std::string& p, int M) {
    float* y = ctx->arena.get<float>((int64_t)M * m->d_model);
    if (SVG_LAUNCHING(ctx)) {
      ProfScope ps(ctx, PK_XF_MISC, s, 0, 12.0 * M * m->d_model);
      xf_add_ln(x, r, W(p + "weight"), W(p + "bias"), y, M, m->d_model, 1e-5f, s);
    }
    return y;
  }
  float* mha(const std::string& p, const float* xq, int Tq, const float* xkv, int Tk, const float* mask, bool self,
             const float* kpad = nullptr) {
    const int d = m->d_model, hd = d / m->heads;
    float* o = ctx->arena.get<float>((int64_t)Tq * B * d);
    if (self) {
      float* qkv = gemm(xq, p + "in_proj_weight", p + "in_proj_bias", Tq * B, 3 * d, d);
      if (SVG_LAUNCHING(ctx)) {
        ProfScope ps(ctx, PK_XF_MISC, s, 0, 0);
        xf_attention(qkv, 3 * d, qkv + d, qkv + 2 * d, 3 * d, mask, o, Tq, Tk, B, m->heads, hd, s, kpad);
      }
    } else {
      float* q = gemm(xq, p + "in_proj_weight", p + "in_proj_bias", Tq * B, d, d);
      float* kv = gemm(xkv, p + "in_proj_weight", p + "in_proj_bias", Tk * B, 2 * d, d, 0, (int64_t)d * d, d);
      if (SVG_LAUNCHING(ctx)) {
        ProfScope ps(ctx, PK_XF_MISC, s, 0, 0);
        xf_attention(q, d, kv, kv + d, 2 * d, mask, o, Tq, Tk, B, m->heads, hd, s);
      }
    }
    return gemm(o, p + "out_proj.weight", p + "out_proj.bias", Tq * B, d, d);
  }
  float* ffn(const std::string& p, const float* x, int M) {
    float* h = gemm(x, p + "linear1.weight", p + "linear1.bias", M, m->ffn, m->d_model);
    return gemm(h, p + "linear2.weight", p + "linear2.bias", M, m->d_model, m->ffn, /*relu_in=*/1);
  }
  float* embed(const float* x, int T, const int32_t* pe_row) {
    const int d = m->d_model, d_img = d - m->text_dim;
    const std::string en = m->text_dim ? "project_image_embedding" : "embedding";
    float* e = gemm(x, en + ".weight", en + ".bias", B * T, d_img, m->d_lat);
    float* y = ctx->arena.get<float>((int64_t)B * T * d);
    if (SVG_LAUNCHING(ctx)) {
      ProfScope ps(ctx, PK_XF_MISC, s, 0, 8.0 * B * T * d);
      xf_embed_post(e, m->pe, pe_row, text, m->text_dim, y, B, T, d, sqrtf((float)d), s);
    }
    return y;
  }
};
}  // namespace

// One chunk of batch rows (B*max(Ts,Tt) <= 336: the rows one pass of the weight stream serves).  pe_row must be non-null here.
static void xf_forward_chunk(svg_ctx* ctx, XfModel* m, const float* src, const float* tgt, int B, int Ts, int Tt,
                             const float* mask, const int32_t* pe_row, float* out_tb, hipStream_t s, const float* text,
                             const float* src_pad, const float* tgt_pad) {
  XfRun r{ctx, m, s, B};
  r.text = text;
  float* xs = r.embed(src, Ts, pe_row);
  float* xt = (tgt == src && Ts == Tt) ? xs : r.embed(tgt, Tt, pe_row);
  // nn.Transformer: src_key_padding_mask -> encoder self-attention keys, tgt_key_padding_mask -> decoder self-attention keys;
  // the cross-attention takes none (memory_key_padding_mask is not passed at models/transformer.py:64)
  const int Ms = Ts * B, Mt = Tt * B;
  for (int i = 0; i < m->enc_layers; ++i) {
    std::string p = "transformer.encoder.layers." + std::to_string(i) + ".";
    xs = r.add_ln(xs, r.mha(p + "self_attn.", xs, Ts, xs, Ts, nullptr, true, src_pad), p + "norm1.", Ms);
    xs = r.add_ln(xs, r.ffn(p, xs, Ms), p + "norm2.", Ms);
  }
  float* mem = r.add_ln(xs, nullptr, "transformer.encoder.norm.", Ms);
  for (int i = 0; i < m->dec_layers; ++i) {
    std::string p = "transformer.decoder.layers." + std::to_string(i) + ".";
    xt = r.add_ln(xt, r.mha(p + "self_attn.", xt, Tt, xt, Tt, mask, true, tgt_pad), p + "norm1.", Mt);
    xt = r.add_ln(xt, r.mha(p + "multihead_attn.", xt, Tt, mem, Ts, nullptr, false), p + "norm2.", Mt);
    xt = r.add_ln(xt, r.ffn(p, xt, Mt), p + "norm3.", Mt);
  }
  xt = r.add_ln(xt, nullptr, "transformer.decoder.norm.", Mt);
  xf_gemm(ctx, xt, r.W("out.weight"), r.W("out.bias"), out_tb, Mt, m->d_lat, m->d_model, 0, s);
}

// ---- the same chunk as ONE launch (xf_walk.hip): the stage table --------------------------------------------------------------------
// Stages per encoder layer: GEMM in_proj | add slabs + bias | attention | GEMM out_proj | add slabs + bias + residual + LayerNorm |
// GEMM linear1 | add slabs + bias, ReLU | GEMM linear2 | add + LayerNorm.  A decoder layer has the cross-attention block in between; its
// K / V projection of the encoder memory shares the stage of the self-attention in_proj (no barrier of its own).  The final encoder /
// decoder LayerNorm rides on the last layer's add + LayerNorm stage (Y2).
static bool xf_walk_usable(const XfModel* m, int B, int Ts, int Tt, hipStream_t s) {
  // off for this device (SVG_XF_WALK=0, ranks sharing it, an earlier give-up, a grid the device cannot hold) or a stream under capture:
  // residency and the knobs are facts established at svg_create (xf_walk.hip: xf_walk_init_device), not looked up per forward
  if (!xf_walk_enabled(s)) return false;
  const int d = m->d_model, d_img = d - m->text_dim, hd = d / m->heads;
  // the stage table: 3 (+2 for a separate target) embedding stages, 9 per encoder layer, 16 per decoder layer, 2 for the output projection,
  // 1 each for an empty encoder / decoder stack
  if (5 + 9 * m->enc_layers + 16 * m->dec_layers + 2 + 2 > kWalkMaxOps) return false;
  const int rows = B * std::max(Ts, Tt);
  if (!(xf_walk_gemm_ok(d, d) && xf_walk_gemm_ok(m->ffn, d) && xf_walk_gemm_ok(d, m->ffn) && xf_walk_gemm_ok(d_img, m->d_lat) &&
        xf_walk_gemm_ok(m->d_lat, d)))
    return false;
  if (d > 3072 || hd % 4 || m->text_dim % 4) return false;
  return xf_walk_available(rows, xf_walk_lds_bytes(rows, std::max(Ts, Tt), std::max(Ts, Tt), hd));
}

static void xf_forward_walk(svg_ctx* ctx, XfModel* m, const float* src, const float* tgt, int B, int Ts, int Tt, const float* mask,
                            const int32_t* pe_row, float* out_tb, hipStream_t s, const float* text, const float* src_pad,
                            const float* tgt_pad) {
  const int d = m->d_model, d_img = d - m->text_dim, ffn = m->ffn, d_lat = m->d_lat, heads = m->heads, hd = d / heads;
  const int Ms = Ts * B, Mt = Tt * B, Mx = std::max(Ms, Mt);
  const bool same = (tgt == src && Ts == Tt);
  auto slabs = [&](int M, int N, int K) { return (int64_t)(K / 128) * M * N; };
  int64_t sa = std::max(slabs(Mx, 3 * d, d), std::max(slabs(Mx, ffn, d), slabs(Mx, d, ffn)));
  sa = std::max(sa, std::max(slabs(Mx, d_img, d_lat), slabs(Mx, d_lat, d)));
  const int64_t sb = std::max(slabs(Ms, 2 * d, d), slabs(Mt, d_img, d_lat));
  float* slabA = ctx->arena.get<float>(sa);
  float* slabB = ctx->arena.get<float>(sb);
  // row buffers with fixed roles (a stage never writes a buffer another workgroup still reads in the same stage): the embeddings; the
  // encoder's norm1 / norm2 results; the memory; the decoder's norm1 / norm2 / norm3 results
  float* xs_e = ctx->arena.get<float>((int64_t)Ms * d);
  float* xt_e = same ? xs_e : ctx->arena.get<float>((int64_t)Mt * d);
  float* e1 = ctx->arena.get<float>((int64_t)Ms * d);
  float* e2 = ctx->arena.get<float>((int64_t)Ms * d);
  float* mem = ctx->arena.get<float>((int64_t)Ms * d);
  float* t1 = ctx->arena.get<float>((int64_t)Mt * d);
  float* t2 = ctx->arena.get<float>((int64_t)Mt * d);
  float* t3 = ctx->arena.get<float>((int64_t)Mt * d);
  float* o = ctx->arena.get<float>((int64_t)Mx * d);
  float* h = ctx->arena.get<float>((int64_t)Mx * ffn);
  float* qkv = ctx->arena.get<float>((int64_t)Mx * 3 * d);     // reduced self-attention projections; the cross-attention's q
  float* kvm = ctx->arena.get<float>((int64_t)Ms * 2 * d);     // reduced K, V of the encoder memory
  if (!SVG_LAUNCHING(ctx)) return;

  std::vector<WalkOp> ops;
  ops.reserve(160);
  auto gemm = [&](const float* X, int ld, const float* W, float* slab, int M, int N, int K, bool bar = true) {
    WalkOp op{};
    op.kind = WK_GEMM; op.bar = bar; op.M = M; op.N = N; op.K = K; op.ld = ld; op.ksplit = K / 128; op.X = X; op.W = W; op.slab = slab;
    ops.push_back(op);
  };
  auto red = [&](const float* slab, int M, int N, int K, const float* bias, float* Y, bool relu, bool bar = true) {
    WalkOp op{};
    op.kind = WK_RED; op.bar = bar; op.M = M; op.N = N; op.ksplit = K / 128; op.slab = (float*)slab; op.bias = bias; op.Y = Y; op.relu = relu;
    ops.push_back(op);
  };
  // y = LN(x + (slabs + bias)) g + b [; y2 = LN(y) g2 + b2]
  auto ln = [&](const float* slab, int M, int K, const float* bias, const float* res, const float* g, const float* b, float* Y,
                const float* g2 = nullptr, const float* b2 = nullptr, float* Y2 = nullptr) {
    WalkOp op{};
    op.kind = WK_LN; op.bar = 1; op.M = M; op.N = d; op.ksplit = slab ? K / 128 : 0; op.slab = (float*)slab; op.bias = bias; op.res = res;
    op.g1 = g; op.b1 = b; op.Y = Y; op.g2 = g2; op.b2 = b2; op.Y2 = Y2; op.eps = 1e-5f;
    ops.push_back(op);
  };
  // q (Tq*B rows of q_ld floats, q_span floats to the end of its buffer), k / v (Tk*B rows of kv_ld floats)
  auto attn = [&](const float* q, int q_ld, int64_t q_span, const float* k, const float* v, int kv_ld, int64_t kv_span, int Tq, int Tk, const float* msk,
                  const float* kpad) {
    WalkOp op{};
    op.kind = WK_ATTN; op.bar = 1; op.Tq = Tq; op.Tk = Tk; op.B = B; op.heads = heads; op.hd = hd;
    op.q_ld = q_ld; op.kv_ld = kv_ld; op.q_span = (int)q_span; op.kv_span = (int)kv_span;
    op.qs = q; op.ks = k; op.vs = v; op.mask = msk; op.kpad = kpad; op.Y = o;
    ops.push_back(op);
  };
  auto embed = [&](const float* slab, int T, float* Y, bool bar) {
    WalkOp op{};
    op.kind = WK_EMBED; op.bar = bar; op.M = B * T; op.N = d_img; op.ksplit = d_lat / 128; op.slab = (float*)slab; op.bias = m->emb_b; op.Y = Y;
    op.pe = m->pe; op.pe_row = pe_row; op.text = text; op.d_txt = m->text_dim; op.T = T; op.B = B; op.scale = sqrtf((float)d);
    ops.push_back(op);
  };

  // embeddings (the launch's inputs: no barrier before the first stage)
  gemm(src, d_lat, m->emb_w, slabA, B * Ts, d_img, d_lat, false);
  if (!same) gemm(tgt, d_lat, m->emb_w, slabB, B * Tt, d_img, d_lat, false);
  embed(slabA, Ts, xs_e, true);
  if (!same) embed(slabB, Tt, xt_e, false);
  const float* xs_cur = xs_e;
  for (int i = 0; i < m->enc_layers; ++i) {
    const XfModel::LayerW& w = m->enc_w[i];
    const bool last = (i + 1 == m->enc_layers);
    gemm(xs_cur, d, w.in_w, slabA, Ms, 3 * d, d);
    red(slabA, Ms, 3 * d, d, w.in_b, qkv, false);
    attn(qkv, 3 * d, (int64_t)Ms * 3 * d, qkv + d, qkv + 2 * d, 3 * d, (int64_t)Ms * 3 * d - d, Ts, Ts, nullptr, src_pad);
    gemm(o, d, w.out_w, slabA, Ms, d, d);
    ln(slabA, Ms, d, w.out_b, xs_cur, w.n_w[0], w.n_b[0], e1);
    gemm(e1, d, w.l1_w, slabA, Ms, ffn, d);
    red(slabA, Ms, ffn, d, w.l1_b, h, true);
    gemm(h, ffn, w.l2_w, slabA, Ms, d, ffn);
    if (last) ln(slabA, Ms, ffn, w.l2_b, e1, w.n_w[1], w.n_b[1], nullptr, m->encn_w, m->encn_b, mem);   // + transformer.encoder.norm
    else ln(slabA, Ms, ffn, w.l2_b, e1, w.n_w[1], w.n_b[1], e2);
    xs_cur = e2;
  }
  if (m->enc_layers == 0) ln(nullptr, Ms, 0, nullptr, xs_cur, m->encn_w, m->encn_b, mem);
  const float* xt_cur = xt_e;
  for (int i = 0; i < m->dec_layers; ++i) {
    const XfModel::LayerW& w = m->dec_w[i];
    const bool last = (i + 1 == m->dec_layers);
    gemm(xt_cur, d, w.in_w, slabA, Mt, 3 * d, d);
    gemm(mem, d, w.cin_w + (int64_t)d * d, slabB, Ms, 2 * d, d, false);                            // K, V of the memory: rows d .. 3d of in_proj
    red(slabA, Mt, 3 * d, d, w.in_b, qkv, false);
    red(slabB, Ms, 2 * d, d, w.cin_b + d, kvm, false, false);
    attn(qkv, 3 * d, (int64_t)Mt * 3 * d, qkv + d, qkv + 2 * d, 3 * d, (int64_t)Mt * 3 * d - d, Tt, Tt, mask, tgt_pad);
    gemm(o, d, w.out_w, slabA, Mt, d, d);
    ln(slabA, Mt, d, w.out_b, xt_cur, w.n_w[0], w.n_b[0], t1);
    gemm(t1, d, w.cin_w, slabA, Mt, d, d);                                                          // q: rows 0 .. d of in_proj
    red(slabA, Mt, d, d, w.cin_b, qkv, false);
    attn(qkv, d, (int64_t)Mt * d, kvm, kvm + d, 2 * d, (int64_t)Ms * 2 * d, Tt, Ts, nullptr, nullptr);
    gemm(o, d, w.cout_w, slabA, Mt, d, d);
    ln(slabA, Mt, d, w.cout_b, t1, w.n_w[1], w.n_b[1], t2);
    gemm(t2, d, w.l1_w, slabA, Mt, ffn, d);
    red(slabA, Mt, ffn, d, w.l1_b, h, true);
    gemm(h, ffn, w.l2_w, slabA, Mt, d, ffn);
    if (last) ln(slabA, Mt, ffn, w.l2_b, t2, w.n_w[2], w.n_b[2], nullptr, m->decn_w, m->decn_b, t3);     // + transformer.decoder.norm
    else ln(slabA, Mt, ffn, w.l2_b, t2, w.n_w[2], w.n_b[2], t3);
    xt_cur = t3;
  }
  if (m->dec_layers == 0) { ln(nullptr, Mt, 0, nullptr, xt_cur, m->decn_w, m->decn_b, t3); xt_cur = t3; }
  gemm(xt_cur, d, m->out_w, slabA, Mt, d_lat, d);
  red(slabA, Mt, d_lat, d, m->out_b, out_tb, false);
  double flops = 0, bytes = 0;                        // the per-GEMM kernels' accounting: W once, X and the product once
  for (const WalkOp& op : ops)
    if (op.kind == WK_GEMM) {
      flops += 2.0 * op.M * (double)op.N * op.K;
      bytes += 4.0 * ((double)op.N * op.K + (double)op.M * op.K + (double)op.M * op.N);
    }
  ProfScope ps(ctx, PK_XF_GEMM, s, flops, bytes, "walk");
  xf_walk_launch(ctx, ops.data(), (int)ops.size(), Mx, xf_walk_lds_bytes(Mx, std::max(Ts, Tt), std::max(Ts, Tt), hd), s);
}

// ---- the small-row form (xf_walk.hip: xf_walk_small_kernel): at most 8 rows per forward — single-clip sampling ------------------------
// Stages per encoder layer: in_proj (LayerNorm of its input folded in; q, k, v as column blocks of one stage) | attention | out_proj + bias +
// residual | linear1 (LayerNorm 1 folded in) + bias + ReLU | linear2 + bias + residual — 5 device-wide barriers instead of 9; a decoder layer
// has 8 instead of 16 (the K / V projection of the encoder memory rides on the self-attention in_proj's barrier, with the encoder's two final
// LayerNorms folded into its input).  What flows between layers is the PRE-LayerNorm sum; the consumer normalises its own LDS copy of the rows
// and publishes the normalised rows (Yln) for the residual of the stage after next.
static bool xf_walk_small_usable(const XfModel* m, int B, int Ts, int Tt, hipStream_t s) {
  // $SVG_XF_WALK_SMALL: 1 always (where the shapes fit), 0 never, unset: where it is ahead of the split-K walk — d_model <= 1024 (measured, one
  // clip of 6 tokens, profiles/r05_walk_small_vs_splitk.txt: d = 256 0.789 -> 0.618 ms, 512 0.811 -> 0.637, 1024 0.897 -> 0.785; d = 2048
  // 1.053 vs 1.059: there a stage is bound by the 64 KB of weights a compute unit has to pull per column block, not by the stage count)
  const int64_t mode = svg_env_i64("SVG_XF_WALK_SMALL", -1);
  if (!xf_walk_enabled(s) || mode == 0 || (mode < 0 && m->d_model > 1024)) return false;
  const int d = m->d_model, hd = d / m->heads;
  if (m->text_dim != 0 || B * std::max(Ts, Tt) > kWalkSmallRows || hd % 4) return false;
  for (int K : {d, m->ffn, m->d_lat})
    if (K % 256 != 0 || K < 256 || K > kWalkSmallMaxK) return false;
  {
    // exact stage count of xf_forward_walk_small's table: a GEMM stage is cut into column blocks of 8 x (workgroups of the grid) columns
    const int blk = 8 * std::max(1, xf_walk_grid());
    auto nb = [&](int N) { return (N + blk - 1) / blk; };
    const int ffn = m->ffn;
    const bool same = Ts == Tt;                      // (src == tgt is the caller's business: count the longer table)
    int64_t n = (int64_t)nb(d) * (same ? 1 : 2);
    n += (int64_t)m->enc_layers * (nb(3 * d) + 1 + nb(d) + nb(ffn) + nb(d));
    n += (int64_t)m->dec_layers * (nb(3 * d) + nb(2 * d) + 1 + nb(d) + nb(d) + 1 + nb(d) + nb(ffn) + nb(d));
    n += nb(m->d_lat) + nb(d);                       // output projection (+ the second embedding when src != tgt at equal lengths)
    if (n > kWalkMaxOps) return false;
  }
  const int T = std::max(Ts, Tt);
  const int64_t lds = std::max<int64_t>((int64_t)kWalkSmallRows * (kWalkSmallMaxK * 4 + 64) + 4 * kWalkSmallMaxK * 4, ((int64_t)3 * T * hd + 2 * 32 * 33 + 32) * 4);
  return xf_walk_grid() >= 8 && xf_walk_available(kWalkSmallRows, lds);
}

static void xf_forward_walk_small(svg_ctx* ctx, XfModel* m, const float* src, const float* tgt, int B, int Ts, int Tt, const float* mask,
                                  const int32_t* pe_row, float* out_tb, hipStream_t s, const float* src_pad, const float* tgt_pad) {
  const int d = m->d_model, ffn = m->ffn, d_lat = m->d_lat, heads = m->heads, hd = d / heads;
  const int Ms = Ts * B, Mt = Tt * B, Mx = std::max(Ms, Mt);
  const bool same = (tgt == src && Ts == Tt);
  float* xs_e = ctx->arena.get<float>((int64_t)Ms * d);                   // embeddings
  float* xt_e = same ? xs_e : ctx->arena.get<float>((int64_t)Mt * d);
  float* qkv = ctx->arena.get<float>((int64_t)Mx * 3 * d);
  float* kvm = ctx->arena.get<float>((int64_t)Ms * 2 * d);
  float* o = ctx->arena.get<float>((int64_t)Mx * d);
  float* h = ctx->arena.get<float>((int64_t)Mx * ffn);
  float* pA = ctx->arena.get<float>((int64_t)Mx * d);                     // pre-LayerNorm sums: attention block, cross-attention block,
  float* pB = ctx->arena.get<float>((int64_t)Mx * d);
  float* pC = ctx->arena.get<float>((int64_t)Mx * d);                     // feed-forward block (the next layer's input)
  float* pM = ctx->arena.get<float>((int64_t)Ms * d);                     // the last encoder layer's (the memory, before its two norms)
  float* lA = ctx->arena.get<float>((int64_t)Mx * d);                     // the normalised rows the consumers publish
  float* lB = ctx->arena.get<float>((int64_t)Mx * d);
  float* lC = ctx->arena.get<float>((int64_t)Mx * d);
  if (!SVG_LAUNCHING(ctx)) return;
  const int blk = 8 * xf_walk_grid();                                     // widest column block a stage serves (8 columns per workgroup)

  std::vector<WalkOp> ops;
  ops.reserve(128);
  struct Ln { const float* g1 = nullptr; const float* b1 = nullptr; const float* g2 = nullptr; const float* b2 = nullptr; float* Yln = nullptr; };
  // Y[:, 0..N) = act(LN(X) W^T + bias) (+ res); first block of a stage: barrier (unless `nobar`), X staged and normalised; further blocks reuse it
  auto gemmf = [&](const float* X, int ld, int M, int K, const float* W, const float* bias, int N, float* Y, int ldy, const Ln& ln, bool relu,
                   const float* res, int ld_res, bool bar) -> WalkOp& {
    size_t first = ops.size();
    for (int n0 = 0; n0 < N; n0 += blk) {
      WalkOp op{};
      op.kind = WK_GEMMF; op.bar = (n0 == 0 && bar) ? 1 : 0; op.reuse_x = n0 == 0 ? 0 : 1;
      op.M = M; op.N = std::min(blk, N - n0); op.K = K; op.ld = ld; op.X = X; op.W = W + (int64_t)n0 * K; op.bias = bias ? bias + n0 : nullptr;
      op.Y = Y + n0; op.ldy = ldy; op.res = res ? res + n0 : nullptr; op.ld_res = ld_res; op.relu = relu ? 1 : 0; op.eps = 1e-5f;
      if (n0 == 0) { op.g1 = ln.g1; op.b1 = ln.b1; op.g2 = ln.g2; op.b2 = ln.b2; op.Yln = ln.Yln; }
      ops.push_back(op);
    }
    return ops[first];
  };
  auto attn = [&](const float* q, int q_ld, int64_t q_span, const float* k, const float* v, int kv_ld, int64_t kv_span, int Tq, int Tk, const float* msk,
                  const float* kpad) {
    WalkOp op{};
    op.kind = WK_ATTN; op.bar = 1; op.Tq = Tq; op.Tk = Tk; op.B = B; op.heads = heads; op.hd = hd;
    op.q_ld = q_ld; op.kv_ld = kv_ld; op.q_span = (int)q_span; op.kv_span = (int)kv_span;
    op.qs = q; op.ks = k; op.vs = v; op.mask = msk; op.kpad = kpad; op.Y = o;
    ops.push_back(op);
  };
  auto embed = [&](const float* x, int T, float* Y, bool bar) {
    WalkOp& op = gemmf(x, d_lat, B * T, d_lat, m->emb_w, m->emb_b, d, Y, d, Ln{}, false, nullptr, 0, bar);
    int n0 = 0;
    for (size_t i = &op - ops.data(); i < ops.size(); ++i) {          // every column block: rows in (b, t) order, out (t, b); scale + PE
      ops[i].perm = 1; ops[i].B = B; ops[i].T = T; ops[i].scale = sqrtf((float)d); ops[i].pe = m->pe + n0; ops[i].ld_res = d; ops[i].pe_row = pe_row;
      n0 += ops[i].N;
    }
  };

  embed(src, Ts, xs_e, false);                                           // the launch's inputs: no barrier before the first stage
  if (!same) embed(tgt, Tt, xt_e, false);
  // ---- encoder.  `cur` = the layer's input rows before their LayerNorm (`cln`: its parameters; none for the embedding), `curl` = where
  // the normalised rows are published (the embedding itself when there is no norm)
  const float* cur = xs_e; Ln cln; const float* curl = xs_e;
  for (int i = 0; i < m->enc_layers; ++i) {
    const XfModel::LayerW& w = m->enc_w[i];
    const bool last = (i + 1 == m->enc_layers);
    Ln l0 = cln; if (l0.g1) { l0.Yln = lA; curl = lA; }
    gemmf(cur, d, Ms, d, w.in_w, w.in_b, 3 * d, qkv, 3 * d, l0, false, nullptr, 0, true);
    attn(qkv, 3 * d, (int64_t)Ms * 3 * d, qkv + d, qkv + 2 * d, 3 * d, (int64_t)Ms * 3 * d - d, Ts, Ts, nullptr, src_pad);
    gemmf(o, d, Ms, d, w.out_w, w.out_b, d, pA, d, Ln{}, false, curl, d, true);
    gemmf(pA, d, Ms, d, w.l1_w, w.l1_b, ffn, h, ffn, Ln{w.n_w[0], w.n_b[0], nullptr, nullptr, lB}, true, nullptr, 0, true);
    float* pout = last ? pM : pC;
    gemmf(h, ffn, Ms, ffn, w.l2_w, w.l2_b, d, pout, d, Ln{}, false, lB, d, true);
    cur = pout; cln = Ln{w.n_w[1], w.n_b[1], nullptr, nullptr, nullptr}; curl = nullptr;
  }
  // the memory = encoder.norm(norm2(last sum)) (or encoder.norm(embedding) for an empty encoder): folded into every consumer
  Ln lmem = m->enc_layers ? Ln{cln.g1, cln.b1, m->encn_w, m->encn_b, nullptr} : Ln{m->encn_w, m->encn_b, nullptr, nullptr, nullptr};
  const float* memp = cur;
  // ---- decoder
  cur = xt_e; cln = Ln{}; curl = xt_e;
  for (int i = 0; i < m->dec_layers; ++i) {
    const XfModel::LayerW& w = m->dec_w[i];
    Ln l0 = cln; if (l0.g1) { l0.Yln = lA; curl = lA; }
    gemmf(cur, d, Mt, d, w.in_w, w.in_b, 3 * d, qkv, 3 * d, l0, false, nullptr, 0, true);
    gemmf(memp, d, Ms, d, w.cin_w + (int64_t)d * d, w.cin_b + d, 2 * d, kvm, 2 * d, lmem, false, nullptr, 0, false);     // K, V of the memory
    attn(qkv, 3 * d, (int64_t)Mt * 3 * d, qkv + d, qkv + 2 * d, 3 * d, (int64_t)Mt * 3 * d - d, Tt, Tt, mask, tgt_pad);
    gemmf(o, d, Mt, d, w.out_w, w.out_b, d, pA, d, Ln{}, false, curl, d, true);
    gemmf(pA, d, Mt, d, w.cin_w, w.cin_b, d, qkv, d, Ln{w.n_w[0], w.n_b[0], nullptr, nullptr, lB}, false, nullptr, 0, true);   // q of the cross-attention
    attn(qkv, d, (int64_t)Mt * d, kvm, kvm + d, 2 * d, (int64_t)Ms * 2 * d, Tt, Ts, nullptr, nullptr);
    gemmf(o, d, Mt, d, w.cout_w, w.cout_b, d, pB, d, Ln{}, false, lB, d, true);
    gemmf(pB, d, Mt, d, w.l1_w, w.l1_b, ffn, h, ffn, Ln{w.n_w[1], w.n_b[1], nullptr, nullptr, lC}, true, nullptr, 0, true);
    gemmf(h, ffn, Mt, ffn, w.l2_w, w.l2_b, d, pC, d, Ln{}, false, lC, d, true);
    cur = pC; cln = Ln{w.n_w[2], w.n_b[2], nullptr, nullptr, nullptr}; curl = nullptr;
  }
  Ln lout = m->dec_layers ? Ln{cln.g1, cln.b1, m->decn_w, m->decn_b, nullptr} : Ln{m->decn_w, m->decn_b, nullptr, nullptr, nullptr};
  gemmf(cur, d, Mt, d, m->out_w, m->out_b, d_lat, out_tb, d_lat, lout, false, nullptr, 0, true);
  double flops = 0, bytes = 0;
  for (const WalkOp& op : ops)
    if (op.kind == WK_GEMMF) {
      flops += 2.0 * op.M * (double)op.N * op.K;
      bytes += 4.0 * ((double)op.N * op.K + (op.reuse_x ? 0.0 : (double)op.M * op.K) + (double)op.M * op.N);
    }
  ProfScope ps(ctx, PK_XF_GEMM, s, flops, bytes, "walk_small");
  const int T = std::max(Ts, Tt);
  xf_walk_small_launch(ctx, ops.data(), (int)ops.size(),
                       std::max<int64_t>((int64_t)kWalkSmallRows * (kWalkSmallMaxK * 4 + 64) + 4 * kWalkSmallMaxK * 4, ((int64_t)3 * T * hd + 2 * 32 * 33 + 32) * 4), s);
}

void XfModel::forward(svg_ctx* ctx, const float* src, const float* tgt, int B, int Ts, int Tt, const float* mask,
                      const int32_t* pe_row, float* out, hipStream_t s, const float* text, const float* src_pad, const float* tgt_pad) {
  SVG_CHECK(ready, "transformer: svg_finalize has not been called");
  SVG_CHECK((text_dim > 0) == (text != nullptr), "transformer: the text-conditioned variant needs (and only it takes) a text embedding");
  SVG_CHECK(B >= 1 && Ts >= 1 && Tt >= 1 && Ts <= 32 && Tt <= 32, "transformer: B=%d Ts=%d Tt=%d unsupported (sequences up to 32 tokens)", B, Ts, Tt);
  SVG_CHECK(pe_row || B <= 64, "transformer: batch %d > max_len 64 of the positional table", B);
  const int Tmax = std::max(Ts, Tt);
  // The layer-walking launch (xf_walk.hip) can serve up to kWalkMaxRows rows, and alone on the device it is ahead of the per-GEMM kernels at
  // every size (28 % at 6-48 rows, 8 % at 168).  But it owns every compute unit while it runs: the sampling loop's two stream groups, whose
  // 168-row forwards overlap on the per-GEMM path, serialise (4250 vs 4709 frames/s without denoising, profiles/README.md).  So by default
  // it takes the latency-bound sizes only (SVG_XF_WALK_ROWS, default 96 rows = 16 clips x 6 tokens); larger batches go through the per-GEMM
  // kernels, which stream W once for up to 336 rows (SVG_XF_WALK_SPLIT=1: through the walk in chunks).
  const int Bw = std::max(1, (int)std::min<int64_t>(kWalkMaxRows, svg_env_i64("SVG_XF_WALK_ROWS", 96)) / Tmax);
  const bool walk = xf_walk_usable(this, std::min(B, Bw), Ts, Tt, s) && (B <= Bw || svg_env_i64("SVG_XF_WALK_SPLIT", 0) != 0);
  // at most 8 rows (one clip): the small-row form — whole-K GEMM stages with LayerNorm / bias / residual folded in, 5 + 8 instead of 9 + 16
  // stages per encoder / decoder layer
  const bool walk_small = walk && !text && xf_walk_small_usable(this, B, Ts, Tt, s);
  const int Bc = walk ? Bw : std::max(1, 336 / Tmax);
  auto chunk = [&](const float* srcc, const float* tgtc, int bc, const int32_t* rows, float* dst, const float* textc, const float* sp, const float* tp) {
    if (walk_small) xf_forward_walk_small(ctx, this, srcc, tgtc, bc, Ts, Tt, mask, rows, dst, s, sp, tp);
    else if (walk) xf_forward_walk(ctx, this, srcc, tgtc, bc, Ts, Tt, mask, rows, dst, s, textc, sp, tp);
    else xf_forward_chunk(ctx, this, srcc, tgtc, bc, Ts, Tt, mask, rows, dst, s, textc, sp, tp);
  };
  run_planned(ctx, [&]() {
    // PE rows: the reference indexes the table by batch row (positional_encoding.py:33-35)
    const int32_t* rows = iota;
    if (pe_row) {
      int32_t* r = ctx->arena.get<int32_t>(B);
      if (SVG_LAUNCHING(ctx)) HIP_OK(hipMemcpyAsync(r, pe_row, B * sizeof(int32_t), hipMemcpyDefault, s));
      rows = r;
    }
    if (B <= Bc) {
      chunk(src, tgt, B, rows, out, text, src_pad, tgt_pad);
    } else {
      for (int b0 = 0; b0 < B; b0 += Bc) {
        const int bc = std::min(Bc, B - b0);
        ctx->arena.push();
        float* tmp = ctx->arena.get<float>((int64_t)Tt * bc * d_lat);
        const float* srcc = src + (int64_t)b0 * Ts * d_lat;
        const float* tgtc = (tgt == src) ? srcc : tgt + (int64_t)b0 * Tt * d_lat;
        chunk(srcc, tgtc, bc, rows + b0, tmp, text ? text + (int64_t)b0 * text_dim : nullptr, src_pad ? src_pad + (int64_t)b0 * Ts : nullptr,
              tgt_pad ? tgt_pad + (int64_t)b0 * Tt : nullptr);
        if (SVG_LAUNCHING(ctx))
          HIP_OK(hipMemcpy2DAsync(out + (int64_t)b0 * d_lat, (size_t)B * d_lat * sizeof(float), tmp,
                                  (size_t)bc * d_lat * sizeof(float), (size_t)bc * d_lat * sizeof(float), Tt,
                                  hipMemcpyDeviceToDevice, s));
        ctx->arena.pop();
      }
    }
  });
}

extern "C" int svg_transformer_forward_text(svg_ctx* ctx, const float* src, const float* tgt, const float* text, int B, int Ts, int Tt,
                                            const float* mask, const int32_t* pe_row, float* out, void* stream) {
  try {
    SVG_CHECK(ctx && ctx->xf, "transformer: model not configured");
    xf_walk_check(ctx);
    ctx->xf->forward(ctx, src, tgt, B, Ts, Tt, mask, pe_row, out, (hipStream_t)stream, text);
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}

extern "C" int svg_transformer_forward_padded(svg_ctx* ctx, const float* src, const float* tgt, const float* text, int B, int Ts, int Tt,
                                              const float* mask, const float* src_pad, const float* tgt_pad, const int32_t* pe_row, float* out,
                                              void* stream) {
  try {
    SVG_CHECK(ctx && ctx->xf, "transformer: model not configured");
    xf_walk_check(ctx);
    ctx->xf->forward(ctx, src, tgt, B, Ts, Tt, mask, pe_row, out, (hipStream_t)stream, text, src_pad, tgt_pad);
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}

extern "C" int svg_transformer_forward(svg_ctx* ctx, const float* src, const float* tgt, int B, int Ts, int Tt, const float* mask,
                                       const int32_t* pe_row, float* out, void* stream) {
  try {
    SVG_CHECK(ctx && ctx->xf, "transformer: model not configured");
    xf_walk_check(ctx);
    ctx->xf->forward(ctx, src, tgt, B, Ts, Tt, mask, pe_row, out, (hipStream_t)stream);
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}
