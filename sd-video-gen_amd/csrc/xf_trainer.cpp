// Training step of the latent Transformer: forward in train mode keeping what backward needs, the criterion, backward through
// every layer, Adam.  Reference: trainers/trainer.py:111-190 (train_loop body: pred = model(new_batch, y_input, tgt_mask);
// loss_fn(pred[-F:], y_expected[-F:]); opt.zero_grad(); loss.backward(); opt.step()), :65-109 (criterion), :192-260
// (validation_loop: the same loss in eval mode) over torch.nn.Transformer (post-norm, ReLU, dropout after the positional
// encoding, on the attention probabilities, after each sublayer and inside the feed-forward) and torch.optim.Adam(lr).
// The Stable Diffusion side stays frozen (encode_batch is the VAE encoder the sampling path already has).
#include "models.h"
#include "../../include/svg_hip.h"
#include <cmath>
#include <cstring>
#include <cstdlib>

struct XfTrain {
  struct Slot { float* g = nullptr; float* m = nullptr; float* v = nullptr; int64_t n = 0; };
  std::unordered_map<std::string, Slot> slots;
  XfAdamTensor* d_tens = nullptr;
  XfAdamChunk* d_chunks = nullptr;
  int n_chunks = 0;
  int step = 0;
  float* d_losses = nullptr;     // [5]
  float* h_losses = nullptr;     // pinned mirror: a device-to-pageable copy goes through the runtime's staging path and its host thread
  // One plan per call signature (shapes + every criterion / dropout parameter except the seed): the workspace high-water mark and,
  // for the loss (+ backward) calls, the captured hipGraph of the whole step.  ~540 launches per step cost more host time than
  // the GPU needs to run them (15.3 ms wall against 9.3 ms of kernels); the graph replays them from fixed staging buffers.
  struct Plan {
    svg_train_cfg cfg{}; int B = 0, Ts = 0, Tt = 0, backward = 0, mode = 0; bool has_mask = false, has_text = false;
    int64_t high = 0;
    hipGraphExec_t exec = nullptr;
    char* arena_base = nullptr;
    float *src = nullptr, *tgt = nullptr, *exp = nullptr, *text = nullptr, *mask = nullptr;
  };
  std::vector<Plan> plans;
  uint64_t* d_seed = nullptr;     // device seed word read by every dropout site
  uint64_t* h_seed = nullptr;     // pinned ring of 16 seeds (a call that does not synchronise must not race the next one's seed)
  hipEvent_t seed_ev[16] = {};    // recorded behind each slot's upload: the 17th unsynchronised call waits for the 1st one's copy
  int seed_slot = 0;
  // The training step's own workspace.  The captured graph bakes workspace addresses: if it lived in the context-wide arena, a
  // VAE / UNet / CLIP call of the same context on ANOTHER stream (or a forward that regrows the arena) could reuse or move the
  // memory while the graph is still running.  loss_pass swaps these in for the duration of its planning and launches.
  Arena ws;
  DevBuf ws_buf;
  // Backward runs dW (xf_gemm_tn) of a layer on a side stream beside its dX chain (they only share the read-only dY): inside the
  // captured step this becomes a fork / join in the graph.  Events are pooled (created before any capture).
  hipStream_t side = nullptr;
  std::vector<hipEvent_t> fork_ev;
  int fork_next = 0;
  hipEvent_t next_event() { hipEvent_t e = fork_ev[fork_next]; fork_next = (fork_next + 1) % (int)fork_ev.size(); return e; }
  std::vector<void*> bufs;       // device allocations of the training state (freed with it)
  void* dalloc(int64_t bytes) {
    void* p = nullptr;
    HIP_OK(hipMalloc(&p, (size_t)std::max<int64_t>(bytes, 256)));
    bufs.push_back(p);
    return p;
  }
  ~XfTrain() {
    for (auto& pl : plans) if (pl.exec) hipGraphExecDestroy(pl.exec);
    for (void* p : bufs) hipFree(p);
    if (h_losses) hipHostFree(h_losses);
    if (h_seed) hipHostFree(h_seed);
    for (auto e : seed_ev) if (e) hipEventDestroy(e);
    for (auto e : fork_ev) if (e) hipEventDestroy(e);
    if (side) hipStreamDestroy(side);
    if (ws_buf.p) hipFree(ws_buf.p);
  }
};

namespace {

constexpr int kAdamChunk = 1 << 16;

struct LinTape { const float* x = nullptr; int M = 0, N = 0, K = 0; std::string w, b; int64_t woff = 0, boff = 0; };
struct LnTape { float* xhat = nullptr; float* rstd = nullptr; std::string p; XfDrop dr{nullptr, 0, 0.f}; int M = 0; };
struct MhaTape {
  LinTape in_q, in_kv, outp;   // self: in_q is the whole packed projection
  float *qkv = nullptr, *q = nullptr, *kv = nullptr, *P = nullptr;
  const float* mask = nullptr;
  int Tq = 0, Tk = 0;
  bool self = true;
  XfDrop dr{nullptr, 0, 0.f};
};
struct FfnTape { LinTape l1, l2; float* r = nullptr; float gate_scale = 1.f; };
struct EncTape { MhaTape sa; LnTape n1; FfnTape ff; LnTape n2; };
struct DecTape { MhaTape sa; LnTape n1; MhaTape ca; LnTape n2; FfnTape ff; LnTape n3; };

struct Run {
  svg_ctx* ctx; XfModel* m; XfTrain* tr; hipStream_t s; int B; const uint64_t* seed; float p; bool grads;
  const float* text = nullptr;
  uint32_t site = 0;
  bool fork = false;             // dW on the side stream (set by loss_pass)
  bool tn_pending = false;
  bool go() const { return SVG_LAUNCHING(ctx); }
  void join() {                  // the side stream's work so far is ordered before whatever `s` launches next
    if (!tn_pending) return;
    hipEvent_t e = tr->next_event();
    HIP_OK(hipEventRecord(e, tr->side));
    HIP_OK(hipStreamWaitEvent(s, e, 0));
    tn_pending = false;
  }
  XfDrop drop() { return XfDrop{seed, site++, p}; }
  const float* W(const std::string& n) { return m->ws.get(n).f32; }
  float* G(const std::string& n) { return tr->slots.at(n).g; }
  template <typename T> T* get(int64_t n) { return ctx->arena.get<T>(n); }

  // ---- linear ------------------------------------------------------------------------------------------------------------
  float* lin(LinTape& t, const float* x, const std::string& w, const std::string& b, int M, int N, int K, int64_t woff = 0, int64_t boff = 0) {
    t = LinTape{x, M, N, K, w, b, woff, boff};
    float* y = get<float>((int64_t)M * N);
    for (int m0 = 0; m0 < M; m0 += 336)       // xf_gemm streams W once per 336 rows (it plans its own split-K slabs in the dry pass)
      xf_gemm(ctx, x + (int64_t)m0 * K, W(w) + woff, W(b) + boff, y + (int64_t)m0 * N, std::min(336, M - m0), N, K, 0, s);
    return y;
  }
  // dW, db of the layer; dx = gate(dy W) + add (dx == nullptr: not wanted)
  void lin_bwd(const LinTape& t, const float* dy, float* dx, const float* add = nullptr, const float* gate = nullptr, float gate_scale = 1.f,
               bool accumulate = false) {
    float* slabs = dx ? get<float>(xf_gemm_nn_slab_floats(t.M, t.N, t.K)) : nullptr;
    if (!go()) return;
    if (fork) {
      // at most one dW in flight beside the main chain: the previous one is joined first, so nothing launched from here on can
      // touch a buffer it still reads; this one waits for dY and then runs beside the dX GEMM and what follows it
      join();
      hipEvent_t e = tr->next_event();
      HIP_OK(hipEventRecord(e, s));
      HIP_OK(hipStreamWaitEvent(tr->side, e, 0));
      xf_gemm_tn(dy, t.N, t.x, t.K, G(t.w) + t.woff, G(t.b) + t.boff, t.M, t.N, t.K, accumulate, tr->side);
      tn_pending = true;
    } else {
      xf_gemm_tn(dy, t.N, t.x, t.K, G(t.w) + t.woff, G(t.b) + t.boff, t.M, t.N, t.K, accumulate, s);
    }
    if (dx) xf_gemm_nn(dy, t.N, W(t.w) + t.woff, slabs, dx, t.M, t.N, t.K, gate, gate_scale, add, s);
  }

  // ---- add + LayerNorm --------------------------------------------------------------------------------------------------
  float* add_ln(LnTape& t, const float* x, const float* r, const std::string& p, int M, bool drop_r) {
    const int d = m->d_model;
    t.p = p; t.M = M;
    t.dr = drop_r ? drop() : XfDrop{seed, 0, 0.f};
    t.xhat = get<float>((int64_t)M * d);
    t.rstd = get<float>(M);
    float* y = get<float>((int64_t)M * d);
    if (go()) xf_add_ln_train(x, r, t.dr, W(p + "weight"), W(p + "bias"), y, t.xhat, t.rstd, M, d, 1e-5f, s);
    return y;
  }
  // returns dz (gradient of both the residual input and, masked in *dz_drop, of the sublayer output)
  float* add_ln_bwd(const LnTape& t, const float* dy, float** dz_drop) {
    const int d = m->d_model;
    float* dz = get<float>((int64_t)t.M * d);
    float* dzd = nullptr;
    if (dz_drop) { dzd = t.dr.p > 0.f ? get<float>((int64_t)t.M * d) : dz; *dz_drop = dzd; }
    if (go())
      xf_ln_bwd(dy, t.xhat, t.rstd, W(t.p + "weight"), dz, dzd == dz ? nullptr : dzd, t.dr, G(t.p + "weight"), G(t.p + "bias"), t.M, d, s);
    return dz;
  }

  // ---- multi-head attention ---------------------------------------------------------------------------------------------
  float* mha(MhaTape& t, const std::string& p, const float* xq, int Tq, const float* xkv, int Tk, const float* mask, bool self) {
    const int d = m->d_model, hd = d / m->heads;
    t.Tq = Tq; t.Tk = Tk; t.self = self; t.mask = mask;
    float* o = get<float>((int64_t)Tq * B * d);
    t.P = get<float>((int64_t)B * m->heads * Tq * Tk);
    if (self) {
      t.qkv = lin(t.in_q, xq, p + "in_proj_weight", p + "in_proj_bias", Tq * B, 3 * d, d);
      t.dr = drop();
      if (go()) xf_attention_train(t.qkv, 3 * d, t.qkv + d, t.qkv + 2 * d, 3 * d, mask, o, t.P, Tq, Tk, B, m->heads, hd, t.dr, s);
    } else {
      t.q = lin(t.in_q, xq, p + "in_proj_weight", p + "in_proj_bias", Tq * B, d, d);
      t.kv = lin(t.in_kv, xkv, p + "in_proj_weight", p + "in_proj_bias", Tk * B, 2 * d, d, (int64_t)d * d, d);
      t.dr = drop();
      if (go()) xf_attention_train(t.q, d, t.kv, t.kv + d, 2 * d, mask, o, t.P, Tq, Tk, B, m->heads, hd, t.dr, s);
    }
    return lin(t.outp, o, p + "out_proj.weight", p + "out_proj.bias", Tq * B, d, d);
  }
  // da: gradient of the block output.  dxq = ... + add_q; self: the k/v gradients flow into the same dxq; cross: dmem (+)= ...
  void mha_bwd(const MhaTape& t, const float* da, float* dxq, const float* add_q, float* dmem, bool dmem_accumulate) {
    const int d = m->d_model, hd = d / m->heads;
    float* d_o = get<float>((int64_t)t.Tq * B * d);
    lin_bwd(t.outp, da, d_o);
    if (t.self) {
      float* dqkv = get<float>((int64_t)t.Tq * B * 3 * d);
      if (go())
        xf_attention_bwd(d_o, t.qkv, 3 * d, t.qkv + d, t.qkv + 2 * d, 3 * d, t.P, dqkv, 3 * d, dqkv + d, dqkv + 2 * d, 3 * d, t.Tq, t.Tk, B,
                         m->heads, hd, t.dr, s);
      lin_bwd(t.in_q, dqkv, dxq, add_q);
    } else {
      float* dq = get<float>((int64_t)t.Tq * B * d);
      float* dkv = get<float>((int64_t)t.Tk * B * 2 * d);
      if (go())
        xf_attention_bwd(d_o, t.q, d, t.kv, t.kv + d, 2 * d, t.P, dq, d, dkv, dkv + d, 2 * d, t.Tq, t.Tk, B, m->heads, hd, t.dr, s);
      lin_bwd(t.in_q, dq, dxq, add_q);
      lin_bwd(t.in_kv, dkv, dmem, dmem_accumulate ? dmem : nullptr);
    }
  }

  // ---- feed-forward: linear2(dropout(relu(linear1(x)))) ---------------------------------------------------------------------
  float* ffn(FfnTape& t, const std::string& p, const float* x, int M) {
    float* h = lin(t.l1, x, p + "linear1.weight", p + "linear1.bias", M, m->ffn, m->d_model);
    t.r = get<float>((int64_t)M * m->ffn);
    const XfDrop dr = drop();
    t.gate_scale = dr.p > 0.f ? 1.f / (1.f - dr.p) : 1.f;
    if (go()) xf_relu_drop(h, t.r, (int64_t)M * m->ffn, dr, s);
    return lin(t.l2, t.r, p + "linear2.weight", p + "linear2.bias", M, m->d_model, m->ffn);
  }
  void ffn_bwd(const FfnTape& t, const float* df, float* dx, const float* add) {
    float* dh = get<float>((int64_t)t.l1.M * m->ffn);
    lin_bwd(t.l2, df, dh, nullptr, t.r, t.gate_scale);        // r > 0 exactly where ReLU and the dropout let the gradient through
    lin_bwd(t.l1, dh, dx, add);
  }

  // ---- embedding + positional encoding --------------------------------------------------------------------------------------
  float* embed(LinTape& t, XfDrop& dr, const float* x, int T, const int32_t* pe_row) {
    const int d = m->d_model, d_img = d - m->text_dim;
    const std::string en = m->text_dim ? "project_image_embedding" : "embedding";
    float* e = lin(t, x, en + ".weight", en + ".bias", B * T, d_img, m->d_lat);
    float* y = get<float>((int64_t)B * T * d);
    dr = drop();
    if (go()) xf_embed_post_train(e, m->pe, pe_row, text, m->text_dim, y, B, T, d, sqrtf((float)d), dr, s);
    return y;
  }
  void embed_bwd(const LinTape& t, const XfDrop& dr, const float* dy, int T, bool accumulate) {
    const int d = m->d_model, d_img = d - m->text_dim;
    float* de = get<float>((int64_t)B * T * d_img);
    if (go()) xf_embed_post_bwd(dy, de, B, T, d, d_img, sqrtf((float)d), dr, s);
    lin_bwd(t, de, nullptr, nullptr, nullptr, 1.f, accumulate);
  }
};

void ensure_train(svg_ctx* ctx, XfModel* m) {
  if (m->train) return;
  SVG_CHECK(m->ready, "transformer: svg_finalize has not been called");
  std::unique_ptr<XfTrain> tr(new XfTrain());
  std::vector<XfAdamTensor> tens;
  std::vector<XfAdamChunk> chunks;
  std::vector<std::string> names;
  for (auto& kv : m->ws.map)
    if (kv.first != "positional_encoder.pos_encoding") names.push_back(kv.first);
  std::sort(names.begin(), names.end());
  for (auto& n : names) {
    const Weight& w = m->ws.get(n);
    XfTrain::Slot sl;
    sl.n = w.numel;
    sl.g = (float*)tr->dalloc(3 * w.numel * sizeof(float));
    sl.m = sl.g + w.numel; sl.v = sl.m + w.numel;
    HIP_OK(hipMemset(sl.g, 0, 3 * w.numel * sizeof(float)));
    tr->slots[n] = sl;
    const int ti = (int)tens.size();
    tens.push_back(XfAdamTensor{w.f32, sl.g, sl.m, sl.v});
    for (int64_t off = 0; off < w.numel; off += kAdamChunk)
      chunks.push_back(XfAdamChunk{ti, (int32_t)std::min<int64_t>(kAdamChunk, w.numel - off), off});
  }
  tr->d_tens = (XfAdamTensor*)tr->dalloc(tens.size() * sizeof(XfAdamTensor));
  tr->d_chunks = (XfAdamChunk*)tr->dalloc(chunks.size() * sizeof(XfAdamChunk));
  tr->d_losses = (float*)tr->dalloc(5 * sizeof(float));
  HIP_OK(hipHostMalloc((void**)&tr->h_losses, 5 * sizeof(float), hipHostMallocDefault));
  tr->d_seed = (uint64_t*)tr->dalloc(sizeof(uint64_t));
  HIP_OK(hipStreamCreateWithFlags(&tr->side, hipStreamNonBlocking));
  tr->fork_ev.resize(64);
  for (auto& e : tr->fork_ev) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIP_OK(hipHostMalloc((void**)&tr->h_seed, 16 * sizeof(uint64_t), hipHostMallocDefault));
  HIP_OK(hipMemcpy(tr->d_tens, tens.data(), tens.size() * sizeof(XfAdamTensor), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(tr->d_chunks, chunks.data(), chunks.size() * sizeof(XfAdamChunk), hipMemcpyHostToDevice));
  tr->n_chunks = (int)chunks.size();
  m->train = tr.release();
}

// expected == nullptr: forward only (train-mode dropout when backward != 0), the prediction (Tt,B,D_lat) goes to pred_out
void loss_pass(svg_ctx* ctx, XfModel* m, const svg_train_cfg& cfg, const float* src, const float* tgt, const float* expected, const float* text,
               int B, int Ts, int Tt, const float* mask, int backward, float* losses_host, hipStream_t s, float* pred_out = nullptr) {
  SVG_CHECK(m->ready, "transformer: svg_finalize has not been called");
  SVG_CHECK((m->text_dim > 0) == (text != nullptr), "transformer: the text-conditioned variant needs (and only it takes) a text embedding");
  SVG_CHECK(B >= 1 && B <= 64 && Ts >= 1 && Tt >= 1 && Ts <= 32 && Tt <= 32, "transformer training: B=%d Ts=%d Tt=%d unsupported (B <= 64, T <= 32)", B, Ts, Tt);
  SVG_CHECK(cfg.dropout_p >= 0.f && cfg.dropout_p < 1.f, "dropout_p %g out of range", cfg.dropout_p);
  if (expected) {
    SVG_CHECK(cfg.frames_to_predict >= 1 && cfg.frames_to_predict <= Tt, "frames_to_predict %d out of range 1..%d", cfg.frames_to_predict, Tt);
    SVG_CHECK(cfg.feat_h > 0 && cfg.feat_w > 0 && 4 * cfg.feat_h * cfg.feat_w == m->d_lat, "criterion: D_lat %d is not 4 x %d x %d", m->d_lat,
              cfg.feat_h, cfg.feat_w);
    SVG_CHECK(cfg.w_contrastive == 0.f || (cfg.temperature > 0.f && cfg.feat_h * cfg.feat_w <= 4096), "contrastive loss: bad temperature / patch count");
  }
  ensure_train(ctx, m);
  XfTrain* tr = m->train;
  const int d = m->d_model, Ms = Ts * B, Mt = Tt * B;
  auto body = [&](const float* src, const float* tgt, const float* expected, const float* text, const float* mask) {
    Run r{ctx, m, tr, s, B, tr->d_seed, backward ? cfg.dropout_p : 0.f, backward != 0};
    r.text = text;
    // off by default: measured 11.1-11.4 ms per step with the fork against 10.46-10.49 ms without (same box, graph replay) — the
    // 75 fork / join pairs cost more inside the graph than the overlapped 1.6 ms of dW kernels give back
    static const bool fork_env = getenv("SVG_TRAIN_FORK") && atoi(getenv("SVG_TRAIN_FORK")) != 0;
    r.fork = fork_env && backward && expected && !ctx->prof;
    LinTape e_src, e_tgt, l_out;
    XfDrop d_src, d_tgt;
    float* xs = r.embed(e_src, d_src, src, Ts, m->iota);
    float* xt = r.embed(e_tgt, d_tgt, tgt, Tt, m->iota);
    std::vector<EncTape> enc(m->enc_layers);
    std::vector<DecTape> dec(m->dec_layers);
    for (int i = 0; i < m->enc_layers; ++i) {
      const std::string p = "transformer.encoder.layers." + std::to_string(i) + ".";
      EncTape& t = enc[i];
      xs = r.add_ln(t.n1, xs, r.mha(t.sa, p + "self_attn.", xs, Ts, xs, Ts, nullptr, true), p + "norm1.", Ms, true);
      xs = r.add_ln(t.n2, xs, r.ffn(t.ff, p, xs, Ms), p + "norm2.", Ms, true);
    }
    LnTape n_enc, n_dec;
    float* mem = r.add_ln(n_enc, xs, nullptr, "transformer.encoder.norm.", Ms, false);
    for (int i = 0; i < m->dec_layers; ++i) {
      const std::string p = "transformer.decoder.layers." + std::to_string(i) + ".";
      DecTape& t = dec[i];
      xt = r.add_ln(t.n1, xt, r.mha(t.sa, p + "self_attn.", xt, Tt, xt, Tt, mask, true), p + "norm1.", Mt, true);
      xt = r.add_ln(t.n2, xt, r.mha(t.ca, p + "multihead_attn.", xt, Tt, mem, Ts, nullptr, false), p + "norm2.", Mt, true);
      xt = r.add_ln(t.n3, xt, r.ffn(t.ff, p, xt, Mt), p + "norm3.", Mt, true);
    }
    xt = r.add_ln(n_dec, xt, nullptr, "transformer.decoder.norm.", Mt, false);
    float* pred = r.lin(l_out, xt, "out.weight", "out.bias", Mt, m->d_lat, d);
    if (!expected) {                                           // forward only
      if (r.go()) HIP_OK(hipMemcpyAsync(pred_out, pred, (size_t)Mt * m->d_lat * sizeof(float), hipMemcpyDeviceToDevice, s));
      return;
    }

    // criterion on the last frames_to_predict positions (trainer.py:145)
    float* dpred = r.get<float>((int64_t)Mt * m->d_lat);
    float* part = r.get<float>((int64_t)Mt * 3);
    float* part2 = r.get<float>(Mt);
    if (r.go())
      xf_criterion(pred, expected, dpred, part, part2, tr->d_losses, Tt, B, m->d_lat, Tt - cfg.frames_to_predict, cfg.feat_h, cfg.feat_w, cfg.w_mse,
                   cfg.w_l1, cfg.w_gdl, cfg.gdl_alpha, cfg.w_contrastive, cfg.temperature, s);
    if (!backward) return;

    // ---- backward ----------------------------------------------------------------------------------------------------------
    float* dx = r.get<float>((int64_t)Mt * d);
    r.lin_bwd(l_out, dpred, dx);
    float* dxt = r.add_ln_bwd(n_dec, dx, nullptr);
    float* dmem = r.get<float>((int64_t)Ms * d);
    bool dmem_set = false;
    for (int i = m->dec_layers - 1; i >= 0; --i) {
      DecTape& t = dec[i];
      float *dzd = nullptr, *dz;
      dz = r.add_ln_bwd(t.n3, dxt, &dzd);                     // x2 + dropout(ff(x2))
      float* dx2 = r.get<float>((int64_t)Mt * d);
      r.ffn_bwd(t.ff, dzd, dx2, dz);
      dz = r.add_ln_bwd(t.n2, dx2, &dzd);                     // x1 + dropout(cross(x1, mem))
      float* dx1 = r.get<float>((int64_t)Mt * d);
      r.mha_bwd(t.ca, dzd, dx1, dz, dmem, dmem_set);
      dmem_set = true;
      dz = r.add_ln_bwd(t.n1, dx1, &dzd);                     // x + dropout(self(x))
      float* dx0 = r.get<float>((int64_t)Mt * d);
      r.mha_bwd(t.sa, dzd, dx0, dz, nullptr, false);
      dxt = dx0;
    }
    if (!dmem_set && r.go()) SDNS::fill_f32(dmem, (int64_t)Ms * d, 0.f, s);
    float* dxs = r.add_ln_bwd(n_enc, dmem, nullptr);
    for (int i = m->enc_layers - 1; i >= 0; --i) {
      EncTape& t = enc[i];
      float *dzd = nullptr, *dz;
      dz = r.add_ln_bwd(t.n2, dxs, &dzd);
      float* dx1 = r.get<float>((int64_t)Ms * d);
      r.ffn_bwd(t.ff, dzd, dx1, dz);
      dz = r.add_ln_bwd(t.n1, dx1, &dzd);
      float* dx0 = r.get<float>((int64_t)Ms * d);
      r.mha_bwd(t.sa, dzd, dx0, dz, nullptr, false);
      dxs = dx0;
    }
    r.embed_bwd(e_src, d_src, dxs, Ts, false);
    r.embed_bwd(e_tgt, d_tgt, dxt, Tt, true);                 // the embedding layer is shared: second contribution accumulates
    r.join();
  };
  // ---- the plan of this call signature ----------------------------------------------------------------------------------------
  struct ArenaSwap {       // the training workspace stands in for the context arena until this call returns (or throws)
    svg_ctx* c; XfTrain* t;
    ArenaSwap(svg_ctx* c_, XfTrain* t_) : c(c_), t(t_) { std::swap(c->arena, t->ws); std::swap(c->arena_buf, t->ws_buf); }
    ~ArenaSwap() { std::swap(c->arena, t->ws); std::swap(c->arena_buf, t->ws_buf); }
  } arena_swap(ctx, tr);
  const int mode = expected ? 0 : 2;
  XfTrain::Plan* pl = nullptr;
  for (auto& c : tr->plans)
    if (c.B == B && c.Ts == Ts && c.Tt == Tt && c.backward == backward && c.mode == mode && c.has_mask == (mask != nullptr) &&
        c.has_text == (text != nullptr) && c.cfg.frames_to_predict == cfg.frames_to_predict && c.cfg.feat_h == cfg.feat_h &&
        c.cfg.feat_w == cfg.feat_w && c.cfg.w_mse == cfg.w_mse && c.cfg.w_l1 == cfg.w_l1 && c.cfg.w_gdl == cfg.w_gdl &&
        c.cfg.gdl_alpha == cfg.gdl_alpha && c.cfg.w_contrastive == cfg.w_contrastive && c.cfg.temperature == cfg.temperature &&
        c.cfg.dropout_p == cfg.dropout_p) { pl = &c; break; }
  if (!pl) {
    tr->plans.emplace_back();
    pl = &tr->plans.back();
    pl->cfg = cfg; pl->B = B; pl->Ts = Ts; pl->Tt = Tt; pl->backward = backward; pl->mode = mode;
    pl->has_mask = mask != nullptr; pl->has_text = text != nullptr;
    ctx->arena.reset(); ctx->arena.dry = true; ctx->arena.high = 0;          // dry pass: the workspace this signature needs
    try { body(src, tgt, expected, text, mask); } catch (...) { ctx->arena.dry = false; tr->plans.pop_back(); throw; }
    ctx->arena.dry = false;
    pl->high = ctx->arena.high;
  }
  ctx->ensure_arena(pl->high);
  tr->seed_slot = (tr->seed_slot + 1) & 15;
  if (!tr->seed_ev[tr->seed_slot]) HIP_OK(hipEventCreateWithFlags(&tr->seed_ev[tr->seed_slot], hipEventDisableTiming));
  else HIP_OK(hipEventSynchronize(tr->seed_ev[tr->seed_slot]));       // the slot's previous upload (16 calls ago) has been read
  tr->h_seed[tr->seed_slot] = cfg.seed;
  HIP_OK(hipMemcpyAsync(tr->d_seed, tr->h_seed + tr->seed_slot, sizeof(uint64_t), hipMemcpyHostToDevice, s));
  HIP_OK(hipEventRecord(tr->seed_ev[tr->seed_slot], s));
  static const bool graph_env = !(getenv("SVG_TRAIN_GRAPH") && atoi(getenv("SVG_TRAIN_GRAPH")) == 0);
  // hipGraph replay of the whole step: needs a capturable (non-null) stream; the inputs go through fixed staging buffers
  if (graph_env && expected && s != nullptr && !ctx->prof) {
    if (!pl->src) {
      pl->src = (float*)tr->dalloc((int64_t)B * Ts * m->d_lat * sizeof(float));
      pl->tgt = (float*)tr->dalloc((int64_t)B * Tt * m->d_lat * sizeof(float));
      pl->exp = (float*)tr->dalloc((int64_t)B * Tt * m->d_lat * sizeof(float));
      if (text) pl->text = (float*)tr->dalloc((int64_t)B * m->text_dim * sizeof(float));
      if (mask) pl->mask = (float*)tr->dalloc((int64_t)Tt * Tt * sizeof(float));
    }
    HIP_OK(hipMemcpyAsync(pl->src, src, (size_t)B * Ts * m->d_lat * sizeof(float), hipMemcpyDefault, s));
    HIP_OK(hipMemcpyAsync(pl->tgt, tgt, (size_t)B * Tt * m->d_lat * sizeof(float), hipMemcpyDefault, s));
    HIP_OK(hipMemcpyAsync(pl->exp, expected, (size_t)B * Tt * m->d_lat * sizeof(float), hipMemcpyDefault, s));
    if (text) HIP_OK(hipMemcpyAsync(pl->text, text, (size_t)B * m->text_dim * sizeof(float), hipMemcpyDefault, s));
    if (mask) HIP_OK(hipMemcpyAsync(pl->mask, mask, (size_t)Tt * Tt * sizeof(float), hipMemcpyDefault, s));
    if (!pl->exec || pl->arena_base != ctx->arena.base) {       // first use, or the workspace moved since the capture
      if (pl->exec) { HIP_OK(hipGraphExecDestroy(pl->exec)); pl->exec = nullptr; }
      hipGraph_t g = nullptr;
      {
        CaptureScope cap;             // shared with other captures, exclusive against device-wide syncs (common.h)
        HIP_OK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        try {
          ctx->arena.reset();
          body(pl->src, pl->tgt, pl->exp, pl->text, pl->mask);
        } catch (...) {
          hipStreamEndCapture(s, &g);
          if (g) hipGraphDestroy(g);
          throw;
        }
        HIP_OK(hipStreamEndCapture(s, &g));
      }
      const hipError_t e = hipGraphInstantiate(&pl->exec, g, nullptr, nullptr, 0);
      hipGraphDestroy(g);
      HIP_OK(e);
      pl->arena_base = ctx->arena.base;
    }
    HIP_OK(hipGraphLaunch(pl->exec, s));
  } else {
    ctx->arena.reset();
    body(src, tgt, expected, text, mask);
  }
  if (losses_host) {
    HIP_OK(hipMemcpyAsync(tr->h_losses, tr->d_losses, 5 * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    memcpy(losses_host, tr->h_losses, 5 * sizeof(float));
  }
}

}  // namespace

// caller holds DeviceWideScope when m->train exists (svg_destroy, svg_model_configure, svg_load_weight)
void xf_train_free(XfModel* m) {
  if (!m->train) return;
  hipDeviceSynchronize();
  delete m->train;
  m->train = nullptr;
}

extern "C" int svg_transformer_loss(svg_ctx* ctx, const svg_train_cfg* cfg, const float* src, const float* tgt, const float* expected,
                                    const float* text, int B, int Ts, int Tt, const float* mask, int backward, float* losses, void* stream) {
  try {
    SVG_CHECK(ctx && ctx->xf, "transformer: model not configured");
    SVG_CHECK(cfg && src && tgt && expected, "svg_transformer_loss: null argument");
    loss_pass(ctx, ctx->xf, *cfg, src, tgt, expected, text, B, Ts, Tt, mask, backward, losses, (hipStream_t)stream);
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}

extern "C" int svg_transformer_forward_train(svg_ctx* ctx, const float* src, const float* tgt, const float* text, int B, int Ts, int Tt,
                                             const float* mask, float dropout_p, uint64_t seed, float* out, void* stream) {
  try {
    SVG_CHECK(ctx && ctx->xf, "transformer: model not configured");
    SVG_CHECK(src && tgt && out, "svg_transformer_forward_train: null argument");
    svg_train_cfg cfg{};
    cfg.dropout_p = dropout_p;
    cfg.seed = seed;
    loss_pass(ctx, ctx->xf, cfg, src, tgt, nullptr, text, B, Ts, Tt, mask, /*train mode*/ 1, nullptr, (hipStream_t)stream, out);
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}

extern "C" int svg_transformer_adam_step(svg_ctx* ctx, float lr, float beta1, float beta2, float eps, void* stream) {
  try {
    SVG_CHECK(ctx && ctx->xf && ctx->xf->train, "adam step: no gradients yet (call svg_transformer_loss with backward=1 first)");
    SVG_CHECK(lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "adam step: bad hyper-parameters");
    XfTrain* tr = ctx->xf->train;
    tr->step += 1;
    xf_adam(tr->d_tens, tr->d_chunks, tr->n_chunks, lr, beta1, beta2, eps, tr->step, (hipStream_t)stream);
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}

extern "C" int svg_transformer_tensor(svg_ctx* ctx, int kind, const char* name, float* out, int64_t numel, void* stream) {
  try {
    SVG_CHECK(ctx && ctx->xf && name && out, "svg_transformer_tensor: null argument");
    XfModel* m = ctx->xf;
    SVG_CHECK(m->ws.has(name), "svg_transformer_tensor: no tensor named %s", name);
    const Weight& w = m->ws.get(name);
    SVG_CHECK(numel == w.numel, "svg_transformer_tensor: %s has %lld elements, caller asked for %lld", name, (long long)w.numel, (long long)numel);
    const float* srcp = w.f32;
    if (kind != SVG_TENSOR_PARAM) {
      SVG_CHECK(kind == SVG_TENSOR_GRAD || kind == SVG_TENSOR_EXP_AVG || kind == SVG_TENSOR_EXP_AVG_SQ, "svg_transformer_tensor: kind %d", kind);
      SVG_CHECK(m->train && m->train->slots.count(name), "svg_transformer_tensor: %s has no training state", name);
      const XfTrain::Slot& sl = m->train->slots.at(name);
      srcp = kind == SVG_TENSOR_GRAD ? sl.g : (kind == SVG_TENSOR_EXP_AVG ? sl.m : sl.v);
    }
    HIP_OK(hipMemcpyAsync(out, srcp, numel * sizeof(float), hipMemcpyDefault, (hipStream_t)stream));
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}

extern "C" int svg_op_dropout_mask(svg_ctx* ctx, uint64_t seed, int site, float p, float* out, int64_t n, void* stream) {
  try {
    SVG_CHECK(ctx && out && n >= 0 && p >= 0.f && p < 1.f && site >= 0, "svg_op_dropout_mask: bad argument");
    if (!ctx->seed_scratch) ctx->seed_scratch = (uint64_t*)ctx->dalloc(sizeof(uint64_t));
    HIP_OK(hipMemcpyAsync(ctx->seed_scratch, &seed, sizeof(uint64_t), hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));            // `seed` is a stack word
    xf_drop_mask(XfDrop{ctx->seed_scratch, (uint32_t)site, p}, out, n, (hipStream_t)stream);
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}
