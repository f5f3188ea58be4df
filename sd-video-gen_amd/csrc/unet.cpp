// SD UNet2DConditionModel graph (diffusers 0.2.x layout) and the DDIM img2img loop
// (utils/sd_utils.py:222-267) on NHWC bf16 activations.
#include "models.h"
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <thread>

namespace SDNS {

void UnetModel::configure(const char* kv) {
  auto m = parse_kv(kv);
  auto geti = [&](const char* k, int& dst) { if (m.count(k)) dst = (int)m[k][0]; };
  if (m.count("block_out")) { block_out.clear(); for (auto v : m["block_out"]) block_out.push_back((int)v); }
  if (m.count("attn")) { attn.clear(); for (auto v : m["attn"]) attn.push_back((int)v); }
  geti("layers", layers); geti("heads", heads); geti("ctx_dim", ctx_dim); geti("groups", groups);
  geti("in_ch", in_ch); geti("out_ch", out_ch);
  fp8 = 0; geti("fp8", fp8);
  ready = false;
}

namespace {

struct TembCollector {
  std::vector<std::string> prefixes;
  std::vector<int> couts;
  int total = 0;
  int add(const std::string& p, int cout) {
    int off = total;
    prefixes.push_back(p); couts.push_back(cout); total += cout;
    return off;
  }
};

// Placement masks of the MX-fp8 convs: bit i (0..3) down_blocks.i | bit 4 mid_block | bit 5 + i up_blocks.i | bit 9 the upsamplers |
// bit 10 conv1 | bit 11 conv2 of a resnet.  $SVG_FP8_SITES (load time: which convs get an e4m3 copy at all) and $SVG_FP8_SITES_GUIDED
// (run time: which of them a classifier-free-guided DDIM loop may use).  Guidance 7.5 multiplies the error of (cond - uncond) by 7.5: with
// every eligible conv in e4m3 the 50-step guided loop ends 0.29 from the fp16 path (0.18 from the fp32 oracle over four frames), and the
// error follows the NUMBER of e4m3 convs, not a particular block (profiles/r05_fp8_sites.txt: 33 convs 0.29, 21-27 convs 0.14-0.26, 12 convs
// 0.12, 10 convs 0.06).  The placement that keeps the guided loop under 1e-1 is the 16 x 16 level alone (10 convs, K = 11 520 ... 23 040) —
// the default under guidance; guidance 0 (configs[2]) keeps every eligible conv (4.2e-2 after 50 steps).
constexpr int64_t kFp8SitesDefault = 0xFFF;                                         // every eligible conv
constexpr int64_t kFp8SitesGuided = (1 << 2) | (1 << 6) | (1 << 10) | (1 << 11);     // down_blocks.2 + up_blocks.1: the 16 x 16 level
ResW load_res_t(svg_ctx* ctx, WeightStore& ws, const std::string& p, int cin, int cout, TembCollector& tc, hipStream_t s, bool fp8, bool fp8_c2) {
  ResW r;
  r.n1 = load_norm(ctx, ws, p + ".norm1", cin);
  r.c1 = load_conv3x3(ctx, ws, p + ".conv1", cin, cout, s, fp8);
  r.n2 = load_norm(ctx, ws, p + ".norm2", cout);
  r.c2 = load_conv3x3(ctx, ws, p + ".conv2", cout, cout, s, fp8_c2);
  r.has_sc = cin != cout;
  if (r.has_sc) r.sc = load_linear(ctx, ws, p + ".conv_shortcut", cout, cin, true, s);
  r.temb_off = tc.add(p + ".time_emb_proj", cout);
  return r;
}

// rows [W0; W1; ...] of same K packed into one bf16 matrix
PackedLinear load_stacked(svg_ctx* ctx, WeightStore& ws, const std::vector<std::string>& names, const std::vector<int>& ns, int K,
                          bool bias, hipStream_t s, const NormW* fold = nullptr, bool release = true) {
  PackedLinear pl;
  int N = 0;
  for (int n : ns) N += n;
  pl.N = (int)align_up(N, 4); pl.K = K; pl.n_valid = N;
  pl.w = (h16*)ctx->dalloc((int64_t)pl.N * K * sizeof(h16));
  HIP_OK(hipMemsetAsync(pl.w, 0, (size_t)pl.N * K * sizeof(h16), s));
  if (bias || fold) {
    pl.b = (float*)ctx->dalloc(pl.N * sizeof(float));
    HIP_OK(hipMemsetAsync(pl.b, 0, pl.N * sizeof(float), s));
  }
  int off = 0;
  for (size_t i = 0; i < names.size(); ++i) {
    const Weight& w = ws.get(names[i] + ".weight");
    SVG_CHECK(w.numel == (int64_t)ns[i] * K, "weight %s.weight: expected [%d,%d]", names[i].c_str(), ns[i], K);
    if (bias)
      HIP_OK(hipMemcpyAsync(pl.b + off, keep_f32(ctx, ws, names[i] + ".bias", ns[i]), ns[i] * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (fold) fold_ln_weights(w.f32, pl.b + off, fold->g, fold->b, pl.b + off, ns[i], K, s);
    pack_linear(w.f32, pl.w + (int64_t)off * K, ns[i], K, ns[i], s);
    off += ns[i];
  }
  if (fold) {
    pl.ln_s = (float*)ctx->dalloc(pl.N * sizeof(float));
    rowsum_h16(pl.w, pl.ln_s, pl.N, K, s);
  }
  HIP_OK(hipStreamSynchronize(s));
  if (release)
    for (auto& n : names) ws.release(n + ".weight");
  return pl;
}

// SVG_LN_FOLD=0 keeps the three LayerNorms of a transformer block as separate kernels (A/B and debugging)
bool ln_fold_enabled() {
  static const int on = getenv("SVG_LN_FOLD") ? atoi(getenv("SVG_LN_FOLD")) : 1;
  return on != 0;
}

// rows per ff1 -> ff2 chunk: SVG_FF_CHUNK_MB bounds the chunk's GEGLU intermediate (rows x 4C x 2 bytes); 0 = unchunked.
// Rounded to a multiple of 4096 rows (whole 256-row tiles of gemm_pp, whole samples at 64 x 64).
int ff_chunk_rows(int M, int C) {
  static const int mb = getenv("SVG_FF_CHUNK_MB") ? atoi(getenv("SVG_FF_CHUNK_MB")) : 0;
  if (mb <= 0) return M;
  const int64_t rows = ((int64_t)mb << 20) / ((int64_t)8 * C);
  return (int)std::max<int64_t>(4096, std::min<int64_t>(M, rows / 4096 * 4096));
}

XfBlockW load_xf(svg_ctx* ctx, WeightStore& ws, const std::string& p, int C, int ctx_dim, hipStream_t s) {
  XfBlockW b;
  b.C = C;
  const bool fold = ln_fold_enabled();
  b.gn = load_norm(ctx, ws, p + ".norm", C);
  {   // f32 copy of proj_in (GroupNorm folding rounds W * gamma * rstd once, per sample, at run time)
    const Weight& w = ws.get(p + ".proj_in.weight");
    SVG_CHECK(w.numel == (int64_t)C * C, "weight %s.proj_in.weight: expected [%d,%d(,1,1)]", p.c_str(), C, C);
    b.proj_in_f32 = (float*)ctx->dalloc((int64_t)C * C * sizeof(float));
    HIP_OK(hipMemcpy(b.proj_in_f32, w.f32, (size_t)C * C * sizeof(float), hipMemcpyDeviceToDevice));
  }
  b.proj_in = load_linear(ctx, ws, p + ".proj_in", C, C, true, s);
  b.proj_out = load_linear(ctx, ws, p + ".proj_out", C, C, true, s);
  const std::string t = p + ".transformer_blocks.0";
  b.ln1 = load_norm(ctx, ws, t + ".norm1", C);
  b.ln2 = load_norm(ctx, ws, t + ".norm2", C);
  b.ln3 = load_norm(ctx, ws, t + ".norm3", C);
  // C = 320 (the 64 x 64 level): q | k | v stacked for the fused projection of gemm_ws.hip (one read of the tokens instead of two).
  // fold_ln_weights scales the f32 copy in place, so the stacked copy is packed from a scratch duplicate of the three matrices
  if (fold && C == 320) {
    WeightStore tmp;
    for (const char* n : {".attn1.to_q", ".attn1.to_k", ".attn1.to_v"}) {
      const Weight& w = ws.get(t + n + ".weight", {C, C});
      std::vector<int64_t> shp{C, C};
      tmp.put(ctx, t + n + ".weight", w.f32, shp.data(), 2);
    }
    b.qkv1 = load_stacked(ctx, tmp, {t + ".attn1.to_q", t + ".attn1.to_k", t + ".attn1.to_v"}, {C, C, C}, C, false, s, &b.ln1);
    tmp.clear();
  }
  b.qk1 = load_stacked(ctx, ws, {t + ".attn1.to_q", t + ".attn1.to_k"}, {C, C}, C, false, s, fold ? &b.ln1 : nullptr);
  b.v1 = load_linear(ctx, ws, t + ".attn1.to_v", C, C, false, s, fold ? &b.ln1 : nullptr);
  b.o1 = load_linear(ctx, ws, t + ".attn1.to_out.0", C, C, true, s);
  if (fold && C == 320) {   // packed copies for the one-launch cross-attention (xattn_fused.hip), made before load_linear releases the f32 originals
    const Weight& wq = ws.get(t + ".attn2.to_q.weight", {C, C});
    const Weight& wo = ws.get(t + ".attn2.to_out.0.weight", {C, C});
    b.xq = (h16*)ctx->dalloc((int64_t)384 * C * sizeof(h16));
    b.xq_s = (float*)ctx->dalloc(384 * sizeof(float));
    b.xq_b = (float*)ctx->dalloc(384 * sizeof(float));
    b.xo = (h16*)ctx->dalloc((int64_t)C * 384 * sizeof(h16));
    b.xqp = (h16*)ctx->dalloc((int64_t)384 * C * sizeof(h16));
    xattn_pack_q(wq.f32, b.ln2.g, b.ln2.b, b.xq, b.xq_s, b.xq_b, 0, s);
    xattn_pack_q(wq.f32, b.ln2.g, b.ln2.b, b.xqp, b.xq_s, b.xq_b, 1, s);      // (row sums and W beta do not depend on the order)
    xattn_pack_o(wo.f32, b.xo, s);
    HIP_OK(hipStreamSynchronize(s));
  }
  b.q2 = load_linear(ctx, ws, t + ".attn2.to_q", C, C, false, s, fold ? &b.ln2 : nullptr);
  b.k2 = load_linear(ctx, ws, t + ".attn2.to_k", C, ctx_dim, false, s);
  b.v2 = load_linear(ctx, ws, t + ".attn2.to_v", C, ctx_dim, false, s);
  b.o2 = load_linear(ctx, ws, t + ".attn2.to_out.0", C, C, true, s);
  // GEGLU: Linear(C -> 8C) rows = [h (4C); gate (4C)] -> 16-row tiles alternating h / gate
  {
    const int F = 4 * C;
    const Weight& w = ws.get(t + ".ff.net.0.proj.weight", {2 * F, C});
    b.ff1.N = 2 * F; b.ff1.K = C; b.ff1.n_valid = 2 * F;
    b.ff1.w = (h16*)ctx->dalloc((int64_t)2 * F * C * sizeof(h16));
    b.ff1.b = (float*)ctx->dalloc(2 * F * sizeof(float));
    float* bias = keep_f32(ctx, ws, t + ".ff.net.0.proj.bias", 2 * F);
    float* btmp = nullptr;
    if (fold) {   // fold LayerNorm 3 on the unpacked rows, then pack weights and bias together
      HIP_OK(hipMalloc(&btmp, (size_t)2 * F * sizeof(float)));
      fold_ln_weights(w.f32, bias, b.ln3.g, b.ln3.b, btmp, 2 * F, C, s);
      bias = btmp;
    }
    pack_geglu(w.f32, bias, b.ff1.w, b.ff1.b, F, C, s);
    if (fold) {
      b.ff1.ln_s = (float*)ctx->dalloc((size_t)2 * F * sizeof(float));
      rowsum_h16(b.ff1.w, b.ff1.ln_s, 2 * F, C, s);
    }
    HIP_OK(hipStreamSynchronize(s));
    if (btmp) HIP_OK(hipFree(btmp));
    ws.release(t + ".ff.net.0.proj.weight");
  }
  if (ff_fused_supported(C, 1 << 30)) {   // k-permuted copy of ff.net.2 for the fused feed-forward
    const Weight& w2 = ws.get(t + ".ff.net.2.weight", {C, 4 * C});
    b.ff2p = (h16*)ctx->dalloc((int64_t)C * 4 * C * sizeof(h16));
    pack_ff2_perm(w2.f32, b.ff2p, C, 4 * C, s);
    HIP_OK(hipStreamSynchronize(s));
  }
  b.ff2 = load_linear(ctx, ws, t + ".ff.net.2", C, 4 * C, true, s);
  return b;
}

}  // namespace

void UnetModel::finalize(svg_ctx* ctx, int64_t* n_params) {
  hipStream_t s = nullptr;
  const int nb = (int)block_out.size();
  SVG_CHECK((int)attn.size() == nb, "unet: attn flags must have one entry per block");
  const int64_t total = ws.total_params();
  const int c0 = block_out[0];
  temb_dim = 4 * c0;
  TembCollector tc;
  // fp8=1: the resnets' 3x3 convs get an MX fp8 copy (conv_halo_fp8.hip) — half of the UNet's FLOPs, K = 2 880 ... 23 040.
  // $SVG_FP8_CONV=0 / $SVG_FP8_PROJ=0 switch the two fp8 placements off separately (A/B)
  const bool fp8_conv = fp8 && svg_env_i64("SVG_FP8_CONV", 1) != 0;
  const bool fp8_proj = fp8 && svg_env_i64("SVG_FP8_PROJ", 0) != 0;
  // Placement of the MX-fp8 convs ($SVG_FP8_SITES, a bit mask; round 5: per-site sensitivity of the guidance-7.5 loop):
  //   bit i (0..3) down_blocks.i | bit 4 mid_block | bit 5 + i up_blocks.i | bit 9 the upsamplers | bit 10 conv1 | bit 11 conv2 of a resnet
  const int64_t sites = svg_env_i64("SVG_FP8_SITES", kFp8SitesDefault);
  auto f8 = [&](int bit, int conv) { return fp8_conv && ((sites >> bit) & 1) && ((sites >> (10 + conv)) & 1); };
  time1 = load_linear(ctx, ws, "time_embedding.linear_1", temb_dim, c0, true, s);
  time2 = load_linear(ctx, ws, "time_embedding.linear_2", temb_dim, temb_dim, true, s);
  conv_in = load_conv3x3(ctx, ws, "conv_in", in_ch, c0, s);
  down_res.clear(); down_attn.clear(); down_s.clear(); up_res.clear(); up_attn.clear(); up_s.clear();
  // ---- down
  std::vector<int> skip_ch{c0};
  int cin = c0;
  for (int i = 0; i < nb; ++i) {
    std::vector<ResW> rs; std::vector<XfBlockW> as;
    const std::string bp = "down_blocks." + std::to_string(i);
    for (int j = 0; j < layers; ++j) {
      rs.push_back(load_res_t(ctx, ws, bp + ".resnets." + std::to_string(j), cin, block_out[i], tc, s, f8(i, 0), f8(i, 1)));
      rs.back().site = i;
      cin = block_out[i];
      if (attn[i]) as.push_back(load_xf(ctx, ws, bp + ".attentions." + std::to_string(j), cin, ctx_dim, s));
      skip_ch.push_back(cin);
    }
    down_res.push_back(rs); down_attn.push_back(as);
    if (i < nb - 1) {
      down_s.push_back(load_conv3x3(ctx, ws, bp + ".downsamplers.0.conv", cin, cin, s));
      skip_ch.push_back(cin);
    }
  }
  // ---- mid
  mid0 = load_res_t(ctx, ws, "mid_block.resnets.0", cin, cin, tc, s, f8(4, 0), f8(4, 1));
  mid_attn = load_xf(ctx, ws, "mid_block.attentions.0", cin, ctx_dim, s);
  mid1 = load_res_t(ctx, ws, "mid_block.resnets.1", cin, cin, tc, s, f8(4, 0), f8(4, 1));
  mid0.site = mid1.site = 4;
  // ---- up (reversed block_out; layers+1 resnets per block, each consuming one skip)
  for (int i = 0; i < nb; ++i) {
    const int bi = nb - 1 - i;
    const int cout = block_out[bi];
    std::vector<ResW> rs; std::vector<XfBlockW> as;
    const std::string bp = "up_blocks." + std::to_string(i);
    for (int j = 0; j < layers + 1; ++j) {
      const int sc = skip_ch.back(); skip_ch.pop_back();
      rs.push_back(load_res_t(ctx, ws, bp + ".resnets." + std::to_string(j), cin + sc, cout, tc, s, f8(5 + i, 0), f8(5 + i, 1)));
      rs.back().site = 5 + i;
      cin = cout;
      if (attn[bi]) as.push_back(load_xf(ctx, ws, bp + ".attentions." + std::to_string(j), cin, ctx_dim, s));
    }
    up_res.push_back(rs); up_attn.push_back(as);
    if (i < nb - 1) up_s.push_back(load_conv3x3(ctx, ws, bp + ".upsamplers.0.conv", cin, cin, s, fp8_conv && ((sites >> 9) & 1)));
  }
  if (fp8_proj) {
    // (round 2-3 placement, off by default since round 4: does not pay, profiles/r03_bench_line_fp8.json) the dense projections that qualify get an MX fp8 copy (attention out-projections, ff.net.2, proj_out,
    // 1x1 shortcuts, cross-attention k / v at the 32 x 32 level and below: K = 640 ... 5120); the 64 x 64 level (K = 320) and
    // every LayerNorm-folded / GEGLU projection stay bf16, as do the convolutions
    auto add_xf = [&](XfBlockW& b) {
      for (PackedLinear* pl : {&b.proj_out, &b.o1, &b.o2, &b.ff2, &b.k2, &b.v2}) add_fp8_copy(ctx, *pl, s);
    };
    for (auto& v : down_attn) for (auto& b : v) add_xf(b);
    for (auto& v : up_attn) for (auto& b : v) add_xf(b);
    add_xf(mid_attn);
    for (auto& v : down_res) for (auto& r : v) if (r.has_sc) add_fp8_copy(ctx, r.sc, s);
    for (auto& v : up_res) for (auto& r : v) if (r.has_sc) add_fp8_copy(ctx, r.sc, s);
  }
  norm_out = load_norm(ctx, ws, "conv_norm_out", c0);
  conv_out = load_conv3x3(ctx, ws, "conv_out", c0, out_ch, s);
  temb_all = load_stacked(ctx, ws, tc.prefixes, tc.couts, temb_dim, true, s);
  // ---- DDIM table, as diffusers' DDIMScheduler builds it: betas = linspace(sqrt(b0), sqrt(b1), 1000, f32)**2,
  // alphas_cumprod = cumprod(1 - betas) in f32
  alphas_cumprod.resize(1000);
  {
    const double b0 = sqrt(0.00085), b1 = sqrt(0.012);
    float prod = 1.f;
    for (int i = 0; i < 1000; ++i) {
      const double step = (b1 - b0) / 999.0;
      float sb = (float)(i * step + b0);
      if (i == 999) sb = (float)b1;
      float beta = sb * sb;
      float alpha = 1.f - beta;
      prod = prod * alpha;
      alphas_cumprod[i] = prod;
    }
  }
  if (n_params) *n_params = total;
  ready = true;
}

void UnetModel::ddim_coefs(int t, int t_prev, float* sa, float* s1a, float* sap, float* s1ap) const {
  SVG_CHECK(t >= 0 && t < 1000, "ddim: timestep %d out of range", t);
  const float a_t = alphas_cumprod[t];
  const float a_p = (t_prev >= 0) ? alphas_cumprod[t_prev] : 1.0f;   // set_alpha_to_one
  *sa = sqrtf(a_t); *s1a = sqrtf(1.f - a_t); *sap = sqrtf(a_p); *s1ap = sqrtf(1.f - a_p);
}

namespace {
struct UnetRun {
  svg_ctx* ctx; UnetModel* m; hipStream_t s; int N;
  const h16* ctxb; int L, Lp;     // context (N*L, ctx_dim) bf16; Lp = L padded to 8
  const float* temb;               // (N, temb_total) f32: every resnet's time_emb_proj(silu(temb))
  int temb_ld;
  KvCache* cache = nullptr;        // cross-attention K / V^T reuse across DDIM steps (constant context)
  int64_t sites = -1;              // placement mask of the MX-fp8 convs for THIS call (a parameter of the call, not model state)
  int xf_idx = 0;

  // an activation tensor with, when its producer could leave them, the GroupNorm column sums of its row tiles
  struct Act {
    const h16* p = nullptr;
    int C = 0;
    GnStats st;
  };
  GnEmit emit_for(int64_t hw, int Cout) {
    GnEmit e;
    if (hw >= 1024) e.buf = ctx->arena.get<float>(gn_part_floats(N, hw, Cout));
    return e;
  }

  // ResnetBlock2D; `skip` (up path): the input is torch.cat([x, skip], dim=1) — never materialised when both tensors carry
  // their column sums: GroupNorm reads the two sources, the 1x1 shortcut takes a two-source A operand
  Act resnet(const Act& x, const Act* skip, const ResW& r, int H, int W) {
    const int64_t P = (int64_t)N * H * W;
    const int HW = H * W;
    const int Cx = x.C, Cs = skip ? skip->C : 0, Cin = Cx + Cs;
    Act out;
    out.C = r.c2.Cout;
    h16* outp = ctx->arena.get<h16>(P * r.c2.Opad);
    GnEmit eo = emit_for(HW, r.c2.Opad);
    ctx->arena.push();
    const bool virt = skip && x.st.valid() && skip->st.valid() && Cx % 64 == 0 && r.has_sc;   // virtual concat
    const h16* xin = x.p;
    if (skip && !virt) {   // torch.cat([hidden, skip], dim=1)
      h16* cat = ctx->arena.get<h16>(P * Cin);
      if (SVG_LAUNCHING(ctx)) { ProfScope ps(ctx, PK_ELT, s, 0, 4.0 * P * Cin); concat_channels(x.p, Cx, skip->p, Cs, cat, P, s); }
      xin = cat;
    }
    h16* t1 = ctx->arena.get<h16>(P * r.c1.Opad);
    GnEmit e1 = emit_for(HW, r.c1.Opad);
    // GroupNorm + SiLU -> conv.  fp8=1 (MX fp8 convs, conv_halo_fp8.hip): the normalised tensor is written as e4m3 + one E8M0 scale per 32
    // channels by the apply pass itself (gn_apply_mx) — or, where the statistics do not come from an epilogue, by a quantising pass over
    // the 16-bit tensor — and the conv runs on v_mfma_scale_f32_16x16x128_f8f6f4 at twice the 16-bit matrix rate.
    auto norm_conv = [&](const h16* a, int Ca, const h16* a2, int Ca2, const NormW& nw, const GnStats* s1, const GnStats* s2, const ConvW& cw, h16* o,
                         const float* bbn, int bbn_ld, const h16* resid, GnEmit* e, int conv_bit) {
      const int Cn = Ca + Ca2;
      ctx->arena.push();
      const bool site_on = ((sites >> r.site) & 1) && ((sites >> conv_bit) & 1);      // this call's placement mask
      if (site_on && conv3x3_fp8_ok(cw, N, H, W)) {
        const int64_t Cp = align_up(Cn, 128);
        uint8_t* q = ctx->arena.get<uint8_t>(P * Cp);
        uint8_t* qs = ctx->arena.get<uint8_t>(P * (Cp / 32));
        if (!groupnorm_mx(ctx, a, Ca, a2, Ca2, nw.g, nw.b, q, qs, N, HW, m->groups, 1e-5f, 1, s, s1, s2)) {
          h16* t = ctx->arena.get<h16>(P * Cn);
          groupnorm(ctx, a, Ca, a2, Ca2, nw.g, nw.b, t, N, HW, m->groups, 1e-5f, 1, s, s1, s2);
          quant_act_mx(ctx, t, Cn, q, qs, P, s);
        }
        conv3x3_fp8(ctx, q, qs, cw, o, N, H, W, bbn, bbn_ld, resid, s, e);
      } else {
        h16* t = ctx->arena.get<h16>(P * Cn);
        groupnorm(ctx, a, Ca, a2, Ca2, nw.g, nw.b, t, N, HW, m->groups, 1e-5f, 1, s, s1, s2);
        conv3x3(ctx, t, cw, o, N, H, W, A_CONV_S1, bbn, bbn_ld, resid, 0, s, e);
      }
      ctx->arena.pop();
    };
    if (virt) norm_conv(x.p, Cx, skip->p, Cs, r.n1, &x.st, &skip->st, r.c1, t1, temb + r.temb_off, temb_ld, nullptr, &e1, 10);
    else norm_conv(xin, Cin, nullptr, 0, r.n1, skip ? nullptr : &x.st, nullptr, r.c1, t1, temb + r.temb_off, temb_ld, nullptr, &e1, 10);
    const h16* res = xin;
    if (r.has_sc) {
      h16* sc = ctx->arena.get<h16>(P * r.sc.N);
      if (virt) linear(ctx, x.p, Cx, r.sc, sc, r.sc.N, (int)P, ACT_NONE, nullptr, 0, 0, s, nullptr, nullptr, nullptr, 0, skip->p, Cs, Cx);
      else linear(ctx, xin, Cin, r.sc, sc, r.sc.N, (int)P, ACT_NONE, nullptr, 0, 0, s);
      res = sc;
    }
    norm_conv(t1, r.n2.C, nullptr, 0, r.n2, &e1.st, nullptr, r.c2, outp, nullptr, 0, res, &eo, 11);
    ctx->arena.pop();
    out.p = outp;
    out.st = eo.st;
    return out;
  }

  void attn_core(const h16* q, int ldq, const h16* k, int ldk, int64_t kb, const h16* vt, int ldvt, int64_t vtb, h16* o,
                 int C, int Sq, int Skv) {
    AttnArgs a;
    a.q = q; a.k = k; a.vt = vt; a.out = o;
    a.B = N; a.heads = m->heads; a.Sq = Sq; a.Skv = Skv; a.d = C / m->heads;
    a.ldq = ldq; a.ldk = ldk; a.ldvt = ldvt; a.ldo = C;
    a.qb = (int64_t)Sq * ldq; a.kb = kb; a.vtb = vtb; a.ob = (int64_t)Sq * C;
    a.scale = 1.f / sqrtf((float)a.d);
    attention(ctx, a, s);
  }

  // V^T[b] (C x SkvPad) = Wv * src_b^T
  void vt_proj_into(const PackedLinear& wv, const h16* src, int rows, int rows_pad, int K, h16* vt,
                    const float* ln_rs = nullptr, const float* ln_rm = nullptr) {
    const int C = wv.N;
    GemmArgs g;
    g.A = wv.w; g.lda = K; g.Wt = src; g.ldb = K; g.M = C; g.N = rows_pad; g.n_valid = rows; g.K = K;
    g.batch = N; g.sA = 0; g.sB = (int64_t)rows * K; g.sC = (int64_t)C * rows_pad;
    g.C = vt; g.ldc = rows_pad;
    SVG_CHECK((wv.ln_s != nullptr) == (ln_rs != nullptr), "vt_proj: LayerNorm-folded weights need the token statistics");
    if (ln_rs) {   // the normalised tokens are the B operand here: statistics per column, sums / bias per row
      g.ln_rs = ln_rs; g.ln_rm = ln_rm; g.ln_s = wv.ln_s; g.ln_swapped = 1; g.ln_zstride = rows;
      g.bias = wv.b; g.bias_row = 1;
    }
    gemm_auto(ctx, g, s, PK_GEMM);
  }
  h16* vt_proj(const PackedLinear& wv, const h16* src, int rows, int rows_pad, int K, const float* ln_rs = nullptr,
                const float* ln_rm = nullptr) {
    h16* vt = ctx->arena.get<h16>((int64_t)N * wv.N * rows_pad);
    vt_proj_into(wv, src, rows, rows_pad, K, vt, ln_rs, ln_rm);
    return vt;
  }

  Act spatial_transformer(const Act& xa, const XfBlockW& b, int H, int W) {
    const h16* x = xa.p;
    const int HW = H * W, C = b.C;
    const int64_t P = (int64_t)N * HW;
    const int M = (int)P;
    h16* out = ctx->arena.get<h16>(P * C);
    GnEmit eo = emit_for(HW, C);
    ctx->arena.push();
    h16* h = ctx->arena.get<h16>(P * C);
    // LayerNorm row partials: the three LayerNorm inputs of the block (proj_in output, the two attention residual sums) are written by
    // GEMM epilogues that leave per-tile row sums here; ln_finish replaces the statistics pass over the tensor (one buffer: a
    // LayerNorm's partials are consumed before the next producer runs)
    LnEmit le;
    le.buf = ctx->arena.get<float>((int64_t)M * 16);
    static const int gn_fold = getenv("SVG_GN_FOLD") ? atoi(getenv("SVG_GN_FOLD")) : 1;
    if (gn_fold && xa.st.valid() && b.proj_in_f32) {
      // GroupNorm (no activation) -> proj_in: the normalisation is folded into per-sample weights, so the normalised tensor
      // is never written: h_b = x_b (W diag(gamma rstd_b))^T + (bias + W (beta - mean_b rstd_b gamma)), one batched GEMM
      ctx->arena.push();
      float* stats = ctx->arena.get<float>((int64_t)N * m->groups * 2);
      h16* wb = ctx->arena.get<h16>((int64_t)N * C * C);
      float* bb = ctx->arena.get<float>((int64_t)N * C);
      gn_finish(ctx, xa.st, C, nullptr, 0, stats, N, HW, m->groups, 1e-6f, s);
      if (SVG_LAUNCHING(ctx)) {
        ProfScope ps(ctx, PK_GNORM, s, 0, (double)N * C * C * 2, "fold_weights");
        gn_fold_weights(b.proj_in_f32, b.proj_in.b, b.gn.g, b.gn.b, stats, wb, bb, N, C, C, m->groups, s);
      }
      GemmArgs g;
      g.A = x; g.lda = C; g.Wt = wb; g.ldb = C; g.M = HW; g.N = C; g.K = C; g.n_valid = C;
      g.batch = N; g.sA = (int64_t)HW * C; g.sB = (int64_t)C * C; g.sC = (int64_t)HW * C;
      g.bias = bb; g.bias_zs = C; g.C = h; g.ldc = C;
      {
        static const int use_ln = getenv("SVG_LN_EPI") ? atoi(getenv("SVG_LN_EPI")) : 1;
        const int tiles = use_ln ? gemm_ln_tiles(g) : 0;
        le.tiles = 0;
        if (tiles > 0 && tiles <= 5) { g.ln_part = le.buf; g.ln_tiles = tiles; le.tiles = tiles; }
      }
      gemm_auto(ctx, g, s, PK_GEMM);
      ctx->arena.pop();
    } else {
      h16* n0 = ctx->arena.get<h16>(P * C);
      groupnorm(ctx, x, C, nullptr, 0, b.gn.g, b.gn.b, n0, N, HW, m->groups, 1e-6f, 0, s, &xa.st, nullptr);
      linear(ctx, n0, C, b.proj_in, h, C, M, ACT_NONE, nullptr, 0, 0, s, nullptr, nullptr, nullptr, 0, nullptr, 0, 0, &le);
    }
    // LayerNorms: folded into the consuming projections (row statistics only) unless SVG_LN_FOLD=0
    const bool fold = b.qk1.ln_s != nullptr;
    h16* ln = fold ? nullptr : ctx->arena.get<h16>(P * C);
    float* rs = fold ? ctx->arena.get<float>(M + 8) : nullptr;
    float* rm = fold ? ctx->arena.get<float>(M + 8) : nullptr;
    auto norm = [&](const h16* src, const NormW& n) -> const h16* {
      if (fold) {
        if (le.tiles > 0) ln_finish(ctx, le.buf, le.tiles, rs, rm, M, C, 1e-5f, s);   // src's producer left its row partials
        else ln_stats(ctx, src, rs, rm, M, C, 1e-5f, s);
        le.tiles = 0;
        return src;
      }
      layernorm(ctx, src, n.g, n.b, ln, M, C, 1e-5f, s);
      return ln;
    };
    h16* ao = ctx->arena.get<h16>(P * C);
    // ---- self-attention
    const h16* a1 = norm(h, b.ln1);
    {
      ctx->arena.push();
      const int HWp = (int)align_up(HW, 8);
      h16* qk = ctx->arena.get<h16>(P * 2 * C);
      h16* vt = nullptr;
      bool fused = false;
      if (b.qkv1.w && HWp == HW && HW % 16 == 0) {
        // q | k | V^T in ONE weight-stationary launch: the V column groups write V^T directly (GemmArgs::vt_out)
        GemmArgs g;
        g.ln_rs = rs; g.ln_rm = rm; g.ln_s = b.qkv1.ln_s;
        g.A = a1; g.lda = C; g.Wt = b.qkv1.w; g.ldb = C; g.M = M; g.N = 3 * C; g.K = C; g.n_valid = 3 * C;
        g.bias = b.qkv1.b; g.C = qk; g.ldc = 2 * C;
        g.vt_n0 = 2 * C; g.vt_rows = HW; g.vt_ld = HWp; g.vt_bs = (int64_t)C * HWp;
        vt = ctx->arena.get<h16>((int64_t)N * C * HWp);
        g.vt_out = vt;
        static const int qkv_env = getenv("SVG_QKV_FUSED") ? atoi(getenv("SVG_QKV_FUSED")) : 1;
        if (qkv_env && gemm_fused_qkv_supported(g)) { gemm_auto(ctx, g, s, PK_GEMM); fused = true; }
      }
      if (!fused) {
        linear(ctx, a1, C, b.qk1, qk, 2 * C, M, ACT_NONE, nullptr, 0, 0, s, rs, rm);
        vt = vt_proj(b.v1, a1, HW, HWp, C, rs, rm);
      }
      attn_core(qk, 2 * C, qk + C, 2 * C, (int64_t)HW * 2 * C, vt, HWp, (int64_t)C * HWp, ao, C, HW, HW);
      ctx->arena.pop();
    }
    h16* h1 = ctx->arena.get<h16>(P * C);
    // C = 320: to_q + attention over the context + to_out + residual in ONE launch (xattn_fused.hip), LayerNorm 2 from the rows it holds
    const bool xa_one = fold && b.xq && xattn_fused_supported(C, m->heads, M, HW, L);
    // ... and, CHAIN form, the self-attention's output projection + residual in front of it: h1 is never written
    const bool xa_chain = xa_one && xattn_chain_enabled() && b.o1.K == C && b.o1.N == C;
    if (!xa_chain) linear(ctx, ao, C, b.o1, h1, C, M, ACT_NONE, h, C, 0, s, nullptr, nullptr, nullptr, 0, nullptr, 0, 0, xa_one ? nullptr : &le);
    h16* h2 = ctx->arena.get<h16>(P * C);
    // ---- cross-attention
    const h16* a2 = xa_one ? h1 : norm(h1, b.ln2);
    {
      ctx->arena.push();
      h16* q = xa_one ? nullptr : ctx->arena.get<h16>(P * C);
      if (!xa_one) linear(ctx, a2, C, b.q2, q, C, M, ACT_NONE, nullptr, 0, 0, s, rs, rm);
      const int idx = xf_idx++;
      const int64_t kn = (int64_t)N * L * C, vn = (int64_t)N * C * Lp;
      h16 *k, *vt;
      h16* const unset = (h16*)(uintptr_t)0x1000;     // dry pass before the cache exists: never dereferenced (nothing is launched)
      // the cache lives outside the arena (it survives the call); plan mode sizes it too, so the first real step allocates nothing
      const bool grow_cache = cache && (SVG_LAUNCHING(ctx) || ctx->plan_only);
      if (cache) {
        if (grow_cache) {
          if ((int)cache->k.size() <= idx) { cache->k.resize(idx + 1, nullptr); cache->vt.resize(idx + 1, nullptr); cache->k_cap.resize(idx + 1, 0); cache->vt_cap.resize(idx + 1, 0); }
          if (cache->k_cap[idx] < kn) { cache->k[idx] = (h16*)ctx->dalloc(kn * sizeof(h16)); cache->k_cap[idx] = kn; cache->valid = false; }
          if (cache->vt_cap[idx] < vn) { cache->vt[idx] = (h16*)ctx->dalloc(vn * sizeof(h16)); cache->vt_cap[idx] = vn; cache->valid = false; }
        }
        const bool have = idx < (int)cache->k.size() && cache->k[idx];
        k = have ? cache->k[idx] : unset; vt = have ? cache->vt[idx] : unset;
      } else {
        k = ctx->arena.get<h16>(kn);
        vt = ctx->arena.get<h16>(vn);
      }
      h16 *kp = nullptr, *vp = nullptr;
      if (xa_one) {
        const int64_t pn = xattn_kv_pack_elems(N);
        if (cache) {
          if (grow_cache) {
            if ((int)cache->kp.size() <= idx) { cache->kp.resize(idx + 1, nullptr); cache->vp.resize(idx + 1, nullptr); cache->kvp_cap.resize(idx + 1, 0); }
            if (cache->kvp_cap[idx] < pn) {
              cache->kp[idx] = (h16*)ctx->dalloc(pn * sizeof(h16)); cache->vp[idx] = (h16*)ctx->dalloc(pn * sizeof(h16));
              cache->kvp_cap[idx] = pn; cache->valid = false;
            }
          }
          const bool have = idx < (int)cache->kp.size() && cache->kp[idx];
          kp = have ? cache->kp[idx] : unset; vp = have ? cache->vp[idx] : unset;
        } else {
          kp = ctx->arena.get<h16>(pn);
          vp = ctx->arena.get<h16>(pn);
        }
      }
      if (!(cache && cache->valid && SVG_LAUNCHING(ctx))) {
        linear(ctx, ctxb, m->ctx_dim, b.k2, k, C, N * L, ACT_NONE, nullptr, 0, 0, s);
        vt_proj_into(b.v2, ctxb, L, Lp, m->ctx_dim, vt);
        if (xa_one && SVG_LAUNCHING(ctx)) xattn_pack_kv(k, C, (int64_t)L * C, vt, Lp, (int64_t)C * Lp, kp, vp, N, L, s);
      }
      if (xa_chain) xattn_fused(ctx, ao, C, h, C, b.o1.w, b.o1.b, nullptr, nullptr, b.xqp, b.xq_s, b.xq_b, kp, vp, b.xo, b.o2.b, h2, C, M, HW, L, s);
      else if (xa_one) xattn_fused(ctx, h1, C, nullptr, 0, nullptr, nullptr, nullptr, nullptr, b.xq, b.xq_s, b.xq_b, kp, vp, b.xo, b.o2.b, h2, C, M, HW, L, s);
      else attn_core(q, C, k, C, (int64_t)L * C, vt, Lp, (int64_t)C * Lp, ao, C, HW, L);
      ctx->arena.pop();
    }
    if (!xa_one) linear(ctx, ao, C, b.o2, h2, C, M, ACT_NONE, h1, C, 0, s, nullptr, nullptr, nullptr, 0, nullptr, 0, 0, &le);
    // ---- GEGLU feed-forward
    const bool ff_one = fold && b.ff2p && ff_fused_supported(C, M);
    const h16* a3 = ff_one ? h2 : norm(h2, b.ln3);     // the fused feed-forward takes its LayerNorm statistics from the rows it holds
    {
      // The GEGLU intermediate is M x 4C (293 MB at 28 clips x 64 x 64 x 1280): written by ff1 and read back by ff2.  Run
      // the pair over row chunks whose intermediate fits the 256 MiB Infinity Cache (with the other stream group's share):
      // the same chunk-sized buffer is rewritten per chunk, so ff2 reads it from the cache instead of HBM.
      if (ff_one) {
        // ff1 -> GEGLU -> ff2 in one kernel: the M x 4C intermediate never leaves the CU (h is free again: reused as h3)
        ff_fused(ctx, h2, C, b.ff1.w, b.ff1.b, b.ff1.ln_s, nullptr, nullptr, b.ff2p, b.ff2.b, h2, C, h, C, M, s);
      } else {
      ctx->arena.push();
      const int rows = ff_chunk_rows(M, C);
      h16* g = ctx->arena.get<h16>((int64_t)std::min(rows, M) * 4 * C);
      for (int m0 = 0; m0 < M; m0 += rows) {
        const int mc = std::min(rows, M - m0);
        linear(ctx, a3 + (int64_t)m0 * C, C, b.ff1, g, 4 * C, mc, ACT_GEGLU, nullptr, 0, 0, s, rs ? rs + m0 : nullptr, rm ? rm + m0 : nullptr);
        linear(ctx, g, 4 * C, b.ff2, h + (int64_t)m0 * C, C, mc, ACT_NONE, h2 + (int64_t)m0 * C, C, 0, s);   // h is free again: reuse as h3
      }
      ctx->arena.pop();
      }
    }
    linear(ctx, h, C, b.proj_out, out, C, M, ACT_NONE, x, C, 0, s, nullptr, nullptr, &eo, HW);
    ctx->arena.pop();
    Act o;
    o.p = out; o.C = C; o.st = eo.st;
    return o;
  }
};
}  // namespace

void UnetModel::forward(svg_ctx* ctx, const float* x, int N, int h, int w, const float* timesteps, const float* ctx_emb,
                        int ctx_len, float* eps_out, hipStream_t s) {
  // A direct call cannot know whether its batch is a classifier-free-guidance pair: with fp8=1 it runs EVERY eligible conv in e4m3 unless
  // $SVG_FP8_SITES_FORWARD narrows the placement (e.g. 3140 = 0xC44, the 16 x 16 level a guided svg_ddim_loop keeps).
  const int64_t sites = svg_env_i64("SVG_FP8_SITES_FORWARD", -1);
  run_planned(ctx, [&]() { run(ctx, x, N, h, w, timesteps, ctx_emb, ctx_len, eps_out, s, nullptr, sites); });
}

void UnetModel::run(svg_ctx* ctx, const float* x, int N, int h, int w, const float* timesteps, const float* ctx_emb,
                    int ctx_len, float* eps_out, hipStream_t s, KvCache* cache, int64_t fp8_sites) {
  SVG_CHECK(ready, "unet: svg_finalize has not been called");
  const int nb = (int)block_out.size();
  const int down = 1 << (nb - 1);
  SVG_CHECK(N >= 1 && h % down == 0 && w % down == 0, "unet: latent %dx%d must be divisible by %d", h, w, down);
  SVG_CHECK(ctx_len >= 1, "unet: empty context");
  const int c0 = block_out[0];
  UnetRun r{ctx, this, s, N};
  r.cache = cache;
  r.sites = fp8_sites;
  r.L = ctx_len; r.Lp = (int)align_up(ctx_len, 8);
  // context -> bf16
  h16* cb = ctx->arena.get<h16>((int64_t)N * ctx_len * ctx_dim);
  if (SVG_LAUNCHING(ctx)) { ProfScope ps(ctx, PK_ELT, s, 0, 0); f32_to_h16(ctx_emb, cb, (int64_t)N * ctx_len * ctx_dim, s); }
  r.ctxb = cb;
  // time embedding: sinusoid -> linear_1 -> SiLU -> linear_2 ; every resnet applies time_emb_proj(SiLU(temb))
  h16* te0 = ctx->arena.get<h16>((int64_t)N * c0);
  if (SVG_LAUNCHING(ctx)) { ProfScope ps(ctx, PK_ELT, s, 0, 0); timestep_embed(timesteps, te0, N, c0, s); }
  h16* te1 = ctx->arena.get<h16>((int64_t)N * temb_dim);
  linear(ctx, te0, c0, time1, te1, temb_dim, N, ACT_SILU, nullptr, 0, 0, s);
  h16* te2 = ctx->arena.get<h16>((int64_t)N * temb_dim);
  linear(ctx, te1, temb_dim, time2, te2, temb_dim, N, ACT_SILU, nullptr, 0, 0, s);   // SiLU(temb), shared by all resnets
  float* tall = ctx->arena.get<float>((int64_t)N * temb_all.N);
  linear(ctx, te2, temb_dim, temb_all, tall, temb_all.N, N, ACT_NONE, nullptr, 0, 1, s);
  r.temb = tall; r.temb_ld = temb_all.N;

  // ---- conv_in
  typedef UnetRun::Act Act;
  h16* x0 = ctx->arena.get<h16>((int64_t)N * h * w * 8);
  if (SVG_LAUNCHING(ctx)) { ProfScope ps(ctx, PK_ELT, s, 0, 0); nchw_to_act(x, x0, N, in_ch, h, w, 8, 1.f, s); }
  int H = h, W = w;
  Act cur;
  {
    h16* y = ctx->arena.get<h16>((int64_t)N * H * W * conv_in.Opad);
    GnEmit e = r.emit_for((int64_t)H * W, conv_in.Opad);
    conv3x3(ctx, x0, conv_in, y, N, H, W, A_CONV_S1, nullptr, 0, nullptr, 0, s, &e);
    cur.p = y; cur.C = c0; cur.st = e.st;
  }
  std::vector<Act> skips{cur};
  // ---- down
  for (int i = 0; i < nb; ++i) {
    for (int j = 0; j < layers; ++j) {
      cur = r.resnet(cur, nullptr, down_res[i][j], H, W);
      if (attn[i]) cur = r.spatial_transformer(cur, down_attn[i][j], H, W);
      skips.push_back(cur);
    }
    if (i < nb - 1) {
      h16* y = ctx->arena.get<h16>((int64_t)N * (H / 2) * (W / 2) * down_s[i].Opad);
      GnEmit e = r.emit_for((int64_t)(H / 2) * (W / 2), down_s[i].Opad);
      conv3x3(ctx, cur.p, down_s[i], y, N, H, W, A_CONV_S2P1, nullptr, 0, nullptr, 0, s, &e);
      cur.p = y; cur.st = e.st; H /= 2; W /= 2;
      skips.push_back(cur);
    }
  }
  // ---- mid
  cur = r.resnet(cur, nullptr, mid0, H, W);
  cur = r.spatial_transformer(cur, mid_attn, H, W);
  cur = r.resnet(cur, nullptr, mid1, H, W);
  // ---- up
  for (int i = 0; i < nb; ++i) {
    const int bi = nb - 1 - i;
    for (int j = 0; j < layers + 1; ++j) {
      const Act sk = skips.back(); skips.pop_back();
      cur = r.resnet(cur, &sk, up_res[i][j], H, W);
      if (attn[bi]) cur = r.spatial_transformer(cur, up_attn[i][j], H, W);
    }
    if (i < nb - 1) {
      h16* y = ctx->arena.get<h16>((int64_t)N * (2 * H) * (2 * W) * up_s[i].Opad);
      GnEmit e = r.emit_for((int64_t)4 * H * W, up_s[i].Opad);
      if (((fp8_sites >> 9) & 1) && conv3x3_fp8_ok(up_s[i], N, H, W, true)) {      // fp8=1: quantise the (small) source image, conv on the MX fp8 path
        ctx->arena.push();
        const int64_t Ps = (int64_t)N * H * W, Cp = align_up(up_s[i].Cin, 128);
        uint8_t* q = ctx->arena.get<uint8_t>(Ps * Cp);
        uint8_t* qs = ctx->arena.get<uint8_t>(Ps * (Cp / 32));
        quant_act_mx(ctx, cur.p, up_s[i].Cin, q, qs, Ps, s);
        conv3x3_fp8(ctx, q, qs, up_s[i], y, N, H, W, nullptr, 0, nullptr, s, &e, true);
        ctx->arena.pop();
      } else
      conv3x3(ctx, cur.p, up_s[i], y, N, H, W, A_CONV_UP2, nullptr, 0, nullptr, 0, s, &e);
      cur.p = y; cur.st = e.st; H *= 2; W *= 2;
    }
  }
  // ---- out
  h16* t = ctx->arena.get<h16>((int64_t)N * H * W * c0);
  groupnorm(ctx, cur.p, c0, nullptr, 0, norm_out.g, norm_out.b, t, N, H * W, groups, 1e-5f, 1, s, &cur.st, nullptr);
  float* o = ctx->arena.get<float>((int64_t)N * H * W * conv_out.Opad);
  conv3x3(ctx, t, conv_out, o, N, H, W, A_CONV_S1, nullptr, 0, nullptr, 1, s);
  if (SVG_LAUNCHING(ctx)) { ProfScope ps(ctx, PK_ELT, s, 0, 0); actf32_to_nchw(o, conv_out.Opad, eps_out, N, out_ch, H, W, s); }
  if (cache && SVG_LAUNCHING(ctx)) cache->valid = true;   // every block's K / V^T of this context is now stored
}

void UnetModel::ddim_loop(svg_ctx* ctx, float* z, int N, int h, int w, const float* text_emb, int ctx_len, int num_steps,
                          int start_step, float guidance, const float* noise, float* hist, hipStream_t s) {
  SVG_CHECK(ready, "unet: svg_finalize has not been called");
  SVG_CHECK(num_steps >= 1 && num_steps <= 1000 && start_step >= 0 && start_step <= num_steps, "ddim: bad steps %d/%d", start_step, num_steps);
  SVG_CHECK(start_step == 0 || noise, "ddim: start_step > 0 needs the add_noise draws");
  const int ratio = 1000 / num_steps;
  const int64_t n = (int64_t)N * in_ch * h * w;
  const bool cfg = guidance != 0.f;
  const int NB = cfg ? 2 * N : N;
  // the e4m3 placement of this loop (see kFp8SitesGuided): handed to every UNet call of the loop as an argument
  const int64_t sites = cfg ? svg_env_i64("SVG_FP8_SITES_GUIDED", kFp8SitesGuided) : -1;
  const int64_t emb_n = (int64_t)N * ctx_len * ctx_dim;
  auto timestep_at = [&](int i) { return (num_steps - 1 - i) * ratio; };   // (arange(n)*ratio)[::-1]

  // One DDIM step as a hipGraph: the first step runs as direct launches (it fills the cross-attention K / V^T cache), the second is
  // CAPTURED on the caller's stream and the remaining ones replay it.  Everything a step needs that changes from step to step —
  // the timestep of the embedding and the four scheduler coefficients — is read from a device table row selected by a device
  // counter (ddim_step_tab / ddim_tvec / ddim_bump), so a replay needs no new kernel arguments.  The graph bakes arena pointers:
  // it lives for this call only (the arena is planned just above and cannot move until the loop ends).  Off: SVG_DDIM_GRAPH=0,
  // the legacy null stream (not capturable), the profiler's event brackets, and the latent history (a per-step copy target).
  const int graph_env = (int)svg_env_i64("SVG_DDIM_GRAPH", 1);      // cached; the tests toggle it in-process and call svg_env_refresh
  const bool use_graph = graph_env && s != nullptr && !ctx->prof && !hist && (num_steps - start_step) >= 3;

  // planned once for the whole loop: every step has the same shapes
  auto body = [&]() {
    kv.valid = false;   // the context is constant over the loop: K / V^T of the cross-attentions are computed once
    float* tvec = ctx->arena.get<float>(NB);
    float* zin = cfg ? ctx->arena.get<float>(2 * n) : nullptr;
    float* eps = ctx->arena.get<float>((int64_t)NB * n / N);
    float* tab = ctx->arena.get<float>((int64_t)5 * num_steps);
    int* idx = ctx->arena.get<int>(1);
    if (SVG_LAUNCHING(ctx)) {
      if (start_step > 0 && start_step < num_steps) {
        const float a = alphas_cumprod[timestep_at(start_step)];
        add_noise(z, noise, z, n, sqrtf(a), sqrtf(1.f - a), s);
      }
      if (hist) HIP_OK(hipMemcpyAsync(hist, z, n * sizeof(float), hipMemcpyDeviceToDevice, s));
      if (use_graph) {
        std::vector<float> h((size_t)5 * num_steps);
        for (int i = 0; i < num_steps; ++i) {
          const int t = timestep_at(i);
          h[5 * i] = (float)t;
          ddim_coefs(t, t - ratio, &h[5 * i + 1], &h[5 * i + 2], &h[5 * i + 3], &h[5 * i + 4]);
        }
        // pageable host memory: the copy is staged before the call returns
        HIP_OK(hipMemcpyAsync(tab, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, s));
        const int first = start_step;
        HIP_OK(hipMemcpyAsync(idx, &first, sizeof(int), hipMemcpyHostToDevice, s));
        HIP_OK(hipStreamSynchronize(s));
      }
    }
    auto one_step = [&](int i, bool tabled) {
      const int t = timestep_at(i);
      ctx->arena.push();
      if (SVG_LAUNCHING(ctx)) {
        if (tabled) ddim_tvec(tvec, NB, tab, idx, s);
        else fill_f32(tvec, NB, (float)t, s);
        if (cfg) {
          HIP_OK(hipMemcpyAsync(zin, z, n * sizeof(float), hipMemcpyDeviceToDevice, s));
          HIP_OK(hipMemcpyAsync(zin + n, z, n * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
      }
      // guidance == 0: noise_pred = uncond + 0*(text - uncond) == uncond — only the uncond half is needed
      std::unique_ptr<ProfScope> step_scope;
      if (SVG_LAUNCHING(ctx)) step_scope.reset(new ProfScope(ctx, PK_UNET_STEP, s, 0, 0));
      run(ctx, cfg ? zin : z, NB, h, w, tvec, text_emb, ctx_len, eps, s, &kv, sites);
      if (SVG_LAUNCHING(ctx)) {
        if (tabled) {
          ddim_step_tab(z, eps, cfg ? eps + n : nullptr, guidance, z, n, tab, idx, s);
          ddim_bump(idx, s);
        } else {
          float sa, s1a, sap, s1ap;
          ddim_coefs(t, t - ratio, &sa, &s1a, &sap, &s1ap);
          ProfScope ps(ctx, PK_ELT, s, 0, 0);
          ddim_step(z, eps, cfg ? eps + n : nullptr, guidance, z, n, sa, s1a, sap, s1ap, s);
          if (hist) HIP_OK(hipMemcpyAsync(hist + (int64_t)(i - start_step + 1) * n, z, n * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
      }
      step_scope.reset();
      ctx->arena.pop();
    };
    if (!use_graph || !SVG_LAUNCHING(ctx)) {
      for (int i = start_step; i < num_steps; ++i) {
        one_step(i, false);
        if (ctx->arena.dry && i > start_step) break;   // two steps are enough to size the arena (the second reuses the K / V^T cache)
      }
      return;
    }
    one_step(start_step, true);                        // direct launches: fills the K / V^T cache, bumps the counter
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    {
      // Shared with other captures, exclusive against every device-wide synchronisation of the library (common.h): HIP rejects
      // hipDeviceSynchronize from ANY thread while this window is open (BENCH_r05: the other stream group's workspace growth).
      CaptureScope cap;
      HIP_OK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      try {
        one_step(start_step + 1, true);
        // test hook: keep the window open so that a test can drive another context's calls into it (tests/test_streams_gpu.py)
        if (const int64_t hold = svg_env_i64("SVG_TEST_CAPTURE_HOLD_MS", 0)) std::this_thread::sleep_for(std::chrono::milliseconds(hold));
      } catch (...) {
        hipStreamEndCapture(s, &graph);
        if (graph) hipGraphDestroy(graph);
        throw;
      }
      HIP_OK(hipStreamEndCapture(s, &graph));
    }
    hipError_t ge = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (ge != hipSuccess) { hipGraphDestroy(graph); HIP_OK(ge); }
    for (int i = start_step + 1; i < num_steps; ++i) {
      ge = hipGraphLaunch(exec, s);
      if (ge != hipSuccess) break;
    }
    // the executable graph is released once the stream has run it (the call stays asynchronous otherwise: the destroy is deferred
    // through a host callback would need a thread-safe queue; a DDIM loop is 1.5 s of GPU work, the sync costs nothing measurable)
    hipError_t se = hipStreamSynchronize(s);
    hipGraphExecDestroy(exec);
    hipGraphDestroy(graph);
    HIP_OK(ge);
    HIP_OK(se);
  };
  (void)emb_n;
  run_planned(ctx, body);
}

}  // namespace SDNS

#if SD_F16
UnetIface* new_unet_f16() { return new sd_f16::UnetModel(); }
#else
UnetIface* new_unet_bf16() { return new sd_bf16::UnetModel(); }
#endif
