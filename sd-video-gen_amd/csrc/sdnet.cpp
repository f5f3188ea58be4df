// Shared pieces of the SD networks (VAE, UNet): weight packing by state_dict name and the
// conv / linear / resnet building blocks over NHWC bf16 activations.
#include "models.h"
#include <algorithm>
#include <cstdlib>

namespace SDNS {

// elements one kernel launch may address per operand (32-bit byte offsets); $SVG_CHUNK_LIMIT lowers it so that the tests can
// drive the batch / row chunking of conv3x3() and linear() at small sizes (cached; the tests toggle it in-process and call svg_env_refresh)
static int64_t chunk_limit() {
  const int64_t full = (1LL << 31) - 1;
  const int64_t v = svg_env_i64("SVG_CHUNK_LIMIT", full);
  return v > 0 && v < full ? v : full;
}

float* keep_f32(svg_ctx* ctx, WeightStore& ws, const std::string& name, int64_t numel) {
  const Weight& w = ws.get(name);
  SVG_CHECK(w.numel == numel, "weight %s has %lld elements, expected %lld", name.c_str(), (long long)w.numel, (long long)numel);
  return w.f32;   // stays in the store (small 1-D parameters)
}

ConvW load_conv3x3(svg_ctx* ctx, WeightStore& ws, const std::string& prefix, int Cin, int Cout, hipStream_t s, bool fp8) {
  const Weight& w = ws.get(prefix + ".weight", {Cout, Cin, 3, 3});
  ConvW cw;
  cw.Cout = Cout;
  cw.Opad = (int)align_up(Cout, 4);
  cw.Cin = (Cin < 64) ? 8 : Cin;       // small-Cin convs (image / latent inputs) run on 8 padded channels
  SVG_CHECK(Cin <= 8 || Cin % 64 == 0, "conv %s: Cin=%d must be <= 8 or a multiple of 64", prefix.c_str(), Cin);
  cw.w = (h16*)ctx->dalloc((int64_t)cw.Opad * 9 * cw.Cin * sizeof(h16));
  pack_conv3x3(w.f32, cw.w, Cout, Cin, cw.Opad, cw.Cin, s);
  cw.b = (float*)ctx->dalloc(cw.Opad * sizeof(float));
  HIP_OK(hipMemsetAsync(cw.b, 0, cw.Opad * sizeof(float), s));
  HIP_OK(hipMemcpyAsync(cw.b, keep_f32(ctx, ws, prefix + ".bias", Cout), Cout * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (fp8 && Cin % 64 == 0 && cw.Opad >= 128) {   // MX fp8 copy, quantised from the f32 weights (one rounding, not two)
    cw.Cp = (int)align_up(Cin, 128);
    cw.w8 = (uint8_t*)ctx->dalloc((int64_t)cw.Opad * 9 * cw.Cp);
    cw.w8s = (uint8_t*)ctx->dalloc((int64_t)9 * (cw.Cp / 128) * cw.Opad * 4);
    pack_conv3x3_mx(w.f32, cw.w8, cw.w8s, Cout, Cin, cw.Opad, cw.Cp, s);
  }
  HIP_OK(hipStreamSynchronize(s));
  ws.release(prefix + ".weight");
  return cw;
}

PackedLinear load_linear(svg_ctx* ctx, WeightStore& ws, const std::string& prefix, int N, int K, bool bias, hipStream_t s,
                         const NormW* fold) {
  const Weight& w = ws.get(prefix + ".weight");
  SVG_CHECK(w.numel == (int64_t)N * K && w.shape[0] == N, "weight %s.weight: expected [%d,%d(,1,1)]", prefix.c_str(), N, K);
  PackedLinear pl;
  pl.N = (int)align_up(N, 4); pl.K = K; pl.n_valid = N;
  pl.w = (h16*)ctx->dalloc((int64_t)pl.N * K * sizeof(h16));
  if (bias || fold) {
    pl.b = (float*)ctx->dalloc(pl.N * sizeof(float));
    HIP_OK(hipMemsetAsync(pl.b, 0, pl.N * sizeof(float), s));
    if (bias) HIP_OK(hipMemcpyAsync(pl.b, keep_f32(ctx, ws, prefix + ".bias", N), N * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  if (fold) {   // b += W beta, W *= gamma (on the f32 copy, which is released below), before the bf16 rounding
    SVG_CHECK(fold->C == K, "fold: norm width %d != K %d", fold->C, K);
    fold_ln_weights(w.f32, pl.b, fold->g, fold->b, pl.b, N, K, s);
  }
  pack_linear(w.f32, pl.w, N, K, pl.N, s);
  if (fold) {
    pl.ln_s = (float*)ctx->dalloc(pl.N * sizeof(float));
    rowsum_h16(pl.w, pl.ln_s, pl.N, K, s);
  }
  HIP_OK(hipStreamSynchronize(s));
  ws.release(prefix + ".weight");
  return pl;
}

void add_fp8_copy(svg_ctx* ctx, PackedLinear& pl, hipStream_t s) {
  if (!pl.w || pl.ln_s || pl.K % 128 != 0 || pl.N % 4 != 0 || pl.w8) return;
  pl.w8 = (uint8_t*)ctx->dalloc((int64_t)pl.N * pl.K);
  pl.w8s = (uint8_t*)ctx->dalloc((int64_t)pl.N * (pl.K / 32));
  quant_mx_h16(ctx, pl.w, pl.K, pl.w8, pl.w8s, pl.N, pl.K, s);
  HIP_OK(hipStreamSynchronize(s));
}

NormW load_norm(svg_ctx* ctx, WeightStore& ws, const std::string& prefix, int C) {
  NormW n;
  n.C = C;
  n.g = keep_f32(ctx, ws, prefix + ".weight", C);
  n.b = keep_f32(ctx, ws, prefix + ".bias", C);
  return n;
}

// fills emit->st when the launch gemm_auto() picks for g can leave the output's GroupNorm column sums (whole row tiles per sample)
static void plan_gn_emit(GemmArgs& g, GnEmit* emit, int rows_per_sample) {
  static const int use_epi = getenv("SVG_GN_EPI") ? atoi(getenv("SVG_GN_EPI")) : 1;   // 0: A/B switch, statistics pass as before
  if (!use_epi || !emit || !emit->buf || rows_per_sample < 1024) return;   // small images take the single-launch GroupNorm (one read)
  const int rows = gemm_emits_gn(g);
  if (rows <= 0 || rows_per_sample % rows != 0) return;
  g.gn_part = emit->buf;
  emit->st.part = emit->buf;
  emit->st.tiles_per_sample = rows_per_sample / rows;
}

bool conv3x3_fp8_ok(const ConvW& cw, int B, int H, int W, bool up2) {
  return cw.w8 != nullptr && conv_halo_fp8_supported(B, up2 ? 2 * H : H, up2 ? 2 * W : W, cw.Cin, cw.Opad);
}

void conv3x3_fp8(svg_ctx* ctx, const uint8_t* x8, const uint8_t* xs, const ConvW& cw, h16* out, int B, int H, int W, const float* bias_bn,
                 int bias_bn_ld, const h16* residual, hipStream_t s, GnEmit* emit, bool up2) {
  SVG_CHECK(conv3x3_fp8_ok(cw, B, H, W, up2), "conv3x3_fp8: %d x %dx%d x %d -> %d does not qualify", B, H, W, cw.Cin, cw.Opad);
  GemmArgs g;
  g.H = H; g.W = W; g.Cin = cw.Cin; g.amode = up2 ? A_CONV_UP2 : A_CONV_S1; g.Ho = up2 ? 2 * H : H; g.Wo = up2 ? 2 * W : W;
  H = g.Ho; W = g.Wo;                       // the output image from here on
  g.K = 9 * cw.Cin; g.M = B * H * W; g.N = cw.Opad; g.n_valid = cw.Opad;
  g.bias = cw.b;
  g.bias_bn = bias_bn; g.bias_bn_ld = bias_bn_ld; g.rows_per_batch = H * W;
  g.residual = residual; g.ldr = cw.Opad;
  g.C = out; g.ldc = cw.Opad;
  // GroupNorm column sums of the output: one partial per 16 x 16 pixel block, like the fp16 halo conv
  static const int use_epi = getenv("SVG_GN_EPI") ? atoi(getenv("SVG_GN_EPI")) : 1;
  if (use_epi && emit && emit->buf && H * W >= 1024 && (H * W) % 256 == 0) {
    g.gn_part = emit->buf;
    emit->st.part = emit->buf;
    emit->st.tiles_per_sample = H * W / 256;
  }
  conv_halo_fp8(ctx, x8, xs, cw.w8, cw.w8s, cw.Opad, g, s);
}

void conv3x3(svg_ctx* ctx, const h16* x, const ConvW& cw, void* out, int B, int H, int W, int amode, const float* bias_bn,
             int bias_bn_ld, const h16* residual, int out_f32, hipStream_t s, GnEmit* emit) {
  // the kernels address an operand with 32-bit byte offsets: an input or output of 2^31 elements or more (the 512 x 512
  // VAE levels beyond ~30 images) is processed in batch chunks
  {
    const int up = amode == A_CONV_UP2 ? 4 : 1;
    const int64_t per_img = std::max<int64_t>((int64_t)H * W * std::max(cw.Cin, 8), (int64_t)H * W * up * cw.Opad);
    const int64_t lim = chunk_limit();
    if ((int64_t)B * per_img > lim && B > 1) {
      const int chunk = (int)std::max<int64_t>(1, lim / per_img);
      const int Ho = amode == A_CONV_UP2 ? 2 * H : ((amode == A_CONV_S2P1 || amode == A_CONV_S2ASYM) ? H / 2 : H);
      const int Wo = amode == A_CONV_UP2 ? 2 * W : ((amode == A_CONV_S2P1 || amode == A_CONV_S2ASYM) ? W / 2 : W);
      const int cin = cw.Cin;
      for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = std::min(chunk, B - b0);
        const int64_t o = (int64_t)b0 * Ho * Wo * cw.Opad;
        conv3x3(ctx, x + (int64_t)b0 * H * W * cin, cw, out_f32 ? (void*)((float*)out + o) : (void*)((h16*)out + o), nb, H, W, amode,
                bias_bn ? bias_bn + (int64_t)b0 * (bias_bn_ld ? bias_bn_ld : cw.Opad) : nullptr, bias_bn_ld,
                residual ? residual + o : nullptr, out_f32, s);
      }
      return;
    }
  }
  GemmArgs g;
  g.A = x; g.H = H; g.W = W; g.Cin = cw.Cin;
  g.amode = (cw.Cin == 8) ? A_CONV_SMALLC : amode;
  SVG_CHECK(cw.Cin != 8 || amode == A_CONV_S1, "small-Cin conv supports stride 1 only");
  switch (amode) {
    case A_CONV_S1: g.Ho = H; g.Wo = W; break;
    case A_CONV_S2P1: case A_CONV_S2ASYM: g.Ho = H / 2; g.Wo = W / 2; break;
    case A_CONV_UP2: g.Ho = 2 * H; g.Wo = 2 * W; break;
    default: throw SvgError("conv3x3: bad mode");
  }
  g.Wt = cw.w; g.ldb = 9 * cw.Cin; g.K = 9 * cw.Cin;
  g.M = B * g.Ho * g.Wo; g.N = cw.Opad; g.n_valid = cw.Opad;
  g.bias = cw.b;
  g.bias_bn = bias_bn; g.bias_bn_ld = bias_bn_ld; g.rows_per_batch = g.Ho * g.Wo;
  g.residual = residual; g.ldr = cw.Opad;
  g.C = out; g.ldc = cw.Opad; g.out_f32 = out_f32;
  plan_gn_emit(g, emit, g.Ho * g.Wo);
  gemm_auto(ctx, g, s, PK_CONV3);
}

void linear(svg_ctx* ctx, const h16* A, int lda, const PackedLinear& pl, void* C, int ldc, int M, int act, const h16* residual,
            int ldr, int out_f32, hipStream_t s, const float* ln_rs, const float* ln_rm, GnEmit* emit, int rows_per_sample,
            const h16* A2, int lda2, int k_split, LnEmit* ln) {
  if (ln) ln->tiles = 0;
  SVG_CHECK((pl.ln_s != nullptr) == (ln_rs != nullptr), "linear: LayerNorm-folded weights need the row statistics (and only they)");
  {   // 32-bit operand offsets in the kernels: split very tall problems (1 x 1 convs on the 512 x 512 VAE levels) by rows
    const int64_t lim = chunk_limit();
    const int64_t per_row = std::max<int64_t>(lda, std::max(ldc, ldr));
    if ((int64_t)M * per_row > lim && M > 1) {
      SVG_CHECK(!A2, "linear: a two-source A operand is not split by rows");
      const int chunk = (int)(lim / per_row) & ~255;
      SVG_CHECK(chunk >= 256, "linear: a %lld-element operand limit is below one 256-row slab of %lld-wide rows", (long long)lim, (long long)per_row);
      const int csz = out_f32 ? 4 : 2;
      for (int m0 = 0; m0 < M; m0 += chunk)
        linear(ctx, A + (int64_t)m0 * lda, lda, pl, (char*)C + (int64_t)m0 * ldc * csz, ldc, std::min(chunk, M - m0), act,
               residual ? residual + (int64_t)m0 * ldr : nullptr, ldr, out_f32, s, ln_rs ? ln_rs + m0 : nullptr, ln_rm ? ln_rm + m0 : nullptr);
      return;
    }
  }
  if (pl.w8 && !A2 && !ln_rs && act != ACT_GEGLU && M >= 1024 && lda == pl.K && gemm_fp8_supported(M, pl.N, pl.K)) {
    // MX fp8: the activations are quantised per 32-element block on the way in (one extra pass over A), f32 accumulate
    ctx->arena.push();
    uint8_t* aq = ctx->arena.get<uint8_t>((int64_t)M * pl.K);
    uint8_t* as = ctx->arena.get<uint8_t>((int64_t)M * (pl.K / 32));
    quant_mx_h16(ctx, A, lda, aq, as, M, pl.K, s);
    GemmArgs g8;
    g8.M = M; g8.N = pl.N; g8.K = pl.K; g8.bias = pl.b; g8.act = act; g8.residual = residual; g8.ldr = ldr; g8.C = C; g8.ldc = ldc; g8.out_f32 = out_f32;
    gemm_fp8(ctx, aq, as, pl.w8, pl.w8s, g8, s);
    ctx->arena.pop();
    return;
  }
  GemmArgs g;
  g.ln_rs = ln_rs; g.ln_rm = ln_rm; g.ln_s = pl.ln_s;
  g.A = A; g.lda = lda; g.Wt = pl.w; g.ldb = pl.K; g.M = M; g.N = pl.N; g.K = pl.K; g.n_valid = pl.N;
  g.bias = pl.b; g.act = act; g.residual = residual; g.ldr = ldr; g.C = C; g.ldc = ldc; g.out_f32 = out_f32;
  g.A2 = A2; g.lda2 = lda2; g.k_split = k_split;
  plan_gn_emit(g, emit, rows_per_sample);
  if (ln && ln->buf) {
    static const int use_ln = getenv("SVG_LN_EPI") ? atoi(getenv("SVG_LN_EPI")) : 1;    // 0: A/B switch, ln_stats pass as before
    const int tiles = use_ln ? gemm_ln_tiles(g) : 0;
    if (tiles > 0 && tiles <= 5) { g.ln_part = ln->buf; g.ln_tiles = tiles; ln->tiles = tiles; }   // C = 1280 (8 tiles): the finish costs what the 8 us pass did
  }
  gemm_auto(ctx, g, s, PK_GEMM);
}

// per-device kernel attributes (dynamic LDS limits) of every kernel instantiation of this namespace
void sd_init_device() {
  gemm_init_device();
  gemm_pp_init_device();
  gemm_ws_init_device();
  vae_attn_init_device();
  conv_halo_init_device();
  ff_fused_init_device();
  xattn_fused_init_device();
  gemm_fp8_init_device();
  conv_halo_fp8_init_device();
}

}  // namespace SDNS
