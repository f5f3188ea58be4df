// C ABI, operator level: the kernels the SD graphs are made of, on caller buffers (used by the parity tests).  Compiled once per
// storage type: the bf16 build exports svg_op_<name>, the -DSVG_F16 build svg_op_<name>_f16 (16-bit buffers are IEEE half there).
#include "models.h"
#include "../../include/svg_hip.h"
#include <cstdlib>

#if SD_F16
#define SVG_OP(name) name##_f16
#else
#define SVG_OP(name) name
#endif
#define API_BEGIN try {
#define API_END(ctx)                                   \
  return 0;                                            \
  }                                                    \
  catch (const std::exception& e) { return svg_fail(ctx, e); }

using namespace SDNS;

extern "C" {

int SVG_OP(svg_op_gemm)(svg_ctx* ctx, const uint16_t* A, const uint16_t* W, const float* bias, const uint16_t* residual, void* C,
                int M, int N, int K, int act, int out_f32, void* stream) {
  API_BEGIN
  hipStream_t s = (hipStream_t)stream;
  run_planned(ctx, [&]() {
    GemmArgs g;
    g.A = (const h16*)A; g.lda = K; g.Wt = (const h16*)W; g.ldb = K; g.M = M; g.N = N; g.K = K; g.n_valid = N;
    g.bias = bias; g.residual = (const h16*)residual; g.act = act; g.out_f32 = out_f32;
    g.C = C;
    if (act == ACT_GEGLU) {
      // caller passes W rows as [h(0..F-1); gate(0..F-1)], F = N/2: pack here (test hook)
      const int F = N / 2;
      h16* wp = ctx->arena.get<h16>((int64_t)N * K);
      float* bp = ctx->arena.get<float>(N);
      float* wf = ctx->arena.get<float>((int64_t)N * K);
      if (SVG_LAUNCHING(ctx)) {
        h16_to_f32((const h16*)W, wf, (int64_t)N * K, s);
        pack_geglu(wf, bias, wp, bp, F, K, s);
      }
      g.Wt = wp; g.bias = bias ? bp : nullptr; g.ldc = F; g.ldr = F;
    } else {
      g.ldc = N; g.ldr = N;
    }
    gemm_auto(ctx, g, s, PK_GEMM);
  });
  API_END(ctx)
}

int SVG_OP(svg_op_conv3x3)(svg_ctx* ctx, const uint16_t* x, const float* w_oihw, const float* bias, uint16_t* out, int B, int H, int W,
                   int Cin, int Cout, int mode, void* stream) {
  API_BEGIN
  hipStream_t s = (hipStream_t)stream;
  run_planned(ctx, [&]() {
    const int Opad = (int)align_up(Cout, 4);
    h16* wp = ctx->arena.get<h16>((int64_t)Opad * 9 * Cin);
    float* wdev = ctx->arena.get<float>((int64_t)Cout * Cin * 9);
    float* bdev = ctx->arena.get<float>(Opad);
    if (SVG_LAUNCHING(ctx)) {
      HIP_OK(hipMemcpyAsync(wdev, w_oihw, (size_t)Cout * Cin * 9 * sizeof(float), hipMemcpyDefault, s));
      HIP_OK(hipMemsetAsync(bdev, 0, Opad * sizeof(float), s));
      if (bias) HIP_OK(hipMemcpyAsync(bdev, bias, Cout * sizeof(float), hipMemcpyDefault, s));
      pack_conv3x3(wdev, wp, Cout, Cin, Opad, Cin, s);
    }
    GemmArgs g;
    g.A = (const h16*)x;
    g.H = H; g.W = W; g.Cin = Cin;
    switch (mode) {
      case 0: g.amode = (Cin == 8) ? A_CONV_SMALLC : A_CONV_S1; g.Ho = H; g.Wo = W; break;
      case 1: g.amode = A_CONV_S2P1; g.Ho = H / 2; g.Wo = W / 2; break;
      case 2: g.amode = A_CONV_S2ASYM; g.Ho = H / 2; g.Wo = W / 2; break;
      case 3: g.amode = A_CONV_UP2; g.Ho = 2 * H; g.Wo = 2 * W; break;
      default: throw SvgError("conv3x3: bad mode");
    }
    g.Wt = wp; g.ldb = 9 * Cin; g.K = 9 * Cin; g.M = B * g.Ho * g.Wo; g.N = Opad; g.n_valid = Opad;
    g.bias = bdev; g.C = out; g.ldc = Cout;
    SVG_CHECK(Cout % 4 == 0, "conv3x3 op: Cout must be a multiple of 4");
    gemm_auto(ctx, g, s, PK_CONV3);
  });
  API_END(ctx)
}

// conv3x3 (stride 1, pad 1) in MX fp8 (conv_halo_fp8.hip): x (B,H,W,Cin) h16 and the f32 OIHW weights are quantised on the device
// (e4m3 + E8M0 per 32 channels), the conv runs on v_mfma_scale_f32_16x16x128_f8f6f4; out h16 (B,H,W,Cout) = conv + bias (+ residual).
// q_out / s_out (optional): the quantised activations ((B*H*W, Cp) bytes, Cp = Cin rounded up to 128) and their scales, for the tests.
// mode 3: nearest-2x upsample fused in front (out is (B,2H,2W,Cout)), as svg_op_conv3x3.
int SVG_OP(svg_op_conv3x3_mx)(svg_ctx* ctx, const uint16_t* x, const float* w_oihw, const float* bias, const uint16_t* residual, uint16_t* out,
                      uint8_t* q_out, uint8_t* s_out, int B, int H, int W, int Cin, int Cout, int mode, void* stream) {
  API_BEGIN
  hipStream_t s = (hipStream_t)stream;
  SVG_CHECK(Cout % 4 == 0 && Cin % 64 == 0, "conv3x3_mx op: Cout %% 4 and Cin %% 64 must be 0");
  SVG_CHECK(mode == 0 || mode == 3, "conv3x3_mx op: mode 0 (stride 1) or 3 (nearest-2x upsample in front)");
  const bool up2 = mode == 3;
  run_planned(ctx, [&]() {
    ConvW cw;
    cw.Cin = Cin; cw.Cout = Cout; cw.Opad = Cout; cw.Cp = (int)align_up(Cin, 128);
    const int64_t P = (int64_t)B * H * W;
    float* wdev = ctx->arena.get<float>((int64_t)Cout * Cin * 9);
    cw.b = ctx->arena.get<float>(Cout);
    cw.w8 = ctx->arena.get<uint8_t>((int64_t)Cout * 9 * cw.Cp);
    cw.w8s = ctx->arena.get<uint8_t>((int64_t)9 * (cw.Cp / 128) * Cout * 4);
    uint8_t* q = ctx->arena.get<uint8_t>(P * cw.Cp);
    uint8_t* qs = ctx->arena.get<uint8_t>(P * (cw.Cp / 32));
    if (SVG_LAUNCHING(ctx)) {
      HIP_OK(hipMemcpyAsync(wdev, w_oihw, (size_t)Cout * Cin * 9 * sizeof(float), hipMemcpyDefault, s));
      HIP_OK(hipMemsetAsync(cw.b, 0, Cout * sizeof(float), s));
      if (bias) HIP_OK(hipMemcpyAsync(cw.b, bias, Cout * sizeof(float), hipMemcpyDefault, s));
      pack_conv3x3_mx(wdev, cw.w8, cw.w8s, Cout, Cin, Cout, cw.Cp, s);
    }
    quant_act_mx(ctx, (const h16*)x, Cin, q, qs, P, s);
    conv3x3_fp8(ctx, q, qs, cw, (h16*)out, B, H, W, nullptr, 0, (const h16*)residual, s, nullptr, up2);
    if (SVG_LAUNCHING(ctx)) {
      if (q_out) HIP_OK(hipMemcpyAsync(q_out, q, (size_t)P * cw.Cp, hipMemcpyDeviceToDevice, s));
      if (s_out) HIP_OK(hipMemcpyAsync(s_out, qs, (size_t)P * (cw.Cp / 32), hipMemcpyDeviceToDevice, s));
    }
  });
  API_END(ctx)
}

// conv3x3 (stride 1) whose epilogue leaves the GroupNorm column sums, followed by the GroupNorm that consumes them
int SVG_OP(svg_op_conv3x3_gn)(svg_ctx* ctx, const uint16_t* x, const float* w_oihw, const float* bias, const float* gamma, const float* beta,
                      uint16_t* conv_out, uint16_t* gn_out, int B, int H, int W, int Cin, int Cout, int groups, float eps, int silu,
                      int* used_epilogue_stats, void* stream) {
  API_BEGIN
  hipStream_t s = (hipStream_t)stream;
  SVG_CHECK(Cout % 4 == 0 && Cin % 64 == 0, "conv3x3_gn op: Cout %% 4 and Cin %% 64 must be 0");
  int used = 0;
  run_planned(ctx, [&]() {
    ConvW cw;
    cw.Cin = Cin; cw.Cout = Cout; cw.Opad = Cout;
    cw.w = ctx->arena.get<h16>((int64_t)Cout * 9 * Cin);
    float* wdev = ctx->arena.get<float>((int64_t)Cout * Cin * 9);
    cw.b = ctx->arena.get<float>(Cout);
    float* gdev = ctx->arena.get<float>(Cout);
    float* bdev = ctx->arena.get<float>(Cout);
    if (SVG_LAUNCHING(ctx)) {
      HIP_OK(hipMemcpyAsync(wdev, w_oihw, (size_t)Cout * Cin * 9 * sizeof(float), hipMemcpyDefault, s));
      HIP_OK(hipMemsetAsync(cw.b, 0, Cout * sizeof(float), s));
      if (bias) HIP_OK(hipMemcpyAsync(cw.b, bias, Cout * sizeof(float), hipMemcpyDefault, s));
      HIP_OK(hipMemcpyAsync(gdev, gamma, Cout * sizeof(float), hipMemcpyDefault, s));
      HIP_OK(hipMemcpyAsync(bdev, beta, Cout * sizeof(float), hipMemcpyDefault, s));
      pack_conv3x3(wdev, cw.w, Cout, Cin, Cout, Cin, s);
    }
    GnEmit e;
    e.buf = ctx->arena.get<float>(gn_part_floats(B, (int64_t)H * W, Cout));
    conv3x3(ctx, (const h16*)x, cw, conv_out, B, H, W, A_CONV_S1, nullptr, 0, nullptr, 0, s, &e);
    used = e.st.valid() ? 1 : 0;
    groupnorm(ctx, (const h16*)conv_out, Cout, nullptr, 0, gdev, bdev, (h16*)gn_out, B, H * W, groups, eps, silu, s, &e.st, nullptr);
  });
  if (used_epilogue_stats) *used_epilogue_stats = used;
  API_END(ctx)
}

// C[M,N] = [A | A2][M, K] * W[N,K]^T + bias: dense GEMM whose A operand is the channel concat of two tensors (A: k_split columns)
int SVG_OP(svg_op_gemm_cat)(svg_ctx* ctx, const uint16_t* A, const uint16_t* A2, const uint16_t* W, const float* bias, uint16_t* C, int M, int N,
                    int K, int k_split, void* stream) {
  API_BEGIN
  run_planned(ctx, [&]() {
    GemmArgs g;
    g.A = (const h16*)A; g.lda = k_split; g.A2 = (const h16*)A2; g.lda2 = K - k_split; g.k_split = k_split;
    g.Wt = (const h16*)W; g.ldb = K; g.M = M; g.N = N; g.K = K; g.n_valid = N; g.bias = bias; g.C = C; g.ldc = N;
    gemm_auto(ctx, g, (hipStream_t)stream, PK_GEMM);
  });
  API_END(ctx)
}

// C = A W^T + bias + residual (h16) with the LayerNorm row statistics of C taken from the epilogue's row partials (GemmArgs::ln_part
// + ln_finish): rs[m] = rstd, rm[m] = rstd * mean over the N columns; *used = column tiles that emitted (0: the launch could not, rs / rm
// then come from the ln_stats pass).  Test hook.
int SVG_OP(svg_op_gemm_lnstats)(svg_ctx* ctx, const uint16_t* A, const uint16_t* W, const float* bias, const uint16_t* residual, uint16_t* C, int M,
                        int N, int K, int batch, float* rs, float* rm, int* used, void* stream) {
  API_BEGIN
  int tiles = 0;
  run_planned(ctx, [&]() {
    GemmArgs g;
    g.A = (const h16*)A; g.lda = K; g.Wt = (const h16*)W; g.ldb = K; g.M = M; g.N = N; g.K = K; g.n_valid = N; g.bias = bias;
    g.residual = (const h16*)residual; g.ldr = N; g.C = C; g.ldc = N;
    if (batch > 1) { g.batch = batch; g.sA = (int64_t)M * K; g.sB = 0; g.sC = (int64_t)M * N; }
    float* part = ctx->arena.get<float>((int64_t)batch * M * 16);
    tiles = gemm_ln_tiles(g);
    if (tiles > 8) tiles = 0;
    if (tiles > 0) { g.ln_part = part; g.ln_tiles = tiles; }
    gemm_auto(ctx, g, (hipStream_t)stream, PK_GEMM);
    if (tiles > 0) ln_finish(ctx, part, tiles, rs, rm, batch * M, N, 1e-5f, (hipStream_t)stream);
    else ln_stats(ctx, (const h16*)C, rs, rm, batch * M, N, 1e-5f, (hipStream_t)stream);
  });
  if (used) *used = tiles;
  API_END(ctx)
}

// fused GEGLU feed-forward of a transformer block at C = 320: out = ff2(GEGLU(ff1(LayerNorm(x)))) + residual.  Weights in the
// state_dict layout (W1 [2*4C][C] = [h; gate], W2 [C][4C]); folding, packing and the row statistics happen here (test hook).
int SVG_OP(svg_op_ff_fused)(svg_ctx* ctx, const uint16_t* x, const float* ln_gamma, const float* ln_beta, const float* w1, const float* b1,
                    const float* w2, const float* b2, const uint16_t* residual, uint16_t* out, int M, int C, void* stream) {
  API_BEGIN
  hipStream_t s = (hipStream_t)stream;
  SVG_CHECK(ff_fused_supported(C, 1 << 30), "ff_fused op: C = %d is not supported (320)", C);
  run_planned(ctx, [&]() {
    const int F = 4 * C;
    float* w1d = ctx->arena.get<float>((int64_t)2 * F * C);
    float* b1d = ctx->arena.get<float>(2 * F);
    float* b1f = ctx->arena.get<float>(2 * F);
    float* gd = ctx->arena.get<float>(C);
    float* bd = ctx->arena.get<float>(C);
    float* w2d = ctx->arena.get<float>((int64_t)C * F);
    float* b2d = ctx->arena.get<float>(C);
    h16* w1p = ctx->arena.get<h16>((int64_t)2 * F * C);
    float* b1p = ctx->arena.get<float>(2 * F);
    float* s1 = ctx->arena.get<float>(2 * F);
    h16* w2p = ctx->arena.get<h16>((int64_t)C * F);
    float* rs = ctx->arena.get<float>(M + 8);
    float* rm = ctx->arena.get<float>(M + 8);
    if (SVG_LAUNCHING(ctx)) {
      HIP_OK(hipMemcpyAsync(w1d, w1, (size_t)2 * F * C * 4, hipMemcpyDefault, s));
      HIP_OK(hipMemcpyAsync(b1d, b1, (size_t)2 * F * 4, hipMemcpyDefault, s));
      HIP_OK(hipMemcpyAsync(gd, ln_gamma, (size_t)C * 4, hipMemcpyDefault, s));
      HIP_OK(hipMemcpyAsync(bd, ln_beta, (size_t)C * 4, hipMemcpyDefault, s));
      HIP_OK(hipMemcpyAsync(w2d, w2, (size_t)C * F * 4, hipMemcpyDefault, s));
      HIP_OK(hipMemcpyAsync(b2d, b2, (size_t)C * 4, hipMemcpyDefault, s));
      fold_ln_weights(w1d, b1d, gd, bd, b1f, 2 * F, C, s);
      pack_geglu(w1d, b1f, w1p, b1p, F, C, s);
      rowsum_h16(w1p, s1, 2 * F, C, s);
      pack_ff2_perm(w2d, w2p, C, F, s);
    }
    // the kernel derives the LayerNorm statistics from the rows it holds (the UNet's path); SVG_FF_LNSTATS=1 feeds it ln_stats' instead
    static const bool ext = getenv("SVG_FF_LNSTATS") && atoi(getenv("SVG_FF_LNSTATS"));
    if (ext) ln_stats(ctx, (const h16*)x, rs, rm, M, C, 1e-5f, s);
    ff_fused(ctx, (const h16*)x, C, w1p, b1p, s1, ext ? rs : nullptr, ext ? rm : nullptr, w2p, b2d, (const h16*)residual, C, (h16*)out, C, M, s);
  });
  API_END(ctx)
}

// one-launch cross-attention of a C = 320 block (xattn_fused.hip).  x (M,320): the block input (pre-LayerNorm rows, also the residual) — or,
// chained form (a != null): x is not given, the kernel starts from a (M,320) = the self-attention's output, r its residual and wp (320,320) /
// bp its output projection.  k (N,L,320) and vt (N,320,Lp) are the projected context of every sample (rows_per_sample rows each); weights
// f32 in the state_dict layout; folding, padding and packing happen here (test hook).
int SVG_OP(svg_op_xattn_fused)(svg_ctx* ctx, const uint16_t* x, const uint16_t* a, const uint16_t* r, const float* wp, const float* bp,
                               const float* ln_gamma, const float* ln_beta, const float* wq, const uint16_t* k, const uint16_t* vt, int Lp,
                               const float* wo, const float* bo, uint16_t* out, int M, int rows_per_sample, int L, void* stream) {
  API_BEGIN
  hipStream_t s = (hipStream_t)stream;
  constexpr int C = 320;
  SVG_CHECK(xattn_fused_supported(C, 8, std::max(M, 128 * 192), rows_per_sample, L) && M % rows_per_sample == 0,
            "xattn_fused op: needs 8 heads of 40, L <= 80, samples of a multiple of 128 rows (M %d, rows per sample %d, L %d)", M, rows_per_sample, L);
  SVG_CHECK((x != nullptr) != (a != nullptr), "xattn_fused op: give either x (plain form) or a, r, wp, bp (chained form)");
  const int N = M / rows_per_sample;
  run_planned(ctx, [&]() {
    float* wqd = ctx->arena.get<float>((int64_t)C * C);
    float* wod = ctx->arena.get<float>((int64_t)C * C);
    float* wpd = ctx->arena.get<float>((int64_t)C * C);
    float* gd = ctx->arena.get<float>(C);
    float* bd = ctx->arena.get<float>(C);
    float* bod = ctx->arena.get<float>(C);
    float* bpd = ctx->arena.get<float>(C);
    h16* Wq = ctx->arena.get<h16>((int64_t)384 * C);
    float* sq = ctx->arena.get<float>(384);
    float* bq = ctx->arena.get<float>(384);
    h16* Wo = ctx->arena.get<h16>((int64_t)C * 384);
    h16* Wp = ctx->arena.get<h16>((int64_t)C * C);
    h16* kp = ctx->arena.get<h16>(xattn_kv_pack_elems(N));
    h16* vp = ctx->arena.get<h16>(xattn_kv_pack_elems(N));
    if (SVG_LAUNCHING(ctx)) {
      HIP_OK(hipMemcpyAsync(wqd, wq, (size_t)C * C * 4, hipMemcpyDefault, s));
      HIP_OK(hipMemcpyAsync(wod, wo, (size_t)C * C * 4, hipMemcpyDefault, s));
      HIP_OK(hipMemcpyAsync(gd, ln_gamma, (size_t)C * 4, hipMemcpyDefault, s));
      HIP_OK(hipMemcpyAsync(bd, ln_beta, (size_t)C * 4, hipMemcpyDefault, s));
      HIP_OK(hipMemcpyAsync(bod, bo, (size_t)C * 4, hipMemcpyDefault, s));
      xattn_pack_q(wqd, gd, bd, Wq, sq, bq, a ? 1 : 0, s);
      xattn_pack_o(wod, Wo, s);
      xattn_pack_kv((const h16*)k, C, (int64_t)L * C, (const h16*)vt, Lp, (int64_t)C * Lp, kp, vp, N, L, s);
      if (a) {
        HIP_OK(hipMemcpyAsync(wpd, wp, (size_t)C * C * 4, hipMemcpyDefault, s));
        HIP_OK(hipMemcpyAsync(bpd, bp, (size_t)C * 4, hipMemcpyDefault, s));
        pack_linear(wpd, Wp, C, C, C, s);
      }
    }
    if (a) xattn_fused(ctx, (const h16*)a, C, (const h16*)r, C, Wp, bpd, nullptr, nullptr, Wq, sq, bq, kp, vp, Wo, bod, (h16*)out, C, M, rows_per_sample, L, s);
    else xattn_fused(ctx, (const h16*)x, C, nullptr, 0, nullptr, nullptr, nullptr, nullptr, Wq, sq, bq, kp, vp, Wo, bod, (h16*)out, C, M, rows_per_sample, L, s);
  });
  API_END(ctx)
}

// MX fp8 quantiser: x (rows,K) bf16 -> q (rows,K) e4m3 bytes + scales (rows,K/32) E8M0 bytes
int SVG_OP(svg_op_quant_mx)(svg_ctx* ctx, const uint16_t* x, uint8_t* q, uint8_t* scales, int64_t rows, int K, void* stream) {
  API_BEGIN
  quant_mx_h16(ctx, (const h16*)x, K, q, scales, rows, K, (hipStream_t)stream);
  API_END(ctx)
}

// C = act(Q(A) Q(W)^T + bias + residual) with both operands quantised to MX fp8 on the fly (test hook / benchmark of the fp8 GEMM)
int SVG_OP(svg_op_gemm_fp8)(svg_ctx* ctx, const uint16_t* A, const uint16_t* W, const float* bias, const uint16_t* residual, void* C, int M, int N,
                    int K, int act, int out_f32, void* stream) {
  API_BEGIN
  hipStream_t s = (hipStream_t)stream;
  run_planned(ctx, [&]() {
    uint8_t* aq = ctx->arena.get<uint8_t>((int64_t)M * K);
    uint8_t* as = ctx->arena.get<uint8_t>((int64_t)M * (K / 32));
    uint8_t* wq = ctx->arena.get<uint8_t>((int64_t)N * K);
    uint8_t* wsc = ctx->arena.get<uint8_t>((int64_t)N * (K / 32));
    quant_mx_h16(ctx, (const h16*)A, K, aq, as, M, K, s);
    quant_mx_h16(ctx, (const h16*)W, K, wq, wsc, N, K, s);
    GemmArgs g;
    g.M = M; g.N = N; g.K = K; g.bias = bias; g.residual = (const h16*)residual; g.ldr = N; g.act = act; g.out_f32 = out_f32; g.C = C; g.ldc = N;
    gemm_fp8(ctx, aq, as, wq, wsc, g, s);
  });
  API_END(ctx)
}

int SVG_OP(svg_op_groupnorm)(svg_ctx* ctx, const uint16_t* x, const float* gamma, const float* beta, uint16_t* out, int B, int HW, int C,
                     int groups, float eps, int silu, void* stream) {
  API_BEGIN
  run_planned(ctx, [&]() {
    groupnorm(ctx, (const h16*)x, C, nullptr, 0, gamma, beta, (h16*)out, B, HW, groups, eps, silu, (hipStream_t)stream);
  });
  API_END(ctx)
}

int SVG_OP(svg_op_layernorm)(svg_ctx* ctx, const uint16_t* x, const float* gamma, const float* beta, uint16_t* out, int M, int C,
                     float eps, void* stream) {
  API_BEGIN
  layernorm(ctx, (const h16*)x, gamma, beta, (h16*)out, M, C, eps, (hipStream_t)stream);
  API_END(ctx)
}

int SVG_OP(svg_op_attention)(svg_ctx* ctx, const uint16_t* q, const uint16_t* k, const uint16_t* vt, uint16_t* out, int B, int heads,
                     int Sq, int Skv, int d, int ldq, int ldk, int ldvt, int ldo, int64_t qb, int64_t kb, int64_t vtb,
                     int64_t ob, float scale, void* stream) {
  API_BEGIN
  AttnArgs a;
  a.q = (const h16*)q; a.k = (const h16*)k; a.vt = (const h16*)vt; a.out = (h16*)out;
  a.B = B; a.heads = heads; a.Sq = Sq; a.Skv = Skv; a.d = d;
  a.ldq = ldq; a.ldk = ldk; a.ldvt = ldvt; a.ldo = ldo; a.qb = qb; a.kb = kb; a.vtb = vtb; a.ob = ob; a.scale = scale;
  attention(ctx, a, (hipStream_t)stream);
  API_END(ctx)
}

}  // extern "C"
