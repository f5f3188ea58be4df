// Inception-v1 I3D (evaluation/pytorch_i3d.py:136-322) and the FVD entry points (evaluation/fvd_2.py) of the library: the network
// whose logits the reference's text loop compares between real and generated clips (prediction/predict_text.py:155-168,289-314).
// f32, channels-last activations, BatchNorm folded into the packed weights at svg_finalize.  Kernels: i3d_kernels.hip.
#include "models.h"
#include "../../include/svg_hip.h"
#include <cmath>

struct Conv3dArgs {
  const float* x; int B, T, H, W, Cin;
  const float* w; const float* bias;
  float* y; int To, Ho, Wo, Cout, ldc, coff;
  int kt, kh, kw, st, sh, sw, pt, ph, pw;
  int relu;
};
void conv3d_launch(const Conv3dArgs& a, hipStream_t s);
void pack_conv3d(const float* w, const float* gamma, const float* beta, const float* mean, const float* var, const float* cbias, float* wout,
                 float* bout, int Cout, int Cin, int Cin_pad, int taps, float eps, hipStream_t s);
void maxpool3d_same(const float* x, float* y, int B, int T, int H, int W, int C, int To, int Ho, int Wo, const int k[3], const int st[3],
                    const int pf[3], hipStream_t s);
void avgpool_thw(const float* x, float* y, int B, int T, int HW, int C, int kt, hipStream_t s);
void time_mean(const float* x, float* y, int B, int To, int C, hipStream_t s);
void ncthw_to_nthwc(const float* x, float* y, int B, int C, int T, int H, int W, int Cp, hipStream_t s);
void fvd_preprocess(const uint8_t* v, float* out, int B, int T, int H, int W, int res, hipStream_t s);
void frechet_distance(const float* x1, int n1, const float* x2, int n2, int d, double* ws, double* out, hipStream_t s);

namespace {
struct MixedCfg { const char* name; int cin; int oc[6]; };
const MixedCfg kMixed[] = {{"Mixed_3b", 192, {64, 96, 128, 16, 32, 32}},   {"Mixed_3c", 256, {128, 128, 192, 32, 96, 64}},
                           {"Mixed_4b", 480, {192, 96, 208, 16, 48, 64}},  {"Mixed_4c", 512, {160, 112, 224, 24, 64, 64}},
                           {"Mixed_4d", 512, {128, 128, 256, 24, 64, 64}}, {"Mixed_4e", 512, {112, 144, 288, 32, 64, 64}},
                           {"Mixed_4f", 528, {256, 160, 320, 32, 128, 128}}, {"Mixed_5b", 832, {256, 160, 320, 32, 128, 128}},
                           {"Mixed_5c", 832, {384, 192, 384, 48, 128, 128}}};

// TF 'SAME' padding of Unit3D / MaxPool3dSamePadding (pytorch_i3d.py:10-36): total pad from the input size, front = total / 2
void same_pad(int size, int k, int s, int* out_size, int* front) {
  const int p = (size % s == 0) ? std::max(k - s, 0) : std::max(k - (size % s), 0);
  *front = p / 2;
  *out_size = (size + p - k) / s + 1;
}
}  // namespace

void I3dModel::configure(const char* kv) {
  auto m = parse_kv(kv);
  num_classes = 400;
  if (m.count("num_classes")) num_classes = (int)m["num_classes"][0];
  ready = false;
}

I3dModel::Unit I3dModel::load_unit(svg_ctx* ctx, const std::string& p, int cin, int cout, int k, bool bn) {
  hipStream_t s = nullptr;
  Unit u;
  u.cin = cin; u.cin_pad = (int)align_up(cin, 4); u.cout = cout; u.k = k;
  const int taps = k * k * k;
  const Weight& w = ws.get(p + ".conv3d.weight", {cout, cin, k, k, k});
  u.w = (float*)ctx->dalloc((int64_t)cout * taps * u.cin_pad * sizeof(float));
  u.b = (float*)ctx->dalloc(cout * sizeof(float));
  if (bn) {
    pack_conv3d(w.f32, ws.get(p + ".bn.weight", {cout}).f32, ws.get(p + ".bn.bias", {cout}).f32, ws.get(p + ".bn.running_mean", {cout}).f32,
                ws.get(p + ".bn.running_var", {cout}).f32, nullptr, u.w, u.b, cout, cin, u.cin_pad, taps, 1e-5f, s);
  } else {
    pack_conv3d(w.f32, nullptr, nullptr, nullptr, nullptr, ws.has(p + ".conv3d.bias") ? ws.get(p + ".conv3d.bias", {cout}).f32 : nullptr, u.w, u.b,
                cout, cin, u.cin_pad, taps, 0.f, s);
  }
  HIP_OK(hipStreamSynchronize(s));
  ws.release(p + ".conv3d.weight");
  return u;
}

void I3dModel::finalize(svg_ctx* ctx, int64_t* n_params) {
  int64_t n = 0;
  for (auto& kv : ws.map)
    if (kv.first.find("num_batches_tracked") == std::string::npos) n += kv.second.numel;
  conv1a = load_unit(ctx, "Conv3d_1a_7x7", 3, 64, 7, true);
  conv2b = load_unit(ctx, "Conv3d_2b_1x1", 64, 64, 1, true);
  conv2c = load_unit(ctx, "Conv3d_2c_3x3", 64, 192, 3, true);
  mixed.clear();
  for (const MixedCfg& c : kMixed) {
    Mixed mx;
    const std::string p = c.name;
    mx.b0 = load_unit(ctx, p + ".b0", c.cin, c.oc[0], 1, true);
    mx.b1a = load_unit(ctx, p + ".b1a", c.cin, c.oc[1], 1, true);
    mx.b1b = load_unit(ctx, p + ".b1b", c.oc[1], c.oc[2], 3, true);
    mx.b2a = load_unit(ctx, p + ".b2a", c.cin, c.oc[3], 1, true);
    mx.b2b = load_unit(ctx, p + ".b2b", c.oc[3], c.oc[4], 3, true);
    mx.b3b = load_unit(ctx, p + ".b3b", c.cin, c.oc[5], 1, true);
    mixed.push_back(mx);
  }
  logits = load_unit(ctx, "logits", 1024, num_classes, 1, false);
  if (n_params) *n_params = n;
  ready = true;
}

namespace {
struct Act3 { float* p; int T, H, W, C; };
struct I3dRun {
  svg_ctx* ctx; hipStream_t s; int B;
  // Unit3D: 'SAME' conv + folded BatchNorm + ReLU into channels [coff, coff + cout) of `out` (allocated when out.p is null)
  Act3 unit(const I3dModel::Unit& u, const Act3& x, int stride, bool relu, Act3 out = {nullptr, 0, 0, 0, 0}, int coff = 0) {
    int To, Ho, Wo, pt, ph, pw;
    same_pad(x.T, u.k, stride, &To, &pt);
    same_pad(x.H, u.k, stride, &Ho, &ph);
    same_pad(x.W, u.k, stride, &Wo, &pw);
    if (!out.p) { out = {ctx->arena.get<float>((int64_t)B * To * Ho * Wo * u.cout), To, Ho, Wo, u.cout}; coff = 0; }
    SVG_CHECK(out.T == To && out.H == Ho && out.W == Wo && x.C == u.cin_pad, "i3d: shape mismatch in a unit (%d,%d,%d,%d vs %d,%d,%d,%d)", out.T, out.H,
              out.W, x.C, To, Ho, Wo, u.cin_pad);
    if (SVG_LAUNCHING(ctx)) {
      Conv3dArgs a{x.p, B, x.T, x.H, x.W, x.C, u.w, u.b, out.p, To, Ho, Wo, u.cout, out.C, coff, u.k, u.k, u.k, stride, stride, stride, pt, ph, pw, relu ? 1 : 0};
      ProfScope ps(ctx, PK_CONV3, s, 2.0 * B * To * Ho * Wo * (double)u.cout * u.k * u.k * u.k * u.cin, 0);
      conv3d_launch(a, s);
    }
    return out;
  }
  Act3 pool(const Act3& x, int kt, int khw, int st, int shw) {
    int To, Ho, Wo, pf[3];
    same_pad(x.T, kt, st, &To, &pf[0]);
    same_pad(x.H, khw, shw, &Ho, &pf[1]);
    same_pad(x.W, khw, shw, &Wo, &pf[2]);
    Act3 y{ctx->arena.get<float>((int64_t)B * To * Ho * Wo * x.C), To, Ho, Wo, x.C};
    const int k[3] = {kt, khw, khw}, sd[3] = {st, shw, shw};
    if (SVG_LAUNCHING(ctx)) { ProfScope ps(ctx, PK_ELT, s, 0, 0); maxpool3d_same(x.p, y.p, B, x.T, x.H, x.W, x.C, To, Ho, Wo, k, sd, pf, s); }
    return y;
  }
  Act3 inception(const I3dModel::Mixed& m, const Act3& x) {
    const int C = m.b0.cout + m.b1b.cout + m.b2b.cout + m.b3b.cout;
    Act3 out{ctx->arena.get<float>((int64_t)B * x.T * x.H * x.W * C), x.T, x.H, x.W, C};
    ctx->arena.push();
    unit(m.b0, x, 1, true, out, 0);
    unit(m.b1b, unit(m.b1a, x, 1, true), 1, true, out, m.b0.cout);
    unit(m.b2b, unit(m.b2a, x, 1, true), 1, true, out, m.b0.cout + m.b1b.cout);
    unit(m.b3b, pool(x, 3, 3, 1, 1), 1, true, out, m.b0.cout + m.b1b.cout + m.b2b.cout);
    ctx->arena.pop();
    return out;
  }
};
}  // namespace

void I3dModel::forward(svg_ctx* ctx, const float* x_ncthw, const uint8_t* video_u8, int B, int T, int H, int W, float* logits_out, hipStream_t s) {
  SVG_CHECK(ready, "i3d: svg_finalize has not been called");
  SVG_CHECK(B >= 1 && T >= 1 && H >= 1 && W >= 1, "i3d: empty input");
  run_planned(ctx, [&]() {
    I3dRun r{ctx, s, B};
    const float* xin = x_ncthw;
    int h = H, w = W;
    if (video_u8) {   // fvd_2.get_fvd_logits: preprocess (resize to 224, crop, [-1,1]) then the network
      float* pre = ctx->arena.get<float>((int64_t)B * 3 * T * 224 * 224);
      if (SVG_LAUNCHING(ctx)) fvd_preprocess(video_u8, pre, B, T, H, W, 224, s);
      xin = pre; h = w = 224;
    }
    Act3 x{ctx->arena.get<float>((int64_t)B * T * h * w * 4), T, h, w, 4};
    if (SVG_LAUNCHING(ctx)) ncthw_to_nthwc(xin, x.p, B, 3, T, h, w, 4, s);
    x = r.unit(conv1a, x, 2, true);
    x = r.pool(x, 1, 3, 1, 2);
    x = r.unit(conv2b, x, 1, true);
    x = r.unit(conv2c, x, 1, true);
    x = r.pool(x, 1, 3, 1, 2);
    x = r.inception(mixed[0], x);
    x = r.inception(mixed[1], x);
    x = r.pool(x, 3, 3, 2, 2);
    for (int i = 2; i <= 6; ++i) x = r.inception(mixed[i], x);
    x = r.pool(x, 2, 2, 2, 2);
    x = r.inception(mixed[7], x);
    x = r.inception(mixed[8], x);
    // avg_pool [2,7,7] stride 1, logits (1x1x1 conv with bias), squeeze, mean over time (pytorch_i3d.py:308-312)
    SVG_CHECK(x.H == 7 && x.W == 7 && x.T >= 2, "i3d: the [2,7,7] average pool needs a (>= 2) x 7 x 7 feature map, got %d x %d x %d (224 x 224 input, >= 9 frames)",
              x.T, x.H, x.W);
    const int To = x.T - 1;
    Act3 ap{ctx->arena.get<float>((int64_t)B * To * x.C), To, 1, 1, x.C};
    if (SVG_LAUNCHING(ctx)) avgpool_thw(x.p, ap.p, B, x.T, 49, x.C, 2, s);
    Act3 lg = r.unit(logits, ap, 1, false);
    if (SVG_LAUNCHING(ctx)) time_mean(lg.p, logits_out, B, To, num_classes, s);
  });
}

extern "C" {
int svg_i3d_forward(svg_ctx* ctx, const float* x, int B, int T, int H, int W, float* logits, void* stream) {
  try {
    SVG_CHECK(ctx && ctx->i3d, "i3d: model not configured");
    SVG_CHECK(x && logits, "i3d: null argument");
    ctx->i3d->forward(ctx, x, nullptr, B, T, H, W, logits, (hipStream_t)stream);
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}
int svg_fvd_logits(svg_ctx* ctx, const uint8_t* videos, int B, int T, int H, int W, float* logits, void* stream) {
  try {
    SVG_CHECK(ctx && ctx->i3d, "i3d: model not configured");
    SVG_CHECK(videos && logits, "fvd_logits: null argument");
    ctx->i3d->forward(ctx, nullptr, videos, B, T, H, W, logits, (hipStream_t)stream);
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}
int svg_fvd_preprocess(svg_ctx* ctx, const uint8_t* videos, int B, int T, int H, int W, float* out, void* stream) {
  try {
    SVG_CHECK(ctx && videos && out && B >= 1 && T >= 1 && H >= 1 && W >= 1, "fvd_preprocess: bad arguments");
    fvd_preprocess(videos, out, B, T, H, W, 224, (hipStream_t)stream);
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}
int svg_frechet_distance(svg_ctx* ctx, const float* x1, int n1, const float* x2, int n2, int d, double* out_host, void* stream) {
  try {
    SVG_CHECK(ctx && x1 && x2 && out_host, "frechet_distance: null argument");
    SVG_CHECK(n1 >= 2 && n2 >= 2 && d >= 2 && d % 2 == 0 && d <= 2048, "frechet_distance: n1=%d n2=%d d=%d (>= 2 samples each, even d <= 2048)", n1, n2, d);
    hipStream_t s = (hipStream_t)stream;
    double* out_dev = nullptr;
    run_planned(ctx, [&]() {
      double* ws = ctx->arena.get<double>((int64_t)6 * d * d + 8 * d + 8);
      out_dev = ws + (int64_t)6 * d * d + 8 * d;
      if (SVG_LAUNCHING(ctx)) frechet_distance(x1, n1, x2, n2, d, ws, out_dev, s);
    });
    HIP_OK(hipMemcpyAsync(out_host, out_dev, sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}
}
