// Element-wise and layout kernels of the sampling path (HBM-bound, vectorised where the layout allows).
#include "kernels.h"

namespace SDNS {


namespace {

#define GRID_STRIDE(idx, total) \
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (total); idx += (int64_t)gridDim.x * blockDim.x)

inline dim3 grid_for(int64_t total) { return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, 8192))); }

// nearest: src = floor(dst * in / out)  (torch F.interpolate mode='nearest', predict.py:158,178)
__device__ __forceinline__ int nearest_src(int d, int in, int out) {
  int s = (int)floorf((float)d * ((float)in / (float)out));
  return min(s, in - 1);
}

// sd_utils.py:135-138: x/255 -> 2*(x-0.5); channels padded 3 -> 8 with zeros
__global__ void img_to_act_kernel(const uint8_t* __restrict__ img, h16* __restrict__ out, int N, int sh, int sw, int H, int W) {
  const int64_t total = (int64_t)N * H * W;
  GRID_STRIDE(idx, total) {
    const int x = (int)(idx % W);
    const int y = (int)((idx / W) % H);
    const int n = (int)(idx / ((int64_t)W * H));
    const int sy = (sh == H) ? y : nearest_src(y, sh, H);
    const int sx = (sw == W) ? x : nearest_src(x, sw, W);
    const uint8_t* p = img + (((int64_t)n * sh + sy) * sw + sx) * 3;
    h16x8 o;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float f = (float)p[c] / 255.0f;
      o[c] = (h16)(2.f * (f - 0.5f));
    }
#pragma unroll
    for (int c = 3; c < 8; ++c) o[c] = (h16)0.f;
    *(h16x8*)(out + idx * 8) = o;
  }
}

// sd_utils.py:164-166: (x/2+.5).clamp(0,1) -> *255 -> round (half to even) -> u8, with nearest resize
__global__ void act_to_img_kernel(const float* __restrict__ x, int ldc, uint8_t* __restrict__ img, float* __restrict__ fout,
                                  int N, int h, int w, int oh, int ow) {
  if (img) {
    const int64_t total = (int64_t)N * oh * ow;
    GRID_STRIDE(idx, total) {
      const int ox = (int)(idx % ow);
      const int oy = (int)((idx / ow) % oh);
      const int n = (int)(idx / ((int64_t)ow * oh));
      const int sy = (oh == h) ? oy : nearest_src(oy, h, oh);
      const int sx = (ow == w) ? ox : nearest_src(ox, w, ow);
      const float* p = x + (((int64_t)n * h + sy) * w + sx) * ldc;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float f = p[c] / 2.f + 0.5f;
        f = fminf(fmaxf(f, 0.f), 1.f);
        img[idx * 3 + c] = (uint8_t)rintf(f * 255.f);
      }
    }
  }
  if (fout) {
    const int64_t total = (int64_t)N * h * w;
    GRID_STRIDE(idx, total) {
      const int px = (int)(idx % ((int64_t)h * w));
      const int n = (int)(idx / ((int64_t)h * w));
      const float* p = x + idx * ldc;
#pragma unroll
      for (int c = 0; c < 3; ++c) fout[((int64_t)n * 3 + c) * h * w + px] = p[c];
    }
  }
}

__global__ void nchw_to_act_kernel(const float* __restrict__ x, h16* __restrict__ out, int N, int C, int hw, int Cpad, float scale) {
  const int64_t total = (int64_t)N * hw;
  GRID_STRIDE(idx, total) {
    const int px = (int)(idx % hw);
    const int n = (int)(idx / hw);
    for (int c = 0; c < Cpad; ++c)
      out[idx * Cpad + c] = (h16)((c < C) ? x[((int64_t)n * C + c) * hw + px] * scale : 0.f);
  }
}

__global__ void nchw_to_actf32_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int C, int hw, float scale) {
  const int64_t total = (int64_t)N * hw;
  GRID_STRIDE(idx, total) {
    const int px = (int)(idx % hw);
    const int n = (int)(idx / hw);
    for (int c = 0; c < C; ++c) out[idx * C + c] = x[((int64_t)n * C + c) * hw + px] * scale;
  }
}
__global__ void actf32_pad_h16_kernel(const float* __restrict__ x, int C, h16* __restrict__ out, int Cpad, int64_t P) {
  GRID_STRIDE(idx, P) {
    for (int c = 0; c < Cpad; ++c) out[idx * Cpad + c] = (h16)((c < C) ? x[idx * C + c] : 0.f);
  }
}

__global__ void actf32_to_nchw_kernel(const float* __restrict__ x, int ld, float* __restrict__ out, int N, int C, int hw) {
  const int64_t total = (int64_t)N * hw;
  GRID_STRIDE(idx, total) {
    const int px = (int)(idx % hw);
    const int n = (int)(idx / hw);
    for (int c = 0; c < C; ++c) out[((int64_t)n * C + c) * hw + px] = x[idx * ld + c];
  }
}

__global__ void pixel_linear_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ b,
                                    float* __restrict__ y, int ldy, int64_t P, int Cin, int Cout) {
  GRID_STRIDE(p, P) {
    float in[8];
    for (int i = 0; i < Cin; ++i) in[i] = x[p * ldx + i];
    for (int o = 0; o < Cout; ++o) {
      float acc = b ? b[o] : 0.f;
      for (int i = 0; i < Cin; ++i) acc += w[o * Cin + i] * in[i];
      y[p * ldy + o] = acc;
    }
  }
}

// DiagonalGaussianDistribution.sample (diffusers vae.py): logvar clamp [-30,20], std = exp(.5 logvar);
// sd_utils.py:142-143: .sample() then *= 0.18215
__global__ void vae_sample_kernel(const float* __restrict__ mom, int ldm, const float* __restrict__ eps, float* __restrict__ z,
                                  float* __restrict__ mom_nchw, int N, int hw) {
  const int64_t total = (int64_t)N * hw;
  GRID_STRIDE(idx, total) {
    const int px = (int)(idx % hw);
    const int n = (int)(idx / hw);
    const float* m = mom + idx * ldm;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float mean = m[c];
      const float lv = fminf(fmaxf(m[4 + c], -30.f), 20.f);
      const int64_t o = ((int64_t)n * 4 + c) * hw + px;
      float v = mean;
      if (eps) v += expf(0.5f * lv) * eps[o];
      z[o] = v * 0.18215f;
      if (mom_nchw) {
        mom_nchw[((int64_t)n * 8 + c) * hw + px] = mean;
        mom_nchw[((int64_t)n * 8 + 4 + c) * hw + px] = m[4 + c];
      }
    }
  }
}

__global__ void concat_kernel(const h16* __restrict__ a, int Ca, const h16* __restrict__ b, int Cb, h16* __restrict__ out, int64_t P) {
  const int CV = (Ca + Cb) / 8;
  const int64_t total = P * CV;
  GRID_STRIDE(idx, total) {
    const int cv = (int)(idx % CV);
    const int64_t p = idx / CV;
    const int c0 = cv * 8;
    h16x8 v = (c0 < Ca) ? *(const h16x8*)(a + p * Ca + c0) : *(const h16x8*)(b + p * Cb + (c0 - Ca));
    *(h16x8*)(out + p * (Ca + Cb) + c0) = v;
  }
}

__global__ void resize_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int N, int sh, int sw, int C, int dh, int dw) {
  const int64_t total = (int64_t)N * dh * dw;
  GRID_STRIDE(idx, total) {
    const int x = (int)(idx % dw);
    const int y = (int)((idx / dw) % dh);
    const int n = (int)(idx / ((int64_t)dw * dh));
    const int sy = nearest_src(y, sh, dh), sx = nearest_src(x, sw, dw);
    for (int c = 0; c < C; ++c) dst[idx * C + c] = src[(((int64_t)n * sh + sy) * sw + sx) * C + c];
  }
}

__global__ void f32_to_h16_kernel(const float* __restrict__ x, h16* __restrict__ y, int64_t n) {
  GRID_STRIDE(i, n) y[i] = (h16)x[i];
}
__global__ void h16_to_f32_kernel(const h16* __restrict__ x, float* __restrict__ y, int64_t n) {
  GRID_STRIDE(i, n) y[i] = (float)x[i];
}
__global__ void silu_kernel(const h16* __restrict__ x, h16* __restrict__ y, int64_t n) {
  GRID_STRIDE(i, n) { float f = (float)x[i]; y[i] = (h16)(f / (1.f + __expf(-f))); }
}

// diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]
__global__ void timestep_embed_kernel(const float* __restrict__ t, h16* __restrict__ out, int N, int dim) {
  const int half = dim / 2;
  const int64_t total = (int64_t)N * half;
  GRID_STRIDE(idx, total) {
    const int i = (int)(idx % half);
    const int n = (int)(idx / half);
    const float freq = expf(-logf(10000.f) * (float)i / (float)half);
    const float a = t[n] * freq;
    out[(int64_t)n * dim + i] = (h16)cosf(a);
    out[(int64_t)n * dim + half + i] = (h16)sinf(a);
  }
}

// DDIMScheduler.step (eta 0, clip_sample) with the classifier-free-guidance combine in front
// (sd_utils.py:256-260)
__global__ void ddim_step_kernel(const float* __restrict__ z, const float* __restrict__ eu, const float* __restrict__ ec, float guidance,
                                 float* __restrict__ zo, int64_t n, float sa, float s1a, float sap, float s1ap) {
  GRID_STRIDE(i, n) {
    float e = eu[i];
    if (ec) e = e + guidance * (ec[i] - e);
    float x0 = (z[i] - s1a * e) / sa;
    x0 = fminf(fmaxf(x0, -1.f), 1.f);
    zo[i] = sap * x0 + s1ap * e;
  }
}
// The same step with its scalars read from a device table row chosen by a device counter, so that a captured graph of one DDIM
// step replays for every timestep without new kernel arguments: tab[i] = {t, sqrt(a_t), sqrt(1 - a_t), sqrt(a_prev), sqrt(1 - a_prev)}
__global__ void ddim_step_tab_kernel(const float* __restrict__ z, const float* __restrict__ eu, const float* __restrict__ ec, float guidance,
                                     float* __restrict__ zo, int64_t n, const float* __restrict__ tab, const int* __restrict__ idx) {
  const float* r = tab + 5 * idx[0];
  const float sa = r[1], s1a = r[2], sap = r[3], s1ap = r[4];
  GRID_STRIDE(i, n) {
    float e = eu[i];
    if (ec) e = e + guidance * (ec[i] - e);
    float x0 = (z[i] - s1a * e) / sa;
    x0 = fminf(fmaxf(x0, -1.f), 1.f);
    zo[i] = sap * x0 + s1ap * e;
  }
}
__global__ void ddim_tvec_kernel(float* __restrict__ tvec, int nb, const float* __restrict__ tab, const int* __restrict__ idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nb) tvec[i] = tab[5 * idx[0]];
}
__global__ void ddim_bump_kernel(int* __restrict__ idx) { idx[0] += 1; }
__global__ void add_noise_kernel(const float* __restrict__ x0, const float* __restrict__ nz, float* __restrict__ out, int64_t n, float sa, float s1a) {
  GRID_STRIDE(i, n) out[i] = sa * x0[i] + s1a * nz[i];
}

}  // namespace

void img_to_act(const uint8_t* img, h16* out, int N, int sh, int sw, int H, int W, hipStream_t s) {
  hipLaunchKernelGGL(img_to_act_kernel, grid_for((int64_t)N * H * W), dim3(256), 0, s, img, out, N, sh, sw, H, W);
  check_launch("img_to_act");
}
void act_to_img(const float* x, int ldc, uint8_t* img, float* fout, int N, int h, int w, int oh, int ow, hipStream_t s) {
  hipLaunchKernelGGL(act_to_img_kernel, grid_for((int64_t)N * std::max(h * w, oh * ow)), dim3(256), 0, s, x, ldc, img, fout, N, h, w, oh, ow);
  check_launch("act_to_img");
}
void nchw_to_act(const float* x, h16* out, int N, int C, int h, int w, int Cpad, float scale, hipStream_t s) {
  hipLaunchKernelGGL(nchw_to_act_kernel, grid_for((int64_t)N * h * w), dim3(256), 0, s, x, out, N, C, h * w, Cpad, scale);
  check_launch("nchw_to_act");
}
void nchw_to_actf32(const float* x, float* out, int N, int C, int h, int w, float scale, hipStream_t s) {
  hipLaunchKernelGGL(nchw_to_actf32_kernel, grid_for((int64_t)N * h * w), dim3(256), 0, s, x, out, N, C, h * w, scale);
  check_launch("nchw_to_actf32");
}
void actf32_pad_h16(const float* x, int C, h16* out, int Cpad, int64_t P, hipStream_t s) {
  hipLaunchKernelGGL(actf32_pad_h16_kernel, grid_for(P), dim3(256), 0, s, x, C, out, Cpad, P);
  check_launch("actf32_pad_h16");
}
void actf32_to_nchw(const float* x, int ld, float* out, int N, int C, int h, int w, hipStream_t s) {
  hipLaunchKernelGGL(actf32_to_nchw_kernel, grid_for((int64_t)N * h * w), dim3(256), 0, s, x, ld, out, N, C, h * w);
  check_launch("actf32_to_nchw");
}
void pixel_linear_f32(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int64_t P, int Cin, int Cout, hipStream_t s) {
  SVG_CHECK(Cin <= 8, "pixel_linear: Cin %d > 8", Cin);
  hipLaunchKernelGGL(pixel_linear_kernel, grid_for(P), dim3(256), 0, s, x, ldx, w, b, y, ldy, P, Cin, Cout);
  check_launch("pixel_linear");
}
void vae_sample(const float* mom, int ldm, const float* eps, float* z, float* mom_nchw, int N, int h, int w, hipStream_t s) {
  hipLaunchKernelGGL(vae_sample_kernel, grid_for((int64_t)N * h * w), dim3(256), 0, s, mom, ldm, eps, z, mom_nchw, N, h * w);
  check_launch("vae_sample");
}
void concat_channels(const h16* a, int Ca, const h16* b, int Cb, h16* out, int64_t P, hipStream_t s) {
  SVG_CHECK(Ca % 8 == 0 && Cb % 8 == 0, "concat: channels must be multiples of 8");
  hipLaunchKernelGGL(concat_kernel, grid_for(P * ((Ca + Cb) / 8)), dim3(256), 0, s, a, Ca, b, Cb, out, P);
  check_launch("concat");
}
// f32 planes (P, h, w) -> (P, oh, ow), bilinear with torch's align_corners=False convention (F.interpolate(mode='bilinear')):
// src = max((dst + 0.5) * in / out - 0.5, 0), the upper neighbour clamped to the last row / column
static __global__ void resize_bilinear_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int P, int h, int w, int oh, int ow) {
  const int64_t total = (int64_t)P * oh * ow;
  const float sy = (float)h / (float)oh, sx = (float)w / (float)ow;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % ow), y = (int)((i / ow) % oh);
    const int64_t p = i / ((int64_t)ow * oh);
    const float fy = fmaxf(((float)y + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf(((float)x + 0.5f) * sx - 0.5f, 0.f);
    const int y0 = min((int)fy, h - 1), x0 = min((int)fx, w - 1);
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float* s = src + p * h * w;
    const float top = s[y0 * w + x0] * (1.f - lx) + s[y0 * w + x1] * lx;
    const float bot = s[y1 * w + x0] * (1.f - lx) + s[y1 * w + x1] * lx;
    dst[i] = top * (1.f - ly) + bot * ly;
  }
}

void resize_bilinear_f32(const float* src, float* dst, int P, int h, int w, int oh, int ow, hipStream_t s) {
  const int64_t total = (int64_t)P * oh * ow;
  hipLaunchKernelGGL(resize_bilinear_f32_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, s, src, dst, P, h, w, oh, ow);
  check_launch("resize_bilinear_f32");
}

void resize_nearest_u8(const uint8_t* src, uint8_t* dst, int N, int sh, int sw, int C, int dh, int dw, hipStream_t s) {
  hipLaunchKernelGGL(resize_u8_kernel, grid_for((int64_t)N * dh * dw), dim3(256), 0, s, src, dst, N, sh, sw, C, dh, dw);
  check_launch("resize_u8");
}
void f32_to_h16(const float* x, h16* y, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(f32_to_h16_kernel, grid_for(n), dim3(256), 0, s, x, y, n);
  check_launch("f32_to_h16");
}
void h16_to_f32(const h16* x, float* y, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(h16_to_f32_kernel, grid_for(n), dim3(256), 0, s, x, y, n);
  check_launch("h16_to_f32");
}
void silu_h16(const h16* x, h16* y, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(silu_kernel, grid_for(n), dim3(256), 0, s, x, y, n);
  check_launch("silu");
}
void timestep_embed(const float* t, h16* out, int N, int dim, hipStream_t s) {
  hipLaunchKernelGGL(timestep_embed_kernel, grid_for((int64_t)N * dim / 2), dim3(256), 0, s, t, out, N, dim);
  check_launch("timestep_embed");
}
void ddim_step(const float* z, const float* eu, const float* ec, float guidance, float* zo, int64_t n, float sa, float s1a,
               float sap, float s1ap, hipStream_t s) {
  hipLaunchKernelGGL(ddim_step_kernel, grid_for(n), dim3(256), 0, s, z, eu, ec, guidance, zo, n, sa, s1a, sap, s1ap);
  check_launch("ddim_step");
}
void ddim_step_tab(const float* z, const float* eu, const float* ec, float guidance, float* zo, int64_t n, const float* tab, const int* idx,
                   hipStream_t s) {
  hipLaunchKernelGGL(ddim_step_tab_kernel, grid_for(n), dim3(256), 0, s, z, eu, ec, guidance, zo, n, tab, idx);
  check_launch("ddim_step_tab");
}
void ddim_tvec(float* tvec, int nb, const float* tab, const int* idx, hipStream_t s) {
  hipLaunchKernelGGL(ddim_tvec_kernel, dim3((nb + 63) / 64), dim3(64), 0, s, tvec, nb, tab, idx);
  check_launch("ddim_tvec");
}
void ddim_bump(int* idx, hipStream_t s) {
  hipLaunchKernelGGL(ddim_bump_kernel, dim3(1), dim3(1), 0, s, idx);
  check_launch("ddim_bump");
}

namespace {
__global__ void fill_f32_kernel(float* __restrict__ p, int64_t n, float v) {
  GRID_STRIDE(i, n) p[i] = v;
}
}  // namespace
void fill_f32(float* p, int64_t n, float v, hipStream_t s) {
  hipLaunchKernelGGL(fill_f32_kernel, grid_for(n), dim3(256), 0, s, p, n, v);
  check_launch("fill_f32");
}
void add_noise(const float* x0, const float* nz, float* out, int64_t n, float sa, float s1a, hipStream_t s) {
  hipLaunchKernelGGL(add_noise_kernel, grid_for(n), dim3(256), 0, s, x0, nz, out, n, sa, s1a);
  check_launch("add_noise");
}

}  // namespace SDNS
