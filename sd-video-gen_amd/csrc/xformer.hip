// Latent-sequence Transformer kernels (f32 end to end).
//
// The model is a weight stream: M = clips*T <= 64 rows against 0.44 G parameters, ~3 FLOP/B, HBM-bound.
// xf_gemm streams W[N][K] once with 16-B loads and feeds v_mfma_f32_16x16x4_f32 (exact f32 fma chain):
//   A operand = W tile (16 output columns n), B operand = X^T (16 rows m), D[i=n][j=m].
//   A lane loads W[n0 + (l&15)][k0 + 4(l>>4) .. +3] as one float4; MFMA j (0..3) consumes element j, so
//   k-slot q = l>>4 of MFMA j is k = k0 + 4q + j — X is loaded with the identical pattern.
// One workgroup = 8 waves = 16 output columns; the waves interleave over K in 16-wide steps and are
// reduced through LDS; an optional 2-way K split across workgroups finishes with f32 atomics onto a
// zeroed output (two addends: order-independent, so results stay bitwise reproducible).
#include "kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int XF_WAVES = 8;
constexpr int XF_MAXMT = 4;   // M <= 64

template <int MT>
__global__ void __launch_bounds__(512) xf_gemm_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                       const float* __restrict__ bias, float* __restrict__ Y, int M,
                                                       int N, int K, int relu_in, int ksplit) {
  __shared__ f32x4 red[XF_WAVES][MT][64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n0 = blockIdx.x * 16;
  const int kz = blockIdx.y;
  const int l15 = lane & 15, lq = lane >> 4;
  const int n = n0 + l15;
  const bool n_ok = n < N;
  const float* wrow = W + (int64_t)(n_ok ? n : 0) * K + 4 * lq;
  const float* xrow[MT];
  bool m_ok[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int m = t * 16 + l15;
    m_ok[t] = m < M;
    xrow[t] = X + (int64_t)(m_ok[t] ? m : 0) * K + 4 * lq;
  }
  f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int steps = K / 16;
  const int stride = XF_WAVES * ksplit;
  for (int st = kz * XF_WAVES + wid; st < steps; st += stride) {
    const int k0 = st * 16;
    f32x4 w = n_ok ? *(const f32x4*)(wrow + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 x[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      x[t] = m_ok[t] ? *(const f32x4*)(xrow[t] + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
      if (relu_in) {
#pragma unroll
        for (int j = 0; j < 4; ++j) x[t][j] = fmaxf(x[t][j], 0.f);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j], x[t][j], acc[t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < MT; ++t) red[wid][t][lane] = acc[t];
  __syncthreads();
  // D layout: col j = lane&15 -> m, row i = (lane>>4)*4 + reg -> n
  for (int idx = tid; idx < MT * 256; idx += 512) {
    const int t = idx >> 8, rem = idx & 255, ln = rem >> 2, rg = rem & 3;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < XF_WAVES; ++w) s += red[w][t][ln][rg];
    const int m = t * 16 + (ln & 15);
    const int nn = n0 + 4 * (ln >> 4) + rg;
    if (m < M && nn < N) {
      if (kz == 0 && bias) s += bias[nn];
      if (ksplit > 1) atomicAdd(Y + (int64_t)m * N + nn, s);
      else Y[(int64_t)m * N + nn] = s;
    }
  }
}

// y = LayerNorm(x + r) * g + b over rows of d (r may be null)
__global__ void __launch_bounds__(256) xf_add_ln_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                         const float* __restrict__ g, const float* __restrict__ b,
                                                         float* __restrict__ y, int d, float eps) {
  __shared__ float red[4];
  __shared__ float stat[2];
  const int row = blockIdx.x;
  const float* xr = x + (int64_t)row * d;
  const float* rr = r ? r + (int64_t)row * d : nullptr;
  float v[12];   // d <= 3072
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const int c = threadIdx.x + 256 * i;
    v[i] = 0.f;
    if (c < d) { v[i] = xr[c] + (rr ? rr[c] : 0.f); s += v[i]; }
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) stat[0] = (red[0] + red[1] + red[2] + red[3]) / (float)d;
  __syncthreads();
  const float mean = stat[0];
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const int c = threadIdx.x + 256 * i;
    if (c < d) { const float t = v[i] - mean; q += t * t; }
  }
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
  __syncthreads();
  if (threadIdx.x == 0) stat[1] = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)d + eps);
  __syncthreads();
  const float rstd = stat[1];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const int c = threadIdx.x + 256 * i;
    if (c < d) y[(int64_t)row * d + c] = (v[i] - mean) * rstd * g[c] + b[c];
  }
}

// emb rows are (b,t) batch-first (d - d_txt wide); out rows are (t,b) sequence-first; PE row chosen per batch row;
// the last d_txt channels of every token are the per-clip text embedding (models/transformer_text.py:82-92)
__global__ void xf_embed_post_kernel(const float* __restrict__ emb, const float* __restrict__ pe, const int32_t* __restrict__ pe_row,
                                     const float* __restrict__ text, int d_txt, float* __restrict__ y, int B, int T, int d, float scale) {
  const int d_img = d - d_txt;
  const int64_t total = (int64_t)B * T * d;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % d);
    const int t = (int)((idx / d) % T);
    const int b = (int)(idx / ((int64_t)d * T));
    const int pr = pe_row ? pe_row[b] : b;
    const float v = (c < d_img) ? emb[((int64_t)b * T + t) * d_img + c] : text[(int64_t)b * d_txt + (c - d_img)];
    y[((int64_t)t * B + b) * d + c] = v * scale + pe[(int64_t)pr * d + c];
  }
}

// one workgroup per (batch row, head); Tq, Tk <= 16
__global__ void __launch_bounds__(256) xf_attention_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                            const float* __restrict__ v, int ldk, const float* __restrict__ mask,
                                                            float* __restrict__ o, int Tq, int Tk, int B, int heads, int hd) {
  __shared__ float sc[16][17];
  const int b = blockIdx.x, hh = blockIdx.y;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const float scale = rsqrtf((float)hd);
  const int d = heads * hd;
  for (int p = wid; p < Tq * Tk; p += 4) {
    const int i = p / Tk, j = p - i * Tk;
    const float* qr = q + ((int64_t)i * B + b) * ldq + hh * hd;
    const float* kr = k + ((int64_t)j * B + b) * ldk + hh * hd;
    float s = 0.f;
    for (int c = lane; c < hd; c += 64) s += qr[c] * kr[c];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) sc[i][j] = s * scale + (mask ? mask[i * Tk + j] : 0.f);
  }
  __syncthreads();
  if (threadIdx.x < Tq) {
    const int i = threadIdx.x;
    float mx = -INFINITY;
    for (int j = 0; j < Tk; ++j) mx = fmaxf(mx, sc[i][j]);
    float sum = 0.f;
    for (int j = 0; j < Tk; ++j) { const float e = expf(sc[i][j] - mx); sc[i][j] = e; sum += e; }
    const float inv = 1.f / sum;
    for (int j = 0; j < Tk; ++j) sc[i][j] *= inv;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < Tq * hd; idx += 256) {
    const int i = idx / hd, c = idx - i * hd;
    float acc = 0.f;
    for (int j = 0; j < Tk; ++j) acc += sc[i][j] * v[((int64_t)j * B + b) * ldk + hh * hd + c];
    o[((int64_t)i * B + b) * d + hh * hd + c] = acc;
  }
}

}  // namespace

void xf_gemm(svg_ctx* ctx, const float* X, const float* W, const float* bias, float* Y, int M, int N, int K, int relu_in,
             hipStream_t s) {
  SVG_CHECK(K % 16 == 0, "xf_gemm: K=%d must be a multiple of 16", K);
  SVG_CHECK(M >= 1 && M <= 16 * XF_MAXMT, "xf_gemm: M=%d must be in 1..%d", M, 16 * XF_MAXMT);
  if (!SVG_LAUNCHING(ctx)) return;
  const int nb = cdiv(N, 16);
  int ksplit = (nb < 192 && K >= 16 * XF_WAVES * 2) ? 2 : 1;
  ProfScope ps(ctx, PK_XF_GEMM, s, 2.0 * M * (double)N * K, 4.0 * ((double)N * K + (double)M * K + (double)M * N));
  if (ksplit > 1) HIP_OK(hipMemsetAsync(Y, 0, (size_t)M * N * sizeof(float), s));
  dim3 grid(nb, ksplit);
  const int mt = cdiv(M, 16);
  switch (mt) {
    case 1: hipLaunchKernelGGL((xf_gemm_kernel<1>), grid, dim3(512), 0, s, X, W, bias, Y, M, N, K, relu_in, ksplit); break;
    case 2: hipLaunchKernelGGL((xf_gemm_kernel<2>), grid, dim3(512), 0, s, X, W, bias, Y, M, N, K, relu_in, ksplit); break;
    case 3: hipLaunchKernelGGL((xf_gemm_kernel<3>), grid, dim3(512), 0, s, X, W, bias, Y, M, N, K, relu_in, ksplit); break;
    default: hipLaunchKernelGGL((xf_gemm_kernel<4>), grid, dim3(512), 0, s, X, W, bias, Y, M, N, K, relu_in, ksplit); break;
  }
  check_launch("xf_gemm");
}

void xf_add_ln(const float* x, const float* r, const float* g, const float* b, float* y, int M, int d, float eps, hipStream_t s) {
  SVG_CHECK(d <= 3072, "xf_add_ln: d=%d too large", d);
  hipLaunchKernelGGL(xf_add_ln_kernel, dim3(M), dim3(256), 0, s, x, r, g, b, y, d, eps);
  check_launch("xf_add_ln");
}

void xf_embed_post(const float* emb, const float* pe, const int32_t* pe_row, const float* text, int d_txt, float* y, int B, int T, int d,
                   float scale, hipStream_t s) {
  const int64_t total = (int64_t)B * T * d;
  hipLaunchKernelGGL(xf_embed_post_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 2048)), dim3(256), 0, s, emb, pe, pe_row, text,
                     d_txt, y, B, T, d, scale);
  check_launch("xf_embed_post");
}

void xf_attention(const float* q, int ldq, const float* k, const float* v, int ldk, const float* mask, float* o, int Tq, int Tk,
                  int B, int heads, int hd, hipStream_t s) {
  SVG_CHECK(Tq <= 16 && Tk <= 16, "xf_attention: sequence length %d/%d > 16", Tq, Tk);
  hipLaunchKernelGGL(xf_attention_kernel, dim3(B, heads), dim3(256), 0, s, q, ldq, k, v, ldk, mask, o, Tq, Tk, B, heads, hd);
  check_launch("xf_attention");
}
