// Training-step kernels of the latent Transformer (f32 end to end, like the reference's trainer: trainers/trainer.py:111-190
// runs the model in fp32, no autocast): the backward GEMMs, LayerNorm / attention / embedding backward, dropout, the criterion
// (trainers/trainer.py:65-109 + models/contrastive_loss.py:7-60) with its gradient, and torch.optim.Adam.
//
// The two backward GEMM forms, both on v_mfma_f32_16x16x4_f32 (exact f32 fma chains) with one float4 global load per operand
// and lane serving four MFMA tiles — the element j of a lane's float4 is column (or row) 4*lane + j of tile j, so a wave-load
// is whole 256-byte row segments and the accumulators of one lane are four adjacent output columns (float4 stores):
//   xf_gemm_tn   dW[n][k] (+)= sum_m dY[m][n] X[m][k]     contraction over the few rows: the N x K gradient write is the traffic
//   xf_gemm_nn   dX[m][k]  =  sum_n dY[m][n] W[n][k]      W streamed once; the contraction is split over waves (LDS reduce in a
//                                                          fixed order) and workgroups (f32 slabs added in a fixed order by the
//                                                          finishing kernel, which also applies the ReLU/dropout gate and the
//                                                          residual-branch add)
// Every reduction here has a fixed order: two runs of a step give the same bits.
#include "kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// ---- dropout: counter-based, keyed by (seed, site, element index) so that backward regenerates the forward mask -------------
__device__ __forceinline__ float drop_keep(const XfDrop d, uint64_t idx) {
  uint64_t x = *d.seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(d.site + 1));
  x += idx * 0xD1B54A32D192ED03ull;
  x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
  x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
  x ^= x >> 32;
  const uint32_t u = (uint32_t)(x >> 40);                     // 24 uniform bits
  return ((float)u * (1.f / 16777216.f)) >= d.p ? 1.f / (1.f - d.p) : 0.f;
}
__device__ __forceinline__ float drop_apply(const XfDrop d, uint64_t idx) { return d.p > 0.f ? drop_keep(d, idx) : 1.f; }

__global__ void drop_mask_kernel(const XfDrop d, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = drop_apply(d, i);
}

// ---- dW = dY^T X (+ db = column sums of dY) -------------------------------------------------------------------------------
// grid (ceil(K/128), ceil(N/128)), 4 waves as 2 x 2, wave tile 64 (n) x 64 (k).  The contraction runs over the few rows m: the
// operands come straight from L2 (dY and X are a few hundred KB), eight row-steps (32 rows) per batch with the next batch's loads in
// flight under this batch's 128 MFMAs.  The waves of k-block 0 also reduce their dY values over m: db[n] (rows ascending per lane
// group, then the four lane groups in order — fixed order).
constexpr int TN_U = 8;
__global__ void __launch_bounds__(256) xf_gemm_tn_kernel(const float* __restrict__ dY, int ldy, const float* __restrict__ X, int ldx,
                                                          float* __restrict__ dW, float* __restrict__ db, int M, int N, int K, int accumulate) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int nb = blockIdx.y * 128 + (wid >> 1) * 64, kb = blockIdx.x * 128 + (wid & 1) * 64;
  const int n4 = nb + 4 * l15, k4 = kb + 4 * l15;
  const bool n_ok = n4 < N, k_ok = k4 < K;                   // N, K multiples of 4
  const bool do_bias = db != nullptr && blockIdx.x == 0 && (wid & 1) == 0;
  const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = zero;
  f32x4 bsum = zero;
  f32x4 a0[TN_U], b0[TN_U], a1[TN_U], b1[TN_U];
  auto load = [&](int m0, f32x4* a, f32x4* b) {
#pragma unroll
    for (int u = 0; u < TN_U; ++u) {
      const int m = m0 + 4 * u + lq;
      a[u] = (m < M && n_ok) ? *(const f32x4*)(dY + (int64_t)m * ldy + n4) : zero;
      b[u] = (m < M && k_ok) ? *(const f32x4*)(X + (int64_t)m * ldx + k4) : zero;
    }
  };
  auto mac = [&](const f32x4* a, const f32x4* b) {
#pragma unroll
    for (int u = 0; u < TN_U; ++u) {
      if (do_bias) bsum += a[u];
#pragma unroll
      for (int jr = 0; jr < 4; ++jr)
#pragma unroll
        for (int jc = 0; jc < 4; ++jc) acc[jr][jc] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][jr], b[u][jc], acc[jr][jc], 0, 0, 0);
    }
  };
  load(0, a0, b0);
  for (int m0 = 0; m0 < M; m0 += 8 * TN_U) {
    load(m0 + 4 * TN_U, a1, b1);
    mac(a0, b0);
    load(m0 + 8 * TN_U, a0, b0);
    mac(a1, b1);
  }
  if (do_bias) {
    // lane groups lq = 0..3 hold the rows m = lq mod 4: add them in order
    f32x4 t = bsum;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v0 = __shfl(bsum[j], l15), v1 = __shfl(bsum[j], l15 + 16), v2 = __shfl(bsum[j], l15 + 32), v3 = __shfl(bsum[j], l15 + 48);
      t[j] = ((v0 + v1) + v2) + v3;
    }
    if (lq == 0 && n_ok) {
      if (accumulate) t += *(const f32x4*)(db + n4);
      *(f32x4*)(db + n4) = t;
    }
  }
  if (!k_ok) return;
#pragma unroll
  for (int jr = 0; jr < 4; ++jr)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = nb + 16 * lq + 4 * e + jr;
      if (n < N) {
        float* p = dW + (int64_t)n * K + k4;
        f32x4 v = f32x4{acc[jr][0][e], acc[jr][1][e], acc[jr][2][e], acc[jr][3][e]};
        if (accumulate) v += *(const f32x4*)p;
        *(f32x4*)p = v;
      }
    }
}

// ---- dX = dY W -----------------------------------------------------------------------------------------------------------
// grid (ceil(K/64), Z): a workgroup owns 64 output columns and the n range [z*chunk, (z+1)*chunk) of the contraction, its 4 waves
// interleave over that range in steps of 16; rows m0 .. m0 + 16*MT of dY.  Partial sums -> slab z of `slabs` ([Z][M][K]).
template <int MT>
__global__ void __launch_bounds__(256) xf_gemm_nn_kernel(const float* __restrict__ dY, int ldy, const float* __restrict__ W,
                                                          float* __restrict__ slabs, int m_base, int M, int N, int K, int chunk) {
  __shared__ f32x4 red[MT][4][64];                           // one wave's accumulators; waves add in turn
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int kb = blockIdx.x * 64, z = blockIdx.y;
  const int k4 = kb + 4 * l15;
  const bool k_ok = k4 < K;
  const int n_lo = z * chunk, n_hi = min(N, n_lo + chunk);
  const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acc[MT][4];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int jc = 0; jc < 4; ++jc) acc[t][jc] = zero;
  f32x4 a[MT], b[4], an[MT], bn[4];
  auto load = [&](int nn0, f32x4* av, f32x4* bv) {
    const int n = nn0 + 4 * lq;                              // N multiple of 4: a float4 is inside or outside
    const bool ok = nn0 < n_hi && n < n_hi;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int m = m_base + 16 * t + l15;
      av[t] = (ok && m < M) ? *(const f32x4*)(dY + (int64_t)m * ldy + n) : zero;
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) bv[jj] = (ok && k_ok) ? *(const f32x4*)(W + (int64_t)(n + jj) * K + k4) : zero;
  };
  int nn0 = n_lo + 16 * wid;
  load(nn0, a, b);
  for (; nn0 < n_hi; nn0 += 64) {
    load(nn0 + 64, an, bn);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int jc = 0; jc < 4; ++jc) acc[t][jc] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][jj], b[jj][jc], acc[t][jc], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < MT; ++t) a[t] = an[t];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) b[jj] = bn[jj];
  }
  // waves 0..3 add into LDS in turn (fixed order)
  for (int w = 0; w < 4; ++w) {
    if (wid == w) {
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int jc = 0; jc < 4; ++jc) {
          if (w == 0) red[t][jc][lane] = acc[t][jc];
          else red[t][jc][lane] += acc[t][jc];
        }
    }
    __syncthreads();
  }
  if (wid != 0 || !k_ok) return;
  float* out = slabs + (int64_t)z * M * K;
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = m_base + 16 * t + 4 * lq + e;
      if (m < M)
        *(f32x4*)(out + (int64_t)m * K + k4) = f32x4{red[t][0][lane][e], red[t][1][lane][e], red[t][2][lane][e], red[t][3][lane][e]};
    }
}

// out[i] = gate( sum_z slabs[z][i] ) + add[i];  gate: x * (gate[i] > 0 ? gate_scale : 0)  (ReLU and its dropout in one test, see
// relu_drop_kernel), z ascending
__global__ void __launch_bounds__(256) xf_nn_finish_kernel(const float* __restrict__ slabs, int Z, int64_t MN, const float* __restrict__ gate,
                                                            float gate_scale, const float* __restrict__ add, float* __restrict__ out) {
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < MN; i += (int64_t)gridDim.x * 1024) {
    f32x4 v = *(const f32x4*)(slabs + i);
    for (int z = 1; z < Z; ++z) v += *(const f32x4*)(slabs + (int64_t)z * MN + i);
    if (gate) {
      const f32x4 g = *(const f32x4*)(gate + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = g[j] > 0.f ? v[j] * gate_scale : 0.f;
    }
    if (add) v += *(const f32x4*)(add + i);
    *(f32x4*)(out + i) = v;
  }
}

// ---- forward pieces that keep what backward needs ---------------------------------------------------------------------------
// r = dropout(relu(h)): r > 0 exactly where the gradient passes (scaled by 1/(1-p))
__global__ void relu_drop_kernel(const float* __restrict__ h, float* __restrict__ r, int64_t n, const XfDrop d) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    r[i] = fmaxf(h[i], 0.f) * drop_apply(d, i);
}

// y = LayerNorm(x + dropout(r)) * g + b; keeps xhat = (z - mean) * rstd and rstd per row
__global__ void __launch_bounds__(256) add_ln_train_kernel(const float* __restrict__ x, const float* __restrict__ r, const XfDrop dr,
                                                            const float* __restrict__ g, const float* __restrict__ b, float* __restrict__ y,
                                                            float* __restrict__ xhat, float* __restrict__ rstd_out, int d, float eps) {
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
    if (c < d) { v[i] = xr[c] + (rr ? rr[c] * drop_apply(dr, (uint64_t)row * d + c) : 0.f); s += v[i]; }
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
  if (threadIdx.x == 0) { stat[1] = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)d + eps); rstd_out[row] = stat[1]; }
  __syncthreads();
  const float rstd = stat[1];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const int c = threadIdx.x + 256 * i;
    if (c < d) {
      const float xh = (v[i] - mean) * rstd;
      xhat[(int64_t)row * d + c] = xh;
      y[(int64_t)row * d + c] = xh * g[c] + b[c];
    }
  }
}

// dz = rstd * (g dy - mean(g dy) - xhat mean(g dy xhat));  dz_drop = dz * dropout mask of the sublayer branch (null: not wanted)
__global__ void __launch_bounds__(256) ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                      const float* __restrict__ g, float* __restrict__ dz, float* __restrict__ dz_drop,
                                                      const XfDrop dr, int d) {
  __shared__ float red[2][4];
  const int row = blockIdx.x;
  float gd[12], xh[12];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const int c = threadIdx.x + 256 * i;
    gd[i] = 0.f; xh[i] = 0.f;
    if (c < d) {
      gd[i] = dy[(int64_t)row * d + c] * g[c];
      xh[i] = xhat[(int64_t)row * d + c];
      s1 += gd[i]; s2 += gd[i] * xh[i];
    }
  }
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  const float m1 = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)d;
  const float m2 = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)d;
  const float rs = rstd[row];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const int c = threadIdx.x + 256 * i;
    if (c < d) {
      const float v = rs * (gd[i] - m1 - xh[i] * m2);
      dz[(int64_t)row * d + c] = v;
      if (dz_drop) dz_drop[(int64_t)row * d + c] = v * drop_apply(dr, (uint64_t)row * d + c);
    }
  }
}

// dgamma[c] = sum_rows dy xhat, dbeta[c] = sum_rows dy: a block owns 64 columns, its 4 waves take the rows m = w mod 4 (ascending)
// and are added in wave order
__global__ void __launch_bounds__(256) ln_bwd_params_kernel(const float* __restrict__ dy, const float* __restrict__ xhat, float* __restrict__ dg,
                                                             float* __restrict__ db, int M, int d) {
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float sg = 0.f, sb = 0.f;
  if (c < d)
    for (int m = wid; m < M; m += 4) { const float v = dy[(int64_t)m * d + c]; sg += v * xhat[(int64_t)m * d + c]; sb += v; }
  red[0][wid][lane] = sg; red[1][wid][lane] = sb;
  __syncthreads();
  if (wid == 0 && c < d) {
    dg[c] = ((red[0][0][lane] + red[0][1][lane]) + red[0][2][lane]) + red[0][3][lane];
    db[c] = ((red[1][0][lane] + red[1][1][lane]) + red[1][2][lane]) + red[1][3][lane];
  }
}

// emb rows (b,t) batch-first, d_img wide -> y (t,b,d) = dropout(v * scale + pe); text channels as in xf_embed_post
__global__ void embed_post_train_kernel(const float* __restrict__ emb, const float* __restrict__ pe, const int32_t* __restrict__ pe_row,
                                        const float* __restrict__ text, int d_txt, float* __restrict__ y, int B, int T, int d, float scale,
                                        const XfDrop dr) {
  const int d_img = d - d_txt;
  const int64_t total = (int64_t)B * T * d;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % d);
    const int t = (int)((idx / d) % T);
    const int b = (int)(idx / ((int64_t)d * T));
    const int pr = pe_row ? pe_row[b] : b;
    const float v = (c < d_img) ? emb[((int64_t)b * T + t) * d_img + c] : text[(int64_t)b * d_txt + (c - d_img)];
    const int64_t o = ((int64_t)t * B + b) * d + c;
    y[o] = (v * scale + pe[(int64_t)pr * d + c]) * drop_apply(dr, (uint64_t)o);
  }
}
// de (b,t) rows, d_img wide = dy (t,b,d)[c < d_img] * mask * scale
__global__ void embed_post_bwd_kernel(const float* __restrict__ dy, float* __restrict__ de, int B, int T, int d, int d_img, float scale,
                                      const XfDrop dr) {
  const int64_t total = (int64_t)B * T * d_img;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % d_img);
    const int t = (int)((idx / d_img) % T);
    const int b = (int)(idx / ((int64_t)d_img * T));
    const int64_t o = ((int64_t)t * B + b) * d + c;
    de[idx] = dy[o] * drop_apply(dr, (uint64_t)o) * scale;
  }
}

// ---- attention, T <= 32: one workgroup per (batch row, head) -----------------------------------------------------------------
// The head's q / k / v (and dO) rows are staged into LDS first with independent coalesced loads: read straight from global inside
// the T x T loops every access is a dependent L2 round trip (36 / 49 us per launch for 11 x 11 tokens).
constexpr int TMAX = 32;
// P (B,H,Tq,Tk) = softmax(q k^T / sqrt(hd) + mask) is kept; o = dropout(P) v
__global__ void __launch_bounds__(256) attn_train_fwd_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                              const float* __restrict__ v, int ldk, const float* __restrict__ mask,
                                                              float* __restrict__ o, float* __restrict__ P, int Tq, int Tk, int B, int heads,
                                                              int hd, const XfDrop dr) {
  extern __shared__ float lds[];                  // sq[Tq][hd], sk[Tk][hd], sv[Tk][hd]
  __shared__ float sc[TMAX][TMAX + 1];
  float* sq = lds;
  float* sk = sq + Tq * hd;
  float* sv = sk + Tk * hd;
  const int b = blockIdx.x, hh = blockIdx.y;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const float scale = rsqrtf((float)hd);
  const int d = heads * hd;
  for (int idx = threadIdx.x; idx < Tq * hd; idx += 256) { const int i = idx / hd, c = idx - i * hd; sq[idx] = q[((int64_t)i * B + b) * ldq + hh * hd + c]; }
  for (int idx = threadIdx.x; idx < Tk * hd; idx += 256) {
    const int j = idx / hd, c = idx - j * hd;
    sk[idx] = k[((int64_t)j * B + b) * ldk + hh * hd + c];
    sv[idx] = v[((int64_t)j * B + b) * ldk + hh * hd + c];
  }
  __syncthreads();
  for (int p = wid; p < Tq * Tk; p += 4) {
    const int i = p / Tk, j = p - i * Tk;
    float s = 0.f;
    for (int c = lane; c < hd; c += 64) s += sq[i * hd + c] * sk[j * hd + c];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) sc[i][j] = s * scale + (mask ? mask[i * Tk + j] : 0.f);
  }
  __syncthreads();
  const int64_t pbase = ((int64_t)b * heads + hh) * Tq * Tk;
  if (threadIdx.x < Tq) {
    const int i = threadIdx.x;
    float mx = -INFINITY;
    for (int j = 0; j < Tk; ++j) mx = fmaxf(mx, sc[i][j]);
    float sum = 0.f;
    for (int j = 0; j < Tk; ++j) { const float e = expf(sc[i][j] - mx); sc[i][j] = e; sum += e; }
    const float inv = 1.f / sum;
    for (int j = 0; j < Tk; ++j) {
      const float pr = sc[i][j] * inv;
      P[pbase + i * Tk + j] = pr;
      sc[i][j] = pr * drop_apply(dr, (uint64_t)(pbase + i * Tk + j));
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < Tq * hd; idx += 256) {
    const int i = idx / hd, c = idx - i * hd;
    float acc = 0.f;
    for (int j = 0; j < Tk; ++j) acc += sc[i][j] * sv[j * hd + c];
    o[((int64_t)i * B + b) * d + hh * hd + c] = acc;
  }
}

// dq, dk, dv of one (batch row, head) from do, q, k, v and the kept P
__global__ void __launch_bounds__(256) attn_train_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ q, int ldq,
                                                              const float* __restrict__ k, const float* __restrict__ v, int ldk,
                                                              const float* __restrict__ P, float* __restrict__ dq, int lddq,
                                                              float* __restrict__ dk, float* __restrict__ dv, int lddk, int Tq, int Tk, int B,
                                                              int heads, int hd, const XfDrop dr) {
  extern __shared__ float lds[];                  // sdo[Tq][hd], sq[Tq][hd], sk[Tk][hd], sv[Tk][hd]
  __shared__ float pd[TMAX][TMAX + 1];            // dropout(P)
  __shared__ float ds[TMAX][TMAX + 1];            // dP, then dS
  float* sdo = lds;
  float* sq = sdo + Tq * hd;
  float* sk = sq + Tq * hd;
  float* sv = sk + Tk * hd;
  const int b = blockIdx.x, hh = blockIdx.y;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const float scale = rsqrtf((float)hd);
  const int d = heads * hd;
  const int64_t pbase = ((int64_t)b * heads + hh) * Tq * Tk;
  for (int idx = threadIdx.x; idx < Tq * hd; idx += 256) {
    const int i = idx / hd, c = idx - i * hd;
    sdo[idx] = dout[((int64_t)i * B + b) * d + hh * hd + c];
    sq[idx] = q[((int64_t)i * B + b) * ldq + hh * hd + c];
  }
  for (int idx = threadIdx.x; idx < Tk * hd; idx += 256) {
    const int j = idx / hd, c = idx - j * hd;
    sk[idx] = k[((int64_t)j * B + b) * ldk + hh * hd + c];
    sv[idx] = v[((int64_t)j * B + b) * ldk + hh * hd + c];
  }
  __syncthreads();
  // dPd[i][j] = <do_i, v_j>; dP = dPd * mask
  for (int p = wid; p < Tq * Tk; p += 4) {
    const int i = p / Tk, j = p - i * Tk;
    float s = 0.f;
    for (int c = lane; c < hd; c += 64) s += sdo[i * hd + c] * sv[j * hd + c];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) {
      const float m = drop_apply(dr, (uint64_t)(pbase + p));
      ds[i][j] = s * m;
      pd[i][j] = P[pbase + p] * m;
    }
  }
  __syncthreads();
  if (threadIdx.x < Tq) {
    const int i = threadIdx.x;
    float dot = 0.f;
    for (int j = 0; j < Tk; ++j) dot += ds[i][j] * P[pbase + i * Tk + j];
    for (int j = 0; j < Tk; ++j) ds[i][j] = P[pbase + i * Tk + j] * (ds[i][j] - dot) * scale;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < Tq * hd; idx += 256) {
    const int i = idx / hd, c = idx - i * hd;
    float acc = 0.f;
    for (int j = 0; j < Tk; ++j) acc += ds[i][j] * sk[j * hd + c];
    dq[((int64_t)i * B + b) * lddq + hh * hd + c] = acc;
  }
  for (int idx = threadIdx.x; idx < Tk * hd; idx += 256) {
    const int j = idx / hd, c = idx - j * hd;
    float ak = 0.f, av = 0.f;
    for (int i = 0; i < Tq; ++i) {
      ak += ds[i][j] * sq[i * hd + c];
      av += pd[i][j] * sdo[i * hd + c];
    }
    dk[((int64_t)j * B + b) * lddk + hh * hd + c] = ak;
    dv[((int64_t)j * B + b) * lddk + hh * hd + c] = av;
  }
}

// ---- criterion (trainers/trainer.py:65-109) -------------------------------------------------------------------------------
// pred (Tt,B,D) rows t >= t0 against expected (B,Tt,D); dpred (Tt,B,D) (rows < t0: zero).  One workgroup per (t,b) row writes
// dpred and its partial sums part[row][3] = {sum (x-y)^2, sum |x-y|, sum gdl terms}; the contrastive part is its own kernel.
__global__ void __launch_bounds__(256) loss_rows_kernel(const float* __restrict__ pred, const float* __restrict__ expected, float* __restrict__ dpred,
                                                         float* __restrict__ part, int Tt, int B, int D, int t0, int fh, int fw, float w_mse,
                                                         float w_l1, float w_gdl, float alpha) {
  __shared__ float red[3][4];
  const int row = blockIdx.x;                     // t * B + b
  const int t = row / B, b = row - t * B;
  float* dp = dpred + (int64_t)row * D;
  if (t < t0) {
    for (int c = threadIdx.x; c < D; c += 256) dp[c] = 0.f;
    if (threadIdx.x < 3) part[row * 3 + threadIdx.x] = 0.f;
    return;
  }
  const float* x = pred + (int64_t)row * D;
  const float* y = expected + ((int64_t)b * Tt + t) * D;
  const float inv_n = 1.f / ((float)(Tt - t0) * (float)B * (float)D);
  const int hw = fh * fw;
  float s_mse = 0.f, s_l1 = 0.f, s_gdl = 0.f;
  // d/dgx of |  |gx| - |gy|  |^alpha
  auto gterm = [&](float gx, float gy, float& val) {
    const float u = fabsf(gx) - fabsf(gy);
    const float au = fabsf(u);
    val = (alpha == 1.f) ? au : (alpha == 2.f ? u * u : powf(au, alpha));
    const float sgn_u = (float)((u > 0.f) - (u < 0.f)), sgn_g = (float)((gx > 0.f) - (gx < 0.f));
    const float mag = (alpha == 1.f) ? 1.f : (alpha == 2.f ? 2.f * au : alpha * powf(au, alpha - 1.f));
    return mag * sgn_u * sgn_g;
  };
  for (int c = threadIdx.x; c < D; c += 256) {
    const float df = x[c] - y[c];
    s_mse += df * df;
    s_l1 += fabsf(df);
    float g = w_mse * 2.f * df * inv_n + w_l1 * (float)((df > 0.f) - (df < 0.f)) * inv_n;
    if (w_gdl != 0.f) {
      const int p = c % hw, r = p / fw, cc = p - r * fw;
      float val, acc = 0.f;
      if (r > 0) acc += gterm(x[c] - x[c - fw], y[c] - y[c - fw], val);                               // this element is the + end
      if (r + 1 < fh) { acc -= gterm(x[c + fw] - x[c], y[c + fw] - y[c], val); s_gdl += val; }        // the - end: count the term once
      if (cc > 0) acc += gterm(x[c] - x[c - 1], y[c] - y[c - 1], val);
      if (cc + 1 < fw) { acc -= gterm(x[c + 1] - x[c], y[c + 1] - y[c], val); s_gdl += val; }
      g += w_gdl * acc * inv_n;
    }
    dp[c] = g;
  }
  for (int o = 32; o > 0; o >>= 1) { s_mse += __shfl_xor(s_mse, o); s_l1 += __shfl_xor(s_l1, o); s_gdl += __shfl_xor(s_gdl, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s_mse; red[1][threadIdx.x >> 6] = s_l1; red[2][threadIdx.x >> 6] = s_gdl; }
  __syncthreads();
  if (threadIdx.x < 3) part[row * 3 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

// BiPatchNCE (models/contrastive_loss.py:29-60) of one (t,b) sample: features = the hw positions, C = 4 channels.
// S[i][j] = <gt_i, pred_j> / tau;  loss1 = CE over rows of S (gradient through the diagonal only: pred is detached off it),
// loss2 = CE over rows of S^T (gradient through pred_i of every S[j][i]).  Adds the gradient into dpred, partial sums to part2[row].
__global__ void __launch_bounds__(256) nce_kernel(const float* __restrict__ pred, const float* __restrict__ expected, float* __restrict__ dpred,
                                                   float* __restrict__ part2, int Tt, int B, int D, int t0, int hw, float inv_tau, float w) {
  extern __shared__ float sm[];                    // gt[hw][4], pr[hw][4], red[4]
  float* gt = sm;
  float* pr = sm + 4 * hw;
  float* red = sm + 8 * hw;
  const int row = blockIdx.x;
  const int t = row / B, b = row - t * B;
  if (t < t0) { if (threadIdx.x == 0) part2[row] = 0.f; return; }
  const float* x = pred + (int64_t)row * D;
  const float* y = expected + ((int64_t)b * Tt + t) * D;
  for (int idx = threadIdx.x; idx < 4 * hw; idx += 256) {
    const int c = idx / hw, p = idx - c * hw;      // latent layout (4, h, w)
    gt[p * 4 + c] = y[idx];
    pr[p * 4 + c] = x[idx];
  }
  __syncthreads();
  const float R = (float)(Tt - t0) * (float)B * (float)hw;   // rows of the flattened cross entropy (mean reduction)
  float lsum = 0.f;
  for (int i = threadIdx.x; i < hw; i += 256) {
    const float g0 = gt[i * 4], g1 = gt[i * 4 + 1], g2 = gt[i * 4 + 2], g3 = gt[i * 4 + 3];
    const float p0 = pr[i * 4], p1 = pr[i * 4 + 1], p2 = pr[i * 4 + 2], p3 = pr[i * 4 + 3];
    // direction 1: row i of S
    float mx = -INFINITY;
    for (int j = 0; j < hw; ++j) mx = fmaxf(mx, (g0 * pr[j * 4] + g1 * pr[j * 4 + 1] + g2 * pr[j * 4 + 2] + g3 * pr[j * 4 + 3]) * inv_tau);
    float se = 0.f;
    for (int j = 0; j < hw; ++j) se += expf((g0 * pr[j * 4] + g1 * pr[j * 4 + 1] + g2 * pr[j * 4 + 2] + g3 * pr[j * 4 + 3]) * inv_tau - mx);
    const float sii = (g0 * p0 + g1 * p1 + g2 * p2 + g3 * p3) * inv_tau;
    const float lse1 = mx + logf(se);
    const float sm_ii = expf(sii - lse1);
    float d0 = (sm_ii - 1.f) * g0, d1 = (sm_ii - 1.f) * g1, d2 = (sm_ii - 1.f) * g2, d3 = (sm_ii - 1.f) * g3;
    // direction 2: row i of S^T, S^T[i][j] = <pred_i, gt_j> / tau
    float mx2 = -INFINITY;
    for (int j = 0; j < hw; ++j) mx2 = fmaxf(mx2, (p0 * gt[j * 4] + p1 * gt[j * 4 + 1] + p2 * gt[j * 4 + 2] + p3 * gt[j * 4 + 3]) * inv_tau);
    float se2 = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int j = 0; j < hw; ++j) {
      const float e = expf((p0 * gt[j * 4] + p1 * gt[j * 4 + 1] + p2 * gt[j * 4 + 2] + p3 * gt[j * 4 + 3]) * inv_tau - mx2);
      se2 += e; a0 += e * gt[j * 4]; a1 += e * gt[j * 4 + 1]; a2 += e * gt[j * 4 + 2]; a3 += e * gt[j * 4 + 3];
    }
    const float lse2 = mx2 + logf(se2);
    const float is2 = 1.f / se2;
    d0 += a0 * is2 - g0; d1 += a1 * is2 - g1; d2 += a2 * is2 - g2; d3 += a3 * is2 - g3;
    lsum += (lse1 - sii) + (lse2 - sii);
    const float sc = w * 0.5f * inv_tau / R;
    float* dp = dpred + (int64_t)row * D;
    dp[i] += sc * d0; dp[hw + i] += sc * d1; dp[2 * hw + i] += sc * d2; dp[3 * hw + i] += sc * d3;
  }
  for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) part2[row] = red[0] + red[1] + red[2] + red[3];
}

// losses[5] = {total, mse, l1, gdl, contrastive} from the per-row partial sums (rows ascending)
__global__ void loss_finish_kernel(const float* __restrict__ part, const float* __restrict__ part2, float* __restrict__ losses, int rows, float n_el,
                                   float n_ce_rows, float w_mse, float w_l1, float w_gdl, float w_nce) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float a = 0.f, b = 0.f, c = 0.f, e = 0.f;
  for (int r = 0; r < rows; ++r) { a += part[r * 3]; b += part[r * 3 + 1]; c += part[r * 3 + 2]; if (part2) e += part2[r]; }
  const float mse = a / n_el, l1 = b / n_el, gdl = c / n_el, nce = part2 ? 0.5f * e / n_ce_rows : 0.f;
  losses[1] = mse; losses[2] = l1; losses[3] = gdl; losses[4] = nce;
  losses[0] = w_mse * mse + w_l1 * l1 + w_gdl * gdl + w_nce * nce;
}

// ---- torch.optim.Adam (no weight decay, no amsgrad), all tensors in one launch ------------------------------------------------
// chunk c covers elements [off, off + n) of tensor `ten`
__global__ void __launch_bounds__(256) adam_kernel(const XfAdamTensor* __restrict__ tens, const XfAdamChunk* __restrict__ chunks, float lr, float beta1,
                                                    float beta2, float eps, float bc1, float bc2_sqrt) {
  const XfAdamChunk ch = chunks[blockIdx.x];
  const XfAdamTensor tn = tens[ch.ten];
  const float step_size = lr / bc1;
  for (int64_t i = ch.off + threadIdx.x; i < ch.off + ch.n; i += 256) {
    const float g = tn.g[i];
    const float m = tn.m[i] + (g - tn.m[i]) * (1.f - beta1);              // exp_avg.lerp_(grad, 1 - beta1)
    const float v = tn.v[i] * beta2 + (1.f - beta2) * g * g;              // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    tn.m[i] = m; tn.v[i] = v;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    tn.p[i] = tn.p[i] - step_size * (m / denom);
  }
}

constexpr int kAttnTrainLds = 150 * 1024;   // + 2 x 4.3 KiB of static score tiles: under the 160 KiB of a CU

int grid_for(int64_t n, int per_block) { return (int)std::min<int64_t>(4096, (n + per_block - 1) / per_block); }

}  // namespace

void xf_train_init_device() {
  HIP_OK(hipFuncSetAttribute((const void*)attn_train_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kAttnTrainLds));
  HIP_OK(hipFuncSetAttribute((const void*)attn_train_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kAttnTrainLds));
  // BiPatchNCE keeps 8 * hw + 4 floats per row in LDS: hw = feat_h * feat_w up to 4096 (FRAME_SIZE 512) needs 128 KiB
  HIP_OK(hipFuncSetAttribute((const void*)nce_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (8 * 4096 + 4) * (int)sizeof(float)));
}

void xf_drop_mask(const XfDrop d, float* out, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(drop_mask_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, d, out, n);
  check_launch("drop_mask");
}

void xf_gemm_tn(const float* dY, int ldy, const float* X, int ldx, float* dW, float* db, int M, int N, int K, int accumulate, hipStream_t s) {
  SVG_CHECK(N % 4 == 0 && K % 4 == 0 && ldy % 4 == 0 && ldx % 4 == 0, "xf_gemm_tn: N %d / K %d / strides must be multiples of 4", N, K);
  hipLaunchKernelGGL(xf_gemm_tn_kernel, dim3(cdiv(K, 128), cdiv(N, 128)), dim3(256), 0, s, dY, ldy, X, ldx, dW, db, M, N, K, accumulate);
  check_launch("xf_gemm_tn");
}

int xf_gemm_nn_splits(int N, int K) {
  int z = std::max(1, 512 / cdiv(K, 64));
  while (z > 1 && cdiv(N, z) < 64) z >>= 1;
  return z;
}
int64_t xf_gemm_nn_slab_floats(int M, int N, int K) { return (int64_t)xf_gemm_nn_splits(N, K) * M * K; }

void xf_gemm_nn(const float* dY, int ldy, const float* W, float* slabs, float* out, int M, int N, int K, const float* gate, float gate_scale,
                const float* add, hipStream_t s) {
  SVG_CHECK(N % 4 == 0 && K % 4 == 0 && ldy % 4 == 0, "xf_gemm_nn: N %d / K %d / stride must be multiples of 4", N, K);
  const int Z = xf_gemm_nn_splits(N, K);
  const int chunk = (cdiv(N, Z) + 63) / 64 * 64;
  for (int m0 = 0; m0 < M; m0 += 96) {
    const int mt = std::min(6, cdiv(M - m0, 16));
    dim3 grid(cdiv(K, 64), Z);
    switch (mt) {
      case 1: hipLaunchKernelGGL((xf_gemm_nn_kernel<1>), grid, dim3(256), 0, s, dY, ldy, W, slabs, m0, M, N, K, chunk); break;
      case 2: hipLaunchKernelGGL((xf_gemm_nn_kernel<2>), grid, dim3(256), 0, s, dY, ldy, W, slabs, m0, M, N, K, chunk); break;
      case 3: hipLaunchKernelGGL((xf_gemm_nn_kernel<3>), grid, dim3(256), 0, s, dY, ldy, W, slabs, m0, M, N, K, chunk); break;
      case 4: hipLaunchKernelGGL((xf_gemm_nn_kernel<4>), grid, dim3(256), 0, s, dY, ldy, W, slabs, m0, M, N, K, chunk); break;
      case 5: hipLaunchKernelGGL((xf_gemm_nn_kernel<5>), grid, dim3(256), 0, s, dY, ldy, W, slabs, m0, M, N, K, chunk); break;
      default: hipLaunchKernelGGL((xf_gemm_nn_kernel<6>), grid, dim3(256), 0, s, dY, ldy, W, slabs, m0, M, N, K, chunk); break;
    }
  }
  const int64_t MN = (int64_t)M * K;
  hipLaunchKernelGGL(xf_nn_finish_kernel, dim3(grid_for(MN, 1024)), dim3(256), 0, s, slabs, Z, MN, gate, gate_scale, add, out);
  check_launch("xf_gemm_nn");
}

void xf_relu_drop(const float* h, float* r, int64_t n, const XfDrop d, hipStream_t s) {
  hipLaunchKernelGGL(relu_drop_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, h, r, n, d);
  check_launch("xf_relu_drop");
}

void xf_add_ln_train(const float* x, const float* r, const XfDrop dr, const float* g, const float* b, float* y, float* xhat, float* rstd, int M,
                     int d, float eps, hipStream_t s) {
  SVG_CHECK(d <= 3072, "xf_add_ln_train: d %d > 3072", d);
  hipLaunchKernelGGL(add_ln_train_kernel, dim3(M), dim3(256), 0, s, x, r, dr, g, b, y, xhat, rstd, d, eps);
  check_launch("xf_add_ln_train");
}

void xf_ln_bwd(const float* dy, const float* xhat, const float* rstd, const float* g, float* dz, float* dz_drop, const XfDrop dr, float* dgamma,
               float* dbeta, int M, int d, hipStream_t s) {
  SVG_CHECK(d <= 3072, "xf_ln_bwd: d %d > 3072", d);
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(M), dim3(256), 0, s, dy, xhat, rstd, g, dz, dz_drop, dr, d);
  hipLaunchKernelGGL(ln_bwd_params_kernel, dim3(cdiv(d, 64)), dim3(256), 0, s, dy, xhat, dgamma, dbeta, M, d);
  check_launch("xf_ln_bwd");
}

void xf_embed_post_train(const float* emb, const float* pe, const int32_t* pe_row, const float* text, int d_txt, float* y, int B, int T, int d,
                         float scale, const XfDrop dr, hipStream_t s) {
  hipLaunchKernelGGL(embed_post_train_kernel, dim3(grid_for((int64_t)B * T * d, 256)), dim3(256), 0, s, emb, pe, pe_row, text, d_txt, y, B, T, d,
                     scale, dr);
  check_launch("xf_embed_post_train");
}
void xf_embed_post_bwd(const float* dy, float* de, int B, int T, int d, int d_img, float scale, const XfDrop dr, hipStream_t s) {
  hipLaunchKernelGGL(embed_post_bwd_kernel, dim3(grid_for((int64_t)B * T * d_img, 256)), dim3(256), 0, s, dy, de, B, T, d, d_img, scale, dr);
  check_launch("xf_embed_post_bwd");
}

void xf_attention_train(const float* q, int ldq, const float* k, const float* v, int ldk, const float* mask, float* o, float* P, int Tq, int Tk,
                        int B, int heads, int hd, const XfDrop dr, hipStream_t s) {
  SVG_CHECK(Tq <= TMAX && Tk <= TMAX, "xf_attention_train: T %d/%d > %d", Tq, Tk, TMAX);
  const size_t lds = (size_t)(Tq + 2 * Tk) * hd * sizeof(float);
  SVG_CHECK(lds <= kAttnTrainLds, "xf_attention_train: %d + 2 x %d tokens of head dim %d do not fit the LDS staging (%zu > %d bytes)", Tq, Tk, hd, lds, kAttnTrainLds);
  hipLaunchKernelGGL(attn_train_fwd_kernel, dim3(B, heads), dim3(256), lds, s, q, ldq, k, v, ldk, mask, o, P, Tq, Tk, B, heads, hd, dr);
  check_launch("xf_attention_train");
}
void xf_attention_bwd(const float* dout, const float* q, int ldq, const float* k, const float* v, int ldk, const float* P, float* dq, int lddq,
                      float* dk, float* dv, int lddk, int Tq, int Tk, int B, int heads, int hd, const XfDrop dr, hipStream_t s) {
  SVG_CHECK(Tq <= TMAX && Tk <= TMAX, "xf_attention_bwd: T %d/%d > %d", Tq, Tk, TMAX);
  const size_t lds = (size_t)(2 * Tq + 2 * Tk) * hd * sizeof(float);
  SVG_CHECK(lds <= kAttnTrainLds, "xf_attention_bwd: 2 x %d + 2 x %d tokens of head dim %d do not fit the LDS staging (%zu > %d bytes)", Tq, Tk, hd, lds, kAttnTrainLds);
  hipLaunchKernelGGL(attn_train_bwd_kernel, dim3(B, heads), dim3(256), lds, s, dout, q, ldq, k, v, ldk, P, dq, lddq, dk, dv, lddk, Tq, Tk, B, heads,
                     hd, dr);
  check_launch("xf_attention_bwd");
}

void xf_criterion(const float* pred, const float* expected, float* dpred, float* part, float* part2, float* losses, int Tt, int B, int D, int t0,
                  int fh, int fw, float w_mse, float w_l1, float w_gdl, float alpha, float w_nce, float temperature, hipStream_t s) {
  SVG_CHECK(D == 4 * fh * fw, "criterion: D_lat %d is not 4 x %d x %d", D, fh, fw);
  const int rows = Tt * B;
  hipLaunchKernelGGL(loss_rows_kernel, dim3(rows), dim3(256), 0, s, pred, expected, dpred, part, Tt, B, D, t0, fh, fw, w_mse, w_l1, w_gdl, alpha);
  const int hw = fh * fw;
  if (w_nce != 0.f)
    hipLaunchKernelGGL(nce_kernel, dim3(rows), dim3(256), (8 * hw + 4) * sizeof(float), s, pred, expected, dpred, part2, Tt, B, D, t0, hw,
                       1.f / temperature, w_nce);
  const float n_el = (float)(Tt - t0) * B * D;
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, s, part, w_nce != 0.f ? part2 : nullptr, losses, rows, n_el,
                     (float)(Tt - t0) * B * hw, w_mse, w_l1, w_gdl, w_nce);
  check_launch("xf_criterion");
}

void xf_adam(const XfAdamTensor* tens, const XfAdamChunk* chunks, int n_chunks, float lr, float beta1, float beta2, float eps, int step,
             hipStream_t s) {
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));          // torch computes the corrections in double
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  hipLaunchKernelGGL(adam_kernel, dim3(n_chunks), dim3(256), 0, s, tens, chunks, lr, beta1, beta2, eps, bc1, bc2_sqrt);
  check_launch("xf_adam");
}
