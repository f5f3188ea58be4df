// Kernels of the FVD evaluation (reference: evaluation/pytorch_i3d.py:8-133 Unit3D / MaxPool3dSamePadding / InceptionModule,
// evaluation/fvd_2.py:7-78,109-136 preprocess and Fréchet distance), f32 like the reference, channels-last (N,T,H,W,C) activations.
//   conv3d_kernel     3-D convolution as an implicit GEMM on v_mfma_f32_16x16x4_f32 (exact f32): a workgroup owns 64 output positions x
//                     64 output channels, K = taps x Cin walked in 8-channel chunks through LDS; TF 'SAME' zero padding by bounds
//                     checks; BatchNorm (eval) is folded into weights / bias at load, ReLU in the epilogue; the output goes to a channel
//                     slice of a wider tensor (the Inception concat is never a copy)
//   maxpool3d_kernel  'SAME' max pooling: the zero padding takes part in the max, as F.pad + MaxPool3d does in the reference
//   avgpool / mean    the [2,7,7] average pool in front of the logits, the time mean behind them
//   fvd_*             uint8 video -> I3D input (bilinear resize of the shorter side, centre crop, [-1,1]); mean / covariance in f64;
//                     symmetric eigen-decomposition by parallel cyclic Jacobi (one workgroup, f64) for the matrix square roots
// An evaluation tool, not the hot path: written for exactness and reasonable speed (a few TFLOP/s), not for the roofline.
#include "kernels.h"
#include <algorithm>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Conv3dArgs {
  const float* x; int B, T, H, W, Cin;         // input (B,T,H,W,Cin), Cin as stored (multiple of 4)
  const float* w;                              // [Cout][kt][kh][kw][Cin]
  const float* bias;                           // [Cout]
  float* y; int To, Ho, Wo, Cout, ldc, coff;   // output (B,To,Ho,Wo,ldc), channels coff .. coff + Cout
  int kt, kh, kw, st, sh, sw, pt, ph, pw;      // kernel, stride, front padding
  int relu;
};

namespace {

template <int CK>     // channels per K chunk: 8 (Cin % 8 == 0) or 4 (the 3-channel input padded to 4)
__global__ void __launch_bounds__(256) conv3d_kernel(const Conv3dArgs a) {
  __shared__ float sx[64][CK + 1];             // [position][channel]   (+1: the fragment reads walk rows)
  __shared__ float sw[64][CK + 1];             // [cout][channel]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int64_t P = (int64_t)a.B * a.To * a.Ho * a.Wo;
  const int64_t p0 = (int64_t)blockIdx.x * 64;
  const int n0 = blockIdx.y * 64;
  // loader roles: thread -> (row r = tid / (CK / 2), pair of channels); rows 0..63 of x for tid < 32 CK, the same for w
  constexpr int TPR = CK / 2;                  // threads per row (2 floats each)
  const int lr = (tid % (64 * TPR)) / TPR, lc = (tid % TPR) * 2;
  const bool load_x = tid < 64 * TPR;          // CK = 8: 256 threads load x AND w (two passes); CK = 4: 128 + 128
  // this thread's output position (for the x loads)
  const int64_t p = p0 + lr;
  int b = 0, to = 0, ho = 0, wo = 0;
  const bool p_ok = p < P;
  if (p_ok) {
    int64_t r = p;
    wo = (int)(r % a.Wo); r /= a.Wo;
    ho = (int)(r % a.Ho); r /= a.Ho;
    to = (int)(r % a.To); b = (int)(r / a.To);
  }
  const int t0 = to * a.st - a.pt, h0 = ho * a.sh - a.ph, w0 = wo * a.sw - a.pw;
  const int n_row = n0 + lr;
  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int taps = a.kt * a.kh * a.kw;
  for (int tap = 0; tap < taps; ++tap) {
    const int dt = tap / (a.kh * a.kw), dh = (tap / a.kw) % a.kh, dw = tap % a.kw;
    const int ti = t0 + dt, hi = h0 + dh, wi = w0 + dw;
    const bool in = p_ok && (unsigned)ti < (unsigned)a.T && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
    const float* xr = a.x + ((((int64_t)b * a.T + ti) * a.H + hi) * a.W + wi) * a.Cin + lc;
    const float* wr = a.w + ((int64_t)n_row * taps + tap) * a.Cin + lc;
    for (int c0 = 0; c0 < a.Cin; c0 += CK) {
      float2 vx = make_float2(0.f, 0.f), vw = make_float2(0.f, 0.f);
      if (CK == 8 || load_x) { if (in) vx = *(const float2*)(xr + c0); }
      if (CK == 8 || !load_x) { if (n_row < a.Cout) vw = *(const float2*)(wr + c0); }
      __syncthreads();                           // the previous chunk's fragments have been read
      if (CK == 8 || load_x) { sx[lr][lc] = vx.x; sx[lr][lc + 1] = vx.y; }
      if (CK == 8 || !load_x) { sw[lr][lc] = vw.x; sw[lr][lc + 1] = vw.y; }
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < CK; kk += 4) {
        const float xb = sx[wid * 16 + l15][kk + lq];          // B operand: column j = position l15, k = lq
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float wa = sw[j * 16 + l15][kk + lq];          // A operand: row i = cout l15, k = lq
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, xb, acc[j], 0, 0, 0);
        }
      }
    }
  }
  // D[i = cout][j = position]: lane holds position wid * 16 + l15, couts j * 16 + 4 lq .. + 3
  const int64_t po = p0 + wid * 16 + l15;
  if (po < P) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + j * 16 + lq * 4;
      if (n >= a.Cout) continue;
      f32x4 v = acc[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (n + e < a.Cout) {
          float o = v[e] + (a.bias ? a.bias[n + e] : 0.f);
          if (a.relu) o = fmaxf(o, 0.f);
          a.y[po * a.ldc + a.coff + n + e] = o;
        }
      }
    }
  }
}

// weights [Cout][Cin][kt][kh][kw] (+ BatchNorm) -> [Cout][kt][kh][kw][Cin_pad] scaled, bias = shift
__global__ void pack_conv3d_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   const float* __restrict__ mean, const float* __restrict__ var, const float* __restrict__ cbias,
                                   float* __restrict__ wout, float* __restrict__ bout, int Cout, int Cin, int Cin_pad, int taps, float eps) {
  const int64_t total = (int64_t)Cout * taps * Cin_pad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin_pad);
    const int tap = (int)((i / Cin_pad) % taps);
    const int co = (int)(i / ((int64_t)Cin_pad * taps));
    const float scale = gamma ? gamma[co] * rsqrtf(var[co] + eps) : 1.f;
    wout[i] = ci < Cin ? w[((int64_t)co * Cin + ci) * taps + tap] * scale : 0.f;
    if (ci == 0 && tap == 0) bout[co] = gamma ? beta[co] - mean[co] * scale : (cbias ? cbias[co] : 0.f);
  }
}

__global__ void maxpool3d_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int T, int H, int W, int C, int To, int Ho, int Wo,
                                 int kt, int kh, int kw, int st, int sh, int sw, int pt, int ph, int pw) {
  const int64_t total = (int64_t)B * To * Ho * Wo * (C / 4);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % (C / 4)) * 4;
    int64_t r = i / (C / 4);
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho); r /= Ho;
    const int to = (int)(r % To);
    const int b = (int)(r / To);
    f32x4 m = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int dt = 0; dt < kt; ++dt)
      for (int dh = 0; dh < kh; ++dh)
        for (int dw = 0; dw < kw; ++dw) {
          const int ti = to * st - pt + dt, hi = ho * sh - ph + dh, wi = wo * sw - pw + dw;
          f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};                    // the zero padding takes part in the max (F.pad, then max_pool3d)
          if ((unsigned)ti < (unsigned)T && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W)
            v = *(const f32x4*)(x + ((((int64_t)b * T + ti) * H + hi) * W + wi) * C + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
        }
    *(f32x4*)(y + ((((int64_t)b * To + to) * Ho + ho) * Wo + wo) * C + c) = m;
  }
}

// (B,T,H,W,C) -> (B,To,C) averages over windows [kt, H, W] (the pool covers the whole 7 x 7 plane), stride 1 in time
__global__ void avgpool_thw_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int T, int HW, int C, int kt) {
  const int To = T - kt + 1;
  const int64_t total = (int64_t)B * To * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int to = (int)((i / C) % To);
    const int b = (int)(i / ((int64_t)C * To));
    float s = 0.f;
    for (int dt = 0; dt < kt; ++dt)
      for (int q = 0; q < HW; ++q) s += x[(((int64_t)b * T + to + dt) * HW + q) * C + c];
    y[i] = s / (float)(kt * HW);
  }
}
__global__ void time_mean_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int To, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i % C;
  float s = 0.f;
  for (int t = 0; t < To; ++t) s += x[((int64_t)b * To + t) * C + c];
  y[i] = s / (float)To;
}
// (B,C,T,H,W) f32 -> (B,T,H,W,Cp) zero padded channels
__global__ void ncthw_to_nthwc_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, int T, int H, int W, int Cp) {
  const int64_t total = (int64_t)B * T * H * W * Cp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cp);
    int64_t r = i / Cp;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H); r /= H;
    const int t = (int)(r % T);
    const int b = (int)(r / T);
    y[i] = c < C ? x[((((int64_t)b * C + c) * T + t) * H + h) * W + w] : 0.f;
  }
}
// fvd_2.py:109-136 + :14: (B,T,H,W,3) uint8 -> (B,3,T,res,res) f32: /255, bilinear (align_corners False) resize of the shorter side to
// res (the other to ceil(side * scale)), centre crop, (v - 0.5) * 2
__global__ void fvd_preprocess_kernel(const uint8_t* __restrict__ v, float* __restrict__ out, int B, int T, int H, int W, int res, int Hs, int Ws) {
  const int64_t total = (int64_t)B * 3 * T * res * res;
  const int h_start = (Hs - res) / 2, w_start = (Ws - res) / 2;
  const float sy = (float)H / (float)Hs, sx = (float)W / (float)Ws;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xo = (int)(i % res);
    int64_t r = i / res;
    const int yo = (int)(r % res); r /= res;
    const int t = (int)(r % T); r /= T;
    const int c = (int)(r % 3);
    const int b = (int)(r / 3);
    // F.interpolate(bilinear, align_corners=False): src = (dst + 0.5) * scale - 0.5, clamped at 0
    float fy = ((float)(yo + h_start) + 0.5f) * sy - 0.5f, fx = ((float)(xo + w_start) + 0.5f) * sx - 0.5f;
    fy = fmaxf(fy, 0.f); fx = fmaxf(fx, 0.f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const uint8_t* f = v + (((int64_t)b * T + t) * H) * W * 3 + c;
    const float p00 = f[((int64_t)y0 * W + x0) * 3] / 255.f, p01 = f[((int64_t)y0 * W + x1) * 3] / 255.f;
    const float p10 = f[((int64_t)y1 * W + x0) * 3] / 255.f, p11 = f[((int64_t)y1 * W + x1) * 3] / 255.f;
    const float val = (1.f - ly) * ((1.f - lx) * p00 + lx * p01) + ly * ((1.f - lx) * p10 + lx * p11);
    out[i] = (val - 0.5f) * 2.f;
  }
}

// ---- Fréchet distance (f64) ----------------------------------------------------------------------------------------------------------
__global__ void col_mean_kernel(const float* __restrict__ x, double* __restrict__ mean, int n, int d) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d) return;
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += (double)x[(int64_t)i * d + c];
  mean[c] = s / n;
}
// cov[a][b] = sum_i (x[i][a] - mean[a]) (x[i][b] - mean[b]) / (n - 1)   (fvd_2.py:35-63)
__global__ void cov_kernel(const float* __restrict__ x, const double* __restrict__ mean, double* __restrict__ cov, int n, int d) {
  const int bcol = blockIdx.x * blockDim.x + threadIdx.x, a = blockIdx.y;
  if (bcol >= d) return;
  double s = 0.0;
  const double ma = mean[a], mb = mean[bcol];
  for (int i = 0; i < n; ++i) s += ((double)x[(int64_t)i * d + a] - ma) * ((double)x[(int64_t)i * d + bcol] - mb);
  cov[(int64_t)a * d + bcol] = s / (n - 1);
}
// C = A * B (d x d, f64), optionally with A's columns scaled by s (A diag(s) B)
__global__ void matmul_f64_kernel(const double* __restrict__ A, const double* __restrict__ s, const double* __restrict__ Bm, int transB,
                                  double* __restrict__ C, int d) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= d) return;
  double acc = 0.0;
  for (int k = 0; k < d; ++k) acc += A[(int64_t)i * d + k] * (s ? s[k] : 1.0) * (transB ? Bm[(int64_t)j * d + k] : Bm[(int64_t)k * d + j]);
  C[(int64_t)i * d + j] = acc;
}
// Parallel cyclic Jacobi for a symmetric matrix (one workgroup of 1024 threads, d even): A -> eigenvalues on its diagonal, V (columns =
// eigenvectors).  A sweep is d - 1 rounds of d / 2 disjoint rotations (round-robin tournament pairing); a round computes its
// rotations from the current matrix, then applies them to the rows, then to the columns (two-sided: A <- J^T A J), and to V.
__global__ void __launch_bounds__(1024) jacobi_eig_kernel(double* __restrict__ A, double* __restrict__ V, int d, int sweeps, double* __restrict__ cs) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < d * d; i += nt) V[i] = (i / d == i % d) ? 1.0 : 0.0;
  __syncthreads();
  const int half = d / 2;
  int* pq = (int*)(cs + 2 * half);             // pairs of this round (after the rotation table)
  for (int sw = 0; sw < sweeps; ++sw) {
    for (int round = 0; round < d - 1; ++round) {
      // tournament pairing: player d - 1 fixed, the others rotate
      for (int k = tid; k < half; k += nt) {
        int p = (k == 0) ? d - 1 : (round + k) % (d - 1);
        int q = (round + d - 1 - k) % (d - 1);
        if (p > q) { const int t = p; p = q; q = t; }
        pq[2 * k] = p; pq[2 * k + 1] = q;
        const double apq = A[(int64_t)p * d + q], app = A[(int64_t)p * d + p], aqq = A[(int64_t)q * d + q];
        double c = 1.0, s = 0.0;
        if (fabs(apq) > 1e-300) {
          const double tau = (aqq - app) / (2.0 * apq);
          const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
          c = 1.0 / sqrt(1.0 + t * t); s = t * c;
        }
        cs[2 * k] = c; cs[2 * k + 1] = s;
      }
      __syncthreads();
      // rows: A <- J^T A  (rows p, q of every column)
      for (int i = tid; i < half * d; i += nt) {
        const int k = i / d, col = i % d;
        const int p = pq[2 * k], q = pq[2 * k + 1];
        const double c = cs[2 * k], s = cs[2 * k + 1];
        const double ap = A[(int64_t)p * d + col], aq = A[(int64_t)q * d + col];
        A[(int64_t)p * d + col] = c * ap - s * aq;
        A[(int64_t)q * d + col] = s * ap + c * aq;
      }
      __syncthreads();
      // columns: A <- A J, V <- V J
      for (int i = tid; i < half * d; i += nt) {
        const int k = i / d, row = i % d;
        const int p = pq[2 * k], q = pq[2 * k + 1];
        const double c = cs[2 * k], s = cs[2 * k + 1];
        const double ap = A[(int64_t)row * d + p], aq = A[(int64_t)row * d + q];
        A[(int64_t)row * d + p] = c * ap - s * aq;
        A[(int64_t)row * d + q] = s * ap + c * aq;
        const double vp = V[(int64_t)row * d + p], vq = V[(int64_t)row * d + q];
        V[(int64_t)row * d + p] = c * vp - s * vq;
        V[(int64_t)row * d + q] = s * vp + c * vq;
      }
      __syncthreads();
    }
    // converged?  sum of squares off the diagonal against the diagonal's (every thread computes the same decision)
    __shared__ double red[2][16];
    double off = 0.0, dg = 0.0;
    for (int i = tid; i < d * d; i += nt) {
      const double v = A[i];
      if (i / d == i % d) dg += v * v; else off += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) { off += __shfl_xor(off, o); dg += __shfl_xor(dg, o); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = off; red[1][tid >> 6] = dg; }
    __syncthreads();
    off = 0.0; dg = 0.0;
    for (int w = 0; w < nt / 64; ++w) { off += red[0][w]; dg += red[1][w]; }
    __syncthreads();
    if (off <= 1e-30 * dg) break;
  }
}
// s[k] = f(A[k][k]) with f = the reference's thresholded square root (fvd_2.py:24-26: s < eps ? s : sqrt(s)) on max(lambda, 0)
__global__ void diag_sqrt_kernel(const double* __restrict__ A, double* __restrict__ s, int d, double eps) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= d) return;
  const double lam = fmax(A[(int64_t)k * d + k], 0.0);
  s[k] = lam < eps ? lam : sqrt(lam);
}
// out[0] = trace(S1 + S2) - 2 * sum_k sqrt_k + |m1 - m2|^2
__global__ void fd_finish_kernel(const double* __restrict__ S1, const double* __restrict__ S2, const double* __restrict__ sq, const double* __restrict__ m1,
                                 const double* __restrict__ m2, double* __restrict__ out, int d) {
  if (threadIdx.x || blockIdx.x) return;
  double tr = 0.0, ts = 0.0, mm = 0.0;
  for (int k = 0; k < d; ++k) {
    tr += S1[(int64_t)k * d + k] + S2[(int64_t)k * d + k];
    ts += sq[k];
    mm += (m1[k] - m2[k]) * (m1[k] - m2[k]);
  }
  out[0] = tr - 2.0 * ts + mm;
}

inline dim3 grid_for(int64_t n) { return dim3((unsigned)std::min<int64_t>((n + 255) / 256, 65535)); }

}  // namespace

void conv3d_launch(const Conv3dArgs& a, hipStream_t s) {
  const int64_t P = (int64_t)a.B * a.To * a.Ho * a.Wo;
  dim3 grid((unsigned)((P + 63) / 64), (unsigned)((a.Cout + 63) / 64));
  if (a.Cin % 8 == 0) hipLaunchKernelGGL((conv3d_kernel<8>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((conv3d_kernel<4>), grid, dim3(256), 0, s, a);
  check_launch("conv3d");
}
void pack_conv3d(const float* w, const float* gamma, const float* beta, const float* mean, const float* var, const float* cbias, float* wout,
                 float* bout, int Cout, int Cin, int Cin_pad, int taps, float eps, hipStream_t s) {
  hipLaunchKernelGGL(pack_conv3d_kernel, grid_for((int64_t)Cout * taps * Cin_pad), dim3(256), 0, s, w, gamma, beta, mean, var, cbias, wout, bout, Cout,
                     Cin, Cin_pad, taps, eps);
  check_launch("pack_conv3d");
}
void maxpool3d_same(const float* x, float* y, int B, int T, int H, int W, int C, int To, int Ho, int Wo, const int k[3], const int st[3],
                    const int pf[3], hipStream_t s) {
  hipLaunchKernelGGL(maxpool3d_kernel, grid_for((int64_t)B * To * Ho * Wo * (C / 4)), dim3(256), 0, s, x, y, B, T, H, W, C, To, Ho, Wo, k[0], k[1], k[2],
                     st[0], st[1], st[2], pf[0], pf[1], pf[2]);
  check_launch("maxpool3d");
}
void avgpool_thw(const float* x, float* y, int B, int T, int HW, int C, int kt, hipStream_t s) {
  hipLaunchKernelGGL(avgpool_thw_kernel, grid_for((int64_t)B * (T - kt + 1) * C), dim3(256), 0, s, x, y, B, T, HW, C, kt);
  check_launch("avgpool_thw");
}
void time_mean(const float* x, float* y, int B, int To, int C, hipStream_t s) {
  hipLaunchKernelGGL(time_mean_kernel, grid_for((int64_t)B * C), dim3(256), 0, s, x, y, B, To, C);
  check_launch("time_mean");
}
void ncthw_to_nthwc(const float* x, float* y, int B, int C, int T, int H, int W, int Cp, hipStream_t s) {
  hipLaunchKernelGGL(ncthw_to_nthwc_kernel, grid_for((int64_t)B * T * H * W * Cp), dim3(256), 0, s, x, y, B, C, T, H, W, Cp);
  check_launch("ncthw_to_nthwc");
}
void fvd_preprocess(const uint8_t* v, float* out, int B, int T, int H, int W, int res, hipStream_t s) {
  const double scale = (double)res / std::min(H, W);
  const int Hs = H < W ? res : (int)ceil(H * scale), Ws = H < W ? (int)ceil(W * scale) : res;
  hipLaunchKernelGGL(fvd_preprocess_kernel, grid_for((int64_t)B * 3 * T * res * res), dim3(256), 0, s, v, out, B, T, H, W, res, Hs, Ws);
  check_launch("fvd_preprocess");
}

// Fréchet distance of two embedding sets (n1,d), (n2,d) f32 -> out[0] f64.  ws: 6 d^2 + 8 d doubles of device scratch.
void frechet_distance(const float* x1, int n1, const float* x2, int n2, int d, double* ws, double* out, hipStream_t s) {
  double* m1 = ws; double* m2 = m1 + d; double* sq = m2 + d; double* cs = sq + d;       // cs: d doubles of rotations + d ints of pairs
  double* S1 = cs + 2 * d; double* S2 = S1 + (int64_t)d * d; double* Wk = S2 + (int64_t)d * d; double* V = Wk + (int64_t)d * d;
  double* R = V + (int64_t)d * d; double* T2 = R + (int64_t)d * d;
  const dim3 g1((d + 127) / 128), g2((d + 127) / 128, d);
  const int sweeps = getenv("SVG_JACOBI_SWEEPS") ? atoi(getenv("SVG_JACOBI_SWEEPS")) : 40;      // an upper bound: the kernel stops when the off-diagonal mass is below 1e-30 of the diagonal's (10-12 sweeps full rank, ~20 rank-deficient)
  hipLaunchKernelGGL(col_mean_kernel, g1, dim3(128), 0, s, x1, m1, n1, d);
  hipLaunchKernelGGL(col_mean_kernel, g1, dim3(128), 0, s, x2, m2, n2, d);
  hipLaunchKernelGGL(cov_kernel, g2, dim3(128), 0, s, x1, m1, S1, n1, d);
  hipLaunchKernelGGL(cov_kernel, g2, dim3(128), 0, s, x2, m2, S2, n2, d);
  // sqrt(S1) = V f(L) V^T  (fvd_2.py:22-26 takes it from the SVD: for a symmetric PSD matrix the same factors)
  HIP_OK(hipMemcpyAsync(Wk, S1, (size_t)d * d * sizeof(double), hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(jacobi_eig_kernel, dim3(1), dim3(1024), 0, s, Wk, V, d, sweeps, cs);
  hipLaunchKernelGGL(diag_sqrt_kernel, g1, dim3(128), 0, s, Wk, sq, d, 1e-10);
  hipLaunchKernelGGL(matmul_f64_kernel, g2, dim3(128), 0, s, V, sq, V, 1, R, d);          // R = V diag(sq) V^T = sqrt(S1)
  hipLaunchKernelGGL(matmul_f64_kernel, g2, dim3(128), 0, s, R, (const double*)nullptr, S2, 0, T2, d);
  hipLaunchKernelGGL(matmul_f64_kernel, g2, dim3(128), 0, s, T2, (const double*)nullptr, R, 0, Wk, d);   // sqrt(S1) S2 sqrt(S1)
  hipLaunchKernelGGL(jacobi_eig_kernel, dim3(1), dim3(1024), 0, s, Wk, V, d, sweeps, cs);
  hipLaunchKernelGGL(diag_sqrt_kernel, g1, dim3(128), 0, s, Wk, sq, d, 1e-10);           // trace of the square root = sum of sqrt(eigenvalues)
  hipLaunchKernelGGL(fd_finish_kernel, dim3(1), dim3(1), 0, s, S1, S2, sq, m1, m2, out, d);
  check_launch("frechet_distance");
}
