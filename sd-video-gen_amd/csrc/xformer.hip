// Latent-sequence Transformer kernels (f32 end to end).
//
// The model is a weight stream: M = clips*T rows against 0.44 G parameters.  f32 MFMA runs at 157 TFLOP/s, so the
// stream is HBM-bound up to M ~ 50 rows (2*M/4 FLOP/B against a ridge of 157e12 / 6.3e12 = 25) and MFMA-bound beyond.
// xf_gemm streams W[N][K] ONCE for any M <= 336 (56 clips x 6 tokens) and feeds v_mfma_f32_16x16x4_f32 (exact f32 fma chain):
//   A operand = W tile (16 output columns n), B operand = X^T (16 rows m), D[i=n][j=m].
//   A K step is 32 floats: lane (l15, lq) loads W[n0 + l15][k0 + 8 lq .. +7] as two float4, so a wave-load covers whole
//   128-byte lines of 16 rows; MFMA (h, j) consumes element j of half h, so k-slot lq of that MFMA is k = k0 + 8 lq + 4 h + j
//   — X is loaded with the identical pattern (from L2: every workgroup reads all of X).
// One workgroup = 8 waves = 16 output columns; the waves interleave over K and are reduced through LDS in a fixed order.
// W loads run one step ahead of the MFMAs (two 2-KiB wave-loads in flight per wave, 32 KiB per workgroup: the bytes in
// flight per CU are what sets a stream's rate).  Optional K split across workgroups (short N): partial sums go to f32
// slabs that a finishing kernel adds in a fixed order (deterministic, unlike atomics with more than two addends).
#include "kernels.h"
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int XF_WAVES = 8;
constexpr int XF_MAXMT = 21;   // M <= 336

template <int MT>
__global__ void __launch_bounds__(512) xf_gemm_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                       const float* __restrict__ bias, const float* __restrict__ residual,
                                                       float* __restrict__ Y, int M, int N, int K, int act_in, int ksplit) {
  constexpr int RB = MT < 4 ? MT : 4;            // m tiles reduced through LDS at a time
  __shared__ f32x4 red[XF_WAVES][RB][64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n0 = blockIdx.x * 16;
  const int kz = blockIdx.y;
  const int l15 = lane & 15, lq = lane >> 4;
  const int n = n0 + l15;
  const bool n_ok = n < N;
  const float* wrow = W + (int64_t)(n_ok ? n : 0) * K + 8 * lq;
  const float* xbase = X + (int64_t)l15 * K + 8 * lq;       // row of m tile t: + t * 16 * K
  f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};

  const int steps = (K + 31) / 32;
  const int stride = XF_WAVES * ksplit;
  int st = kz * XF_WAVES + wid;
  f32x4 w0 = zero, w1 = zero;
  auto load_w = [&](int step, f32x4& a, f32x4& b) {
    const int k = step * 32 + 8 * lq;
    const bool ok = n_ok && step < steps && k < K;          // K % 8 == 0: both halves or none
    a = ok ? *(const f32x4*)(wrow + step * 32) : zero;
    b = ok ? *(const f32x4*)(wrow + step * 32 + 4) : zero;
  };
  load_w(st, w0, w1);
  for (; st < steps; st += stride) {
    f32x4 nw0, nw1;
    load_w(st + stride, nw0, nw1);                           // next step's weights: in flight under this step's MFMAs
    const int k0 = st * 32;
    const bool k_ok = k0 + 8 * lq < K;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const bool ok = k_ok && (t * 16 + l15 < M);
      const float* xr = xbase + (int64_t)t * 16 * K + k0;
      f32x4 x0 = ok ? *(const f32x4*)xr : zero;
      f32x4 x1 = ok ? *(const f32x4*)(xr + 4) : zero;
      if (act_in == 1) {          // ReLU on the input (nn.Transformer's feed-forward)
#pragma unroll
        for (int j = 0; j < 4; ++j) { x0[j] = fmaxf(x0[j], 0.f); x1[j] = fmaxf(x1[j], 0.f); }
      } else if (act_in == 2) {   // quick-GELU x * sigmoid(1.702 x) on the input (CLIP's MLP)
#pragma unroll
        for (int j = 0; j < 4; ++j) { x0[j] = x0[j] / (1.f + __expf(-1.702f * x0[j])); x1[j] = x1[j] / (1.f + __expf(-1.702f * x1[j])); }
      } else if (act_in == 3) {   // exact GELU x * Phi(x) on the input (BERT's intermediate activation)
#pragma unroll
        for (int j = 0; j < 4; ++j) { x0[j] = 0.5f * x0[j] * (1.f + erff(x0[j] * 0.70710678118654752f)); x1[j] = 0.5f * x1[j] * (1.f + erff(x1[j] * 0.70710678118654752f)); }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[j], x0[j], acc[t], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[j], x1[j], acc[t], 0, 0, 0);
    }
    w0 = nw0; w1 = nw1;
  }
  // D layout: col j = lane&15 -> m, row i = (lane>>4)*4 + reg -> n.  Fixed summation order over the waves.
  float* out = Y + (ksplit > 1 ? (int64_t)kz * M * N : 0);
#pragma unroll
  for (int t0 = 0; t0 < MT; t0 += RB) {
    if (t0) __syncthreads();
#pragma unroll
    for (int tt = 0; tt < RB; ++tt)
      if (t0 + tt < MT) red[wid][tt][lane] = acc[t0 + tt];
    __syncthreads();
    for (int idx = tid; idx < RB * 256; idx += 512) {
      const int tt = idx >> 8, rem = idx & 255, ln = rem >> 2, rg = rem & 3;
      if (t0 + tt >= MT) continue;
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < XF_WAVES; ++w) sum += red[w][tt][ln][rg];
      const int m = (t0 + tt) * 16 + (ln & 15);
      const int nn = n0 + 4 * (ln >> 4) + rg;
      if (m < M && nn < N) {
        if (ksplit == 1) {
          if (bias) sum += bias[nn];
          if (residual) sum += residual[(int64_t)m * N + nn];
        }
        out[(int64_t)m * N + nn] = sum;
      }
    }
  }
}

// ---- column-block form: a workgroup = 4 waves = 64 output columns (one 16-column tile per wave) over a K range; the X rows of
// a 32-wide K step are staged ONCE per workgroup in LDS (LDS-direct loads, swizzled 128-byte rows) and shared by the four
// waves — the form above re-reads all of X from L2 in every 16-column workgroup, which bounds it at M >= 48.  W streams
// straight to registers, four steps ahead (8 KiB per wave in flight).  lane (l15, lq) holds k = k0 + 4 lq .. + 3 and
// k0 + 16 + 4 lq .. + 3 of its W row (the two float4 a row's 128-byte line is read with); X uses the same k order: LDS chunks lq
// and lq + 4, conflict free under the row & 7 swizzle.  Always split-K capable: partial sums to slabs, fixed-order finish.
// NST LDS stages: the X rows of step s + NST - 1 and the W fragments of step s + NST - 1 are requested during step s, so NST - 1
// steps of both streams are in flight (vmcnt retires in order: whatever is older than the stage awaited is complete, so the
// W prefetch can only be as deep as the X staging).  Small M (the HBM-bound regime) affords five 8-KiB stages.
template <int MT, int NST>
__global__ void __launch_bounds__(256) xf_gemm2_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                        const float* __restrict__ bias, const float* __restrict__ residual,
                                                        float* __restrict__ Y, int M, int N, int K, int act_in, int ksplit) {
  extern __shared__ __attribute__((aligned(16))) char xs[];   // NST stages x (MT * 16 rows) x 128 B, then a 4-KiB sink
  constexpr int STAGE = MT * 16 * 128;
  constexpr int D = NST - 1;                                // prefetch distance in K steps
  constexpr int NPIECE = MT * 2;                            // 8-row pieces of a stage
  constexpr int NPW = (NPIECE + 3) / 4;                     // DMA instructions per wave and stage (padded with sink writes)
  constexpr int VMC = (D - 1) * (NPW + 2) + 2;              // operations younger than the DMAs of stage s + 1 at the end of step s
  static_assert(VMC <= 63, "vmcnt immediate");
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wid);
  const int l15 = lane & 15, lq = lane >> 4;
  const int n0 = blockIdx.x * 64 + wave_u * 16;
  const int kz = blockIdx.y;
  const int steps_total = (K + 31) / 32;
  const int per = (steps_total + ksplit - 1) / ksplit;
  const int st0 = kz * per, st1 = min(steps_total, st0 + per);
  const int n = n0 + l15;
  const bool n_ok = n < N;
  const float* wrow = W + (int64_t)(n_ok ? n : 0) * K + 4 * lq;
  const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};

  // X staging: a wave instruction fills 8 LDS rows (64 lanes x 16 B); lane (r = lane >> 3, p = lane & 7) fetches chunk p ^ (row & 7)
  const __amdgpu_buffer_rsrc_t srdX = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (unsigned)((int64_t)M * K * 4), 0x00020000);
  constexpr unsigned INVALID = 0x80000000u;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  auto stage_x = [&](int step) {                            // always NPW instructions per wave (the counted wait relies on it)
    const int buf = (step - st0) % NST;
    const int k0 = step * 32;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int piece = i * 4 + wave_u;
      const int row = piece * 8 + (lane >> 3);
      const int ch = (lane & 7) ^ (row & 7);
      const bool ok = step < st1 && piece < NPIECE && row < M && k0 + ch * 4 < K;        // K % 4 == 0
      const unsigned voff = ok ? (unsigned)(((int64_t)row * K + k0 + ch * 4) * 4) : INVALID;
      char* dst = piece < NPIECE ? xs + buf * STAGE + piece * 1024 : xs + NST * STAGE + wave_u * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdX, (lds_ptr_t)dst, 16, voff, 0, 0, 0);
    }
  };
  f32x4 wq[D][2];
  auto load_w = [&](int step, f32x4 (&w)[2]) {              // always two loads (out of range: row 0 / masked to zero below)
    const int k = step * 32 + 4 * lq;
    const bool in = step < st1;
    const f32x4 a = *(const f32x4*)(wrow + (in && k < K ? step * 32 : 0));
    const f32x4 b = *(const f32x4*)(wrow + (in && k + 16 < K ? step * 32 + 16 : 0));
    w[0] = (n_ok && in && k < K) ? a : zero;
    w[1] = (n_ok && in && k + 16 < K) ? b : zero;
  };
  f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = zero;

  if (st0 < st1) {
    // prologue: stages and W fragments of steps st0 .. st0 + D - 1, in the steady-state issue order
#pragma unroll
    // (the compiler barriers pin the issue order DMAs -> W loads of a step: the counted waits below count on it)
    for (int d = 0; d < D; ++d) { stage_x(st0 + d); asm volatile("" ::: "memory"); load_w(st0 + d, wq[d]); asm volatile("" ::: "memory"); }
    if (D > 1) {
      // stage st0 must have landed: younger than its DMAs are load_w(st0) and the (D - 1) later stage / W pairs
      constexpr int C0 = (D - 1) * (NPW + 2) + 2;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C0) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const int xoff0 = l15 * 128 + ((lq ^ (l15 & 7)) << 4), xoff1 = l15 * 128 + (((lq + 4) ^ (l15 & 7)) << 4);
    for (int st = st0; st < st1; st += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int step = st + d;
        if (step < st1) {
          const int buf = (step - st0) % NST;
          stage_x(step + D);                                 // into the stage last read in step - 1 (a barrier ago)
          asm volatile("" ::: "memory");
          const f32x4 w0 = wq[d][0], w1 = wq[d][1];
          load_w(step + D, wq[d]);
          asm volatile("" ::: "memory");
          const char* xb = xs + buf * STAGE;
#pragma unroll
          for (int t = 0; t < MT; ++t) {
            f32x4 x0 = *(const f32x4*)(xb + t * 2048 + xoff0), x1 = *(const f32x4*)(xb + t * 2048 + xoff1);
            if (act_in == 1) {
#pragma unroll
              for (int j = 0; j < 4; ++j) { x0[j] = fmaxf(x0[j], 0.f); x1[j] = fmaxf(x1[j], 0.f); }
            } else if (act_in == 2) {
#pragma unroll
              for (int j = 0; j < 4; ++j) { x0[j] = x0[j] / (1.f + __expf(-1.702f * x0[j])); x1[j] = x1[j] / (1.f + __expf(-1.702f * x1[j])); }
            } else if (act_in == 3) {
#pragma unroll
              for (int j = 0; j < 4; ++j) { x0[j] = 0.5f * x0[j] * (1.f + erff(x0[j] * 0.70710678118654752f)); x1[j] = 0.5f * x1[j] * (1.f + erff(x1[j] * 0.70710678118654752f)); }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[j], x0[j], acc[t], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[j], x1[j], acc[t], 0, 0, 0);
          }
          // stage step + 1 has to have landed (everything older with it), and every wave has to be done reading this stage
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMC) : "memory");
          __syncthreads();
        }
      }
    }
  }
  // D layout: col j = lane&15 -> m, row i = (lane>>4)*4 + reg -> n
  float* out = Y + (ksplit > 1 ? (int64_t)kz * M * N : 0);
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int m = t * 16 + l15;
    const int nn = n0 + 4 * lq;
    if (m < M && nn < N) {
      f32x4 v = acc[t];
      if (ksplit == 1) {
        if (bias) v += *(const f32x4*)(bias + nn);
        if (residual) v += *(const f32x4*)(residual + (int64_t)m * N + nn);
      }
      *(f32x4*)(out + (int64_t)m * N + nn) = v;
    }
  }
}

// Y[m][n] = sum_z slabs[z][m][n] + bias[n], z ascending (deterministic)
__global__ void __launch_bounds__(256) xf_splitk_finish_kernel(const float* __restrict__ slabs, const float* __restrict__ bias,
                                                                const float* __restrict__ residual, float* __restrict__ Y, int64_t MN,
                                                                int N, int ksplit) {
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < MN; i += (int64_t)gridDim.x * 1024) {
    f32x4 v = *(const f32x4*)(slabs + i);
    for (int z = 1; z < ksplit; ++z) v += *(const f32x4*)(slabs + (int64_t)z * MN + i);
    if (bias) v += *(const f32x4*)(bias + (i % N));
    if (residual) v += *(const f32x4*)(residual + i);
    *(f32x4*)(Y + i) = v;
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

// one workgroup per (batch row, head); Tq, Tk <= 32 (the text loop of prediction/predict_text.py conditions on 16 frames + SOS)
// kpad (B,Tk) or null: additive key-padding bias of batch row b (nn.Transformer's *_key_padding_mask: 0 / -inf, or a float bias)
__global__ void __launch_bounds__(256) xf_attention_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                            const float* __restrict__ v, int ldk, const float* __restrict__ mask,
                                                            const float* __restrict__ kpad,
                                                            float* __restrict__ o, int Tq, int Tk, int B, int heads, int hd) {
  __shared__ float sc[32][33];
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
    if (lane == 0) sc[i][j] = s * scale + (mask ? mask[i * Tk + j] : 0.f) + (kpad ? kpad[b * Tk + j] : 0.f);
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

// CLIP text embeddings: y[b][t] = token_embedding[ids[b][t]] + position_embedding[t]
__global__ void xf_embed_tokens_kernel(const int32_t* __restrict__ ids, const float* __restrict__ tok, const float* __restrict__ pos,
                                       float* __restrict__ y, int T, int d, int vocab) {
  const int row = blockIdx.x;                       // b * T + t
  int id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const float* tr = tok + (int64_t)id * d;
  const float* pr = pos + (int64_t)(row % T) * d;
  for (int c = threadIdx.x * 4; c < d; c += blockDim.x * 4)
    *(f32x4*)(y + (int64_t)row * d + c) = *(const f32x4*)(tr + c) + *(const f32x4*)(pr + c);
}

// batch-first causal self-attention for the CLIP text tower: qkv rows (b*T + t) hold [q | k | v] (ld = 3d), head dim <= 64,
// T <= 128.  One workgroup per (b, head): K and V of the head in LDS; a wave owns query rows i = wid, wid+4, ...; lane j
// scores key j (and j + 64), wave-shuffle max / sum, then lane c accumulates channel c.  q is scaled by hd^-1/2 (CLIPAttention).
// causal = 0 with lens (B): bidirectional attention over the first lens[b] keys (BERT's padding mask: MiniLM sentence encoder).
__global__ void __launch_bounds__(256) xf_attention_causal_kernel(const float* __restrict__ qkv, float* __restrict__ o, int T, int heads, int hd,
                                                                   int causal, const int32_t* __restrict__ lens) {
  __shared__ float sk[128][65];
  __shared__ float sv[128][64];
  __shared__ float sp[4][128];
  const int b = blockIdx.x, hh = blockIdx.y;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int d = heads * hd, ld = 3 * d;
  const float* base = qkv + (int64_t)b * T * ld + hh * hd;
  for (int idx = threadIdx.x; idx < T * hd; idx += 256) {
    const int j = idx / hd, c = idx - j * hd;
    sk[j][c] = base[(int64_t)j * ld + d + c];
    sv[j][c] = base[(int64_t)j * ld + 2 * d + c];
  }
  __syncthreads();
  const float scale = rsqrtf((float)hd);
  const int len = lens ? min(max(lens[b], 1), T) : T;
  for (int i0 = wid; i0 < T; i0 += 4) {
    const float* qr = base + (int64_t)i0 * ld;
    const int i = causal ? i0 : len - 1;          // last visible key of query i0
    float s0 = -INFINITY, s1 = -INFINITY;
    if (lane <= i) {
      float a = 0.f;
      for (int c = 0; c < hd; ++c) a += qr[c] * scale * sk[lane][c];
      s0 = a;
    }
    if (lane + 64 <= i) {
      float a = 0.f;
      for (int c = 0; c < hd; ++c) a += qr[c] * scale * sk[lane + 64][c];
      s1 = a;
    }
    float mx = fmaxf(s0, s1);
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    const float e0 = (lane <= i) ? __expf(s0 - mx) : 0.f, e1 = (lane + 64 <= i) ? __expf(s1 - mx) : 0.f;
    float sum = e0 + e1;
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float inv = 1.f / sum;
    sp[wid][lane] = e0 * inv;
    sp[wid][lane + 64] = e1 * inv;
    __builtin_amdgcn_wave_barrier();
    if (lane < hd) {
      float acc = 0.f;
      for (int j = 0; j <= i; ++j) acc += sp[wid][j] * sv[j][lane];
      o[((int64_t)b * T + i0) * d + hh * hd + lane] = acc;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

template <int MT>
static void xf_launch(dim3 grid, hipStream_t s, const float* X, const float* W, const float* bias, const float* residual, float* Y,
                      int M, int N, int K, int act_in, int ksplit) {
  hipLaunchKernelGGL((xf_gemm_kernel<MT>), grid, dim3(512), 0, s, X, W, bias, residual, Y, M, N, K, act_in, ksplit);
}

template <int MT>
static void xf2_launch(dim3 grid, hipStream_t s, const float* X, const float* W, const float* bias, const float* residual, float* Y,
                       int M, int N, int K, int act_in, int ksplit) {
  constexpr int NST = MT <= 4 ? 5 : (MT <= 8 ? 3 : 2);
  constexpr int smem = NST * MT * 16 * 128 + 4096;          // above 64 KB for MT = 16 / 21: opted in by xformer_init_device()
  hipLaunchKernelGGL((xf_gemm2_kernel<MT, NST>), grid, dim3(256), smem, s, X, W, bias, residual, Y, M, N, K, act_in, ksplit);
}

template <int MT>
static void xf2_attr() {
  constexpr int NST = MT <= 4 ? 5 : (MT <= 8 ? 3 : 2);
  constexpr int smem = NST * MT * 16 * 128 + 4096;
  HIP_OK(hipFuncSetAttribute((const void*)xf_gemm2_kernel<MT, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
}
// per-device kernel attributes, set when a context is created on the device (svg_create): nothing but launches happens later,
// in particular inside the stream capture of the training-step graph or on the sampling worker threads
void xformer_init_device() {
  xf2_attr<1>(); xf2_attr<2>(); xf2_attr<3>(); xf2_attr<4>(); xf2_attr<6>(); xf2_attr<8>(); xf2_attr<11>(); xf2_attr<16>(); xf2_attr<21>();
}

// column-block form (X shared through LDS): grid (N / 64, ksplit)
static void xf_gemm_v2(svg_ctx* ctx, const float* X, const float* W, const float* bias, float* Y, int M, int N, int K, int act_in,
                       hipStream_t s, const float* residual) {
  SVG_CHECK(K >= 32 && K % 4 == 0, "xf_gemm (column-block form): K = %d must be a multiple of 4 and at least one 32-wide step (masked W loads read row offset 0)", K);
  const int nb = cdiv(N, 64);
  const int steps = cdiv(K, 32);
  // enough workgroups for two per CU while each keeps >= 8 steps (two rounds of its 4-deep pipeline)
  int ksplit = 1;
  while (nb * ksplit < 512 && steps / (ksplit * 2) >= 8 && ksplit < 16) ksplit *= 2;
  static const int ks_env = getenv("SVG_XF_KSPLIT") ? atoi(getenv("SVG_XF_KSPLIT")) : 0;
  if (ks_env > 0) ksplit = ks_env;
  float* slabs = nullptr;
  if (ksplit > 1) { ctx->arena.push(); slabs = ctx->arena.get<float>((int64_t)ksplit * M * N); ctx->arena.pop(); }
  if (!SVG_LAUNCHING(ctx)) return;
  char tag[64];
  snprintf(tag, sizeof(tag), "v2_M%d_N%d_K%d_ks%d", M, N, K, ksplit);
  ProfScope ps(ctx, PK_XF_GEMM, s, 2.0 * M * (double)N * K, 4.0 * ((double)N * K + (double)M * K + (double)M * N), tag);
  dim3 grid(nb, ksplit);
  float* dst = ksplit > 1 ? slabs : Y;
  const int mt = cdiv(M, 16);
  if (mt <= 1) xf2_launch<1>(grid, s, X, W, bias, residual, dst, M, N, K, act_in, ksplit);
  else if (mt <= 2) xf2_launch<2>(grid, s, X, W, bias, residual, dst, M, N, K, act_in, ksplit);
  else if (mt <= 3) xf2_launch<3>(grid, s, X, W, bias, residual, dst, M, N, K, act_in, ksplit);
  else if (mt <= 4) xf2_launch<4>(grid, s, X, W, bias, residual, dst, M, N, K, act_in, ksplit);
  else if (mt <= 6) xf2_launch<6>(grid, s, X, W, bias, residual, dst, M, N, K, act_in, ksplit);
  else if (mt <= 8) xf2_launch<8>(grid, s, X, W, bias, residual, dst, M, N, K, act_in, ksplit);
  else if (mt <= 11) xf2_launch<11>(grid, s, X, W, bias, residual, dst, M, N, K, act_in, ksplit);
  else if (mt <= 16) xf2_launch<16>(grid, s, X, W, bias, residual, dst, M, N, K, act_in, ksplit);
  else xf2_launch<21>(grid, s, X, W, bias, residual, dst, M, N, K, act_in, ksplit);
  check_launch("xf_gemm2");
  if (ksplit > 1) {
    const int64_t MN = (int64_t)M * N;
    hipLaunchKernelGGL(xf_splitk_finish_kernel, dim3((unsigned)std::min<int64_t>((MN / 4 + 255) / 256, 1024)), dim3(256), 0, s, slabs, bias, residual,
                       Y, MN, N, ksplit);
    check_launch("xf_splitk_finish");
  }
}

void xf_gemm(svg_ctx* ctx, const float* X, const float* W, const float* bias, float* Y, int M, int N, int K, int act_in,
             hipStream_t s, const float* residual) {
  SVG_CHECK(K % 8 == 0 && N % 4 == 0, "xf_gemm: K=%d must be a multiple of 8 and N=%d of 4", K, N);
  SVG_CHECK(M >= 1 && M <= 16 * XF_MAXMT, "xf_gemm: M=%d must be in 1..%d", M, 16 * XF_MAXMT);
  // Two forms (same-box kbench, profiles/README.md): up to 47 rows the 16-column form below (every wave streams its own K slice,
  // X straight from L2: 1.6-3.1 TB/s at 6 rows) is ahead; from 48 rows on X re-reads bound it and the column-block form (X
  // shared through LDS) wins: 71 vs 53 TFLOP/s at 168 x 6144 x 2048.  SVG_XF_V forces one.
  static const int ver = getenv("SVG_XF_V") ? atoi(getenv("SVG_XF_V")) : 0;
  if (ver == 2 || (ver == 0 && M >= 48 && K >= 32)) { xf_gemm_v2(ctx, X, W, bias, Y, M, N, K, act_in, s, residual); return; }
  const int nb = cdiv(N, 16);
  // K split: enough workgroups to put >= 2 on every CU while every wave keeps >= 2 steps of 32 (its load pipeline)
  const int steps = cdiv(K, 32);
  int ksplit = 1;
  while (nb * ksplit < 512 && steps / (XF_WAVES * ksplit * 2) >= 2 && ksplit < 8) ksplit *= 2;
  static const int ks_env = getenv("SVG_XF_KSPLIT") ? atoi(getenv("SVG_XF_KSPLIT")) : 0;
  if (ks_env > 0) ksplit = ks_env;
  float* slabs = nullptr;
  if (ksplit > 1) { ctx->arena.push(); slabs = ctx->arena.get<float>((int64_t)ksplit * M * N); ctx->arena.pop(); }   // stream order protects it
  if (!SVG_LAUNCHING(ctx)) return;
  ProfScope ps(ctx, PK_XF_GEMM, s, 2.0 * M * (double)N * K, 4.0 * ((double)N * K + (double)M * K + (double)M * N));
  dim3 grid(nb, ksplit);
  float* dst = ksplit > 1 ? slabs : Y;
  const int mt = cdiv(M, 16);
  if (mt <= 1) xf_launch<1>(grid, s, X, W, bias, residual, dst, M, N, K, act_in, ksplit);
  else if (mt <= 2) xf_launch<2>(grid, s, X, W, bias, residual, dst, M, N, K, act_in, ksplit);
  else if (mt <= 3) xf_launch<3>(grid, s, X, W, bias, residual, dst, M, N, K, act_in, ksplit);
  else if (mt <= 4) xf_launch<4>(grid, s, X, W, bias, residual, dst, M, N, K, act_in, ksplit);
  else if (mt <= 6) xf_launch<6>(grid, s, X, W, bias, residual, dst, M, N, K, act_in, ksplit);
  else if (mt <= 8) xf_launch<8>(grid, s, X, W, bias, residual, dst, M, N, K, act_in, ksplit);
  else if (mt <= 11) xf_launch<11>(grid, s, X, W, bias, residual, dst, M, N, K, act_in, ksplit);
  else if (mt <= 16) xf_launch<16>(grid, s, X, W, bias, residual, dst, M, N, K, act_in, ksplit);
  else xf_launch<21>(grid, s, X, W, bias, residual, dst, M, N, K, act_in, ksplit);
  check_launch("xf_gemm");
  if (ksplit > 1) {
    const int64_t MN = (int64_t)M * N;
    hipLaunchKernelGGL(xf_splitk_finish_kernel, dim3((unsigned)std::min<int64_t>((MN / 4 + 255) / 256, 1024)), dim3(256), 0, s, slabs, bias, residual,
                       Y, MN, N, ksplit);
    check_launch("xf_splitk_finish");
  }
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
                  int B, int heads, int hd, hipStream_t s, const float* kpad) {
  SVG_CHECK(Tq <= 32 && Tk <= 32, "xf_attention: sequence length %d/%d > 32", Tq, Tk);
  hipLaunchKernelGGL(xf_attention_kernel, dim3(B, heads), dim3(256), 0, s, q, ldq, k, v, ldk, mask, kpad, o, Tq, Tk, B, heads, hd);
  check_launch("xf_attention");
}

void xf_embed_tokens(const int32_t* ids, const float* tok, const float* pos, float* y, int rows, int T, int d, int vocab, hipStream_t s) {
  hipLaunchKernelGGL(xf_embed_tokens_kernel, dim3(rows), dim3(192), 0, s, ids, tok, pos, y, T, d, vocab);
  check_launch("xf_embed_tokens");
}

void xf_attention_causal(const float* qkv, float* o, int B, int T, int heads, int hd, hipStream_t s) {
  SVG_CHECK(T >= 1 && T <= 128 && hd >= 1 && hd <= 64, "xf_attention_causal: T=%d (<= 128) / head dim %d (<= 64) unsupported", T, hd);
  hipLaunchKernelGGL(xf_attention_causal_kernel, dim3(B, heads), dim3(256), 0, s, qkv, o, T, heads, hd, 1, (const int32_t*)nullptr);
  check_launch("xf_attention_causal");
}

void xf_attention_padded(const float* qkv, const int32_t* lens, float* o, int B, int T, int heads, int hd, hipStream_t s) {
  SVG_CHECK(T >= 1 && T <= 128 && hd >= 1 && hd <= 64 && lens, "xf_attention_padded: T=%d (<= 128) / head dim %d (<= 64) unsupported", T, hd);
  hipLaunchKernelGGL(xf_attention_causal_kernel, dim3(B, heads), dim3(256), 0, s, qkv, o, T, heads, hd, 0, lens);
  check_launch("xf_attention_padded");
}

namespace {
// BERT embeddings: y[b][t] = word[ids] + position[t] + token_type[0]   (LayerNorm follows as xf_add_ln)
__global__ void xf_embed_bert_kernel(const int32_t* __restrict__ ids, const float* __restrict__ word, const float* __restrict__ pos,
                                     const float* __restrict__ type0, float* __restrict__ y, int T, int d, int vocab) {
  const int row = blockIdx.x;
  int id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const float* wr = word + (int64_t)id * d;
  const float* pr = pos + (int64_t)(row % T) * d;
  for (int c = threadIdx.x * 4; c < d; c += blockDim.x * 4)
    *(f32x4*)(y + (int64_t)row * d + c) = *(const f32x4*)(wr + c) + *(const f32x4*)(pr + c) + *(const f32x4*)(type0 + c);
}
// sentence-transformers Pooling(mean) + Normalize: out[b] = normalize(sum_{t < len_b} x[b][t] / max(len_b, 1e-9)), one workgroup per row
__global__ void __launch_bounds__(256) xf_mean_pool_norm_kernel(const float* __restrict__ x, const int32_t* __restrict__ lens, float* __restrict__ out,
                                                                 int T, int d) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const int len = min(max(lens[b], 0), T);
  float sq = 0.f;
  for (int c = threadIdx.x; c < d; c += 256) {
    float a = 0.f;
    for (int t = 0; t < len; ++t) a += x[((int64_t)b * T + t) * d + c];
    a /= fmaxf((float)len, 1e-9f);
    out[(int64_t)b * d + c] = a;
    sq += a * a;
  }
  for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
  __syncthreads();
  const float nrm = fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), 1e-12f);      // F.normalize eps
  for (int c = threadIdx.x; c < d; c += 256) out[(int64_t)b * d + c] /= nrm;
}
}  // namespace

void xf_embed_bert(const int32_t* ids, const float* word, const float* pos, const float* type0, float* y, int rows, int T, int d, int vocab,
                   hipStream_t s) {
  hipLaunchKernelGGL(xf_embed_bert_kernel, dim3(rows), dim3(96), 0, s, ids, word, pos, type0, y, T, d, vocab);
  check_launch("xf_embed_bert");
}
void xf_mean_pool_norm(const float* x, const int32_t* lens, float* out, int B, int T, int d, hipStream_t s) {
  hipLaunchKernelGGL(xf_mean_pool_norm_kernel, dim3(B), dim3(256), 0, s, x, lens, out, T, d);
  check_launch("xf_mean_pool_norm");
}
