// Cross-attention of a BasicTransformerBlock in ONE kernel for gfx950 (C = 320: 8 heads x 40, the 64 x 64 level of the SD UNet):
//
//     out = x + to_out( softmax( to_q(LayerNorm2(x)) K^T / sqrt(40) ) V ) + b_out              (diffusers CrossAttention)
//
// As three launches (to_q, attention over the 77 context keys, to_out + residual) the M x 320 query and attention-output tensors
// (73 MB each at 28 clips) are written and read back, and each launch is HBM-side work: 61.6 + 67.6 + 87.6 us.  Here a workgroup
// owns 128 rows of one sample (8 waves x 16 rows, the rows in registers as the B operand, like ff_fused.hip) and everything else
// streams through an 8-slab LDS ring (LDS-DMA, counted vmcnt, one barrier per slab):
//   phase 1  q = LN2(x) Wq^T for all heads (15 slabs of Wq): the head dimension is padded 40 -> 48 (three 16-wide tiles per head), the
//            LayerNorm is folded (W' = W diag(gamma), row sums, W beta) and 1/sqrt(d) log2(e) is applied on the accumulators;
//   phase 2  per head: S^T = K_h q_h^T (80 padded keys x 16 rows, K as the A operand), the softmax of a row lives in the four lanes
//            l15 + 16 lq (two shuffles), O_h^T = V_h^T P^T; the f32 accumulators converted to h16 ARE the next B operand (no LDS round
//            trip): K / V^T / W_out are packed with the k order in which a lane's accumulator registers line up (xattn_pack_*);
//   phase 3  out = O W_out^T + b_out + x (18 slabs of W_out over the padded 384-wide contraction).
// K and V^T of the (constant) context are packed once per DDIM loop, per sample: [head][128 rows][64] for either.
// CHAIN form (SVG_XATTN_FUSED=2; correct, not faster than the plain form: the K = 320 projection it absorbs already runs at the HBM
// rate in gemm_ws.hip): the kernel starts one projection earlier, at the self-attention's output —
//   phase 0  x = r + to_out1(a) + b1 (15 slabs of the self-attention's W_out): the block input of the cross-attention is never written;
//            its h16 rounding stays in registers as the B operand of phase 1 (W_q packed in accumulator order for that) and as the
//            residual of the epilogue, and LayerNorm 2's statistics come from those registers.
#include "igemm_epi.h"
#include <cstdlib>

namespace SDNS {

namespace {

constexpr int XC = 320, XH = 8, XD = 40, XDP = 48;
constexpr int XQ = XH * XDP;              // 384: padded q / o width
constexpr int X_SLAB = 16384;             // 128 rows x 128 B
constexpr int X_RING = 8, X_DEPTH = 4;
constexpr int X_KS = XC / 32;
constexpr int X_P0 = 15, X_P1 = 15, X_P2 = 2 * XH, X_P3 = 18;
constexpr int X_OFF_C = X_RING * X_SLAB;  // sq[384], bq[384], bo[320] (f32)
constexpr int X_LDS = X_OFF_C + (2 * XQ + XC) * 4;
constexpr int X_HEAD_ELEMS = 128 * 64;    // one packed K_h or V_h^T slab, in elements

struct XaArgs {
  const h16* X; int ldx;                 // plain form: the block input (pre-LayerNorm rows, also the residual); CHAIN: the self-attention output rows
  const h16* R; int ldr;                 // CHAIN: residual of the self-attention's output projection
  const h16* Wp; const float* bp;        // CHAIN: that projection [320][320] (natural k order) and its bias
  const float* rs; const float* rm;
  const h16* Wq; const float* sq; const float* bq;
  const h16* Kp; const h16* Vp;
  const h16* Wo; const float* bo;
  h16* out; int ldo;
  int M, rows_per_sample, L;
  float scale_log2e;
};

__device__ __forceinline__ h16x8 cat8(h16x4 a, h16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }

template <bool CHAIN>
__global__ void __launch_bounds__(512, 2) xattn_fused_kernel(const XaArgs a) {
  constexpr int Q0 = CHAIN ? X_P0 : 0;      // slabs of phase 0
  constexpr int X_NQ = Q0 + X_P1 + X_P2 + X_P3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wid);
  const int l15 = lane & 15, lq = lane >> 4;
  const int m0 = blockIdx.x * 128;
  const int sample = m0 / a.rows_per_sample;
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  constexpr unsigned INVALID = 0x80000000u;

  const uint64_t pq = (uint64_t)a.Wq, po = (uint64_t)a.Wo, pp = (uint64_t)a.Wp;
  const v4i srdp = {(int)(unsigned)pp, (int)((pp >> 32) & 0xffff), (int)((unsigned)XC * XC * 2u), 0x00020000};
  const uint64_t pk = (uint64_t)(a.Kp + (int64_t)sample * XH * X_HEAD_ELEMS), pv = (uint64_t)(a.Vp + (int64_t)sample * XH * X_HEAD_ELEMS);
  const v4i srdq = {(int)(unsigned)pq, (int)((pq >> 32) & 0xffff), (int)((unsigned)XQ * XC * 2u), 0x00020000};
  const v4i srdo = {(int)(unsigned)po, (int)((po >> 32) & 0xffff), (int)((unsigned)XC * XQ * 2u), 0x00020000};
  const v4i srdk = {(int)(unsigned)pk, (int)((pk >> 32) & 0xffff), (int)((unsigned)XH * X_HEAD_ELEMS * 2u), 0x00020000};
  const v4i srdv = {(int)(unsigned)pv, (int)((pv >> 32) & 0xffff), (int)((unsigned)XH * X_HEAD_ELEMS * 2u), 0x00020000};

  // ring slab q: [0, 15) Wq k-slab q / 3, row block q % 3; [15, 31) head (q - 15) / 2: K_h then V_h^T; [31, 49) W_out k-slab, row block
  const int prow = lane >> 3;
  auto dma_slab = [&](int qq) {
    const unsigned dst = lds0 + (qq & (X_RING - 1)) * X_SLAB;
    const int q = qq - Q0;                                 // < 0: phase 0 (W_out of the self-attention: k-slab, row block)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int blk = i * 8 + wave_u;
      const int row = blk * 8 + prow;
      const int ch = (lane & 7) ^ (row & 7);
      if (CHAIN && q < 0) {
        const int ks = qq / 3, rb = qq - 3 * ks;
        const int r = rb * 128 + row;
        dma16(srdp, r < XC ? (unsigned)((r * XC + ch * 8) * 2) : INVALID, ks * 128, dst + blk * 1024);
      } else if (q < X_P1) {
        const int ks = q / 3, rb = q - 3 * ks;
        dma16(srdq, (unsigned)(((rb * 128 + row) * XC + ch * 8) * 2), ks * 128, dst + blk * 1024);
      } else if (q < X_P1 + X_P2) {
        const int j = q - X_P1, h = j >> 1;
        const unsigned voff = (unsigned)((row * 64 + ch * 8) * 2);
        if (j & 1) dma16(srdv, voff, h * X_HEAD_ELEMS * 2, dst + blk * 1024);
        else dma16(srdk, voff, h * X_HEAD_ELEMS * 2, dst + blk * 1024);
      } else {
        const int j = q - X_P1 - X_P2, ks = j / 3, rb = j - 3 * ks;
        const int x = rb * 128 + row;                        // MFMA operand row: tile x >> 4, row x & 15
        // plain form: the output column that operand row stands for — the quads of a tile pair are 8 consecutive columns per lane (16-byte
        // epilogue, as gemm_ws.hip); CHAIN keeps the natural order (its residual hb[] comes from phase 0's accumulators in that order)
        const int r = CHAIN ? x : 32 * (x >> 5) + 8 * ((x & 15) >> 2) + 4 * ((x >> 4) & 1) + (x & 3);
        dma16(srdo, x < XC ? (unsigned)((r * XQ + ch * 8) * 2) : INVALID, ks * 128, dst + blk * 1024);
      }
    }
  };
#pragma unroll
  for (int q = 0; q < X_DEPTH; ++q) dma_slab(q);

  float* const sSq = (float*)(smem + X_OFF_C);
  float* const sBq = sSq + XQ;
  float* const sBo = sBq + XQ;
  for (int i = tid; i < XQ; i += 512) { sSq[i] = a.sq[i]; sBq[i] = a.bq[i]; }
  for (int i = tid; i < XC; i += 512) sBo[i] = a.bo[i];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  const int m = m0 + wave_u * 16 + l15;
  const bool m_ok = m < a.M;
  h16x8 xf[X_KS];
#pragma unroll
  for (int ks = 0; ks < X_KS; ++ks) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (m_ok) v = *(const uint4*)(a.X + (int64_t)m * a.ldx + ks * 32 + lq * 8);
    xf[ks] = *(h16x8*)&v;
  }
  int fsw[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) fsw[kk] = l15 * 128 + (((kk * 4 + lq) ^ (l15 & 7)) << 4);

  int q = 0;
  auto step_begin = [&]() {                                // retire slab q, barrier, refill the slot X_DEPTH ahead
    if (q + X_DEPTH - 1 < X_NQ) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else wait_vm((X_NQ - 1 - q) * 2);
    bar();
    if (q + X_DEPTH < X_NQ) dma_slab(q + X_DEPTH);
  };
  const h16x4 zero4 = {(h16)0.f, (h16)0.f, (h16)0.f, (h16)0.f};

  // ---- phase 0 (CHAIN): x = r + a W_p^T + b_p, kept as h16 in registers (tile t: columns 16 t + 4 lq .. + 3 of row l15)
  h16x4 hb[XC / 16];
  float rs, rm;
  if (CHAIN) {
    f32x4 acc0[XC / 16];
#pragma unroll
    for (int t = 0; t < XC / 16; ++t) acc0[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 5; ++ks)
#pragma unroll
      for (int rb = 0; rb < 3; ++rb) {
        step_begin();
        const char* sl = smem + (q & (X_RING - 1)) * X_SLAB;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int t = 0; t < 8; ++t)
            if (rb * 8 + t < XC / 16) {
              const h16x8 wf = *(const h16x8*)(sl + t * 2048 + fsw[kk]);
              acc0[rb * 8 + t] = MFMA_16x16x32(wf, xf[ks * 2 + kk], acc0[rb * 8 + t]);
            }
        ++q;
      }
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < XC / 16; ++t) {
      const int n = t * 16 + lq * 4;
      f32x4 v = acc0[t] + *(const f32x4*)(a.bp + n);
      if (m_ok) {
        const h16x4 r = *(const h16x4*)(a.R + (int64_t)m * a.ldr + n);
        v[0] += (float)r[0]; v[1] += (float)r[1]; v[2] += (float)r[2]; v[3] += (float)r[3];
      }
      hb[t] = to_h16x4(v);
#pragma unroll
      for (int e = 0; e < 4; ++e) sum += (float)hb[t][e];
    }
    sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.f / XC);
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < XC / 16; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float dlt = (float)hb[t][e] - mean; sq += dlt * dlt; }
    sq += __shfl_xor(sq, 16); sq += __shfl_xor(sq, 32);
    rs = m_ok ? rsqrtf(sq * (1.f / XC) + 1e-5f) : 0.f;
    rm = rs * mean;
  } else if (a.rs) {
    rs = m_ok ? a.rs[m] : 0.f; rm = m_ok ? a.rm[m] : 0.f;
  } else {                                                  // two-pass row statistics over the four lanes that hold the row (eps 1e-5)
    float sum = 0.f;
#pragma unroll
    for (int ks = 0; ks < X_KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += (float)xf[ks][j];
    sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.f / XC);
    float sq = 0.f;
#pragma unroll
    for (int ks = 0; ks < X_KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float dlt = (float)xf[ks][j] - mean; sq += dlt * dlt; }
    sq += __shfl_xor(sq, 16); sq += __shfl_xor(sq, 32);
    rs = m_ok ? rsqrtf(sq * (1.f / XC) + 1e-5f) : 0.f;
    rm = rs * mean;
  }

  // ---- phase 1: q^T tiles (24 x 16 padded columns) ------------------------------------------------------------------------------
  h16x4 qb[3 * XH];
  {
    f32x4 accq[3 * XH];
#pragma unroll
    for (int t = 0; t < 3 * XH; ++t) accq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 5; ++ks)
#pragma unroll
      for (int rb = 0; rb < 3; ++rb) {
        step_begin();
        const char* sl = smem + (q & (X_RING - 1)) * X_SLAB;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          h16x8 wf[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) wf[t] = *(const h16x8*)(sl + t * 2048 + fsw[kk]);
#pragma unroll
          for (int t = 0; t < 8; ++t)
            accq[rb * 8 + t] = MFMA_16x16x32(wf[t], CHAIN ? cat8(hb[2 * (ks * 2 + kk)], hb[2 * (ks * 2 + kk) + 1]) : xf[ks * 2 + kk], accq[rb * 8 + t]);
        }
        ++q;
      }
#pragma unroll
    for (int t = 0; t < 3 * XH; ++t) {
      const int n = t * 16 + lq * 4;
      const f32x4 sv = *(const f32x4*)(sSq + n), bv = *(const f32x4*)(sBq + n);
      qb[t] = to_h16x4((accq[t] * rs - sv * rm + bv) * a.scale_log2e);
    }
  }

  // ---- phase 2: one head at a time ------------------------------------------------------------------------------------------------
  h16x4 ob[3 * XH];
#pragma unroll
  for (int h = 0; h < XH; ++h) {
    step_begin();                                          // K_h: rows = keys, 64 (48 used) permuted head channels
    f32x4 sacc[5];
    {
      const char* sl = smem + (q & (X_RING - 1)) * X_SLAB;
#pragma unroll
      for (int t = 0; t < 5; ++t) sacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const h16x8 bop = cat8(qb[3 * h + 2 * kk], kk == 0 ? qb[3 * h + 1] : zero4);
#pragma unroll
        for (int t = 0; t < 5; ++t) {
          const h16x8 wf = *(const h16x8*)(sl + t * 2048 + fsw[kk]);
          sacc[t] = MFMA_16x16x32(wf, bop, sacc[t]);
        }
      }
      ++q;
    }
    // softmax over the keys of row l15: this lane holds keys 16 t + 4 lq + e, the other three quarters hold the rest
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (16 * t + 4 * lq + e >= a.L) sacc[t][e] = -INFINITY;
        mx = fmaxf(mx, sacc[t][e]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
    h16x4 pb[5];
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      f32x4 p;
#pragma unroll
      for (int e = 0; e < 4; ++e) p[e] = __builtin_amdgcn_exp2f(sacc[t][e] - mx);
      pb[t] = to_h16x4(p);
#pragma unroll
      for (int e = 0; e < 4; ++e) sum += (float)pb[t][e];    // the denominator sums what the numerator multiplies
    }
    sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
    step_begin();                                          // V_h^T: rows (key chunk c, channel), 64 permuted keys per row
    {
      const char* sl = smem + (q & (X_RING - 1)) * X_SLAB;
      f32x4 oacc[3];
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 3; ++s) {                         // 32 keys per step: (chunk 0, k half 0), (chunk 0, k half 1), (chunk 1, k half 0)
        const int c = s >> 1, kk = s & 1;
        const h16x8 bop = cat8(pb[2 * s], s < 2 ? pb[2 * s + 1] : zero4);
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
          const h16x8 wf = *(const h16x8*)(sl + (c * 3 + dt) * 2048 + fsw[kk]);
          oacc[dt] = MFMA_16x16x32(wf, bop, oacc[dt]);
        }
      }
      ++q;
      const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) ob[3 * h + dt] = to_h16x4(oacc[dt] * inv);
    }
  }

  // ---- phase 3: out = O W_out^T + b_out + x ------------------------------------------------------------------------------------------
  f32x4 acco[XC / 16];
#pragma unroll
  for (int t = 0; t < XC / 16; ++t) acco[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 6; ++ks)
#pragma unroll
    for (int rb = 0; rb < 3; ++rb) {
      step_begin();
      const char* sl = smem + (q & (X_RING - 1)) * X_SLAB;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int s = 2 * ks + kk;                          // 32-wide step of the padded contraction: accumulator tiles 2 s, 2 s + 1
        const h16x8 bop = cat8(ob[2 * s], ob[2 * s + 1]);
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (rb * 8 + t < XC / 16) {
            const h16x8 wf = *(const h16x8*)(sl + t * 2048 + fsw[kk]);
            acco[rb * 8 + t] = MFMA_16x16x32(wf, bop, acco[rb * 8 + t]);
          }
      }
      ++q;
    }

  if (m_ok) {
    if constexpr (CHAIN) {
#pragma unroll
      for (int t = 0; t < XC / 16; ++t) {
        const int n = t * 16 + lq * 4;
        f32x4 v = acco[t] + *(const f32x4*)(sBo + n);
        const h16x4 r = hb[t];
        v[0] += (float)r[0]; v[1] += (float)r[1]; v[2] += (float)r[2]; v[3] += (float)r[3];
        *(h16x4*)(a.out + (int64_t)m * a.ldo + n) = to_h16x4(v);
      }
    } else {
      // 8 consecutive columns per lane and tile pair: one 16-byte residual load and one 16-byte store (W_out's rows were dealt accordingly)
#pragma unroll
      for (int p = 0; p < XC / 32; ++p) {
        const int n = 32 * p + 8 * lq;
        f32x4 v0 = acco[2 * p] + *(const f32x4*)(sBo + n), v1 = acco[2 * p + 1] + *(const f32x4*)(sBo + n + 4);
        const h16x8 r = *(const h16x8*)(a.X + (int64_t)m * a.ldx + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v0[e] += (float)r[e]; v1[e] += (float)r[4 + e]; }
        *(h16x8*)(a.out + (int64_t)m * a.ldo + n) = cat8(to_h16x4(v0), to_h16x4(v1));
      }
    }
  }
}

// position p = 32 b + 8 lq + j of a packed row <- logical index 32 b + 16 (j >> 2) + 4 lq + (j & 3): the order in which the accumulator
// registers of a lane (four consecutive indices of tile 2 b, then of tile 2 b + 1) line up as a 16x16x32 B operand
__device__ __forceinline__ int perm32(int p) {
  const int b = p >> 5, r = p & 31, lq = r >> 3, j = r & 7;
  return 32 * b + 16 * (j >> 2) + 4 * lq + (j & 3);
}

// Wq'' [384][320] = to_q rows of head h at 48 h + j (j < 40, zeros above), LayerNorm folded (W diag(gamma)); bq = W beta
// (perm: the contraction index in accumulator order — the CHAIN form's B operand of phase 1 is built from accumulator registers)
__global__ void xattn_pack_q_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta,
                                    h16* __restrict__ Wq, float* __restrict__ bq, int perm) {
  const int r = blockIdx.x;                                 // padded row
  const int h = r / XDP, j = r - h * XDP;
  float acc = 0.f;
  for (int p = threadIdx.x; p < XC; p += blockDim.x) {
    const int k = perm ? perm32(p) : p;
    const float v = j < XD ? w[(int64_t)(h * XD + j) * XC + k] : 0.f;
    Wq[(int64_t)r * XC + p] = (h16)(v * gamma[k]);
    acc += v * beta[k];
  }
  __shared__ float red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) bq[r] = red[0];
}

// Wo'' [320][384]: column p of the packed row <- to_out column of padded index perm32(p) (48 h + dd -> 40 h + dd, zeros for dd >= 40)
__global__ void xattn_pack_o_kernel(const float* __restrict__ w, h16* __restrict__ Wo) {
  const int n = blockIdx.x;
  for (int p = threadIdx.x; p < XQ; p += blockDim.x) {
    const int c = perm32(p), h = c / XDP, dd = c - h * XDP;
    Wo[(int64_t)n * XQ + p] = (h16)(dd < XD ? w[(int64_t)n * XC + h * XD + dd] : 0.f);
  }
}

// per sample and head: Kp [128 keys][64] (channels permuted, zeros past 40 / past L) and Vp [2 x 48 rows + pad][64] (row = key chunk * 48 + channel,
// 64 permuted keys of the chunk per row)
__global__ void xattn_pack_kv_kernel(const h16* __restrict__ K, int ldk, int64_t k_bs, const h16* __restrict__ Vt, int ldv, int64_t v_bs,
                                     h16* __restrict__ Kp, h16* __restrict__ Vp, int L) {
  const int n = blockIdx.y, h = blockIdx.x;
  h16* kp = Kp + ((int64_t)n * XH + h) * X_HEAD_ELEMS;
  h16* vp = Vp + ((int64_t)n * XH + h) * X_HEAD_ELEMS;
  for (int i = threadIdx.x; i < X_HEAD_ELEMS; i += blockDim.x) {
    const int row = i >> 6, p = i & 63, c = perm32(p);
    kp[i] = (row < L && c < XD) ? K[(int64_t)n * k_bs + (int64_t)row * ldk + h * XD + c] : (h16)0.f;
    const int chunk = row / XDP, dd = row - chunk * XDP, key = 64 * chunk + c;
    vp[i] = (chunk < 2 && dd < XD && key < L) ? Vt[(int64_t)n * v_bs + (int64_t)(h * XD + dd) * ldv + key] : (h16)0.f;
  }
}

}  // namespace

void xattn_fused_init_device() {
  HIP_OK(hipFuncSetAttribute((const void*)xattn_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, X_LDS));
  HIP_OK(hipFuncSetAttribute((const void*)xattn_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, X_LDS));
}

// C = 320 with 8 heads of 40, at most 80 context keys, samples of a multiple of 128 rows, enough 128-row tiles for every CU; SVG_XATTN_FUSED
// (cached, svg_env_refresh): 0 keeps the three-launch form, 1 / unset the fused form (block input from memory), 2 the CHAIN form
bool xattn_fused_supported(int C, int heads, int M, int rows_per_sample, int L) {
  if (svg_env_i64("SVG_XATTN_FUSED", 1) == 0) return false;
  return C == XC && heads == XH && L > 0 && L <= 80 && rows_per_sample % 128 == 0 && M % rows_per_sample == 0 && M >= 128 * 192;
}
bool xattn_chain_enabled() {       // measured, same box: 18.46 frames/s chained, 18.49-18.52 plain fused, 18.33-18.35 three launches -> off by default
  return svg_env_i64("SVG_XATTN_FUSED", 1) >= 2;
}

int64_t xattn_kv_pack_elems(int N) { return (int64_t)N * XH * X_HEAD_ELEMS; }

void xattn_pack_q(const float* w, const float* gamma, const float* beta, h16* Wq, float* sq, float* bq, int perm, hipStream_t s) {
  hipLaunchKernelGGL(xattn_pack_q_kernel, dim3(XQ), dim3(256), 0, s, w, gamma, beta, Wq, bq, perm);
  check_launch("xattn_pack_q");
  rowsum_h16(Wq, sq, XQ, XC, s);
}

void xattn_pack_o(const float* w, h16* Wo, hipStream_t s) {
  hipLaunchKernelGGL(xattn_pack_o_kernel, dim3(XC), dim3(128), 0, s, w, Wo);
  check_launch("xattn_pack_o");
}

void xattn_pack_kv(const h16* K, int ldk, int64_t k_bs, const h16* Vt, int ldv, int64_t v_bs, h16* Kp, h16* Vp, int N, int L, hipStream_t s) {
  hipLaunchKernelGGL(xattn_pack_kv_kernel, dim3(XH, N), dim3(256), 0, s, K, ldk, k_bs, Vt, ldv, v_bs, Kp, Vp, L);
  check_launch("xattn_pack_kv");
}

// plain form (Wp == nullptr): out = X + attn2(LN2(X)), LayerNorm statistics from rs / rm or (null) from the rows.
// CHAIN form: x = R + X Wp^T + bp first (X = the self-attention's output rows, Wp [320][320] row-major), Wq in accumulator order.
void xattn_fused(svg_ctx* ctx, const h16* X, int ldx, const h16* R, int ldr, const h16* Wp, const float* bp, const float* rs, const float* rm,
                 const h16* Wq, const float* sq, const float* bq, const h16* Kp, const h16* Vp, const h16* Wo, const float* bo, h16* out, int ldo,
                 int M, int rows_per_sample, int L, hipStream_t s) {
  SVG_CHECK(ldx % 8 == 0 && ldo % 8 == 0 && (int64_t)M * ldx < (1LL << 31) && (!Wp || (R && ldr % 4 == 0)), "xattn_fused: strides / size unsupported");
  if (!SVG_LAUNCHING(ctx)) return;
  char tag[64];
  snprintf(tag, sizeof(tag), "xattn_%s_M%d_C%d_L%d", Wp ? "chain" : "fused", M, XC, L);
  ProfScope ps(ctx, PK_ATTN, s, 2.0 * M * (double)XC * XC * (Wp ? 3 : 2) + 4.0 * M * (double)L * XC, 2.0 * ((double)M * XC * (Wp ? 3 : 2) + (Wp ? 3.0 : 2.0) * XC * XC), tag);
  XaArgs a{X, ldx, R, ldr, Wp, bp, rs, rm, Wq, sq, bq, Kp, Vp, Wo, bo, out, ldo, M, rows_per_sample, L, 1.4426950408889634f / sqrtf((float)XD)};
  if (Wp) hipLaunchKernelGGL(xattn_fused_kernel<true>, dim3(cdiv(M, 128)), dim3(512), X_LDS, s, a);
  else hipLaunchKernelGGL(xattn_fused_kernel<false>, dim3(cdiv(M, 128)), dim3(512), X_LDS, s, a);
  check_launch("xattn_fused");
}

}  // namespace SDNS
