// Fused attention for gfx950: softmax(Q K^T * scale) V, flash-style (no S x S matrix in HBM).
//
// Workgroup = 4 waves = 128 query rows of one (sample, head); each wave owns 32 query rows.
// KV is walked in tiles of 64 keys staged through LDS (K row-major, V TRANSPOSED: [d][kv]).
//   S^T (32 kv x 32 q) = K_tile * Q^T      v_mfma_f32_32x32x16_bf16, A = K rows (LDS), B = Q^T (registers)
//     -> each lane holds 16 scores of ONE query column; its partner lane (lane^32) holds the other 16:
//        row max / row sum are in-register reductions plus one cross-half shuffle.
//   O^T (d x 32 q) += V^T_tile * P^T       the f32 S^T accumulator, converted to bf16, IS the B operand
//     (k order inside a step: element j of lane-half h = kv row 16s + 8(j>>2) + 4h + (j&3)); the A
//     operand reads V^T from LDS with that same k order (two 8-byte reads).
// LDS row strides are padded (K: odd multiple of 16 B, V^T: 136 B) so fragment reads are conflict free.
#include "kernels.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int KVT = 64;     // keys per tile
constexpr int VROW = 136;   // bytes per V^T LDS row (64 kv * 2 B + 8)

template <int KS>  // KS = ceil(d / 16): k-steps of the QK^T contraction
__global__ void __launch_bounds__(256) attn_kernel(const AttnArgs a) {
  constexpr int DK = KS * 16;
  constexpr int DVT = (KS + 1) / 2;           // 32-row tiles of O^T
  constexpr int KROW = DK * 2 + 16;           // bytes per K LDS row
  constexpr int K_BYTES = KVT * KROW;
  constexpr int V_BYTES = DVT * 32 * VROW;
  constexpr int NLD = DVT;                    // 16-B chunks per thread per tile, for K and for V^T
  __shared__ __attribute__((aligned(16))) char smem[K_BYTES + V_BYTES];
  char* const sK = smem;
  char* const sV = smem + K_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int b = blockIdx.z, head = blockIdx.y;
  const int q0 = blockIdx.x * 128 + wid * 32;
  const int d = a.d;
  const int CPR = d >> 3;                      // 16-B chunks per K row

  const bf16* __restrict__ Q = a.q + (int64_t)b * a.qb + head * d;
  const bf16* __restrict__ Kp = a.k + (int64_t)b * a.kb + head * d;
  const bf16* __restrict__ Vt = a.vt + (int64_t)b * a.vtb + (int64_t)head * d * a.ldvt;

  // ---- Q^T fragments (B operand): lane (r,h) holds Q[q0+r][16ks + 8h .. +7] -------------------
  bf16x8 qf[KS];
  {
    const int q = q0 + r;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int col = ks * 16 + h * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (q < a.Sq && col < d) v = *(const uint4*)(Q + (int64_t)q * a.ldq + col);
      qf[ks] = *(bf16x8*)&v;
    }
  }

  // ---- staging maps --------------------------------------------------------------------------------
  int k_row[NLD], k_ch[NLD];       // K tile: idx -> (row, chunk); row = -1 when idx is past the tile
  int v_row[NLD], v_ch[NLD];       // V^T tile: idx -> (d row, chunk)
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int idx = tid + 256 * i;
    if (idx < KVT * CPR) { k_row[i] = idx / CPR; k_ch[i] = idx - k_row[i] * CPR; } else { k_row[i] = -1; k_ch[i] = 0; }
    v_row[i] = idx >> 3; v_ch[i] = idx & 7;
    if (v_row[i] >= d) v_row[i] = -1;
  }
  // zero the K pad columns (d .. DK-1) once
  if (CPR * 8 < DK && tid < KVT) *(uint4*)(sK + tid * KROW + CPR * 16) = make_uint4(0, 0, 0, 0);

  uint4 rk[NLD], rv[NLD];
  auto load_regs = [&](int t) {
    const int kv0 = t * KVT;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      rk[i] = make_uint4(0, 0, 0, 0);
      rv[i] = make_uint4(0, 0, 0, 0);
      if (k_row[i] >= 0 && kv0 + k_row[i] < a.Skv)
        rk[i] = *(const uint4*)(Kp + (int64_t)(kv0 + k_row[i]) * a.ldk + k_ch[i] * 8);
      if (v_row[i] >= 0 && kv0 + v_ch[i] * 8 < a.Skv)
        rv[i] = *(const uint4*)(Vt + (int64_t)v_row[i] * a.ldvt + kv0 + v_ch[i] * 8);
    }
  };
  auto write_lds = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      if (k_row[i] >= 0) *(uint4*)(sK + k_row[i] * KROW + k_ch[i] * 16) = rk[i];
      if (v_row[i] >= 0) {
        char* p = sV + v_row[i] * VROW + v_ch[i] * 16;
        *(uint2*)p = make_uint2(rv[i].x, rv[i].y);
        *(uint2*)(p + 8) = make_uint2(rv[i].z, rv[i].w);
      }
    }
  };

  f32x16 O[DVT];
#pragma unroll
  for (int i = 0; i < DVT; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) O[i][j] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float c = a.scale * 1.4426950408889634f;   // exp(x*scale) = exp2(x*c)

  const int ntiles = (a.Skv + KVT - 1) / KVT;
  load_regs(0);
  write_lds();
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const bool more = (t + 1 < ntiles);
    if (more) load_regs(t + 1);

    // ---- S^T = K_tile * Q^T for the two 32-key sub-tiles ------------------------------------------
    f32x16 S0, S1;
#pragma unroll
    for (int j = 0; j < 16; ++j) { S0[j] = 0.f; S1[j] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 a0 = *(const bf16x8*)(sK + r * KROW + (ks * 2 + h) * 16);
      const bf16x8 a1 = *(const bf16x8*)(sK + (32 + r) * KROW + (ks * 2 + h) * 16);
      S0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, qf[ks], S0, 0, 0, 0);
      S1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, qf[ks], S1, 0, 0, 0);
    }
    // ---- mask keys past Skv (last tile only) ----------------------------------------------------
    if (t * KVT + KVT > a.Skv) {
      const int base = t * KVT + 4 * h;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int row = (j & 3) + 8 * (j >> 2);
        if (base + row >= a.Skv) S0[j] = -INFINITY;
        if (base + 32 + row >= a.Skv) S1[j] = -INFINITY;
      }
    }
    // ---- online softmax (per query column; lanes l and l^32 share a column) ----------------------
    float mx = -1e30f;
#pragma unroll
    for (int j = 0; j < 16; ++j) mx = fmaxf(mx, fmaxf(S0[j], S1[j]));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
    const float mc = m_new * c;
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      S0[j] = __builtin_amdgcn_exp2f(fmaf(S0[j], c, -mc));
      S1[j] = __builtin_amdgcn_exp2f(fmaf(S1[j], c, -mc));
      psum += S0[j] + S1[j];
    }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int i = 0; i < DVT; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) O[i][j] *= alpha;

    // ---- O^T += V^T_tile * P^T -------------------------------------------------------------------
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (bf16)(st == 0 ? S0[8 * s2 + j] : S1[8 * s2 + j]);
        const int kvoff = (32 * st + 16 * s2 + 4 * h) * 2;   // bytes
#pragma unroll
        for (int dt = 0; dt < DVT; ++dt) {
          const char* p = sV + (32 * dt + r) * VROW + kvoff;
          uint2 lo = *(const uint2*)p;
          uint2 hi = *(const uint2*)(p + 16);
          uint4 v = make_uint4(lo.x, lo.y, hi.x, hi.y);
          O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8*)&v, pf, O[dt], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (more) write_lds();
    __syncthreads();
  }

  // ---- normalise and store O[q][head*d + dd] -------------------------------------------------------
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.f / l_tot;
  const int q = q0 + r;
  if (q < a.Sq) {
    bf16* orow = a.out + (int64_t)b * a.ob + (int64_t)q * a.ldo + head * d;
#pragma unroll
    for (int dt = 0; dt < DVT; ++dt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = 32 * dt + 8 * g + 4 * h;
        if (dd < d) {
          bf16x4 w;
          w[0] = (bf16)(O[dt][4 * g + 0] * inv);
          w[1] = (bf16)(O[dt][4 * g + 1] * inv);
          w[2] = (bf16)(O[dt][4 * g + 2] * inv);
          w[3] = (bf16)(O[dt][4 * g + 3] * inv);
          *(bf16x4*)(orow + dd) = w;
        }
      }
    }
  }
}

template <int KS>
void launch(const AttnArgs& a, hipStream_t s) {
  dim3 grid(cdiv(a.Sq, 128), a.heads, a.B);
  hipLaunchKernelGGL((attn_kernel<KS>), grid, dim3(256), 0, s, a);
}

}  // namespace

void attention(svg_ctx* ctx, const AttnArgs& a, hipStream_t s) {
  SVG_CHECK(a.d % 8 == 0 && a.d >= 8 && a.d <= 160, "attention: head dim %d unsupported (multiple of 8, <= 160)", a.d);
  SVG_CHECK(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldvt % 8 == 0 && a.ldo % 4 == 0, "attention: strides must be 16-byte aligned");
  SVG_CHECK(a.ldvt >= (a.Skv + 7) / 8 * 8, "attention: V^T rows must be padded to a multiple of 8 keys");
  SVG_CHECK(a.Skv > 0 && a.Sq > 0, "attention: empty");
  if (!SVG_LAUNCHING(ctx)) return;
  ProfScope ps(ctx, PK_ATTN, s, 4.0 * a.B * a.heads * (double)a.Sq * a.Skv * a.d,
               2.0 * a.B * a.heads * ((double)a.Sq * a.d * 2 + (double)a.Skv * a.d * 2));
  const int ks = (a.d + 15) / 16;
  switch (ks) {
    case 1: launch<1>(a, s); break;
    case 2: launch<2>(a, s); break;
    case 3: launch<3>(a, s); break;
    case 4: launch<4>(a, s); break;
    case 5: launch<5>(a, s); break;
    case 6: launch<6>(a, s); break;
    case 8: launch<8>(a, s); break;
    case 10: launch<10>(a, s); break;
    default: throw SvgError("attention: head dim not instantiated");
  }
  check_launch("attention");
}
