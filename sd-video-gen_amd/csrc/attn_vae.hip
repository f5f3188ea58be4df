// Fused single-head attention of the VAE mid blocks (d = C = 512, one head, S = HW tokens; diffusers AttentionBlock):
//   O = softmax(Q K^T / sqrt(512)) V   without the S x S score matrix in HBM (it was 2 x 1.88 GB per 28-frame group at 512 x 512).
//
// d = 512 does not fit one wave's registers flash-style (the O^T tile alone is 512 x 32 f32 = 256 registers), so the HEAD DIMENSION
// is split over the four waves of a workgroup (SURVEY §7 risk 8): a workgroup owns 64 queries; wave w owns channels 128 w .. + 127
//   1. partial S^T (32 keys x 64 queries) = K_tile[:, slice] * Q^T[slice, :]        v_mfma_f32_32x32x16, Q^T fragments resident
//   2. the four partial tiles meet in LDS; wave w adds them for ITS 16 queries, runs the online softmax there (f32, exp2 domain:
//      running max / sum per query, rescale factor), and leaves P (16-bit, [query][key]) and the factors in LDS
//   3. O^T[slice, 64 queries] = alpha * O^T + V^T_tile[slice, 32 keys] * P^T          (the factor is per query = per lane column)
// K / V^T tiles of 32 keys are fetched one tile ahead into registers and staged through LDS (K row-major, V^T [channel][key]);
// four barriers per tile.  One workgroup per CU (111 KB of LDS, 1 wave per SIMD: MFMA-dominated at d = 512).
#include "kernels.h"
#include "igemm_epi.h"
#include <cstdlib>

namespace SDNS {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int VA_D = 512, VA_Q = 64, VA_KV = 32, VA_SL = 128;      // head dim, queries per workgroup, keys per tile, channels per wave
constexpr int VA_KROW = VA_D * 2 + 16;                              // bytes per K row in LDS (odd multiple of 16)
constexpr int VA_VROW = VA_KV * 2 + 16;                             // bytes per V^T row (32 keys)
constexpr int VA_PROW = VA_KV * 2 + 16;                             // bytes per P row (one query, 32 keys)
constexpr int VA_OFF_V = VA_KV * VA_KROW;
constexpr int VA_OFF_S = VA_OFF_V + VA_D * VA_VROW;
constexpr int VA_OFF_P = VA_OFF_S + 4 * VA_KV * VA_Q * 4;
constexpr int VA_OFF_A = VA_OFF_P + VA_Q * VA_PROW;
constexpr int VA_LDS = VA_OFF_A + 2 * VA_Q * 4 + 16;                // alpha[64], 1 / l[64], rescale flags[4]

__global__ void __launch_bounds__(256) vae_attn_kernel(const h16* __restrict__ q, const h16* __restrict__ k, int ldqk, int64_t qkb,
                                                        const h16* __restrict__ vt, int ldvt, int64_t vtb, h16* __restrict__ out, int ldo,
                                                        int64_t ob, int S, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;
  char* sV = smem + VA_OFF_V;
  float* sS = (float*)(smem + VA_OFF_S);          // [wave][kv 32][q 64]
  char* sP = smem + VA_OFF_P;                     // [q 64][kv 32] 16-bit
  float* sA = (float*)(smem + VA_OFF_A);          // alpha per query, then 1 / l at the end
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int qtiles = S / VA_Q;
  const int b = blockIdx.x / qtiles, q0 = (blockIdx.x % qtiles) * VA_Q;
  const h16* Q = q + (int64_t)b * qkb;
  const h16* K = k + (int64_t)b * qkb;
  const h16* Vt = vt + (int64_t)b * vtb;
  const float c = scale * 1.4426950408889634f;

  // Q^T fragments of this wave's channel slice (B operand of S^T = K Q^T: lane (query r, half h) holds 8 consecutive channels)
  h16x8 qf[2][8];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      qf[qt][ks] = *(const h16x8*)(Q + (int64_t)(q0 + 32 * qt + r) * ldqk + wid * VA_SL + ks * 16 + h * 8);

  f32x16 O[4][2];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int j = 0; j < 16; ++j) O[dt][qt][j] = 0.f;
  // online-softmax state of the 16 queries this wave finishes: lane (query 16 w + (lane & 15), key group lane >> 4) — every lane of a
  // query's four keeps the same copy
  float m_run = -1e30f, l_run = 0.f;

  // tile staging: 2048 16-byte chunks each for K (32 rows x 64) and V^T (512 rows x 4): 8 per thread
  uint4 rk[8], rv[8];
  auto load_regs = [&](int t) {
    const int kv0 = t * VA_KV;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + 256 * i;
      rk[i] = *(const uint4*)(K + (int64_t)(kv0 + (idx >> 6)) * ldqk + (idx & 63) * 8);
      rv[i] = *(const uint4*)(Vt + (int64_t)(idx >> 2) * ldvt + kv0 + (idx & 3) * 8);
    }
  };
  auto store_lds = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + 256 * i;
      *(uint4*)(sK + (idx >> 6) * VA_KROW + (idx & 63) * 16) = rk[i];
      *(uint4*)(sV + (idx >> 2) * VA_VROW + (idx & 3) * 16) = rv[i];
    }
  };
  const int ntiles = S / VA_KV;
  load_regs(0);
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();                               // every wave is done with the previous tile's K / V^T / P
    store_lds();
    __syncthreads();
    if (t + 1 < ntiles) load_regs(t + 1);
    // ---- 1. partial S^T over this wave's 128 channels
    f32x16 Sp[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int j = 0; j < 16; ++j) Sp[qt][j] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const h16x8 a = *(const h16x8*)(sK + r * VA_KROW + (wid * VA_SL + ks * 16 + h * 8) * 2);
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) Sp[qt] = MFMA_32x32x16(a, qf[qt][ks], Sp[qt]);
    }
    // D[i = key][j = query]: lane holds query column r (+ 32 qt), keys (j & 3) + 8 (j >> 2) + 4 h
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int kv = (j & 3) + 8 * (j >> 2) + 4 * h;
        sS[(wid * VA_KV + kv) * VA_Q + 32 * qt + r] = Sp[qt][j];
      }
    __syncthreads();
    // ---- 2. this wave's 16 queries: sum of the four partials, online softmax, P and the rescale factor
    {
      const int ql = wid * 16 + (lane & 15), kg = lane >> 4;       // query, key group (8 keys)
      float s[8];
      float mx = -1e30f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int kv = kg * 8 + e;
        const float v = sS[(0 * VA_KV + kv) * VA_Q + ql] + sS[(1 * VA_KV + kv) * VA_Q + ql] + sS[(2 * VA_KV + kv) * VA_Q + ql] +
                        sS[(3 * VA_KV + kv) * VA_Q + ql];
        s[e] = v * c;
        mx = fmaxf(mx, s[e]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      float psum = 0.f;
      h16x8 p;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float pe = __builtin_amdgcn_exp2f(s[e] - m_new);
        p[e] = (h16)pe;
        psum += (float)p[e];                       // the sum of what the PV product actually multiplies
      }
      psum += __shfl_xor(psum, 16);
      psum += __shfl_xor(psum, 32);
      l_run = l_run * alpha + psum;
      m_run = m_new;
      *(h16x8*)(sP + ql * VA_PROW + kg * 16) = p;
      if (kg == 0) sA[ql] = alpha;
      // does any of this wave's queries rescale?  (after the first tiles the running maxima rarely move: the O-wide multiply is skipped then)
      const bool any_rs = __any(alpha != 1.f);
      if (lane == 0) sA[2 * VA_Q + wid] = any_rs ? 1.f : 0.f;
    }
    __syncthreads();
    // ---- 3. O^T = alpha O^T + V^T_tile P^T
    {
      const bool rescale = (sA[2 * VA_Q] + sA[2 * VA_Q + 1] + sA[2 * VA_Q + 2] + sA[2 * VA_Q + 3]) != 0.f;      // workgroup-uniform
      if (rescale) {
        const float a0 = sA[r], a1 = sA[32 + r];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int j = 0; j < 16; ++j) { O[dt][0][j] *= a0; O[dt][1][j] *= a1; }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const h16x8 p0 = *(const h16x8*)(sP + r * VA_PROW + (ks * 16 + h * 8) * 2);
        const h16x8 p1 = *(const h16x8*)(sP + (32 + r) * VA_PROW + (ks * 16 + h * 8) * 2);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const h16x8 v = *(const h16x8*)(sV + (wid * VA_SL + dt * 32 + r) * VA_VROW + (ks * 16 + h * 8) * 2);
          O[dt][0] = MFMA_32x32x16(v, p0, O[dt][0]);
          O[dt][1] = MFMA_32x32x16(v, p1, O[dt][1]);
        }
      }
    }
  }
  // ---- epilogue: 1 / l per query from its owner wave, then out[q][channel] = O^T[channel][q] / l
  __syncthreads();
  if ((lane >> 4) == 0) sA[VA_Q + wid * 16 + (lane & 15)] = 1.f / l_run;
  __syncthreads();
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const float inv = sA[VA_Q + 32 * qt + r];
    h16* orow = out + (int64_t)b * ob + (int64_t)(q0 + 32 * qt + r) * ldo + wid * VA_SL;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        h16x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (h16)(O[dt][qt][g4 * 4 + e] * inv);
        *(h16x4*)(orow + dt * 32 + 8 * g4 + 4 * h) = w;             // rows (j & 3) + 8 (j >> 2) + 4 h of the 32-channel tile
      }
  }
}

}  // namespace

bool vae_attention_supported(int S, int C, int ldqk, int ldvt, int ldo) {
  // Default OFF: built, correct (parity tests run it), and 3.3x SLOWER than the three-GEMM path it replaces — 9.5 vs 2.9 ms per
  // 28 x 4096-token attention (profiles/README.md): four barriers per 32-key tile at one wave per SIMD, the cross-wave S reduction
  // through LDS, and 272 bytes of scratch per lane (the Q fragments spill).  It removes 7.5 GB of HBM traffic per frame group, but
  // the time is what the frame pays: SVG_VAE_ATTN_FUSED=1 selects it (cached; the tests toggle it and call svg_env_refresh).
  const int on = (int)svg_env_i64("SVG_VAE_ATTN_FUSED", 0);
  return on && C == VA_D && S % VA_Q == 0 && S >= VA_Q && ldqk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0;
}

void vae_attn_init_device() {
  HIP_OK(hipFuncSetAttribute((const void*)vae_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, VA_LDS));
}

// q, k: (B, S, ·) rows of stride ldqk (batch stride qkb); vt: (B, C, ·) V transposed, row stride ldvt; out (B, S, C) row stride ldo
void vae_attention(svg_ctx* ctx, const h16* q, const h16* k, int ldqk, int64_t qkb, const h16* vt, int ldvt, int64_t vtb, h16* out, int ldo,
                   int64_t ob, int B, int S, int C, hipStream_t s) {
  SVG_CHECK(vae_attention_supported(S, C, ldqk, ldvt, ldo), "vae_attention: S=%d C=%d unsupported", S, C);
  if (!SVG_LAUNCHING(ctx)) return;
  char tag[64];
  snprintf(tag, sizeof(tag), "B%d_h1_Sq%d_Skv%d_d%d_fused", B, S, S, C);
  ProfScope ps(ctx, PK_ATTN, s, 4.0 * B * (double)S * S * C, 2.0 * B * ((double)S * C * 2 * 2), tag);
  hipLaunchKernelGGL(vae_attn_kernel, dim3(B * (S / VA_Q)), dim3(256), VA_LDS, s, q, k, ldqk, qkb, vt, ldvt, vtb, out, ldo, ob, S,
                     1.f / sqrtf((float)C));
  check_launch("vae_attention");
}

}  // namespace SDNS
