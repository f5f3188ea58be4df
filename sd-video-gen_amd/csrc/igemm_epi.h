// Shared device helpers of the igemm kernels (gemm.hip, conv_halo.hip): vector typedefs, the XOR-swizzled LDS
// addressing of 128-byte rows, activations, and the fused epilogue.
#pragma once
#include "kernels.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128;
constexpr int BK = 64;

// exact-erf GELU with erf from Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 output rounding):
// ~12 VALU + exp + rcp per element instead of the ~40-instruction libm erff — the GEGLU epilogue applies it to
// 64 values per thread.
__device__ __forceinline__ float gelu_erf(float x) {
#ifdef GELU_ABL
  return x;
#endif
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = 1.f - poly * __expf(-z * z);
  return 0.5f * x * (1.f + copysignf(e, x));
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }

__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// bias / per-sample bias / residual / activation for 4 consecutive columns n..n+3 of row m
__device__ __forceinline__ f32x4 epi_value(const GemmArgs& g, int z, int m, int n, f32x4 v) {
  if (g.bias) {
    if (g.bias_row) {
      float b = g.bias[m];
      v += b;
    } else {
      f32x4 b = *(const f32x4*)(g.bias + n);
      v += b;
    }
  }
  if (g.bias_bn) {
    f32x4 b = *(const f32x4*)(g.bias_bn + (int64_t)(m / g.rows_per_batch) * (g.bias_bn_ld ? g.bias_bn_ld : g.N) + n);
    v += b;
  }
  if (g.residual) {
    bf16x4 r = *(const bf16x4*)(g.residual + (int64_t)z * g.sC + (int64_t)m * g.ldr + n);
    v[0] += (float)r[0]; v[1] += (float)r[1]; v[2] += (float)r[2]; v[3] += (float)r[3];
  }
  if (g.act == ACT_SILU) {
    for (int i = 0; i < 4; ++i) v[i] = silu_f(v[i]);
  } else if (g.act == ACT_GELU) {
    for (int i = 0; i < 4; ++i) v[i] = gelu_erf(v[i]);
  }
  return v;
}
__device__ __forceinline__ bf16x4 to_bf16x4(f32x4 v) {
  bf16x4 w;
  w[0] = (bf16)v[0]; w[1] = (bf16)v[1]; w[2] = (bf16)v[2]; w[3] = (bf16)v[3];
  return w;
}
__device__ __forceinline__ void epi_store(const GemmArgs& g, int z, int m, int n, f32x4 v) {
  v = epi_value(g, z, m, n, v);
  int64_t o = (int64_t)z * g.sC + (int64_t)m * g.ldc + n;
  if (g.out_f32) *(f32x4*)((float*)g.C + o) = v;
  else *(bf16x4*)((bf16*)g.C + o) = to_bf16x4(v);
}
__device__ __forceinline__ f32x4 geglu_value(const GemmArgs& g, int nh, f32x4 h, f32x4 gt) {
  if (g.bias) {
    h += *(const f32x4*)(g.bias + nh);
    gt += *(const f32x4*)(g.bias + nh + 16);
  }
  f32x4 v;
  for (int i = 0; i < 4; ++i) v[i] = h[i] * gelu_erf(gt[i]);
  return v;
}

}  // namespace
