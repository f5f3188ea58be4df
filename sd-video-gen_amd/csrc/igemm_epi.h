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

// ---- helpers of the LDS-DMA pipelines (conv_halo.hip, gemm_pp.hip) ----------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));

// LDS-direct 16-byte buffer load issued from inline asm: hipcc's waitcnt pass does not see it, so it cannot add its own
// conservative vmcnt(0) in front of the fragment reads (it does for the builtin form here: the stage index is dynamic) —
// every wait for these DMAs is one of the explicit counted s_waitcnt below.  M0 (LDS base of the wave's 1-KiB piece) is
// written in the same statement that uses it and restored afterwards.
__device__ __forceinline__ void dma16(v4i srd, unsigned voff, int soff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(srd), "s"(soff), "s"(lds_addr)
               : "memory");
}

// LDS rows are 128 B; 16-byte chunk c of row r sits at chunk c ^ (r & 7).  For a ds_read_b128 lane group (rows l = 0-3
// and 12-15 at chunk c0, rows 4-11 at chunk c0+1) over ANY 16 consecutive rows this is conflict free: rows r and r+8 share
// a key but sit in different chunk classes, and keys of equal parity never differ by exactly 1.  (gemm.hip's (r>>1)&7 key
// needs 16-aligned windows; the tap-shifted patch reads here start anywhere: it measured 25 % conflict cycles.)
__device__ __forceinline__ int lds_off7(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// wave-uniform counted wait (the count has to be an immediate)
__device__ __forceinline__ void wait_vm(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
  }
}
// raw barrier that the compiler may not move LDS accesses across
__device__ __forceinline__ void bar() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

}  // namespace
