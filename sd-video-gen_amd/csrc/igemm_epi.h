// Shared device helpers of the igemm kernels (gemm.hip, conv_halo.hip): vector typedefs, the XOR-swizzled LDS
// addressing of 128-byte rows, activations, and the fused epilogue.
#pragma once
#include "kernels.h"

namespace SDNS {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128;
constexpr int BK = 64;

// exact (erf) GELU without erf: gelu(x) = x Phi(x) = max(x, 0) - |x| Phi(-|x|), and the normal tail is smooth in the log domain:
// log2 Phi(-a) is fitted on [0, 6] by a degree-6 polynomial (weighted for the error of a Phi(-a); |x| > 6 reuses the value at 6,
// where |x| Phi(-|x|) < 1e-8 |x|).  |gelu - exact| <= 2.7e-7 in f32 (Abramowitz-Stegun 7.1.26, the previous form: 4.7e-7), and
// the whole thing is one min, six FMAs (v_pk_fma_f32: two elements per instruction), one v_exp_f32, one max and one FMA —
// 8 issue slots per element against 18 (rcp + exp + 14 VALU).  The GEGLU epilogues run it on 16-64 values per lane: at K = 320
// (ff_fused) that was three quarters of the matrix time.  Coefficients: tools/fit_gelu.py.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 gelu_erf4(f32x4 x) {
#ifdef GELU_ABL
  return x;
#endif
#ifdef GELU_AS      // A/B switch: the Abramowitz-Stegun form of rounds 1-2 (tools/build_variant.sh as "-DGELU_AS")
  f32x4 o;
  for (int i = 0; i < 4; ++i) {
    const float z = fabsf(x[i]) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    o[i] = 0.5f * x[i] * (1.f + copysignf(1.f - poly * __expf(-z * z), x[i]));
  }
  return o;
#endif
  f32x4 a, out;
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = fminf(fabsf(x[i]), 6.f);
  f32x4 p = f32x4{3.4645448e-05f, 3.4645448e-05f, 3.4645448e-05f, 3.4645448e-05f};
  auto step = [&](float c) { p = __builtin_elementwise_fma(p, a, f32x4{c, c, c, c}); };
  step(-0.000782622703f); step(0.00812418268f); step(-0.0534785727f); step(-0.458721816f); step(-1.15121768f); step(-0.999991402f);
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = fmaf(-fabsf(x[i]), __builtin_amdgcn_exp2f(p[i]), fmaxf(x[i], 0.f));
  return out;
}
__device__ __forceinline__ float gelu_erf(float x) {
#ifdef GELU_ABL
  return x;
#endif
  const float a = fminf(fabsf(x), 6.f);
  float p = 3.4645448e-05f;
  p = fmaf(p, a, -0.000782622703f); p = fmaf(p, a, 0.00812418268f); p = fmaf(p, a, -0.0534785727f);
  p = fmaf(p, a, -0.458721816f); p = fmaf(p, a, -1.15121768f); p = fmaf(p, a, -0.999991402f);
  return fmaf(-fabsf(x), __builtin_amdgcn_exp2f(p), fmaxf(x, 0.f));
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }

__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// bias / per-sample bias / residual / activation for 4 consecutive columns n..n+3 of row m
// folded LayerNorm: acc -> rstd * acc - rstd * mean * s (see GemmArgs)
__device__ __forceinline__ f32x4 ln_fold(const GemmArgs& g, int z, int m, int n, f32x4 v) {
  if (g.ln_swapped) {
    const float sm = g.ln_s[m];
    const int64_t t = z * g.ln_zstride + n;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = n + i < g.n_valid;
      const float rs = ok ? g.ln_rs[t + i] : 0.f, rm = ok ? g.ln_rm[t + i] : 0.f;
      v[i] = v[i] * rs - rm * sm;
    }
    return v;
  }
  const float rs = g.ln_rs[m], rm = g.ln_rm[m];
  const f32x4 sv = *(const f32x4*)(g.ln_s + n);
  return v * rs - sv * rm;
}

// the same for a wave's whole accumulator tile (rows mr + 16 i, columns nc + 16 j .. +3), statistics and column sums
// loaded once up front: inside the store loop they would be re-fetched after every store (possible aliasing with C)
template <int MT_, int NT_, typename ColFn>
__device__ __forceinline__ void ln_fold_tile(const GemmArgs& g, int z, int mr, int mstep, ColFn cj, float alpha, f32x4 (&acc)[MT_][NT_]) {
  if (g.ln_swapped) {
    float sm[MT_];
    f32x4 rs[NT_], rm[NT_];
#pragma unroll
    for (int i = 0; i < MT_; ++i) sm[i] = (mr + mstep * i < g.M) ? g.ln_s[mr + mstep * i] : 0.f;
#pragma unroll
    for (int j = 0; j < NT_; ++j) {
      const int n = cj(j);
      const int64_t t = z * g.ln_zstride + n;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = n + e < g.n_valid;
        rs[j][e] = ok ? g.ln_rs[t + e] : 0.f;
        rm[j][e] = ok ? g.ln_rm[t + e] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < MT_; ++i)
#pragma unroll
      for (int j = 0; j < NT_; ++j) acc[i][j] = acc[i][j] * alpha * rs[j] - rm[j] * sm[i];
  } else {
    float rs[MT_], rm[MT_];
    f32x4 sv[NT_];
#pragma unroll
    for (int i = 0; i < MT_; ++i) {
      const bool ok = mr + mstep * i < g.M;
      rs[i] = ok ? g.ln_rs[mr + mstep * i] : 0.f;
      rm[i] = ok ? g.ln_rm[mr + mstep * i] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NT_; ++j) sv[j] = (cj(j) < g.N) ? *(const f32x4*)(g.ln_s + cj(j)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MT_; ++i)
#pragma unroll
      for (int j = 0; j < NT_; ++j) acc[i][j] = acc[i][j] * (alpha * rs[i]) - sv[j] * rm[i];
  }
}

// (the kernels fold LayerNorm into their accumulators with ln_fold_tile before calling this; the split-K reduce with ln_fold)
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
    h16x4 r = *(const h16x4*)(g.residual + (int64_t)z * g.sC + (int64_t)m * g.ldr + n);
    v[0] += (float)r[0]; v[1] += (float)r[1]; v[2] += (float)r[2]; v[3] += (float)r[3];
  }
  if (g.act == ACT_SILU) {
    for (int i = 0; i < 4; ++i) v[i] = silu_f(v[i]);
  } else if (g.act == ACT_GELU) {
    v = gelu_erf4(v);
  }
  return v;
}
__device__ __forceinline__ h16x4 to_h16x4(f32x4 v) {
  h16x4 w;
  w[0] = (h16)v[0]; w[1] = (h16)v[1]; w[2] = (h16)v[2]; w[3] = (h16)v[3];
  return w;
}
__device__ __forceinline__ void epi_store(const GemmArgs& g, int z, int m, int n, f32x4 v) {
  v = epi_value(g, z, m, n, v);
  int64_t o = (int64_t)z * g.sC + (int64_t)m * g.ldc + n;
  if (g.out_f32) *(f32x4*)((float*)g.C + o) = v;
  else *(h16x4*)((h16*)g.C + o) = to_h16x4(v);
}
__device__ __forceinline__ f32x4 geglu_value(const GemmArgs& g, int m, int nh, f32x4 h, f32x4 gt) {
  if (g.bias) {
    h += *(const f32x4*)(g.bias + nh);
    gt += *(const f32x4*)(g.bias + nh + 16);
  }
  return h * gelu_erf4(gt);
}

// sum over the 16 lanes of a DPP row (lanes sharing lane >> 4), result in every lane: four v_add_f32 with DPP operand
// swizzles (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror) instead of four ds_bpermute round trips
__device__ __forceinline__ float row16_sum(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));
  return x;
}

// raw barrier that the compiler may not move LDS accesses across
__device__ __forceinline__ void bar() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// The epilogue of a wave's whole accumulator tile: rows mr + mstep * i (i < MT_), columns nc + 16 j .. +3 (j < NT_).
// Every load (bias, per-sample bias, residual, LayerNorm statistics) is issued BEFORE the first store: written as a
// per-fragment loop of load -> use -> store, each load has to wait behind the previous store (the compiler cannot prove
// that C does not alias them), i.e. one exposed L2 round trip per fragment and operand — 40-60 of them per tile, more
// than the whole K loop of a short-K GEMM.
// GN_: additionally emit the GroupNorm column sums of the stored values (GemmArgs::gn_part): each lane sums its rows, a
// 16-lane butterfly finishes the wave's rows, the waves of the tile meet in LDS (`lds`: the tile's staging memory, free after
// the K loop) and the first BN_ threads add them in a fixed order.  WM_ = waves along M, wm / wn = this wave's position,
// BNH_ = columns per wave, tile_m = row-tile index, n0 = first column of the tile.
// LN_ (dense GEMMs only): additionally emit the LayerNorm row partials of the stored values (GemmArgs::ln_part).
// WIDE_ = 2 of epi_tile: weight row (inside a wave's run of NT 16-column tiles) that belongs in MFMA operand row x = 16 j + r of that run
template <int NT_>
__device__ __forceinline__ int epi_perm_col(int x) {
  const int j = x >> 4, r = x & 15;
  return j < (NT_ & ~1) ? 32 * (j >> 1) + 8 * (r >> 2) + 4 * (j & 1) + (r & 3) : x;
}

template <int MT_, int NT_, bool LN_ = false, int WIDE_ = 0>
__device__ __forceinline__ void epi_tile(const GemmArgs& g, int z, int mr, int mstep, int nc, f32x4 (&acc)[MT_][NT_],
                                         char* lds = nullptr, int WM_ = 0, int wm = 0, int wn = 0, int tile_m = 0, int n0 = 0) {
  const bool geglu = g.act == ACT_GEGLU;
  const bool emit_gn = g.gn_part != nullptr && lds != nullptr;
  const bool emit_ln = LN_ && g.ln_part != nullptr && lds != nullptr;
  const float* const bias_z = g.bias ? g.bias + (int64_t)z * g.bias_zs : nullptr;
  float alpha = g.alpha;
  // ---- wide form (16-bit outputs, no GEGLU): a lane of a 16 x 16 MFMA tile holds 4 consecutive output columns (8 bytes of a row);
  // v_permlane16_swap_b32 between the tiles j and j + 1 of a pair (odd 16-lane rows of tile j <-> even rows of tile j + 1, one swap per
  // register) leaves every lane with 8 CONSECUTIVE columns — lane rows 0, 2 take columns 0-7 / 8-15 of tile j, rows 1, 3 columns 0-7 /
  // 8-15 of tile j + 1 — so that a row is stored (and its residual loaded) 16 bytes per lane: half the store / load instructions of the
  // epilogue, which is issue-bound (cdna_hip_programming.md T21; the K = 320 / 640 launches spend half to two thirds of their time
  // here).  cj(j) = first of the 4 columns acc[i][j] holds in this lane, before or after the exchange.
  // WIDE_ is a compile-time choice of the kernel instantiation (the launcher picks it when the output is 16-bit and the activation is
  // not GEGLU): with the two forms selected at run time inside one kernel the allocator spills 250+ registers at NT_ = 5
  // WIDE_ = 2 (round 5): the SAME lane layout without the exchange — the kernel dealt the rows of W to the MFMA operand rows of a tile pair so
  // that a lane's quads of tiles (j, j + 1) ARE 8 consecutive columns (epi_perm_col below: operand row r of tile j <-> column 32 (j >> 1) +
  // 8 (r >> 2) + 4 (j & 1) + (r & 3); an unpaired last tile stays natural).  No permlane instructions, no extra registers: the form for the
  // 160-column tiles, whose exchange form spills.
  constexpr bool wide = WIDE_ != 0 && NT_ >= 2;
  // column of acc[i][j][0] in this lane: paired tiles cb + 16 (j & ~1) + (j & 1) * cstep, an unpaired last tile nc + 16 j (two registers,
  // not a table: these kernels sit at the 256-register limit)
  int cb = nc, cstep = 16;
  if constexpr (wide && WIDE_ == 2) {
    const int lq_ = (threadIdx.x & 63) >> 4;
    cb = nc + 4 * lq_;                                    // nc = base + 4 lq: the lane's 8 columns start at base + 8 lq
    cstep = 4;
  }
  if constexpr (wide && WIDE_ == 1) {
    const int lq_ = (threadIdx.x & 63) >> 4;
    cb = (lq_ & 1) ? nc - 4 * lq_ + 16 + 4 * (lq_ - 1) : nc;
    cstep = 4;
#pragma unroll
    for (int j = 0; j + 1 < NT_; j += 2) {
#pragma unroll
      for (int i = 0; i < MT_; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // inline asm, not __builtin_amdgcn_permlane16_swap: hipcc (ROCm 7.2) folds the unrolled builtin calls on vector elements into
          // ONE swap whose result it copies to every element (tools/probe/probe_permlane16.hip shows the instruction itself is fine).
          // "s_nop 1": the two wait states between a VALU write of an operand and the permlane read (hipcc pads nothing inside asm).
          float x = acc[i][j][e], y = acc[i][j + 1][e];
          asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
          acc[i][j][e] = x;
          acc[i][j + 1][e] = y;
        }
    }
  }
  auto cj = [&](int j) -> int { return j < (NT_ & ~1) ? cb + 16 * (j & ~1) + (j & 1) * cstep : nc + 16 * j; };
  if (g.ln_rs) { ln_fold_tile<MT_, NT_>(g, z, mr, mstep, cj, alpha, acc); alpha = 1.f; }
  // ---- loads
  f32x4 bj[NT_];
  float bi[MT_];
#pragma unroll
  for (int j = 0; j < NT_; ++j) {
    const int n = cj(j);
    bj[j] = (bias_z && !g.bias_row && n < g.N) ? *(const f32x4*)(bias_z + n) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int i = 0; i < MT_; ++i) {
    const int m = mr + mstep * i;
    bi[i] = (bias_z && g.bias_row && m < g.M) ? bias_z[m] : 0.f;
  }
  const bool has_bbn = g.bias_bn != nullptr, has_res = g.residual != nullptr && !geglu;
  const int ldbn = g.bias_bn_ld ? g.bias_bn_ld : g.N;
  // two row groups: halves the registers the prefetched operands need (one more exposed round trip, not twenty)
  constexpr int RG = (MT_ + 1) / 2;
  constexpr int NTP = NT_ & ~1;          // tiles that belong to a pair in the wide form
#pragma unroll
  for (int i0 = 0; i0 < MT_; i0 += RG) {
    f32x4 ex[RG][NT_];        // per-sample bias + residual of this row group (one array: a conv has one or the other)
#pragma unroll
    for (int ii = 0; ii < RG; ++ii)
#pragma unroll
      for (int j = 0; j < NT_; ++j) ex[ii][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (has_bbn) {
#pragma unroll
      for (int ii = 0; ii < RG; ++ii) {
        const int m = mr + mstep * (i0 + ii);
        const float* row = g.bias_bn + (int64_t)(m / g.rows_per_batch) * ldbn;
#pragma unroll
        for (int j = 0; j < NT_; ++j) {
          const int n = cj(j);
          if (i0 + ii < MT_ && m < g.M && n < g.N) ex[ii][j] = *(const f32x4*)(row + n);
        }
      }
    }
    if constexpr (wide) {
      // ================= wide form: pairs of tiles as 8 consecutive columns per lane, 16-byte residual loads and stores
      if (has_res) {
#pragma unroll
        for (int ii = 0; ii < RG; ++ii) {
          const int m = mr + mstep * (i0 + ii);
          const h16* rrow = g.residual + (int64_t)z * g.sC + (int64_t)m * g.ldr;
          if (!(i0 + ii < MT_ && m < g.M)) continue;
#pragma unroll
          for (int j = 0; j < NTP; j += 2) {
            const int n = cj(j);
            if (n + 4 < g.N) {
              const h16x8 r = *(const h16x8*)(rrow + n);
#pragma unroll
              for (int e = 0; e < 4; ++e) { ex[ii][j][e] += (float)r[e]; ex[ii][j + 1][e] += (float)r[4 + e]; }
            } else if (n < g.N) {
              const h16x4 r = *(const h16x4*)(rrow + n);
#pragma unroll
              for (int e = 0; e < 4; ++e) ex[ii][j][e] += (float)r[e];
            }
          }
          if constexpr (NT_ & 1) {
            const int n = cj(NT_ - 1);
            if (n < g.N) {
              const h16x4 r = *(const h16x4*)(rrow + n);
#pragma unroll
              for (int e = 0; e < 4; ++e) ex[ii][NT_ - 1][e] += (float)r[e];
            }
          }
        }
      }
#pragma unroll
      for (int ii = 0; ii < RG; ++ii) {
        const int i = i0 + ii;
        if (i >= MT_) continue;
        const int m = mr + mstep * i;
        if (m >= g.M) continue;
        auto value = [&](int j) -> h16x4 {
          f32x4 v = acc[i][j] * alpha + bj[j] + bi[i];
          v += ex[ii][j];
          if (g.act == ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
          } else if (g.act == ACT_GELU) {
            v = gelu_erf4(v);
          }
          const h16x4 w = to_h16x4(v);
          // the accumulator is dead from here: keep what the consumer will read (the rounded values) in it for the column / row sums below
          if (emit_gn || emit_ln) { acc[i][j][0] = (float)w[0]; acc[i][j][1] = (float)w[1]; acc[i][j][2] = (float)w[2]; acc[i][j][3] = (float)w[3]; }
          return w;
        };
        h16* crow = (h16*)g.C + (int64_t)z * g.sC + (int64_t)m * g.ldc;
#pragma unroll
        for (int j = 0; j < NTP; j += 2) {
          const int n = cj(j);
          if (n + 4 < g.N) {
            const h16x4 w0 = value(j), w1 = value(j + 1);
            h16x8 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) { w[e] = w0[e]; w[4 + e] = w1[e]; }
            *(h16x8*)(crow + n) = w;
          } else if (n < g.N) {
            *(h16x4*)(crow + n) = value(j);
          }
        }
        if constexpr (NT_ & 1) {
          const int n = cj(NT_ - 1);
          if (n < g.N) *(h16x4*)(crow + n) = value(NT_ - 1);
        }
      }
    } else {
    // ================= narrow form (GEGLU, f32 outputs, single-tile waves): 4 columns per lane and tile
    if (has_res) {
#pragma unroll
      for (int ii = 0; ii < RG; ++ii) {
        const int m = mr + mstep * (i0 + ii);
#pragma unroll
        for (int j = 0; j < NT_; ++j) {
          const int n = nc + 16 * j;
          if (i0 + ii < MT_ && m < g.M && n < g.N) {
            const h16x4 r = *(const h16x4*)(g.residual + (int64_t)z * g.sC + (int64_t)m * g.ldr + n);
            ex[ii][j][0] += (float)r[0]; ex[ii][j][1] += (float)r[1]; ex[ii][j][2] += (float)r[2]; ex[ii][j][3] += (float)r[3];
          }
        }
      }
    }
    // ---- arithmetic and stores of this row group
#pragma unroll
    for (int ii = 0; ii < RG; ++ii) {
      const int i = i0 + ii;
      if (i >= MT_) continue;
      const int m = mr + mstep * i;
      if (m >= g.M) continue;
      if (geglu) {
        if constexpr ((NT_ & 1) == 0) {
#pragma unroll
          for (int j = 0; j < NT_; j += 2) {
            const int nh = nc + 16 * j;                       // packed column of the h tile; the gate tile follows
            if (nh >= g.N) continue;
            const f32x4 h = acc[i][j] * alpha + bj[j], gt = acc[i][j + 1] * alpha + bj[j + 1];
            const f32x4 v = h * gelu_erf4(gt);
            const int oc = (nh >> 5) * 16 + (nh & 15);
            *(h16x4*)((h16*)g.C + (int64_t)z * g.sC + (int64_t)m * g.ldc + oc) = to_h16x4(v);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < NT_; ++j) {
          const int n = nc + 16 * j;
          if (n >= g.N) continue;
          f32x4 v = acc[i][j] * alpha + bj[j] + bi[i];
          v += ex[ii][j];
          if (g.act == ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
          } else if (g.act == ACT_GELU) {
            v = gelu_erf4(v);
          }
          const int64_t o = (int64_t)z * g.sC + (int64_t)m * g.ldc + n;
          if (g.out_f32) *(f32x4*)((float*)g.C + o) = v;
          else {
            const h16x4 w = to_h16x4(v);
            *(h16x4*)((h16*)g.C + o) = w;
            if (emit_gn || emit_ln) { acc[i][j][0] = (float)w[0]; acc[i][j][1] = (float)w[1]; acc[i][j][2] = (float)w[2]; acc[i][j][3] = (float)w[3]; }
          }
        }
      }
    }
    }
  }
  if (emit_gn) {
    // acc[i][j] now holds the stored values (rows past M were skipped above: zero them).  One column tile at a time: sum the
    // wave's rows in the lane, then a butterfly over the 16 lanes that share lane >> 4 (lane bits 0-3).
    const int lane = threadIdx.x & 63;
    const int BNH = NT_ * 16;                               // columns of a wave tile
    float* sc = (float*)lds;                                // [WM_][waves_n * BNH][2]
    const int ncols = (blockDim.x >> 6) / WM_ * BNH;
    bar();
#pragma unroll
    for (int j = 0; j < NT_; ++j) {
      f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, b = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < MT_; ++i) {
        const bool ok = (mr + mstep * i < g.M) && (cj(j) < g.N);
        const f32x4 v = ok ? acc[i][j] : f32x4{0.f, 0.f, 0.f, 0.f};
        a += v; b += v * v;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = row16_sum(a[e]), y = row16_sum(b[e]);
        if ((lane & 15) == 0) {
          const int col = cj(j) - n0 + e;                   // column of the tile this lane's register e of tile j holds
          sc[(wm * ncols + col) * 2] = x;
          sc[(wm * ncols + col) * 2 + 1] = y;
        }
      }
    }
    // the ds_writes above must have LANDED before any other wave passes the barrier: s_barrier alone does not wait for them on gfx950
    // (back-off barrier: the compiler inserts no s_waitcnt in front of a bare __builtin_amdgcn_s_barrier).  Without this wait the last
    // entries written — the final tile's columns — were occasionally read stale by the summing threads when the LDS was busy with a
    // co-resident workgroup of ANOTHER context: GroupNorm statistics off in the last bits, run to run (found by tools/stress_pair.py, round 6)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bar();
    for (int col = threadIdx.x; col < ncols; col += blockDim.x) {
      const int n = n0 + col;
      if (n >= g.N) continue;
      float a = 0.f, b = 0.f;
      for (int w = 0; w < WM_; ++w) { a += sc[(w * ncols + col) * 2]; b += sc[(w * ncols + col) * 2 + 1]; }
      *(float2*)(g.gn_part + ((int64_t)tile_m * g.N + n) * 2) = make_float2(a, b);
    }
  }
  if (emit_ln) {
    // LayerNorm row partials of this tile's columns: the four lanes l15 + 16 * lq hold a row's columns of the wave tile; the
    // waves along N meet in LDS and are added in order.  acc holds the stored (rounded) values (rows / columns past M / N: skipped).
    const int lane = threadIdx.x & 63;
    const int WN_ = (blockDim.x >> 6) / WM_;
    const int trows = WM_ * MT_ * 16;
    float* sc = (float*)lds;                                // [WN_][trows][2]
    bar();
#pragma unroll
    for (int i = 0; i < MT_; ++i) {
      const int m = mr + mstep * i;
      float a = 0.f, b = 0.f;
      if (m < g.M) {
#pragma unroll
        for (int j = 0; j < NT_; ++j)
          if (cj(j) < g.N) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float v = acc[i][j][e]; a += v; b += v * v; }
          }
      }
      a += __shfl_xor(a, 16); b += __shfl_xor(b, 16);
      a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
      if ((lane >> 4) == 0) {
        const int rl = m - tile_m * trows;
        *(float2*)(sc + ((size_t)wn * trows + rl) * 2) = make_float2(a, b);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (as in the GroupNorm sums above: the writes have landed before anyone passes the barrier)
    bar();
    const int tile_n = n0 / (WN_ * NT_ * 16);
    for (int rl = threadIdx.x; rl < trows; rl += blockDim.x) {
      const int m = tile_m * trows + rl;
      if (m >= g.M) continue;
      float a = 0.f, b = 0.f;
      for (int w = 0; w < WN_; ++w) { a += sc[((size_t)w * trows + rl) * 2]; b += sc[((size_t)w * trows + rl) * 2 + 1]; }
      *(float2*)(g.ln_part + (((int64_t)z * g.M + m) * g.ln_tiles + tile_n) * 2) = make_float2(a, b);
    }
  }
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
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
  }
}

}  // namespace

}  // namespace SDNS
