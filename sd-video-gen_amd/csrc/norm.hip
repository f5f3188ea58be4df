// GroupNorm (+SiLU) on NHWC bf16, LayerNorm over rows, row softmax.  HBM-bound: 16-B vector
// loads/stores, f32 statistics, wave-shuffle reductions.
#include "kernels.h"
#include <cstdlib>

namespace SDNS {


namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// ---- GroupNorm stage 1: per (sample, pixel-chunk) partial sums per channel-group ----------------
// grid (nchunk, B); block = CV*PL threads where CV = C/8 channel vectors, PL pixel lanes.
// partial[b][chunk][group][2]
__global__ void gn_stats_kernel(const h16* __restrict__ x, int C1, const h16* __restrict__ x2, int C2,
                                float* __restrict__ partial, int HW, int groups, int nchunk, int CV, int PL) {
  extern __shared__ float sh[];   // [PL][C][2] then reused
  const int C = C1 + C2;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int tid = threadIdx.x;
  const int pix_per_chunk = (HW + nchunk - 1) / nchunk;
  const int p_begin = chunk * pix_per_chunk;
  const int p_end = min(HW, p_begin + pix_per_chunk);
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
  const int nthr = CV * PL;
  if (tid < nthr) {
    const int cv = tid % CV, pl = tid / CV;
    const int c0 = cv * 8;
    const h16* src; int ld, coff;
    if (c0 < C1) { src = x; ld = C1; coff = c0; } else { src = x2; ld = C2; coff = c0 - C1; }
    // four pixels per trip: four independent 16-byte loads in flight per thread (a one-load-per-trip loop is a chain of
    // exposed memory latencies: ~10 trips x ~1 us)
    const h16* sp = src + ((int64_t)b * HW + p_begin + pl) * ld + coff;
    const int64_t st = (int64_t)PL * ld;
    int p = p_begin + pl;
    for (; p + 3 * PL < p_end; p += 4 * PL, sp += 4 * st) {
      const h16x8 v0 = *(const h16x8*)sp, v1 = *(const h16x8*)(sp + st), v2 = *(const h16x8*)(sp + 2 * st), v3 = *(const h16x8*)(sp + 3 * st);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f0 = (float)v0[j], f1 = (float)v1[j], f2 = (float)v2[j], f3 = (float)v3[j];
        s[j] += (f0 + f1) + (f2 + f3);
        q[j] += (f0 * f0 + f1 * f1) + (f2 * f2 + f3 * f3);
      }
    }
    for (; p < p_end; p += PL, sp += st) {
      const h16x8 v = *(const h16x8*)sp;
#pragma unroll
      for (int j = 0; j < 8; ++j) { float f = (float)v[j]; s[j] += f; q[j] += f * f; }
    }
    float* dst = sh + ((int64_t)pl * C + c0) * 2;
#pragma unroll
    for (int j = 0; j < 8; ++j) { dst[2 * j] = s[j]; dst[2 * j + 1] = q[j]; }
  }
  __syncthreads();
  // reduce over pixel lanes and over the channels of each group
  const int cpg = C / groups;
  for (int g = tid; g < groups; g += blockDim.x) {
    float ss = 0.f, qq = 0.f;
    for (int pl = 0; pl < PL; ++pl)
      for (int cc = 0; cc < cpg; ++cc) {
        const float* e = sh + ((int64_t)pl * C + g * cpg + cc) * 2;
        ss += e[0]; qq += e[1];
      }
    float* o = partial + (((int64_t)b * nchunk + chunk) * groups + g) * 2;
    o[0] = ss; o[1] = qq;
  }
}

// ---- GroupNorm stage 2: finish the statistics, normalise, affine, optional SiLU ------------------
// grid (nblk, B); block = CV*PL threads like stage 1.  Each thread owns one 8-channel vector: it folds
// (mean, rstd, gamma, beta) into 8 (scale, shift) pairs once, then streams its pixels with 16-B loads/stores
// and 8 FMAs per vector — no divisions or table lookups in the loop.
__global__ void gn_apply_kernel(const h16* __restrict__ x, int C1, const h16* __restrict__ x2, int C2,
                                const float* __restrict__ partial, const float* __restrict__ gamma,
                                const float* __restrict__ beta, h16* __restrict__ out, int HW, int groups,
                                int nchunk, float eps, int silu, int CV, int PL) {
  __shared__ float mean_s[64], rstd_s[64];
  const int C = C1 + C2;
  const int b = blockIdx.y;
  const int cpg = C / groups;
  for (int g = threadIdx.x; g < groups; g += blockDim.x) {
    if (nchunk == 0) {   // finished statistics (gn_finish_kernel): (mean, rstd) per (sample, group)
      mean_s[g] = partial[((int64_t)b * groups + g) * 2];
      rstd_s[g] = partial[((int64_t)b * groups + g) * 2 + 1];
      continue;
    }
    float ss = 0.f, qq = 0.f;
    for (int ch = 0; ch < nchunk; ++ch) {
      const float* e = partial + (((int64_t)b * nchunk + ch) * groups + g) * 2;
      ss += e[0]; qq += e[1];
    }
    const float n = (float)cpg * (float)HW;
    const float mean = ss / n;
    const float var = fmaxf(qq / n - mean * mean, 0.f);
    mean_s[g] = mean;
    rstd_s[g] = rsqrtf(var + eps);
  }
  __syncthreads();
  const int tid = threadIdx.x;
  if (tid >= CV * PL) return;
  const int cv = tid % CV, pl = tid / CV;
  const int c0 = cv * 8;
  float sc[8], sh[8];
  {
    int g = c0 / cpg, rem = c0 - g * cpg;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (rem == cpg) { rem = 0; ++g; }
      ++rem;
      const float a = rstd_s[g] * gamma[c0 + j];
      sc[j] = a;
      sh[j] = beta[c0 + j] - mean_s[g] * a;
    }
  }
  const h16* src; int ld, coff;
  if (c0 < C1) { src = x; ld = C1; coff = c0; } else { src = x2; ld = C2; coff = c0 - C1; }
  const int per_blk = (HW + gridDim.x - 1) / gridDim.x;
  const int p_begin = blockIdx.x * per_blk;
  const int p_end = min(HW, p_begin + per_blk);
  const h16* sp = src + ((int64_t)b * HW + p_begin + pl) * ld + coff;
  h16* dp = out + ((int64_t)b * HW + p_begin + pl) * C + c0;
  const int64_t sstep = (int64_t)PL * ld, dstep = (int64_t)PL * C;
  auto apply = [&](const h16x8& v) -> h16x8 {
    h16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float f = fmaf((float)v[j], sc[j], sh[j]);
      if (silu) f = f * __builtin_amdgcn_rcpf(1.f + __expf(-f));
      o[j] = (h16)f;
    }
    return o;
  };
  int p = p_begin + pl;
  // four pixels per trip: the four loads are issued together (see gn_stats_kernel)
  for (; p + 3 * PL < p_end; p += 4 * PL, sp += 4 * sstep, dp += 4 * dstep) {
    const h16x8 v0 = *(const h16x8*)sp, v1 = *(const h16x8*)(sp + sstep), v2 = *(const h16x8*)(sp + 2 * sstep), v3 = *(const h16x8*)(sp + 3 * sstep);
    *(h16x8*)dp = apply(v0);
    *(h16x8*)(dp + dstep) = apply(v1);
    *(h16x8*)(dp + 2 * dstep) = apply(v2);
    *(h16x8*)(dp + 3 * dstep) = apply(v3);
  }
  for (; p < p_end; p += PL, sp += sstep, dp += dstep) *(h16x8*)dp = apply(*(const h16x8*)sp);
}

// ---- The same apply pass with an MX fp8 output (the input of conv_halo_fp8.hip): e4m3 [pixel][Cp] + one E8M0 scale per 32 channels
// [pixel][Cp / 32], Cp = C rounded up to 128 (padding channels: zero data, scale 1).  A thread owns one 8-channel vector as above; the four
// threads of a 32-channel block are adjacent lanes and agree on the block's shared exponent with two shuffles (OCP MX v1.0 section 6.3:
// floor(log2(amax)) - 8, elements RNE + saturation at +-448 — the conversion of quant_mx_kernel / quant_act_mx_kernel, bit for bit).
// 8 + 0.25 bytes written per vector instead of 16: the pass moves 40 % fewer bytes than the 16-bit one and no quantisation pass follows.
__global__ void gn_apply_mx_kernel(const h16* __restrict__ x, int C1, const h16* __restrict__ x2, int C2, const float* __restrict__ stats,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, uint8_t* __restrict__ q,
                                   uint8_t* __restrict__ sc, int HW, int groups, int silu, int CVp, int PL) {
  __shared__ float mean_s[64], rstd_s[64];
  const int C = C1 + C2, Cp = CVp * 8;
  const int b = blockIdx.y;
  const int cpg = C / groups;
  for (int g = threadIdx.x; g < groups; g += blockDim.x) {
    mean_s[g] = stats[((int64_t)b * groups + g) * 2];
    rstd_s[g] = stats[((int64_t)b * groups + g) * 2 + 1];
  }
  __syncthreads();
  const int tid = threadIdx.x;
  const bool live = tid < CVp * PL;                      // (idle lanes of the last wave still take part in the shuffles)
  const int cv = live ? tid % CVp : 0, pl = live ? tid / CVp : 0;
  const int c0 = cv * 8;
  const bool real = live && c0 < C;                      // padding vectors write zeros
  float scl[8], sh[8];
  if (real) {
    int g = c0 / cpg, rem = c0 - g * cpg;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (rem == cpg) { rem = 0; ++g; }
      ++rem;
      const float a = rstd_s[g] * gamma[c0 + j];
      scl[j] = a;
      sh[j] = beta[c0 + j] - mean_s[g] * a;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) { scl[j] = 0.f; sh[j] = 0.f; }
  }
  const h16* src = x; int ld = C1, coff = c0;
  if (real && c0 >= C1) { src = x2; ld = C2; coff = c0 - C1; }
  const int per_blk = (HW + gridDim.x - 1) / gridDim.x;
  const int p_begin = blockIdx.x * per_blk;
  const int p_end = min(HW, p_begin + per_blk);
  const int SB = Cp >> 5;
  for (int p0 = p_begin; p0 < p_end; p0 += PL) {         // every thread walks the same trips (the shuffles need whole quads)
    const int p = p0 + pl;
    const bool on = live && p < p_end;
    float f[8];
    if (on && real) {
      const h16x8 v = *(const h16x8*)(src + ((int64_t)b * HW + p) * ld + coff);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = fmaf((float)v[j], scl[j], sh[j]);
        if (silu) t = t * __builtin_amdgcn_rcpf(1.f + __expf(-t));
        f[j] = (float)(h16)t;                            // the value the 16-bit pass would have stored: same numbers into the quantiser
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = 0.f;
    }
    float am = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) am = fmaxf(am, fabsf(f[j]));
    am = fmaxf(am, __shfl_xor(am, 1));
    am = fmaxf(am, __shfl_xor(am, 2));
    int e = am > 0.f ? ((__float_as_int(am) >> 23) & 0xff) - 127 - 8 : 0;
    e = e < -127 ? -127 : (e > 126 ? 126 : e);
    const float inv = __int_as_float((127 - e) << 23);
    float t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = fminf(fmaxf(f[j] * inv, -448.f), 448.f);
    unsigned lo = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], 0, false) & 0xffffu;
    lo |= ((unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], 0, false) & 0xffffu) << 16;
    unsigned hi = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(t[4], t[5], 0, false) & 0xffffu;
    hi |= ((unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(t[6], t[7], 0, false) & 0xffffu) << 16;
    if (on) {
      const int64_t pix = (int64_t)b * HW + p;
      *(uint2*)(q + pix * Cp + c0) = make_uint2(lo, hi);
      if ((cv & 3) == 0) sc[pix * SB + (cv >> 2)] = (uint8_t)(e + 127);
    }
  }
}

// ---- GroupNorm statistics from the producers' column sums (GemmArgs::gn_part): one workgroup per (group, sample) adds the
// tiles_per_sample x cpg (sum, sumsq) pairs of its channels in a fixed order; a group may straddle the two sources of a concat.
__global__ void __launch_bounds__(256) gn_finish_kernel(const float* __restrict__ p1, int C1, int tps1, const float* __restrict__ p2, int C2,
                                                         int tps2, float* __restrict__ stats, int HW, int groups, float eps) {
  __shared__ float red[8];
  const int C = C1 + C2, cpg = C / groups;
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  float s = 0.f, q = 0.f;
  const int c_lo = g * cpg, c_hi = c_lo + cpg;
  // source 1 channels [c_lo, min(c_hi, C1)), source 2 channels [max(c_lo, C1), c_hi) - C1
  const int a1 = min(c_lo, C1), b1 = min(c_hi, C1), n1 = b1 - a1;
  for (int idx = tid; idx < tps1 * n1; idx += 256) {
    const int t = idx / n1, c = a1 + (idx - t * n1);
    const float2 v = *(const float2*)(p1 + (((int64_t)b * tps1 + t) * C1 + c) * 2);
    s += v.x; q += v.y;
  }
  const int a2 = max(c_lo, C1) - C1, b2 = max(c_hi, C1) - C1, n2 = b2 - a2;
  for (int idx = tid; idx < tps2 * n2; idx += 256) {
    const int t = idx / n2, c = a2 + (idx - t * n2);
    const float2 v = *(const float2*)(p2 + (((int64_t)b * tps2 + t) * C2 + c) * 2);
    s += v.x; q += v.y;
  }
  s = wave_sum(s); q = wave_sum(q);
  if ((tid & 63) == 0) { red[(tid >> 6) * 2] = s; red[(tid >> 6) * 2 + 1] = q; }
  __syncthreads();
  if (tid == 0) {
    s = (red[0] + red[2]) + (red[4] + red[6]);
    q = (red[1] + red[3]) + (red[5] + red[7]);
    const float n = (float)cpg * (float)HW;
    const float mean = s / n;
    stats[((int64_t)b * groups + g) * 2] = mean;
    stats[((int64_t)b * groups + g) * 2 + 1] = rsqrtf(fmaxf(q / n - mean * mean, 0.f) + eps);
  }
}

// ---- GroupNorm folded into the 1x1 projection that follows it (SpatialTransformer: norm -> proj_in, no activation between):
// one workgroup per (output row n, sample b)
__global__ void __launch_bounds__(256) gn_fold_weights_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ stats, h16* __restrict__ Wb, float* __restrict__ bb,
                                                               int N, int C, int groups) {
  __shared__ float red[4];
  const int n = blockIdx.x, b = blockIdx.y, cpg = C / groups;
  const float* wr = W + (int64_t)n * C;
  h16* o = Wb + ((int64_t)b * N + n) * C;
  const float* st = stats + (int64_t)b * groups * 2;
  float acc = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cpg;
    const float mean = st[2 * g], rstd = st[2 * g + 1];
    const float w = wr[c], a = gamma[c] * rstd;
    o[c] = (h16)(w * a);
    acc += w * (beta[c] - mean * a);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) bb[(int64_t)b * N + n] = (bias ? bias[n] : 0.f) + ((red[0] + red[1]) + (red[2] + red[3]));
}

// ---- GroupNorm for small images (one launch, one read): a workgroup owns one (sample, group) — HW x cpg values that
// fit its registers (<= 256 x 80 here: the 16 x 16 and 8 x 8 UNet levels, where the two-stage pair above is all launch
// latency: 22-27 us for 2.6-10 MB).  Values are fetched as 8-byte (4-channel) pieces, so a group may straddle the two
// sources of a virtual concat.  Deterministic: fixed per-thread order, shuffle + LDS tree.
typedef h16 h16x4n __attribute__((ext_vector_type(4)));
typedef h16 h16x8n __attribute__((ext_vector_type(8)));
// VW = channels per piece: 4 (8-byte pieces: any cpg % 4 == 0) or 8 (16-byte pieces, cpg % 8 == 0 — round 5: half the vector-memory
// instructions and half the predicated trips: 256 x 40 channels = 5 pieces per thread instead of 10)
template <int MAXCH, int VW>
__global__ void __launch_bounds__(256) gn_small_kernel(const h16* __restrict__ x, int C1, const h16* __restrict__ x2, int C2,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       h16* __restrict__ out, int HW, int groups, float eps, int silu) {
  typedef h16 vec_t __attribute__((ext_vector_type(VW)));
  __shared__ float red[8];
  const int C = C1 + C2, cpg = C / groups, nch = cpg / VW, total = HW * nch;
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  vec_t v[MAXCH];
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int id = tid + 256 * i;
    if (id < total) {
      const int p = id / nch, c = g * cpg + (id - p * nch) * VW;
      const h16* src = (c < C1) ? x + ((int64_t)b * HW + p) * C1 + c : x2 + ((int64_t)b * HW + p) * C2 + (c - C1);
      v[i] = *(const vec_t*)src;
#pragma unroll
      for (int j = 0; j < VW; ++j) { const float f = (float)v[i][j]; s += f; q += f * f; }
    }
  }
  s = wave_sum(s); q = wave_sum(q);
  if ((tid & 63) == 0) { red[(tid >> 6) * 2] = s; red[(tid >> 6) * 2 + 1] = q; }
  __syncthreads();
  s = (red[0] + red[2]) + (red[4] + red[6]);
  q = (red[1] + red[3]) + (red[5] + red[7]);
  const float n = (float)cpg * (float)HW;
  const float mean = s / n;
  const float rstd = rsqrtf(fmaxf(q / n - mean * mean, 0.f) + eps);
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int id = tid + 256 * i;
    if (id < total) {
      const int p = id / nch, c = g * cpg + (id - p * nch) * VW;
      vec_t o;
#pragma unroll
      for (int j4 = 0; j4 < VW; j4 += 4) {
        const float4 ga = *(const float4*)(gamma + c + j4), be = *(const float4*)(beta + c + j4);
        const float gg[4] = {ga.x, ga.y, ga.z, ga.w}, bb[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = rstd * gg[j];
          float f = fmaf((float)v[i][j4 + j], a, bb[j] - mean * a);
          if (silu) f = f * __builtin_amdgcn_rcpf(1.f + __expf(-f));
          o[j4 + j] = (h16)f;
        }
      }
      *(vec_t*)(out + ((int64_t)b * HW + p) * C + c) = o;
    }
  }
}

// ---- LayerNorm: one wave per row -------------------------------------------------------------------
__global__ void __launch_bounds__(256) layernorm_kernel(const h16* __restrict__ x, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, h16* __restrict__ out, int M, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int CV = C / 8;
  const h16* xr = x + (int64_t)row * C;
  // C <= 8*64*4 = 2048: up to 4 vectors per lane
  h16x8 v[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cv = lane + 64 * i;
    if (cv < CV) {
      v[i] = *(const h16x8*)(xr + cv * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (float)v[i][j];
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cv = lane + 64 * i;
    if (cv < CV) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { float d = (float)v[i][j] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cv = lane + 64 * i;
    if (cv < CV) {
      h16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int cc = cv * 8 + j;
        o[j] = (h16)(((float)v[i][j] - mean) * rstd * gamma[cc] + beta[cc]);
      }
      *(h16x8*)(out + (int64_t)row * C + cv * 8) = o;
    }
  }
}

// ---- LayerNorm statistics only (the normalisation itself is folded into the consuming GEMM) ---------------------------
__global__ void __launch_bounds__(256) ln_stats_kernel(const h16* __restrict__ x, float* __restrict__ rs, float* __restrict__ rm,
                                                       int M, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int CV = C / 8;
  const h16* xr = x + (int64_t)row * C;
  h16x8 v[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cv = lane + 64 * i;
    if (cv < CV) {
      v[i] = *(const h16x8*)(xr + cv * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (float)v[i][j];
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cv = lane + 64 * i;
    if (cv < CV) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { float d = (float)v[i][j] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) { rs[row] = rstd; rm[row] = rstd * mean; }
}

// rs / rm from the row partials the producing GEMM's epilogue left (GemmArgs::ln_part): tiles are added in order; the variance is
// E[x^2] - mean^2 of the bf16 values the consumer will read, in f32 (C <= 1280 terms of O(1..100) magnitude)
__global__ void ln_finish_kernel(const float* __restrict__ part, int tiles, float* __restrict__ rs, float* __restrict__ rm, int M, int C, float eps) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  float a = 0.f, b = 0.f;
  for (int t = 0; t < tiles; ++t) { const float2 p = *(const float2*)(part + ((int64_t)m * tiles + t) * 2); a += p.x; b += p.y; }
  const float mean = a / (float)C;
  const float var = fmaxf(b / (float)C - mean * mean, 0.f);
  const float rstd = rsqrtf(var + eps);
  rs[m] = rstd; rm[m] = rstd * mean;
}

// ---- LayerNorm folding at load time --------------------------------------------------------------------------------
// one block per output row: bias_out[n] = bias_in[n] + sum_k W[n][k] beta[k], then W[n][k] *= gamma[k]
__global__ void __launch_bounds__(256) fold_ln_kernel(float* __restrict__ w, const float* __restrict__ bin, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ bout, int K) {
  __shared__ float red[4];
  const int n = blockIdx.x;
  float* wr = w + (int64_t)n * K;
  float acc = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float v = wr[k];
    acc += v * beta[k];
    wr[k] = v * gamma[k];
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) bout[n] = (bin ? bin[n] : 0.f) + ((red[0] + red[1]) + (red[2] + red[3]));
}
__global__ void __launch_bounds__(256) rowsum_h16_kernel(const h16* __restrict__ w, float* __restrict__ out, int K) {
  __shared__ float red[4];
  const int n = blockIdx.x;
  const h16* wr = w + (int64_t)n * K;
  float acc = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) acc += (float)wr[k];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[n] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- row softmax: f32 scores -> bf16 probabilities, one block per row ------------------------------
__global__ void __launch_bounds__(256) softmax_rows_kernel(const float* __restrict__ sin, h16* __restrict__ pout, int cols,
                                    int ld_in, int ld_out, float scale) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const float* sr = sin + row * ld_in;
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, sr[c] * scale);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) sum += __expf(sr[c] * scale - mx);
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
  h16* pr = pout + row * ld_out;
  for (int c = threadIdx.x; c < ld_out; c += 256) pr[c] = (h16)((c < cols) ? __expf(sr[c] * scale - mx) * inv : 0.f);
}

}  // namespace

void gn_finish(svg_ctx* ctx, const GnStats& st1, int C1, const GnStats* st2, int C2, float* stats, int B, int HW, int groups, float eps,
               hipStream_t s) {
  if (!SVG_LAUNCHING(ctx)) return;
  char tag[96];
  snprintf(tag, sizeof(tag), "finish_B%d_HW%d_C%d", B, HW, C1 + C2);
  ProfScope ps(ctx, PK_GNORM, s, 0, 8.0 * B * ((double)st1.tiles_per_sample * C1 + (st2 ? (double)st2->tiles_per_sample * C2 : 0.0)), tag);
  hipLaunchKernelGGL(gn_finish_kernel, dim3(groups, B), dim3(256), 0, s, st1.part, C1, st1.tiles_per_sample, st2 ? st2->part : nullptr, C2,
                     st2 ? st2->tiles_per_sample : 0, stats, HW, groups, eps);
  check_launch("gn_finish");
}

void gn_fold_weights(const float* W, const float* bias, const float* gamma, const float* beta, const float* stats, h16* Wb, float* bb,
                     int B, int N, int C, int groups, hipStream_t s) {
  hipLaunchKernelGGL(gn_fold_weights_kernel, dim3(N, B), dim3(256), 0, s, W, bias, gamma, beta, stats, Wb, bb, N, C, groups);
  check_launch("gn_fold_weights");
}

void groupnorm(svg_ctx* ctx, const h16* x, int C1, const h16* x2, int C2, const float* gamma, const float* beta,
               h16* out, int B, int HW, int groups, float eps, int silu, hipStream_t s, const GnStats* st1, const GnStats* st2) {
  const int C = C1 + C2;
  SVG_CHECK(C % groups == 0 && C % 8 == 0 && C1 % 8 == 0 && groups <= 64, "groupnorm: C=%d groups=%d unsupported", C, groups);
  const int CV = C / 8;
  SVG_CHECK(CV <= 1024, "groupnorm: C too large");
  {
    const int cpg = C / groups;
    static const int no_small = getenv("SVG_GN_NOSMALL") ? atoi(getenv("SVG_GN_NOSMALL")) : 0;
    if (!no_small && HW <= 256 && cpg % 4 == 0 && C1 % 4 == 0 && (int64_t)HW * (cpg / 4) <= 256 * 20) {
      if (!SVG_LAUNCHING(ctx)) return;
      char tag[96];
      snprintf(tag, sizeof(tag), "small_B%d_HW%d_C%d", B, HW, C);
      ProfScope ps(ctx, PK_GNORM, s, 0, 2.0 * B * HW * C * 2, tag);
      static const int vw8 = getenv("SVG_GN_SMALL_VW8") ? atoi(getenv("SVG_GN_SMALL_VW8")) : 1;
      if (vw8 && cpg % 8 == 0 && C1 % 8 == 0) {             // 16-byte pieces
        if ((int64_t)HW * (cpg / 8) <= 256 * 5)
          hipLaunchKernelGGL((gn_small_kernel<5, 8>), dim3(groups, B), dim3(256), 0, s, x, C1, x2, C2, gamma, beta, out, HW, groups, eps, silu);
        else
          hipLaunchKernelGGL((gn_small_kernel<10, 8>), dim3(groups, B), dim3(256), 0, s, x, C1, x2, C2, gamma, beta, out, HW, groups, eps, silu);
      } else if ((int64_t)HW * (cpg / 4) <= 256 * 5)
        hipLaunchKernelGGL((gn_small_kernel<5, 4>), dim3(groups, B), dim3(256), 0, s, x, C1, x2, C2, gamma, beta, out, HW, groups, eps, silu);
      else
        hipLaunchKernelGGL((gn_small_kernel<20, 4>), dim3(groups, B), dim3(256), 0, s, x, C1, x2, C2, gamma, beta, out, HW, groups, eps, silu);
      check_launch("gn_small");
      return;
    }
  }
  const int PL = std::max(1, 256 / CV);
  static const int use_epi = getenv("SVG_GN_EPI") ? atoi(getenv("SVG_GN_EPI")) : 1;
  if (use_epi && st1 && st1->valid() && (C2 == 0 || (st2 && st2->valid()))) {
    // the producers' epilogues left per-tile column sums: no statistics pass over the tensor, one apply pass
    ctx->arena.push();
    float* stats = ctx->arena.get<float>((int64_t)B * groups * 2);
    gn_finish(ctx, *st1, C1, C2 ? st2 : nullptr, C2, stats, B, HW, groups, eps, s);
    if (SVG_LAUNCHING(ctx)) {
      char tag[96];
      snprintf(tag, sizeof(tag), "apply_B%d_HW%d_C%d", B, HW, C);
      ProfScope ps(ctx, PK_GNORM, s, 0, 2.0 * B * HW * C * 2, tag);
      const int threads = std::max((CV * PL + 63) / 64 * 64, 64);
      int nblk = std::max(1, std::min(HW / PL, std::max(HW / (PL * 16), (2048 + B - 1) / B)));
      hipLaunchKernelGGL(gn_apply_kernel, dim3(nblk, B), dim3(threads), 0, s, x, C1, x2, C2, stats, gamma, beta, out, HW, groups, 0, eps,
                         silu, CV, PL);
      check_launch("gn_apply");
    }
    ctx->arena.pop();
    return;
  }
  int nchunk = std::max(1, std::min(64, HW / (PL * 8)));
  // enough blocks to fill the chip at small batch
  while (nchunk * 2 <= 64 && (int64_t)nchunk * B < 512 && HW / (nchunk * 2) >= PL * 2) nchunk *= 2;
  ctx->arena.push();
  float* partial = ctx->arena.get<float>((int64_t)B * nchunk * groups * 2);
  if (SVG_LAUNCHING(ctx)) {
    const double bytes = (double)B * HW * C * 2;
    char tag[96];
    snprintf(tag, sizeof(tag), "stats_B%d_HW%d_C%d", B, HW, C);
    {
      ProfScope ps(ctx, PK_GNORM, s, 0, bytes, tag);
      const int threads = (CV * PL + 63) / 64 * 64;
      const size_t sh = (size_t)PL * C * 2 * sizeof(float);
      hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunk, B), dim3(std::max(threads, 64)), sh, s, x, C1, x2, C2, partial, HW,
                         groups, nchunk, CV, PL);
      check_launch("gn_stats");
    }
    {
      tag[0] = 'a'; tag[1] = 'p'; tag[2] = 'p'; tag[3] = 'l'; tag[4] = 'y';
      ProfScope ps(ctx, PK_GNORM, s, 0, 2 * bytes, tag);
      const int threads = std::max((CV * PL + 63) / 64 * 64, 64);
      // ~16 pixels per thread, at least enough blocks to fill the chip
      int nblk = std::max(1, std::min(HW / PL, std::max(HW / (PL * 16), (2048 + B - 1) / B)));
      hipLaunchKernelGGL(gn_apply_kernel, dim3(nblk, B), dim3(threads), 0, s, x, C1, x2, C2, partial, gamma, beta, out, HW,
                         groups, nchunk, eps, silu, CV, PL);
      check_launch("gn_apply");
    }
  }
  ctx->arena.pop();
}

// GroupNorm (+SiLU) whose output is MX fp8 (e4m3 [B*HW][Cp] + E8M0 [B*HW][Cp/32], Cp = C rounded up to 128): the statistics must
// come from the producers' epilogues (st1 / st2 valid); returns false when they do not, or the shape does not fit the apply kernel —
// the caller then runs groupnorm() and quant_act_mx().  tmp_stats: B * groups * 2 floats of workspace.
bool groupnorm_mx(svg_ctx* ctx, const h16* x, int C1, const h16* x2, int C2, const float* gamma, const float* beta, uint8_t* q, uint8_t* sc,
                  int B, int HW, int groups, float eps, int silu, hipStream_t s, const GnStats* st1, const GnStats* st2) {
  const int C = C1 + C2;
  static const int use_epi = getenv("SVG_GN_EPI") ? atoi(getenv("SVG_GN_EPI")) : 1;
  static const int fused = getenv("SVG_GN_MX") ? atoi(getenv("SVG_GN_MX")) : 1;
  if (!fused || !use_epi || !st1 || !st1->valid() || (C2 != 0 && !(st2 && st2->valid()))) return false;
  if (C % groups != 0 || C % 32 != 0 || C1 % 8 != 0 || groups > 64) return false;
  const int Cp = (int)align_up(C, 128), CVp = Cp / 8;
  if (CVp > 1024) return false;
  const int PL = std::max(1, 256 / CVp);
  ctx->arena.push();
  float* stats = ctx->arena.get<float>((int64_t)B * groups * 2);
  gn_finish(ctx, *st1, C1, C2 ? st2 : nullptr, C2, stats, B, HW, groups, eps, s);
  if (SVG_LAUNCHING(ctx)) {
    char tag[96];
    snprintf(tag, sizeof(tag), "apply_mx_B%d_HW%d_C%d", B, HW, C);
    ProfScope ps(ctx, PK_GNORM, s, 0, (double)B * HW * (2.0 * C + Cp + Cp / 32), tag);
    const int threads = std::max((CVp * PL + 63) / 64 * 64, 64);
    int nblk = std::max(1, std::min(HW / PL, std::max(HW / (PL * 16), (2048 + B - 1) / B)));
    hipLaunchKernelGGL(gn_apply_mx_kernel, dim3(nblk, B), dim3(threads), 0, s, x, C1, x2, C2, stats, gamma, beta, q, sc, HW, groups, silu, CVp, PL);
    check_launch("gn_apply_mx");
  }
  ctx->arena.pop();
  return true;
}

void layernorm(svg_ctx* ctx, const h16* x, const float* gamma, const float* beta, h16* out, int M, int C, float eps,
               hipStream_t s) {
  SVG_CHECK(C % 8 == 0 && C <= 2048, "layernorm: C=%d unsupported", C);
  if (!SVG_LAUNCHING(ctx)) return;
  ProfScope ps(ctx, PK_LNORM, s, 0, 4.0 * M * C);
  hipLaunchKernelGGL(layernorm_kernel, dim3(cdiv(M, 4)), dim3(256), 0, s, x, gamma, beta, out, M, C, eps);
  check_launch("layernorm");
}

void ln_stats(svg_ctx* ctx, const h16* x, float* rs, float* rm, int M, int C, float eps, hipStream_t s) {
  SVG_CHECK(C % 8 == 0 && C <= 2048, "ln_stats: C=%d unsupported", C);
  if (!SVG_LAUNCHING(ctx)) return;
  char tag[64];
  snprintf(tag, sizeof(tag), "stats_M%d_C%d", M, C);
  ProfScope ps(ctx, PK_LNORM, s, 0, 2.0 * M * C, tag);
  hipLaunchKernelGGL(ln_stats_kernel, dim3(cdiv(M, 4)), dim3(256), 0, s, x, rs, rm, M, C, eps);
  check_launch("ln_stats");
}

void ln_finish(svg_ctx* ctx, const float* part, int tiles, float* rs, float* rm, int M, int C, float eps, hipStream_t s) {
  if (!SVG_LAUNCHING(ctx)) return;
  char tag[64];
  snprintf(tag, sizeof(tag), "finish_M%d_C%d_t%d", M, C, tiles);
  ProfScope ps(ctx, PK_LNORM, s, 0, 8.0 * M * tiles, tag);
  hipLaunchKernelGGL(ln_finish_kernel, dim3(cdiv(M, 256)), dim3(256), 0, s, part, tiles, rs, rm, M, C, eps);
  check_launch("ln_finish");
}

void fold_ln_weights(float* w, const float* bias_in, const float* gamma, const float* beta, float* bias_out, int N, int K, hipStream_t s) {
  hipLaunchKernelGGL(fold_ln_kernel, dim3(N), dim3(256), 0, s, w, bias_in, gamma, beta, bias_out, K);
  check_launch("fold_ln");
}
void rowsum_h16(const h16* w, float* out, int N, int K, hipStream_t s) {
  hipLaunchKernelGGL(rowsum_h16_kernel, dim3(N), dim3(256), 0, s, w, out, K);
  check_launch("rowsum_h16");
}

void softmax_rows(svg_ctx* ctx, const float* s_in, h16* p_out, int64_t rows, int cols, int ld_in, int ld_out, float scale,
                  hipStream_t s) {
  if (!SVG_LAUNCHING(ctx)) return;
  ProfScope ps(ctx, PK_SOFTMAX, s, 0, (double)rows * cols * 10.0);
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, s, s_in, p_out, cols, ld_in, ld_out, scale);
  check_launch("softmax_rows");
}

}  // namespace SDNS
