// Fused attention for gfx950: softmax(Q K^T * scale) V, flash-style (no S x S matrix in HBM).
//
// Workgroup = 4 waves = 128*QB query rows of one (sample, head); each wave owns QB independent blocks of 32 query
// rows (QB = 2 for d <= 80): the K / V^T fragments read from LDS are reused by both blocks and the two softmax
// dependency chains (QK^T -> max -> shuffle -> exp2 -> PV) interleave, which is what hides their latency at 2 waves/SIMD.
// KV is walked in tiles of 64 keys staged through LDS (K row-major, V TRANSPOSED: [d][kv]).
//   S^T (32 kv x 32 q) = K_tile * Q^T      v_mfma_f32_32x32x16_bf16, A = K rows (LDS), B = Q^T (registers)
//     -> each lane holds 16 scores of ONE query column; its partner lane (lane^32) holds the other 16:
//        the row max is an in-register reduction plus one cross-half shuffle.
//   O^T (d x 32 q) += V^T_tile * P^T       the f32 S^T accumulator, converted to bf16, IS the B operand
//     (k order inside a step: element j of lane-half h = kv row 16s + 8(j>>2) + 4h + (j&3)); the A
//     operand reads V^T from LDS with that same k order (two 8-byte reads).
//   Row sums ride on the matrix pipe: when d is not a multiple of 32 the O^T tile has spare rows; LDS row d of
//   V^T holds ones, so O^T[row d] accumulates sum_k P (and is rescaled together with O) — no VALU adds.
//   The online-softmax rescale (exp2 + O-wide multiply) runs only when some lane's running max actually grows
//   (wave-uniform branch): bit-identical to rescaling every tile with alpha = 1.
// LDS row strides are padded (K: odd multiple of 16 B, V^T: 136 B) so fragment reads are conflict free.
#include "kernels.h"
#include "igemm_epi.h"
#include <cstdlib>

namespace SDNS {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int KVT = 64;     // keys per tile
constexpr int VROW = 136;   // bytes per V^T LDS row (64 kv * 2 B + 8)

__device__ __forceinline__ float max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// max over the lane pair (l, l^32) without the LDS crossbar: v_permlane32_swap exchanges the upper half of one register
// with the lower half of the other, leaving {lo,lo} and {hi,hi}
__device__ __forceinline__ float max_xor32(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// BC ("bias column", needs >= 3 spare columns in the padded contraction, i.e. d = 40 or 8): the softmax runs VALU-bound
// (per tile and query block: 32 exp2 + 32 fma + ~25 max + 16 cvt against 14 MFMAs), so the scale-and-shift fma is moved
// onto the matrix pipe: Q is pre-multiplied by scale*log2(e), the K pad columns d..d+2 hold 1.0 and the matching Q pad
// elements hold -m (running shift) split into three bf16 pieces (24 bits, exact in the f32 accumulate), so the MFMA
// delivers S*c - m ready for exp2.  The shift is lazy: it moves only when a score exceeds it by more than 2^TAU (first
// tile: always), which is exact for the result (softmax is shift invariant; P keeps bf16's relative precision).
template <int D, int QB, int NSTW, bool BC, bool HV = false>  // head dim (multiple of 8); 32-query blocks per wave; LDS stages wanted; HV: V^T fragments read ahead of the softmax
__global__ void __launch_bounds__(256) attn_kernel(const AttnArgs a) {
  constexpr int KS = (D + 15) / 16;           // k-steps of the QK^T contraction
  constexpr int DK = KS * 16;
  constexpr int DVT = (D + 31) / 32;          // 32-row tiles of O^T
  constexpr bool ONES = (D % 32) != 0;        // spare O^T row available for the row sums
  constexpr int LT = D / 32;                  // tile, lane-half and register holding O^T row D
  constexpr int LH = ((D % 32) >> 2) & 1;
  constexpr int LR = ((D % 32) & 3) + 4 * ((D % 32) >> 3);
  constexpr int CPR = D / 8;                  // 16-B chunks per K row
  constexpr int KROW = DK * 2 + 16;           // bytes per K LDS row
  constexpr int K_BYTES = KVT * KROW;
  constexpr int V_BYTES = DVT * 32 * VROW;
  constexpr int NLD = (KVT * CPR + 255) / 256;   // 16-B chunks per thread per tile, for K and for V^T
  // two LDS stages (a K + V^T tile is ~10-40 KB): tile t+1 is written into the other stage right after the MFMAs of
  // tile t, so the loop needs ONE barrier per tile
  constexpr int STAGE = K_BYTES + V_BYTES;
  constexpr int NST = (NSTW == 2 && 2 * STAGE <= 65536) ? 2 : 1;   // static LDS limit: d = 160 keeps one stage and two barriers
  __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  // 1-D grid, XCD-aware: workgroups are dealt to the 8 XCDs round robin (bid % 8), and the query tiles of one (sample,
  // head) all stream the same K / V^T — give each XCD a contiguous run of logical ids (query tile fastest) so that they
  // share one L2 instead of fetching K / V^T eight times (PMC: 1.25 GB fetched per launch against 0.22 GB of Q, K, V).
  int b, head, qt;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, rr = nblk & 7, xcd = bid & 7;
    const int w = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    const int nqt = (a.Sq + 128 * QB - 1) / (128 * QB);
    qt = w % nqt;
    const int bh = w / nqt;
    head = bh % a.heads;
    b = bh / a.heads;
  }
  const int q0 = qt * (128 * QB) + wid * (32 * QB);

  const h16* __restrict__ Q = a.q + (int64_t)b * a.qb + head * D;
  const h16* __restrict__ Kp = a.k + (int64_t)b * a.kb + head * D;
  const h16* __restrict__ Vt = a.vt + (int64_t)b * a.vtb + (int64_t)head * D * a.ldvt;

  // ---- Q^T fragments (B operand): lane (r,h) holds Q[q0+r][16ks + 8h .. +7] -------------------
  static_assert(!BC || (DK - D >= 3 && (D % 16) == 8), "bias-column form needs the pad columns in lane-half 1 of the last k-step");
  const float c = a.scale * 1.4426950408889634f;   // exp(x*scale) = exp2(x*c)
  h16x8 qf[QB][KS];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int q = q0 + 32 * qb + r;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int col = ks * 16 + h * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (q < a.Sq && col < D) v = *(const uint4*)(Q + (int64_t)q * a.ldq + col);
      qf[qb][ks] = *(h16x8*)&v;
      if (BC) {
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[qb][ks][j] = (h16)((float)qf[qb][ks][j] * c);
      }
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
    if (v_row[i] >= D) v_row[i] = -1;
  }
  // once: zero the K pad columns (D .. DK-1); fill V^T row D with ones (bf16 1.0 = 0x3F80)
#pragma unroll
  for (int st = 0; st < NST; ++st) {
    char* const sK = smem + st * STAGE;
    char* const sV = sK + K_BYTES;
    if (CPR * 8 < DK && tid < KVT)   // BC: columns d, d+1, d+2 = 1.0 (they multiply the three pieces of -m in Q)
      *(uint4*)(sK + tid * KROW + CPR * 16) = BC ? make_uint4(H16_ONE2, H16_ONE1, 0, 0) : make_uint4(0, 0, 0, 0);
    if (ONES && tid < 16) *(uint2*)(sV + D * VROW + tid * 8) = make_uint2(H16_ONE2, H16_ONE2);
  }

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
  auto write_lds = [&](int stage) {
    char* const sK = smem + stage * STAGE;
    char* const sV = sK + K_BYTES;
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

  f32x16 O[QB][DVT];
  float m_run[QB], l_run[QB], mc[QB];
  constexpr float TAU = 6.f;
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
    for (int i = 0; i < DVT; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) O[qb][i][j] = 0.f;
    m_run[qb] = BC ? 0.f : -1e30f; l_run[qb] = 0.f; mc[qb] = m_run[qb] * c;   // BC: m_run is the shift in the exp2 domain
  }

  const int ntiles = (a.Skv + KVT - 1) / KVT;
  load_regs(0);
  write_lds(0);
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const bool more = (t + 1 < ntiles);
    if (more) load_regs(t + 1);
    const char* const sK = smem + (t & (NST - 1)) * STAGE;
    const char* const sV = sK + K_BYTES;

    // ---- S^T = K_tile * Q^T for the two 32-key sub-tiles (K fragments shared by the QB query blocks) ------
    f32x16 S0[QB], S1[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int j = 0; j < 16; ++j) { S0[qb][j] = 0.f; S1[qb][j] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const h16x8 a0 = *(const h16x8*)(sK + r * KROW + (ks * 2 + h) * 16);
      const h16x8 a1 = *(const h16x8*)(sK + (32 + r) * KROW + (ks * 2 + h) * 16);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        S0[qb] = MFMA_32x32x16(a0, qf[qb][ks], S0[qb]);
        S1[qb] = MFMA_32x32x16(a1, qf[qb][ks], S1[qb]);
      }
    }
    // ---- HV: the V^T fragments of this tile do not depend on the softmax — issue their LDS reads now, so that the
    //      latency runs under the QK^T MFMAs and the max / exp2 phase instead of in front of every PV MFMA
    uint4 vfr[HV ? 4 * DVT : 1];
    if (HV) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int dt = 0; dt < DVT; ++dt) {
          const char* p = sV + (32 * dt + r) * VROW + (16 * i + 4 * h) * 2;
          const uint2 lo = *(const uint2*)p;
          const uint2 hi = *(const uint2*)(p + 16);
          vfr[i * DVT + dt] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- mask keys past Skv (last tile only) ----------------------------------------------------
    if (t * KVT + KVT > a.Skv) {
      const int base = t * KVT + 4 * h;
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int row = (j & 3) + 8 * (j >> 2);
          if (base + row >= a.Skv) S0[qb][j] = -INFINITY;
          if (base + 32 + row >= a.Skv) S1[qb][j] = -INFINITY;
        }
    }
    // ---- online softmax (per query column; lanes l and l^32 share a column) ----------------------
    float mx[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float v = fmaxf(S0[qb][0], S1[qb][0]);
#pragma unroll
      for (int j = 1; j < 16; ++j) v = max3(v, S0[qb][j], S1[qb][j]);
      mx[qb] = v;
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) mx[qb] = max_xor32(mx[qb]);
    if (BC) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        if (t == 0 || __any(mx[qb] > TAU)) {   // wave-uniform and rare: move the shift of the columns that need it
          const float delta = (t == 0 || mx[qb] > 0.f) ? mx[qb] : 0.f;
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          m_run[qb] += delta;
          if (!ONES) l_run[qb] *= alpha;
#pragma unroll
          for (int i = 0; i < DVT; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) O[qb][i][j] *= alpha;
#pragma unroll
          for (int j = 0; j < 16; ++j) { S0[qb][j] -= delta; S1[qb][j] -= delta; }
          // -m as three bf16 pieces in Q columns d, d+1, d+2 (held by lane-half 1 of the last k-step)
          const float v0 = -m_run[qb];
          const h16 p1 = (h16)v0;
          const float r1 = v0 - (float)p1;
          const h16 p2 = (h16)r1;
          const h16 p3 = (h16)(r1 - (float)p2);
          if (h == 1) { qf[qb][KS - 1][0] = p1; qf[qb][KS - 1][1] = p2; qf[qb][KS - 1][2] = p3; }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          S0[qb][j] = __builtin_amdgcn_exp2f(S0[qb][j]);
          S1[qb][j] = __builtin_amdgcn_exp2f(S1[qb][j]);
        }
        if (!ONES) {
          float psum = 0.f;
#pragma unroll
          for (int j = 0; j < 16; ++j) psum += S0[qb][j] + S1[qb][j];
          l_run[qb] += psum;
        }
      }
    } else {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      if (__any(mx[qb] > m_run[qb])) {   // wave-uniform: some column's running max grows -> rescale what is accumulated
        const float m_new = fmaxf(m_run[qb], mx[qb]);
        const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * c);
        m_run[qb] = m_new;
        mc[qb] = m_new * c;
        if (!ONES) l_run[qb] *= alpha;
#pragma unroll
        for (int i = 0; i < DVT; ++i)
#pragma unroll
          for (int j = 0; j < 16; ++j) O[qb][i][j] *= alpha;
      }
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        S0[qb][j] = __builtin_amdgcn_exp2f(fmaf(S0[qb][j], c, -mc[qb]));
        S1[qb][j] = __builtin_amdgcn_exp2f(fmaf(S1[qb][j], c, -mc[qb]));
      }
      if (!ONES) {
        float psum = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) psum += S0[qb][j] + S1[qb][j];
        l_run[qb] += psum;
      }
    }
    }

    // ---- O^T += V^T_tile * P^T (V^T fragments shared by the QB query blocks) -----------------------------
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        h16x8 pf[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[qb][j] = (h16)(st == 0 ? S0[qb][8 * s2 + j] : S1[qb][8 * s2 + j]);
        const int kvoff = (32 * st + 16 * s2 + 4 * h) * 2;   // bytes
#pragma unroll
        for (int dt = 0; dt < DVT; ++dt) {
          uint4 v;
          if (HV) {
            v = vfr[(2 * st + s2) * DVT + dt];
          } else {
            const char* p = sV + (32 * dt + r) * VROW + kvoff;
            const uint2 lo = *(const uint2*)p;
            const uint2 hi = *(const uint2*)(p + 16);
            v = make_uint4(lo.x, lo.y, hi.x, hi.y);
          }
#pragma unroll
          for (int qb = 0; qb < QB; ++qb)
            O[qb][dt] = MFMA_32x32x16(*(h16x8*)&v, pf[qb], O[qb][dt]);
        }
      }
    }
    if (NST == 1) __syncthreads();
    if (more) write_lds((t + 1) & (NST - 1));
    __syncthreads();
  }

  // ---- normalise and store O[q][head*D + dd] -------------------------------------------------------
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    float l_tot;
    if (ONES) {
      l_tot = __shfl(O[qb][LT][LR], r + 32 * LH);      // O^T row D = sum_k P, held by lane-half LH
    } else {
      l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32);
    }
    const float inv = 1.f / l_tot;
    const int q = q0 + 32 * qb + r;
    if (q < a.Sq) {
      h16* orow = a.out + (int64_t)b * a.ob + (int64_t)q * a.ldo + head * D;
#pragma unroll
      for (int dt = 0; dt < DVT; ++dt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dd = 32 * dt + 8 * g + 4 * h;
          if (dd < D) {
            h16x4 w;
            w[0] = (h16)(O[qb][dt][4 * g + 0] * inv);
            w[1] = (h16)(O[qb][dt][4 * g + 1] * inv);
            w[2] = (h16)(O[qb][dt][4 * g + 2] * inv);
            w[3] = (h16)(O[qb][dt][4 * g + 3] * inv);
            *(h16x4*)(orow + dd) = w;
          }
        }
      }
    }
  }
}

// ---- d = 40: K / V^T staged by LDS-DMA through a three-stage ring ---------------------------------------------------------
// The register-staged form above spends a fifth of its time on staging (ablation: no loads / LDS writes / second barrier = -20 %):
// address VALU, 6 ds_writes, a vmcnt(0) and two barriers per key tile.  Here every wave issues three `buffer_load ... lds` per tile
// (7 wave-instructions of K rows + 5 of V^T rows, 12 = 3 x 4 waves) two tiles ahead, waits with a counted vmcnt (in order: the three
// DMAs of the next tile stay in flight) and meets the other waves at ONE barrier per tile.  Two layout changes make the LDS side
// DMA-friendly (a DMA writes 64 consecutive 16-byte chunks):
//   * K rows sit in LDS in PERMUTED order: slot 16 s + 8 g + 4 h + e holds key 16 s + 8 h + 4 g + e.  Softmax does not care, and
//     the eight P values a lane feeds into one PV MFMA (slots 16 s + {4 h + e, 8 + 4 h + e}) then belong to eight CONSECUTIVE keys
//     16 s + 8 h .. + 7 — one 16-byte chunk of a natural-order V^T row instead of two 8-byte pieces.
//   * V^T rows are 128 bytes, chunk c of row r at position c ^ ((r >> 1) & 7) (the DMA lane picks its global chunk accordingly):
//     conflict-free ds_read_b128 for 16 consecutive rows at one chunk.
// Lanes that would read past Skv get an out-of-range buffer offset (zero fill).  Same arithmetic as attn_kernel<40, 2, 1, true>.
// Measured (28 x 8 x 4096^2 x 40, same box): 0.955-0.984 ms against 1.008-1.020 ms for the register-staged form.  Ablations of THIS
// kernel (0.965 ms): no v_exp -0.155, no row max -0.045, no PV MFMAs -0.34, no QK^T MFMAs -0.205, no DMA / barrier -0.165 ms, QK^T
// alone 0.54 ms — the costs add up: on this instruction mix the matrix phase and the softmax VALU do not overlap, neither inside a
// wave nor between the two unsynchronised waves of a SIMD.  A variant that runs the two query blocks of a wave half a tile apart
// (every MFMA group beside the other block's independent exp2 / max VALU in the same stream, sched_group_barrier interleave, four
// LDS stages, tail keys masked through a -1e30 pad column instead of a pass over S) was correct and NOT faster (1.02-1.06 ms);
// AGPR-form accumulators (no -amdgpu-mfma-vgpr-form) 1.26-1.65 ms.  tools/probe/probe_overlap.hip shows why the costs add: in ONE
// wave's stream 4 MFMAs + 16 v_exp take 180 cycles (140 / 160 alone), but an MFMA-phase wave beside a VALU-phase wave on the same
// SIMD takes 400 — two unsynchronised workgroups per CU are exactly that.  The consequent form — one workgroup per CU, each region
// a hand-ordered volatile-asm stream of 14 MFMAs with 4 v_exp / 2 cvt / 3 max between them and the LDS fragment reads six groups
// ahead — was also correct and slower still (1.23-1.28 ms): at one wave per SIMD nothing covers the per-tile barrier, the DMA issue
// code and the dependent max chain.
constexpr int DM_NST = 3;
__global__ void __launch_bounds__(256) attn_dma40_kernel(const AttnArgs a) {
  constexpr int D = 40, QB = 2, KS = 3, DVT = 2;
  constexpr int KROW = 112;                    // 7 chunks: 5 data, 1 pad columns (bias columns), 1 spare
  constexpr int K_BYTES = KVT * KROW;          // 7168
  constexpr int V_BYTES = 64 * 128;            // rows 0..39 data, 40 ones, 41..63 zero
  constexpr int STAGE = K_BYTES + V_BYTES;     // 15360
  constexpr int LT = D / 32, LH = ((D % 32) >> 2) & 1, LR = ((D % 32) & 3) + 4 * ((D % 32) >> 3);
  __shared__ __attribute__((aligned(16))) char smem[DM_NST * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  int b, head, qt;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, rr = nblk & 7, xcd = bid & 7;
    const int w = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    const int nqt = (a.Sq + 255) / 256;
    qt = w % nqt;
    const int bh = w / nqt;
    head = bh % a.heads;
    b = bh / a.heads;
  }
  const int q0 = qt * 256 + wid * 64;
  const h16* __restrict__ Q = a.q + (int64_t)b * a.qb + head * D;
  const h16* __restrict__ Kp = a.k + (int64_t)b * a.kb + head * D;
  const h16* __restrict__ Vt = a.vt + (int64_t)b * a.vtb + (int64_t)head * D * a.ldvt;
  const uint64_t pk = (uint64_t)Kp, pv = (uint64_t)Vt;
  const unsigned k_bytes = (unsigned)(((int64_t)a.Skv - 1) * a.ldk * 2 + D * 2);
  const unsigned v_bytes = (unsigned)(((int64_t)D - 1) * a.ldvt * 2 + ((a.Skv + 7) / 8 * 8) * 2);
  const v4i srdK = {(int)(unsigned)pk, (int)((pk >> 32) & 0xffff), (int)k_bytes, 0x00020000};
  const v4i srdV = {(int)(unsigned)pv, (int)((pv >> 32) & 0xffff), (int)v_bytes, 0x00020000};
  constexpr unsigned OOB = 0x80000000u;

  // ---- this wave's three DMA instructions of a tile: indices wid, wid + 4, wid + 8 of (K0..K6, V0..V4) -------------------------
  unsigned voff[3];            // per lane: byte offset inside the (sample, head) K or V^T view at tile 0 (OOB: lane inactive)
  int vrow_kv[3];              // K: key (within the tile) this lane's row holds; V: first key of this lane's chunk
  unsigned lds_off[3];         // wave-uniform: byte offset of the instruction's 1-KiB piece inside a stage
  bool is_k[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int idx = wid + 4 * j;
    is_k[j] = idx < 7;
    if (is_k[j]) {
      const int slot = idx * 64 + lane;                 // 448 = 64 rows x 7 chunks
      const int rs = slot / 7, c = slot - rs * 7;
      const int rho = rs & 31;
      const int kv = (rs & 32) + (rho & 16) + ((rho & 4) << 1) + ((rho & 8) >> 1) + (rho & 3);   // swap bits 2 and 3
      vrow_kv[j] = kv;
      voff[j] = c < 5 ? (unsigned)(kv * a.ldk * 2 + c * 16) : OOB;
      lds_off[j] = (unsigned)(idx * 1024);
    } else {
      const int slot = (idx - 7) * 64 + lane;           // 320 = 40 rows x 8 chunk positions
      const int row = slot >> 3, pos = slot & 7;
      const int c = pos ^ ((row >> 1) & 7);
      vrow_kv[j] = 8 * c;
      voff[j] = (unsigned)(row * a.ldvt * 2 + c * 16);
      lds_off[j] = (unsigned)(K_BYTES + (idx - 7) * 1024);
    }
  }
  const unsigned smem_base = (unsigned)(uintptr_t)smem;
  const int ntiles = (a.Skv + KVT - 1) / KVT;
  const bool ragged = (a.Skv % KVT) != 0;
  auto issue = [&](int t) {
    const unsigned st = smem_base + (unsigned)((t % DM_NST) * STAGE);
    const bool last = ragged && t == ntiles - 1;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      unsigned vo = voff[j];
      if (last && t * KVT + vrow_kv[j] >= a.Skv) vo = OOB;
      if (is_k[j]) {
        // pad-column lanes (chunks 5, 6) are masked off: their LDS bytes keep the bias-column constants
        if (voff[j] != OOB) dma16(srdK, vo, t * KVT * a.ldk * 2, __builtin_amdgcn_readfirstlane(st + lds_off[j]));
      } else {
        dma16(srdV, vo, t * KVT * 2, __builtin_amdgcn_readfirstlane(st + lds_off[j]));
      }
    }
  };

  // ---- Q^T fragments, pre-multiplied by scale * log2(e) ----------------------------------------------------------------------
  const float c = a.scale * 1.4426950408889634f;
  h16x8 qf[QB][KS];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int q = q0 + 32 * qb + r;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int col = ks * 16 + h * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (q < a.Sq && col < D) v = *(const uint4*)(Q + (int64_t)q * a.ldq + col);
      qf[qb][ks] = *(h16x8*)&v;
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[qb][ks][j] = (h16)((float)qf[qb][ks][j] * c);
    }
  }
  // ---- once per stage: K bias columns (1, 1, 1, 0...) and the spare chunk; V^T ones row 40 and zero rows 41..63 --------------
#pragma unroll
  for (int st = 0; st < DM_NST; ++st) {
    char* const sK = smem + st * STAGE;
    char* const sV = sK + K_BYTES;
    if (tid < KVT) {
      *(uint4*)(sK + tid * KROW + 80) = make_uint4(H16_ONE2, H16_ONE1, 0, 0);
      *(uint4*)(sK + tid * KROW + 96) = make_uint4(0, 0, 0, 0);
    }
    for (int i = tid; i < 24 * 8; i += 256) {
      const int row = 40 + i / 8, pos = i & 7;
      *(uint4*)(sV + row * 128 + pos * 16) = row == 40 ? make_uint4(H16_ONE2, H16_ONE2, H16_ONE2, H16_ONE2) : make_uint4(0, 0, 0, 0);
    }
  }
  __syncthreads();   // the constants are in place before any DMA lands next to them (and LDS-DMA writes are not ordered with ds_writes)

  f32x16 O[QB][DVT];
  float m_run[QB];
  constexpr float TAU = 6.f;
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
    for (int i = 0; i < DVT; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) O[qb][i][j] = 0.f;
    m_run[qb] = 0.f;
  }

  issue(0);
  if (ntiles > 1) issue(1);
  for (int t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                       // tile t is in LDS for everybody; everybody is done reading tile t - 1
    if (t + 2 < ntiles) issue(t + 2);      // into the stage tile t - 1 used
    const char* const sK = smem + (t % DM_NST) * STAGE;
    const char* const sV = sK + K_BYTES;

    f32x16 S0[QB], S1[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int j = 0; j < 16; ++j) { S0[qb][j] = 0.f; S1[qb][j] = 0.f; }
#ifndef ATTN_PRIO
#define ATTN_PRIO 0
#endif
    if (ATTN_PRIO & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const h16x8 a0 = *(const h16x8*)(sK + r * KROW + (ks * 2 + h) * 16);
      const h16x8 a1 = *(const h16x8*)(sK + (32 + r) * KROW + (ks * 2 + h) * 16);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        S0[qb] = MFMA_32x32x16(a0, qf[qb][ks], S0[qb]);
        S1[qb] = MFMA_32x32x16(a1, qf[qb][ks], S1[qb]);
      }
    }
    if (ATTN_PRIO & 1) __builtin_amdgcn_s_setprio(0);
    // V^T fragments of this tile: one 16-byte read each, issued now so that their latency runs under the softmax
    uint4 vfr[4 * DVT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int dt = 0; dt < DVT; ++dt) {
        const int row = 32 * dt + r;
        vfr[i * DVT + dt] = *(const uint4*)(sV + row * 128 + (((2 * i + h) ^ ((row >> 1) & 7)) << 4));
      }
    __builtin_amdgcn_sched_barrier(0);
    if (t * KVT + KVT > a.Skv) {           // keys past Skv: slot 8 (j >> 2) + 4 h + (j & 3) of a 32-slot block holds key 16 (j >> 3) + 8 h + 4 ((j >> 2) & 1) + (j & 3)
      const int base = t * KVT + 8 * h;
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int kv = 16 * (j >> 3) + 4 * ((j >> 2) & 1) + (j & 3);
          if (base + kv >= a.Skv) S0[qb][j] = -INFINITY;
          if (base + 32 + kv >= a.Skv) S1[qb][j] = -INFINITY;
        }
    }
    if (ATTN_PRIO & 4) __builtin_amdgcn_s_setprio(1);
    float mx[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float v = fmaxf(S0[qb][0], S1[qb][0]);
#pragma unroll
      for (int j = 1; j < 16; ++j) v = max3(v, S0[qb][j], S1[qb][j]);
      mx[qb] = max_xor32(v);
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      if (t == 0 || __any(mx[qb] > TAU)) {
        const float delta = (t == 0 || mx[qb] > 0.f) ? mx[qb] : 0.f;
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        m_run[qb] += delta;
#pragma unroll
        for (int i = 0; i < DVT; ++i)
#pragma unroll
          for (int j = 0; j < 16; ++j) O[qb][i][j] *= alpha;
#pragma unroll
        for (int j = 0; j < 16; ++j) { S0[qb][j] -= delta; S1[qb][j] -= delta; }
        const float v0 = -m_run[qb];
        const h16 p1 = (h16)v0;
        const float r1 = v0 - (float)p1;
        const h16 p2 = (h16)r1;
        const h16 p3 = (h16)(r1 - (float)p2);
        if (h == 1) { qf[qb][KS - 1][0] = p1; qf[qb][KS - 1][1] = p2; qf[qb][KS - 1][2] = p3; }
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        S0[qb][j] = __builtin_amdgcn_exp2f(S0[qb][j]);
        S1[qb][j] = __builtin_amdgcn_exp2f(S1[qb][j]);
      }
    }
    if (ATTN_PRIO & 4) __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        h16x8 pf[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[qb][j] = (h16)(st == 0 ? S0[qb][8 * s2 + j] : S1[qb][8 * s2 + j]);
        if (ATTN_PRIO & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int dt = 0; dt < DVT; ++dt) {
          const uint4 v = vfr[(2 * st + s2) * DVT + dt];
#pragma unroll
          for (int qb = 0; qb < QB; ++qb)
            O[qb][dt] = MFMA_32x32x16(*(const h16x8*)&v, pf[qb], O[qb][dt]);
        }
        if (ATTN_PRIO & 2) __builtin_amdgcn_s_setprio(0);
      }
  }

#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float l_tot = __shfl(O[qb][LT][LR], r + 32 * LH);
    const float inv = 1.f / l_tot;
    const int q = q0 + 32 * qb + r;
    if (q < a.Sq) {
      h16* orow = a.out + (int64_t)b * a.ob + (int64_t)q * a.ldo + head * D;
#pragma unroll
      for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dd = 32 * dt + 8 * g + 4 * h;
          if (dd < D) {
            h16x4 w;
            w[0] = (h16)(O[qb][dt][4 * g + 0] * inv);
            w[1] = (h16)(O[qb][dt][4 * g + 1] * inv);
            w[2] = (h16)(O[qb][dt][4 * g + 2] * inv);
            w[3] = (h16)(O[qb][dt][4 * g + 3] * inv);
            *(h16x4*)(orow + dd) = w;
          }
        }
    }
  }
}

template <int D>
void launch(const AttnArgs& a, hipStream_t s) {
  // Two query blocks per wave (QB = 2) when the registers allow two waves per SIMD (d <= 64: 222 VGPRs):
  // 0.67 vs 0.73 ms at 16 x 8 x 4096^2 x 40.  SVG_ATTN_QB=1 forces the single-block form.
  static const int nst_env = getenv("SVG_ATTN_NST") ? atoi(getenv("SVG_ATTN_NST")) : 1;   // same-box A/B: one stage + two barriers is 1-2 % faster than two stages + one barrier
  static const int qb_env = getenv("SVG_ATTN_QB") ? atoi(getenv("SVG_ATTN_QB")) : 2;
  static const int bc_env = getenv("SVG_ATTN_BC") ? atoi(getenv("SVG_ATTN_BC")) : 1;
  static const int hv_env = getenv("SVG_ATTN_HV") ? atoi(getenv("SVG_ATTN_HV")) : 1;   // same-box A/B at 28 x 8 x 4096^2 x 40: 1.006 vs 1.026 ms
  constexpr int QB = (D <= 64) ? 2 : 1;
  constexpr bool CAN_BC = (D % 16) == 8;        // three spare pad columns in lane-half 1 of the last k-step (d = 8, 40)
  static const int dma_env = getenv("SVG_ATTN_DMA") ? atoi(getenv("SVG_ATTN_DMA")) : 1;
  if (D == 40 && dma_env && qb_env == 2 && bc_env && a.Sq >= 512) {
    hipLaunchKernelGGL(attn_dma40_kernel, dim3(cdiv(a.Sq, 256) * a.heads * a.B), dim3(256), 0, s, a);
    return;
  }
  if (QB == 2 && qb_env == 2 && a.Sq >= 512) {
    dim3 grid(cdiv(a.Sq, 256) * a.heads * a.B);
    if (CAN_BC && bc_env && hv_env) hipLaunchKernelGGL((attn_kernel<D, QB, 1, CAN_BC, true>), grid, dim3(256), 0, s, a);
    else if (CAN_BC && bc_env) hipLaunchKernelGGL((attn_kernel<D, QB, 1, CAN_BC>), grid, dim3(256), 0, s, a);
    else if (nst_env == 2) hipLaunchKernelGGL((attn_kernel<D, QB, 2, false>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((attn_kernel<D, QB, 1, false>), grid, dim3(256), 0, s, a);
  } else {
    dim3 grid(cdiv(a.Sq, 128) * a.heads * a.B);
    if (CAN_BC && bc_env) hipLaunchKernelGGL((attn_kernel<D, 1, 1, CAN_BC>), grid, dim3(256), 0, s, a);
    else if (nst_env == 2) hipLaunchKernelGGL((attn_kernel<D, 1, 2, false>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((attn_kernel<D, 1, 1, false>), grid, dim3(256), 0, s, a);
  }
}

}  // namespace

void attention(svg_ctx* ctx, const AttnArgs& a, hipStream_t s) {
  SVG_CHECK(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldvt % 8 == 0 && a.ldo % 4 == 0, "attention: strides must be 16-byte aligned");
  SVG_CHECK(a.ldvt >= (a.Skv + 7) / 8 * 8, "attention: V^T rows must be padded to a multiple of 8 keys");
  SVG_CHECK(a.Skv > 0 && a.Sq > 0, "attention: empty");
  if (!SVG_LAUNCHING(ctx)) {
    switch (a.d) { case 8: case 16: case 32: case 40: case 64: case 80: case 160: return; default: break; }
    SVG_CHECK(false, "attention: head dim %d unsupported (instantiated: 8, 16, 32, 40, 64, 80, 160)", a.d);
  }
  char tag[96];
  snprintf(tag, sizeof(tag), "B%d_h%d_Sq%d_Skv%d_d%d", a.B, a.heads, a.Sq, a.Skv, a.d);
  ProfScope ps(ctx, PK_ATTN, s, 4.0 * a.B * a.heads * (double)a.Sq * a.Skv * a.d,
               2.0 * a.B * a.heads * ((double)a.Sq * a.d * 2 + (double)a.Skv * a.d * 2), tag);
  switch (a.d) {
    case 8: launch<8>(a, s); break;
    case 16: launch<16>(a, s); break;
    case 32: launch<32>(a, s); break;
    case 40: launch<40>(a, s); break;
    case 64: launch<64>(a, s); break;
    case 80: launch<80>(a, s); break;
    case 160: launch<160>(a, s); break;
    default: SVG_CHECK(false, "attention: head dim %d unsupported (instantiated: 8, 16, 32, 40, 64, 80, 160)", a.d);
  }
  check_launch("attention");
}

}  // namespace SDNS
