// Fused GEGLU feed-forward of a BasicTransformerBlock for gfx950 (C = 320, the 64 x 64 level of the SD UNet):
//
//     out = ff.net.2( GEGLU( LayerNorm3(x) W1^T + b1 ) ) + b2 + residual          (diffusers FeedForward)
//
// As two GEMMs the M x 4C intermediate (293 MB at 28 clips) is written by one kernel and read back by the next.  Here a
// workgroup owns 128 rows and walks the 1280 hidden units in chunks of 64: GEMM1 (K = 320) produces the chunk's h | gate
// tiles, GEGLU runs on the accumulator registers, and the bf16 result IS the B operand of GEMM2 (no LDS round trip: the k
// order inside a 32-wide step is permuted, W2 is packed with the same permutation), which accumulates the 128 x 320 output.
//   8 waves x 16 rows; the wave's 16 x 320 slice of x stays in registers for the whole kernel (40 VGPRs, MFMA B operand);
//   W1 / W2 stream through an 8-slab LDS ring (16 KiB slabs = 128 rows x 64 k, asm LDS-DMA, counted vmcnt, 4 slabs in flight);
//   LayerNorm is folded as in gemm.hip (W1' = W1 diag(gamma), row statistics applied to the accumulator);
//   v_mfma_f32_16x16x32_bf16 with the weight fragment as the A operand (4 consecutive output columns per lane).
// Per hidden chunk and wave: 80 + 40 MFMAs against 120 ds_read_b128 — LDS bandwidth and the matrix pipe saturate together.
#include "igemm_epi.h"
#include <cstdlib>
#include <vector>

namespace SDNS {

namespace {

constexpr int FC = 320;                 // channels
constexpr int FH = 1280;                // hidden units (4C)
constexpr int F_SLAB = 16384;           // 128 rows x 128 B
constexpr int F_RING = 8;
#ifndef FF_DEPTH
#define FF_DEPTH 4
#endif
constexpr int F_DEPTH = FF_DEPTH;       // slabs in flight
constexpr int F_KS = FC / 32;           // k-steps of GEMM1
constexpr int F_NCH = FH / 64;          // hidden chunks
constexpr int F_SPC = 5 + 3;            // slabs per chunk: 5 of W1 (k slabs), 3 of W2 (row blocks of 128 covering 320 rows)
constexpr int F_OFF_S1 = F_RING * F_SLAB;              // s1 (2560 f32) then b1 (2560 f32)
constexpr int F_LDS = F_OFF_S1 + 2 * 2 * FH * 4;

struct FfArgs {
  const h16* X; int ldx;               // pre-LayerNorm input rows
  const h16* W1; const float* b1; const float* s1;   // packed GEGLU weights [2*FH][FC] (h / gate tiles of 16 rows alternate), folded bias, row sums
  const float* rs; const float* rm;     // LayerNorm row statistics: rstd, rstd * mean; both null: computed here from the rows in registers
  const h16* W2p; const float* b2;     // [FC][FH], k permuted inside 32-blocks (pack_ff2_perm)
  const h16* residual; int ldr;
  h16* out; int ldo;
  int M;
};

__global__ void __launch_bounds__(512, 2) ff_fused_kernel(const FfArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wid);
  const int l15 = lane & 15, lq = lane >> 4;
  const int m0 = blockIdx.x * 128;
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  constexpr unsigned INVALID = 0x80000000u;

  const uint64_t p1 = (uint64_t)a.W1, p2 = (uint64_t)a.W2p;
  const v4i srd1 = {(int)(unsigned)p1, (int)((p1 >> 32) & 0xffff), (int)(2u * FH * FC * 2u), 0x00020000};
  const v4i srd2 = {(int)(unsigned)p2, (int)((p2 >> 32) & 0xffff), (int)((unsigned)FC * FH * 2u), 0x00020000};

  // ---- ring DMA: slab q (global counter) = chunk c = q / 8, piece s = q % 8: s < 5 -> W1 rows [128c, +128) x k [64s, +64);
  //      s >= 5 -> W2p rows [128(s-5), +128) (rows >= 320 read as zeros) x k [64c, +64).  Two 1-KiB pieces per wave and slab.
  const int prow = lane >> 3;                              // row within an 8-row piece
  auto dma_slab = [&](int q) {
    const int c = q >> 3, s = q & 7;
    const unsigned dst = lds0 + (q & (F_RING - 1)) * F_SLAB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int blk = i * 8 + wave_u;                      // 8-row block of the slab
      const int row = blk * 8 + prow;
      const int ch = (lane & 7) ^ (row & 7);               // swizzle on the source chunk
      if (s < 5) {
        const unsigned voff = (unsigned)((128 * c + row) * FC + ch * 8) * 2u;
        dma16(srd1, voff, s * 128, dst + blk * 1024);
      } else {
        const int r = 128 * (s - 5) + row;
        const unsigned voff = r < FC ? (unsigned)(r * FH + ch * 8) * 2u : INVALID;
        dma16(srd2, voff, c * 128, dst + blk * 1024);
      }
    }
  };
  constexpr int NQ = F_NCH * F_SPC;
#pragma unroll
  for (int q = 0; q < F_DEPTH; ++q) dma_slab(q);

  // ---- s1 / b1 to LDS, the wave's x rows and LayerNorm statistics to registers
  {
    float* ss = (float*)(smem + F_OFF_S1);
    for (int i = tid; i < 2 * FH; i += 512) { ss[i] = a.s1[i]; ss[2 * FH + i] = a.b1[i]; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the ds_writes have left before this wave's first barrier
  }
  const int m = m0 + wave_u * 16 + l15;
  const bool m_ok = m < a.M;
  h16x8 xf[F_KS];
#pragma unroll
  for (int ks = 0; ks < F_KS; ++ks) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (m_ok) v = *(const uint4*)(a.X + (int64_t)m * a.ldx + ks * 32 + lq * 8);
    xf[ks] = *(h16x8*)&v;
  }
  float rs, rm;
  if (a.rs) {
    rs = m_ok ? a.rs[m] : 0.f; rm = m_ok ? a.rm[m] : 0.f;
  } else {
    // the four lanes l15 + 16 * lq hold the whole row (80 values each): two-pass statistics like ln_stats_kernel (eps 1e-5), without
    // the extra pass over the tensor
    float sum = 0.f;
#pragma unroll
    for (int ks = 0; ks < F_KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += (float)xf[ks][j];
    sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.f / FC);
    float sq = 0.f;
#pragma unroll
    for (int ks = 0; ks < F_KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float dlt = (float)xf[ks][j] - mean; sq += dlt * dlt; }
    sq += __shfl_xor(sq, 16); sq += __shfl_xor(sq, 32);
    rs = m_ok ? rsqrtf(sq * (1.f / FC) + 1e-5f) : 0.f;
    rm = rs * mean;
  }

  f32x4 acc2[FC / 16];
#pragma unroll
  for (int t = 0; t < FC / 16; ++t) acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frag0 = l15 * 128;                             // row l15 of a 16-row tile; chunk (kk*4 + lq) ^ (l15 & 7)
  int fsw[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) fsw[kk] = frag0 + (((kk * 4 + lq) ^ (l15 & 7)) << 4);
  const float* const sS1 = (const float*)(smem + F_OFF_S1);
  const float* const sB1 = sS1 + 2 * FH;

  // ---- main loop.  All eight waves walk the slabs in lock step (one barrier per slab); the two waves of a SIMD alternate by
  // themselves: while one issues its 8 MFMAs of a k half the other reads its 8 fragments.  Measured alternatives, all SLOWER
  // than this form (0.448 ms for 114688 rows, kbench): a finer software pipeline (fragments of the next quarter step read
  // before the MFMAs of the current one, pinned with sched_barrier) 0.495-0.533 ms; six slabs in flight instead of four
  // 0.473 ms; the ping-pong schedule of conv_halo.hip (waves 4-7 one barrier behind, reads and MFMAs in separate phases)
  // 0.483 ms.  Per slab a wave issues 2 LDS-DMA pieces + 16 fragment reads for only 16 MFMAs (one fragment read per MFMA:
  // 16 rows per wave): the in-order issue stream of each wave, not a single resource, sets the pace.  Four waves x 32 rows (every
  // fragment feeding two MFMAs, one wave per SIMD): 304 accumulator / operand registers, i.e. 142 of them parked in AGPRs with copies
  // around the MFMAs — 0.132-0.136 ms against 0.125-0.126 ms for 32768 rows.
  int q = 0;
#ifndef FF_PAIR
#define FF_PAIR 0
#endif
#ifndef FF_ABL            // timing ablations (results are garbage): 1 no ring DMA after the prologue, 2 no MFMAs, 3 no fragment reads
#define FF_ABL 0
#endif
  auto step_begin = [&](int sidx) {                        // retire slab q, barrier, refill the slot F_DEPTH ahead
    if (FF_PAIR) {                                         // experiment: one barrier per TWO slabs (sidx = slab index inside the chunk)
      if (sidx & 1) return;
      if (q + F_DEPTH <= NQ) wait_vm(2 * (F_DEPTH - 2));
      else wait_vm(NQ - q - 2 > 0 ? (NQ - q - 2) * 2 : 0);
      bar();
      if (q + F_DEPTH < NQ && FF_ABL != 1) { dma_slab(q + F_DEPTH); dma_slab(q + F_DEPTH + 1); }
      return;
    }
    if (q + F_DEPTH - 1 < NQ) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // 2 * (F_DEPTH - 1): the steady state
    else wait_vm((NQ - 1 - q) * 2);
    bar();
    if (q + F_DEPTH < NQ && FF_ABL != 1) dma_slab(q + F_DEPTH);
  };

  for (int c = 0; c < F_NCH; ++c) {
    // ---- GEMM1: 16 rows x 128 packed columns (h0 g0 h1 g1 h2 g2 h3 g3), K = 320 in 5 slabs
    f32x4 acc1[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      step_begin(s);
      const char* sl = smem + (q & (F_RING - 1)) * F_SLAB;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        h16x8 wf[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) { if (FF_ABL != 3) wf[t] = *(const h16x8*)(sl + t * 2048 + fsw[kk]); else asm volatile("" : "=v"(wf[t])); }
#pragma unroll
        for (int t = 0; t < 8; ++t) { if (FF_ABL != 2) acc1[t] = MFMA_16x16x32(wf[t], xf[s * 2 + kk], acc1[t]); else asm volatile("" :: "v"(wf[t])); }
      }
      ++q;
    }
    // ---- folded LayerNorm + bias, GEGLU: p = h * gelu(gate) -> the B operand of GEMM2
    h16x8 pf[2];
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
      const int n = 128 * c + 32 * pr + 4 * lq;            // packed column of the h tile; the gate tile follows 16 later
      const f32x4 sh = *(const f32x4*)(sS1 + n), sg = *(const f32x4*)(sS1 + n + 16);
      const f32x4 bh = *(const f32x4*)(sB1 + n), bg = *(const f32x4*)(sB1 + n + 16);
      const f32x4 h = acc1[2 * pr] * rs - sh * rm + bh;
      const f32x4 gt = acc1[2 * pr + 1] * rs - sg * rm + bg;
      const f32x4 pv = h * gelu_erf4(gt);
#pragma unroll
      for (int e = 0; e < 4; ++e) pf[pr >> 1][(pr & 1) * 4 + e] = (h16)pv[e];
    }
    // ---- GEMM2: acc2[320 columns] += P (16 x 64) W2p[:, chunk]^T, 3 slabs of 128 W2 rows
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      step_begin(5 + s);
      const char* sl = smem + (q & (F_RING - 1)) * F_SLAB;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          if (s * 8 + t < FC / 16) {
            h16x8 wf;
            if (FF_ABL != 3) wf = *(const h16x8*)(sl + t * 2048 + fsw[kk]); else asm volatile("" : "=v"(wf));
            if (FF_ABL != 2) acc2[s * 8 + t] = MFMA_16x16x32(wf, pf[kk], acc2[s * 8 + t]); else asm volatile("" :: "v"(wf));
          }
        }
      }
      ++q;
    }
  }

  // ---- epilogue: + b2 + residual, h16 store (4 consecutive columns per lane)
  if (m_ok) {
#pragma unroll
    for (int t = 0; t < FC / 16; ++t) {
      const int n = t * 16 + lq * 4;
      f32x4 v = acc2[t] + *(const f32x4*)(a.b2 + n);
      if (a.residual) {
        const h16x4 r = *(const h16x4*)(a.residual + (int64_t)m * a.ldr + n);
        v[0] += (float)r[0]; v[1] += (float)r[1]; v[2] += (float)r[2]; v[3] += (float)r[3];
      }
      *(h16x4*)(a.out + (int64_t)m * a.ldo + n) = to_h16x4(v);
    }
  }
}

// ---- the PAIRED form: two waves share 32 rows -----------------------------------------------------------------------------------
// Ablations of ff_fused_kernel on MI355X (114688 rows, 0.455 ms): no MFMAs 0.274, no fragment reads 0.321, no ring DMA 0.419, a GELU
// at 8 instead of 18 issue slots -3 %, half the barriers 0 %, a deeper ring 0 %.  At 16 rows per wave a weight fragment feeds ONE
// MFMA (16 FLOP per LDS byte: the LDS pipe and the matrix pipe both have to run at 100 %), and the two take turns instead of
// overlapping.  32 rows per wave with the whole 320-wide output needs 304 registers per lane (one wave per SIMD, accumulators in
// AGPRs: built, 0.555 ms — the compiler shuttles the x fragments through AGPRs and scratch).  Here the waves of a pair (2p, 2p + 1)
// own the same 32 rows and split the COLUMNS: of a hidden chunk's 128 packed GEMM1 columns each takes 64 (so each runs GEGLU on 32
// hidden units = one k-step of GEMM2), they swap their P tiles through 2 KiB of LDS, and each accumulates 160 of the 320 output
// columns.  Every fragment read feeds two MFMAs (60 reads + 4 swap accesses per 120 MFMAs), 80 + 32 + 80 accumulator / operand
// registers per lane, still two waves per SIMD.  W2's rows are dealt to the slabs so that each slab carries 64 rows of either half
// (slab j: rows 64 j .. of columns 0-159 in LDS rows 0-63, of columns 160-319 in LDS rows 64-127) — a DMA source map, no repacking.
constexpr int P_RING = 7;                // 7 x 16 KiB slabs + s1/b1 (20 KiB) + the swap buffer (16 KiB) = 148 KiB
constexpr int P_OFF_S1 = P_RING * F_SLAB;
constexpr int P_OFF_EX = P_OFF_S1 + 2 * 2 * FH * 4;
constexpr int P_LDS = P_OFF_EX + 8 * 2 * 64 * 16;

// In-kernel stamps (diagnostic build -DFF_STAMP; VERDICT r03 #1d): per wave, summed over the kernel, the cycles of ring wait (counted
// vmcnt) / barrier / DMA issue / GEMM1 slab (fragment reads + 16 MFMAs) / GEGLU + swap write / GEMM2 slab — printed by the launcher.
#ifdef FF_STAMP
#define FSTAMP(v) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define FSTAMP(v) do { } while (0)
#endif

__global__ void __launch_bounds__(512, 2) ff_pair_kernel(const FfArgs a, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wid);
  const int pair_u = wave_u >> 1, half_u = wave_u & 1;
  const int l15 = lane & 15, lq = lane >> 4;
  const int m0 = blockIdx.x * 128;
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  constexpr unsigned INVALID = 0x80000000u;

  const uint64_t p1 = (uint64_t)a.W1, p2 = (uint64_t)a.W2p;
  const v4i srd1 = {(int)(unsigned)p1, (int)((p1 >> 32) & 0xffff), (int)(2u * FH * FC * 2u), 0x00020000};
  const v4i srd2 = {(int)(unsigned)p2, (int)((p2 >> 32) & 0xffff), (int)((unsigned)FC * FH * 2u), 0x00020000};

  const int prow = lane >> 3;
  auto dma_slab = [&](int q, int slot) {                   // slab q = (chunk q / 8, piece q % 8) into ring slot `slot`
    const int c = q >> 3, s = q & 7;
    const unsigned dst = lds0 + slot * F_SLAB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int blk = i * 8 + wave_u;
      const int row = blk * 8 + prow;                      // LDS row of the slab
      const int ch = (lane & 7) ^ (row & 7);
      if (s < 5) {
        const unsigned voff = (unsigned)((128 * c + row) * FC + ch * 8) * 2u;
        dma16(srd1, voff, s * 128, dst + blk * 1024);
      } else {
        const int rr = 64 * (s - 5) + (row & 63);          // MFMA operand row inside the column half: tile rr >> 4, row rr & 15
        // the output column that operand row stands for: the quads of a tile pair are 8 consecutive columns per lane (16-byte epilogue, as
        // gemm_ws.hip / igemm_epi.h: epi_perm_col) — ten tiles per half, all paired
        const int cperm = 32 * (rr >> 5) + 8 * ((rr & 15) >> 2) + 4 * ((rr >> 4) & 1) + (rr & 3);
        const int r = (row >> 6) * 160 + cperm;
        const unsigned voff = rr < 160 ? (unsigned)(r * FH + ch * 8) * 2u : INVALID;
        dma16(srd2, voff, c * 128, dst + blk * 1024);
      }
    }
  };
  constexpr int NQ = F_NCH * F_SPC;
#pragma unroll
  for (int q = 0; q < F_DEPTH; ++q) dma_slab(q, q);

  {
    float* ss = (float*)(smem + P_OFF_S1);
    for (int i = tid; i < 2 * FH; i += 512) { ss[i] = a.s1[i]; ss[2 * FH + i] = a.b1[i]; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  int mrow[2]; bool m_ok[2];
  h16x8 xf[2][F_KS];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    mrow[rt] = m0 + pair_u * 32 + rt * 16 + l15;
    m_ok[rt] = mrow[rt] < a.M;
#pragma unroll
    for (int ks = 0; ks < F_KS; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (m_ok[rt]) v = *(const uint4*)(a.X + (int64_t)mrow[rt] * a.ldx + ks * 32 + lq * 8);
      xf[rt][ks] = *(h16x8*)&v;
    }
  }
  float rs[2], rm[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    if (a.rs) {
      rs[rt] = m_ok[rt] ? a.rs[mrow[rt]] : 0.f; rm[rt] = m_ok[rt] ? a.rm[mrow[rt]] : 0.f;
    } else {
      float sum = 0.f;
#pragma unroll
      for (int ks = 0; ks < F_KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += (float)xf[rt][ks][j];
      sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
      const float mean = sum * (1.f / FC);
      float sq = 0.f;
#pragma unroll
      for (int ks = 0; ks < F_KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float dlt = (float)xf[rt][ks][j] - mean; sq += dlt * dlt; }
      sq += __shfl_xor(sq, 16); sq += __shfl_xor(sq, 32);
      rs[rt] = m_ok[rt] ? rsqrtf(sq * (1.f / FC) + 1e-5f) : 0.f;
      rm[rt] = rs[rt] * mean;
    }
  }

  f32x4 acc2[2][10];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int t = 0; t < 10; ++t) acc2[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment address inside a slab: row tile (4 half + j) of 16 rows, chunk (kk * 4 + lq) ^ (l15 & 7)
  int fsw[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) fsw[kk] = (half_u * 64 + l15) * 128 + (((kk * 4 + lq) ^ (l15 & 7)) << 4);
  const float* const sS1 = (const float*)(smem + P_OFF_S1);
  const float* const sB1 = sS1 + 2 * FH;
  char* const ex_own = smem + P_OFF_EX + ((wave_u * 2) * 64 + lane) * 16;            // [wave][rt][lane] 16 B
  const char* const ex_other = smem + P_OFF_EX + (((wave_u ^ 1) * 2) * 64 + lane) * 16;

  int q = 0, slot = 0, slot_in = F_DEPTH;                  // slot of slab q; slot the next DMA goes to
  unsigned long long t_wait = 0, t_bar = 0, t_dma = 0, t_g1 = 0, t_gelu = 0, t_g2 = 0, t_all0 = 0, t_all1 = 0, t_epi = 0;
  // A slab step = wait for the slab (counted vmcnt) + barrier, then the refill DMA of the slot F_DEPTH ahead.  The two are separate
  // because the LATE waves (below) put matrix work between them.
  auto step_sync = [&]() {
    unsigned long long u0 = 0, u1 = 0, u2 = 0;
    FSTAMP(u0);
    if (q + F_DEPTH - 1 < NQ) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else wait_vm((NQ - 1 - q) * 2);
    FSTAMP(u1);
    bar();
    FSTAMP(u2);
    t_wait += u1 - u0; t_bar += u2 - u1;
  };
  auto step_dma = [&]() {
    unsigned long long u2 = 0, u3 = 0;
    FSTAMP(u2);
    if (q + F_DEPTH < NQ) dma_slab(q + F_DEPTH, slot_in);
    FSTAMP(u3);
    t_dma += u3 - u2;
    slot_in = slot_in + 1 == P_RING ? 0 : slot_in + 1;
  };
  auto step_end = [&]() { ++q; slot = slot + 1 == P_RING ? 0 : slot + 1; };
  // Stagger (round 5; MI355X_MICROARCH.md "two waves that run the SAME program with one barrier per block"): the eight waves walk the
  // slabs in lock step, and the two waves of a SIMD (w and w + 4) used to reach their fragment reads, their MFMAs, the DMA issue and the
  // GEGLU at the same time — matrix pipe idle while both read, contended while both multiply (stamps of round 4: 566 cycles per slab for
  // 256 matrix cycles, barrier waits 18 %).  Waves 4-7 (LATE) now run half a slab behind: they carry the fragments of a slab's second k
  // half across the barrier and multiply them FIRST in the next interval, while their SIMD partner issues its DMA and waits for its
  // reads; then they issue their own DMA and read while the partner multiplies.  Same MFMAs into the same accumulators in the same
  // order (bit-identical results), no extra LDS, one more fragment set live across the barrier in the late waves only.
#ifndef FF_STAGGER
#define FF_STAGGER 1
#endif
  const bool late = FF_STAGGER && wave_u >= 4;

  FSTAMP(t_all0);
  for (int c = 0; c < F_NCH; ++c) {
    // ---- GEMM1: 32 rows x 64 packed columns (h g h g of this half), K = 320 in 5 slabs
    f32x4 acc1[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc1[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    h16x8 wpend[4];                                        // late waves: the k-half-1 fragments of the previous slab
    auto rd1 = [&](const char* sl, int kk, h16x8 (&wf)[4]) {
#pragma unroll
      for (int t = 0; t < 4; ++t) wf[t] = *(const h16x8*)(sl + t * 2048 + fsw[kk]);
    };
    auto mm1 = [&](int ks, const h16x8 (&wf)[4]) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) acc1[rt][t] = MFMA_16x16x32(wf[t], xf[rt][ks], acc1[rt][t]);
    };
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      step_sync();
      unsigned long long v0 = 0, v1 = 0;
      FSTAMP(v0);
      const char* sl = smem + slot * F_SLAB;
      // one instruction stream for both kinds of wave (two copies of the slab body cost 113 spilled registers): the k-half-1 MFMAs sit
      // behind a wave-uniform branch, before the DMA for the late waves (previous slab's fragments) and after the reads for the early ones
      if (late && s > 0) mm1((s - 1) * 2 + 1, wpend);
      step_dma();
      {
        h16x8 wf[4];
        rd1(sl, 0, wf);
        mm1(s * 2, wf);
      }
      rd1(sl, 1, wpend);
      if (!late || s == 4) mm1(s * 2 + 1, wpend);              // (s == 4: the chunk's GEMM1 is complete before the GEGLU)
#ifdef FF_STAMP
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(acc1[rt][t]));
#endif
      FSTAMP(v1);
      t_g1 += v1 - v0;
      step_end();
    }
    unsigned long long w0 = 0, w1 = 0;
    FSTAMP(w0);
    // ---- folded LayerNorm + bias, GEGLU on this half's 32 hidden units: the B operand of GEMM2's k-step `half`
    h16x8 pown[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const int n = 128 * c + 64 * half_u + 32 * pr + 4 * lq;
        const f32x4 sh = *(const f32x4*)(sS1 + n), sg = *(const f32x4*)(sS1 + n + 16);
        const f32x4 bh = *(const f32x4*)(sB1 + n), bg = *(const f32x4*)(sB1 + n + 16);
        const f32x4 h = acc1[rt][2 * pr] * rs[rt] - sh * rm[rt] + bh;
        const f32x4 gt = acc1[rt][2 * pr + 1] * rs[rt] - sg * rm[rt] + bg;
        const f32x4 pv = h * gelu_erf4(gt);
#pragma unroll
        for (int e = 0; e < 4; ++e) pown[rt][pr * 4 + e] = (h16)pv[e];
      }
      *(h16x8*)(ex_own + rt * 1024) = pown[rt];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the swap writes have landed before the barrier of the next step
    FSTAMP(w1);
    t_gelu += w1 - w0;
    // ---- GEMM2: acc2[160 columns of this half] += P (32 x 64) W2p[:, chunk]^T, 3 slabs of 64 + 64 W2 rows
    h16x8 pf[2][2];
    auto rd2 = [&](const char* sl, int kk, h16x8 (&wf)[4]) {
#pragma unroll
      for (int t = 0; t < 4; ++t) wf[t] = *(const h16x8*)(sl + t * 2048 + fsw[kk]);
    };
    auto mm2 = [&](int s, int kk, const h16x8 (&wf)[4]) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (s * 4 + t < 10) {
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) acc2[rt][s * 4 + t] = MFMA_16x16x32(wf[t], pf[rt][kk], acc2[rt][s * 4 + t]);
        }
      }
    };
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      step_sync();
      unsigned long long v0 = 0, v1 = 0;
      FSTAMP(v0);
      if (s == 0) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const h16x8 po = *(const h16x8*)(ex_other + rt * 1024);
          pf[rt][0] = half_u ? po : pown[rt];
          pf[rt][1] = half_u ? pown[rt] : po;
        }
      }
      const char* sl = smem + slot * F_SLAB;
      if (late && s > 0) mm2(s - 1, 1, wpend);
      step_dma();
      {
        h16x8 wf[4];
        rd2(sl, 0, wf);
        mm2(s, 0, wf);
      }
      rd2(sl, 1, wpend);
      if (!late || s == 2) mm2(s, 1, wpend);
#ifdef FF_STAMP
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int t = 0; t < 4; ++t) if (s * 4 + t < 10) asm volatile("" : "+v"(acc2[rt][s * 4 + t]));
#endif
      FSTAMP(v1);
      t_g2 += v1 - v0;
      step_end();
    }
  }
  FSTAMP(t_all1);

  // ---- epilogue: + b2 + residual, h16 store — 8 consecutive columns per lane and tile pair (the W2 rows were dealt accordingly): one
  // 16-byte residual load and one 16-byte store instead of two 8-byte ones (the epilogue was 10 % of the kernel's cycles)
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    if (!m_ok[rt]) continue;
    const int m = mrow[rt];
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      const int n = half_u * 160 + 32 * p + 8 * lq;
      f32x4 v0 = acc2[rt][2 * p] + *(const f32x4*)(a.b2 + n);
      f32x4 v1 = acc2[rt][2 * p + 1] + *(const f32x4*)(a.b2 + n + 4);
      if (a.residual) {
        const h16x8 r = *(const h16x8*)(a.residual + (int64_t)m * a.ldr + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v0[e] += (float)r[e]; v1[e] += (float)r[4 + e]; }
      }
      h16x8 w;
      const h16x4 w0 = to_h16x4(v0), w1 = to_h16x4(v1);
#pragma unroll
      for (int e = 0; e < 4; ++e) { w[e] = w0[e]; w[4 + e] = w1[e]; }
      *(h16x8*)(a.out + (int64_t)m * a.ldo + n) = w;
    }
  }
#ifdef FF_STAMP
  FSTAMP(t_epi);
  if (stamps && lane == 0) {
    unsigned long long* o = stamps + ((size_t)blockIdx.x * 8 + wid) * 8;
    o[0] = t_wait; o[1] = t_bar; o[2] = t_dma; o[3] = t_g1; o[4] = t_gelu; o[5] = t_g2; o[6] = t_all1 - t_all0; o[7] = t_epi - t_all1;
  }
#endif
}

// W2p[n][32 b + 8 lq + j] = W2[n][32 b + 16 (j >> 2) + 4 lq + (j & 3)]: the k order in which the GEGLU registers of a lane line up
__global__ void pack_ff2_perm_kernel(const float* __restrict__ w, h16* __restrict__ out, int N, int K) {
  const int64_t total = (int64_t)N * K;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(idx / K), k = (int)(idx - (int64_t)n * K);
    const int b = k >> 5, p = k & 31, lq = p >> 3, j = p & 7;
    out[idx] = (h16)w[(int64_t)n * K + 32 * b + 16 * (j >> 2) + 4 * lq + (j & 3)];
  }
}

}  // namespace

void ff_fused_init_device() {
  HIP_OK(hipFuncSetAttribute((const void*)ff_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS));
  HIP_OK(hipFuncSetAttribute((const void*)ff_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS));
}

bool ff_fused_supported(int C, int M) {
  static const int on = getenv("SVG_FF_FUSED") ? atoi(getenv("SVG_FF_FUSED")) : 1;
  return on && C == FC && M >= 128 * 192;     // enough 128-row tiles to give every CU a workgroup
}

void pack_ff2_perm(const float* w, h16* out, int N, int K, hipStream_t s) {
  const int64_t total = (int64_t)N * K;
  hipLaunchKernelGGL(pack_ff2_perm_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, s, w, out, N, K);
  check_launch("pack_ff2_perm");
}

void ff_fused(svg_ctx* ctx, const h16* X, int ldx, const h16* W1, const float* b1, const float* s1, const float* rs, const float* rm,
              const h16* W2p, const float* b2, const h16* residual, int ldr, h16* out, int ldo, int M, hipStream_t s) {
  SVG_CHECK(ldx % 8 == 0 && ldr % 8 == 0 && ldo % 8 == 0 && (int64_t)M * ldx < (1LL << 31), "ff_fused: strides / size unsupported");
  if (!SVG_LAUNCHING(ctx)) return;
  char tag[64];
  snprintf(tag, sizeof(tag), "ff_fused_M%d_C%d", M, FC);
  ProfScope ps(ctx, PK_GEMM, s, 2.0 * M * (double)FC * (2 * FH) + 2.0 * M * (double)FH * FC,
               2.0 * ((double)M * FC * 3 + 3.0 * FC * FH), tag);
  FfArgs a{X, ldx, W1, b1, s1, rs, rm, W2p, b2, residual, ldr, out, ldo, M};
  // SVG_FF_PAIR: 1 (default) the paired 32-row form, 0 the 16-rows-per-wave form
  unsigned long long* stamps = nullptr;
#ifdef FF_STAMP
  const size_t n_st = (size_t)cdiv(M, 128) * 64;
  HIP_OK(hipMalloc(&stamps, n_st * sizeof(unsigned long long)));
  HIP_OK(hipMemsetAsync(stamps, 0, n_st * sizeof(unsigned long long), s));
#endif
  if (svg_env_i64("SVG_FF_PAIR", 1) != 0) hipLaunchKernelGGL(ff_pair_kernel, dim3(cdiv(M, 128)), dim3(512), P_LDS, s, a, stamps);
  else hipLaunchKernelGGL(ff_fused_kernel, dim3(cdiv(M, 128)), dim3(512), F_LDS, s, a);
  check_launch("ff_fused");
#ifdef FF_STAMP
  {
    HIP_OK(hipStreamSynchronize(s));
    std::vector<unsigned long long> h(n_st);
    HIP_OK(hipMemcpy(h.data(), stamps, n_st * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIP_OK(hipFree(stamps));
    double sum[8] = {0};
    for (size_t w = 0; w < n_st / 8; ++w) for (int k = 0; k < 8; ++k) sum[k] += (double)h[w * 8 + k];
    const double nw = (double)(n_st / 8);
    fprintf(stderr, "[ff stamps] M %d: per wave (cycles, 160 slab steps): ring wait %.0f | barrier %.0f | DMA issue %.0f | GEMM1 slabs (100 x: 8 reads + 16 MFMA) %.0f | "
            "LN-fold + GEGLU + swap write (20 x) %.0f | GEMM2 slabs (60 x: reads + up to 16 MFMA) %.0f || main loop %.0f, epilogue %.0f; matrix time of the loop = 2400 MFMA x 16 = 38400\n",
            M, sum[0] / nw, sum[1] / nw, sum[2] / nw, sum[3] / nw, sum[4] / nw, sum[5] / nw, sum[6] / nw, sum[7] / nw);
  }
#endif
}

}  // namespace SDNS
