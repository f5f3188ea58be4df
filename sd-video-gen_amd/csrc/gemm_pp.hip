// Dense NT GEMM for gfx950 with the ping-pong schedule of conv_halo.hip: C[M,N] = A[M,K] * W[N,K]^T (+ epilogue).
//
// gemm.hip's 128 x 160 tile with two 4-wave workgroups per CU moves (128 + 160) x 128 B per 640 matrix-pipe cycles
// through L2 -> LDS — 58 B/clk/CU, the whole L2 rate — and every K step ends in vmcnt(0) + barrier, so it tops out near
// 0.65-0.85 PFLOP/s.  Here one 8-wave workgroup per CU owns a 256 x BN tile (42 B/clk at BN = 160), operand slabs
// (64 of K) stream through THREE LDS stages with counted vmcnt, and the two wave groups alternate: while waves 0-3
// multiply (20 MFMAs between two barriers) waves 4-7 issue the next phase's fragment reads and their share of the DMAs.
// Used for long K (>= 16 slabs), where the unhidden prologue / epilogue of a tile (one workgroup per CU: nothing else
// on the CU covers them) is small; short K stays with gemm.hip.
//   waves 4 (M) x 2 (N); wave tile 64 x BN/2; v_mfma_f32_16x16x32_bf16 with the W fragment as the A operand.
//   LDS rows are 128 B, chunk c of row r at c ^ (r & 7) (swizzle applied on the DMA source side).
#include "igemm_epi.h"
#include <algorithm>
#include <cstdlib>
#include <vector>

namespace SDNS {

namespace {

// In-kernel stamps (diagnostic build only: -DPP_STAMP, tools/build_variant.sh; cdna_hip_programming.md section 7): where a slab of the merged
// ping-pong loop spends its cycles — per wave the sums over all slabs of {fragment-read + DMA issue, counted waits, barrier into the MFMA
// segment, the MFMA segment, barrier out of it}, written to (g.slabs)[workgroup][wave][8] and printed by the launcher.
#ifdef PP_STAMP
#define PSTAMP(v) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PSTAMP(v) do { } while (0)
#endif

// MFMA (0-based, of the 2 nm in a merged segment) behind which DMA slot o sits: the first three at the middle of the first k half, the
// second MFMA and the middle of the second k half (the placement measured best), the next three between them
__host__ __device__ constexpr int pp_slot_at(int o, int nm) {
  return o == 0 ? nm / 2 - 1 : o == 1 ? nm + 1 : o == 2 ? nm + nm / 2 - 1 : o == 3 ? (3 * nm) / 4 : o == 4 ? 2 * nm - 5 : o == 5 ? 3 : nm + nm / 4;
}

// WIDE: the wide tile epilogue (igemm_epi.h; 16-bit output without GEGLU — the launcher decides)
template <int BN, bool WIDE>
__global__ void __launch_bounds__(512, 2) gemm_pp_kernel(const GemmArgs g) {
  constexpr int NT = BN / 32;            // 16-wide n tiles per wave
  constexpr int MT = 4;                  // 16-high m tiles per wave
  constexpr int BMP = 256;
  constexpr int A_BYTES = BMP * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int NWS = 3;
  constexpr int NA = 4;                  // A DMA instructions per wave and slab (64 rows each across the 8 waves)
  constexpr int BIT = BN / 64;
  constexpr bool B_TAIL = (BN % 64) != 0;
  constexpr int NB = BIT + (B_TAIL ? 1 : 0);
  constexpr int NW = NA + NB;            // DMA instructions per wave and slab
  constexpr int OFF_SINK = NWS * STAGE;  // 1 KiB sink for the padding DMAs of the ragged B group
  constexpr unsigned INVALID = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int wave_u = __builtin_amdgcn_readfirstlane(wid);
  const int l15 = lane & 15, lq = lane >> 4;

  const int tiles_n = (g.N + BN - 1) / BN;
  int tile;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // Tile order inside an XCD's contiguous run: groups of g.group_m row tiles, the column tile walking slowest inside a group
  // and the row tile fastest, so the ~32 workgroups an XCD runs at a time cover a compact group_m x (32 / group_m) rectangle
  // whose A and W panels fit its 4 MiB L2 (row-major order at N = 10240 keeps 33 panels live per XCD: 780 MB per launch
  // from beyond L2 against 26 + 18 MB of operands).  group_m <= 1: plain row-major order.
  int tm, tn;
  if (g.group_m > 1) {
    const int tiles_m = (g.M + BMP - 1) / BMP;
    const int per = g.group_m * tiles_n;
    const int grp_i = tile / per, in = tile - grp_i * per;
    const int gm = min(g.group_m, tiles_m - grp_i * g.group_m);
    tn = in / gm; tm = grp_i * g.group_m + (in - tn * gm);
  } else {
    tm = tile / tiles_n; tn = tile - tm * tiles_n;
  }
  const int m0 = tm * BMP, n0 = tn * BN;
  const int KT = g.K >> 6;

  const unsigned a_bytes = (unsigned)(((int64_t)(g.M - 1) * g.lda + g.K) * 2);
  const unsigned b_bytes = (unsigned)(((int64_t)(g.n_valid - 1) * g.ldb + g.K) * 2);
  const uint64_t pa = (uint64_t)g.A, pw = (uint64_t)g.Wt;
  const v4i srdA = {(int)(unsigned)pa, (int)((pa >> 32) & 0xffff), (int)a_bytes, 0x00020000};
  const v4i srdB = {(int)(unsigned)pw, (int)((pw >> 32) & 0xffff), (int)b_bytes, 0x00020000};
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);

  // DMA maps: a wave instruction fills 8 LDS rows (64 lanes x 16 B); lane (r0, p) fetches logical chunk p ^ (r0 & 7).
  // Rows past M / n_valid fall outside the descriptor's range and read as zeros.
  const int r0 = tid >> 3;                                   // 0..63
  const int cs = (tid & 7) ^ (r0 & 7);
  const unsigned a_voff0 = (unsigned)((m0 + r0) * g.lda + cs * 8) * 2u;
  const unsigned a_step = (unsigned)(64 * g.lda) * 2u;
  const unsigned b_voff0 = (unsigned)((n0 + r0) * g.ldb + cs * 8) * 2u;
  const unsigned b_step = (unsigned)(64 * g.ldb) * 2u;
  // (a row offset past the range stays past it: offsets are < 2^31 by the host-side checks, INVALID is 2^31)
  auto dma_part = [&](int kt, int stage, int i0, int i1) {   // instructions [i0, i1) of the slab's NW
    const unsigned dst = lds0 + stage * STAGE + wave_u * 1024;
    const int soff = kt * 128;
#pragma unroll
    for (int i = i0; i < i1; ++i) {
      if (i < NA) {
        const int m = m0 + r0 + 64 * i;
        dma16(srdA, m < g.M ? a_voff0 + i * a_step : INVALID, soff, dst + i * 8192);
      } else {
        const int ib = i - NA;
        const int n = n0 + r0 + 64 * ib;
        if (ib < BIT) dma16(srdB, n < g.n_valid ? b_voff0 + ib * b_step : INVALID, soff, dst + A_BYTES + ib * 8192);
        else dma16(srdB, (r0 < 32 && n < g.n_valid) ? b_voff0 + ib * b_step : INVALID, soff,
                   (wave_u < 4) ? (dst + A_BYTES + BIT * 8192) : (lds0 + OFF_SINK));
      }
    }
  };

  // fragment addresses: the swizzle key of every row a lane reads is l15 & 7 (tile offsets are multiples of 8)
  int xaddr[2], waddr[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int sw = ((kk * 4 + lq) ^ (l15 & 7)) << 4;
    xaddr[kk] = (wm * 64 + l15) * 128 + sw;
    waddr[kk] = A_BYTES + (wn * (BN / 2) + l15) * 128 + sw;
  }
  auto rd = [&](int stage, int kk, h16x8 (&xf)[MT], h16x8 (&wf)[NT]) {
    const char* sx = smem + stage * STAGE + xaddr[kk];
    const char* sw = smem + stage * STAGE + waddr[kk];
#pragma unroll
    for (int i = 0; i < MT; ++i) xf[i] = *(const h16x8*)(sx + i * 2048);
#pragma unroll
    for (int j = 0; j < NT; ++j) wf[j] = *(const h16x8*)(sw + j * 2048);
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mma_phase = [&](const h16x8 (&xf)[MT], const h16x8 (&wf)[NT]) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[i][j] = MFMA_16x16x32(wf[j], xf[i], acc[i][j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  // ---- main loop: same phase structure, RAW / WAR argument and vmcnt accounting as conv_halo.hip's ping-pong loop:
  // step s (slab s, stage s % 3) = phase 0 [reads of its second k half; first NA DMAs of slab s+2; wait until only those
  // are in flight -> slab s+1 landed] + phase 1 [reads of the first k half of slab s+1; the other DMAs of slab s+2].
  if (g.pp_merge) {
    // merged form (as conv_halo.hip's MODE 2): a slab is ONE phase of 2 * MT * NT MFMAs between two barriers; its fragments are read in
    // its own load segment (one register set), whose latency runs beside the other group's MFMA segment.
    //   L(s): read slab s; DMA slab s + 2 -> stage (s + 2) % 3 (slab s - 1: drained by both groups before the barriers since);
    //         vmcnt(own issues) -> slab s + 1 landed; lgkmcnt(0).
    const int grp = wave_u >> 2;
    const int km_cfg = min(__builtin_amdgcn_readfirstlane(g.pp_dma_m), NW);
    unsigned long long ps_rd = 0, ps_wait = 0, ps_b1 = 0, ps_mm = 0, ps_b2 = 0, ps_n = 0, ps_t0 = 0, ps_t1 = 0;
    PSTAMP(ps_t0);
    dma_part(0, 0, 0, NW);
    dma_part(1, 1, 0, NW);
    wait_vm(NW);
    bar();
    if (grp == 1) bar();
    int st = 0;
    for (int s = 0; s < KT; ++s) {
      const bool more = s + 2 < KT;
      const int st2 = st == 0 ? 2 : st - 1;
      h16x8 x0[MT], w0[NT], x1[MT], w1[NT];
      unsigned long long p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, p5 = 0;
      PSTAMP(p0);
      rd(st, 0, x0, w0);
      rd(st, 1, x1, w1);
      // km of the slab's NW DMA instructions ride in the MFMA segment (round 6): a load segment of 18 fragment reads + 7 DMA issues
      // (~110 cycles each with four waves issuing: profiles/r06_probe_ldsdma.txt) is longer than the 40-MFMA segment it should hide
      // under, and every barrier interval lasts max(load, matrix).  A DMA among the MFMAs stalls that wave's matrix stream while it
      // issues, so the optimum moves only as many as balance the two segments (g.pp_dma_m, tuned per tile width by the launcher).
      // Order / safety: the moved instructions are issued AFTER the barrier that closes L(s) (later than before: WAR on stage
      // (s + 2) % 3 holds a fortiori) and BEFORE L(s + 1)'s issues, whose counted wait therefore covers them (RAW unchanged).
      const int km = more ? km_cfg : 0;
      if (more) {
#pragma unroll
        for (int i = 0; i < NW; ++i)
          if (i < NW - km_cfg) dma_part(s + 2, st2, i, i + 1);
      }
      PSTAMP(p1);
      wait_vm(more ? NW - km_cfg : 0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PSTAMP(p2);
      bar();
      PSTAMP(p3);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      // slot o (compile-time position pp_slot_at(o)) carries DMA instruction NW - 1 - o when km > o.  (A run-time slot mask with a
      // run-time instruction index was measured too: its address arithmetic and ten scalar slot tests per segment cost more than the
      // move gains — profiles/r06_pp_dma_in_mfma.txt.)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            acc[i][j] = h == 0 ? MFMA_16x16x32(w0[j], x0[i], acc[i][j]) : MFMA_16x16x32(w1[j], x1[i], acc[i][j]);
            const int idx = (h * MT + i) * NT + j;
#pragma unroll
            for (int o = 0; o < 7; ++o)
              if (idx == pp_slot_at(o, MT * NT) && o < NW) {
                __builtin_amdgcn_sched_barrier(0);
                if (km > o) dma_part(s + 2, st2, NW - 1 - o, NW - o);
                __builtin_amdgcn_sched_barrier(0);
              }
          }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
#ifdef PP_STAMP
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(acc[i][j]));
#endif
      PSTAMP(p4);
      bar();
      PSTAMP(p5);
      ps_rd += p1 - p0; ps_wait += p2 - p1; ps_b1 += p3 - p2; ps_mm += p4 - p3; ps_b2 += p5 - p4; ps_n += 1;
      st = st == 2 ? 0 : st + 1;
    }
    if (grp == 0) bar();
#ifdef PP_STAMP
    PSTAMP(ps_t1);
    if (g.slabs && lane == 0) {
      unsigned long long* o = (unsigned long long*)g.slabs + ((size_t)blockIdx.x * 8 + wid) * 8;
      o[0] = ps_rd; o[1] = ps_wait; o[2] = ps_b1; o[3] = ps_mm; o[4] = ps_b2; o[5] = ps_t1 - ps_t0; o[6] = 0; o[7] = ps_n;
    }
#endif
  } else {
    const int grp = wave_u >> 2;
    dma_part(0, 0, 0, NW);
    dma_part(1, 1, 0, NW);
    wait_vm(NW);                              // NW <= 7: slab 0 landed, slab 1 may be in flight
    bar();
    if (grp == 1) bar();
    h16x8 xa[MT], wa[NT], xb[MT], wb[NT];
    rd(0, 0, xa, wa);
    int st = 0;                               // stage of slab s
    for (int s = 0; s < KT; ++s) {
      const bool more = s + 2 < KT;
      const int st2 = st == 0 ? 2 : st - 1;   // (s + 2) % 3
      const int st1 = st == 2 ? 0 : st + 1;   // (s + 1) % 3
      rd(st, 1, xb, wb);
      if (more) dma_part(s + 2, st2, 0, NA);
      wait_vm(more ? NA : 0);
      bar();
      mma_phase(xa, wa);
      bar();
      if (s + 1 < KT) rd(st1, 0, xa, wa);
      if (more) dma_part(s + 2, st2, NA, NW);
      bar();
      mma_phase(xb, wb);
      bar();
      st = st1;
    }
    if (grp == 0) bar();
  }

  // ---- epilogue ---------------------------------------------------------------------------------------------------
  epi_tile<MT, NT, true, WIDE>(g, 0, m0 + wm * 64 + l15, 16, n0 + wn * (BN / 2) + lq * 4, acc, smem, 4, wm, wn, tm, n0);
}

template <int BN>
void launch_pp(const GemmArgs& g, hipStream_t s) {
  constexpr int smem = 3 * (256 * 128 + BN * 128) + 1024;
  const int tiles = cdiv(g.M, 256) * cdiv(g.N, BN);
  if (g.act == ACT_GEGLU || g.out_f32) hipLaunchKernelGGL((gemm_pp_kernel<BN, false>), dim3(tiles), dim3(512), smem, s, g);
  else hipLaunchKernelGGL((gemm_pp_kernel<BN, true>), dim3(tiles), dim3(512), smem, s, g);
}

}  // namespace

void gemm_pp_init_device() {
  HIP_OK(hipFuncSetAttribute((const void*)gemm_pp_kernel<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (256 * 128 + 128 * 128) + 1024));
  HIP_OK(hipFuncSetAttribute((const void*)gemm_pp_kernel<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (256 * 128 + 128 * 128) + 1024));
  HIP_OK(hipFuncSetAttribute((const void*)gemm_pp_kernel<160, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (256 * 128 + 160 * 128) + 1024));
  HIP_OK(hipFuncSetAttribute((const void*)gemm_pp_kernel<160, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (256 * 128 + 160 * 128) + 1024));
}

int gemm_pp_bn(const GemmArgs& g) {
  if (g.act == ACT_GEGLU) return 128;
  // fewest serial rounds of workgroups (one per CU) x tile width; ties -> fewer padded columns
  const int64_t tm = cdiv(g.M, 256);
  int best = 128;
  int64_t best_cost = -1, best_pad = 0;
  for (int bn : {128, 160}) {
    const int64_t tn = cdiv(g.N, bn);
    const int64_t cost = ((tm * tn + 255) / 256) * bn, pad = tn * bn - g.N;
    if (best_cost < 0 || cost < best_cost || (cost == best_cost && pad < best_pad)) { best = bn; best_cost = cost; best_pad = pad; }
  }
  return best;
}

// dense, unbatched, K a multiple of 64 and long enough, enough 256-row tiles to give every CU a workgroup
bool gemm_pp_supported(const GemmArgs& g) {
  static const int on = getenv("SVG_GEMM_PP") ? atoi(getenv("SVG_GEMM_PP")) : 1;
  static const int min_kt = getenv("SVG_GEMM_PP_MINKT") ? atoi(getenv("SVG_GEMM_PP_MINKT")) : 16;
  // wide outputs (the GEGLU projection at 32 x 32: N = 5120, K = 640) amortise the tile's unhidden prologue over enough columns at
  // 10 slabs already: 0.248 against 0.258-0.262 ms on the 128-row kernel, same box; N = 1280 at K = 640 loses (0.081 against 0.073)
  const int need_kt = g.N >= 2560 ? std::min(min_kt, 10) : min_kt;
  if (!on || g.amode != A_DENSE || g.batch != 1 || g.splitk > 1 || g.out_f32 || (g.K & 63) != 0 || (g.K >> 6) < need_kt || g.A2) return false;
  if (g.lda % 8 != 0 || g.ldb % 8 != 0 || g.bias_row) return false;
  return (int64_t)cdiv(g.M, 256) * cdiv(g.N, gemm_pp_bn(g)) >= 192;
}

void launch_gemm_pp(const GemmArgs& g0, hipStream_t s) {
  static const int gm_env = getenv("SVG_PP_GROUPM") ? atoi(getenv("SVG_PP_GROUPM")) : 4;
  GemmArgs g = g0;
  g.group_m = gm_env;
  g.pp_merge = (int)svg_env_i64("SVG_PP_MERGE", 1);        // 0 = the two-phase loop
  // DMA instructions of a slab that ride among the MFMAs (see the merged loop)
  g.pp_dma_m = std::max(0, std::min(7, (int)svg_env_i64("SVG_PP_DMA_M", 6)));
#ifdef PP_STAMP
  const int tiles_dbg = cdiv(g.M, 256) * cdiv(g.N, gemm_pp_bn(g));
  unsigned long long* dbg = nullptr;
  HIP_OK(hipMalloc(&dbg, (size_t)tiles_dbg * 64 * sizeof(unsigned long long)));
  HIP_OK(hipMemsetAsync(dbg, 0, (size_t)tiles_dbg * 64 * sizeof(unsigned long long), s));
  g.slabs = (float*)dbg;
#endif
  if (gemm_pp_bn(g) == 160) launch_pp<160>(g, s);
  else launch_pp<128>(g, s);
#ifdef PP_STAMP
  {
    HIP_OK(hipStreamSynchronize(s));
    std::vector<unsigned long long> h((size_t)tiles_dbg * 64);
    HIP_OK(hipMemcpy(h.data(), dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIP_OK(hipFree(dbg));
    double sum[2][8] = {{0}};
    for (size_t w = 0; w < (size_t)tiles_dbg * 8; ++w)
      for (int k = 0; k < 8; ++k) sum[(w & 7) >> 2][k] += (double)h[w * 8 + k];
    for (int grp = 0; grp < 2; ++grp) {
      const double n = sum[grp][7] > 0 ? sum[grp][7] : 1, nw = (double)tiles_dbg * 4;
      fprintf(stderr, "[pp stamps] M%d N%d K%d BN%d dma_m %d grp %d: per slab (cycles) reads + DMA issue %.0f | waits %.0f | barrier in %.0f | MFMA segment %.0f | barrier out %.0f  = %.0f || loop per tile %.0f, slabs %.0f\n",
              g.M, g.N, g.K, gemm_pp_bn(g), g.pp_dma_m, grp, sum[grp][0] / n, sum[grp][1] / n, sum[grp][2] / n, sum[grp][3] / n, sum[grp][4] / n,
              (sum[grp][0] + sum[grp][1] + sum[grp][2] + sum[grp][3] + sum[grp][4]) / n, sum[grp][5] / nw, n / nw);
    }
  }
#endif
}

}  // namespace SDNS
