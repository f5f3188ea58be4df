// Halo 3x3 convolution (stride 1, pad 1) for gfx950: the L2-traffic-lean form of the implicit GEMM.
//
// The plain implicit GEMM (gemm.hip) re-loads its 128-pixel A tile once per tap: 9x the input through L2 -> LDS.
// Measured on MI355X that traffic runs at ~39 % of the L2 rate while the matrix pipe is ~38 % busy — at a 128 x 160
// tile the two limits coincide.  Here a workgroup owns a 16 x 16 PIXEL BLOCK (256 output pixels) x BN channels:
//   * per 64-channel chunk it DMAs the block's 18 x 18 halo patch ONCE (41 KB) and reads all nine taps out of LDS at
//     shifted pixel addresses (the activation fragment of tap (ky,kx), pixel (y,x) is patch pixel (y+ky, x+kx));
//   * only the weights stream per tap (BN x 64, THREE LDS stages, LDS-direct buffer loads, counted vmcnt: the slab of
//     step s+2 is issued while step s is multiplied — one workgroup per CU has no neighbour to hide DMA latency);
//   => (41.5 + 9*20) KB per 47 MFLOP instead of 9*(16+20) KB per 23.6 MFLOP: 2.8x fewer L2 bytes per FLOP.
// 8 waves as 4 (pixel rows) x 2 (channels); wave tile = 4 image rows x 16 px x BN/2 channels; v_mfma_f32_16x16x32_bf16
// with the weight fragment as the A operand (4 consecutive output channels per lane, like gemm.hip).
// Image borders and ragged channel tiles are zero-filled by the buffer range check (voffset 0x80000000).
#include "igemm_epi.h"
#include <cstdlib>

namespace SDNS {

namespace {

// compile-time ablation switch for tools/abl_halo.sh (-DHALO_ABL=n): 1 no stores, 2 no MFMA, 3 no DMA, 4 no fragment reads, 6 no barrier
#ifndef HALO_ABL
#define HALO_ABL 0
#endif
constexpr int ABL = HALO_ABL;
constexpr int PW = 18;                 // patch width / height in pixels
constexpr int PPIX = PW * PW;          // 324
constexpr int NPD = 6;                 // patch DMA instructions per wave: 8 waves * 64 lanes * 6 = 3072 >= 324 * 8 chunks
// MFMA (0-based, of the 2 nm in a merged segment) behind which DMA slot o of the merged loop sits (the placement gemm_pp.hip measured best)
__host__ __device__ constexpr int halo_slot_at(int o, int nm) {
  return o == 0 ? nm / 2 - 1 : o == 1 ? nm + 1 : o == 2 ? nm + nm / 2 - 1 : o == 3 ? (3 * nm) / 4 : 2 * nm - 5;
}

// PP = ping-pong schedule: the 8 waves run as two groups of four (one wave of each group per SIMD) staggered by one
// barrier, so that while one group multiplies (20 MFMAs between two barriers, s_setprio 1) the other issues its fragment
// reads and DMAs — the matrix pipe always has a wave feeding it instead of all eight loading, then all eight multiplying.
// MODE 0: lock-step loop, 1: ping-pong (PP), 2: merged ping-pong (below, BN = 128 with an even number of channel chunks)
template <int BN, int MODE>
__global__ void __launch_bounds__(512, 2) conv_halo_kernel(const GemmArgs g) {
  constexpr bool PP = MODE != 0;
  constexpr int NT = BN / 32;            // 16-wide channel tiles per wave
  constexpr int MT = 4;                  // image rows per wave
  constexpr int BIT = BN / 64;           // weight row groups per thread (64 rows per DMA instruction of all 8 waves)
  constexpr int B_BYTES = BN * 128;
  constexpr int PBUF = PPIX * 128;       // one patch buffer: 324 px * 128 B = 41472 B (lanes past it are EXEC-masked)
  constexpr int NWS = 3;                 // weight stages
  constexpr int NW = (BN + 63) / 64;     // weight DMA instructions per wave and slab (BN = 160: waves 4-7 pad with a dummy)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sP = smem;                 // 2 patch buffers
  char* const sB = smem + 2 * PBUF;      // NWS weight stages
  char* const sDummy = sB + NWS * B_BYTES;   // 8 KiB sink for the padding DMAs (1 KiB per wave)
  constexpr unsigned INVALID = 0x80000000u;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int wave_u = __builtin_amdgcn_readfirstlane(wid);
  const int l15 = lane & 15, lq = lane >> 4;
  const int ks = blockIdx.z;

  // tile order: the channel tiles of one pixel block are adjacent (they share the patch through L2)
  const int tiles_n = (g.N + BN - 1) / BN;
  // nearest-2x upsample fused in front (A_CONV_UP2: the VAE / UNet upsamplers): the conv runs over the OH x OW upsampled image, patch pixel
  // (yy, xx) of it is source pixel (yy >> 1, xx >> 1) — only the DMA source offsets change, the LDS patch holds the upsampled pixels
  const bool up2 = g.amode == A_CONV_UP2;
  const int OH = up2 ? g.Ho : g.H, OW = up2 ? g.Wo : g.W;
  const int bx_n = OW >> 4, by_n = OH >> 4;
  int tile;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int tn, pb;
  if (g.tn_major) {   // weight-heavy (small images): the XCD's run of tiles keeps one channel tile's weights in its L2
    const int npb = gridDim.x / tiles_n;
    tn = tile / npb; pb = tile - tn * npb;
  } else {
    tn = tile % tiles_n; pb = tile / tiles_n;
  }
  const int tile_m = pb;                 // linear index of the 16 x 16 pixel block: (b * by_n + by) * bx_n + bx
  const int bx = pb % bx_n; pb /= bx_n;
  const int by = pb % by_n;
  const int b = pb / by_n;
  const int y0 = by << 4, x0 = bx << 4, n0 = tn * BN;

  // channel chunks of this K split
  const int CC = g.Cin >> 6;
  const int cc_per = (CC + g.splitk - 1) / g.splitk;
  const int cc_begin = ks * cc_per;
  const int cc_end = min(CC, cc_begin + cc_per);

  const unsigned a_bytes = (unsigned)((int64_t)(g.M / (OH * OW)) * g.H * g.W * g.Cin * 2);
  const unsigned b_bytes = (unsigned)(((int64_t)(g.n_valid - 1) * g.ldb + g.K) * 2);
  // buffer descriptors: {base[31:0], base[47:32] (stride 0), num_records, flags}
  const uint64_t pa = (uint64_t)g.A, pw = (uint64_t)g.Wt;
  const v4i srdA = {(int)(unsigned)pa, (int)((pa >> 32) & 0xffff), (int)a_bytes, 0x00020000};
  const v4i srdB = {(int)(unsigned)pw, (int)((pw >> 32) & 0xffff), (int)b_bytes, 0x00020000};
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);   // LDS address of smem[0]

  // ---- patch DMA map: linear LDS image [pixel][8 chunks]; lane id -> (pp, position); the swizzle rides on the source chunk
  auto patch_voff = [&](int i) -> unsigned {
    int ln = lane;
    if (PP) asm volatile("" : "+v"(ln));               // keeps LICM from hoisting the six offsets back into registers
    const int id = (i * 8 + wid) * 64 + ln;            // 16-byte slot in the patch buffer
    const int pp = id >> 3, pos = id & 7;
    const int c = pos ^ (pp & 7);
    const int py = pp / PW, px = pp - py * PW;
    const int yy = y0 - 1 + py, xx = x0 - 1 + px;
    const bool ok = pp < PPIX && (unsigned)yy < (unsigned)OH && (unsigned)xx < (unsigned)OW;
    const int sy = up2 ? yy >> 1 : yy, sx = up2 ? xx >> 1 : xx;
    return ok ? (unsigned)(((b * g.H + sy) * g.W + sx) * g.Cin + c * 8) * 2u : INVALID;
  };
  // the lock-step loop keeps the six offsets in registers; the ping-pong loop (one piece per step, registers are its
  // scarce resource) recomputes the piece's offset when it issues it
  unsigned p_voff[NPD];
#pragma unroll
  for (int i = 0; i < NPD; ++i) p_voff[i] = PP ? 0u : patch_voff(i);
  // ---- weight DMA map (as gemm.hip): rows r0 + 64 i, chunk swizzled on the source
  const int r0 = tid >> 3;                                // 0..63
  const int cB = (tid & 7) ^ (r0 & 7);
  unsigned b_voff[5];
  // PERM (BN = 160, NT = 5: round 5): LDS row rb of the weight slab holds the weight row its MFMA operand row stands for in the 16-byte epilogue
  // (igemm_epi.h: epi_perm_col — the quads of a tile pair are 8 consecutive output channels per lane, without lane exchanges or registers)
  constexpr bool PERM = (NT & 1) != 0;
  auto w_row = [&](int rb) {
    if (!PERM) return n0 + rb;
    const int run = rb / (BN / 2);
    return n0 + run * (BN / 2) + epi_perm_col<NT>(rb - run * (BN / 2));
  };
#pragma unroll
  for (int i = 0; i < BIT; ++i) {
    const int n = w_row(r0 + 64 * i);
    b_voff[i] = (n < g.n_valid) ? (unsigned)(n * g.ldb + cB * 8) * 2u : INVALID;
  }
  constexpr bool B_TAIL = (BN % 64) != 0;                 // BN = 160: a last half group of 32 rows
  unsigned b_voff_tail = INVALID;
  if (B_TAIL) {
    const int n = w_row(BIT * 64 + (r0 & 31));
    if (r0 < 32 && n < g.n_valid) b_voff_tail = (unsigned)(n * g.ldb + cB * 8) * 2u;
  }

  constexpr int OFF_B = 2 * PBUF, OFF_DUMMY = OFF_B + NWS * B_BYTES;
  auto dma_patch = [&](int cc, int buf) {
    const unsigned dst = lds0 + buf * PBUF + wave_u * 1024;
    const int soff = cc * 128;
#pragma unroll
    for (int i = 0; i < NPD - 1; ++i) dma16(srdA, p_voff[i], soff, dst + i * 8192);
    // last piece: only slots < 324*8 exist (32 of them, on wave 0).  Every wave still issues the instruction so that all
    // waves count the same number of DMAs: wave 0 with lanes 32-63 switched off (a DMA writes active lanes only), the
    // other waves into the sink with an out-of-range source.
    const unsigned d5 = (wave_u == 0) ? (dst + (NPD - 1) * 8192) : (lds0 + OFF_DUMMY + wave_u * 1024);
    if (wid != 0 || lane < 32) dma16(srdA, wid == 0 ? p_voff[NPD - 1] : INVALID, soff, d5);
  };
  auto dma_patch_piece = [&](int cc, int buf, int i) {      // one of the NPD pieces (i is a compile-time constant after unrolling)
    const unsigned dst = lds0 + buf * PBUF + wave_u * 1024;
    const int soff = cc * 128;
    const unsigned voff = patch_voff(i);
    if (i < NPD - 1) dma16(srdA, voff, soff, dst + i * 8192);
    else {
      const unsigned d5 = (wave_u == 0) ? (dst + (NPD - 1) * 8192) : (lds0 + OFF_DUMMY + wave_u * 1024);
      if (wid != 0 || lane < 32) dma16(srdA, wid == 0 ? voff : INVALID, soff, d5);
    }
  };
  auto dma_weights = [&](int cc, int tap, int stage) {
    const unsigned dst = lds0 + OFF_B + stage * B_BYTES + wave_u * 1024;
    const int soff = (tap * g.Cin + cc * 64) * 2;
#pragma unroll
    for (int i = 0; i < BIT; ++i) dma16(srdB, b_voff[i], soff, dst + i * 8192);
    if (B_TAIL) {
      // rows BIT*64 .. BIT*64+31 ride on waves 0-3 (r0 < 32); waves 4-7 issue a padding DMA into the sink so that
      // every wave has the same number of DMAs per slab (the counted vmcnt below relies on it)
      const unsigned tdst = (wave_u < 4) ? (dst + BIT * 8192) : (lds0 + OFF_DUMMY + wave_u * 1024);
      dma16(srdB, b_voff_tail, soff, tdst);
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // patch pixel of (image row wm*4 + i, column l15) for tap (0,0); tap (ky,kx) adds ky*18 + kx
  int pp0[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) pp0[i] = (wm * 4 + i) * PW + l15;

  auto compute = [&](int pbuf, int stage, int toff) {
    const char* sp = sP + pbuf * PBUF;
    const char* sb = sB + stage * B_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      h16x8 xf[MT], wf[NT];
      if (ABL != 4) {
#pragma unroll
      for (int i = 0; i < MT; ++i) xf[i] = *(const h16x8*)(sp + lds_off7(pp0[i] + toff, kk * 4 + lq));
#pragma unroll
      for (int j = 0; j < NT; ++j) wf[j] = *(const h16x8*)(sb + lds_off7(wn * (BN / 2) + j * 16 + l15, kk * 4 + lq));
      } else {
#pragma unroll
      for (int i = 0; i < MT; ++i) asm volatile("" : "=v"(xf[i]));
#pragma unroll
      for (int j = 0; j < NT; ++j) asm volatile("" : "=v"(wf[j]));
      }
      if (ABL == 2) {
#pragma unroll
        for (int i = 0; i < MT; ++i) asm volatile("" :: "v"(xf[i]));
#pragma unroll
        for (int j = 0; j < NT; ++j) asm volatile("" :: "v"(wf[j]));
        continue;
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = MFMA_16x16x32(wf[j], xf[i], acc[i][j]);
    }
  };

  // the operands were read a phase ago (the compiler's own counted lgkmcnt covers them, not the reads just issued)
  auto mma_phase = [&](const h16x8 (&xf)[MT], const h16x8 (&wf)[NT]) {
    __builtin_amdgcn_sched_barrier(0);
    if (ABL == 2) {
#pragma unroll
      for (int i = 0; i < MT; ++i) asm volatile("" :: "v"(xf[i]));
#pragma unroll
      for (int j = 0; j < NT; ++j) asm volatile("" :: "v"(wf[j]));
      return;
    }
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

  // ---- ping-pong main loop ------------------------------------------------------------------------------------------
  // Step s = (cc, tap) is two phases (the k halves of the 64-channel chunk); a phase is
  //     [issue the fragment reads of the NEXT phase + this phase's share of the DMAs (+ counted wait in phase 0)]
  //     barrier  [20 MFMAs]  barrier
  // and group 1 (waves 4-7) runs one barrier behind group 0, so its load segment coincides with group 0's MFMAs.
  // Fragments are read one phase ahead into the other register set, so an MFMA segment never waits on LDS latency; every
  // wave drains its own reads (lgkmcnt(0), long since landed) before the barrier that closes its MFMA segment.
  //   RAW: slab s+1 is retired by every wave's counted vmcnt in the load segment of phase (s,0); its first reads are
  //        issued in the load segment of phase (s,1), two barriers later for that wave's group, one for the other group.
  //   WAR: slab s+2 (issued from phase (s,0) on) overwrites the stage of slab s-1, whose last reads were issued in phase
  //        (s-1,0) and drained before that phase's closing barrier — two barriers earlier for the issuing group, one for
  //        the other.  The next chunk's patch goes into the other patch buffer one piece per step (taps 0-5).
  //   DMA issue order per step: phase 0: the first N0 instructions of slab s+2; phase 1: the rest, then a patch piece.
  //   vmcnt in phase (s,0), after its issues; oldest first: [slab s+1 ...][patch piece of s-1][first N0 of slab s+2].
  // The load segment has to be shorter than the MFMA segment it hides under, so its address arithmetic is hoisted: the
  // swizzle key of a patch read is (l15 + d) & 7 with d = (18 i + tap offset) & 7 a compile-time constant, so 8 x 2
  // precomputed lane addresses + an immediate serve all 72 reads of a chunk; the weight reads need 2.
  if (MODE == 1 && cc_begin < cc_end) {
    constexpr int N0 = (NW + 1) / 2;
    const int grp = wave_u >> 2;
#pragma unroll
    for (int i = 0; i < NPD; ++i) dma_patch_piece(cc_begin, 0, i);
    dma_weights(cc_begin, 0, 0);
    dma_weights(cc_begin, 1, 1);
    wait_vm(NW);
    bar();
    if (grp == 1) bar();

    int xaddr[8][2];                     // patch lane address for key offset d and k half (buffer 0; flipped per chunk)
#pragma unroll
    for (int d = 0; d < 8; ++d)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        xaddr[d][kk] = (wm * 4 * PW + l15) * 128 + (((kk * 4 + lq) ^ ((l15 + d) & 7)) << 4);
    int waddr[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) waddr[kk] = OFF_B + (wn * (BN / 2) + l15) * 128 + (((kk * 4 + lq) ^ (l15 & 7)) << 4);
    auto rd_x = [&](int tap, int kk, h16x8 (&xf)[MT]) {
      const int toff = (tap / 3) * PW + (tap % 3);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int rel = i * PW + toff;
        if (ABL != 4) xf[i] = *(const h16x8*)(smem + xaddr[rel & 7][kk] + rel * 128);
      }
    };
    auto rd_w = [&](int stage, int kk, h16x8 (&wf)[NT]) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
        if (ABL != 4) wf[j] = *(const h16x8*)(smem + waddr[kk] + stage * B_BYTES + j * 2048);
    };
    auto dma_w_part = [&](int cc, int tap, int stage, int i0, int i1) {     // instructions [i0, i1) of a slab's NW
      const unsigned dst = lds0 + OFF_B + stage * B_BYTES + wave_u * 1024;
      const int soff = (tap * g.Cin + cc * 64) * 2;
#pragma unroll
      for (int i = i0; i < i1; ++i) {
        if (i < BIT) dma16(srdB, b_voff[i], soff, dst + i * 8192);
        else dma16(srdB, b_voff_tail, soff, (wave_u < 4) ? (dst + BIT * 8192) : (lds0 + OFF_DUMMY + wave_u * 1024));
      }
    };

    h16x8 xa[MT], wa[NT], xb[MT], wb[NT];
    rd_x(0, 0, xa);
    rd_w(0, 0, wa);
    for (int cc = cc_begin; cc < cc_end; ++cc) {
      const int pbuf = (cc - cc_begin) & 1;
      const int hn = (cc + 1 < cc_end) ? 1 : 0;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const bool more_w = tap < 7 || hn;                    // slab s+2 exists
        const int wcc = tap < 7 ? cc : cc + 1, wtap = tap < 7 ? tap + 2 : tap - 7, wst = (tap + 2) % NWS;
        // phase 0: second half of this step's fragments, first share of the DMAs, counted wait
        rd_x(tap, 1, xb);
        rd_w(tap % NWS, 1, wb);
        if (ABL != 3 && more_w) dma_w_part(wcc, wtap, wst, 0, N0);
        wait_vm(more_w && ABL != 3 ? N0 : 0);
        bar();
        mma_phase(xa, wa);
        bar();
        // phase 1: first half of the next step's fragments, the rest of the DMAs
        if (tap == 8) {                                        // the next step reads the other patch buffer
          const int flip = pbuf ? -PBUF : PBUF;
#pragma unroll
          for (int d = 0; d < 8; ++d) { xaddr[d][0] += flip; xaddr[d][1] += flip; }
        }
        if (tap < 8 || hn) {
          rd_x((tap + 1) % 9, 0, xa);
          rd_w((tap + 1) % NWS, 0, wa);
        }
        if (ABL != 3 && more_w) dma_w_part(wcc, wtap, wst, N0, NW);
        if (ABL != 3 && tap < NPD && hn) dma_patch_piece(cc + 1, pbuf ^ 1, tap);
        bar();
        mma_phase(xb, wb);
        bar();
      }
    }
    if (grp == 0) bar();
  }

  // ---- merged ping-pong loop (MODE 2) ---------------------------------------------------------------------------------------------
  // A phase of the loop above is 16 (BN = 128) or 20 (BN = 160) MFMAs per wave — 256 / 320 matrix cycles — between two barriers,
  // against a load segment of 8-9 fragment reads + DMA issue that does not get shorter: the VAE's 128-channel convs at 512 x 512 ran
  // at 0.29-0.35 of the peak.  Here a step (tap) is ONE phase: both k halves, 32 / 40 MFMAs between two barriers, half the barriers.
  // The fragments of a step are read in the step's own load segment (one register set, not two): its latency runs beside the OTHER
  // group's MFMA segment, which is what the stagger is for.
  //   group 0:  [L(s)] P [M(s)] P [L(s+1)] ...      group 1 one barrier behind: its L(s) runs beside group 0's M(s).
  //   L(s): read the 2 (MT + NT) fragments of step s (slab s: own pieces retired by the counted vmcnt of L(s - 1), the others'
  //         published by the barriers since); DMA slab s + 2 -> stage (s + 2) % 3 (slab s - 1: its reads were drained — lgkmcnt(0) —
  //         before the barrier that closed L(s - 1) in BOTH groups); patch piece (taps 0-5) of the next chunk; vmcnt(own issues): slab
  //         s + 1 and the previous patch piece have landed; lgkmcnt(0).
  if (MODE == 2 && cc_begin < cc_end) {
    const int grp = wave_u >> 2;
    // HALO_NOPAD = 1: no padding DMAs (round 5, VERDICT r04 #5: the scheme of conv_halo_fp8.hip).  The counted vmcnt of a load segment only
    // has to leave THIS wave's just-issued DMAs in flight — a wave's counter sees its own operations, in order — so the waves need not
    // issue equal numbers: at BN = 160 the slab's last 32 rows are DMA'd by waves 0-3 alone, the 6th patch piece (32 slots) by wave 0
    // alone, instead of every other wave issuing those instructions with an out-of-range source into the sink region (10 of the 33 DMAs of
    // a 64-channel chunk in waves 4-7).  Built, bit-identical, and SLOWER on every BN = 160 shape of the UNet (same box, 28 clips,
    // profiles/r05_halo_nopad_ab.txt): 64^2 x 320 -> 320 0.187 -> 0.193 ms, 32^2 x 1920 -> 640 0.459 -> 0.478, 16^2 x 2560 -> 1280
    // 0.295 -> 0.311 (-3 ... -5 %); BN = 128 shapes unchanged.  A reading, not a measurement: with unequal DMA counts the two wave
    // groups' load segments differ in length; the second group's shorter segment cannot shorten the step (the barrier waits for the
    // first group's), it only shifts its MFMAs against the first group's.  The padding DMAs stay (HALO_NOPAD 0).
#ifndef HALO_NOPAD
#define HALO_NOPAD 0
#endif
    const bool tail_w = B_TAIL && wave_u < 4;              // this wave carries a piece of the slab's 32-row tail
    auto dma_w_own = [&](int cc, int tap, int stage) {
      if (!HALO_NOPAD) { dma_weights(cc, tap, stage); return; }
      const unsigned dst = lds0 + OFF_B + stage * B_BYTES + wave_u * 1024;
      const int soff = (tap * g.Cin + cc * 64) * 2;
#pragma unroll
      for (int i = 0; i < BIT; ++i) dma16(srdB, b_voff[i], soff, dst + i * 8192);
      if (tail_w) dma16(srdB, b_voff_tail, soff, dst + BIT * 8192);
    };
    auto dma_p_own = [&](int cc, int buf, int i) {          // patch piece i; the last piece exists on wave 0 only
      if (!HALO_NOPAD) { dma_patch_piece(cc, buf, i); return; }
      const unsigned dst = lds0 + buf * PBUF + wave_u * 1024;
      const unsigned voff = patch_voff(i);
      if (i < NPD - 1) dma16(srdA, voff, cc * 128, dst + i * 8192);
      else if (wave_u == 0 && lane < 32) dma16(srdA, voff, cc * 128, dst + (NPD - 1) * 8192);
    };
    auto dma_w_one = [&](int cc, int tap, int stage, int i) {   // instruction i (compile-time after unrolling) of the slab's NW
      const unsigned dst = lds0 + OFF_B + stage * B_BYTES + wave_u * 1024;
      const int soff = (tap * g.Cin + cc * 64) * 2;
      if (i < BIT) dma16(srdB, b_voff[i], soff, dst + i * 8192);
      else dma16(srdB, b_voff_tail, soff, (wave_u < 4) ? (dst + BIT * 8192) : (lds0 + OFF_DUMMY + wave_u * 1024));
    };
    const int km_cfg = min(__builtin_amdgcn_readfirstlane(g.pp_dma_m), NW + 1);
    // DMAs this wave issues for a slab / for patch piece i
    const int nw_own = HALO_NOPAD ? BIT + (tail_w ? 1 : 0) : NW;
    auto np_own = [&](int i) { return (!HALO_NOPAD || i < NPD - 1 || wave_u == 0) ? 1 : 0; };
#pragma unroll
    for (int i = 0; i < NPD; ++i) dma_p_own(cc_begin, 0, i);
    dma_w_own(cc_begin, 0, 0);
    dma_w_own(cc_begin, 1, 1);
    wait_vm(nw_own);                                      // patch and slab 0 have landed; slab 1 may be in flight

    bar();
    if (grp == 1) bar();

    int xaddr[8][2];
#pragma unroll
    for (int d = 0; d < 8; ++d)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        xaddr[d][kk] = (wm * 4 * PW + l15) * 128 + (((kk * 4 + lq) ^ ((l15 + d) & 7)) << 4);
    int waddr[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) waddr[kk] = OFF_B + (wn * (BN / 2) + l15) * 128 + (((kk * 4 + lq) ^ (l15 & 7)) << 4);
    auto rd_x = [&](int tap, int kk, h16x8 (&xf)[MT]) {
      const int toff = (tap / 3) * PW + (tap % 3);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int rel = i * PW + toff;
        xf[i] = *(const h16x8*)(smem + xaddr[rel & 7][kk] + rel * 128);
      }
    };
    auto rd_w = [&](int stage, int kk, h16x8 (&wf)[NT]) {
#pragma unroll
      for (int j = 0; j < NT; ++j) wf[j] = *(const h16x8*)(smem + waddr[kk] + stage * B_BYTES + j * 2048);
    };

    for (int cc = cc_begin; cc < cc_end; ++cc) {
      const int pbuf = (cc - cc_begin) & 1;
      const int hn = (cc + 1 < cc_end) ? 1 : 0;
      if (cc != cc_begin) {                                // this chunk's patch sits in the other buffer
        const int flip = pbuf ? PBUF : -PBUF;
#pragma unroll
        for (int d = 0; d < 8; ++d) { xaddr[d][0] += flip; xaddr[d][1] += flip; }
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const bool more_w = tap < 7 || hn;                    // slab s + 2 exists
        const int wcc = tap < 7 ? cc : cc + 1, wtap = (tap + 2) % 9;
        const bool pp = tap < NPD && hn;
        h16x8 x0[MT], x1[MT], w0[NT], w1[NT];
        rd_x(tap, 0, x0); rd_w(tap % NWS, 0, w0);
        rd_x(tap, 1, x1); rd_w(tap % NWS, 1, w1);
        // km of the step's DMA instructions (the slab's last ones, then the patch piece) ride among the MFMAs instead (round 6, as in
        // gemm_pp.hip): the load segment — 18 fragment reads + 3-4 DMA issues of ~110 cycles each — is longer than the 40-MFMA segment
        // it hides under and every barrier interval lasts max(load, matrix).  Issued later than before (behind this step's barrier) and
        // ahead of the next load segment's issues, whose counted wait therefore covers them: the RAW / WAR argument above is unchanged.
        const int kw = more_w ? (km_cfg < NW ? km_cfg : NW) : 0;          // weight instructions moved
        const bool p_in_m = pp && km_cfg > NW;                             // the patch piece moved
        if (HALO_NOPAD) {
          if (more_w) dma_w_own(wcc, wtap, (tap + 2) % NWS);
          if (pp) dma_p_own(cc + 1, pbuf ^ 1, tap);
          wait_vm((more_w ? nw_own : 0) + (pp ? np_own(tap) : 0));
        } else {
          if (more_w) {
#pragma unroll
            for (int i = 0; i < NW; ++i)
              if (i < NW - kw) dma_w_one(wcc, wtap, (tap + 2) % NWS, i);
          }
          if (pp && !p_in_m) dma_p_own(cc + 1, pbuf ^ 1, tap);
          wait_vm((more_w ? NW - kw : 0) + ((pp && !p_in_m) ? 1 : 0));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bar();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              acc[i][j] = hh == 0 ? MFMA_16x16x32(w0[j], x0[i], acc[i][j]) : MFMA_16x16x32(w1[j], x1[i], acc[i][j]);
              const int idx = (hh * MT + i) * NT + j;
              if (!HALO_NOPAD) {
#pragma unroll
                for (int o = 0; o <= NW; ++o)
                  if (idx == halo_slot_at(o, MT * NT)) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (o < NW) { if (kw > o) dma_w_one(wcc, wtap, (tap + 2) % NWS, NW - 1 - o); }
                    else if (p_in_m) dma_p_own(cc + 1, pbuf ^ 1, tap);
                    __builtin_amdgcn_sched_barrier(0);
                  }
              }
            }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        bar();
      }
    }
    if (grp == 0) bar();
  }

  // ---- lock-step main loop over steps s = (cc, tap): weight slab s lives in stage s % 3; the patch alternates per chunk
  // iteration s:  wait until slab s (and anything older) has landed, leaving only slab s+1 (and a just-issued patch)
  //               in flight -> barrier (everyone's pieces landed; stage (s+2)%3 and, at tap 0, the other patch buffer
  //               were last read in iteration s-1) -> issue patch(cc+1) [tap 0] and slab s+2 -> multiply slab s.
  if (!PP && cc_begin < cc_end) {
    const int nsteps = (cc_end - cc_begin) * 9;
    auto issue_w = [&](int sidx) {
      if (ABL == 3) return;
      if (sidx < nsteps) dma_weights(cc_begin + sidx / 9, sidx % 9, sidx % NWS);
    };
    dma_patch(cc_begin, 0);
    issue_w(0);
    issue_w(1);
    bool patch_just_issued = false;
    int s = 0;
    for (int cc = cc_begin; cc < cc_end; ++cc) {
      const int pbuf = (cc - cc_begin) & 1;
      for (int tap = 0; tap < 9; ++tap, ++s) {
        // outstanding, oldest first: [slab s] [patch (if issued last iteration)] [slab s+1]
        if (s + 1 >= nsteps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (patch_just_issued) {
          if (NW == 3) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
          if (NW == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
        if (ABL != 6) __builtin_amdgcn_s_barrier();
        patch_just_issued = false;
        if (tap == 0 && cc + 1 < cc_end && ABL != 3) { dma_patch(cc + 1, pbuf ^ 1); patch_just_issued = true; }
        issue_w(s + 2);
        const int ky = tap / 3, kx = tap - ky * 3;
        compute(pbuf, s % NWS, ky * PW + kx);
      }
    }
    __syncthreads();
  }

  // ---- epilogue -----------------------------------------------------------------------------------------------
  if (ABL == 1 && acc[0][0][0] != 12345.f) return;
  const int mr = (b * OH + y0 + wm * 4) * OW + x0 + l15;        // image row i of the wave adds i * OW
  const int nc = n0 + wn * (BN / 2) + lq * 4;
  if (g.splitk > 1) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        // (PERM: the quads of a tile pair are columns 32 (j >> 1) + 8 lq + 4 (j & 1) .. + 3 of the wave's run, the unpaired tile natural)
        const int n = (PERM && j < (NT & ~1)) ? n0 + wn * (BN / 2) + 32 * (j >> 1) + 8 * lq + 4 * (j & 1) : nc + j * 16;
        if (n < g.N) *(f32x4*)(g.slabs + ((int64_t)ks * g.M + mr + i * OW) * g.N + n) = acc[i][j];
      }
  } else {
    // 16-bit output, no GEGLU: the wide epilogue — at BN = 128 through lane exchanges; at BN = 160 (NT = 5, the kernel sits at 254 VGPRs, the
    // exchange form spills 38 registers) through the permuted weight rows above (round 5)
    epi_tile<MT, NT, false, (NT % 2 == 0) ? 1 : 2>(g, 0, mr, OW, nc, acc, smem, 4, wm, wn, tile_m, n0);
  }
}

template <int BN, int MODE>
void launch_halo(const GemmArgs& g, dim3 grid, hipStream_t s) {
  constexpr int smem = 2 * PPIX * 128 + 3 * BN * 128 + 8192;
  hipLaunchKernelGGL((conv_halo_kernel<BN, MODE>), grid, dim3(512), smem, s, g);
}

}  // namespace

int conv_halo_bn(const GemmArgs& g);

// stride-1 3x3 convs on images whose sides are multiples of 16, with enough 16x16-pixel blocks x channel tiles to give
// every CU a workgroup (below that the 128-row implicit GEMM with its split-K is faster: same-box A/B at 16x16 images)
bool conv_halo_supported(const GemmArgs& g) {
  static const int off = getenv("SVG_NO_HALO") ? atoi(getenv("SVG_NO_HALO")) : 0;
  static const int up_on = getenv("SVG_HALO_UP2") ? atoi(getenv("SVG_HALO_UP2")) : 1;
  const bool s1 = g.amode == A_CONV_S1 && g.Ho == g.H && g.Wo == g.W;
  const bool up = up_on && g.amode == A_CONV_UP2 && g.Ho == 2 * g.H && g.Wo == 2 * g.W && !g.A2;
  if (off || !(s1 || up) || g.Cin % 64 != 0 || g.Ho % 16 != 0 || g.Wo % 16 != 0 || g.batch != 1 || g.out_f32 ||
      g.act == ACT_GEGLU || g.N < 128)
    return false;
  const int min_wg = (int)svg_env_i64("SVG_HALO_MIN", 192);      // (cached lookup; svg_env_refresh re-reads it: the parity tests force the kernel at batch 1-2)
  return (int64_t)(g.M / 256) * cdiv(g.N, conv_halo_bn(g)) >= min_wg;
}

// channel-tile width: fewest serial rounds of workgroups (one per CU) times tile width; ties -> fewer padded columns
int conv_halo_bn(const GemmArgs& g) {
  const int64_t pb = g.M / 256;
  int best = 128;
  int64_t best_cost = -1, best_pad = 0;
  for (int bn : {128, 160}) {
    const int64_t tn = cdiv(g.N, bn);
    const int64_t cost = ((pb * tn + 255) / 256) * bn, pad = tn * bn - g.N;
    if (best_cost < 0 || cost < best_cost || (cost == best_cost && pad < best_pad)) { best = bn; best_cost = cost; best_pad = pad; }
  }
  return best;
}

void conv_halo_init_device() {
  HIP_OK(hipFuncSetAttribute((const void*)conv_halo_kernel<128, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PPIX * 128 + 3 * 128 * 128 + 8192));
  HIP_OK(hipFuncSetAttribute((const void*)conv_halo_kernel<128, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PPIX * 128 + 3 * 128 * 128 + 8192));
  HIP_OK(hipFuncSetAttribute((const void*)conv_halo_kernel<128, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PPIX * 128 + 3 * 128 * 128 + 8192));
  HIP_OK(hipFuncSetAttribute((const void*)conv_halo_kernel<160, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PPIX * 128 + 3 * 160 * 128 + 8192));
  HIP_OK(hipFuncSetAttribute((const void*)conv_halo_kernel<160, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PPIX * 128 + 3 * 160 * 128 + 8192));
  HIP_OK(hipFuncSetAttribute((const void*)conv_halo_kernel<160, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PPIX * 128 + 3 * 160 * 128 + 8192));
}

void launch_conv_halo(const GemmArgs& g, dim3 grid, hipStream_t s) {
  static const int pp = getenv("SVG_HALO_PP") ? atoi(getenv("SVG_HALO_PP")) : 1;
  const bool w160 = conv_halo_bn(g) == 160;
  // merged ping-pong (one 32 / 40-MFMA phase per tap); SVG_HALO_MERGE (cached, svg_env_refresh): 0 off, 1 BN = 128 only, 2 / unset both widths
  const int mm = (int)svg_env_i64("SVG_HALO_MERGE", 2);
  if (pp && mm >= 1 && (mm >= 2 || !w160)) {
    GemmArgs gm = g;
    gm.pp_dma_m = std::max(0, std::min(4, (int)svg_env_i64("SVG_HALO_DMA_M", 4)));   // DMA instructions of a step issued among its MFMAs (merged loop)
    if (w160) launch_halo<160, 2>(gm, grid, s); else launch_halo<128, 2>(gm, grid, s);
    return;
  }
  if (false) {}
  else if (pp) { if (w160) launch_halo<160, 1>(g, grid, s); else launch_halo<128, 1>(g, grid, s); }
  else    { if (w160) launch_halo<160, 0>(g, grid, s); else launch_halo<128, 0>(g, grid, s); }
}

}  // namespace SDNS
