// Halo 3x3 convolution (stride 1, pad 1) in OCP MX block-scaled fp8 for gfx950 — BASELINE configs[4] ("fp8 MFMA on CDNA4") on the
// convolutions, where K is long (K = 9 Cin = 2 880 ... 23 040) and the matrix pipe is the limit (VERDICT r03 #4, row g).
//
// Same data movement as conv_halo.hip's merged ping-pong loop (MODE 2), at half the bytes per channel:
//   * activations: e4m3 NHWC [pixel][Cp] (Cp = Cin rounded up to 128, padding zero) + one E8M0 scale per 32 channels
//     [pixel][Cp / 32]; a "chunk" is 128 channels = 128 B per pixel, so the 18 x 18 halo patch of a 16 x 16 pixel block is the
//     same 324 x 128 B LDS image, DMA'd once per chunk, all nine taps read from it at shifted pixel addresses;
//   * weights: e4m3 [N][tap][Cp] + E8M0 [tap][chunk][N][4]; a slab is BN x 128 B per (tap, chunk), three LDS stages;
//   * v_mfma_scale_f32_16x16x128_f8f6f4: lane (r = lane & 15, q = lane >> 4) supplies the 16-byte chunks q and q + 4 of row r's
//     128 bytes and the scale of block q (channels [32 q, 32 q + 32): operand map probed with exact integer data,
//     tools/probe/probe_mx.hip) — i.e. exactly the two fragment reads per row the fp16 kernel issues per 64-channel chunk, here
//     covering 128 channels; the instruction takes 32 cycles against 2 x 16 for the two fp16 MFMAs of HALF that K: twice the rate;
//   * the scales travel through LDS as well: one 4-byte DMA per patch pixel and chunk (the patch's seventh piece), one per weight
//     row and slab; a lane reads its byte with ds_read_u8.
// 8 waves as 4 (pixel rows) x 2 (channels), two groups of four staggered by one barrier: while one group multiplies (MT x NT =
// 16 / 20 MFMAs of 32 cycles between two barriers) the other reads the next step's fragments and issues its DMAs.
// Epilogue: the shared tile epilogue (bias, per-sample time-embedding bias, residual, GroupNorm column sums, h16 store).
#include "igemm_epi.h"
#include <cstdlib>

namespace SDNS {

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));

constexpr int PW8 = 18;
constexpr int PPIX8 = PW8 * PW8;       // 324
constexpr int NPD8 = 6;                // 16-byte patch pieces per wave (as conv_halo.hip); piece 6 is the scale piece

struct Fp8ConvArgs {
  GemmArgs g;                          // M, N, H, W, Ho, Wo, epilogue (bias, bias_bn, residual, gn_part, C, ldc), tn_major; A / Wt / K unused
  const uint8_t* A8; const uint8_t* As;     // [B][H][W][Cp] e4m3, [B][H][W][Cp / 32] E8M0
  const uint8_t* W8; const uint8_t* Ws;     // [Npad][9][Cp] e4m3, [9][Cp / 128][Npad][4] E8M0
  int Cp, Npad;
};

// LDS-direct 4-byte buffer load (one dword per lane: LDS address = M0 + 4 * lane); see dma16 in igemm_epi.h
__device__ __forceinline__ void dma4(v4i srd, unsigned voff, int soff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(srd), "s"(soff), "s"(lds_addr)
               : "memory");
}

// In-kernel stamps (diagnostic build only: -DFP8_STAMP; cdna_hip_programming.md section 7): where a step of the ping-pong loop spends
// its cycles — per wave the sums over all steps of {fragment reads + DMA issue, counted DMA wait, barrier into the MFMA segment, the
// MFMA segment, barrier out of it}, written to a(g.slabs)[workgroup][wave][8] and printed by the launcher.
#ifdef FP8_STAMP
#define STAMP(v) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(v) do { } while (0)
#endif

constexpr int fp8_ws_bytes(int bn) { return (9 * bn * 4 + 1023) / 1024 * 1024; }

template <int BN>
__global__ void __launch_bounds__(512, 2) conv_halo_fp8_kernel(const Fp8ConvArgs a) {
  constexpr int NT = BN / 32;            // 16-wide channel tiles per wave
  constexpr int MT = 4;                  // image rows per wave
  constexpr int BIT = BN / 64;           // 64-row weight groups: one 16-byte DMA per wave each
  constexpr int B_BYTES = BN * 128;
  constexpr int PBUF = PPIX8 * 128;      // 41472
  constexpr int NWS = 3;
  constexpr bool B_TAIL = (BN % 64) != 0;          // BN = 160: a last group of 32 rows, served by waves 0-3
  constexpr int PS_BYTES = 6 * 256;                // patch scales: 324 pixels x 4 B, one dword DMA of waves 0-5
  // weight scales of a chunk: [tap][row][4 B], padded to whole 1-KiB DMA pieces — the last participating wave issues a full 64-lane DMA whose
  // surplus lanes (INVALID offset) write zeros: they must land inside the buffer, not on the next one or past the allocation
  constexpr int WS_BYTES = fp8_ws_bytes(BN);
  constexpr int OFF_B = 2 * PBUF, OFF_PS = OFF_B + NWS * B_BYTES, OFF_WS = OFF_PS + 2 * PS_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr unsigned INVALID = 0x80000000u;
  const GemmArgs& g = a.g;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int wave_u = __builtin_amdgcn_readfirstlane(wid);
  const int l15 = lane & 15, lq = lane >> 4;

  const int tiles_n = (g.N + BN - 1) / BN;
  // nearest-2x upsample fused in front (A_CONV_UP2, the UNet upsamplers): the conv runs over the OH x OW upsampled image, patch pixel (yy, xx)
  // of it is source pixel (yy >> 1, xx >> 1) — only the DMA source offsets (data and scales) change, as in conv_halo.hip
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
  if (g.tn_major) {
    const int npb = gridDim.x / tiles_n;
    tn = tile / npb; pb = tile - tn * npb;
  } else {
    tn = tile % tiles_n; pb = tile / tiles_n;
  }
  const int tile_m = pb;
  const int bx = pb % bx_n; pb /= bx_n;
  const int by = pb % by_n;
  const int b = pb / by_n;
  const int y0 = by << 4, x0 = bx << 4, n0 = tn * BN;

  const int Cp = a.Cp, CC = Cp >> 7, SB = Cp >> 5;          // chunks of 128 channels; scale bytes per pixel
  const int nimg = g.M / (OH * OW);
  const unsigned a_bytes = (unsigned)((int64_t)nimg * g.H * g.W * Cp);
  const unsigned as_bytes = (unsigned)((int64_t)nimg * g.H * g.W * SB);
  const unsigned w_bytes = (unsigned)((int64_t)a.Npad * 9 * Cp);
  const unsigned ws_bytes = (unsigned)((int64_t)9 * CC * a.Npad * 4);
  auto mk = [](const void* p, unsigned bytes) -> v4i {
    const uint64_t u = (uint64_t)p;
    return v4i{(int)(unsigned)u, (int)((u >> 32) & 0xffff), (int)bytes, 0x00020000};
  };
  const v4i srdA = mk(a.A8, a_bytes), srdAs = mk(a.As, as_bytes), srdB = mk(a.W8, w_bytes), srdBs = mk(a.Ws, ws_bytes);
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);

  // A load segment waits for everything but the BIT weight DMAs it has just issued (the same count in every wave), so the DMAs issued
  // in the matrix segments need not be equally many per wave: no padding DMAs (and no sink region) as in conv_halo.hip's loops.
  // ---- patch pieces of a chunk: 0-4 all waves (8 KiB each), 5 wave 0's lanes 0-31 (slots 2560-2591), 6 = the scale bytes (waves 0-5)
  auto patch_voff = [&](int i) -> unsigned {
    int ln = lane;
    asm volatile("" : "+v"(ln));                        // keeps LICM from hoisting the offsets into registers this kernel does not have
    const int id = (i * 8 + wid) * 64 + ln;
    const int pp = id >> 3, pos = id & 7;
    const int c = pos ^ (pp & 7);
    const int py = pp / PW8, px = pp - py * PW8;
    const int yy = y0 - 1 + py, xx = x0 - 1 + px;
    const bool ok = pp < PPIX8 && (unsigned)yy < (unsigned)OH && (unsigned)xx < (unsigned)OW;
    const int sy = up2 ? yy >> 1 : yy, sx = up2 ? xx >> 1 : xx;
    return ok ? (unsigned)(((b * g.H + sy) * g.W + sx) * Cp + c * 16) : INVALID;
  };
  // source offset of this lane's share of piece i (computed in a load segment: ~40 VALU instructions with an integer division that
  // must not sit between the MFMAs of a matrix segment, where nothing else of this wave can cover them)
  auto patch_piece_voff = [&](int i) -> unsigned {
    if (i < NPD8) return patch_voff(i);
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int pp = wid * 64 + ln;                         // the scale piece: slot = patch pixel
    const int py = pp / PW8, px = pp - py * PW8;
    const int yy = y0 - 1 + py, xx = x0 - 1 + px;
    const bool ok = pp < PPIX8 && (unsigned)yy < (unsigned)OH && (unsigned)xx < (unsigned)OW;
    const int sy = up2 ? yy >> 1 : yy, sx = up2 ? xx >> 1 : xx;
    return ok ? (unsigned)(((b * g.H + sy) * g.W + sx) * SB) : INVALID;      // pixels outside the image read zeros (2^-127 beside zero data)
  };
  auto dma_patch_issue = [&](int cc, int buf, int i, unsigned voff) {
    if (i < NPD8 - 1) {
      dma16(srdA, voff, cc * 128, lds0 + buf * PBUF + wave_u * 1024 + i * 8192);
    } else if (i == NPD8 - 1) {
      if (wave_u == 0) {
        if (lane < 32) dma16(srdA, voff, cc * 128, lds0 + buf * PBUF + i * 8192);
      }
    } else if (wave_u < 6) {
      dma4(srdAs, voff, cc * 4, lds0 + OFF_PS + buf * PS_BYTES + wave_u * 256);
    }
  };
  auto dma_patch_piece = [&](int cc, int buf, int i) { dma_patch_issue(cc, buf, i, patch_piece_voff(i)); };
  // ---- weight slab (tap, chunk): rows r0 + 64 i, chunk swizzled on the source
  const int r0 = tid >> 3;
  const int cB = (tid & 7) ^ (r0 & 7);
  const int ldb8 = 9 * Cp;
  unsigned b_voff[BIT];
#pragma unroll
  for (int i = 0; i < BIT; ++i) {
    const int n = n0 + r0 + 64 * i;
    b_voff[i] = (n < a.Npad) ? (unsigned)(n * ldb8 + cB * 16) : INVALID;
  }
  unsigned b_voff_tail = INVALID;
  if (B_TAIL) {
    const int n = n0 + BIT * 64 + r0;
    if (r0 < 32 && n < a.Npad) b_voff_tail = (unsigned)(n * ldb8 + cB * 16);
  }
  auto dma_w_main = [&](int cc, int tap, int stage, int i) {
    dma16(srdB, b_voff[i], tap * Cp + cc * 128, lds0 + OFF_B + stage * B_BYTES + wave_u * 1024 + i * 8192);
  };
  auto dma_w_tail = [&](int cc, int tap, int stage) {
    if (B_TAIL && wave_u < 4) dma16(srdB, b_voff_tail, tap * Cp + cc * 128, lds0 + OFF_B + stage * B_BYTES + wave_u * 1024 + BIT * 8192);
  };
  // ---- weight scales of a whole chunk in one 16-byte DMA per lane: lane -> (tap, 4 rows); LDS image [tap][row][4 B]
  unsigned bs_voff = INVALID;
  {
    const int id = wid * 64 + lane;                       // < 9 * BN / 4 (= 360 for BN 160: waves 0-5)
    const int tap = id / (BN / 4), quad = id - tap * (BN / 4);
    if (tap < 9 && n0 + quad * 4 < a.Npad) bs_voff = (unsigned)((tap * CC * a.Npad + n0 + quad * 4) * 4);
  }
  auto dma_w_scales = [&](int cc, int buf) {
    if (wave_u * 64 < 9 * (BN / 4)) dma16(srdBs, bs_voff, cc * a.Npad * 4, lds0 + OFF_WS + buf * WS_BYTES + wave_u * 1024);
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  unsigned long long st_rd = 0, st_wait = 0, st_b1 = 0, st_mm = 0, st_b2 = 0, st_t0 = 0, st_t1 = 0, st_steps = 0;
  STAMP(st_t0);
  if (CC > 0 && g.dbg != 2) {
    const int grp = wave_u >> 2;
#pragma unroll
    for (int i = 0; i <= NPD8; ++i) dma_patch_piece(0, 0, i);
    dma_w_scales(0, 0);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int i = 0; i < BIT; ++i) dma_w_main(0, t, t, i);
      dma_w_tail(0, t, t);
    }
    wait_vm(0);
    bar();
    if (grp == 1) bar();

    // patch lane address of chunk lq for key offset d (buffer 0; flipped per chunk); chunk lq + 4 of the same row sits at that
    // address ^ 64 (the swizzle XORs the chunk index, and (lq + 4) ^ key == (lq ^ key) ^ 4): one v_xor per read instead of a second
    // table of 8 registers — this kernel has none to spare
    int xaddr[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) xaddr[d] = (wm * 4 * PW8 + l15) * 128 + ((lq ^ ((l15 + d) & 7)) << 4);
    const int waddr = OFF_B + (wn * (BN / 2) + l15) * 128 + ((lq ^ (l15 & 7)) << 4);
    const int waddr1 = waddr ^ 64;
    int xs_addr = OFF_PS + (wm * 4 * PW8 + l15) * 4 + lq;                 // scale byte of (pixel, block lq), buffer 0
    int ws_addr = OFF_WS + (wn * (BN / 2) + l15) * 4 + lq;                // scale byte of (weight row, block lq), tap 0, buffer 0

    // Step s = (chunk, tap).  Group 0:  [L(s)] P [M(s)] P [L(s+1)] ...; group 1 runs one barrier behind, so its load segment L
    // lies beside group 0's matrix segment M and vice versa.
    //   L(s): read the 2 (MT + NT) fragments + MT + NT scale bytes of step s; issue the BIT main weight DMAs of slab s + 2 (stage
    //         (s + 2) % 3, whose last readers — L(s - 1) of both groups — drained their reads before the previous barrier of each);
    //         vmcnt(BIT): everything this wave issued before them — L(s - 1), M(s - 1) — has landed; lgkmcnt(0); barrier.
    //   M(s): MT x NT scaled MFMAs with the remaining DMAs of the step between their rows: the tail rows of slab s + 2 (waves
    //         0-3, BN 160), the next chunk's patch piece (taps 0-6: into the other patch buffer) and, at tap 7, the next chunk's
    //         weight scales.  They are retired by the counted wait of L(s + 1).
    //   RAW: slab s + 2 is complete in LDS for a reader at L(s + 2): every wave's pieces were waited for in L(s + 1) at the latest,
    //        at least one barrier before any wave's L(s + 2).  WAR: see the stage argument above; patch / scale buffers alternate
    //        per chunk and are rewritten from M(cc, 0) on, after the last reads of chunk cc - 1.
    // Why the split: an LDS-DMA issue costs ~120 cycles in a load segment and ~90 among MFMAs (stamped, profiles/r04_fp8_conv_stamps.txt);
    // with all five in L the segment took 1144 cycles against 700 of MFMAs, with all five in M 531 against 1140: the matrix pipe idles
    // for the difference either way.  Two in L and the rest in M balance the two segments.
    for (int cc = 0; cc < CC; ++cc) {
      const int pbuf = cc & 1;
      const int hn = (cc + 1 < CC) ? 1 : 0;
      if (cc != 0) {
        const int flip = pbuf ? PBUF : -PBUF;
#pragma unroll
        for (int d = 0; d < 8; ++d) xaddr[d] += flip;
        xs_addr += pbuf ? PS_BYTES : -PS_BYTES;
        ws_addr += pbuf ? WS_BYTES : -WS_BYTES;
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const bool more_w = tap < 7 || hn;
        const int wcc = tap < 7 ? cc : cc + 1, wtap = (tap + 2) % 9, wst = (tap + 2) % NWS;
        const bool pp = tap <= NPD8 && hn;
        const int toff = (tap / 3) * PW8 + (tap % 3);
        const int stage = tap % NWS;
        v8i xv[MT], wv[NT];
        int sx[MT], sw[NT];
        unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0;
        STAMP(s0);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int rel = i * PW8 + toff;
          // (rel * 128 leaves bit 6 alone: the XOR applies to the table entry, the row offset stays an immediate.  Opaque to the
          // optimiser on purpose: left to itself it hoists all 36 XORed addresses of a chunk out of the tap loop and spills)
          int a1;
          asm volatile("v_xor_b32 %0, 64, %1" : "=v"(a1) : "v"(xaddr[rel & 7]));
          const v4i lo = *(const v4i*)(smem + xaddr[rel & 7] + rel * 128);
          const v4i hi = *(const v4i*)(smem + a1 + rel * 128);
          xv[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          sx[i] = *(const uint8_t*)(smem + xs_addr + rel * 4);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const v4i lo = *(const v4i*)(smem + waddr + stage * B_BYTES + j * 2048);
          const v4i hi = *(const v4i*)(smem + waddr1 + stage * B_BYTES + j * 2048);
          wv[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          sw[j] = *(const uint8_t*)(smem + ws_addr + tap * (BN * 4) + j * 64);
        }
        if (more_w) {
#pragma unroll
          for (int i = 0; i < BIT; ++i) dma_w_main(wcc, wtap, wst, i);
        }
        unsigned pvoff = INVALID;
        if (pp) pvoff = patch_piece_voff(tap);
        STAMP(s1);
        wait_vm(more_w ? BIT : 0);                          // all but the BIT DMAs just issued (every wave issues exactly BIT here)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        STAMP(s2);
        bar();
        STAMP(s3);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv[j], xv[i], acc[i][j], 0, 0, 0, sw[j], 0, sx[i]);
          __builtin_amdgcn_sched_barrier(0);
          if (i == 0 && more_w) dma_w_tail(wcc, wtap, wst);
          if (i == 1 && pp) dma_patch_issue(cc + 1, pbuf ^ 1, tap, pvoff);
          if (i == 2 && tap == 7 && hn) dma_w_scales(cc + 1, pbuf ^ 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        // pin the products to this segment: hipcc otherwise SINKS the scaled MFMAs of all nine taps below the loop's barriers (their
        // results are only needed by the next tap's accumulation) and spills every fragment it has to keep alive for them
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(acc[i][j]));
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(s4);
        bar();
        STAMP(s5);
        st_rd += s1 - s0; st_wait += s2 - s1; st_b1 += s3 - s2; st_mm += s4 - s3; st_b2 += s5 - s4; st_steps += 1;
      }
    }
    if (grp == 0) bar();
  }
  STAMP(st_t1);

  const int mr = (b * OH + y0 + wm * 4) * OW + x0 + l15;
  const int nc = n0 + wn * (BN / 2) + lq * 4;
  if (g.dbg == 1 && acc[0][0][0] != 12345.f) return;       // ablation (SVG_FP8_DBG=1): no epilogue
  epi_tile<MT, NT, false, true>(g, 0, mr, OW, nc, acc, smem, 4, wm, wn, tile_m, n0);
#ifdef FP8_STAMP
  unsigned long long st_t2 = 0;
  STAMP(st_t2);
  if (g.slabs && lane == 0) {
    unsigned long long* o = (unsigned long long*)g.slabs + ((size_t)blockIdx.x * 8 + wid) * 8;
    o[0] = st_rd; o[1] = st_wait; o[2] = st_b1; o[3] = st_mm; o[4] = st_b2; o[5] = st_t1 - st_t0; o[6] = st_t2 - st_t1; o[7] = st_steps;
  }
#endif
}

template <int BN>
constexpr int fp8_halo_smem() { return 2 * PPIX8 * 128 + 3 * BN * 128 + 2 * 6 * 256 + 2 * fp8_ws_bytes(BN); }

// ---- quantisers --------------------------------------------------------------------------------------------------------------
// activations: x [P][C] h16 -> q [P][Cp] e4m3 + sc [P][Cp / 32] E8M0 (OCP MX v1.0 section 6.3, like quant_mx_kernel); one thread per
// 8-channel vector, four adjacent threads share a block of 32; the padding vectors (channels >= C) are written as zeros with scale 1
__global__ void __launch_bounds__(256) quant_act_mx_kernel(const h16* __restrict__ x, int C, uint8_t* __restrict__ q, uint8_t* __restrict__ sc,
                                                           int64_t P, int Cp) {
  const int CVp = Cp >> 3;
  const int64_t total = P * CVp;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < ((total + 3) & ~3LL); t += (int64_t)gridDim.x * blockDim.x) {
    const bool live = t < total;
    const int64_t p = live ? t / CVp : 0;
    const int cv = live ? (int)(t - p * CVp) : 0;
    float v[8];
    const bool real = live && cv * 8 < C;
    if (real) {
      const h16x8 h = *(const h16x8*)(x + p * C + cv * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (float)h[j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
    float am = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) am = fmaxf(am, fabsf(v[j]));
    am = fmaxf(am, __shfl_xor(am, 1));
    am = fmaxf(am, __shfl_xor(am, 2));
    int e = am > 0.f ? ((__float_as_int(am) >> 23) & 0xff) - 127 - 8 : 0;
    e = e < -127 ? -127 : (e > 126 ? 126 : e);
    const float inv = __int_as_float((127 - e) << 23);
    unsigned lo = 0, hi = 0;
    {
      float s[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] = fminf(fmaxf(v[j] * inv, -448.f), 448.f);
      lo = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(s[0], s[1], 0, false) & 0xffffu;
      lo |= ((unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(s[2], s[3], 0, false) & 0xffffu) << 16;
      hi = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(s[4], s[5], 0, false) & 0xffffu;
      hi |= ((unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(s[6], s[7], 0, false) & 0xffffu) << 16;
    }
    if (live) {
      *(uint2*)(q + p * Cp + cv * 8) = make_uint2(lo, hi);
      if ((cv & 3) == 0) sc[p * (Cp >> 5) + (cv >> 2)] = (uint8_t)(e + 127);
    }
  }
}

// weights: f32 OIHW -> e4m3 [Npad][tap][Cp] + E8M0 [tap][Cp / 128][Npad][4]; one thread per (n, tap, block of 32 channels)
__global__ void __launch_bounds__(256) pack_conv3x3_mx_kernel(const float* __restrict__ w, uint8_t* __restrict__ q, uint8_t* __restrict__ sc,
                                                              int O, int I, int Npad, int Cp) {
  const int KB = Cp >> 5;
  const int64_t total = (int64_t)Npad * 9 * KB;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int kb = (int)(t % KB);
    const int tap = (int)((t / KB) % 9);
    const int n = (int)(t / ((int64_t)KB * 9));
    float v[32];
    float am = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int c = kb * 32 + j;
      v[j] = (n < O && c < I) ? w[((int64_t)n * I + c) * 9 + tap] : 0.f;
      am = fmaxf(am, fabsf(v[j]));
    }
    int e = am > 0.f ? ((__float_as_int(am) >> 23) & 0xff) - 127 - 8 : 0;
    e = e < -127 ? -127 : (e > 126 ? 126 : e);
    const float inv = __int_as_float((127 - e) << 23);
    uint8_t* dst = q + ((int64_t)n * 9 + tap) * Cp + kb * 32;
#pragma unroll
    for (int j = 0; j < 32; j += 2) {
      const float s0 = fminf(fmaxf(v[j] * inv, -448.f), 448.f), s1 = fminf(fmaxf(v[j + 1] * inv, -448.f), 448.f);
      *(uint16_t*)(dst + j) = (uint16_t)(__builtin_amdgcn_cvt_pk_fp8_f32(s0, s1, 0, false) & 0xffff);
    }
    sc[(((int64_t)tap * (Cp >> 7) + (kb >> 2)) * Npad + n) * 4 + (kb & 3)] = (uint8_t)(e + 127);
  }
}

}  // namespace

int conv_halo_fp8_bn(int64_t M, int N) {
  const int64_t pb = M / 256;
  int best = 128;
  int64_t best_cost = -1, best_pad = 0;
  for (int bn : {128, 160}) {
    const int64_t tn = cdiv(N, bn);
    const int64_t cost = ((pb * tn + 255) / 256) * bn, pad = tn * bn - N;
    if (best_cost < 0 || cost < best_cost || (cost == best_cost && pad < best_pad)) { best = bn; best_cost = cost; best_pad = pad; }
  }
  return best;
}

// stride-1 3x3 convs on images whose sides are multiples of 16 with at least one workgroup per most CUs (as conv_halo_supported)
// H, W: the OUTPUT image (2x the input under the fused nearest upsample)
bool conv_halo_fp8_supported(int B, int H, int W, int Cin, int N) {
  const int min_wg = (int)svg_env_i64("SVG_HALO_MIN", 192);      // (cached lookup; svg_env_refresh re-reads it: the parity tests force the kernel at batch 1-2)
  if (Cin % 64 != 0 || H % 16 != 0 || W % 16 != 0 || N < 128 || N % 4 != 0) return false;
  const int64_t M = (int64_t)B * H * W;
  const int64_t Cp = align_up(Cin, 128);
  if (M * Cp >= (1LL << 31) || (int64_t)N * 9 * Cp >= (1LL << 31) || M * N >= (1LL << 31)) return false;
  return (M / 256) * cdiv(N, conv_halo_fp8_bn(M, N)) >= min_wg;
}

void conv_halo_fp8_init_device() {
  HIP_OK(hipFuncSetAttribute((const void*)conv_halo_fp8_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, fp8_halo_smem<128>()));
  HIP_OK(hipFuncSetAttribute((const void*)conv_halo_fp8_kernel<160>, hipFuncAttributeMaxDynamicSharedMemorySize, fp8_halo_smem<160>()));
}

void quant_act_mx(svg_ctx* ctx, const h16* x, int C, uint8_t* q, uint8_t* sc, int64_t P, hipStream_t s) {
  SVG_CHECK(C % 32 == 0, "quant_act_mx: C=%d must be a multiple of 32", C);
  if (!SVG_LAUNCHING(ctx)) return;
  const int Cp = (int)align_up(C, 128);
  char tag[64];
  snprintf(tag, sizeof(tag), "quant_act_mx_P%lld_C%d", (long long)P, C);
  ProfScope ps(ctx, PK_ELT, s, 0, (double)P * (2.0 * C + Cp + Cp / 32), tag);
  const int64_t threads = P * (Cp / 8);
  hipLaunchKernelGGL(quant_act_mx_kernel, dim3((unsigned)std::min<int64_t>((threads + 255) / 256, 65536)), dim3(256), 0, s, x, C, q, sc, P, Cp);
  check_launch("quant_act_mx");
}

void pack_conv3x3_mx(const float* w_oihw, uint8_t* q, uint8_t* sc, int O, int I, int Npad, int Cp, hipStream_t s) {
  const int64_t total = (int64_t)Npad * 9 * (Cp / 32);
  hipLaunchKernelGGL(pack_conv3x3_mx_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65536)), dim3(256), 0, s, w_oihw, q, sc, O, I, Npad, Cp);
  check_launch("pack_conv3x3_mx");
}

// g: the GemmArgs conv3x3() builds for the fp16 conv (M, N, H, W, Ho, Wo, Cin, epilogue); operands in MX fp8
void conv_halo_fp8(svg_ctx* ctx, const uint8_t* A8, const uint8_t* As, const uint8_t* W8, const uint8_t* Ws, int Npad, const GemmArgs& g0, hipStream_t s) {
  const int B = g0.M / (g0.Ho * g0.Wo);
  const bool s1 = g0.amode == A_CONV_S1 && g0.Ho == g0.H && g0.Wo == g0.W, up = g0.amode == A_CONV_UP2 && g0.Ho == 2 * g0.H && g0.Wo == 2 * g0.W;
  SVG_CHECK((s1 || up) && conv_halo_fp8_supported(B, g0.Ho, g0.Wo, g0.Cin, g0.N) && !g0.out_f32 &&
            g0.act == ACT_NONE && !g0.ln_rs, "conv_halo_fp8: %dx%d Cin %d N %d unsupported", g0.H, g0.W, g0.Cin, g0.N);
  if (!SVG_LAUNCHING(ctx)) return;
  Fp8ConvArgs a;
  a.g = g0;
  a.g.splitk = 1; a.g.n_valid = g0.N;
  a.g.dbg = (int)svg_env_i64("SVG_FP8_DBG", 0);             // ablations: 1 no epilogue, 2 no main loop, 3 no GroupNorm sums, 4 no residual
  if (a.g.dbg == 3) a.g.gn_part = nullptr;
  if (a.g.dbg == 4) a.g.residual = nullptr;
  a.A8 = A8; a.As = As; a.W8 = W8; a.Ws = Ws;
  a.Cp = (int)align_up(g0.Cin, 128); a.Npad = Npad;
  const int bn = conv_halo_fp8_bn(g0.M, g0.N);
  const int tiles_n = cdiv(g0.N, bn);
  // weight-heavy (small images, weights larger than the activations): keep one channel tile's weights in an XCD's L2 (as gemm_auto)
  a.g.tn_major = ((int64_t)g0.N * 9 > (int64_t)g0.M) ? 1 : 0;
  char tag[112];
  snprintf(tag, sizeof(tag), "conv_fp8%s_B%d_%dx%d_Cin%d_Cout%d_res%d", up ? "_up2" : "", B, g0.H, g0.W, g0.Cin, g0.N, g0.residual ? 1 : 0);
  ProfScope ps(ctx, PK_CONV3, s, 2.0 * g0.M * (double)g0.N * 9.0 * g0.Cin,
               (double)g0.M * a.Cp + (double)g0.N * 9 * a.Cp + 2.0 * g0.M * g0.N * (g0.residual ? 2 : 1), tag);
  const dim3 grid((unsigned)((g0.M / 256) * tiles_n));
#ifdef FP8_STAMP
  unsigned long long* dbg = nullptr;
  HIP_OK(hipMalloc(&dbg, (size_t)grid.x * 64 * sizeof(unsigned long long)));
  HIP_OK(hipMemsetAsync(dbg, 0, (size_t)grid.x * 64 * sizeof(unsigned long long), s));
  a.g.slabs = (float*)dbg;
#endif
  if (bn == 160) hipLaunchKernelGGL((conv_halo_fp8_kernel<160>), grid, dim3(512), fp8_halo_smem<160>(), s, a);
  else hipLaunchKernelGGL((conv_halo_fp8_kernel<128>), grid, dim3(512), fp8_halo_smem<128>(), s, a);
  check_launch("conv_halo_fp8");
#ifdef FP8_STAMP
  {
    HIP_OK(hipStreamSynchronize(s));
    std::vector<unsigned long long> h((size_t)grid.x * 64);
    HIP_OK(hipMemcpy(h.data(), dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIP_OK(hipFree(dbg));
    double sum[2][8] = {{0}};
    for (size_t w = 0; w < (size_t)grid.x * 8; ++w)
      for (int k = 0; k < 8; ++k) sum[(w & 7) >> 2][k] += (double)h[w * 8 + k];
    for (int grp = 0; grp < 2; ++grp) {
      const double n = sum[grp][7] > 0 ? sum[grp][7] : 1, nw = (double)grid.x * 4;
      fprintf(stderr, "[fp8 stamps] %s grp %d: per step (cycles) reads+DMA issue %.0f | DMA wait %.0f | barrier in %.0f | MFMA %.0f | barrier out %.0f || per tile: loop+prologue %.0f epilogue %.0f steps %.0f\n",
              tag, grp, sum[grp][0] / n, sum[grp][1] / n, sum[grp][2] / n, sum[grp][3] / n, sum[grp][4] / n, sum[grp][5] / nw, sum[grp][6] / nw, n / nw);
    }
  }
#endif
}

}  // namespace SDNS
