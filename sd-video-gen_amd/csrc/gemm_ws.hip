// Weight-stationary dense NT GEMM for the short-K projections of the UNet's transformer blocks (K = 320 / 640, M in the tens of
// thousands):   C[M,N] = epilogue( A[M,K] * W[N,K]^T ).
//
// These launches move their operands once and do about one MFMA cycle per byte: HBM-bound.  What the tiled kernel (gemm.hip) gets
// out of them — 2.9 TB/s — is set by the CU's load path, not by the matrix pipe (profiles/r03_gemm_exp.json): every 128-row tile
// re-streams its W tiles from L2 (1.6 KB per row at N = K = 320 against 1.3 KB of A + residual), waits for six dependent DMA
// round trips and then for its residual behind the K loop.  A first rewrite that kept 16 rows per wave in registers and streamed W
// through an LDS ring (row-stationary, one 64-row tile per workgroup) was correct and SLOWER — 118 us with a store per stage
// (vmcnt retires in order: a slow store in front of a DMA stalls the ring), 94 us with the stores deferred, against 76 us:
// 3.2 KB of W per row through the same load path.  Hence:
//   * ONE persistent workgroup per CU (8 waves) keeps a column group of W — 100 KiB: 160 columns at K = 320, 80 at K = 640 —
//     in LDS for its whole life, laid out fragment-major (1-KiB piece = the MFMA A-operand fragment of one n-tile and k-step in
//     lane order: fragment reads are linear ds_read_b128, conflict free).  W is fetched once per CU, not once per row tile.
//   * a wave owns 16 rows of a 128-row tile: it loads them (whole K) straight into registers as MFMA B-operand fragments, with
//     the residual fragments of its column group, ONE TILE AHEAD (double-buffered register sets): the loads of tile i + 1 are
//     issued before the MFMAs of tile i and are older than tile i's stores in the in-order queue, so no wait ever depends on a
//     store.  No LDS staging of A, no barriers, no DMA in the loop: the compiler's own waitcnt insertion is exact here.
//   * column groups of the same rows run on workgroups that share an XCD (blockIdx.x % 8), so the re-read of A by the other
//     groups is served by that XCD's L2.
//   * a wave sees whole rows of its column group: LayerNorm row partials (GemmArgs::ln_part, one per row and column group) are
//     per-lane sums + two shuffles; GroupNorm column sums (gn_part, 128-row tiles) meet in LDS in wave order.
// v_mfma_f32_16x16x32 with the W fragment as the A operand (D[i = n][j = m]: 4 consecutive columns per lane, as gemm.hip).
//   * 16-byte epilogue traffic (round 5): the rows of W are dealt to the MFMA A-operand rows of an n-tile PAIR (2p, 2p + 1) so that a
//     lane's accumulator quads of the two tiles are 8 CONSECUTIVE output columns — tile 2p + h, operand row r <-> column
//     32 p + 8 (r >> 2) + 4 h + (r & 3).  W is fetched by per-lane DMA addresses, so this is an address map, not a repack; the residual
//     comes in and the result goes out as ONE 16-byte access per lane and tile pair instead of two 8-byte ones (22 instead of 32
//     vector-memory instructions per 16 x 160 wave-tile: the launch is bound by the CU's address path, SQ counters in DESIGN.md §4.1).
#include "igemm_epi.h"
#include <cstdlib>

namespace SDNS {

namespace {

constexpr int WS_ROWS = 128;
constexpr int WS_PIECES = 100;                 // (n-tile, k-step) pieces of a column group: 10 x 10 at K = 320, 5 x 20 at K = 640
constexpr int WS_W_BYTES = WS_PIECES * 1024;

// KS = K / 32; RES: residual present; QKV: the fused q | k | V^T projection (column groups from GemmArgs::vt_n0 on write V^T)
template <int KS, bool RES, bool QKV = false>
__global__ void __launch_bounds__(512, 2) gemm_ws_kernel(const GemmArgs g, int n_groups, int n_slices) {
  constexpr int NTG = WS_PIECES / KS;           // n-tiles (16 columns) per column group
  constexpr int GC = NTG * 16;                  // columns per group
  static_assert(NTG % 2 == 0, "n-tiles are processed in pairs (K = 320: 10 tiles of 16 columns)");
  constexpr int NTP = NTG / 2;                  // n-tile pairs
  // column (inside the group) of operand row r of n-tile j
  auto col_of = [](int j, int r) { return 32 * (j >> 1) + 8 * (r >> 2) + 4 * (j & 1) + (r & 3); };
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS: W group [100 KiB] | bias f32[GC] | ln_s f32[GC] | GroupNorm scratch 2 x [8 waves][GC][2] f32
  float* s_bias = (float*)(smem + WS_W_BYTES);
  float* s_lns = s_bias + GC;
  float* s_gn = s_lns + GC;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  // workgroup -> (column group, row slice): the groups of one slice sit on one XCD (workgroups are dealt round robin: bid % 8)
  const int bid = blockIdx.x;
  const int xcd = bid & 7, idx = bid >> 3;
  const int grp = idx % n_groups;
  const int slice = xcd + 8 * (idx / n_groups);
  if (slice >= n_slices) return;
  const int n0 = grp * GC;
  const bool is_vt = QKV && n0 >= g.vt_n0;                      // workgroup-uniform: a V^T column group of a fused q | k | V^T launch
  const int tiles = (g.M + WS_ROWS - 1) / WS_ROWS;
  const bool fold = g.ln_rs != nullptr, emit_ln = g.ln_part != nullptr, emit_gn = g.gn_part != nullptr;

  // ---- the column group of W, once: piece p = (n-tile j = p / KS, k-step ks = p % KS); lane (r = lane & 15, q = lane >> 4) fetches
  // W[n0 + 16 j + r][32 ks + 8 q .. +7] into bytes [16 lane, +16) of the piece
  {
    const unsigned w_bytes = (unsigned)(((int64_t)(g.n_valid - 1) * g.ldb + g.K) * 2);
    const uint64_t pw = (uint64_t)g.Wt;
    const v4i srdW = {(int)(unsigned)pw, (int)((pw >> 32) & 0xffff), (int)w_bytes, 0x00020000};
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    for (int p = wid; p < WS_PIECES; p += 8) {
      const int j = p / KS, ks = p - j * KS;
      const int n = n0 + col_of(j, l15);
      const unsigned voff = n < g.n_valid ? (unsigned)(n * g.ldb + ks * 32 + lq * 8) * 2u : 0x80000000u;
      dma16(srdW, voff, 0, lds0 + (unsigned)p * 1024u);
    }
    for (int c = tid; c < GC; c += 512) {
      const int n = n0 + c;
      s_bias[c] = (g.bias && n < g.N) ? g.bias[n] : 0.f;
      s_lns[c] = (fold && n < g.N) ? g.ln_s[n] : 0.f;
    }
  }

  // Branch-free operand access through buffer descriptors: a lane outside the problem (row >= M, tile past the end) carries an
  // out-of-range offset — loads return zeros, stores are dropped — so no load sits behind a branch (hipcc waits vmcnt(0) around
  // conditional loads, which would serialise the prefetch) and every wave issues the same instruction stream.
  typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (unsigned)((((int64_t)g.M - 1) * g.lda + g.K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t srdR = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? g.residual : g.A), 0,
                                                                        RES ? (unsigned)((((int64_t)g.M - 1) * g.ldr + g.N) * 2) : 0u, 0x00020000);
  const int n_c = QKV ? g.vt_n0 : g.N;                           // columns that go to C
  const __amdgpu_buffer_rsrc_t srdC = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, (unsigned)((((int64_t)g.M - 1) * g.ldc + n_c) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t srdS = __builtin_amdgcn_make_buffer_rsrc((void*)(fold ? g.ln_rs : (const float*)g.A), 0, fold ? (unsigned)g.M * 4u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdT = __builtin_amdgcn_make_buffer_rsrc((void*)(fold ? g.ln_rm : (const float*)g.A), 0, fold ? (unsigned)g.M * 4u : 0u, 0x00020000);
  auto load_tile = [&](int t, h16x8 (&af)[KS], uint4 (&rf)[NTP], float& lrs, float& lrm) {
    const int m = t * WS_ROWS + wid * 16 + l15;
    const bool ok = t < tiles && m < g.M;
    const unsigned a_off = ok ? (unsigned)(m * g.lda + lq * 8) * 2u : OOB;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(srdA, a_off, ks * 64, 0);      // (OOB + small soffset stays out of range)
      af[ks] = *(const h16x8*)&v;
    }
    if (RES) {
      const unsigned r_off = ok ? (unsigned)(m * g.ldr + n0 + lq * 8) * 2u : OOB;
#pragma unroll
      for (int p = 0; p < NTP; ++p) {
        const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(srdR, r_off, p * 64, 0);
        rf[p] = make_uint4(v.x, v.y, v.z, v.w);
      }
    }
    const unsigned s_off = ok ? (unsigned)m * 4u : OOB;
    lrs = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srdS, s_off, 0, 0));
    lrm = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srdT, s_off, 0, 0));
  };

  h16x8 af0[KS], af1[KS];
  uint4 rf0[NTP], rf1[NTP];
  float rs0, rm0, rs1, rm1;
  load_tile(slice, af0, rf0, rs0, rm0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // W landed (and the first tile's fragments)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  bar();

  // one 128-row tile: prefetch the next one into the other register set, multiply, finish, store
  auto step = [&](int t, h16x8 (&af)[KS], uint4 (&rf)[NTP], float lrs, float lrm, int tn, h16x8 (&afn)[KS], uint4 (&rfn)[NTP],
                  float& lrsn, float& lrmn, int parity) {
    load_tile(tn, afn, rfn, lrsn, lrmn);
    const int m = t * WS_ROWS + wid * 16 + l15;
    const bool m_ok = m < g.M;
    const char* wl = smem + lane * 16;
    float ln1 = 0.f, ln2 = 0.f;
    float* sc = s_gn + parity * (8 * GC * 2);
    if (QKV && is_vt) {
      // V^T group: the same fragments with the MFMA operand roles swapped — D[i = token][j = column]: a lane holds 4 consecutive
      // tokens of ONE V column, which is 8 contiguous bytes of a V^T row
      const int mq = t * WS_ROWS + wid * 16 + lq * 4;                // first of this lane's 4 tokens
      const int smp = mq / g.vt_rows, tok = mq - smp * g.vt_rows;
      // LayerNorm fold needs the statistics of the lane's 4 tokens (not of row l15): fetched from the wave's registers by shuffle
      float rs4[4], rm4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { rs4[e] = __shfl(lrs, lq * 4 + e); rm4[e] = __shfl(lrm, lq * 4 + e); }
#pragma unroll
      for (int j = 0; j < NTG; ++j) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const h16x8 wf = *(const h16x8*)(wl + (j * KS + ks) * 1024);
          acc = MFMA_16x16x32(af[ks], wf, acc);
        }
        const int c = col_of(j, l15);                                // this lane's column inside the group
        const float bv = s_bias[c], sv = s_lns[c];
        h16x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[e] * g.alpha;
          if (fold) v = v * rs4[e] - sv * rm4[e];
          w[e] = (h16)(v + bv);
        }
        if (mq < g.M) *(h16x4*)(g.vt_out + (int64_t)smp * g.vt_bs + (int64_t)(n0 - g.vt_n0 + c) * g.vt_ld + tok) = w;
        __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
#pragma unroll
    for (int p = 0; p < NTP; ++p) {
      h16x8 w8;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int j = 2 * p + hh;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const h16x8 wf = *(const h16x8*)(wl + (j * KS + ks) * 1024);
          acc = MFMA_16x16x32(wf, af[ks], acc);
        }
        const int c = 32 * p + 8 * lq + 4 * hh;             // column inside the group: the lane's quads of the pair are 8 consecutive columns
        f32x4 v = acc * g.alpha;
        if (fold) v = v * lrs - *(const f32x4*)(s_lns + c) * lrm;
        v += *(const f32x4*)(s_bias + c);
        if (RES) {
          const uint2 rr = hh ? make_uint2(rf[p].z, rf[p].w) : make_uint2(rf[p].x, rf[p].y);
          const h16x4 r = *(const h16x4*)&rr;
          v[0] += (float)r[0]; v[1] += (float)r[1]; v[2] += (float)r[2]; v[3] += (float)r[3];
        }
        if (g.act == ACT_SILU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
        } else if (g.act == ACT_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        }
        const h16x4 w = to_h16x4(v);
#pragma unroll
        for (int e = 0; e < 4; ++e) w8[4 * hh + e] = w[e];
        __builtin_amdgcn_sched_barrier(0);                  // one n-tile's fragment reads at a time (the scheduler otherwise hoists tens of them: spills)
        if (emit_ln || emit_gn) {
          float x[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) { x[e] = m_ok ? (float)w[e] : 0.f; ln1 += x[e]; ln2 += x[e] * x[e]; }
          if (emit_gn) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float a = row16_sum(x[e]), b = row16_sum(x[e] * x[e]);
              if (l15 == 0) *(float2*)(sc + (wid * GC + c + e) * 2) = make_float2(a, b);
            }
          }
        }
      }
      rf[p] = *(const uint4*)&w8;                         // kept until the stores below (after the last MFMA of the tile)
    }
    {
      const unsigned c_off = m_ok ? (unsigned)(m * g.ldc + n0 + lq * 8) * 2u : OOB;
#pragma unroll
      for (int p = 0; p < NTP; ++p) {
        u32x4v v; v.x = rf[p].x; v.y = rf[p].y; v.z = rf[p].z; v.w = rf[p].w;
        __builtin_amdgcn_raw_buffer_store_b128(v, srdC, c_off, p * 64, 0);
      }
    }
    if (emit_ln) {
      ln1 += __shfl_xor(ln1, 16); ln2 += __shfl_xor(ln2, 16);
      ln1 += __shfl_xor(ln1, 32); ln2 += __shfl_xor(ln2, 32);
      if (lq == 0 && m_ok) *(float2*)(g.ln_part + ((int64_t)m * n_groups + grp) * 2) = make_float2(ln1, ln2);
    }
    if (emit_gn) {
      // the eight waves' column sums of this tile, added in wave order (scratch double-buffered by tile parity: one barrier per tile)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      bar();
      for (int c = tid; c < GC; c += 512) {
        if (n0 + c >= g.N) continue;
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) { a += sc[(w * GC + c) * 2]; b += sc[(w * GC + c) * 2 + 1]; }
        *(float2*)(g.gn_part + ((int64_t)t * g.N + n0 + c) * 2) = make_float2(a, b);
      }
    }
  };

  for (int t = slice; t < tiles; t += 2 * n_slices) {
    step(t, af0, rf0, rs0, rm0, t + n_slices, af1, rf1, rs1, rm1, 0);
    if (t + n_slices < tiles) step(t + n_slices, af1, rf1, rs1, rm1, t + 2 * n_slices, af0, rf0, rs0, rm0, 1);
  }
}

int ws_smem(int gc) { return WS_W_BYTES + 2 * gc * 4 + 2 * 8 * gc * 2 * 4; }

int g_ws_cus = 0;       // compute units of the device (persistent grid = one workgroup per CU)

}  // namespace

int gemm_ws_groups(const GemmArgs& g) { return g.N / (WS_PIECES / (g.K / 32) * 16); }

// problems the weight-stationary kernel takes: dense, 16-bit output, K = 320 / 640, whole column groups, unbatched (one W for all
// rows), no split-K / GEGLU / row bias / per-sample bias / swapped LayerNorm / two-source A
bool gemm_ws_supported(const GemmArgs& g) {
  static const int on = getenv("SVG_GEMM_WS") ? atoi(getenv("SVG_GEMM_WS")) : 1;
  if (!on) return false;
  if (g.amode != A_DENSE || g.A2 || g.out_f32 || g.act == ACT_GEGLU || g.bias_row || g.bias_bn || g.ln_swapped || g.batch != 1) return false;
  // K = 640 (80-column groups) builds and is correct but loses: two register sets of 80 A-fragment registers spill, and eight /
  // sixteen column groups re-read A that often (69 vs 53 us at 28672 x 640 x 640): the tiled kernel keeps those shapes
  if (g.K != 320) return false;
  const int gc = WS_PIECES / (g.K / 32) * 16;
  if (g.N % gc != 0 || g.N > 1280 || (g.n_valid > 0 && g.n_valid < g.N)) return false;
  if (g.residual && (g.ldr & 7)) return false;            // 16-byte residual loads / stores (8 consecutive columns per lane)
  if ((g.lda & 7) || (g.ldc & 7) || (g.ldb & 7)) return false;
  if (g.vt_out && (g.vt_n0 % gc != 0 || g.vt_rows % 16 != 0 || (g.vt_ld & 3) || g.residual || g.gn_part || g.ln_part || g.act != ACT_NONE)) return false;
  if (g.M < 16384) return false;                        // the tiled kernel's territory: too few 128-row tiles per CU to amortise the W load
  return true;
}

void gemm_ws_init_device() {
  const int smem = ws_smem(160);
  HIP_OK(hipFuncSetAttribute((const void*)gemm_ws_kernel<10, false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  HIP_OK(hipFuncSetAttribute((const void*)gemm_ws_kernel<10, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  HIP_OK(hipFuncSetAttribute((const void*)gemm_ws_kernel<10, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  int dev = 0;
  hipDeviceProp_t prop;
  HIP_OK(hipGetDevice(&dev));
  HIP_OK(hipGetDeviceProperties(&prop, dev));
  g_ws_cus = prop.multiProcessorCount;
}

void launch_gemm_ws(svg_ctx* ctx, const GemmArgs& g, hipStream_t s) {
  const int ng = gemm_ws_groups(g);
  const int gc = g.N / ng;
  const int cus = g_ws_cus > 0 ? g_ws_cus : 256;
  const int tiles = cdiv(g.M, WS_ROWS);
  // row slices: a multiple of 8 (one per XCD position), as many as the CUs allow, not more than the tiles
  int n_slices = std::max(8, (cus / ng) / 8 * 8);
  n_slices = std::min(n_slices, (int)align_up(tiles, 8));
  dim3 grid(n_slices * ng);
  const int smem = ws_smem(gc);
  const bool res = g.residual != nullptr;
  if (g.vt_out) {
    hipLaunchKernelGGL((gemm_ws_kernel<10, false, true>), grid, dim3(512), smem, s, g, ng, n_slices);
  } else {                                               // K = 320 (gemm_ws_supported: the K = 640 form lost to the tiled kernel and is gone)
    SVG_CHECK(g.K == 320, "gemm_ws: K = %d", g.K);
    if (res) hipLaunchKernelGGL((gemm_ws_kernel<10, true>), grid, dim3(512), smem, s, g, ng, n_slices);
    else hipLaunchKernelGGL((gemm_ws_kernel<10, false>), grid, dim3(512), smem, s, g, ng, n_slices);
  }
}

}  // namespace SDNS
