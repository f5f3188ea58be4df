// MFMA implicit GEMM for gfx950 (CDNA4): dense NT GEMM and 3x3 convolution on NHWC bf16.
//
//   C[M,N] = epilogue( A[M,K] * Wt[N,K]^T ),  f32 accumulate on v_mfma_f32_16x16x32_bf16.
//
// Tile 128 x BN x 64 per 256-thread workgroup (4 waves as 2x2, each 64 x BN/2), two LDS stages filled by LDS-direct
// buffer loads (no VGPR round trip): the DMA of K-step t+1 is issued before the MFMAs of K-step t.
// (A 256 x 320 tile — 4 waves x 512 registers, half the L2 traffic per FLOP — was tried: hipcc cannot allocate its
//  320 accumulator registers without hundreds of AGPR<->VGPR copies and scratch spills in the loop; it needs hand asm.)
// LDS rows are 128 B (64 h16); 16-B chunk c of row r is stored at chunk c ^ ((r>>1)&7) so the
// ds_read_b128 fragment reads (16 rows x 2 k-chunks per lane group) are bank-conflict free.
// The MFMA is issued with the WEIGHT fragment as the A operand and the activation fragment as
// the B operand, so D[i][j] has i = output column n (4 consecutive n per lane in registers) and
// j = output row m: the epilogue stores 4 contiguous outputs per lane (8 B bf16 / 16 B f32).
// 3x3 conv = same GEMM with the A tile gathered from NHWC: K index = (tap, cin), cin contiguous.
#include <cstdlib>

#include "igemm_epi.h"

namespace SDNS {

// conv_halo.hip
bool conv_halo_supported(const GemmArgs& g);
int conv_halo_bn(const GemmArgs& g);
void launch_conv_halo(const GemmArgs& g, dim3 grid, hipStream_t s);
bool gemm_pp_supported(const GemmArgs& g);
bool gemm_ws_supported(const GemmArgs& g);
int gemm_ws_groups(const GemmArgs& g);
void launch_gemm_ws(svg_ctx* ctx, const GemmArgs& g, hipStream_t s);
int gemm_pp_bn(const GemmArgs& g);
void launch_gemm_pp(const GemmArgs& g, hipStream_t s);

namespace {

// GEGLU: h and gate are 4 consecutive packed columns nh.. / nh+16..; output column oc..oc+3
__device__ __forceinline__ void epi_store_geglu(const GemmArgs& g, int z, int m, int nh, int oc, f32x4 h, f32x4 gt) {
  int64_t o = (int64_t)z * g.sC + (int64_t)m * g.ldc + oc;
  *(h16x4*)((h16*)g.C + o) = to_h16x4(geglu_value(g, m, nh, h, gt));
}

// WIDE: the wide tile epilogue (igemm_epi.h) — dense GEMMs with a 16-bit output and no GEGLU (the launcher decides)
// WIDE = 2: the same 16-byte epilogue through a permutation of the W rows at load time instead of lane exchanges (igemm_epi.h: epi_perm_col) —
// no extra registers: the form of the 160-column dense tiles (launched without split-K only: the slab path keeps natural columns)
template <int BN, int AMODE, int WIDE = 0>
__global__ void __launch_bounds__(256, 2) igemm_kernel(const GemmArgs g) {
  constexpr int NT = BN / 32;       // 16-wide n tiles per wave
  constexpr int MT = 4;             // 16-high m tiles per wave
  constexpr int BIT = BN / 32;      // B-tile rows per thread
  constexpr int A_BYTES = BM * 128;
  constexpr int B_BYTES = BN * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int z = blockIdx.y, ks = blockIdx.z;

  // XCD-aware tile order: blocks that share an XCD (bid % 8) get a contiguous run of tiles,
  // consecutive tiles share the A row panel (same tm) so its re-reads hit that XCD's L2.
  const int tiles_n = (g.N + BN - 1) / BN;
  int tile;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int tm, tn;
  if (g.tn_major) {   // weight-heavy problem: the XCD's run of tiles walks the rows for a fixed column tile
    const int tiles_m = (g.M + BM - 1) / BM;
    tn = tile / tiles_m; tm = tile - tn * tiles_m;
  } else {
    tm = tile / tiles_n; tn = tile - tm * tiles_n;
  }
  const int m0 = tm * BM, n0 = tn * BN;

  const int KT = (g.K + BK - 1) / BK;
  const int kt_per = (KT + g.splitk - 1) / g.splitk;
  const int kt_begin = ks * kt_per;
  const int kt_end = min(KT, kt_begin + kt_per);

  const h16* __restrict__ Ap = g.A + (int64_t)z * g.sA;
  const h16* __restrict__ Bp = g.Wt + (int64_t)z * g.sB;

  // ---- loader state ---------------------------------------------------------------------------
  // Global -> register staging uses buffer loads: per-lane byte offset in voffset, the K-slab offset (uniform)
  // in the scalar soffset, and the hardware range check for zero fill — an invalid chunk (padding tap, row or
  // column out of range, K tail) gets voffset 0x80000000, which is past num_records and reads as zeros.
  constexpr unsigned INVALID = 0x80000000u;
  // LDS-direct staging (buffer_load ... lds): a wave instruction writes 64 x 16 B = 8 LDS rows contiguously in
  // lane order, so the XOR swizzle is applied to the SOURCE chunk: the lane landing at position tid&7 of row r
  // fetches logical chunk (tid&7) ^ ((r>>1)&7); fragment reads apply the same XOR (lds_off).
  const int r0 = tid >> 3;       // 0..31: row within a 32-row group (rows r0 + 32 i)
  const int c = (tid & 7) ^ ((r0 >> 1) & 7);   // logical 16-B chunk of the 64-wide K slab this lane fetches
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned a_bytes, b_bytes;
  if (AMODE == A_DENSE) a_bytes = (unsigned)(((int64_t)(g.M - 1) * g.lda + (g.A2 ? g.k_split : g.K)) * 2);
  else a_bytes = (unsigned)((int64_t)(g.M / (g.Ho * g.Wo)) * g.H * g.W * g.Cin * 2);
  b_bytes = (unsigned)(((int64_t)(g.n_valid - 1) * g.ldb + g.K) * 2);
  const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc((void*)Ap, 0, a_bytes, 0x00020000);
  // second source of a two-source dense A (channel concat [A | A2]): K slabs from k_split on
  const bool two_src = AMODE == A_DENSE && g.A2 != nullptr;
  const unsigned a2_bytes = two_src ? (unsigned)(((int64_t)(g.M - 1) * g.lda2 + (g.K - g.k_split)) * 2) : 0u;
  const __amdgpu_buffer_rsrc_t srdA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(two_src ? g.A2 : Ap), 0, two_src ? a2_bytes : a_bytes, 0x00020000);
  const int kt_split = two_src ? g.k_split / BK : 0x7fffffff;
  const __amdgpu_buffer_rsrc_t srdB = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, b_bytes, 0x00020000);

  // A rows
  int a_base[4];                 // conv: element offset of (b,0,0,0)
  int a_yx[4];                   // conv: (y<<16)|x of the output pixel, -1 = row out of range
  unsigned a_voff[4];            // byte offset of this lane's chunk (dense: row; conv: current tap's pixel) or INVALID
  unsigned a2_voff[4];           // the same in the second source of a two-source dense A
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + r0 + 32 * i;
    if (AMODE == A_DENSE) {
      a_voff[i] = (m < g.M) ? (unsigned)(m * g.lda + c * 8) * 2u : INVALID;
      a2_voff[i] = (two_src && m < g.M) ? (unsigned)(m * g.lda2 + c * 8) * 2u : INVALID;
      a_base[i] = 0; a_yx[i] = 0;
    } else {
      if (m < g.M) {
        int hw = g.Ho * g.Wo;
        int b = m / hw;
        int rem = m - b * hw;
        int y = rem / g.Wo;
        int x = rem - y * g.Wo;
        a_yx[i] = (y << 16) | x;
        a_base[i] = b * g.H * g.W * g.Cin;
      } else {
        a_yx[i] = -1; a_base[i] = 0;
      }
      a_voff[i] = INVALID;
    }
  }
  // B rows
  unsigned b_voff[5];            // fixed extent (BIT <= 5): a template-dependent extent here makes hipcc's host pass drop the kernel stub
#pragma unroll
  for (int i = 0; i < BIT; ++i) {
    int n = n0 + r0 + 32 * i;
    if (WIDE == 2) {                                     // LDS row rb = r0 + 32 i of the B tile holds the weight row its MFMA operand row stands for
      const int rb = r0 + 32 * i, run = rb / (BN / 2);
      n = n0 + run * (BN / 2) + epi_perm_col<NT>(rb - run * (BN / 2));
    }
    b_voff[i] = (n < g.n_valid) ? (unsigned)(n * g.ldb + c * 8) * 2u : INVALID;
  }

  // element offset of the pixel under `tap` for row i, or -1 (padding / row out of range)
  auto tap_offset = [&](int i, int tap) -> int {
    if (a_yx[i] < 0 || tap >= 9) return -1;
    int y = a_yx[i] >> 16, x = a_yx[i] & 0xffff;
    int ky = tap / 3, kx = tap - ky * 3;
    int yy, xx;
    if (AMODE == A_CONV_S1 || AMODE == A_CONV_SMALLC) {
      yy = y + ky - 1; xx = x + kx - 1;
    } else if (AMODE == A_CONV_S2P1) {
      yy = 2 * y + ky - 1; xx = 2 * x + kx - 1;
    } else if (AMODE == A_CONV_S2ASYM) {
      yy = 2 * y + ky; xx = 2 * x + kx;
    } else {  // A_CONV_UP2: conv over the nearest-2x upsampled image, read the source pixel
      int uy = y + ky - 1, ux = x + kx - 1;
      if ((unsigned)uy >= (unsigned)(2 * g.H) || (unsigned)ux >= (unsigned)(2 * g.W)) return -1;
      yy = uy >> 1; xx = ux >> 1;
    }
    if ((unsigned)yy >= (unsigned)g.H || (unsigned)xx >= (unsigned)g.W) return -1;
    return a_base[i] + (yy * g.W + xx) * g.Cin;
  };
  auto tap_voff = [&](int i, int tap) -> unsigned {
    const int off = tap_offset(i, tap);
    return off >= 0 ? (unsigned)(off + c * 8) * 2u : INVALID;
  };

  // incremental (tap, cin0) of the K slab being loaded (conv modes with Cin % 64 == 0)
  int ld_tap = 0, ld_cin0 = 0;
  if (AMODE != A_DENSE && AMODE != A_CONV_SMALLC) {
    int k0 = kt_begin * BK;
    ld_tap = k0 / g.Cin;
    ld_cin0 = k0 - ld_tap * g.Cin;
#pragma unroll
    for (int i = 0; i < 4; ++i) a_voff[i] = tap_voff(i, ld_tap);
  }

  const bool k_tail = (g.K & (BK - 1)) != 0;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  // issue the DMA of K slab `kt` into LDS stage `buf`: 4 + BIT wave instructions of 1 KiB each, no VGPR staging
  auto dma_tiles = [&](int kt, int buf) {
    if (g.dbg == 3) return;
    char* sa = smem + buf * (A_BYTES + B_BYTES) + wave_u * 1024;
    char* sb = sa + A_BYTES;
    const int ksoff = kt * (BK * 2);                        // bytes, wave-uniform
    const bool k_ok = !k_tail || (kt * BK + c * 8 < g.K);   // only the last slab of a ragged K can fail
    if (AMODE == A_DENSE) {
      if (kt >= kt_split) {      // wave-uniform: this slab lies in the second source
        const int ksoff2 = (kt - kt_split) * (BK * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA2, (lds_ptr_t)(sa + i * 4096), 16, k_ok ? a2_voff[i] : INVALID, ksoff2, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (lds_ptr_t)(sa + i * 4096), 16, k_ok ? a_voff[i] : INVALID, ksoff, 0, 0);
      }
    } else if (AMODE == A_CONV_SMALLC) {
      const int tap = kt * 8 + c;                           // Cin == 8: one 16-B chunk per tap
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int off = tap_offset(i, tap);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (lds_ptr_t)(sa + i * 4096), 16, off >= 0 ? (unsigned)off * 2u : INVALID, 0, 0, 0);
      }
    } else {
      const int csoff = ld_cin0 * 2;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (lds_ptr_t)(sa + i * 4096), 16, a_voff[i], csoff, 0, 0);
      ld_cin0 += BK;
      if (ld_cin0 >= g.Cin) {
        ld_cin0 = 0;
        ++ld_tap;
#pragma unroll
        for (int i = 0; i < 4; ++i) a_voff[i] = tap_voff(i, ld_tap);
      }
    }
#pragma unroll
    for (int i = 0; i < BIT; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdB, (lds_ptr_t)(sb + i * 4096), 16, k_ok ? b_voff[i] : INVALID, ksoff, 0, 0);
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, lq = lane >> 4;

  auto compute = [&](int buf) {
    if (g.dbg == 2) return;
    const char* sa = smem + buf * (A_BYTES + B_BYTES);
    const char* sb = sa + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      h16x8 af[MT], bfr[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = *(const h16x8*)(sa + lds_off(wm * 64 + i * 16 + l15, kk * 4 + lq));
#pragma unroll
      for (int j = 0; j < NT; ++j) bfr[j] = *(const h16x8*)(sb + lds_off(wn * (BN / 2) + j * 16 + l15, kk * 4 + lq));
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = MFMA_16x16x32(bfr[j], af[i], acc[i][j]);
    }
  };

  // Two LDS stages: the DMA of slab t+1 is issued into the other stage before the MFMAs of slab t; it may start
  // only after the barrier that ended iteration t-1 (every wave has finished reading that stage), and slab t+1 is
  // read only after this wave's vmcnt(0) AND the barrier (every wave's DMA has landed).
  if (kt_begin < kt_end) {
    dma_tiles(kt_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      const int buf = (kt - kt_begin) & 1;
      if (kt + 1 < kt_end) dma_tiles(kt + 1, buf ^ 1);
      compute(buf);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // ---- epilogue --------------------------------------------------------------------------------
  if (g.dbg == 1 && acc[0][0][0] != 12345.f) return;
  const bool geglu = g.act == ACT_GEGLU;
  const int n_out = geglu ? (g.N >> 1) : g.N;          // columns of C
  if (g.splitk > 1) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + wm * 64 + i * 16 + l15;
      if (m >= g.M) continue;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 16 + lq * 4;
        if (n >= g.N) continue;
        float* sl = g.slabs + ((int64_t)(ks * g.batch + z) * g.M + m) * g.N + n;
        *(f32x4*)sl = acc[i][j];
      }
    }
    return;
  }
  // (an LDS-staged, 16-byte coalesced store variant measured no faster: L2 merges the 8-byte pieces)
  epi_tile<MT, NT, AMODE == A_DENSE, WIDE>(g, z, m0 + wm * 64 + l15, 16, n0 + wn * (BN / 2) + lq * 4, acc, smem, 2, wm, wn, tm, n0);
}

// sums the split-K slabs and applies the epilogue
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const GemmArgs g) {
  const int z = blockIdx.y;
  const int n4 = g.N >> 2;
  const int64_t total = (int64_t)g.M * n4;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(idx / n4);
    const int n = (int)(idx - (int64_t)m * n4) * 4;
    if (g.act == ACT_GEGLU) {
      // packed columns: 16-wide tiles alternate h / gate.  Visit only h tiles.
      if ((n >> 4) & 1) continue;
      f32x4 h = {0, 0, 0, 0}, gt = {0, 0, 0, 0};
      for (int s = 0; s < g.splitk; ++s) {
        const float* sl = g.slabs + ((int64_t)(s * g.batch + z) * g.M + m) * g.N + n;
        h += *(const f32x4*)sl;
        gt += *(const f32x4*)(sl + 16);
      }
      const int oc = (n >> 5) * 16 + (n & 15);
      h = h * g.alpha; gt = gt * g.alpha;
      if (g.ln_rs) { h = ln_fold(g, z, m, n, h); gt = ln_fold(g, z, m, n + 16, gt); }
      epi_store_geglu(g, z, m, n, oc, h, gt);
    } else {
      f32x4 v = {0, 0, 0, 0};
      for (int s = 0; s < g.splitk; ++s)
        v += *(const f32x4*)(g.slabs + ((int64_t)(s * g.batch + z) * g.M + m) * g.N + n);
      v = v * g.alpha;
      if (g.ln_rs) v = ln_fold(g, z, m, n, v);
      epi_store(g, z, m, n, v);
    }
  }
}

// ---- packing -------------------------------------------------------------------------------------
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, h16* __restrict__ out, int O, int I, int Opad, int Ipad) {
  int64_t total = (int64_t)Opad * 9 * Ipad;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int ci = (int)(idx % Ipad);
    int t = (int)((idx / Ipad) % 9);
    int o = (int)(idx / ((int64_t)Ipad * 9));
    float v = (o < O && ci < I) ? w[((int64_t)o * I + ci) * 9 + t] : 0.f;
    out[idx] = (h16)v;
  }
}
__global__ void pack_linear_kernel(const float* __restrict__ w, h16* __restrict__ out, int N, int K, int Npad) {
  int64_t total = (int64_t)Npad * K;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int n = (int)(idx / K);
    out[idx] = (h16)((n < N) ? w[idx] : 0.f);
  }
}
__global__ void pack_geglu_kernel(const float* __restrict__ w, const float* __restrict__ b, h16* __restrict__ wout,
                                  float* __restrict__ bout, int F, int K) {
  // packed row p: tile = p/16; pair = tile/2; which = tile&1 (0 h, 1 gate); src row = which*F + pair*16 + p%16
  int64_t total = (int64_t)2 * F * K;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int p = (int)(idx / K), k = (int)(idx - (int64_t)p * K);
    int tile = p >> 4, pair = tile >> 1, which = tile & 1;
    int src = which * F + pair * 16 + (p & 15);
    wout[idx] = (h16)w[(int64_t)src * K + k];
    if (k == 0 && b) bout[p] = b[src];
  }
}

// instantiations that take the wide (16-byte) tile epilogue: the dense 64 / 128-column tiles, and the 8-channel-input conv at 128 columns
// (the VAE's conv_in at 512 x 512: 1.9 GB of output per 28-clip launch — a store-bound launch)
template <int BN, int AMODE>
constexpr bool kWideInst = (AMODE == A_DENSE && (BN == 64 || BN == 128)) || (AMODE == A_CONV_SMALLC && BN == 128);
template <int BN, int AMODE>
constexpr bool kPermInst = AMODE == A_DENSE && BN == 160;      // WIDE = 2 (W rows permuted at load): dense 160-column tiles

template <int BN, int AMODE>
void launch_inst(const GemmArgs& g, dim3 grid, hipStream_t s) {
  constexpr int smem = 2 * (BM * 128 + BN * 128);
  if constexpr (kWideInst<BN, AMODE>) {      // (BN = 160: 48 spilled registers in the exchange form — the permuted-row form below)
    if (g.act != ACT_GEGLU && !g.out_f32) { hipLaunchKernelGGL((igemm_kernel<BN, AMODE, 1>), grid, dim3(256), smem, s, g); return; }
  }
  if constexpr (kPermInst<BN, AMODE>) {
    static const int perm_on = getenv("SVG_IGEMM_PERM") ? atoi(getenv("SVG_IGEMM_PERM")) : 1;
    if (perm_on && g.act != ACT_GEGLU && !g.out_f32 && g.splitk == 1) { hipLaunchKernelGGL((igemm_kernel<BN, AMODE, 2>), grid, dim3(256), smem, s, g); return; }
  }
  hipLaunchKernelGGL((igemm_kernel<BN, AMODE>), grid, dim3(256), smem, s, g);
}

template <int BN>
void launch_bn(const GemmArgs& g, dim3 grid, hipStream_t s) {
  switch (g.amode) {
    case A_DENSE: launch_inst<BN, A_DENSE>(g, grid, s); break;
    case A_CONV_S1: launch_inst<BN, A_CONV_S1>(g, grid, s); break;
    case A_CONV_S2P1: launch_inst<BN, A_CONV_S2P1>(g, grid, s); break;
    case A_CONV_S2ASYM: launch_inst<BN, A_CONV_S2ASYM>(g, grid, s); break;
    case A_CONV_UP2: launch_inst<BN, A_CONV_UP2>(g, grid, s); break;
    case A_CONV_SMALLC: launch_inst<BN, A_CONV_SMALLC>(g, grid, s); break;
    default: throw SvgError("bad amode");
  }
}

// dynamic-LDS limits are a per-device function attribute: set for every instantiation when a context is created on a
// device (svg_create), not lazily behind a process-wide flag (a second device, or two host threads launching at once)
template <int BN, int AMODE>
void attr_inst() {
  HIP_OK(hipFuncSetAttribute((const void*)igemm_kernel<BN, AMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (BM * 128 + BN * 128)));
  if constexpr (kWideInst<BN, AMODE>)
    HIP_OK(hipFuncSetAttribute((const void*)igemm_kernel<BN, AMODE, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (BM * 128 + BN * 128)));
  if constexpr (kPermInst<BN, AMODE>)
    HIP_OK(hipFuncSetAttribute((const void*)igemm_kernel<BN, AMODE, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (BM * 128 + BN * 128)));
}
template <int BN>
void attr_bn() {
  attr_inst<BN, A_DENSE>(); attr_inst<BN, A_CONV_S1>(); attr_inst<BN, A_CONV_S2P1>(); attr_inst<BN, A_CONV_S2ASYM>();
  attr_inst<BN, A_CONV_UP2>(); attr_inst<BN, A_CONV_SMALLC>();
}

int pick_bn(const GemmArgs& g) {
  static const int force = getenv("SVG_GEMM_BN") ? atoi(getenv("SVG_GEMM_BN")) : 0;
  if (force && g.act != ACT_GEGLU && g.N > 64) return force;
  if (g.act == ACT_GEGLU) return 128;
  if (g.N <= 32) return 32;
  if (g.N <= 64) return 64;
  // per-CU serial work ~ ceil(blocks / 256) * BN (blocks beyond one per CU share the matrix pipe); ties go to the
  // width with fewer padded columns, then to the wider tile (the A panel is re-read once per column tile)
  const int64_t tm = (int64_t)cdiv(g.M, BM) * g.batch;
  // the 8 x 8 level's convolutions (1792 rows, K = 11520 / 23040): 14 x 8 tiles of 160 columns x split-K 4 = 448 workgroups, two per CU in
  // one wave of the grid: 0.067 / 0.115 ms against 0.073 / 0.121 for 128 columns x split-K 3 (profiles/r04_kbench_8x8_sweep.txt)
  if (g.amode != A_DENSE && g.N % 160 == 0 && tm * (g.N / 160) < 192) return 160;
  int best = 128;
  int64_t best_cost = -1, best_pad = 0;
  for (int bn : {128, 160}) {
    const int64_t tn = cdiv(g.N, bn);
    const int64_t cost = ((tm * tn + 255) / 256) * bn;
    const int64_t pad = tn * bn - g.N;
    if (best_cost < 0 || cost < best_cost || (cost == best_cost && (pad < best_pad || (pad == best_pad && bn > best)))) {
      best = bn; best_cost = cost; best_pad = pad;
    }
  }
  return best;
}

}  // namespace

static int plan_splitk(const GemmArgs& g);

int gemm_emits_gn(const GemmArgs& g0) {
  GemmArgs g = g0;
  if (g.n_valid <= 0) g.n_valid = g.N;
  if (g.out_f32 || g.act == ACT_GEGLU || g.N > g.ldc) return 0;
  if (g.batch != 1) return 0;
  if (plan_splitk(g) > 1) return 0;
  g.splitk = 1;
  if (conv_halo_supported(g)) return 256;
  if (gemm_ws_supported(g)) return 128;
  if (gemm_pp_supported(g)) return 256;
  return pick_bn(g) >= 32 ? BM : 0;
}

int gemm_ln_tiles(const GemmArgs& g0) {
  GemmArgs g = g0;
  if (g.n_valid <= 0) g.n_valid = g.N;
  if (g.out_f32 || g.act == ACT_GEGLU || g.N > g.ldc || g.amode != A_DENSE || g.bias_row) return 0;
  if (plan_splitk(g) > 1) return 0;
  g.splitk = 1;
  if (gemm_ws_supported(g)) return gemm_ws_groups(g);      // one partial per row and column group
  if (gemm_pp_supported(g)) return cdiv(g.N, gemm_pp_bn(g));
  const int bn = pick_bn(g);
  return bn >= 128 ? cdiv(g.N, bn) : 0;
}

void gemm_init_device() { attr_bn<32>(); attr_bn<64>(); attr_bn<128>(); attr_bn<160>(); }

void launch_gemm(svg_ctx* ctx, const GemmArgs& g, hipStream_t s, int prof_kind) {
  SVG_CHECK(g.N % 4 == 0 && g.K % 8 == 0, "gemm: N (%d) must be a multiple of 4 and K (%d) of 8", g.N, g.K);
  SVG_CHECK(g.M > 0 && g.N > 0 && g.K > 0, "gemm: empty problem %d %d %d", g.M, g.N, g.K);
  if (g.amode != A_DENSE) {
    SVG_CHECK(g.amode == A_CONV_SMALLC ? (g.Cin == 8 && g.K == 72) : (g.Cin % 64 == 0 && g.K == 9 * g.Cin),
              "conv: unsupported Cin %d / K %d for mode %d", g.Cin, g.K, g.amode);
    SVG_CHECK((int64_t)g.M / (g.Ho * g.Wo) * g.H * g.W * g.Cin < (1LL << 31), "conv: input too large for 32-bit offsets");
    SVG_CHECK(g.Ho < 32768 && g.Wo < 32768, "conv: spatial dims too large");
    SVG_CHECK(g.batch == 1, "conv: batch must be folded into M");
  } else {
    SVG_CHECK((int64_t)g.M * g.lda < (1LL << 31) && g.lda % 8 == 0, "gemm: lda %d / M %d unsupported", g.lda, g.M);
    if (g.A2) SVG_CHECK(g.k_split > 0 && g.k_split % BK == 0 && g.k_split < g.K && g.lda2 % 8 == 0 && (int64_t)g.M * g.lda2 < (1LL << 31) && g.batch == 1,
                        "gemm: two-source A needs k_split %d to be a multiple of %d inside K %d", g.k_split, BK, g.K);
  }
  SVG_CHECK((int64_t)g.N * g.ldb < (1LL << 31) && g.ldb % 8 == 0, "gemm: ldb %d unsupported", g.ldb);
  if (g.act == ACT_GEGLU) SVG_CHECK(g.N % 128 == 0, "geglu: packed N must be a multiple of 128");
  if (!SVG_LAUNCHING(ctx)) return;
  GemmArgs a = g;
  static const int dbg_env = getenv("SVG_GEMM_DBG") ? atoi(getenv("SVG_GEMM_DBG")) : 0;
  a.dbg = dbg_env;
  if (a.splitk < 1) a.splitk = 1;
  if (a.n_valid <= 0) a.n_valid = a.N;
  {
    // Each XCD has its own L2: with the A rows adjacent every XCD streams ALL the weights (8 x N*K*2 bytes per launch),
    // with the weights adjacent every XCD streams all of A.  Pick the cheaper (16 x 16 / 8 x 8 convs: 30 MB of weights
    // against 5-18 MB of image).
    static const int tn_env = getenv("SVG_TN_MAJOR") ? atoi(getenv("SVG_TN_MAJOR")) : -1;
    const double a_bytes_phys = (a.amode == A_DENSE ? (double)a.M * a.K : (double)(a.M / (a.Ho * a.Wo)) * a.H * a.W * a.Cin) * 2.0;
    const double w_bytes = (double)a.N * a.K * 2.0;
    a.tn_major = tn_env >= 0 ? tn_env : (a.batch == 1 && w_bytes > a_bytes_phys);
  }
  // algorithmic bytes: every operand once (a conv reads its image once, not once per tap)
  const double a_elems = g.amode == A_DENSE ? (double)g.M * g.K : (double)(g.M / (g.Ho * g.Wo)) * g.H * g.W * g.Cin;
  char tag[160] = "";
  if (ctx->prof_detail) {
    const char* kern = conv_halo_supported(a) ? "halo" : (gemm_ws_supported(a) ? "ws" : (gemm_pp_supported(a) ? "pp" : "igemm"));
    if (g.amode == A_DENSE)
      snprintf(tag, sizeof(tag), "M%d_N%d_K%d_b%d_act%d_res%d_ln%d_sk%d_%s", g.M, g.N, g.K, g.batch, g.act, g.residual ? 1 : 0, g.ln_rs ? 1 : 0, a.splitk, kern);
    else
      snprintf(tag, sizeof(tag), "conv%d_B%d_%dx%d_Cin%d_Cout%d_res%d_sk%d_%s", g.amode, g.M / (g.Ho * g.Wo), g.H, g.W, g.Cin, g.N, g.residual ? 1 : 0, a.splitk, kern);
  }
  ProfScope ps(ctx, prof_kind, s, 2.0 * g.M * (double)g.N * g.K * g.batch,
               2.0 * (a_elems + (double)g.N * g.K + (double)g.M * g.N) * g.batch, tag);
  if (conv_halo_supported(a)) {
    // 16 x 16 pixel blocks x channel tiles; splitk partitions the 64-channel chunks
    const int blocks = (a.M / 256) * cdiv(a.N, conv_halo_bn(a));
    launch_conv_halo(a, dim3(blocks, 1, a.splitk), s);
  } else if (gemm_ws_supported(a)) {
    launch_gemm_ws(ctx, a, s);
  } else if (gemm_pp_supported(a)) {
    launch_gemm_pp(a, s);
  } else {
    const int bn = pick_bn(a);
    const int tiles = cdiv(a.M, BM) * cdiv(a.N, bn);
    dim3 grid(tiles, a.batch, a.splitk);
    switch (bn) {
      case 32: launch_bn<32>(a, grid, s); break;
      case 64: launch_bn<64>(a, grid, s); break;
      case 128: launch_bn<128>(a, grid, s); break;
      default: launch_bn<160>(a, grid, s); break;
    }
  }
  check_launch("igemm");
  if (a.splitk > 1) {
    int64_t total = (int64_t)a.M * (a.N / 4);
    dim3 rg((unsigned)std::min<int64_t>((total + 255) / 256, 2048), a.batch);
    hipLaunchKernelGGL(splitk_reduce_kernel, rg, dim3(256), 0, s, a);
    check_launch("splitk_reduce");
  }
}

// split-K factor gemm_auto() uses for g (1 = none)
static int plan_splitk(const GemmArgs& g) {
  if (conv_halo_supported(g)) {
    const int64_t blocks = (int64_t)(g.M / 256) * cdiv(g.N, conv_halo_bn(g));
    const int CC = g.Cin / 64;
    static const int tgt = getenv("SVG_HALO_SPLIT_TGT") ? atoi(getenv("SVG_HALO_SPLIT_TGT")) : 320;
    if (blocks < 192 && CC >= 4) return (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>((tgt + blocks - 1) / blocks, CC / 2), 8));
    return 1;
  }
  if (gemm_ws_supported(g) || gemm_pp_supported(g)) return 1;
  const int bn = pick_bn(g);
  const int64_t blocks = (int64_t)cdiv(g.M, BM) * cdiv(g.N, bn) * g.batch;
  const int KT = cdiv(g.K, BK);
  {
    const int force = (int)svg_env_i64("SVG_IGEMM_SK", 0);     // experiments: split-K of the tiled kernel for launches below 192 tiles
    if (force > 0 && blocks < 192 && KT >= 8) return std::min(force, KT / 4);
  }
  if (blocks < 192 && KT >= 8) {
    const int tgt = g.amode != A_DENSE ? 448 : 384;
    const int sk = (int)std::min<int64_t>((tgt + blocks - 1) / blocks, KT / 4);
    return std::max(1, std::min(sk, 16));
  }
  // about one workgroup (4 waves) per CU and a long K: a single wave per SIMD cannot hide its own load phases, so
  // split in two for two co-resident workgroups (same-box A/B at 16 x 16 x 1280 convs: 0.149 -> 0.122 ms)
  if (blocks < 300 && KT >= 64) return 2;
  return 1;
}

bool gemm_fused_qkv_supported(const GemmArgs& g) { return g.vt_out != nullptr && gemm_ws_supported(g); }

void gemm_auto(svg_ctx* ctx, GemmArgs g, hipStream_t s, int prof_kind) {
  if (g.n_valid <= 0) g.n_valid = g.N;
  SVG_CHECK(!g.vt_out || gemm_ws_supported(g), "gemm: a fused q | k | V^T problem (vt_out) needs the weight-stationary kernel (ask gemm_fused_qkv_supported)");
  g.splitk = 1;
  const int sk = plan_splitk(g);
  g.splitk = sk;
  if (sk > 1) {
    SVG_CHECK(!g.gn_part, "gemm: GroupNorm statistics cannot be emitted by a split-K launch (ask gemm_emits_gn first)");
    ctx->arena.push();
    g.slabs = ctx->arena.get<float>((int64_t)sk * g.batch * g.M * g.N);
    launch_gemm(ctx, g, s, prof_kind);
    ctx->arena.pop();   // stream order protects the slabs until the reduce has run
  } else {
    launch_gemm(ctx, g, s, prof_kind);
  }
}

void pack_conv3x3(const float* w, h16* out, int O, int I, int Opad, int Ipad, hipStream_t s) {
  int64_t total = (int64_t)Opad * 9 * Ipad;
  hipLaunchKernelGGL(pack_conv3x3_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, s, w, out, O, I, Opad, Ipad);
  check_launch("pack_conv3x3");
}
void pack_linear(const float* w, h16* out, int N, int K, int Npad, hipStream_t s) {
  int64_t total = (int64_t)Npad * K;
  hipLaunchKernelGGL(pack_linear_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, s, w, out, N, K, Npad);
  check_launch("pack_linear");
}
void pack_geglu(const float* w, const float* b, h16* wout, float* bout, int F, int K, hipStream_t s) {
  int64_t total = (int64_t)2 * F * K;
  hipLaunchKernelGGL(pack_geglu_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, s, w, b, wout, bout, F, K);
  check_launch("pack_geglu");
}

}  // namespace SDNS
