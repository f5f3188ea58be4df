// MX block-scaled fp8 GEMM for gfx950 (BASELINE configs[4]: "fp8 MFMA on CDNA4"): C[M,N] = epi(A[M,K] W[N,K]^T) with both
// operands in OCP e4m3 and one E8M0 scale per 32 consecutive K elements (the OCP MX format), accumulated in f32 by
// v_mfma_scale_f32_16x16x128_f8f6f4 — the only fp8 MFMA form that runs at twice the bf16 rate (MI355X_MICROARCH.md).
//
// Operand map of the instruction, probed on MI355X with exact integer data (tools/probe/probe_mx.hip, profiles/README.md):
//   data : lane (r = lane & 15, q = lane >> 4) supplies 32 bytes of row r: K elements [16 q, 16 q + 16) and
//          [64 + 16 q, 64 + 16 q + 16) of the 128-wide step (NOT 32 consecutive ones);
//   scale: byte 0 of the scale operand of lane (r, q) is the E8M0 scale of K block q = elements [32 q, 32 q + 32) of row r.
//   C/D  : as every 16x16 MFMA (column = lane & 15, rows 4 (lane >> 4) + i).
// So with K-contiguous rows in memory, 128-byte LDS rows and one scale byte per 32 elements, a lane reads the 16-byte chunks
// q and q + 4 of its row and passes scale byte q.
//
// Kernel: gemm.hip's structure at K step 128: tile 128 x BN x 128 per 256-thread workgroup (2 x 2 waves, each 64 x BN/2), two LDS
// stages filled by LDS-direct buffer loads, the same XOR swizzle (rows are 128 B here too), weight fragment as the A operand so
// that a lane owns 4 consecutive output columns, shared tile epilogue (bias / residual / activation, h16 or f32 out).
// The scales ([rows][K/32] bytes) ride in registers: one byte load per fragment row and K step.
#include "igemm_epi.h"
#include <cstdlib>

namespace SDNS {

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));

struct Fp8Args {
  const uint8_t* A; const uint8_t* As;     // [M][K] e4m3, [M][K/32] E8M0
  const uint8_t* W; const uint8_t* Ws;     // [N][K] e4m3, [N][K/32] E8M0
  GemmArgs g;                              // M, N, K, epilogue (bias, residual, act, C, ldc, out_f32); A / Wt unused
};

template <int BN>
__global__ void __launch_bounds__(256, 2) gemm_fp8_kernel(const Fp8Args a) {
  constexpr int NT = BN / 32, MT = 4, BIT = BN / 32;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const GemmArgs& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int l15 = lane & 15, lq = lane >> 4;

  const int tiles_n = (g.N + BN - 1) / BN;
  int tile;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int KT = g.K >> 7;                                  // K % 128 == 0

  constexpr unsigned INVALID = 0x80000000u;
  const int r0 = tid >> 3;                                  // 0..31
  const int c = (tid & 7) ^ ((r0 >> 1) & 7);                // logical 16-byte chunk this lane fetches (swizzle on the source)
  const int wave_u = __builtin_amdgcn_readfirstlane(wid);
  const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, (unsigned)((int64_t)g.M * g.K), 0x00020000);
  const __amdgpu_buffer_rsrc_t srdB = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, (unsigned)((int64_t)g.N * g.K), 0x00020000);
  unsigned a_voff[4], b_voff[5];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + r0 + 32 * i;
    a_voff[i] = m < g.M ? (unsigned)(m * g.K + c * 16) : INVALID;
  }
#pragma unroll
  for (int i = 0; i < BIT; ++i) {
    const int n = n0 + r0 + 32 * i;
    b_voff[i] = n < g.N ? (unsigned)(n * g.K + c * 16) : INVALID;
  }
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  auto dma_tiles = [&](int kt, int buf) {
    char* sa = smem + buf * (A_BYTES + B_BYTES) + wave_u * 1024;
    char* sb = sa + A_BYTES;
    const int ksoff = kt * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (lds_ptr_t)(sa + i * 4096), 16, a_voff[i], ksoff, 0, 0);
#pragma unroll
    for (int i = 0; i < BIT; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(srdB, (lds_ptr_t)(sb + i * 4096), 16, b_voff[i], ksoff, 0, 0);
  };

  // scale bytes of this lane's fragment rows: A rows m0 + wm*64 + 16 i + l15, W rows n0 + wn*(BN/2) + 16 j + l15; block lq of step kt
  const int kb_row = g.K >> 5;
  const uint8_t* as_p[MT];
  const uint8_t* ws_p[NT];
  bool as_ok[MT], ws_ok[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) { const int m = m0 + wm * 64 + i * 16 + l15; as_ok[i] = m < g.M; as_p[i] = a.As + (int64_t)(as_ok[i] ? m : 0) * kb_row + lq; }
#pragma unroll
  for (int j = 0; j < NT; ++j) { const int n = n0 + wn * (BN / 2) + j * 16 + l15; ws_ok[j] = n < g.N; ws_p[j] = a.Ws + (int64_t)(ws_ok[j] ? n : 0) * kb_row + lq; }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto frag = [&](const char* base, int row) -> v8i {        // chunks lq and lq + 4 of a 128-byte LDS row
    const u32x4 lo = *(const u32x4*)(base + lds_off(row, lq));
    const u32x4 hi = *(const u32x4*)(base + lds_off(row, lq + 4));
    return v8i{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };

  if (KT > 0) {
    dma_tiles(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < KT) dma_tiles(kt + 1, buf ^ 1);
      int sa_[MT], sb_[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) sa_[i] = as_ok[i] ? as_p[i][kt * 4] : 127;
#pragma unroll
      for (int j = 0; j < NT; ++j) sb_[j] = ws_ok[j] ? ws_p[j][kt * 4] : 127;
      const char* sA = smem + buf * (A_BYTES + B_BYTES);
      const char* sB = sA + A_BYTES;
      v8i af[MT], bfr[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = frag(sA, wm * 64 + i * 16 + l15);
#pragma unroll
      for (int j = 0; j < NT; ++j) bfr[j] = frag(sB, wn * (BN / 2) + j * 16 + l15);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bfr[j], af[i], acc[i][j], 0, 0, 0, sb_[j], 0, sa_[i]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  epi_tile<MT, NT>(g, 0, m0 + wm * 64 + l15, 16, n0 + wn * (BN / 2) + lq * 4, acc);
}

// ---- MX quantiser: x (rows, K) bf16 or f32 -> e4m3 (rows, K) + E8M0 (rows, K/32).  One wave-quarter (16 lanes x 2 elements) per
// block of 32: shared exponent = floor(log2(amax)) - 8 (e4m3's largest power of two is 2^8), elements = RNE(x * 2^-shared),
// saturated at +-448 (OCP MX v1.0, section 6.3).
template <typename T>
__global__ void __launch_bounds__(256) quant_mx_kernel(const T* __restrict__ x, int ldx, uint8_t* __restrict__ q, uint8_t* __restrict__ sc,
                                                       int64_t rows, int K) {
  const int kb = K >> 5;
  const int64_t nblk = rows * kb;
  const int sub = threadIdx.x & 15;                          // 16 lanes per block, 2 consecutive elements each
  for (int64_t b = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); b < nblk; b += (int64_t)gridDim.x * 16) {
    const int64_t row = b / kb;
    const int k0 = (int)(b - row * kb) * 32 + sub * 2;
    const float v0 = (float)x[row * ldx + k0], v1 = (float)x[row * ldx + k0 + 1];
    float am = fmaxf(fabsf(v0), fabsf(v1));
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) am = fmaxf(am, __shfl_xor(am, o));
    int e = am > 0.f ? ((__float_as_int(am) >> 23) & 0xff) - 127 - 8 : -127;     // floor(log2(amax)) - emax_elem (8 for e4m3)
    e = e < -127 ? -127 : (e > 126 ? 126 : e);
    const float inv = __int_as_float((127 - e) << 23);                         // 2^-e: exponent field 1 .. 254
    const float s0 = fminf(fmaxf(v0 * inv, -448.f), 448.f), s1 = fminf(fmaxf(v1 * inv, -448.f), 448.f);
    const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(s0, s1, 0, false);
    *(uint16_t*)(q + row * K + k0) = (uint16_t)(pk & 0xffff);
    if (sub == 0) sc[b] = (uint8_t)(e + 127);
  }
}

template <int BN>
void launch_fp8(const Fp8Args& a, hipStream_t s) {
  constexpr int smem = 2 * (BM * 128 + BN * 128);
  const int tiles = cdiv(a.g.M, BM) * cdiv(a.g.N, BN);
  hipLaunchKernelGGL((gemm_fp8_kernel<BN>), dim3(tiles), dim3(256), smem, s, a);
}

}  // namespace

void gemm_fp8_init_device() {
  HIP_OK(hipFuncSetAttribute((const void*)gemm_fp8_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (BM * 128 + 128 * 128)));
  HIP_OK(hipFuncSetAttribute((const void*)gemm_fp8_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (BM * 128 + 64 * 128)));
}

void quant_mx_h16(svg_ctx* ctx, const h16* x, int ldx, uint8_t* q, uint8_t* sc, int64_t rows, int K, hipStream_t s) {
  SVG_CHECK(K % 32 == 0 && ldx % 2 == 0, "quant_mx: K=%d must be a multiple of 32", K);
  if (!SVG_LAUNCHING(ctx)) return;
  char tag[64];
  snprintf(tag, sizeof(tag), "quant_mx_rows%lld_K%d", (long long)rows, K);
  ProfScope ps(ctx, PK_ELT, s, 0, 3.0 * rows * K, tag);
  const int64_t nblk = rows * (K / 32);
  hipLaunchKernelGGL((quant_mx_kernel<h16>), dim3((unsigned)std::min<int64_t>((nblk + 15) / 16, 16384)), dim3(256), 0, s, x, ldx, q, sc, rows, K);
  check_launch("quant_mx");
}
void quant_mx_f32(const float* x, int ldx, uint8_t* q, uint8_t* sc, int64_t rows, int K, hipStream_t s) {
  const int64_t nblk = rows * (K / 32);
  hipLaunchKernelGGL((quant_mx_kernel<float>), dim3((unsigned)std::min<int64_t>((nblk + 15) / 16, 16384)), dim3(256), 0, s, x, ldx, q, sc, rows, K);
  check_launch("quant_mx");
}

bool gemm_fp8_supported(int M, int N, int K) { return K % 128 == 0 && N % 4 == 0 && (int64_t)M * K < (1LL << 31) && (int64_t)N * K < (1LL << 31); }

// g: M, N, K and the epilogue fields of a dense GemmArgs (bias, residual, act != GEGLU, C, ldc, out_f32); operands in MX fp8
void gemm_fp8(svg_ctx* ctx, const uint8_t* A, const uint8_t* As, const uint8_t* W, const uint8_t* Ws, const GemmArgs& g, hipStream_t s) {
  SVG_CHECK(gemm_fp8_supported(g.M, g.N, g.K) && g.act != ACT_GEGLU && g.batch == 1 && !g.ln_rs && !g.gn_part,
            "gemm_fp8: M %d N %d K %d (K %% 128, no GEGLU / folded LayerNorm / GroupNorm sums) unsupported", g.M, g.N, g.K);
  if (!SVG_LAUNCHING(ctx)) return;
  Fp8Args a{A, As, W, Ws, g};
  a.g.splitk = 1; a.g.n_valid = g.N; a.g.gn_part = nullptr;
  char tag[96];
  snprintf(tag, sizeof(tag), "fp8_M%d_N%d_K%d_act%d_res%d", g.M, g.N, g.K, g.act, g.residual ? 1 : 0);
  ProfScope ps(ctx, PK_GEMM, s, 2.0 * g.M * (double)g.N * g.K, (double)g.M * g.K + (double)g.N * g.K + 2.0 * g.M * g.N, tag);
  if (g.N <= 64) launch_fp8<64>(a, s); else launch_fp8<128>(a, s);
  check_launch("gemm_fp8");
}

}  // namespace SDNS
