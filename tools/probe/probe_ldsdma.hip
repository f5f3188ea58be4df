// What does an LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KiB per wave-instruction) cost on gfx950 — in bytes per clock per CU, and
// in matrix-pipe time of the wave that shares its SIMD?  One 512-thread workgroup per CU (waves w and w + 4 share a SIMD, as in
// conv_halo.hip / gemm_pp.hip): waves 0-3 load, waves 4-7 multiply, alone and together.  The question behind it:
// every ping-pong kernel here measures  step time = matrix cycles + ~100 cycles per DMA instruction of the SIMD's two waves.
// Is that the L2 -> LDS path's bandwidth (then fewer BYTES per MFMA is the only lever) or issue blocking (then placement is)?
//   hipcc --offload-arch=gfx950 -O3 -o probe_ldsdma tools/probe/probe_ldsdma.hip && ./probe_ldsdma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void dma16(v4i srd, unsigned voff, int soff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(voff), "s"(srd), "s"(soff), "s"(lds_addr) : "memory");
}

// loader: 0 none | 1 LDS-DMA | 2 global_load_dwordx4 into registers | 3 ds_read_b128 of LDS
// n_load / n_mfma: loop counts of the loading waves (0-3, or all 8 when all8) and of the multiplying waves (4-7)
// span: bytes of the source each workgroup walks through (<= 64 KiB: L2 / L1 resident after the first pass; large: streams from HBM)
template <int LOADER, int INFLIGHT, int AG = 0>
__global__ void __launch_bounds__(512) k(const char* src, long long span, long long wg_stride, int n_load, int n_mfma, int all8, long long* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint64_t p = (uint64_t)(src + (long long)blockIdx.x * wg_stride);
  const v4i srd = {(int)(unsigned)p, (int)((p >> 32) & 0xffff), (int)span, 0x00020000};
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(tid * 0.001f + j); b[j] = (__bf16)(j * 0.5f + lane * 0.01f); }
  const bool loader = all8 || wid < 4;
  const int nl = all8 ? 8 : 4;
  const int lw = all8 ? wid : wid;           // loading wave index
  __syncthreads();
  long long t0 = __builtin_readcyclecounter(), t1 = t0;
  if (loader && LOADER != 0 && n_load > 0) {
    f32x4 sink = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned off = (unsigned)(lw * 1024 + lane * 16);
    const unsigned step = (unsigned)(nl * 1024);
    for (int it = 0; it < n_load; ++it) {
      if (LOADER == 1) {
        dma16(srd, off, 0, lds0 + (unsigned)(wid * 16 + (it & 15)) * 1024u);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(INFLIGHT) : "memory");
      } else if (LOADER == 2) {
        f32x4 v;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(off), "s"(srd) : "memory");
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(INFLIGHT) : "memory");
        asm volatile("" :: "v"(v));
      } else {
        f32x4 v = *(const f32x4*)(smem + ((wid * 16 + (it & 15)) * 1024 + lane * 16));
        asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(INFLIGHT > 15 ? 15 : INFLIGHT) : "memory");
        asm volatile("" :: "v"(v));
      }
      off += step;
      if ((long long)off + 1024 > span) off = (unsigned)(lw * 1024 + lane * 16);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    t1 = __builtin_readcyclecounter();
  } else if (!loader && n_mfma > 0) {
    for (int it = 0; it < n_mfma; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (AG) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[u]) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a), "v"(b));
      }
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    t1 = __builtin_readcyclecounter();
  }
  if (lane == 0) out[(long long)blockIdx.x * 8 + wid] = t1 - t0;
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0];
  if (s == 12345.678f) out[0] = 0;
}

struct Res { double load_cyc, mfma_cyc; };

template <int LOADER, int INFLIGHT, int AG = 0>
Res run(const char* src, long long span, long long wg_stride, int n_load, int n_mfma, int all8) {
  const int nwg = 256;
  long long* d;
  CK(hipMalloc(&d, nwg * 8 * sizeof(long long)));
  CK(hipFuncSetAttribute((const void*)k<LOADER, INFLIGHT, AG>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((k<LOADER, INFLIGHT, AG>), dim3(nwg), dim3(512), 136 * 1024, 0, src, span, wg_stride, n_load, n_mfma, all8, d);
    CK(hipDeviceSynchronize());
  }
  std::vector<long long> h(nwg * 8);
  CK(hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
  CK(hipFree(d));
  std::vector<double> l, m;
  for (int w = 0; w < nwg; ++w)
    for (int i = 0; i < 8; ++i) {
      if (all8 || i < 4) l.push_back((double)h[w * 8 + i]); else m.push_back((double)h[w * 8 + i]);
    }
  auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  return Res{med(l), med(m)};
}

int main() {
  const long long big = 2LL << 30;
  char* src;
  CK(hipMalloc(&src, big));
  CK(hipMemset(src, 1, big));
  const int NM = 4096;           // x 8 MFMAs
  printf("one 512-thread workgroup per CU, 256 workgroups; cycles = s_memtime ticks (shader clock); medians over waves\n");
  {
    Res r = run<0, 0>(src, 65536, 65536, 0, NM, 0);
    printf("MFMA alone (waves 4-7, 16x16x32 bf16, 8 accumulators): %.2f cycles per MFMA\n", r.mfma_cyc / (NM * 8.0));
  }
  auto report = [&](const char* name, Res lo, int n_load, int nl, Res hi, int n_mfma) {
    // lo: loaders measured while the MFMA waves still run (MFMA loop long); hi: MFMA waves measured while the loaders still run
    const double cyc_per_load = lo.load_cyc / n_load;
    printf("%-58s load: %6.1f cyc per wave-instruction = %5.1f B/clk/CU | partner MFMA: %.2f cyc per MFMA\n", name, cyc_per_load,
           nl * 1024.0 / cyc_per_load, hi.mfma_cyc / (n_mfma * 8.0));
  };
  // L2-resident source: every workgroup re-reads its own 48 KiB; HBM: every workgroup streams its own 8 MiB
  for (int hbm = 0; hbm < 2; ++hbm) {
    const long long span = hbm ? (8LL << 20) : 49152, stride = hbm ? (8LL << 20) : 65536;
    const char* tag = hbm ? "HBM stream" : "L2-resident";
    char name[128];
    {
      Res a = run<1, 8>(src, span, stride, 4096, 0, 0);
      snprintf(name, sizeof name, "[%s] LDS-DMA alone, 4 waves, 8 in flight", tag);
      printf("%-58s load: %6.1f cyc per wave-instruction = %5.1f B/clk/CU\n", name, a.load_cyc / 4096, 4 * 1024.0 / (a.load_cyc / 4096));
      Res b = run<1, 8>(src, span, stride, 4096, 0, 1);
      snprintf(name, sizeof name, "[%s] LDS-DMA alone, 8 waves, 8 in flight", tag);
      printf("%-58s load: %6.1f cyc per wave-instruction = %5.1f B/clk/CU\n", name, b.load_cyc / 4096, 8 * 1024.0 / (b.load_cyc / 4096));
      Res c = run<1, 2>(src, span, stride, 4096, 0, 1);
      snprintf(name, sizeof name, "[%s] LDS-DMA alone, 8 waves, 2 in flight", tag);
      printf("%-58s load: %6.1f cyc per wave-instruction = %5.1f B/clk/CU\n", name, c.load_cyc / 4096, 8 * 1024.0 / (c.load_cyc / 4096));
    }
    {
      Res lo = run<1, 8>(src, span, stride, 2048, 4 * NM, 0), hi = run<1, 8>(src, span, stride, 1 << 16, NM / 4, 0);
      snprintf(name, sizeof name, "[%s] LDS-DMA (waves 0-3, 8 in flight) beside MFMA (4-7)", tag);
      report(name, lo, 2048, 4, hi, NM / 4);
    }
    {
      Res lo = run<1, 2>(src, span, stride, 2048, 4 * NM, 0), hi = run<1, 2>(src, span, stride, 1 << 16, NM / 4, 0);
      snprintf(name, sizeof name, "[%s] LDS-DMA (waves 0-3, 2 in flight) beside MFMA (4-7)", tag);
      report(name, lo, 2048, 4, hi, NM / 4);
    }
    {
      Res lo = run<2, 8>(src, span, stride, 2048, 4 * NM, 0), hi = run<2, 8>(src, span, stride, 1 << 16, NM / 4, 0);
      snprintf(name, sizeof name, "[%s] register loads (waves 0-3, 8 in flight) beside MFMA", tag);
      report(name, lo, 2048, 4, hi, NM / 4);
    }
  }
  {
    Res lo = run<3, 8>(src, 65536, 65536, 4096, 4 * NM, 0), hi = run<3, 8>(src, 65536, 65536, 1 << 17, NM / 4, 0);
    report("ds_read_b128 (waves 0-3, 8 in flight) beside MFMA (4-7)", lo, 4096, 4, hi, NM / 4);
  }
  {
    // the same with the accumulators in AGPRs (the MFMA's C / D traffic on the other half of the register file than the LDS returns)
    Res hv = run<3, 8, 0>(src, 65536, 65536, 1 << 17, NM / 4, 0), ha = run<3, 8, 1>(src, 65536, 65536, 1 << 17, NM / 4, 0);
    printf("MFMA beside a ds_read_b128 partner: VGPR accumulators %.2f, AGPR accumulators %.2f cycles per MFMA\n", hv.mfma_cyc / (NM / 4 * 8.0), ha.mfma_cyc / (NM / 4 * 8.0));
    Res dv = run<1, 8, 0>(src, 49152, 65536, 1 << 16, NM / 4, 0), da = run<1, 8, 1>(src, 49152, 65536, 1 << 16, NM / 4, 0);
    printf("MFMA beside an LDS-DMA partner:     VGPR accumulators %.2f, AGPR accumulators %.2f cycles per MFMA\n", dv.mfma_cyc / (NM / 4 * 8.0), da.mfma_cyc / (NM / 4 * 8.0));
  }
  CK(hipFree(src));
  return 0;
}
