// Does VALU issue overlap with MFMA execution on gfx950?  One wave per SIMD and two waves per SIMD, same-wave interleave and
// role-split pairs.  Prints cycles per loop iteration (s_memtime, wave 0 of workgroup 0; every CU runs the same workgroup).
//   hipcc --offload-arch=gfx950 -O3 -o probe_overlap tools/probe/probe_overlap.hip && ./probe_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// mode: 0 MFMA only (4 independent accumulators) | 1 v_exp only (NV per iteration) | 2 v_fma only | 3 MFMA + NV v_exp in one stream
//       4 MFMA + NV v_fma in one stream | 5 role split: waves 0-3 MFMA only, waves 4-7 v_exp only | 6 role split with v_fma
// PRIO (role-split modes 5 / 6, round 6): 0 both roles at priority 0 | 1 the MFMA waves at s_setprio 3 | 2 the VALU waves at s_setprio 3
template <int MODE, int NV, int PRIO = 0>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, int iters) {
  const int wid = threadIdx.x >> 6;
  if (PRIO == 1 && wid < 4) __builtin_amdgcn_s_setprio(3);
  if (PRIO == 2 && wid >= 4) __builtin_amdgcn_s_setprio(3);
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(j * 0.5f); }
  float v[NV > 0 ? NV : 1];
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) v[i] = threadIdx.x * 0.01f + i;
  const bool do_m = MODE == 0 || MODE == 3 || MODE == 4 || ((MODE == 5 || MODE == 6) && wid < 4);
  const bool do_e = MODE == 1 || MODE == 3 || (MODE == 5 && wid >= 4);
  const bool do_f = MODE == 2 || MODE == 4 || (MODE == 6 && wid >= 4);
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (do_m) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a), "v"(b));
      if (do_e) {
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
      }
      if (do_f) {
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i]));
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wid] = t1 - t0;
}

template <int MODE, int NV, int PRIO = 0>
void run(const char* name, int threads) {
  const int iters = 2000, grid = 256;
  float* out; long long* cyc;
  (void)hipMalloc(&out, grid * 512 * 4); (void)hipMalloc(&cyc, grid * 8 * 8);
  (void)hipMemset(cyc, 0, grid * 8 * 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, NV, PRIO>), dim3(grid), dim3(threads), 0, 0, out, cyc, iters);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<MODE, NV, PRIO>), dim3(grid), dim3(threads), 0, 0, out, cyc, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(grid * 8);
  (void)hipMemcpy(h.data(), cyc, grid * 8 * 8, hipMemcpyDeviceToHost);
  // s_memtime counts at a fixed 100 MHz-class rate on some parts: report both raw and relative numbers
  printf("%-58s threads %3d  wave0 %8.1f  wave4 %8.1f ticks/iteration   kernel %7.1f ns/iteration\n", name, threads, (double)h[0] / iters,
         (double)h[4] / iters, ms * 1e6 / iters);
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  for (int threads : {256, 512}) {
    run<0, 0>("MFMA only (4 x 32x32x16 per iteration)", threads);
    run<1, 4>("v_exp only (4 x 4)", threads);
    run<2, 8>("v_fma only (4 x 8)", threads);
    run<3, 2>("MFMA + 2 v_exp each, one stream", threads);
    run<3, 4>("MFMA + 4 v_exp each, one stream", threads);
    run<4, 4>("MFMA + 4 v_fma each, one stream", threads);
    run<4, 8>("MFMA + 8 v_fma each, one stream", threads);
  }
  run<5, 4>("role split: waves 0-3 MFMA, waves 4-7 4 x 4 v_exp", 512);
  run<6, 8>("role split: waves 0-3 MFMA, waves 4-7 4 x 8 v_fma", 512);
  run<5, 4, 1>("role split, MFMA waves at s_setprio 3: MFMA | 4 x 4 v_exp", 512);
  run<5, 4, 2>("role split, VALU waves at s_setprio 3: MFMA | 4 x 4 v_exp", 512);
  run<6, 8, 1>("role split, MFMA waves at s_setprio 3: MFMA | 4 x 8 v_fma", 512);
  run<6, 8, 2>("role split, VALU waves at s_setprio 3: MFMA | 4 x 8 v_fma", 512);
  run<5, 2, 1>("role split, MFMA waves at s_setprio 3: MFMA | 4 x 2 v_exp", 512);
  run<5, 2, 0>("role split: MFMA | 4 x 2 v_exp", 512);
  return 0;
}
