// Cycles per v_mfma_f32_16x16x32_bf16 issued back to back by ONE wave per SIMD, by where the accumulator lives (VGPR form "+v" as
// -amdgpu-mfma-vgpr-form builds it, AGPR form "+a") and by how many independent accumulators / distinct operand registers rotate.
// probe_ldsdma.hip measured 18.0 cycles in the VGPR form against the 16 of MI355X_MICROARCH.md: is that the register form?
//   hipcc --offload-arch=gfx950 -O3 -o probe_mfma_acc tools/probe/probe_mfma_acc.hip && ./probe_mfma_acc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// FORM 0: VGPR accumulators, 1: AGPR accumulators.  NACC accumulators in rotation, NOP distinct (a, b) operand pairs in rotation.
// SHAPE 0: 16x16x32, 1: 32x32x16.  WAVES: waves per SIMD (1 or 2).
template <int FORM, int NACC, int NOP, int SHAPE>
__global__ void __launch_bounds__(512) k(long long* out, int iters, int waves_per_simd, float* sink) {
  const int tid = threadIdx.x, wid = tid >> 6;
  bf16x8 a[NOP], b[NOP];
  for (int i = 0; i < NOP; ++i)
    for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)(tid * 0.001f + j + i); b[i][j] = (__bf16)(j * 0.5f + tid * 0.01f - i); }
  f32x4 acc4[NACC];
  f32x16 acc16[NACC];
  for (int i = 0; i < NACC; ++i) {
    acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < 16; ++j) acc16[i][j] = 0.f;
  }
  const bool active = wid < 4 * waves_per_simd;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (active) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (SHAPE == 0) {
          if (FORM == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc4[u % NACC]) : "v"(a[u % NOP]), "v"(b[u % NOP]));
          else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc4[u % NACC]) : "v"(a[u % NOP]), "v"(b[u % NOP]));
        } else {
          if (FORM == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc16[u % NACC]) : "v"(a[u % NOP]), "v"(b[u % NOP]));
          else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc16[u % NACC]) : "v"(a[u % NOP]), "v"(b[u % NOP]));
        }
      }
    }
  }
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  if ((tid & 63) == 0) out[(long long)blockIdx.x * 8 + wid] = t1 - t0;
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc4[i][0] + acc16[i][0];
  if (s == 12345.678f) sink[0] = s;
}

template <int FORM, int NACC, int NOP, int SHAPE>
void run(const char* name, int wps) {
  const int nwg = 256, iters = 2048;
  long long* d; float* sink;
  CK(hipMalloc(&d, nwg * 8 * sizeof(long long)));
  CK(hipMalloc(&sink, 64));
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((k<FORM, NACC, NOP, SHAPE>), dim3(nwg), dim3(512), 0, 0, d, iters, wps, sink);
    CK(hipDeviceSynchronize());
  }
  std::vector<long long> h(nwg * 8);
  CK(hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
  std::vector<double> v;
  for (int w = 0; w < nwg; ++w) for (int i = 0; i < 4 * wps; ++i) v.push_back((double)h[w * 8 + i]);
  std::sort(v.begin(), v.end());
  const double per = v[v.size() / 2] / (iters * 16.0);
  printf("%-86s %6.2f cycles per MFMA per wave = %6.2f per SIMD\n", name, per, per / wps);
  CK(hipFree(d)); CK(hipFree(sink));
}

int main() {
  printf("256 workgroups (one per CU); s_memtime ticks; median over the issuing waves\n");
  run<0, 8, 1, 0>("16x16x32 VGPR acc, 8 accumulators, 1 operand pair, 1 wave/SIMD", 1);
  run<1, 8, 1, 0>("16x16x32 AGPR acc, 8 accumulators, 1 operand pair, 1 wave/SIMD", 1);
  run<0, 8, 4, 0>("16x16x32 VGPR acc, 8 accumulators, 4 operand pairs, 1 wave/SIMD", 1);
  run<1, 8, 4, 0>("16x16x32 AGPR acc, 8 accumulators, 4 operand pairs, 1 wave/SIMD", 1);
  run<0, 16, 4, 0>("16x16x32 VGPR acc, 16 accumulators, 4 operand pairs, 1 wave/SIMD", 1);
  run<1, 16, 4, 0>("16x16x32 AGPR acc, 16 accumulators, 4 operand pairs, 1 wave/SIMD", 1);
  run<0, 2, 1, 0>("16x16x32 VGPR acc, 2 accumulators, 1 operand pair, 1 wave/SIMD", 1);
  run<1, 2, 1, 0>("16x16x32 AGPR acc, 2 accumulators, 1 operand pair, 1 wave/SIMD", 1);
  run<0, 8, 4, 0>("16x16x32 VGPR acc, 8 accumulators, 4 operand pairs, 2 waves/SIMD", 2);
  run<1, 8, 4, 0>("16x16x32 AGPR acc, 8 accumulators, 4 operand pairs, 2 waves/SIMD", 2);
  run<0, 4, 2, 1>("32x32x16 VGPR acc, 4 accumulators, 2 operand pairs, 1 wave/SIMD", 1);
  run<1, 4, 2, 1>("32x32x16 AGPR acc, 4 accumulators, 2 operand pairs, 1 wave/SIMD", 1);
  return 0;
}
