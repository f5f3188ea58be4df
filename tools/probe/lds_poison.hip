// Fills the LDS of every CU with a chosen 32-bit pattern, over and over, from a stream of its own: whatever kernel runs next on a CU inherits
// that content.  A kernel whose result depends on LDS it never wrote (a reduction scratch with unwritten entries, a padded fragment row)
// then shows LARGE differences instead of last-bit ones.  Built as a shared library and driven from Python (tools/lds_poison_test.py).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/probe/bin/liblds_poison.so tools/probe/lds_poison.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void __launch_bounds__(256) poison_kernel(uint32_t pattern, int bytes, int spin) {
  extern __shared__ uint32_t lds[];
  for (int i = threadIdx.x; i < bytes / 4; i += 256) lds[i] = pattern;
  __syncthreads();
  // stay resident a little so that the blocks spread over all compute units
  uint32_t acc = 0;
  for (int k = 0; k < spin; ++k) acc += lds[(threadIdx.x + k) % (bytes / 4)];
  if (acc == 0x12345u) lds[0] = acc;
}

extern "C" int lds_poison(uint32_t pattern, int bytes, int blocks, int spin, void* stream) {
  static bool init = false;
  if (!init) { if (hipFuncSetAttribute((const void*)poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1; init = true; }
  hipLaunchKernelGGL(poison_kernel, dim3(blocks), dim3(256), (size_t)bytes, (hipStream_t)stream, pattern, bytes, spin);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
