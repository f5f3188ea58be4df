// Probe of v_permlane16_swap_b32 on gfx950: prints, per 16-lane row, which (operand, row) each result register holds.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe/probe_permlane16.hip -o gpurun_out/probe_permlane16 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned lane = threadIdx.x;
  unsigned a = 0x100 + lane, b = 0x200 + lane;           // a: operand 1, b: operand 2
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[lane] = r[0]; out[64 + lane] = r[1];
}
int main() {
  unsigned* d; unsigned h[128];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int res = 0; res < 2; ++res)
    for (int row = 0; row < 4; ++row) {
      unsigned v = h[res * 64 + row * 16];
      printf("result %d row %d <- operand %d row %d (lane0 of row holds 0x%x)\n", res, row, (v >> 8), (v & 0xff) / 16, v);
    }
  return 0;
}
