// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 (OCP e4m3, E8M0 block scales) operand and scale lane maps on gfx950.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe/probe_mx.hip -o gpurun_out/probe_mx ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// e4m3 encode of small non-negative integers 0..15 (exact)
__host__ __device__ inline uint8_t e4m3(int v) {
  if (v == 0) return 0;
  int s = v < 0; if (s) v = -v;
  int e = 0; while ((v >> (e + 1)) != 0) ++e;          // floor(log2 v)
  int man = ((v << 3) >> e) & 7;                        // 3 mantissa bits (exact for v <= 15)
  return (uint8_t)((s << 7) | ((e + 7) << 3) | man);
}

__global__ void probe(const uint8_t* A, const uint8_t* B, const uint8_t* sa, const uint8_t* sb, float* out, int mode) {
  const int lane = threadIdx.x;
  const int r = lane & 15, kb = lane >> 4;
  v8i a, b;
  // hypothesis: lane (r, kb) holds A[r][32 kb .. 32 kb + 31] and B[32 kb .. +31][r] (B stored here as Bt[n][k])
  const int* pa = (const int*)(A + r * 128 + kb * 32);
  const int* pb = (const int*)(B + r * 128 + kb * 32);
  for (int i = 0; i < 8; ++i) { a[i] = pa[i]; b[i] = pb[i]; }
  int scale_a = sa[r * 4 + kb], scale_b = sb[r * 4 + kb];       // byte 0 of the operand = E8M0 scale of this lane's block
  if (mode == 1) { scale_a |= 0x7f7f7f00; scale_b |= 0x7f7f7f00; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = c[i];
}

int main() {
  std::vector<uint8_t> A(16 * 128), Bt(16 * 128), sa(64, 127), sb(64, 127);
  std::vector<float> Af(16 * 128), Bf(16 * 128);
  for (int r = 0; r < 16; ++r) for (int k = 0; k < 128; ++k) {
    int va = (r * 7 + k * 3) % 5, vb = (r * 5 + k) % 4 + (k % 3 == 0);
    A[r * 128 + k] = e4m3(va); Af[r * 128 + k] = (float)va;
    Bt[r * 128 + k] = e4m3(vb); Bf[r * 128 + k] = (float)vb;
  }
  uint8_t *dA, *dB, *dsa, *dsb; float* dout;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, Bt.size()); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dout, 256 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), Bt.size(), hipMemcpyHostToDevice);
  for (int test = 0; test < 4; ++test) {
    std::fill(sa.begin(), sa.end(), 127); std::fill(sb.begin(), sb.end(), 127);
    if (test == 1) for (auto& v : sa) v = 128;                       // all A blocks x2
    if (test == 2) for (int r = 0; r < 16; ++r) sa[r * 4 + 1] = 129;    // A block kb=1 x4, every row
    if (test == 3) sb[5 * 4 + 2] = 126;                                 // B column 5, block 2 x0.5
    hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dout, test == 0 ? 1 : 0);
    std::vector<float> out(256);
    hipMemcpy(out.data(), dout, 1024, hipMemcpyDeviceToHost);
    // expected with the hypothesised maps; C/D: col = lane & 15 (-> n), row = (lane >> 4) * 4 + i (-> m)
    int bad = 0; double maxerr = 0;
    for (int lane = 0; lane < 64; ++lane) for (int i = 0; i < 4; ++i) {
      const int n = lane & 15, m = (lane >> 4) * 4 + i;
      double ref = 0;
      for (int k = 0; k < 128; ++k) {
        double wa = std::ldexp(1.0, sa[m * 4 + k / 32] - 127), wb = std::ldexp(1.0, sb[n * 4 + k / 32] - 127);
        ref += Af[m * 128 + k] * wa * Bf[n * 128 + k] * wb;
      }
      double e = std::fabs(ref - out[lane * 4 + i]);
      if (e > 1e-3 * std::fabs(ref) + 1e-3) ++bad;
      if (e > maxerr) maxerr = e;
    }
    printf("test %d: %d / 256 mismatches, max |err| %.4g  (out[0..3] = %.1f %.1f %.1f %.1f)\n", test, bad, maxerr, out[0], out[1], out[2], out[3]);
  }
  return 0;
}
