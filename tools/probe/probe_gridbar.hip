// Probe for a layer-walking persistent kernel on MI355X: what does a device-wide barrier cost, and how fast does a weight stream
// run when every 16.7 MB (one 2048 x 2048 f32 matrix) is followed by TWO barriers?
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probe/probe_gridbar.hip -o /tmp/probe_gridbar && /tmp/probe_gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Bar { unsigned cnt; unsigned abort_; unsigned pad[14]; unsigned xcd[8][16]; unsigned rel[8][16]; };

// flat: one counter; barrier k completes at cnt == k * nwg.  Called by ONE lane of a workgroup.
template <int F> __device__ __forceinline__ void rel_fence() {
  if (F == 0) __atomic_thread_fence(__ATOMIC_RELEASE);
  else if (F == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}
template <int F> __device__ __forceinline__ void acq_fence() {
  if (F == 0) __atomic_thread_fence(__ATOMIC_ACQUIRE);
  else if (F == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
template <int F, int SLEEP>
__device__ __forceinline__ bool bar_flat(Bar* b, unsigned target) {
  rel_fence<F>();
  __hip_atomic_fetch_add(&b->cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned spins = 0;
  while (__hip_atomic_load(&b->cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    if (SLEEP) __builtin_amdgcn_s_sleep(1);
    if (++spins > (1u << 22)) { __hip_atomic_store(&b->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
    if ((spins & 1023) == 0 && __hip_atomic_load(&b->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
  }
  acq_fence<F>();
  return true;
}

// two-level: workgroups of group g = blockIdx & 7 (the XCD under round-robin dispatch) count on xcd[g]; the last arrival of a group
// counts on cnt; the last arrival overall releases every group by bumping rel[g]
template <int F, int SLEEP>
__device__ __forceinline__ bool bar_two(Bar* b, unsigned k, unsigned nwg) {
  const unsigned g = blockIdx.x & 7, per = nwg / 8;
  rel_fence<F>();
  const unsigned a = __hip_atomic_fetch_add(&b->xcd[g][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (a == k * per - 1) {
    const unsigned c = __hip_atomic_fetch_add(&b->cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (c == k * 8 - 1)
      for (int i = 0; i < 8; ++i) __hip_atomic_store(&b->rel[i][0], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  unsigned spins = 0;
  while (__hip_atomic_load(&b->rel[g][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k) {
    if (SLEEP) __builtin_amdgcn_s_sleep(1);
    if (++spins > (1u << 22)) { __hip_atomic_store(&b->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
    if ((spins & 1023) == 0 && __hip_atomic_load(&b->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
  }
  acq_fence<F>();
  return true;
}

// mode 0: barriers only (flat); 1: barriers only (two-level); 2/3: stream 64 KB per workgroup and stage into registers of waves 0..3, the
// barriers by wave 4 (flat / two-level); 4: stream only (no barriers); 5/6: as 2/3 with the next stage's loads in flight across the barriers
template <int F, int SLEEP>
__global__ void __launch_bounds__(320) walk(Bar* b, const f32x4* __restrict__ W, size_t wvec, float* out, int stages, int mode, int bars_per_stage, int pattern) {
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
  const unsigned nwg = gridDim.x;
  unsigned k = 0;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  __shared__ int dead;
  if (tid == 0) dead = 0;
  __syncthreads();
  const bool pre = mode >= 5;                       // 5 / 6: the NEXT stage's 64 KB are requested before this stage's barriers
  const bool two = (mode == 1 || mode == 3 || mode == 6);
  f32x4 w[16];
  auto load = [&](int s) {
    // 64 KB per workgroup and stage: 4 waves x 16 x (64 lanes x 16 B)
    if (pattern == 0) {
      const size_t base = ((size_t)s * nwg + blockIdx.x) * 4096 + wid * 1024 + lane;
#pragma unroll
      for (int i = 0; i < 16; ++i) w[i] = W[(base + i * 64) % wvec];
    } else {
      // a 2048 x 2048 f32 matrix per stage (row = 8 KB = 512 vectors); tile t = blockIdx: 128 rows x 512 B; wave -> 32 rows; lane (l15, lq)
      // loads 16 B at row l15 (+16), vector 4 * step + lq (+ 4 .. : the second half of the 128-byte line)
      const int nblk = blockIdx.x >> 4, kz = blockIdx.x & 15, l15 = lane & 15, lq = lane >> 4;
      const size_t mat = (size_t)s * (2048 * 512);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const size_t row = nblk * 128 + wid * 32 + u * 16 + l15;
            const size_t vec = pattern == 1 ? (size_t)kz * 32 + st * 8 + h * 4 + lq        // the kernel's mapping: 64 B per row and instruction
                                            : (size_t)kz * 32 + (st * 2 + h) * 4 + lq;
            w[(u * 4 + st) * 2 + h] = W[(mat + row * 512 + vec) % wvec];
          }
    }
  };
  if (pre && wid < 4) load(0);
  for (int s = 0; s < stages; ++s) {
    if (mode >= 2 && wid < 4) {
      if (!pre) load(s);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc += w[i];
      if (pre && s + 1 < stages) load(s + 1);
    }
    if (mode != 4) {
      for (int r = 0; r < bars_per_stage; ++r) {
        ++k;
        if (wid == 4 && lane == 0) {
          const bool ok = two ? bar_two<F, SLEEP>(b, k, nwg) : bar_flat<F, SLEEP>(b, k * nwg);
          if (!ok) dead = 1;
        }
        __syncthreads();
        if (dead) return;
      }
    }
  }
  if (wid < 4) out[blockIdx.x * 256 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
  Bar* bar; OK(hipMalloc(&bar, sizeof(Bar)));
  const size_t wbytes = 2ull << 30;
  f32x4* W; OK(hipMalloc(&W, wbytes)); OK(hipMemset(W, 0, wbytes));
  float* out; OK(hipMalloc(&out, 512 * 256 * 4));
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  const int stages = 200;
  hipDeviceProp_t prop; OK(hipGetDeviceProperties(&prop, 0));
  const int nwg = prop.multiProcessorCount;
  printf("CUs %d\n", nwg);
  for (int pattern = 0; pattern < 2; ++pattern)
  for (int flav = (pattern ? 4 : 0); flav < 5; ++flav) {
    const char* names[5] = {"system fences, sleep", "agent fences, sleep", "no fences, sleep", "agent fences, spin", "no fences, spin"};
    for (int mode = 0; mode <= 6; ++mode) {
      for (int bps : {1, 2}) {
        if (mode == 4 && (bps == 2 || (flav > 0 && !pattern))) continue;
        if (pattern && mode < 2) continue;
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
          OK(hipMemset(bar, 0, sizeof(Bar)));
          OK(hipEventRecord(e0));
          switch (flav) {
            case 0: hipLaunchKernelGGL((walk<0, 1>), dim3(nwg), dim3(320), 0, 0, bar, W, wbytes / 16, out, stages, mode, bps, pattern); break;
            case 1: hipLaunchKernelGGL((walk<1, 1>), dim3(nwg), dim3(320), 0, 0, bar, W, wbytes / 16, out, stages, mode, bps, pattern); break;
            case 2: hipLaunchKernelGGL((walk<2, 1>), dim3(nwg), dim3(320), 0, 0, bar, W, wbytes / 16, out, stages, mode, bps, pattern); break;
            case 3: hipLaunchKernelGGL((walk<1, 0>), dim3(nwg), dim3(320), 0, 0, bar, W, wbytes / 16, out, stages, mode, bps, pattern); break;
            default: hipLaunchKernelGGL((walk<2, 0>), dim3(nwg), dim3(320), 0, 0, bar, W, wbytes / 16, out, stages, mode, bps, pattern); break;
          }
          OK(hipEventRecord(e1));
          OK(hipEventSynchronize(e1));
          float ms; OK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
        }
        Bar h; OK(hipMemcpy(&h, bar, sizeof(Bar), hipMemcpyDeviceToHost));
        const double us_stage = best * 1e3 / stages;
        printf("pattern %d [%-20s] mode %d bars/stage %d: %8.3f ms  %6.2f us/stage  abort %u", pattern, names[flav], mode, bps, best, us_stage, h.abort_);
        if (mode >= 2) printf("  stream %7.1f GB/s", (double)nwg * 65536 * stages / (best * 1e-3) / 1e9);
        else printf("  %6.2f us/barrier", us_stage / bps);
        printf("\n");
      }
    }
  }
  return 0;
}
