// Probe: what does it cost to hand 16 KB from one workgroup to another on a different XCD inside one launch, per cache policy?
// Workgroup A writes a 16-KB block (256 lanes x 4 x 16 B), drains its stores, raises a flag; workgroup B (blockIdx chosen so that it sits
// on another XCD) polls the flag, reads the block, checks it, and answers the same way.  Reported: one-way time = round trip / 2.
//   policy 0: sc1 stores / sc1 loads (agent-coherent accesses, no cache maintenance)            <- what xf_walk.hip uses
//   policy 1: sc0 sc1 stores / loads (system scope)
//   policy 2: plain stores + buffer_wbl2 sc1 on the writer; buffer_inv sc1 + plain loads on the reader (release / acquire fences)
//   policy 3: sc1 stores (write-through) on the writer; buffer_inv sc1 + plain loads on the reader
//   policy 4: nt stores / nt loads
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/probe/probe_pingpong.hip -o /tmp/probe_pingpong && /tmp/probe_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int AUX> __device__ __forceinline__ f32x4 ld(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX));
}
template <int AUX> __device__ __forceinline__ void st(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, AUX);
}

template <int POL>
__global__ void __launch_bounds__(256) pingpong(float* buf, unsigned* flags, int iters, int peer, unsigned* err, int loads_only_one) {
  const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == (unsigned)peer ? 1 : -1);
  if (me < 0) return;
  constexpr int LAUX = POL == 0 ? 16 : (POL == 1 ? 17 : (POL == 4 ? 2 : 0));
  constexpr int SAUX = POL == 0 ? 16 : (POL == 1 ? 17 : (POL == 3 ? 16 : (POL == 4 ? 2 : 0)));
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, 1 << 20, 0x00020000);
  const int tid = threadIdx.x;
  __shared__ int dead;
  if (tid == 0) dead = 0;
  __syncthreads();
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    const int writer = it & 1;               // A writes on even iterations, B on odd ones
    if (me == writer) {
      f32x4 v = {(float)it, (float)tid, 1.f, 2.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) st<SAUX>(r, (unsigned)((i * 256 + tid) * 16), v);
      if (POL == 2) asm volatile("buffer_wbl2 sc1" ::: "memory");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flags, (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (tid == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(it + 1)) {
          if (++spins > (1u << 21)) { dead = 1; break; }
        }
      }
      __syncthreads();
      if (dead) { if (tid == 0) err[1] = 1; return; }
      if (POL == 2 || POL == 3) asm volatile("buffer_inv sc1" ::: "memory");
      f32x4 a[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = ld<LAUX>(r, (unsigned)(((loads_only_one ? 0 : i) * 256 + tid) * 16));
#pragma unroll
      for (int i = 0; i < 4; ++i) bad += (a[i][0] != (float)it) || (a[i][1] != (float)tid);
    }
  }
  if (bad) atomicAdd(err, bad);
}

template <int POL>
void run(const char* name, float* buf, unsigned* flags, unsigned* err, int peer) {
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  const int iters = 2000;
  float best = 1e9f;
  unsigned herr[2] = {0, 0};
  for (int rep = 0; rep < 3; ++rep) {
    OK(hipMemset(flags, 0, 64)); OK(hipMemset(err, 0, 8)); OK(hipMemset(buf, 0, 1 << 20));
    OK(hipEventRecord(e0));
    hipLaunchKernelGGL((pingpong<POL>), dim3(16), dim3(256), 0, 0, buf, flags, iters, peer, err, 0);
    OK(hipEventRecord(e1)); OK(hipEventSynchronize(e1));
    float ms; OK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
    OK(hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost));
  }
  printf("%-62s peer wg %2d: %6.2f us one way   (stale reads %u, gave up %u)\n", name, peer, best * 1e3 / iters, herr[0], herr[1]);
}

int main() {
  float* buf; unsigned *flags, *err;
  OK(hipMalloc(&buf, 1 << 20)); OK(hipMalloc(&flags, 64)); OK(hipMalloc(&err, 8));
  for (int peer : {1, 8}) {                 // wg 1: another XCD (round-robin dispatch); wg 8: the same XCD as wg 0
    run<0>("sc1 stores, sc1 loads", buf, flags, err, peer);
    run<1>("sc0 sc1 stores, sc0 sc1 loads", buf, flags, err, peer);
    run<2>("plain stores + buffer_wbl2 sc1 | buffer_inv sc1 + plain loads", buf, flags, err, peer);
    run<3>("sc1 stores | buffer_inv sc1 + plain loads", buf, flags, err, peer);
    run<4>("nt stores, nt loads", buf, flags, err, peer);
  }
  return 0;
}
