// Layer-walking kernel of the latent Transformer (reference: models/transformer.py:47-68 over torch.nn.Transformer — post-norm,
// ReLU, final encoder / decoder LayerNorm, sequence-first).  f32 end to end, v_mfma_f32_16x16x4_f32 (an exact f32 fma chain).
//
// Why one launch.  At 48 rows (8 clips x 6 tokens) the forward is a 1.75-GB weight stream that HBM delivers in 0.28 ms; as ~200
// launches (GEMM, split-K finish, attention, add + LayerNorm) it took 1.95 ms: every launch pays its own fill, first-byte latency and
// drain.  Here ONE workgroup per compute unit stays resident and walks a stage table (xf_walk.h); stages that depend on each other are
// separated by a device-wide barrier (2.2 us, tools/probe/probe_gridbar.hip) instead of a kernel boundary (fill + drain + cache
// maintenance), and the weights of the NEXT GEMM tile of a workgroup are already in flight while it computes the current one —
// across barriers and across layers, since weights depend on nothing.
//
// Data movement rules (what makes the barrier cheap).  The L2 of an XCD is not coherent with the seven others; a release / acquire
// pair at agent scope costs a write-back plus an invalidate of the L2 (3.3 us measured).  Instead every buffer that one workgroup writes
// and another reads INSIDE the launch (split-K slabs, the residual stream, attention output, FF hidden) is only ever touched with
// sc1 (agent-coherent) loads / stores — the instructions the memory model assigns to relaxed agent-scope atomics — so the barrier needs
// ordering only: s_waitcnt vmcnt(0) in every wave, then one arrive + poll.  Weights, biases, tables and the launch's inputs were
// written before the launch and use ordinary cached loads.
//
// GEMM stage.  The N x K weight matrix is cut into 128-column x 128-k tiles (64 KB), tile t -> workgroup t mod #workgroups: a
// 2048 x 2048 matrix is exactly one tile per compute unit.  Wave w owns 32 columns (two MFMA A operands of 16 weight rows), lane
// (l15, lq) loads W[n][k0 + 4 lq .. + 3] and W[n][k0 + 16 + 4 lq .. + 3] per 32-wide step: 16 x 16 B per lane = the whole tile in
// registers, requested one tile ahead.  The X slice (rows x 128 k, <= 88 KB) goes through LDS by LDS-direct loads (swizzled 128-byte
// rows, as xformer.hip's column-block form) and is shared by the four waves.  Each tile is one K slice of the product: partial sums go
// to slab kz, and the NEXT stage (reduce, reduce + LayerNorm, attention, embedding epilogue) adds the slabs in ascending order —
// deterministic — while applying bias / residual / activation.  So a GEMM never needs a "finish" stage of its own.
#include "kernels.h"
#include "xf_walk.h"
#include <mutex>
#include <type_traits>
#include <vector>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int SC1 = 16;                       // cache-policy bit of the buffer builtins: sc1 = agent-coherent
constexpr unsigned INVALID = 0x80000000u;     // buffer offset beyond num_records: the load returns 0
constexpr int WK_THREADS = 256;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct WalkSync { unsigned cnt; unsigned abort_; unsigned pad0[14]; unsigned grp[8][16]; unsigned rel[8][16]; };

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, unsigned bytes = 0x7FFFFFFFu) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 ldc(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, SC1));
}
__device__ __forceinline__ void stc(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, SC1);
}
// Read-only operands (weights, biases, tables, masks): ordinary cached loads, but through a buffer descriptor of the operand's exact size —
// an index that is off by a tile reads zeros instead of faulting the device (a faulting kernel can take every GPU of the host down)
__device__ __forceinline__ f32x4 ldr4(const void* p, unsigned n_elems, unsigned idx) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_of(p, n_elems * 4u), idx * 4u, 0, 0));
}
__device__ __forceinline__ float ldr1(const void* p, unsigned n_elems, unsigned idx) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_of(p, n_elems * 4u), idx * 4u, 0, 0));
}
// sum of ks slabs (stride zs bytes) at byte offset off, ascending z.  All loads of a batch of 16 are requested before the first add: a
// coherent load is a ~0.8 us round trip (stamps), so dependent rounds are what a reducing stage costs.  Slabs past ks are requested at
// an out-of-range offset: they return +0 without touching memory, and x + 0 leaves x unchanged.
__device__ __forceinline__ f32x4 slab_sum(__amdgpu_buffer_rsrc_t r, unsigned off, int ks, unsigned zs) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int z0 = 0; z0 < ks; z0 += 16) {
    f32x4 v[16];
#pragma unroll
    for (int z = 0; z < 16; ++z) v[z] = ldc(r, (z0 + z < ks && off != INVALID) ? off + (unsigned)(z0 + z) * zs : INVALID);
#pragma unroll
    for (int z = 0; z < 16; ++z) acc += v[z];
  }
  return acc;
}

// two vectors at once: 32 loads in flight (one round trip instead of two)
__device__ __forceinline__ void slab_sum2(__amdgpu_buffer_rsrc_t r, unsigned off0, unsigned off1, int ks, unsigned zs, f32x4& a0, f32x4& a1) {
  a0 = f32x4{0.f, 0.f, 0.f, 0.f};
  a1 = a0;
  for (int z0 = 0; z0 < ks; z0 += 16) {
    f32x4 v[16], w[16];
#pragma unroll
    for (int z = 0; z < 16; ++z) {
      const bool in = z0 + z < ks;
      v[z] = ldc(r, in && off0 != INVALID ? off0 + (unsigned)(z0 + z) * zs : INVALID);
      w[z] = ldc(r, in && off1 != INVALID ? off1 + (unsigned)(z0 + z) * zs : INVALID);
    }
#pragma unroll
    for (int z = 0; z < 16; ++z) { a0 += v[z]; a1 += w[z]; }
  }
}

// Device-wide barrier number k (1-based), called by ONE lane per workgroup after the workgroup's stores have completed.  Two levels:
// the workgroups of group g = blockIdx & 7 (one XCD under round-robin dispatch) count on grp[g]; the last of a group counts on cnt; the
// last overall releases every group.  Gives up (and tells everybody) after `timeout` ticks of the 100 MHz steady counter (2 s by default,
// wall clock, not iterations): a workgroup that never became resident must not hang the device.
__device__ __forceinline__ bool wk_barrier(WalkSync* sy, unsigned k, unsigned nwg, unsigned long long timeout) {
  const unsigned g = blockIdx.x & 7, per = nwg >> 3;
  const unsigned a = __hip_atomic_fetch_add(&sy->grp[g][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (a == k * per - 1) {
    const unsigned c = __hip_atomic_fetch_add(&sy->cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (c == k * 8 - 1)
      for (int i = 0; i < 8; ++i) __hip_atomic_store(&sy->rel[i][0], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  unsigned spins = 0;
  unsigned long long t0 = 0;
  while (__hip_atomic_load(&sy->rel[g][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 255) == 0) {                       // every 256 polls (~30 us): the clock and the give-up word of the others
      const unsigned long long now = __builtin_readsteadycounter();
      if (t0 == 0) t0 = now;
      if (now - t0 > timeout) { __hip_atomic_store(&sy->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
      if (__hip_atomic_load(&sy->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
    }
  }
  return true;
}

// In-kernel stamps (diagnostic build -DWK_STAMP, tools/build_variant.sh): workgroup 0, per stage, s_memtime (it counts shader clocks here, ≈ 2.2 GHz) at stage entry, after
// this workgroup's memory operations have drained, after the device-wide barrier, and at the end of the stage's work
#ifdef WK_STAMP
#define WSTAMP(slot) do { if (stamps && blockIdx.x == 0 && tid == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamps[s * 4 + (slot)] = t_; } } while (0)
#else
#define WSTAMP(slot) do { } while (0)
#endif

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- attention stage (shared by both walking kernels): q (Tq*B rows, stride q_ld) and k, v (Tk*B rows, stride kv_ld) are projections
// already reduced (+ bias) by the stage before; one (batch row, head) job per workgroup at a time
template <class Op>
__device__ __forceinline__ void wk_attention_stage(const Op& op, char* smem, unsigned nwg) {
  const int tid = threadIdx.x, lane = tid & 63;

  // q (Tq*B rows, stride q_ld) and k, v (Tk*B rows, stride kv_ld): projections already reduced (+ bias) by the stage before
  const int Tq = op.Tq, Tk = op.Tk, B = op.B, heads = op.heads, hd = op.hd;
  const int d = heads * hd, hv = hd / 4;
  float* sq = (float*)smem;
  float* sk = sq + Tq * hd;
  float* sv = sk + Tk * hd;
  float* sc = sv + Tk * hd;                              // [32][33]
  float* smk = sc + 32 * 33;                             // [32][33] the (Tq, Tk) mask
  float* skp = smk + 32 * 33;                            // [32] this batch row's key-padding bias
  const float scale = rsqrtf((float)hd);
  const __amdgpu_buffer_rsrc_t rQ = rsrc_of(op.qs, (unsigned)op.q_span * 4u), rK = rsrc_of(op.ks, (unsigned)op.kv_span * 4u),
                               rV = rsrc_of(op.vs, (unsigned)(op.kv_span - (int)(op.vs - op.ks)) * 4u), rO = rsrc_of(op.Y, (unsigned)(Tq * B * d) * 4u);
  for (int job = (int)blockIdx.x; job < B * heads; job += (int)nwg) {
    const int b = job / heads, hh = job - b * heads;
    __syncthreads();                                     // the previous job's LDS reads are done
    for (int idx = tid; idx < Tq * Tk; idx += WK_THREADS) smk[(idx / Tk) * 33 + idx % Tk] = op.mask ? ldr1(op.mask, Tq * Tk, idx) : 0.f;
    if (tid < Tk) skp[tid] = op.kpad ? ldr1(op.kpad, B * Tk, b * Tk + tid) : 0.f;
    for (int idx0 = 0; idx0 < (Tq + 2 * Tk) * hv; idx0 += 8 * WK_THREADS) {
      f32x4 a[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {                      // up to 8 rounds' worth of loads in flight at once
        const int idx = idx0 + i * WK_THREADS + tid;
        const int r = idx / hv, c = (idx - r * hv) * 4;
        unsigned off = INVALID;
        if (r < Tq) off = (unsigned)(((r * B + b) * op.q_ld + hh * hd + c) * 4);
        else if (r < Tq + 2 * Tk) off = (unsigned)((((r < Tq + Tk ? r - Tq : r - Tq - Tk) * B + b) * op.kv_ld + hh * hd + c) * 4);
        a[i] = r < Tq ? ldc(rQ, off) : (r < Tq + Tk ? ldc(rK, off) : ldc(rV, off));
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = idx0 + i * WK_THREADS + tid;
        if (idx < (Tq + 2 * Tk) * hv) *(f32x4*)(sq + idx * 4) = a[i];          // sq, sk, sv are contiguous: row r at r * hd
      }
    }
    __syncthreads();
    // scores: a 16-lane row per (i, j) pair, lane l takes channels 4 l + 64 k; the row sum by four DPP steps (a 64-lane butterfly of
    // ds_bpermute shuffles per pair cost ~0.4 us each: stamps)
    {
      const int rowi = lane >> 4, l = lane & 15;
      for (int p0 = (tid >> 6) * 4; p0 < Tq * Tk; p0 += 16) {
        const int p = p0 + rowi;
        const bool ok = p < Tq * Tk;
        const int pp = ok ? p : 0;
        const int i = pp / Tk, j = pp - i * Tk;
        f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
        for (int c = 4 * l; c < hd; c += 64) a4 += *(const f32x4*)(sq + i * hd + c) * *(const f32x4*)(sk + j * hd + c);
        float sacc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
        auto dpp_add = [&](auto ctrl) {
          sacc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sacc), decltype(ctrl)::value, 0xf, 0xf, true));
        };
        dpp_add(std::integral_constant<int, 0xB1>{});     // quad_perm [1,0,3,2]
        dpp_add(std::integral_constant<int, 0x4E>{});     // quad_perm [2,3,0,1]
        dpp_add(std::integral_constant<int, 0x141>{});    // row_half_mirror
        dpp_add(std::integral_constant<int, 0x140>{});    // row_mirror
        if (ok && l == 0) sc[i * 33 + j] = sacc * scale + smk[i * 33 + j] + skp[j];
      }
    }
    __syncthreads();
    if (tid < Tq) {
      float mx = -INFINITY;
      for (int j = 0; j < Tk; ++j) mx = fmaxf(mx, sc[tid * 33 + j]);
      float sum = 0.f;
      for (int j = 0; j < Tk; ++j) { const float e = expf(sc[tid * 33 + j] - mx); sc[tid * 33 + j] = e; sum += e; }
      const float inv = 1.f / sum;
      for (int j = 0; j < Tk; ++j) sc[tid * 33 + j] *= inv;
    }
    __syncthreads();
    for (int idx = tid; idx < Tq * hv; idx += WK_THREADS) {
      const int i = idx / hv, c = (idx - i * hv) * 4;
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      for (int j = 0; j < Tk; ++j) a += sc[i * 33 + j] * *(const f32x4*)(sv + j * hd + c);
      stc(rO, (unsigned)(((i * B + b) * d + hh * hd + c) * 4), a);
    }
  }
}

template <int MT>
__global__ void __launch_bounds__(WK_THREADS) xf_walk_kernel(const WalkOp* __restrict__ ops_g, int n_ops, WalkSync* sy, unsigned* host_abort, unsigned long long* stamps,
                                                              unsigned expect_wg, unsigned long long timeout) {
  // the stage table is read-only for the whole launch: through the constant address space every (uniform) field read is a scalar load,
  // which neither waits on nor disturbs the vector-memory counter the pipeline below counts on
  typedef const __attribute__((address_space(4))) WalkOp* cops_t;
  const cops_t ops = (cops_t)ops_g;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int dead;
  __shared__ float red[8];
  constexpr int STAGE = MT * 16 * 128;               // one 32-wide k step of the X tile
  constexpr int NPIECE = MT * 2;                     // 8-row pieces of a step
  constexpr int NPW = (NPIECE + 3) / 4;              // LDS-direct loads per wave and step (padded with sink writes)
  constexpr int SINK = 4 * STAGE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const unsigned nwg = gridDim.x;
  // workgroups the barrier counts: gridDim.x — except under the give-up test hook ($SVG_XF_WALK_TEST_GIVEUP), which asks for 8 more than exist
  const unsigned bar_wg = expect_wg;
  unsigned bar_k = 0;
  if (tid == 0) dead = 0;
  __syncthreads();

  // ---- the weight stream: tile (js, jt) is the next GEMM tile of this workgroup; its weights are requested into wn one tile ahead
  f32x4 wn[2][4][2], wc[2][4][2];
  auto tiles_of = [&](int s) { return (ops[s].N >> 7) * (ops[s].K >> 7); };
  auto load_w = [&](int s, int t) {
    const int K = ops[s].K, ksplit = K >> 7;
    const int nblk = t / ksplit, kz = t - nblk * ksplit;
    const __amdgpu_buffer_rsrc_t rW = rsrc_of(ops[s].W, (unsigned)(ops[s].N * K) * 4u);
    const unsigned voff = (unsigned)(((nblk * 128 + wave * 32 + l15) * K + kz * 128 + 4 * lq) * 4);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        wn[u][st][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, voff + st * 128, u * 64 * K, 0));
        wn[u][st][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, voff + st * 128 + 64, u * 64 * K, 0));
      }
  };
  int js = ops[0].kind == WK_GEMM ? 0 : ops[0].next_gemm, jt = (int)blockIdx.x;
  auto settle = [&]() {                               // move (js, jt) forward to a valid GEMM tile (or js = n_ops)
    while (js < n_ops && jt >= tiles_of(js)) { js = ops[js].next_gemm; jt = (int)blockIdx.x; }
  };
  settle();
  if (js < n_ops) load_w(js, jt);

  for (int s = 0; s < n_ops; ++s) {
    const auto& op = ops[s];
    WSTAMP(0);
    if (op.bar) {
      wait_vm<0>();                                    // this wave's stores (and loads) are complete
      __builtin_amdgcn_s_barrier();
      WSTAMP(1);
      ++bar_k;
      if (tid == 0 && !wk_barrier(sy, bar_k, bar_wg, timeout)) {
        dead = 1;
        __hip_atomic_store(host_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // host-mapped word: read without a copy
      }
      __syncthreads();
      if (dead) {
        // gave up: the host learns it at its next launch (host_abort); whoever reads THIS launch's result first must not mistake it for
        // one — the final stage's output becomes NaN
        const auto& last = ops[n_ops - 1];
        const __amdgpu_buffer_rsrc_t rL = rsrc_of(last.Y, (unsigned)(last.M * last.N) * 4u);
        const float qn = __builtin_bit_cast(float, 0x7FC00000u);   // the bit pattern (this file is built with -fno-honor-nans: no NaN literal)
        for (int v = (int)blockIdx.x * WK_THREADS + tid; v < last.M * last.N / 4; v += (int)nwg * WK_THREADS) stc(rL, (unsigned)v * 16u, f32x4{qn, qn, qn, qn});
        return;
      }
    }
    WSTAMP(2);
    const int kind = op.kind;
    if (kind == WK_GEMM) {
      const int M = op.M, N = op.N, K = op.K, ld = op.ld;
      const int ksplit = K >> 7, ntiles = (N >> 7) * ksplit;
      const __amdgpu_buffer_rsrc_t rX = rsrc_of(op.X, (unsigned)((int64_t)(M - 1) * ld + K) * 4u);
      const __amdgpu_buffer_rsrc_t rS = rsrc_of(op.slab, (unsigned)(ksplit * M * N) * 4u);
      int staged_kz = -1;
      for (int t = (int)blockIdx.x; t < ntiles; t += (int)nwg) {
        const int nblk = t / ksplit, kz = t - nblk * ksplit;
        const bool restage = kz != staged_kz;
        // wc <- wn through opaque moves, HERE: as plain assignments the compiler renames registers and puts the physical copies (and the wait
        // for the weights just requested) into the middle of this tile's MFMAs — the request then lives for one tile's compute instead of
        // staying in flight until the next tile (stamps: 6 us per GEMM stage at 6 rows instead of ~2)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
              for (int j = 0; j < 4; ++j) asm volatile("v_mov_b32 %0, %1" : "=v"(wc[u][st][hh][j]) : "v"(wn[u][st][hh][j]) : "memory");
        if (restage) {
          __builtin_amdgcn_s_barrier();                // every wave is done with the X tile in LDS
#pragma unroll
          for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int i = 0; i < NPW; ++i) {
              const int piece = i * 4 + wave;
              const int row = piece * 8 + (lane >> 3);
              const int ch = (lane & 7) ^ (row & 7);
              const bool ok = piece < NPIECE && row < M;
              const unsigned voff = ok ? (unsigned)((row * ld + kz * 128 + st * 32 + ch * 4) * 4) : INVALID;
              char* dst = piece < NPIECE ? smem + st * STAGE + piece * 1024 : smem + SINK + wave * 1024;
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_ptr_t)dst, 16, voff, 0, 0, SC1);
            }
          staged_kz = kz;
        }
        asm volatile("" ::: "memory");
        // the next tile's weights (the current one again at the end of the table: the counted waits below count on 16 loads)
        jt += (int)nwg;
        settle();
        auto prefetch = [&]() {
          if (js < n_ops) load_w(js, jt); else load_w(s, t);
          asm volatile("" ::: "memory");
        };
        prefetch();      // (holding the request back until X has landed was tried: neutral within 2 %, profiles/README.md)

        f32x4 acc[2][MT];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[u][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int xoff0 = l15 * 128 + ((lq ^ (l15 & 7)) << 4), xoff1 = l15 * 128 + (((lq + 4) ^ (l15 & 7)) << 4);
        auto step = [&]<int ST>() {
          if (restage) {
            wait_vm<(3 - ST) * NPW + 16>();            // this wave's pieces of step ST have landed (younger: later steps + 16 W loads)
            __builtin_amdgcn_s_barrier();              // ... and so have the other waves'
          }
          const char* xb = smem + ST * STAGE;
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const f32x4 x0 = *(const f32x4*)(xb + m * 2048 + xoff0), x1 = *(const f32x4*)(xb + m * 2048 + xoff1);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[u][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[u][ST][0][j], x0[j], acc[u][m], 0, 0, 0);
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[u][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[u][ST][1][j], x1[j], acc[u][m], 0, 0, 0);
            }
          }
        };
        step.template operator()<0>();
        step.template operator()<1>();
        step.template operator()<2>();
        step.template operator()<3>();
        // D: column j = lane & 15 -> row m of X, rows i = 4 lq + r -> weight row n.  Partial sums of K slice kz -> slab kz.
        const unsigned zoff = (unsigned)kz * (unsigned)(M * N * 4);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int row = m * 16 + l15;
          if (row < M) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
              stc(rS, zoff + (unsigned)((row * N + nblk * 128 + wave * 32 + u * 16 + 4 * lq) * 4), acc[u][m]);
          }
        }
      }
    } else if (kind == WK_RED) {
      const int N = op.N, ks = op.ksplit;
      const int total4 = op.M * N / 4;
      const unsigned zs = (unsigned)(op.M * N * 4);
      const unsigned mn4 = (unsigned)(op.M * N) * 4u;
      const __amdgpu_buffer_rsrc_t rS = rsrc_of(op.slab, mn4 * (unsigned)ks), rR = rsrc_of(op.res, mn4), rY = rsrc_of(op.Y, mn4);
      for (int v = (int)blockIdx.x * WK_THREADS + tid; v < total4; v += (int)nwg * WK_THREADS) {
        const unsigned off = (unsigned)v * 16u;
        f32x4 a = slab_sum(rS, off, ks, zs);
        if (op.bias) a += ldr4(op.bias, N, (v * 4) % N);
        if (op.res) a += ldc(rR, off);
        if (op.relu) { a[0] = fmaxf(a[0], 0.f); a[1] = fmaxf(a[1], 0.f); a[2] = fmaxf(a[2], 0.f); a[3] = fmaxf(a[3], 0.f); }
        stc(rY, off, a);
      }
    } else if (kind == WK_LN) {
      // a row per workgroup: v = sum of slabs + bias + residual; y = LN(v) g1 + b1; optionally y2 = LN(y) g2 + b2 (the final norm)
      const int N = op.N, ks = op.ksplit;
      const int nv = N / 4;                                  // N <= 3072: at most 3 vectors per thread
      const unsigned zs = (unsigned)(op.M * N * 4);
      const unsigned mn4 = (unsigned)(op.M * N) * 4u;
      const __amdgpu_buffer_rsrc_t rS = rsrc_of(op.slab, mn4 * (unsigned)ks), rR = rsrc_of(op.res, mn4), rY = rsrc_of(op.Y, mn4), rY2 = rsrc_of(op.Y2, mn4);
      for (int row = (int)blockIdx.x; row < op.M; row += (int)nwg) {
        // every load of the row is requested before the first use: slabs of vectors 0 and 1 as one batch of 32, bias / residual / gamma /
        // beta beside them (an out-of-range index returns 0: inactive vectors cost nothing)
        f32x4 v[3], bia[3], rs[3], gg[3], bb[3];
        unsigned off[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int c4 = tid + i * WK_THREADS;
          const bool in = c4 < nv;
          off[i] = in ? (unsigned)((row * N + c4 * 4) * 4) : INVALID;
          const unsigned ci = in ? (unsigned)(c4 * 4) : 0x3FFFFFF0u;
          bia[i] = op.bias ? ldr4(op.bias, N, ci) : f32x4{0.f, 0.f, 0.f, 0.f};
          rs[i] = op.res ? ldc(rR, off[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
          gg[i] = ldr4(op.g1, N, ci);
          bb[i] = ldr4(op.b1, N, ci);
        }
        v[0] = v[1] = v[2] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ks) {
          slab_sum2(rS, off[0], off[1], ks, zs, v[0], v[1]);
          if (nv > 2 * WK_THREADS) v[2] = slab_sum(rS, off[2], ks, zs);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) { v[i] += bia[i]; v[i] += rs[i]; }
        auto block_sum = [&](float x) {
          for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
          __syncthreads();
          if (lane == 0) red[wave] = x;
          __syncthreads();
          return red[0] + red[1] + red[2] + red[3];
        };
        auto norm = [&](const f32x4 (&g)[3], const f32x4 (&b)[3]) {
          float sm = 0.f;
#pragma unroll
          for (int i = 0; i < 3; ++i)
            if (tid + i * WK_THREADS < nv) sm += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
          const float mean = block_sum(sm) / (float)N;
          float q = 0.f;
#pragma unroll
          for (int i = 0; i < 3; ++i)
            if (tid + i * WK_THREADS < nv) {
              const f32x4 t = v[i] - mean;
              q += (t[0] * t[0] + t[1] * t[1]) + (t[2] * t[2] + t[3] * t[3]);
            }
          const float rstd = rsqrtf(block_sum(q) / (float)N + op.eps);
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const int c4 = tid + i * WK_THREADS;
            if (c4 < nv) v[i] = (v[i] - mean) * rstd * g[i] + b[i];
          }
        };
        auto store = [&](__amdgpu_buffer_rsrc_t r) {
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const int c4 = tid + i * WK_THREADS;
            if (c4 < nv) stc(r, (unsigned)((row * N + c4 * 4) * 4), v[i]);
          }
        };
        norm(gg, bb);
        if (op.Y) store(rY);
        if (op.g2) {
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const unsigned ci = tid + i * WK_THREADS < nv ? (unsigned)((tid + i * WK_THREADS) * 4) : 0x3FFFFFF0u;
            gg[i] = ldr4(op.g2, N, ci);
            bb[i] = ldr4(op.b2, N, ci);
          }
          norm(gg, bb);
          store(rY2);
        }
      }
    } else if (kind == WK_EMBED) {
      // slab rows are (b, t) batch-first and N = d - d_txt wide; Y rows are (t, b) sequence-first and d wide
      const int d_img = op.N, d = op.N + op.d_txt, T = op.T, B = op.B, ks = op.ksplit;
      const int total4 = B * T * d / 4;
      const unsigned zs = (unsigned)(op.M * d_img * 4);
      const __amdgpu_buffer_rsrc_t rS = rsrc_of(op.slab, zs * (unsigned)ks), rY = rsrc_of(op.Y, (unsigned)(B * T * d) * 4u);
      for (int v = (int)blockIdx.x * WK_THREADS + tid; v < total4; v += (int)nwg * WK_THREADS) {
        const int c = (v * 4) % d;
        const int bt = (v * 4) / d;
        const int t = bt % T, b = bt / T;
        f32x4 a;
        if (c < d_img) {
          a = slab_sum(rS, (unsigned)(((b * T + t) * d_img + c) * 4), ks, zs);
          a += ldr4(op.bias, d_img, c);
        } else {
          a = ldr4(op.text, B * op.d_txt, b * op.d_txt + (c - d_img));
        }
        const int pr = op.pe_row ? __builtin_bit_cast(int, ldr1(op.pe_row, B, b)) : b;
        a = a * op.scale + ldr4(op.pe, 64 * d, pr * d + c);
        stc(rY, (unsigned)(((t * B + b) * d + c) * 4), a);
      }
    } else if (kind == WK_ATTN) {
      wk_attention_stage(op, smem, nwg);
    }
    WSTAMP(3);
  }
}

// ---- the small-row kernel (at most 8 rows: single-clip sampling, `python -m prediction.predict`) --------------------------------------------
// At 6 rows the split-K walk above is a chain of 168 dependent stages of ~6 us, and what a stage costs is its device-wide barrier and one
// coherent round trip, not its bytes.  Here a GEMM stage is NOT split over K: workgroup w owns ceil(N / #workgroups) <= 8
// output columns and reads whole rows of W for them (64 KB at 2048 x 2048: the same bytes per workgroup as a 128 x 128 tile), so its result is
// complete — bias, ReLU, the residual add and the embedding's scale + positional row ride in the GEMM's epilogue and the `reduce` stages are
// gone; and it holds ALL (<= 8) rows of X in LDS, so the LayerNorm in front of a GEMM is computed by the consumer on its own copy of the rows
// (every workgroup redundantly: 8 x 2048 values) and the `reduce + LayerNorm` stages are gone too: 5 instead of 9 stages per encoder layer,
// 8 instead of 16 per decoder layer.
// Matrix tile.  v_mfma_f32_16x16x4_f32 with 8 valid weight rows and <= 8 valid X rows would run 19 % full; instead the two halves of a wave's K
// range share one tile: A row (c, h) = weight column c over k-half h, B column (m, h') = X row m over k-half h'; the diagonal blocks h = h' are
// the two halves' partial sums of out[m][c], the off-diagonal blocks are discarded.  Half the MFMAs per wave, all 64 lanes load weights.
// Four waves split K; eight partial sums per output meet in LDS in a fixed order.
constexpr int SM_STEPS = kWalkSmallMaxK / 128;          // 16-wide k steps of a wave's half range (K / 128)
constexpr int SM_PAD = 64;                              // bytes between X rows in LDS (rows would otherwise share all banks)
constexpr int SM_GB = kWalkSmallRows * (kWalkSmallMaxK * 4 + SM_PAD);     // LDS offset of the LayerNorm parameters (4 vectors of K floats)

__global__ void __launch_bounds__(WK_THREADS) xf_walk_small_kernel(const WalkOp* __restrict__ ops_g, int n_ops, WalkSync* sy, unsigned* host_abort,
                                                                    unsigned expect_wg, unsigned long long timeout) {
  typedef const __attribute__((address_space(4))) WalkOp* cops_t;
  const cops_t ops = (cops_t)ops_g;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int dead;
  __shared__ __attribute__((aligned(16))) float red[4][2][8][8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int c8 = l15 & 7, h = l15 >> 3;                 // operand row / column (c or m, k-half)
  const unsigned nwg = gridDim.x;
  unsigned bar_k = 0;
  if (tid == 0) dead = 0;
  __syncthreads();

  // ---- the weight stream: this workgroup's column block of the next GEMM stage, requested one stage ahead
  f32x4 wn[SM_STEPS], wc[SM_STEPS];
  auto load_w = [&](int s) {
    const int N = ops[s].N, K = ops[s].K;
    const int ct = (N + (int)nwg - 1) / (int)nwg;
    const int n = (int)blockIdx.x * ct + c8;
#ifndef SM_ABL            // timing ablations of the small-row kernel (results are garbage): 1 no weight loads, 2 no X staging
#define SM_ABL 0
#endif
    const bool okn = SM_ABL != 1 && c8 < ct && n < N;
    const __amdgpu_buffer_rsrc_t rW = rsrc_of(ops[s].W, (unsigned)(N * K) * 4u);
    const int kh = wave * (K >> 2) + h * (K >> 3) + 4 * lq;
    const int steps = K >> 7;
#pragma unroll
    for (int st = 0; st < SM_STEPS; ++st) {
      const unsigned off = (okn && st < steps) ? (unsigned)((n * K + kh + 16 * st) * 4) : INVALID;
      wn[st] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, off, 0, 0));
    }
  };
  int js = ops[0].kind == WK_GEMMF ? 0 : ops[0].next_gemm;
  if (js < n_ops) load_w(js);

  for (int s = 0; s < n_ops; ++s) {
    const auto& op = ops[s];
    if (op.bar) {
      wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      ++bar_k;
      if (tid == 0 && !wk_barrier(sy, bar_k, expect_wg, timeout)) {
        dead = 1;
        __hip_atomic_store(host_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      __syncthreads();
      if (dead) {                                        // gave up: the final stage's output becomes NaN (see xf_walk_kernel)
        // the final stage may be cut into column blocks (reuse_x marks the continuation blocks): poison ALL of its columns, not the last block's
        const auto& last = ops[n_ops - 1];
        int f = n_ops - 1;
        while (f > 0 && ops[f].reuse_x) --f;
        float* const Y0 = ops[f].Y;
        const int ntot = (int)(last.Y - Y0) + last.N;
        const __amdgpu_buffer_rsrc_t rL = rsrc_of(Y0, (unsigned)((last.M - 1) * last.ldy + ntot) * 4u);
        const unsigned qn = 0x7FC00000u;
        for (int v = (int)blockIdx.x * WK_THREADS + tid; v < last.M * ntot; v += (int)nwg * WK_THREADS)
          __builtin_amdgcn_raw_buffer_store_b32(qn, rL, (unsigned)(((v / ntot) * last.ldy + v % ntot) * 4), 0, SC1);
        return;
      }
    }
    if (op.kind == WK_ATTN) {
      wk_attention_stage(op, smem, nwg);
      continue;
    }
    // ---- WK_GEMMF
    const int M = op.M, N = op.N, K = op.K;
    const int rowb = K * 4 + SM_PAD;                      // bytes per X row in LDS
    const int steps = K >> 7;
    // wc <- wn through opaque moves (as in xf_walk_kernel: keeps the request of the NEXT block in flight across this stage)
#pragma unroll
    for (int st = 0; st < SM_STEPS; ++st)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("v_mov_b32 %0, %1" : "=v"(wc[st][j]) : "v"(wn[st][j]) : "memory");
    js = op.next_gemm;
    if (!op.reuse_x) {
      __builtin_amdgcn_s_barrier();                      // every wave is done with the previous X tile / attention slices in LDS
      const __amdgpu_buffer_rsrc_t rX = rsrc_of(op.X, (unsigned)((M - 1) * op.ld + K) * 4u);
      const int segs = K >> 8;                           // 1-KiB pieces per row
      const int total = M * segs;
      for (int p = wave; p < total; p += 4) {
        const int r = p / segs, seg = p - r * segs;
        const int src = op.perm ? (r % op.B) * op.T + r / op.B : r;      // the launch's inputs are batch-first, the rows in flight (t, b)
        const unsigned voff = SM_ABL == 2 ? INVALID : (unsigned)((src * op.ld + seg * 256 + lane * 4) * 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_ptr_t)(smem + r * rowb + seg * 1024), 16, voff, 0, 0, SC1);
      }
      // the LayerNorm parameters ride along (plain cached loads: they were written before the launch): gamma1 | beta1 | gamma2 | beta2 behind the
      // rows, so the normalising pass reads them from LDS instead of waiting on a global round trip per pass
      if (op.g1) {
        const int nvec = op.g2 ? 4 : 2;
        for (int p = wave; p < nvec * segs; p += 4) {
          const int v = p / segs, seg = p - v * segs;
          const float* src = v == 0 ? op.g1 : (v == 1 ? op.b1 : (v == 2 ? op.g2 : op.b2));
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_of(src, (unsigned)K * 4u), (lds_ptr_t)(smem + SM_GB + v * K * 4 + seg * 1024), 16,
                                                   (unsigned)((seg * 256 + lane * 4) * 4), 0, 0, 0);
        }
      }
    }
    asm volatile("" ::: "memory");
    if (js < n_ops) load_w(js); else load_w(s);           // 16 loads either way: the counted wait below counts on them
    asm volatile("" ::: "memory");
    if (!op.reuse_x) {
      wait_vm<SM_STEPS>();                               // the X pieces (older than the 16 weight loads just requested) have landed
      __builtin_amdgcn_s_barrier();
      if (op.g1) {
        // LayerNorm(s) of the rows, in place: 32 threads per row, two-pass statistics; then this workgroup's share of the normalised rows
        // goes to Yln (the residual of a later stage reads it)
        const int r = tid >> 5, j = tid & 31;
        const bool rok = r < M;
        float* xr = (float*)(smem + r * rowb);
        auto ln_pass = [&](int which) {
          const float* g = (const float*)(smem + SM_GB + (2 * which) * K * 4);
          const float* b = (const float*)(smem + SM_GB + (2 * which + 1) * K * 4);
          f32x4 v[SM_STEPS];
          float sum = 0.f;
#pragma unroll
          for (int i = 0; i < SM_STEPS; ++i) {
            v[i] = (rok && i < steps) ? *(const f32x4*)(xr + j * 4 + i * 128) : f32x4{0.f, 0.f, 0.f, 0.f};
            sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
          }
          for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
          const float mean = sum / (float)K;
          float q = 0.f;
#pragma unroll
          for (int i = 0; i < SM_STEPS; ++i)
            if (i < steps) { const f32x4 t = v[i] - mean; q += (t[0] * t[0] + t[1] * t[1]) + (t[2] * t[2] + t[3] * t[3]); }
          for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o);
          const float rstd = rsqrtf(q / (float)K + op.eps);
#pragma unroll
          for (int i = 0; i < SM_STEPS; ++i)
            if (rok && i < steps) {
              const int k = j * 4 + i * 128;
              *(f32x4*)(xr + k) = (v[i] - mean) * rstd * *(const f32x4*)(g + k) + *(const f32x4*)(b + k);
            }
        };
        ln_pass(0);
        if (op.g2) ln_pass(1);                           // (a thread re-reads only what it wrote itself)
        __syncthreads();
        if (op.Yln) {
          const int cl = (K + (int)nwg - 1) / (int)nwg, k0 = (int)blockIdx.x * cl;
          const __amdgpu_buffer_rsrc_t rL = rsrc_of(op.Yln, (unsigned)(M * K) * 4u);
          for (int t = tid; t < M * cl; t += WK_THREADS) {
            const int rr = t / cl, k = k0 + t - rr * cl;
            if (k < K) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, *(const float*)(smem + rr * rowb + k * 4)), rL, (unsigned)((rr * K + k) * 4), 0, SC1);
          }
        }
      }
    }
    // ---- the epilogue's operands (bias, positional row, residual) are requested BEFORE the product: three dependent global round trips
    // (~1 us each) otherwise sit on the stage's critical path
    const int ct = (N + (int)nwg - 1) / (int)nwg;
    const int e_m = tid >> 3, e_n = (int)blockIdx.x * ct + (tid & 7);
    const bool e_ok = tid < 64 && e_m < M && (tid & 7) < ct && e_n < N;
    float e_bias = 0.f, e_pe = 0.f, e_res = 0.f;
    if (e_ok) {
      if (op.bias) e_bias = ldr1(op.bias, N, e_n);
      if (op.pe) {
        const int b = e_m % op.B;
        const int pr = op.pe_row ? __builtin_bit_cast(int, ldr1(op.pe_row, op.B, b)) : b;
        e_pe = ldr1(op.pe, 63 * op.ld_res + N, pr * op.ld_res + e_n);
      }
      if (op.res) e_res = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_of(op.res, (unsigned)((M - 1) * op.ld_res + N) * 4u), (unsigned)((e_m * op.ld_res + e_n) * 4), 0, SC1));
    }
    // ---- the product: this wave's quarter of K, two halves in one tile
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    {
      const char* xb = smem + c8 * rowb + (wave * (K >> 2) + h * (K >> 3) + 4 * lq) * 4;
#pragma unroll
      for (int st = 0; st < SM_STEPS; ++st)
        if (st < steps) {
          const f32x4 x = *(const f32x4*)(xb + st * 64);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[st][j], x[j], acc, 0, 0, 0);
        }
    }
    // D[i = 4 lq + r][j = l15]: valid where the halves agree, (lq >> 1) == h: out[m = l15 & 7][c = 4 (lq & 1) + r], half h
    __syncthreads();                                     // the previous stage's readers of `red` are done
    if ((lq >> 1) == h) *(f32x4*)&red[wave][h][c8][4 * (lq & 1)] = acc;
    __syncthreads();
    if (e_ok) {
      const int m = e_m, cc = tid & 7;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += red[w][0][m][cc] + red[w][1][m][cc];
      v += e_bias;
      if (op.pe) v = v * op.scale + e_pe;                 // embedding epilogue: rows are (t, b); positional row of batch row b
      if (op.relu) v = fmaxf(v, 0.f);
      v += e_res;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_of(op.Y, (unsigned)((M - 1) * op.ldy + N) * 4u), (unsigned)((m * op.ldy + e_n) * 4), 0, SC1);
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------------
constexpr int kRing = 16;                              // table / barrier-word slots: a launch reuses the slot of the 16th launch before it
constexpr int kMts[] = {1, 2, 3, 4, 6, 8, 11};        // the instantiated accumulator heights
struct WalkDev {
  bool init = false;
  bool pending = false;              // a give-up has been seen (walk turned off, logged) but not yet raised to a Transformer caller
  bool disabled = false;             // set when a launch gave up at a barrier, or when ranks share the device: the per-GEMM kernels serve
  bool coop_ok = false;              // the device reports hipDeviceAttributeCooperativeLaunch
  bool coop = false;                 // launches go through hipLaunchCooperativeKernel (the runtime refuses a grid that cannot be co-resident)
  bool test_giveup = false;
  int n_wg = 0;                      // grid: min over the instantiations of (resident workgroups per compute unit at the launch's LDS) x compute units
  unsigned long long timeout = 0;    // barrier give-up time in 100 MHz ticks
  WalkSync* sync[kRing] = {};
  WalkOp* dops[kRing] = {};
  WalkOp* hops[kRing] = {};          // pinned
  unsigned* habort = nullptr;        // pinned, one word per slot: set by a launch that gave up at a barrier
  hipEvent_t done[kRing] = {};
  bool used[kRing] = {};
  int next = 0;
  int last = -1;                     // slot of the previous launch (its `done` event orders the next one)
};
std::mutex g_mu;
WalkDev g_dev[16];
constexpr int64_t kMaxLds = 150 * 1024;
constexpr int64_t kMinLds = 84 * 1024;                // above half a compute unit's LDS: exactly one workgroup per compute unit

// The knobs, looked up once per device at svg_create (and again at svg_env_refresh), never on the forward path:
//   SVG_XF_WALK          0 = per-GEMM kernels only.  Default 1 — except when ranks share the device (the one-GPU rehearsal of the N > 1 path sets
//                        SVG_DEVICE_OVERRIDE): their resident grids would starve each other of compute units
//   SVG_XF_WALK_COOP     1 = launch through hipLaunchCooperativeKernel.  Default 0: measured +23 ... +31 us per forward (1.049 -> 1.080 ms at 6
//                        rows, 1.416 -> 1.439 at 48; profiles/r05_walk_coop_ab.txt) for a check the occupancy query at init already makes — a
//                        plain launch of the same grid has the same residency (MI355X_MICROARCH.md, coop-launch row)
//   SVG_XF_WALK_TIMEOUT_MS  wall-clock give-up time of a barrier (default 2000)
//   SVG_XF_WALK_TEST_GIVEUP  test hook: barriers wait for 8 workgroups more than the grid has
void apply_env(WalkDev& D) {
  D.coop = D.coop_ok && svg_env_i64("SVG_XF_WALK_COOP", 0) != 0;
  D.timeout = (unsigned long long)std::max<int64_t>(1, svg_env_i64("SVG_XF_WALK_TIMEOUT_MS", 2000)) * 100000ull;
  D.test_giveup = svg_env_i64("SVG_XF_WALK_TEST_GIVEUP", 0) != 0;
  D.disabled = D.n_wg < 8 || svg_env_i64("SVG_XF_WALK", getenv("SVG_DEVICE_OVERRIDE") ? 0 : 1) == 0;
}

template <int MT>
int resident_per_cu() {
  HIP_OK(hipFuncSetAttribute((const void*)xf_walk_kernel<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
  int n = 0;
  HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)xf_walk_kernel<MT>, WK_THREADS, (size_t)kMaxLds));
  return n;
}
template <int MT>
void launch(const WalkDev& D, int64_t lds, hipStream_t s, const WalkOp* ops, int n_ops, WalkSync* sy, unsigned* habort, unsigned expect_wg) {
  unsigned long long* stamps = nullptr;
#ifdef WK_STAMP
  static unsigned long long* g_stamps = nullptr;
  if (!g_stamps) HIP_OK(hipMalloc((void**)&g_stamps, 4 * kWalkMaxOps * sizeof(unsigned long long)));
  HIP_OK(hipMemsetAsync(g_stamps, 0, 4 * kWalkMaxOps * sizeof(unsigned long long), s));
  stamps = g_stamps;
#endif
  unsigned long long timeout = D.timeout;
  if (D.coop) {
    void* args[] = {(void*)&ops, (void*)&n_ops, (void*)&sy, (void*)&habort, (void*)&stamps, (void*)&expect_wg, (void*)&timeout};
    HIP_OK(hipLaunchCooperativeKernel((const void*)xf_walk_kernel<MT>, dim3(D.n_wg), dim3(WK_THREADS), args, (unsigned)lds, s));
  } else {
    hipLaunchKernelGGL((xf_walk_kernel<MT>), dim3(D.n_wg), dim3(WK_THREADS), (size_t)lds, s, ops, n_ops, sy, habort, stamps, expect_wg, timeout);
  }
#ifdef WK_STAMP
  if (const char* path = getenv("SVG_XF_WALK_STAMPS")) {       // diagnostic build only: synchronous dump of the last launch
    HIP_OK(hipStreamSynchronize(s));
    std::vector<unsigned long long> h(4 * n_ops);
    HIP_OK(hipMemcpy(h.data(), g_stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<WalkOp> hop(n_ops);
    HIP_OK(hipMemcpy(hop.data(), ops, n_ops * sizeof(WalkOp), hipMemcpyDeviceToHost));
    if (FILE* f = fopen(path, "w")) {
      fprintf(f, "# stage kind bar M N K | entry drained after_barrier end (s_memtime ticks, relative to stage 0 entry)\n");
      for (int i = 0; i < n_ops; ++i)
        fprintf(f, "%d %d %d %d %d %d | %llu %llu %llu %llu\n", i, hop[i].kind, hop[i].bar, hop[i].M, hop[i].N, hop[i].K, h[4 * i] - h[0],
                h[4 * i + 1] ? h[4 * i + 1] - h[0] : 0ull, h[4 * i + 2] - h[0], h[4 * i + 3] - h[0]);
      fclose(f);
    }
  }
#endif
}
int walk_mt(int rows) {
  const int mt = (rows + 15) / 16;
  return mt <= 4 ? mt : (mt <= 6 ? 6 : (mt <= 8 ? 8 : 11));
}
const char* kGaveUp = "xf_walk: a layer-walking launch gave up at a device-wide barrier (a workgroup never became resident: something else is "
                      "holding compute units of this device).  The output of that forward is invalid (NaN-filled).  The layer-walking launch "
                      "is now OFF for this device in this process — later forwards run the per-GEMM kernels; re-issue the failed forward";

// Every stage against what the kernel assumes about it.  The kernel reads its operands through exact-size descriptors (a wrong index
// returns zeros instead of faulting), but a table that breaks these rules would still compute garbage or overrun LDS: refuse it on the host.
void validate_table(const WalkOp* ops, int n_ops, int rows, int64_t lds, bool small, int n_wg) {
  const int mt = walk_mt(rows);
  auto span_ok = [](int64_t elems) { return elems > 0 && elems * 4 < (int64_t)1 << 31; };
  for (int i = 0; i < n_ops; ++i) {
    const WalkOp& op = ops[i];
    switch (op.kind) {
      case WK_GEMM:
        SVG_CHECK(xf_walk_gemm_ok(op.N, op.K) && op.M >= 1 && op.M <= mt * 16 && op.ld >= op.K && op.ksplit == op.K / 128 && op.X && op.W && op.slab,
                  "xf_walk: stage %d: GEMM %d x %d x %d (ld %d) does not fit the kernel (tiles of 128 x 128, at most %d rows)", i, op.M, op.N, op.K, op.ld, mt * 16);
        SVG_CHECK(span_ok((int64_t)op.N * op.K) && span_ok((int64_t)(op.M - 1) * op.ld + op.K) && span_ok((int64_t)op.ksplit * op.M * op.N),
                  "xf_walk: stage %d: an operand of GEMM %d x %d x %d exceeds a 2 GB buffer descriptor", i, op.M, op.N, op.K);
        SVG_CHECK((int64_t)4 * mt * 16 * 128 + 4096 <= lds, "xf_walk: stage %d: X tile of %d rows needs more than %lld bytes of LDS", i, mt * 16, (long long)lds);
        break;
      case WK_RED:
        SVG_CHECK(op.M >= 1 && op.N % 4 == 0 && op.slab && op.Y && op.ksplit >= 1 && span_ok((int64_t)op.ksplit * op.M * op.N), "xf_walk: stage %d: reduce %d x %d x %d slabs", i, op.M, op.N, op.ksplit);
        break;
      case WK_LN:
        SVG_CHECK(op.M >= 1 && op.N % 4 == 0 && op.N <= 3072 && op.g1 && op.b1 && (op.Y || op.Y2) && (!op.g2 || (op.b2 && op.Y2)) && (op.ksplit == 0 || op.slab) &&
                      span_ok((int64_t)std::max(op.ksplit, 1) * op.M * op.N),
                  "xf_walk: stage %d: LayerNorm over %d x %d (at most 3072 columns)", i, op.M, op.N);
        break;
      case WK_ATTN:
        SVG_CHECK(op.Tq >= 1 && op.Tq <= 32 && op.Tk >= 1 && op.Tk <= 32 && op.hd % 4 == 0 && op.heads >= 1 && op.B >= 1 && op.qs && op.ks && op.vs && op.Y &&
                      op.vs >= op.ks && op.q_span > 0 && op.kv_span > (int)(op.vs - op.ks),
                  "xf_walk: stage %d: attention Tq %d Tk %d head dim %d", i, op.Tq, op.Tk, op.hd);
        SVG_CHECK(((int64_t)(op.Tq + 2 * op.Tk) * op.hd + 2 * 32 * 33 + 32) * 4 <= lds, "xf_walk: stage %d: attention slices exceed %lld bytes of LDS", i, (long long)lds);
        break;
      case WK_EMBED:
        SVG_CHECK(op.M == op.B * op.T && op.N % 4 == 0 && op.d_txt % 4 == 0 && op.slab && op.Y && op.pe && op.bias && op.ksplit >= 1 && (op.d_txt == 0 || op.text),
                  "xf_walk: stage %d: embedding %d x %d (+%d)", i, op.M, op.N, op.d_txt);
        break;
      case WK_GEMMF:
        SVG_CHECK(small && op.M >= 1 && op.M <= kWalkSmallRows && op.K % 256 == 0 && op.K >= 256 && op.K <= kWalkSmallMaxK && op.N >= 1 &&
                      (op.N + n_wg - 1) / n_wg <= 8 && op.ld >= op.K && op.ldy >= op.N && op.X && op.W && op.Y && (!op.res || op.ld_res >= op.N),
                  "xf_walk: stage %d: small-row GEMM %d x %d x %d does not fit the kernel (at most %d rows, K a multiple of 256 up to %d, at most 8 columns per workgroup)",
                  i, op.M, op.N, op.K, kWalkSmallRows, kWalkSmallMaxK);
        SVG_CHECK((!op.g1 || op.b1) && (!op.g2 || (op.g1 && op.b2)) && (!op.Yln || op.g1) && (!op.reuse_x || (i > 0 && ops[i - 1].kind == WK_GEMMF && ops[i - 1].X == op.X && ops[i - 1].K == op.K && !op.bar)) &&
                      (!op.perm || (op.B >= 1 && op.T >= 1 && op.B * op.T == op.M)) && (!op.pe || op.B >= 1),
                  "xf_walk: stage %d: small-row GEMM flags are inconsistent", i);
        SVG_CHECK(span_ok((int64_t)op.N * op.K) && (int64_t)kWalkSmallRows * (kWalkSmallMaxK * 4 + 64) + 4 * kWalkSmallMaxK * 4 <= lds, "xf_walk: stage %d: operands of the small-row GEMM exceed a descriptor / LDS", i);
        break;
      default: SVG_CHECK(false, "xf_walk: stage %d: unknown kind %d", i, op.kind);
    }
    SVG_CHECK(small ? (op.kind == WK_GEMMF || op.kind == WK_ATTN) : op.kind != WK_GEMMF, "xf_walk: stage %d: kind %d does not belong to this launch form", i, op.kind);
  }
  SVG_CHECK(!ops[0].bar, "xf_walk: the first stage reads the launch's inputs and takes no barrier");
}

}  // namespace

int64_t xf_walk_lds_bytes(int rows, int Tq, int Tk, int hd) {
  const int mt = walk_mt(rows);
  const int64_t gemm = (int64_t)4 * mt * 16 * 128 + 4096;
  const int64_t attn = ((int64_t)(Tq + 2 * Tk) * hd + 2 * 32 * 33 + 32) * 4;
  return std::max(gemm, attn);
}

bool xf_walk_available(int rows, int64_t lds_bytes) { return rows >= 1 && rows <= kWalkMaxRows && lds_bytes <= kMaxLds; }

void xf_walk_init_device() {
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_mu);
  SVG_CHECK(dev >= 0 && dev < 16, "xf_walk: device index %d", dev);
  WalkDev& D = g_dev[dev];
  if (D.init) return;
  hipDeviceProp_t prop;
  HIP_OK(hipGetDeviceProperties(&prop, dev));
  // Residency is a launch-time fact, not an assumption: the grid is what the occupancy calculator says is co-resident for EVERY instantiation at
  // the largest LDS a launch may ask for, capped at one workgroup per compute unit (the kernel's tiling: a 2048 x 2048 matrix = 256 tiles), a
  // multiple of 8 (the barrier's per-XCD groups).  With $SVG_XF_WALK_COOP=1 the launch itself goes through
  // hipLaunchCooperativeKernel, which repeats that check per launch (and costs 23-31 us per forward: off by default).
  int small_per_cu = 0;
  HIP_OK(hipFuncSetAttribute((const void*)xf_walk_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
  HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&small_per_cu, (const void*)xf_walk_small_kernel, WK_THREADS, (size_t)kMaxLds));
  const int per_cu = std::min({small_per_cu, resident_per_cu<1>(), resident_per_cu<2>(), resident_per_cu<3>(), resident_per_cu<4>(), resident_per_cu<6>(),
                               resident_per_cu<8>(), resident_per_cu<11>()});
  D.n_wg = per_cu >= 1 ? (prop.multiProcessorCount / 8) * 8 : 0;
  int coop_attr = 0;
  HIP_OK(hipDeviceGetAttribute(&coop_attr, hipDeviceAttributeCooperativeLaunch, dev));
  D.coop_ok = coop_attr != 0;
  apply_env(D);
  for (int i = 0; i < kRing; ++i) {
    HIP_OK(hipMalloc((void**)&D.sync[i], sizeof(WalkSync)));
    HIP_OK(hipMalloc((void**)&D.dops[i], sizeof(WalkOp) * kWalkMaxOps));
    HIP_OK(hipHostMalloc((void**)&D.hops[i], sizeof(WalkOp) * kWalkMaxOps, hipHostMallocDefault));
    HIP_OK(hipEventCreateWithFlags(&D.done[i], hipEventDisableTiming));
  }
  HIP_OK(hipHostMalloc((void**)&D.habort, sizeof(unsigned) * kRing, hipHostMallocDefault));
  for (int i = 0; i < kRing; ++i) D.habort[i] = 0;
  D.init = true;
}

void xf_walk_env_refresh() {
  std::lock_guard<std::mutex> lk(g_mu);
  for (WalkDev& D : g_dev)
    if (D.init) apply_env(D);
}

bool xf_walk_enabled(hipStream_t s) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    const WalkDev& D = g_dev[dev];
    if (!D.init || D.disabled) return false;
  }
  // a launch stages its table through a pinned ring slot, waits on other streams' events and may block the host on a slot's event: none of
  // that belongs in a stream capture (the graph would bake a stale table) — the per-GEMM kernels serve there
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return false;
  return true;
}

// throw_it = true (the latent Transformer's entry points, svg_transformer_status): raise the give-up as that call's error.
// throw_it = false (VAE / UNet / DDIM entry points, which did nothing wrong and must still run): turn the walk off, say so on stderr, and keep
// the event for the next Transformer call or status query to raise.
bool xf_walk_check(svg_ctx* ctx, bool throw_it) {
  (void)ctx;
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return false;
  std::lock_guard<std::mutex> lk(g_mu);
  WalkDev& D = g_dev[dev];
  if (!D.init) return false;
  bool gave_up = false;
  for (int i = 0; i < kRing; ++i)
    if (__atomic_load_n(&D.habort[i], __ATOMIC_RELAXED) != 0) { gave_up = true; __atomic_store_n(&D.habort[i], 0u, __ATOMIC_RELAXED); }
  if (gave_up) {
    D.disabled = true;                                 // one contention event must not make every later forward raise: fall back for good
    D.pending = true;
    fprintf(stderr, "[svg_hip] %s\n", kGaveUp);
  }
  if (D.pending && throw_it) {
    D.pending = false;
    throw SvgHipError(std::string(kGaveUp));          // a device-side failure (SVG_ERR_RUNTIME), not a bad argument
  }
  return D.pending;
}

int xf_walk_grid() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  return g_dev[dev].init ? g_dev[dev].n_wg : 0;
}

static void walk_launch_impl(const WalkOp* ops, int n_ops, int rows, int64_t lds_bytes, hipStream_t s, bool small);

void xf_walk_launch(svg_ctx* ctx, const WalkOp* ops, int n_ops, int rows, int64_t lds_bytes, hipStream_t s) {
  (void)ctx;
  walk_launch_impl(ops, n_ops, rows, lds_bytes, s, false);
}

void xf_walk_small_launch(svg_ctx* ctx, const WalkOp* ops, int n_ops, int64_t lds_bytes, hipStream_t s) {
  (void)ctx;
  walk_launch_impl(ops, n_ops, kWalkSmallRows, lds_bytes, s, true);
}

static void walk_launch_impl(const WalkOp* ops, int n_ops, int rows, int64_t lds_bytes, hipStream_t s, bool small) {
  SVG_CHECK(n_ops >= 1 && n_ops <= kWalkMaxOps, "xf_walk: %d stages (at most %d)", n_ops, kWalkMaxOps);
  SVG_CHECK(xf_walk_available(rows, lds_bytes), "xf_walk: %d rows / %lld bytes of LDS unsupported", rows, (long long)lds_bytes);
  // LDS above half the compute unit's: one workgroup per compute unit
  const int64_t lds = std::max<int64_t>(lds_bytes, kMinLds);
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  validate_table(ops, n_ops, rows, lds, small, std::max(1, xf_walk_grid()));
  std::lock_guard<std::mutex> lk(g_mu);
  WalkDev& D = g_dev[dev];
  SVG_CHECK(D.init, "xf_walk: xf_walk_init_device() has not run on device %d", dev);
  const int slot = D.next;
  D.next = (D.next + 1) % kRing;
  // the slot's previous launch (16 launches ago) must be over before its pinned table and barrier words are reused — in practice it is:
  // the query succeeds and the host never blocks here
  if (D.used[slot] && hipEventQuery(D.done[slot]) != hipSuccess) HIP_OK(hipEventSynchronize(D.done[slot]));
  memcpy(D.hops[slot], ops, sizeof(WalkOp) * n_ops);
  for (int i = n_ops - 1, nxt = n_ops; i >= 0; --i) {     // the weight stream's chain: the next GEMM stage after each stage
    D.hops[slot][i].next_gemm = nxt;
    if (D.hops[slot][i].kind == (small ? WK_GEMMF : WK_GEMM)) nxt = i;
  }
  // one walk at a time on the device: a second one could take compute units the first still needs for its unplaced workgroups
  if (D.last >= 0 && D.last != slot) HIP_OK(hipStreamWaitEvent(s, D.done[D.last], 0));
  HIP_OK(hipMemcpyAsync(D.dops[slot], D.hops[slot], sizeof(WalkOp) * n_ops, hipMemcpyHostToDevice, s));
  HIP_OK(hipMemsetAsync(D.sync[slot], 0, sizeof(WalkSync), s));
  const int mt = walk_mt(rows);
  WalkOp* dops = D.dops[slot];
  WalkSync* sy = D.sync[slot];
  unsigned* ha = D.habort + slot;
  // test hook: a barrier that waits for 8 workgroups more than the grid has never completes — exercises the give-up path end to end
  const unsigned expect = (unsigned)D.n_wg + (D.test_giveup ? 8u : 0u);
  if (small) {
    unsigned long long timeout = D.timeout;
    const WalkOp* cops = dops;
    if (D.coop) {
      void* args[] = {(void*)&cops, (void*)&n_ops, (void*)&sy, (void*)&ha, (void*)&expect, (void*)&timeout};
      HIP_OK(hipLaunchCooperativeKernel((const void*)xf_walk_small_kernel, dim3(D.n_wg), dim3(WK_THREADS), args, (unsigned)lds, s));
    } else {
      hipLaunchKernelGGL(xf_walk_small_kernel, dim3(D.n_wg), dim3(WK_THREADS), (size_t)lds, s, cops, n_ops, sy, ha, expect, timeout);
    }
  } else switch (mt) {
    case 1: launch<1>(D, lds, s, dops, n_ops, sy, ha, expect); break;
    case 2: launch<2>(D, lds, s, dops, n_ops, sy, ha, expect); break;
    case 3: launch<3>(D, lds, s, dops, n_ops, sy, ha, expect); break;
    case 4: launch<4>(D, lds, s, dops, n_ops, sy, ha, expect); break;
    case 6: launch<6>(D, lds, s, dops, n_ops, sy, ha, expect); break;
    case 8: launch<8>(D, lds, s, dops, n_ops, sy, ha, expect); break;
    default: launch<11>(D, lds, s, dops, n_ops, sy, ha, expect); break;
  }
  check_launch("xf_walk");
  HIP_OK(hipEventRecord(D.done[slot], s));
  D.used[slot] = true;
  D.last = slot;
}
