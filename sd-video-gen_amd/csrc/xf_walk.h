// Stage table of the layer-walking kernel of the latent Transformer (xf_walk.hip): ONE launch walks embedding -> encoder layers ->
// decoder layers -> output projection of models/transformer.py:47-68 (torch.nn.Transformer, post-norm, ReLU).
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

struct svg_ctx;

enum WalkKind : int {
  WK_GEMM = 0,    // slab[kz][M][N] = X[M][kz-th 128-wide K slice] . W[N][same slice]^T      (tiles of 128 columns x 128 k)
  WK_RED = 1,     // Y = act(sum_z slab[z] + bias + res)                                     (element-wise over all workgroups)
  WK_LN = 2,      // Y = LN(sum_z slab[z] + bias + res) g1 + b1;  Y2 = LN(Y) g2 + b2          (a row per workgroup)
  WK_ATTN = 3,    // o = softmax(q k^T / sqrt(hd) + mask + kpad) v per (batch row, head)
  WK_EMBED = 4,   // Y[t][b] = [sum_z slab[z][b][t] + bias | text[b]] * scale + pe[pe_row[b]]
  // small-row launch (xf_walk_small_kernel, at most 8 rows): a whole-K GEMM with the LayerNorm of its input and its bias / ReLU / residual
  // (or the embedding's scale + positional row) folded in — no split-K slabs, so no reduce stages:
  //   x = LN2(LN1(X)) (each optional; Yln <- the normalised rows);  Y[m][n] = act(x[m] . W[n] + bias[n]) (+ res[m][n])
  WK_GEMMF = 5,
};

struct alignas(16) WalkOp {
  int kind, bar;                 // bar: device-wide barrier before the stage (its inputs come from the stage before)
  int next_gemm, pad_;           // index of the next GEMM stage (filled in by xf_walk_launch)
  int M, N, K, ld;               // GEMM: X is M x K with row stride ld, W is N x K;  RED / LN / EMBED: an M x N result
  int ksplit, relu;              // slabs to add up (0: none);  RED: ReLU on the result
  const float* X; const float* W; float* slab;
  const float* bias; const float* res; float* Y;
  const float* g1; const float* b1; const float* g2; const float* b2; float* Y2;
  float eps;
  // ATTN: q / k / v point at their column blocks of already reduced projections; rows are (t * B + b) with strides q_ld / kv_ld floats;
  // q_span / kv_span: floats from the base to the end of the buffer
  int Tq, Tk, B, heads, hd;
  int q_ld, kv_ld, q_span, kv_span;
  const float* qs; const float* ks; const float* vs;
  const float* mask; const float* kpad;
  // EMBED
  const float* pe; const int32_t* pe_row; const float* text; int d_txt, T; float scale;
  // GEMMF: X rows are (b, t) batch-first when perm (the launch's inputs), output rows (t, b); ldy / ld_res: row strides of Y / res;
  // reuse_x: the X tile (and its LayerNorm) of the previous stage is still in LDS; Yln: where the normalised rows go (M x K)
  int ldy, ld_res, reuse_x, perm;
  float* Yln;
};

// rows one launch serves (the accumulator tiles of a workgroup: 11 x 16) and the shapes a GEMM stage takes
constexpr int kWalkMaxRows = 176;
constexpr int kWalkMaxOps = 512;                      // stages of one launch (the pinned / device table ring is sized for it)
inline bool xf_walk_gemm_ok(int N, int K) { return N % 128 == 0 && K % 128 == 0 && N >= 128 && K >= 128; }
// dynamic LDS of a launch: the X tile of a GEMM stage (rows rounded to 16, 4 steps of 128 B) + a sink for padding DMAs, or the q/k/v
// slices of one attention job
int64_t xf_walk_lds_bytes(int rows, int Tq, int Tk, int hd);
bool xf_walk_available(int rows, int64_t lds_bytes);
// ops: host array (copied to the device through a pinned ring); the launch is ordered after the previous walk of this process on any
// stream (two resident walks could starve each other of compute units while spinning at their barriers)
void xf_walk_launch(svg_ctx* ctx, const WalkOp* ops, int n_ops, int rows, int64_t lds_bytes, hipStream_t s);
// the small-row form (WK_GEMMF / WK_ATTN stages only): at most kWalkSmallRows rows, K <= 2048 in multiples of 128, at most 8 output
// columns per workgroup and stage (N <= 8 x workgroups: the host cuts wider matrices into column blocks)
constexpr int kWalkSmallRows = 8;
constexpr int kWalkSmallMaxK = 2048;
int xf_walk_grid();                                     // workgroups of a launch on the current device (0: walk unavailable)
void xf_walk_small_launch(svg_ctx* ctx, const WalkOp* ops, int n_ops, int64_t lds_bytes, hipStream_t s);
void xf_walk_init_device();
// false when the layer-walking launch must not be used now: off for this device ($SVG_XF_WALK=0, ranks sharing the device, an earlier
// give-up, a device that cannot hold the grid) or the stream is being captured
bool xf_walk_enabled(hipStream_t s);
void xf_walk_env_refresh();                           // svg_env_refresh(): re-reads $SVG_XF_WALK* for every initialised device (re-arms the walk after a give-up)
// When an earlier walk on this device gave up at a barrier (a workgroup never became resident): clears the flag, turns the walk off for
// the device (later forwards take the per-GEMM kernels), logs it, and — throw_it — raises ONCE.  The Transformer's entry points and
// svg_transformer_status raise; the other model entry points (which must still run) call it with throw_it = false and leave the event
// pending for the Transformer's caller.  Returns whether an event is still pending.
bool xf_walk_check(svg_ctx* ctx, bool throw_it = true);
