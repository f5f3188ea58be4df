// Internal runtime types of libsvg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <unordered_map>
#include <vector>
#include <stdexcept>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <atomic>
#include <mutex>
#include <shared_mutex>

// Storage type of the Stable-Diffusion side (activations and packed weights; accumulation is always f32).  Every SD source is
// compiled twice into libsvg_hip.so: once with h16 = bf16 (namespace sd_bf16, the default) and once with -DSVG_F16, h16 = IEEE
// half (namespace sd_f16: the reference's autocast arithmetic, utils/sd_utils.py:246).  A model picks its namespace when it is
// configured (kv key f16=1); files compiled once see the bf16 namespace.
#ifdef SVG_F16
typedef _Float16 h16;
#define SDNS sd_f16
#define SD_F16 1
#define H16_ONE1 0x00003C00u          /* 1.0 as one 16-bit pattern / as a pair */
#define H16_ONE2 0x3C003C00u
#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#else
typedef __bf16 h16;
#define SDNS sd_bf16
#define SD_F16 0
#define H16_ONE1 0x00003F80u
#define H16_ONE2 0x3F803F80u
#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef h16 h16x4 __attribute__((ext_vector_type(4)));
typedef uint16_t u16;

// SvgError: the caller handed over something the library cannot take (shape, size, missing weight, call order): the
// C ABI returns SVG_ERR_INVALID (-2) and the Python facade raises ValueError.  SvgHipError: a HIP call or a kernel launch
// failed: SVG_ERR_RUNTIME (-1), RuntimeError.
struct SvgError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct SvgHipError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

#define SVG_CHECK(cond, ...)                                   \
  do {                                                         \
    if (!(cond)) {                                             \
      char _b[512];                                            \
      snprintf(_b, sizeof(_b), __VA_ARGS__);                   \
      throw SvgError(std::string(_b));                         \
    }                                                          \
  } while (0)

#define HIP_OK(expr)                                                                   \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      char _b[512];                                                                    \
      snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),  \
               __FILE__, __LINE__);                                                    \
      throw SvgHipError(std::string(_b));                                              \
    }                                                                                  \
  } while (0)

// Tuning / debugging knobs ($SVG_*): looked up ONCE per name and cached (a launch path must not call getenv — VERDICT r03 hygiene);
// svg_env_refresh() (C ABI) drops the cache so that a test that changes a knob in-process is seen by the next call.
int64_t svg_env_i64(const char* name, int64_t dflt);

static inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- device memory owned by the library (weights) ---------------------------------------------
struct DevBuf {
  void* p = nullptr;
  int64_t bytes = 0;
};

// A weight as handed over (f32, original shape) and/or its packed forms.
struct Weight {
  std::vector<int64_t> shape;
  float* f32 = nullptr;   // device, original layout (kept for 1-D params / f32 models)
  int64_t numel = 0;
};

// ---- workspace arena: bump allocator with scopes; "dry" mode measures the high-water mark ----
struct Arena {
  char* base = nullptr;
  int64_t cap = 0;
  int64_t top = 0;
  int64_t high = 0;
  bool dry = false;
  std::vector<int64_t> marks;
  void* alloc(int64_t bytes) {
    int64_t off = align_up(top, 256);
    top = off + bytes;
    if (top > high) high = top;
    if (dry) return (void*)(uintptr_t)(0x1000 + off);   // never dereferenced: launches are skipped
    SVG_CHECK(top <= cap, "workspace arena overflow: need %lld, have %lld", (long long)top, (long long)cap);
    return base + off;
  }
  template <typename T> T* get(int64_t n) { return (T*)alloc(n * (int64_t)sizeof(T)); }
  void push() { marks.push_back(top); }
  void pop() { top = marks.back(); marks.pop_back(); }
  void reset() { top = 0; marks.clear(); }
};

// ---- profiling: hipEvent brackets per kernel family -------------------------------------------
struct ProfEntry {
  std::string name;
  int64_t calls = 0;
  double flops = 0, bytes = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
};

struct svg_ctx {
  int device = 0;
  std::string err;
  Arena arena;
  DevBuf arena_buf;
  bool prof = false;
  bool prof_detail = false;            // svg_prof_enable(ctx, 2): additionally one entry per call-site signature ("@kind|shape")
  std::vector<ProfEntry> prof_entries;
  std::unordered_map<std::string, ProfEntry> prof_shapes;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  // models (opaque here; defined in their own translation units)
  struct XfModel* xf = nullptr;
  struct VaeIface* vae = nullptr;     // sd_bf16::VaeModel or sd_f16::VaeModel (configure key f16=1)
  struct UnetIface* unet = nullptr;
  struct ClipTextModel* clip = nullptr;
  struct MiniLmModel* minilm = nullptr;
  struct I3dModel* i3d = nullptr;
  static constexpr int kCtxSlot = 6;                 // owned[] index of allocations that belong to the context itself
  std::vector<void*> owned[7];   // device allocations per model id (kCtxSlot = context) freed at reconfigure / destroy
  int cur_model = kCtxSlot;
  uint64_t* seed_scratch = nullptr;   // device word for svg_op_dropout_mask
  // Workspace growth never frees and never synchronises: the outgrown block may still be read by this context's queued kernels and
  // another thread of the process may be inside a stream capture (where a device-wide sync is an error).  Outgrown blocks are parked
  // here and released by release_retired() at the points that hold the device-wide lock (reserve / plan end / destroy).
  std::vector<DevBuf> arena_retired;
  int64_t arena_growths = 0;          // (re)allocations of the workspace since svg_create: constant in steady state
  // svg_plan_begin .. svg_plan_end: every planned call runs its dry pass only (nothing is launched) and the largest workspace
  // need is kept; svg_plan_end sizes the arena for it once
  bool plan_only = false;
  int64_t plan_high = 0;
  void* dalloc(int64_t bytes);
  void ensure_arena(int64_t bytes);
  void release_retired();             // caller holds DeviceWideScope
};

// One process-wide lock between stream captures and device-wide operations.  HIP rejects hipDeviceSynchronize (and anything that
// implies it: hipFree of live memory, re-finalize) while ANY stream of the process is capturing, whichever thread asks.  Capture
// windows (the DDIM loop's second step, the training step) hold it shared; every device-wide synchronisation / free holds it
// exclusive and therefore waits the few milliseconds a capture window takes to record.
std::shared_mutex& svg_capture_mutex();
extern std::atomic<int> g_captures_active;          // capture windows open right now (svg_debug_captures_active)
struct CaptureScope {
  std::shared_lock<std::shared_mutex> lk;
  CaptureScope() : lk(svg_capture_mutex()) { g_captures_active.fetch_add(1); }
  ~CaptureScope() { g_captures_active.fetch_sub(1); }
  CaptureScope(const CaptureScope&) = delete;
};
struct DeviceWideScope {
  std::unique_lock<std::shared_mutex> lk;
  DeviceWideScope() : lk(svg_capture_mutex()) {}
  DeviceWideScope(const DeviceWideScope&) = delete;
};

// Kernel families for the profiler (index into prof_entries)
enum ProfKind {
  PK_GEMM = 0, PK_CONV3, PK_ATTN, PK_GNORM, PK_LNORM, PK_ELT, PK_XF_GEMM, PK_XF_MISC, PK_SOFTMAX,
  PK_UNET_STEP,   // outer bracket: one whole UNet call + scheduler step of the DDIM loop (contains the families above)
  PK_COUNT
};
extern const char* kProfNames[PK_COUNT];

struct ProfScope {
  svg_ctx* c; int kind; hipStream_t s; hipEvent_t e0 = nullptr, e1 = nullptr;
  ProfScope(svg_ctx* c_, int kind_, hipStream_t s_, double flops, double bytes, const char* tag = nullptr);
  ~ProfScope();
};

// C-ABI failure path: records the message on the context (or process-wide when there is none) and returns the code
int svg_fail(svg_ctx* ctx, const std::exception& e);

// Launch guard: in arena-dry mode nothing is launched.
#define SVG_LAUNCHING(ctx) (!(ctx)->arena.dry)

static inline void check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    char b[256];
    snprintf(b, sizeof(b), "launch of %s failed: %s", what, hipGetErrorString(e));
    throw SvgHipError(std::string(b));
  }
}
