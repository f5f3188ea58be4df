// Context, workspace arena, weight store, profiler, C-ABI error plumbing.
#include <mutex>
#include "models.h"
#include "xf_walk.h"
#include "../../include/svg_hip.h"
#include <algorithm>
#include <sstream>
#include "build/srchash.h"

using namespace SDNS;   // the storage-independent element-wise kernels (resize, scheduler steps) of the bf16 namespace

const char* kProfNames[PK_COUNT] = {"gemm", "conv3x3", "attention", "groupnorm", "layernorm", "eltwise",
                                    "xf_gemm", "xf_misc", "softmax", "unet_step"};

static std::string g_err;   // errors without a context

int svg_fail(svg_ctx* ctx, const std::exception& e) {
  if (ctx) ctx->err = e.what(); else g_err = e.what();
  return dynamic_cast<const SvgError*>(&e) ? SVG_ERR_INVALID : SVG_ERR_RUNTIME;
}

void* svg_ctx::dalloc(int64_t bytes) {
  void* p = nullptr;
  HIP_OK(hipMalloc(&p, (size_t)std::max<int64_t>(bytes, 256)));
  owned[cur_model].push_back(p);
  return p;
}

std::shared_mutex& svg_capture_mutex() {
  static std::shared_mutex mu;
  return mu;
}
std::atomic<int> g_captures_active{0};

// Grows the workspace WITHOUT freeing and WITHOUT synchronising: the outgrown block stays alive (kernels this context queued may
// still read it; another thread may be capturing, where hipDeviceSynchronize / hipFree are errors) and is parked in arena_retired.
// Callers that size the workspace up front (svg_plan_begin/_end, svg_reserve_workspace) never get here in steady state.
void svg_ctx::ensure_arena(int64_t bytes) {
  bytes = align_up(bytes + (1 << 20), 1 << 20);
  if (arena_buf.bytes >= bytes) return;
  int64_t parked = 0;
  for (const DevBuf& b : arena_retired) parked += b.bytes;
  if (parked > ((int64_t)8 << 30)) {            // bound what growth-by-growth use can strand: wait for the captures to end, release
    DeviceWideScope lk;
    HIP_OK(hipDeviceSynchronize());
    release_retired();
  }
  void* p = nullptr;
  HIP_OK(hipMalloc(&p, (size_t)bytes));
  if (arena_buf.p) arena_retired.push_back(arena_buf);
  arena_buf.p = p;
  arena_buf.bytes = bytes;
  arena.base = (char*)p;
  arena.cap = bytes;
  ++arena_growths;
}

void svg_ctx::release_retired() {
  for (DevBuf& b : arena_retired) hipFree(b.p);
  arena_retired.clear();
}

ProfScope::ProfScope(svg_ctx* c_, int kind_, hipStream_t s_, double flops, double bytes, const char* tag) : c(c_), kind(kind_), s(s_) {
  if (!c->prof) return;
  auto next_event = [&]() {
    if (c->ev_used == c->ev_pool.size()) {
      hipEvent_t e;
      HIP_OK(hipEventCreate(&e));
      c->ev_pool.push_back(e);
    }
    return c->ev_pool[c->ev_used++];
  };
  e0 = next_event();
  e1 = next_event();
  ProfEntry& pe = c->prof_entries[kind];
  pe.calls++;
  pe.flops += flops;
  pe.bytes += bytes;
  pe.ev.push_back({e0, e1});
  if (c->prof_detail && tag) {
    ProfEntry& d = c->prof_shapes[std::string("@") + kProfNames[kind] + "|" + tag];
    d.calls++; d.flops += flops; d.bytes += bytes;
    d.ev.push_back({e0, e1});
  }
  hipEventRecord(e0, s);
}
ProfScope::~ProfScope() {
  if (e1) hipEventRecord(e1, s);
}

// ---- weight store ---------------------------------------------------------------------------------
void WeightStore::put(svg_ctx* ctx, const std::string& name, const float* data, const int64_t* shape, int ndim) {
  Weight w;
  w.numel = 1;
  for (int i = 0; i < ndim; ++i) { w.shape.push_back(shape[i]); w.numel *= shape[i]; }
  SVG_CHECK(w.numel > 0, "weight %s is empty", name.c_str());
  auto it = map.find(name);
  if (it != map.end() && it->second.numel == w.numel && it->second.f32) {
    w.f32 = it->second.f32;   // reload in place
  } else {
    HIP_OK(hipMalloc((void**)&w.f32, (size_t)w.numel * sizeof(float)));
  }
  HIP_OK(hipMemcpy(w.f32, data, (size_t)w.numel * sizeof(float), hipMemcpyDefault));
  map[name] = w;
}
const Weight& WeightStore::get(const std::string& name) const {
  auto it = map.find(name);
  SVG_CHECK(it != map.end() && it->second.f32, "missing weight: %s", name.c_str());
  return it->second;
}
const Weight& WeightStore::get(const std::string& name, std::initializer_list<int64_t> shape) const {
  const Weight& w = get(name);
  bool ok = w.shape.size() == shape.size();
  if (ok) {
    size_t i = 0;
    for (int64_t s : shape) ok = ok && (w.shape[i++] == s);
  }
  if (!ok) {
    std::ostringstream os;
    os << "weight " << name << " has shape (";
    for (auto s : w.shape) os << s << ",";
    os << ") expected (";
    for (auto s : shape) os << s << ",";
    os << ")";
    throw SvgError(os.str());
  }
  return w;
}
void WeightStore::release(const std::string& name) {
  auto it = map.find(name);
  if (it != map.end() && it->second.f32) {
    hipFree(it->second.f32);
    it->second.f32 = nullptr;
  }
}
void WeightStore::clear() {
  for (auto& kv : map)
    if (kv.second.f32) hipFree(kv.second.f32);
  map.clear();
}
int64_t WeightStore::total_params() const {
  int64_t n = 0;
  for (auto& kv : map) n += kv.second.numel;
  return n;
}

// ---- "key=v,v;key=v" parser -------------------------------------------------------------------------
std::unordered_map<std::string, std::vector<int64_t>> parse_kv(const char* kv) {
  std::unordered_map<std::string, std::vector<int64_t>> out;
  if (!kv) return out;
  std::string s(kv), item;
  std::stringstream ss(s);
  while (std::getline(ss, item, ';')) {
    if (item.empty()) continue;
    size_t eq = item.find('=');
    SVG_CHECK(eq != std::string::npos, "bad config item '%s'", item.c_str());
    std::string key = item.substr(0, eq), vals = item.substr(eq + 1), v;
    std::stringstream vs(vals);
    std::vector<int64_t> arr;
    while (std::getline(vs, v, ',')) arr.push_back(std::stoll(v));
    out[key] = arr;
  }
  return out;
}

// ---- C ABI: context ------------------------------------------------------------------------------------
#define API_BEGIN try {
#define API_END(ctx)                                   \
  return 0;                                            \
  }                                                    \
  catch (const std::exception& e) { return svg_fail(ctx, e); }

static std::mutex g_env_mu;
static std::unordered_map<std::string, int64_t> g_env;      // values of the $SVG_* knobs already looked up (absent -> its default)

int64_t svg_env_i64(const char* name, int64_t dflt) {
  std::lock_guard<std::mutex> lk(g_env_mu);
  std::string key(name);
  key += '\x01';
  key += std::to_string(dflt);
  auto it = g_env.find(key);
  if (it != g_env.end()) return it->second;
  const char* e = getenv(name);
  const int64_t v = (e && *e) ? atoll(e) : dflt;
  g_env.emplace(std::move(key), v);
  return v;
}

extern "C" {

void svg_env_refresh(void) {
  {
    std::lock_guard<std::mutex> lk(g_env_mu);
    g_env.clear();
  }
  xf_walk_env_refresh();
}

const char* svg_version(void) { return "svg_hip 0.3 (gfx950, bf16+fp16) src " SVG_SRC_HASH; }

int svg_create(int device_id, svg_ctx** out) {
  svg_ctx* ctx = nullptr;
  API_BEGIN
  SVG_CHECK(out, "svg_create: out is NULL");
  int n = 0;
  HIP_OK(hipGetDeviceCount(&n));
  SVG_CHECK(device_id >= 0 && device_id < n, "svg_create: device %d of %d", device_id, n);
  HIP_OK(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  HIP_OK(hipGetDeviceProperties(&prop, device_id));
  SVG_CHECK(std::string(prop.gcnArchName).find("gfx950") != std::string::npos,
            "svg_create: this library is built for gfx950 only, device reports %s", prop.gcnArchName);
  sd_bf16::sd_init_device();
  sd_f16::sd_init_device();
  xf_train_init_device();
  xformer_init_device();
  xf_walk_init_device();
  ctx = new svg_ctx();
  ctx->device = device_id;
  ctx->prof_entries.resize(PK_COUNT);
  for (int i = 0; i < PK_COUNT; ++i) ctx->prof_entries[i].name = kProfNames[i];
  *out = ctx;
  API_END((svg_ctx*)nullptr)
}

void svg_destroy(svg_ctx* ctx) {
  if (!ctx) return;
  DeviceWideScope lk;                 // waits for capture windows of other threads: a device-wide sync inside one is an error
  hipDeviceSynchronize();
  destroy_models(ctx);
  for (auto& v : ctx->owned) for (void* p : v) hipFree(p);
  ctx->release_retired();
  if (ctx->arena_buf.p) hipFree(ctx->arena_buf.p);
  for (auto e : ctx->ev_pool) hipEventDestroy(e);
  delete ctx;
}

const char* svg_last_error(svg_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int64_t svg_workspace_bytes(svg_ctx* ctx) { return ctx ? ctx->arena_buf.bytes : 0; }
int64_t svg_workspace_growths(svg_ctx* ctx) { return ctx ? ctx->arena_growths : 0; }
int svg_debug_captures_active(void) { return g_captures_active.load(); }

int svg_reserve_workspace(svg_ctx* ctx, int64_t bytes) {
  API_BEGIN
  SVG_CHECK(ctx && bytes >= 0, "svg_reserve_workspace: bad arguments");
  HIP_OK(hipSetDevice(ctx->device));
  ctx->ensure_arena(bytes);
  if (!ctx->arena_retired.empty()) {
    DeviceWideScope lk;
    HIP_OK(hipDeviceSynchronize());   // kernels queued on the outgrown blocks
    ctx->release_retired();
  }
  API_END(ctx)
}
int svg_plan_begin(svg_ctx* ctx) {
  API_BEGIN
  SVG_CHECK(ctx, "svg_plan_begin: null context");
  ctx->plan_only = true;
  ctx->plan_high = 0;
  API_END(ctx)
}
int svg_plan_end(svg_ctx* ctx, int64_t* bytes) {
  API_BEGIN
  SVG_CHECK(ctx && ctx->plan_only, "svg_plan_end without svg_plan_begin");
  ctx->plan_only = false;
  if (bytes) *bytes = ctx->plan_high;
  const int rc = svg_reserve_workspace(ctx, ctx->plan_high);
  if (rc != 0) return rc;
  API_END(ctx)
}

int svg_prof_enable(svg_ctx* ctx, int on) {
  if (!ctx) return -1;
  ctx->prof = on != 0;
  ctx->prof_detail = on == 2;
  return 0;
}
// the brackets of this context only: no device-wide synchronisation (another thread may be capturing)
static void prof_wait(svg_ctx* ctx) {
  for (size_t i = 0; i < ctx->ev_used; ++i) HIP_OK(hipEventSynchronize(ctx->ev_pool[i]));
}
int svg_prof_reset(svg_ctx* ctx) {
  API_BEGIN
  prof_wait(ctx);
  for (auto& e : ctx->prof_entries) { e.calls = 0; e.flops = 0; e.bytes = 0; e.ev.clear(); }
  ctx->prof_shapes.clear();
  ctx->ev_used = 0;
  API_END(ctx)
}
int svg_prof_report(svg_ctx* ctx, char* buf, int buflen) {
  API_BEGIN
  prof_wait(ctx);
  std::ostringstream os;
  for (auto& e : ctx->prof_entries) {
    if (!e.calls) continue;
    double ms = 0;
    for (auto& p : e.ev) {
      float t = 0;
      if (hipEventElapsedTime(&t, p.first, p.second) == hipSuccess) ms += t;
    }
    os << e.name << " " << e.calls << " " << ms << " " << e.flops << " " << e.bytes << "\n";
  }
  for (auto& kv : ctx->prof_shapes) {
    double ms = 0;
    for (auto& p : kv.second.ev) {
      float t = 0;
      if (hipEventElapsedTime(&t, p.first, p.second) == hipSuccess) ms += t;
    }
    os << kv.first << " " << kv.second.calls << " " << ms << " " << kv.second.flops << " " << kv.second.bytes << "\n";
  }
  std::string s = os.str();
  SVG_CHECK((int)s.size() + 1 <= buflen, "svg_prof_report: buffer too small");
  memcpy(buf, s.c_str(), s.size() + 1);
  API_END(ctx)
}

// ---- C ABI: operator level (the hooks on 16-bit buffers are in sd_ops.cpp, once per storage type) ----
int svg_op_xf_gemm(svg_ctx* ctx, const float* X, const float* W, const float* bias, float* Y, int M, int N, int K, int relu_in,
                   void* stream) {
  API_BEGIN
  run_planned(ctx, [&]() { xf_gemm(ctx, X, W, bias, Y, M, N, K, relu_in, (hipStream_t)stream); });
  API_END(ctx)
}

int svg_resize_bilinear_f32(svg_ctx* ctx, const float* src, int planes, int h, int w, float* dst, int oh, int ow, void* stream) {
  try {
    SVG_CHECK(ctx && src && dst && planes > 0 && h > 0 && w > 0 && oh > 0 && ow > 0, "svg_resize_bilinear_f32: bad arguments");
    resize_bilinear_f32(src, dst, planes, h, w, oh, ow, (hipStream_t)stream);
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}

int svg_resize_nearest_u8(svg_ctx* ctx, const uint8_t* src, int N, int sh, int sw, int C, uint8_t* dst, int dh, int dw,
                          void* stream) {
  API_BEGIN
  resize_nearest_u8(src, dst, N, sh, sw, C, dh, dw, (hipStream_t)stream);
  API_END(ctx)
}

}  // extern "C"
