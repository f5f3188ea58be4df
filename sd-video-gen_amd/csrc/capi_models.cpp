// C ABI: model configure / load / finalize dispatch.
#include "models.h"
#include "xf_walk.h"
#include "../../include/svg_hip.h"

using namespace SDNS;   // ddim_step: storage-independent f32 kernel

void destroy_models(svg_ctx* ctx) {
  if (ctx->xf) { xf_train_free(ctx->xf); ctx->xf->ws.clear(); delete ctx->xf; ctx->xf = nullptr; }
  if (ctx->vae) { ctx->vae->ws.clear(); delete ctx->vae; ctx->vae = nullptr; }
  if (ctx->unet) { ctx->unet->ws.clear(); delete ctx->unet; ctx->unet = nullptr; }
  if (ctx->clip) { ctx->clip->ws.clear(); delete ctx->clip; ctx->clip = nullptr; }
  if (ctx->minilm) { ctx->minilm->ws.clear(); delete ctx->minilm; ctx->minilm = nullptr; }
  if (ctx->i3d) { ctx->i3d->ws.clear(); delete ctx->i3d; ctx->i3d = nullptr; }
}

static WeightStore* store_of(svg_ctx* ctx, int model, bool create) {
  switch (model) {
    case SVG_TRANSFORMER: if (!ctx->xf && create) ctx->xf = new XfModel(); return ctx->xf ? &ctx->xf->ws : nullptr;
    case SVG_VAE: if (!ctx->vae && create) ctx->vae = new_vae_bf16(); return ctx->vae ? &ctx->vae->ws : nullptr;
    case SVG_UNET: if (!ctx->unet && create) ctx->unet = new_unet_bf16(); return ctx->unet ? &ctx->unet->ws : nullptr;
    case SVG_CLIP_TEXT: if (!ctx->clip && create) ctx->clip = new ClipTextModel(); return ctx->clip ? &ctx->clip->ws : nullptr;
    case SVG_MINILM: if (!ctx->minilm && create) ctx->minilm = new MiniLmModel(); return ctx->minilm ? &ctx->minilm->ws : nullptr;
    case SVG_I3D: if (!ctx->i3d && create) ctx->i3d = new I3dModel(); return ctx->i3d ? &ctx->i3d->ws : nullptr;
    default: throw SvgError("unknown model id " + std::to_string(model));
  }
}

extern "C" {

int svg_model_configure(svg_ctx* ctx, int model, const char* kv) {
  try {
    SVG_CHECK(ctx, "null context");
    // a new configuration starts a fresh model: drop the previous weights and packed buffers
    {
      DeviceWideScope lk;             // frees live device memory: not while another thread's stream is capturing
      HIP_OK(hipDeviceSynchronize());
      if (WeightStore* old = store_of(ctx, model, true)) old->clear();
      for (void* p : ctx->owned[model]) hipFree(p);
      ctx->owned[model].clear();
      if (model == SVG_TRANSFORMER) xf_train_free(ctx->xf);
    }
    if (model == SVG_TRANSFORMER) { ctx->xf->pe = nullptr; ctx->xf->iota = nullptr; }
    if (model == SVG_VAE || model == SVG_UNET) {
      // storage type of the SD networks: f16=1 -> IEEE half (the reference's autocast arithmetic), default bf16
      auto m = parse_kv(kv);
      const bool f16 = m.count("f16") && m["f16"][0] != 0;
      if (model == SVG_VAE) { delete ctx->vae; ctx->vae = f16 ? new_vae_f16() : new_vae_bf16(); }
      else { delete ctx->unet; ctx->unet = f16 ? new_unet_f16() : new_unet_bf16(); }
    }
    if (model == SVG_TRANSFORMER) ctx->xf->configure(kv);
    else if (model == SVG_VAE) ctx->vae->configure(kv);
    else if (model == SVG_CLIP_TEXT) ctx->clip->configure(kv);
    else if (model == SVG_MINILM) ctx->minilm->configure(kv);
    else if (model == SVG_I3D) ctx->i3d->configure(kv);
    else ctx->unet->configure(kv);
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}

int svg_load_weight(svg_ctx* ctx, int model, const char* name, const float* data, const int64_t* shape, int ndim) {
  try {
    SVG_CHECK(ctx && name && data && shape && ndim >= 1 && ndim <= 5, "svg_load_weight: bad arguments");
    WeightStore* ws = store_of(ctx, model, true);
    HIP_OK(hipSetDevice(ctx->device));
    ws->put(ctx, name, data, shape, ndim);
    if (model == SVG_TRANSFORMER) {   // new weights: a fresh optimizer state
      if (ctx->xf->train) { DeviceWideScope lk; xf_train_free(ctx->xf); }
      ctx->xf->ready = false;
    }
    else if (model == SVG_VAE) ctx->vae->ready = false;
    else if (model == SVG_CLIP_TEXT) ctx->clip->ready = false;
    else if (model == SVG_MINILM) ctx->minilm->ready = false;
    else if (model == SVG_I3D) ctx->i3d->ready = false;
    else ctx->unet->ready = false;
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}

int svg_finalize(svg_ctx* ctx, int model, int64_t* n_params) {
  try {
    SVG_CHECK(ctx, "null context");
    SVG_CHECK(store_of(ctx, model, false), "svg_finalize: model %d has no weights", model);
    HIP_OK(hipSetDevice(ctx->device));
    ctx->cur_model = model;
    try {
      if (model == SVG_TRANSFORMER) ctx->xf->finalize(ctx, n_params);
      else if (model == SVG_VAE) ctx->vae->finalize(ctx, n_params);
      else if (model == SVG_CLIP_TEXT) ctx->clip->finalize(ctx, n_params);
      else if (model == SVG_MINILM) ctx->minilm->finalize(ctx, n_params);
      else if (model == SVG_I3D) ctx->i3d->finalize(ctx, n_params);
      else ctx->unet->finalize(ctx, n_params);
    } catch (...) { ctx->cur_model = svg_ctx::kCtxSlot; throw; }
    ctx->cur_model = svg_ctx::kCtxSlot;
    {
      DeviceWideScope lk;
      HIP_OK(hipDeviceSynchronize());
    }
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}

/* ---- VAE / UNet / DDIM entry points: dispatch on the storage type the model was configured with ---- */
int svg_vae_encode(svg_ctx* ctx, const uint8_t* img, int N, int srcH, int srcW, int H, int W, const float* eps, float* z_out,
                   float* moments_out, void* stream) {
  try {
    SVG_CHECK(ctx && ctx->vae, "vae: model not configured");
    xf_walk_check(ctx, false);                       // an earlier layer-walking forward that gave up: walk off + logged; raised to the Transformer's caller
    ctx->vae->encode(ctx, img, N, srcH, srcW, H, W, eps, z_out, moments_out, (hipStream_t)stream);
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}
int svg_vae_decode(svg_ctx* ctx, const float* z, int N, int h, int w, uint8_t* img_out, int outH, int outW, float* float_out,
                   void* stream) {
  try {
    SVG_CHECK(ctx && ctx->vae, "vae: model not configured");
    xf_walk_check(ctx, false);                       // an earlier layer-walking forward that gave up: walk off + logged; raised to the Transformer's caller
    ctx->vae->decode(ctx, z, N, h, w, img_out, outH, outW, float_out, (hipStream_t)stream);
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}
int svg_unet_forward(svg_ctx* ctx, const float* x, int N, int h, int w, const float* timesteps, const float* ctx_emb, int ctx_len,
                     float* eps_out, void* stream) {
  try {
    SVG_CHECK(ctx && ctx->unet, "unet: model not configured");
    xf_walk_check(ctx, false);                       // an earlier layer-walking forward that gave up: walk off + logged; raised to the Transformer's caller
    ctx->unet->forward(ctx, x, N, h, w, timesteps, ctx_emb, ctx_len, eps_out, (hipStream_t)stream);
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}
int svg_ddim_loop(svg_ctx* ctx, float* z, int N, int h, int w, const float* text_emb, int ctx_len, int num_steps, int start_step,
                  float guidance, const float* noise, float* hist, void* stream) {
  try {
    SVG_CHECK(ctx && ctx->unet, "unet: model not configured");
    xf_walk_check(ctx, false);                       // an earlier layer-walking forward that gave up: walk off + logged; raised to the Transformer's caller
    ctx->unet->ddim_loop(ctx, z, N, h, w, text_emb, ctx_len, num_steps, start_step, guidance, noise, hist, (hipStream_t)stream);
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}
int svg_ddim_step(svg_ctx* ctx, const float* x, const float* eps, float* prev, int64_t n, int t, int t_prev, void* stream) {
  try {
    SVG_CHECK(ctx && ctx->unet && ctx->unet->ready, "unet: model not finalized");
    float sa, s1a, sap, s1ap;
    ctx->unet->ddim_coefs(t, t_prev, &sa, &s1a, &sap, &s1ap);
    ddim_step(x, eps, nullptr, 0.f, prev, n, sa, s1a, sap, s1ap, (hipStream_t)stream);
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}
/* Has a layer-walking Transformer forward on this device given up since the last Transformer call / status query?  Call it where
 * the forward's result is consumed (after the stream has been synchronised): 0 = no, SVG_ERR_RUNTIME = yes (svg_last_error says what;
 * that forward's output is NaN-filled and must be re-issued: the walk is now off, the per-GEMM kernels serve). */
int svg_transformer_status(svg_ctx* ctx) {
  try {
    SVG_CHECK(ctx, "null context");
    xf_walk_check(ctx, true);
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}
/* "bf16" / "fp16": the storage type of a configured SD model (SVG_VAE, SVG_UNET); "f32" for the Transformer / CLIP; NULL if absent */
const char* svg_model_dtype(svg_ctx* ctx, int model) {
  if (!ctx) return nullptr;
  if (model == SVG_VAE) return ctx->vae ? ctx->vae->dtype() : nullptr;
  if (model == SVG_UNET) return ctx->unet ? ctx->unet->dtype() : nullptr;
  if (model == SVG_TRANSFORMER) return ctx->xf ? "f32" : nullptr;
  if (model == SVG_CLIP_TEXT) return ctx->clip ? "f32" : nullptr;
  if (model == SVG_MINILM) return ctx->minilm ? "f32" : nullptr;
  if (model == SVG_I3D) return ctx->i3d ? "f32" : nullptr;
  return nullptr;
}

}  // extern "C"
