// C ABI: model configure / load / finalize dispatch.
#include "models.h"
#include "../../include/svg_hip.h"

void destroy_models(svg_ctx* ctx) {
  if (ctx->xf) { xf_train_free(ctx->xf); ctx->xf->ws.clear(); delete ctx->xf; ctx->xf = nullptr; }
  if (ctx->vae) { ctx->vae->ws.clear(); delete ctx->vae; ctx->vae = nullptr; }
  if (ctx->unet) { ctx->unet->ws.clear(); delete ctx->unet; ctx->unet = nullptr; }
  if (ctx->clip) { ctx->clip->ws.clear(); delete ctx->clip; ctx->clip = nullptr; }
}

static WeightStore* store_of(svg_ctx* ctx, int model, bool create) {
  switch (model) {
    case SVG_TRANSFORMER: if (!ctx->xf && create) ctx->xf = new XfModel(); return ctx->xf ? &ctx->xf->ws : nullptr;
    case SVG_VAE: if (!ctx->vae && create) ctx->vae = new VaeModel(); return ctx->vae ? &ctx->vae->ws : nullptr;
    case SVG_UNET: if (!ctx->unet && create) ctx->unet = new UnetModel(); return ctx->unet ? &ctx->unet->ws : nullptr;
    case SVG_CLIP_TEXT: if (!ctx->clip && create) ctx->clip = new ClipTextModel(); return ctx->clip ? &ctx->clip->ws : nullptr;
    default: throw SvgError("unknown model id " + std::to_string(model));
  }
}

extern "C" {

int svg_model_configure(svg_ctx* ctx, int model, const char* kv) {
  try {
    SVG_CHECK(ctx, "null context");
    // a new configuration starts a fresh model: drop the previous weights and packed buffers
    if (WeightStore* old = store_of(ctx, model, true)) old->clear();
    HIP_OK(hipDeviceSynchronize());
    for (void* p : ctx->owned[model]) hipFree(p);
    ctx->owned[model].clear();
    if (model == SVG_TRANSFORMER) { xf_train_free(ctx->xf); ctx->xf->pe = nullptr; ctx->xf->iota = nullptr; }
    if (model == SVG_TRANSFORMER) ctx->xf->configure(kv);
    else if (model == SVG_VAE) ctx->vae->configure(kv);
    else if (model == SVG_CLIP_TEXT) ctx->clip->configure(kv);
    else ctx->unet->configure(kv);
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}

int svg_load_weight(svg_ctx* ctx, int model, const char* name, const float* data, const int64_t* shape, int ndim) {
  try {
    SVG_CHECK(ctx && name && data && shape && ndim >= 1 && ndim <= 4, "svg_load_weight: bad arguments");
    WeightStore* ws = store_of(ctx, model, true);
    HIP_OK(hipSetDevice(ctx->device));
    ws->put(ctx, name, data, shape, ndim);
    if (model == SVG_TRANSFORMER) { xf_train_free(ctx->xf); ctx->xf->ready = false; }   // new weights: a fresh optimizer state
    else if (model == SVG_VAE) ctx->vae->ready = false;
    else if (model == SVG_CLIP_TEXT) ctx->clip->ready = false;
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
      else ctx->unet->finalize(ctx, n_params);
    } catch (...) { ctx->cur_model = svg_ctx::kCtxSlot; throw; }
    ctx->cur_model = svg_ctx::kCtxSlot;
    HIP_OK(hipDeviceSynchronize());
    return 0;
  } catch (const std::exception& e) { return svg_fail(ctx, e); }
}

}  // extern "C"
