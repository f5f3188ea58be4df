"""MI355X-native sampling path of sd-video-gen: VAE encode -> latent Transformer -> SD-UNet DDIM
denoise -> VAE decode, executed by hand-written HIP kernels (gfx950) behind a C ABI
(``include/svg_hip.h``, ``libsvg_hip.so``).  Host code mirrors the reference's Python surface:

  sd_video_gen_amd.config        utils/config.py       flags + YAML
  sd_video_gen_amd.transformer   models/transformer.py Transformer (same state_dict keys)
  sd_video_gen_amd.sd_utils      utils/sd_utils.py     SDUtils
  sd_video_gen_amd.predict       prediction/predict.py predict() + the per-clip loop
  sd_video_gen_amd.sharding      clip sharding over ranks + the one all-gather

There is no CPU fallback: anything that computes needs the built library and a gfx950 GPU.
"""
__version__ = "0.1.0"
