"""`evaluation.fvd_2` of the reference (evaluation/fvd_2.py:7-107) -> the MI355X-native mirror: FVD preprocessing, I3D logits and the
Frechet distance run in the library (sd_video_gen_amd/fvd.py -> svg_fvd_preprocess / svg_i3d_forward / svg_frechet_distance)."""
from sd_video_gen_amd.fvd import (preprocess, get_fvd_logits, get_logits, frechet_distance, load_i3d_pretrained, all_gather)  # noqa: F401
