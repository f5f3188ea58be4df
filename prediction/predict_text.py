"""`prediction.predict_text` of the reference (prediction/predict_text.py) -> the MI355X-native mirror: `predict(model,
X, cls_list)` and the class-name helper; the sampling loop itself is `sd_video_gen_amd.predict.sample_clips(...,
cls_list=...)` (the reference's loop differs from predict.py only by that argument)."""
import re

from sd_video_gen_amd.transformer_text import predict  # noqa: F401


def splitClassNames(classes):
    """'WallPushups' -> 'Wall Pushups' (predict_text.py:18-33: a break before every capital letter)."""
    return [" ".join(w for w in re.sub(r"([A-Z])", r"*\1", s).split("*") if w != "") for s in classes]
