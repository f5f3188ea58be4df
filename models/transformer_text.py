"""`models.transformer_text` of the reference (models/transformer_text.py) -> the MI355X-native mirror."""
from sd_video_gen_amd.transformer_text import Transformer  # noqa: F401
