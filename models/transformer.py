"""`models.transformer` of the reference (models/transformer.py) -> the MI355X-native mirror."""
from sd_video_gen_amd.transformer import Transformer, PositionalEncoding  # noqa: F401
