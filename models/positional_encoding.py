"""`models.positional_encoding` of the reference -> the MI355X-native mirror (buffer only; the add is in the HIP forward)."""
from sd_video_gen_amd.transformer import PositionalEncoding  # noqa: F401
