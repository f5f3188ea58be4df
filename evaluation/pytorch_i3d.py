"""`evaluation.pytorch_i3d` of the reference (evaluation/pytorch_i3d.py:136-322) -> the MI355X-native mirror of InceptionI3d."""
from sd_video_gen_amd.fvd import InceptionI3d  # noqa: F401
