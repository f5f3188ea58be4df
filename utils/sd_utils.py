"""`utils.sd_utils` of the reference (utils/sd_utils.py) -> the MI355X-native SDUtils."""
from sd_video_gen_amd.sd_utils import SDUtils  # noqa: F401
