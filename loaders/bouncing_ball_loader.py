"""module path of the reference (loaders/bouncing_ball_loader.py): re-export of the MI355X build's dataset front-end"""
from sd_video_gen_amd.loaders import BouncingBall  # noqa: F401
