"""module path of the reference (loaders/kitti_loader.py): re-export of the MI355X build's dataset front-end"""
from sd_video_gen_amd.loaders import Kitti  # noqa: F401
