"""`utils.config` of the reference (utils/config.py:8-49): same flags, same YAML surface."""
from sd_video_gen_amd.config import load_config, remove_config_index, parse_config_args, set_args  # noqa: F401
