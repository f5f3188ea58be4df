"""Flag + YAML surface of the sampling path.

Mirrors the reference's ``utils/config.py:8-49``: the same flags, the same
``type=bool`` quirk (any non-empty string is True), YAML read from
``./config/<name>.yml`` relative to the cwd, and ``CONFIG_NAME`` attached.
``sys.argv`` and the cwd are part of the API there (constructors re-parse);
``parse_config_args(argv)`` additionally accepts an explicit argv so tests and
the bench need not mutate ``sys.argv``.
"""
import argparse
import os
from types import SimpleNamespace

import yaml

_CONFIG_DIRS = ["./config", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config")]


def load_config(config_name):
    for d in _CONFIG_DIRS:
        path = os.path.join(d, config_name + ".yml")
        if os.path.exists(path):
            break
    else:
        raise FileNotFoundError(os.path.join("./config", config_name + ".yml"))
    with open(path, "r") as stream:
        data = yaml.safe_load(stream)
    cfg = SimpleNamespace(**data)
    cfg.CONFIG_NAME = config_name
    return cfg


def remove_config_index(config_name):
    if config_name[-1].isdigit():
        config_name = config_name[:config_name.rfind("_")]
    return config_name


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--dataset", type=str, required=True)
    p.add_argument("--save_best", type=bool, default=False)
    p.add_argument("--folder", type=str, default=None)
    p.add_argument("--config", type=str, required=True)
    p.add_argument("--resume", type=bool, default=False)
    p.add_argument("--debug", type=bool, default=False)
    p.add_argument("--flip", type=bool, default=False)
    p.add_argument("--pred_frames", type=int, default=1)
    p.add_argument("--show", type=bool, default=False)
    p.add_argument("--old_name", type=str, default="old_name_default")
    p.add_argument("--fullscreen", type=bool, default=False)
    p.add_argument("--save_output", type=bool, default=False)
    p.add_argument("--index", type=int, default=0)
    p.add_argument("--denoise", type=bool, default=False)
    p.add_argument("--mode", type=str, default="")
    p.add_argument("--denoise_start_step", type=int, default=40)
    return p


_OVERRIDE = None


def set_args(argv):
    """Pin the argv every later ``parse_config_args()`` sees (None = sys.argv)."""
    global _OVERRIDE
    _OVERRIDE = list(argv) if argv is not None else None


def parse_config_args(argv=None):
    if argv is None:
        argv = _OVERRIDE
    args = build_parser().parse_args(argv)
    return load_config(args.config), args
