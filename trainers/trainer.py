"""`python -m trainers.trainer --dataset <ball|kitti> --config <name>` (reference README.md:34; trainers/trainer.py).
Re-export of the package's trainer: the training step itself runs in libsvg_hip.so."""
from sd_video_gen_amd.trainer import Adam, Criterion, Trainer, main, make_loaders  # noqa: F401

if __name__ == "__main__":
    main()
