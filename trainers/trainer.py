"""`python -m trainers.trainer …` (reference README.md:34).  Training is outside this round's hot path
("next"): the command line parses exactly like the reference's and then stops."""
from sd_video_gen_amd.config import parse_config_args


def main():
    config, args = parse_config_args()
    raise NotImplementedError(
        "training (trainers/trainer.py:111-190 of the reference) is not part of the MI355X sampling path yet; "
        "parsed --config %s --dataset %s. Use `python -m prediction.predict` for sampling." % (args.config, args.dataset))


if __name__ == "__main__":
    main()
