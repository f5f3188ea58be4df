"""`python -m prediction.predict --dataset … --config …` (reference README.md:41, prediction/predict.py)."""
from sd_video_gen_amd.predict import predict, main  # noqa: F401

if __name__ == "__main__":
    main()
