"""`prediction.predict_text` of the reference (prediction/predict_text.py) -> the MI355X-native mirror: `predict(model, X, cls_list)`
(:48-74), the class-name helpers (:18-46) and `main()` — the text-conditioned sampling loop with the FVD evaluation behind it
(:76-321): `python -m prediction.predict_text --dataset ucf-wallpushups --config 11_27_ucf_text_final --mode test --save_output True`."""
from sd_video_gen_amd.transformer_text import predict  # noqa: F401
from sd_video_gen_amd.predict_text import find_classes, main, splitClassNames  # noqa: F401

if __name__ == "__main__":
    main()
