"""The sampling loop of the reference (prediction/predict.py:16-42 and :117-197), host side.

``predict(model, X)`` and the per-clip loop keep the reference's semantics, quirks included
(SURVEY §9): src == tgt, SOS only on the first iteration, window hard-coded to 5, last conditioning
frame dropped from the emitted clip.  The clip-batched loop runs C independent clips in lock step
(every GEMM gets C times the rows); each clip still sees PE row 0 exactly as at batch 1.
"""
import torch


def predict(model, input_sequence, pe_row=None):
    """prediction/predict.py:16-42 -> (D_lat,) for batch row 0 (or (B,D_lat) rows when ``pe_row`` is given)."""
    model.eval()
    with torch.no_grad():
        tgt_mask = model.get_tgt_mask(input_sequence.size(1)).to(input_sequence.device)
        pred = model(input_sequence, input_sequence, tgt_mask, pe_row=pe_row) if pe_row is not None \
            else model(input_sequence, input_sequence, tgt_mask)
        pred = pred.permute(1, 0, 2)                       # (B, T, D)
    if pe_row is not None:
        return pred[:, -1]
    return pred[0, -1]


def rollout_latents(model, new_batch, pred_frames, post=None):
    """predict.py:124-197 for one clip with the VAE taken out: ``new_batch`` (1,6,D) = SOS + 5 encoded frames.
    ``post(pred)`` is the optional denoise round trip (predict.py:145-185).  Returns (all_latents, trace)."""
    X = new_batch
    inputs = new_batch[:, 1:]
    preds = new_batch.new_zeros((1, 0, new_batch.shape[-1]))
    trace, all_latents = [], None
    for _ in range(pred_frames):
        shape_in = tuple(X.shape)
        pred = predict(model, X)
        if post is not None:
            pred = post(pred)
        preds = torch.cat((preds, pred.reshape(1, 1, -1)), dim=1)
        all_latents = torch.cat([inputs[:, :-1], preds], dim=1)      # predict.py:193
        X = all_latents[:, -5:]                                      # predict.py:196
        trace.append((shape_in, tuple(all_latents.shape)))
    return all_latents, trace
