"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's sliding-window accumulation.

Follows /root/reference/nnunetv2/inference/predict_from_raw_data.py:
  _internal_maybe_mirror_and_predict :549-564 and _internal_predict_sliding_window_return_logits :566-643,
with tile geometry from nnunetv2/inference/sliding_window_prediction.py:10-58 (restated in
nnuzoo_amd/inference/sliding_window_prediction.py, host-side set-up code checked against the same fixtures).
Pinned by tests/golden/sliding_window.npz: outputs of the reference's own methods run on CPU half tensors around
tests/golden_util.toy_seg_network (tools/make_golden.py gen_sliding_window).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline may import this module; the product path (nnuzoo_amd/inference) never does.
"""
import itertools

import torch


def mirror_and_predict(network, x, mirror_axes):
    """x (1, C, *spatial); returns the mirror-averaged prediction with the reference's in-place half arithmetic"""
    prediction = network(x)
    if mirror_axes is not None:
        combos = [c for i in range(len(mirror_axes)) for c in itertools.combinations([m + 2 for m in mirror_axes], i + 1)]
        for axes in combos:
            prediction += torch.flip(network(torch.flip(x, (*axes,))), (*axes,))
        prediction /= (len(combos) + 1)
    return prediction


def predict_sliding_window(network, data, slicers, num_heads, gaussian, mirror_axes):
    """data (C, *image) on CPU; slicers as produced by the predictor; gaussian: half tensor or None.
    Returns half logits (num_heads, *image) - the reference's result before the padding is reverted."""
    predicted_logits = torch.zeros((num_heads, *data.shape[1:]), dtype=torch.half)
    n_predictions = torch.zeros(data.shape[1:], dtype=torch.half)
    for sl in slicers:
        workon = data[sl][None]
        prediction = mirror_and_predict(network, workon, mirror_axes)[0]
        if gaussian is not None:
            prediction *= gaussian
        predicted_logits[sl] += prediction
        n_predictions[sl[1:]] += gaussian if gaussian is not None else 1
    predicted_logits /= n_predictions
    if torch.any(torch.isinf(predicted_logits)):
        raise RuntimeError('Encountered inf in predicted array.')
    return predicted_logits
