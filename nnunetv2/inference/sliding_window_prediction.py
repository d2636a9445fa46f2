"""`nnunetv2.inference.sliding_window_prediction` of the reference (/root/reference/nnunetv2/inference/sliding_window_prediction.py:10-58) -> native implementation in `nnuzoo_amd.inference.sliding_window_prediction`."""
from nnuzoo_amd.inference.sliding_window_prediction import compute_gaussian, compute_steps_for_sliding_window  # noqa: F401

__all__ = ['compute_gaussian', 'compute_steps_for_sliding_window']
