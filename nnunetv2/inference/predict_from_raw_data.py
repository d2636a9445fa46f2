"""`nnunetv2.inference.predict_from_raw_data` of the reference (/root/reference/nnunetv2/inference/predict_from_raw_data.py:37-692) -> native implementation in `nnuzoo_amd.inference.predict_from_raw_data`."""
from nnuzoo_amd.inference.predict_from_raw_data import nnUNetPredictor  # noqa: F401

__all__ = ['nnUNetPredictor']
