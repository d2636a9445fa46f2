"""`nnunetv2.training.dataloading.data_loader` of the reference (/root/reference/nnunetv2/training/dataloading/data_loader.py:19-262) -> device-resident implementation in `nnuzoo_amd.dataloading.device_loader` (same constructor arguments and batch contract; the batch tensors are CUDA tensors)."""
from nnuzoo_amd.dataloading.device_loader import DeviceCaseStore, nnUNetDataLoader  # noqa: F401

__all__ = ['DeviceCaseStore', 'nnUNetDataLoader']
