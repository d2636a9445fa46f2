"""`dynamic_network_architectures.architectures.unet.PlainConvUNet` -> nnuzoo_amd.nets.plain_conv_unet.PlainConvUNet
(same constructor keywords as the planner emits, default_experiment_planner.py:285-305; same state_dict keys)."""
from nnuzoo_amd.nets.plain_conv_unet import PlainConvUNet  # noqa: F401

__all__ = ["PlainConvUNet"]
