"""`nnunetv2.nets.seg_mamba.selective_scan_interface` of the reference (/root/reference/nnunetv2/nets/seg_mamba/selective_scan_interface.py:14-152, 640-674) -> native implementation in `nnuzoo_amd.selective_scan`."""
from nnuzoo_amd.selective_scan import selective_scan_fn  # noqa: F401
from nnuzoo_amd.mamba_block import causal_conv1d_fn, mamba_inner_fn, mamba_inner_fn_no_out_proj  # noqa: F401

__all__ = ['selective_scan_fn']
