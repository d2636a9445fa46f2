"""`nnunetv2.nets.seg_mamba.mamba_simple` of the reference (/root/reference/nnunetv2/nets/seg_mamba/mamba_simple.py:37-357) -> native implementation in `nnuzoo_amd.nets.mamba_simple`."""
from nnuzoo_amd.nets.mamba_simple import Mamba  # noqa: F401

__all__ = ['Mamba']
