"""`nnunetv2.nets.seg_mamba.segmamba` of the reference (/root/reference/nnunetv2/nets/seg_mamba/segmamba.py:27-411) -> native implementation in `nnuzoo_amd.nets.segmamba`."""
from nnuzoo_amd.nets.segmamba import GSC, InstanceNorm, LayerNorm, MambaEncoder, MambaLayer, MlpChannel, SegMamba, get_seg_mamba_from_plans  # noqa: F401

__all__ = ['GSC', 'InstanceNorm', 'LayerNorm', 'MambaEncoder', 'MambaLayer', 'MlpChannel', 'SegMamba', 'get_seg_mamba_from_plans']
