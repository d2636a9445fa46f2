"""`nnunetv2.nets.swt2net` of the reference (/root/reference/nnunetv2/nets/swt2net.py) -> native implementation in `nnuzoo_amd.nets.swt2net`."""
from nnuzoo_amd.nets.swt2net import REBNCONV, RSU4F, DropPath, PatchEmbedding, PatchMerging, PatchExpanding, FinalPatchExpanding, Mlp, WindowAttention, SwinTransformerBlock, BasicBlock, BasicBlockUp, SwinTransformerUnet, SwT2Net, get_swt2net_from_plans  # noqa: F401

__all__ = ['REBNCONV', 'RSU4F', 'DropPath', 'PatchEmbedding', 'PatchMerging', 'PatchExpanding', 'FinalPatchExpanding', 'Mlp', 'WindowAttention', 'SwinTransformerBlock', 'BasicBlock', 'BasicBlockUp', 'SwinTransformerUnet', 'SwT2Net', 'get_swt2net_from_plans']
