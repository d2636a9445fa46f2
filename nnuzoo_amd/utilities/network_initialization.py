"""He initialisation used by every network factory of the reference.

Behaviour of /root/reference/nnunetv2/utilities/network_initialization.py:4-12: kaiming_normal_(a=neg_slope) on the
weights of (transposed) convolutions, zero biases; everything else keeps torch's default init.
"""
from torch import nn

_CONV_TYPES = (nn.Conv3d, nn.Conv2d, nn.ConvTranspose2d, nn.ConvTranspose3d)


class InitWeights_He:
    def __init__(self, neg_slope: float = 1e-2):
        self.neg_slope = neg_slope

    def __call__(self, module: nn.Module) -> None:
        if not isinstance(module, _CONV_TYPES):
            return
        nn.init.kaiming_normal_(module.weight, a=self.neg_slope)
        if module.bias is not None:
            nn.init.constant_(module.bias, 0)
