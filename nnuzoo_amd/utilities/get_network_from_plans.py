"""Network factory with the reference's signature and behaviour
(/root/reference/nnunetv2/utilities/get_network_from_plans.py:18-62): resolve the class named in plans.json,
instantiate it with (input_channels, num_classes, **arch_kwargs), then `network.apply(network.initialize)`.

The reference's default class, dynamic_network_architectures.architectures.unet.PlainConvUNet, is resolved to the
MI355X-native nnuzoo_amd.nets.plain_conv_unet.PlainConvUNet (same constructor, same state_dict keys), so existing
plans.json files work unchanged.
"""
import pydoc
from typing import Union

_NATIVE = {
    "dynamic_network_architectures.architectures.unet.PlainConvUNet": "nnuzoo_amd.nets.plain_conv_unet.PlainConvUNet",
    "PlainConvUNet": "nnuzoo_amd.nets.plain_conv_unet.PlainConvUNet",
}


def get_network_from_plans(arch_class_name, arch_kwargs, arch_kwargs_req_import, input_channels, output_channels,
                           allow_init=True, deep_supervision: Union[bool, None] = None,
                           up_sample_type: str = "convtranspose"):
    architecture_kwargs = dict(**arch_kwargs)
    for ri in arch_kwargs_req_import:
        if architecture_kwargs[ri] is not None and isinstance(architecture_kwargs[ri], str):
            architecture_kwargs[ri] = pydoc.locate(architecture_kwargs[ri])
    nw_class = pydoc.locate(_NATIVE.get(arch_class_name, arch_class_name))
    if nw_class is None:
        raise ImportError(f'Network class {arch_class_name} could not be found, please check/correct your plans file')
    if deep_supervision is not None:
        architecture_kwargs['deep_supervision'] = deep_supervision
    if up_sample_type != "convtranspose":
        raise NotImplementedError("only up_sample_type='convtranspose' has a HIP schedule")
    architecture_kwargs.pop("up_sample_type", None)  # same effect as the reference's retry (:51-57)
    network = nw_class(input_channels=input_channels, num_classes=output_channels, **architecture_kwargs)
    if hasattr(network, 'initialize') and allow_init:
        network.apply(network.initialize)
    return network
