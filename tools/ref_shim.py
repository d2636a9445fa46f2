"""Import shim used ONLY in the build container (where /root/reference exists) to import the reference's pure-Python
model zoo without its third-party CUDA/medical-imaging dependencies, so that golden vectors can be generated from
the reference's own code.  Nothing here travels into the product or is needed on the GPU box.

Substitutions (arithmetic that is actually replaced):
  timm.layers.DropPath            -> standard stochastic depth (identity in eval())
  timm.layers.trunc_normal_       -> torch.nn.init.trunc_normal_
  monai Convolution(conv_only)    -> nn.Sequential with a child named `conv` = nn.Conv{2,3}d(same padding)
  monai UpSample(nontrainable)    -> nn.Upsample(size, mode, align_corners=False); interp_mode LINEAR -> bi/trilinear by dims
  mamba_ssm selective_scan_fn     -> the reference's own selective_scan_ref
                                     (nnunetv2/nets/seg_mamba/selective_scan_interface.py:86-152, extracted by ast)
  dynamic_network_architectures init_last_bn_before_add_to_0 -> no-op (no residual-BN blocks in these nets)
  batchgenerators ...file_and_folder_operations (star-imported) -> typing names + os.path helpers only
Everything else that is missing becomes a MagicMock module (never executed on the fixture paths).
"""
import ast
import enum
import importlib.abc
import importlib.machinery
import sys
import types
from unittest import mock

import torch
from torch import nn

REF = "/root/reference"
_MOCKED_ROOTS = ("timm", "monai", "mamba_ssm", "dynamic_network_architectures", "batchgenerators", "batchgeneratorsv2",
                 "acvl_utils", "causal_conv1d", "torchinfo", "deep_utils", "SimpleITK", "nibabel", "skimage", "numba",
                 "blosc2", "tifffile", "seaborn", "prettytable", "matplotlib", "imageio", "dicom2nifti", "graphviz",
                 "hiddenlayer", "IPython", "cv2", "nnunetv2.utilities.plans_handling.plans_handler")


class _MockFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in _MOCKED_ROOTS or name in _MOCKED_ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__path__ = []
        m.__name__ = spec.name
        m.__spec__ = spec
        m.__loader__ = self
        return m

    def exec_module(self, module):
        pass


class DropPath(nn.Module):
    def __init__(self, drop_prob: float = 0., scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        rt = x.new_empty(shape).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            rt.div_(keep)
        return x * rt


class Convolution(nn.Sequential):
    def __init__(self, spatial_dims, in_channels, out_channels, strides=1, kernel_size=3, bias=True, conv_only=False,
                 groups=1, dilation=1, padding=None, **kw):
        super().__init__()
        assert conv_only
        conv = nn.Conv2d if spatial_dims == 2 else nn.Conv3d
        ks = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        pad = (ks - 1) // 2 * dilation if padding is None else padding
        self.add_module("conv", conv(in_channels, out_channels, kernel_size, strides, pad, dilation, groups, bias))


class UpSample(nn.Module):
    def __init__(self, spatial_dims, size=None, mode="nontrainable", interp_mode="bilinear", align_corners=False, **kw):
        super().__init__()
        im = interp_mode.value if isinstance(interp_mode, enum.Enum) else interp_mode
        if im == "linear":
            im = {1: "linear", 2: "bilinear", 3: "trilinear"}[spatial_dims]
        self.up = nn.Upsample(size=size, mode=im, align_corners=align_corners)

    def forward(self, x):
        return self.up(x)


class UpsampleMode(str, enum.Enum):
    NONTRAINABLE = "nontrainable"
    DECONV = "deconv"


class InterpolateMode(str, enum.Enum):
    BILINEAR = "bilinear"
    TRILINEAR = "trilinear"
    NEAREST = "nearest"
    LINEAR = "linear"   # monai resolves "linear" by spatial_dims: bilinear (2-D) / trilinear (3-D)


def load_selective_scan_ref():
    src = open(f"{REF}/nnunetv2/nets/seg_mamba/selective_scan_interface.py").read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "selective_scan_ref"][0]
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {}
    import torch.nn.functional as F
    from einops import rearrange, repeat
    ns.update(torch=torch, F=F, rearrange=rearrange, repeat=repeat)
    exec(compile(mod, "selective_scan_ref(reference)", "exec"), ns)
    return ns["selective_scan_ref"]


def load_mamba_inner_ref():
    """The reference's mamba_inner_ref (selective_scan_interface.py:640-674), extracted by ast like selective_scan_ref
    (the module itself needs the selective_scan_cuda extension).  Its two free names are bound to reference code:
    selective_scan_fn := the reference's selective_scan_ref; causal_conv1d_fn := the fallback formula of the reference's
    own Mamba.forward (mamba_simple.py:318-321): act(conv1d(x, padding=W-1)[..., :L])."""
    src = open(f"{REF}/nnunetv2/nets/seg_mamba/selective_scan_interface.py").read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "mamba_inner_ref"][0]
    mod = ast.Module(body=[fn], type_ignores=[])
    import torch.nn.functional as F
    from einops import rearrange, repeat

    def causal_conv1d_fn(x, weight, bias, activation):
        assert activation in ("silu", "swish")
        W = weight.shape[-1]
        return F.silu(F.conv1d(x, weight.unsqueeze(1), bias, padding=W - 1, groups=x.shape[1])[..., :x.shape[-1]])

    ns = dict(torch=torch, F=F, rearrange=rearrange, repeat=repeat, causal_conv1d_fn=causal_conv1d_fn,
              selective_scan_fn=load_selective_scan_ref())
    exec(compile(mod, "mamba_inner_ref(reference)", "exec"), ns)
    return ns["mamba_inner_ref"]


def install():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sys.meta_path.insert(0, _MockFinder())
    import timm.layers
    timm.layers.DropPath = DropPath
    timm.layers.trunc_normal_ = nn.init.trunc_normal_
    import timm.models.layers
    timm.models.layers.DropPath = DropPath
    timm.models.layers.trunc_normal_ = nn.init.trunc_normal_
    timm.models.layers.to_2tuple = lambda x: (x, x) if not isinstance(x, (tuple, list)) else tuple(x)
    import monai.networks.blocks
    monai.networks.blocks.Convolution = Convolution
    monai.networks.blocks.UpSample = UpSample
    import monai.networks.blocks.upsample
    monai.networks.blocks.upsample.UpSample = UpSample
    import monai.utils
    monai.utils.UpsampleMode = UpsampleMode
    monai.utils.InterpolateMode = InterpolateMode
    import dynamic_network_architectures.initialization.weight_init as wi
    wi.init_last_bn_before_add_to_0 = lambda m: None
    import mamba_ssm.ops.selective_scan_interface as ssi
    ssi.selective_scan_fn = load_selective_scan_ref()
    # `from batchgenerators.utilities.file_and_folder_operations import *` is how several reference modules obtain the
    # typing names and os.path helpers: give the mocked module a real namespace for the star import
    import os
    import typing
    import batchgenerators.utilities.file_and_folder_operations as ffo
    names = {n: getattr(typing, n) for n in ("List", "Tuple", "Union", "Optional", "Dict", "Callable", "Iterable")}
    names.update(join=os.path.join, isdir=os.path.isdir, isfile=os.path.isfile, os=os)
    for k, v in names.items():
        setattr(ffo, k, v)
    ffo.__all__ = list(names)
    return ssi.selective_scan_fn
