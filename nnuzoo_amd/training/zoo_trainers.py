"""Trainer plugins of the X^2-Net zoo models, same class names and hooks as the reference's plugin files:
  nnUNetTrainerM2Net / nnUNetTrainerM2NetP   /root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainerM2Net.py
  nnUNetTrainerSwT2Net                        /root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainerSwT2Net.py
Overrides, as there: build_network_architecture (legacy signature; the live one is accepted too, SURVEY.md §8b
quirk 2), AdamW(lr 1e-4, wd 5e-2, eps 1e-5) + CosineAnnealingLR(eta_min 1e-6), the fixed 7-entry deep-supervision
scale list, `network.deep_supervision` toggle, and - for the Swin model - the fp32 train_step without autocast or
GradScaler (nnUNetTrainerSwT2Net.py:112-130).  The M2Net plugins inherit the autocast step like the reference.
"""
from __future__ import annotations

import contextlib
import os

import numpy as np

import torch

from ..token_linear import deferred_wgrads
from torch.optim import AdamW
from torch.optim.lr_scheduler import CosineAnnealingLR

from ..nets.m2net import get_m2net_from_plans, get_m2netp_from_plans
from ..nets.swt2net import get_swt2net_from_plans
from ..ddp import allreduce_gradients, prepare_autograd_network_for_ddp
from .nnUNetTrainer import nnUNetTrainer, _num_input_channels

_X2_SCALES = [[1.0, 1.0], [1.0, 1.0], [0.5, 0.5], [0.25, 0.25], [0.125, 0.125], [0.0625, 0.0625], [0.03125, 0.03125]]


def _legacy_or_live(factory, args, kwargs):
    """(plans_manager, dataset_json, configuration_manager, num_input_channels, enable_deep_supervision=True) or the
    live (architecture_class_name, arch_init_kwargs, req_import, num_input_channels, num_output_channels, ds)."""
    if args and isinstance(args[0], str):
        num_in, num_out = args[3], args[4]
        ds = args[5] if len(args) > 5 else kwargs.get("enable_deep_supervision", True)
        dataset_json = {"labels": {str(i): i for i in range(num_out)}}
        return factory(None, dataset_json, kwargs.get("configuration_manager"), num_in, deep_supervision=ds)
    names = ["plans_manager", "dataset_json", "configuration_manager", "num_input_channels", "enable_deep_supervision"]
    a = dict(zip(names, args))
    a.update(kwargs)
    return factory(a.get("plans_manager"), a["dataset_json"], a.get("configuration_manager"), a["num_input_channels"],
                   deep_supervision=a.get("enable_deep_supervision", True))


class _X2Trainer(nnUNetTrainer):
    _factory = None
    _fp32_step = False
    # True for the nets whose small-channel fp32 conv blocks fault inside MIOpen's immediate-mode backward on this stack
    # (see nnUNetTrainerMambaND2Net): their steps run with the library path disabled for the DURATION of the step only
    # (`torch.backends.cudnn.flags`), never as a process-wide switch - other networks, predictors and tests in the same
    # process keep MIOpen.
    _no_miopen = False
    _fp32_validation = False

    def _library_scope(self):
        if self.device.type != 'cuda':
            return contextlib.nullcontext()
        # NNZ_LIBRARY_DETERMINISTIC=1: the convolutions that stay on the library (stage heads, 1-channel stems, small-channel
        # blocks) are restricted to its deterministic solvers - tools/probes/zoo_module_determinism.py found them to be the only
        # modules of the SwT2Net step whose forward / backward is not bit-reproducible from call to call
        det = os.environ.get("NNZ_LIBRARY_DETERMINISTIC", "0")
        if self._no_miopen or det == "2":           # 2: ATen's own kernels instead of the library (also reproducible)
            return torch.backends.cudnn.flags(enabled=False)
        if det == "1":                              # 1: the library's deterministic solvers (measured: 11x slower steps)
            return torch.backends.cudnn.flags(enabled=True, benchmark=False, deterministic=True)
        return contextlib.nullcontext()

    def __init__(self, plans: dict, configuration: str, fold: int, dataset_json: dict, unpack_dataset: bool = True,
                 device: torch.device = torch.device('cuda'), num_epochs: int = 250):
        super().__init__(plans, configuration, fold, dataset_json, unpack_dataset, device, num_epochs=num_epochs)
        self.initial_lr = 1e-4
        self.weight_decay = 5e-2
        self.freeze_encoder_epochs = -1
        self.early_stop_epoch = 25
        if self._fp32_step:
            self.grad_scaler = None
        # forward + loss + backward replayed as ONE hipGraph (training/graph_step.py): the zoo steps issue 5 000 - 10 000
        # small kernels and are host-bound in eager mode.  Captured memset nodes (ATen reductions, library paths) are
        # rewritten into kernel nodes before instantiation (csrc/graph_tools.hip), which is what made replay exact on
        # this stack.  NNZ_HIP_GRAPH=0 (or `use_hip_graph = False`) selects the eager step.
        self.use_hip_graph = self.device.type == 'cuda' and os.environ.get("NNZ_HIP_GRAPH", "1") != "0"
        self._graphed = None

    def initialize(self):
        if self.was_initialized:
            raise RuntimeError("You have called self.initialize even though the trainer was already initialized.")
        self._set_batch_size_and_oversample()
        self.num_input_channels = _num_input_channels(self.dataset_json)
        self.network = self.build_network_architecture(None, self.dataset_json, self.configuration_manager,
                                                       self.num_input_channels, self.enable_deep_supervision
                                                       ).to(self.device)
        self.optimizer, self.lr_scheduler = self.configure_optimizers()
        if self.is_ddp:
            # SyncBatchNorm + rank-0 parameters; gradients are averaged after backward (nnuzoo_amd/ddp.py) - parameters
            # without gradients (inner seg_layers) are skipped on every rank alike
            self.network = prepare_autograd_network_for_ddp(self.network)
            self.optimizer, self.lr_scheduler = self.configure_optimizers()
        self.loss = self._build_loss()
        # fp16-autocast steps: one multi-tensor cast of the plain torch convolutions' parameters per step instead of one cast
        # launch per parameter and direction (nnuzoo_amd/param_shadow.py)
        from ..param_shadow import ParamShadow
        self._forward = ParamShadow(self.network) if (self.device.type == 'cuda' and not self._fp32_step) else self.network
        self.was_initialized = True

    def _get_deep_supervision_scales(self):
        if not self.enable_deep_supervision:
            return None
        nd = len(self.configuration_manager.patch_size)   # the N-D plugins (SSND2Net) use the same 7 scales per axis
        return [[s[0]] * nd for s in _X2_SCALES]

    def configure_optimizers(self):
        # same optimizer and hyper-parameters as the reference's plugins (nnUNetTrainerM2Net.py:58-65); on the GPU the
        # update runs as torch's fused multi-tensor kernel (one launch per dtype group, step counters on the device)
        # instead of the foreach path, whose ~3 000 host-side `.item()` / dispatch calls per step for 1 526 parameter
        # tensors cost more host time than the whole backward (tools/profile_ops.py)
        fused = self.device.type == 'cuda' and os.environ.get("NNZ_FUSED_ADAMW", "1") != "0"
        if fused and os.environ.get("NNZ_HIP_ADAMW", "1") != "0":
            # round 4: an AdamW (same class hierarchy, state under torch's names) whose unscale + clip + step tail is two HIP
            # launches over a device chunk table instead of ~260 multi-tensor launches (training/fused_adamw.py)
            from .fused_adamw import FusedAdamW
            optimizer = FusedAdamW(self.network.parameters(), lr=self.initial_lr, weight_decay=self.weight_decay, eps=1e-5,
                                   betas=(0.9, 0.999))
        else:
            optimizer = AdamW(self.network.parameters(), lr=self.initial_lr, weight_decay=self.weight_decay, eps=1e-5,
                              betas=(0.9, 0.999), fused=fused)
        return optimizer, CosineAnnealingLR(optimizer, T_max=self.num_epochs, eta_min=1e-6)

    def _optimizer_tail(self, grads_token=None):
        """unscale_ -> clip_grad_norm_(12) -> step -> scaler update (nnUNetTrainer.py:1131-1139); the fused HIP tail when the
        optimizer offers it, torch's sequence otherwise"""
        from .nnUNetTrainer import _scaler_internals_ok
        opt, sc = self.optimizer, self.grad_scaler
        hip = hasattr(opt, "fused_step") and opt.fused_available() and (sc is None or _scaler_internals_ok(sc))
        if hip:
            if sc is not None:
                inv_scale = sc._scale.double().reciprocal().float()
                found_inf = opt.fused_step(inv_scale, 12, grads_token)
                torch._amp_update_scale_(sc._scale, sc._growth_tracker, found_inf, sc._growth_factor, sc._backoff_factor,
                                         sc._growth_interval)
            else:
                opt.fused_step(None, 12, grads_token)
            return
        if sc is not None:
            sc.unscale_(opt)
            torch.nn.utils.clip_grad_norm_(self.network.parameters(), 12)
            sc.step(opt)
            sc.update()
        else:
            torch.nn.utils.clip_grad_norm_(self.network.parameters(), 12)
            opt.step()

    def set_deep_supervision_enabled(self, enabled: bool):
        self.network.deep_supervision = enabled

    def train_step(self, batch: dict) -> dict:
        with self._library_scope():
            return self._train_step(batch)

    def _autocast_context(self):
        # a plugin whose reference class overrides validation_step to run WITHOUT autocast (nnUNetTrainerLightMUNet.py) validates
        # in fp32 here as well; the other fp32-step plugins inherit the base trainer's autocast validation there, and here
        if self._fp32_validation:
            return contextlib.nullcontext()
        return super()._autocast_context()

    def validation_step(self, batch: dict) -> dict:
        with self._library_scope():
            return super().validation_step(batch)

    def _train_step(self, batch: dict) -> dict:
        data = batch['data'].to(self.device, non_blocking=True)
        target = [i.to(self.device, non_blocking=True) for i in batch['target']] \
            if isinstance(batch['target'], list) else batch['target'].to(self.device, non_blocking=True)
        if self.use_hip_graph:
            from .graph_step import GraphedForwardBackward
            if self._graphed is None:
                self._graphed = GraphedForwardBackward(self.network, self.loss, self.grad_scaler,
                                                       autocast=not self._fp32_step, forward_fn=self._forward)
            tl = target if isinstance(target, list) else [target]
            l = self._graphed(data, tl)
            if self.is_ddp:
                allreduce_gradients(self.network.parameters())
            self._optimizer_tail(self._graphed.grads_token())
            return {'loss': l.detach().cpu().numpy()}
        self.optimizer.zero_grad(set_to_none=True)
        if self._fp32_step:
            output = self.network(data)
            l = self.loss(list(output) if isinstance(output, (tuple, list)) else output, target)
            with deferred_wgrads():  # fp32 Linear weight gradients of the pass: one grouped launch (token_linear.py)
                l.backward()
            if self.is_ddp:
                allreduce_gradients(self.network.parameters())
            self._optimizer_tail()
        else:
            with torch.autocast('cuda'):
                output = self._forward(data)
                l = self.loss(list(output) if isinstance(output, (tuple, list)) else output, target)
            with deferred_wgrads():  # the same grouped weight-gradient launches as the captured step (training/graph_step.py)
                self.grad_scaler.scale(l).backward()
            if self.is_ddp:
                allreduce_gradients(self.network.parameters())    # scaled gradients, like torch DDP under AMP
            self._optimizer_tail()
        return {'loss': l.detach().cpu().numpy()}


class nnUNetTrainerM2Net(_X2Trainer):
    @staticmethod
    def build_network_architecture(*args, **kwargs):
        return _legacy_or_live(get_m2net_from_plans, args, kwargs)


class nnUNetTrainerM2NetP(_X2Trainer):
    @staticmethod
    def build_network_architecture(*args, **kwargs):
        return _legacy_or_live(get_m2netp_from_plans, args, kwargs)


class nnUNetTrainerSwT2Net(_X2Trainer):
    _fp32_step = True

    @staticmethod
    def build_network_architecture(*args, **kwargs):
        return _legacy_or_live(get_swt2net_from_plans, args, kwargs)


class nnUNetTrainerSSND2Net(_X2Trainer):
    """reference: training/nnUNetTrainer/nnUNetTrainerSSND2Net.py:18-118 (2-D and 3-D, AdamW 1e-4 / wd 5e-2, cosine)"""

    @staticmethod
    def build_network_architecture(*args, **kwargs):
        from ..nets.ssnd2net import get_ssnd2net_from_plans
        return _legacy_or_live(lambda *a, **k: get_ssnd2net_from_plans(*a, small_mode=False, **k), args, kwargs)


class nnUNetTrainerSSND2NetP(nnUNetTrainerSSND2Net):
    """reference: nnUNetTrainerSSND2Net.py:121-142 (small_mode=True)"""

    @staticmethod
    def build_network_architecture(*args, **kwargs):
        from ..nets.ssnd2net import get_ssnd2net_from_plans
        return _legacy_or_live(lambda *a, **k: get_ssnd2net_from_plans(*a, small_mode=True, **k), args, kwargs)


class nnUNetTrainerMambaND2Net(_X2Trainer):
    """reference: training/nnUNetTrainer/nnUNetTrainerMambaND2Net.py:15-131 (N-D; fp32 step without autocast / GradScaler
    :111-131; AdamW 1e-4 / wd 5e-2, cosine; the 7-entry deep-supervision scale list per axis)"""
    _fp32_step = True
    # The UNETR-style blocks of this net are fp32 convolutions with 4..128 channels (1x1, 3x3, k = s transposed).
    # On this stack (ROCm 7.2 MIOpen through PyTorch's immediate mode) their backward faults with an out-of-bounds
    # access inside the full network (each block alone passes; MIOpen logs "workspace required ... provided ..." for
    # the solver it then runs anyway) - tools/probes/mambaND_stage_probe.py.  ATen's native convolution path is
    # exact and, at these channel counts, not slower, so the library path is off INSIDE this trainer's steps.
    _no_miopen = True

    @staticmethod
    def build_network_architecture(*args, **kwargs):
        from ..nets.mamba_nd2net import get_mamband2net_from_plans
        return _legacy_or_live(lambda *a, **k: get_mamband2net_from_plans(*a, small_mode=False, **k), args, kwargs)


class nnUNetTrainerMambaND2NetP(nnUNetTrainerMambaND2Net):
    """reference :134-156: small_mode=True, for which the reference's factory raises NotImplementedError (:1926)"""

    @staticmethod
    def build_network_architecture(*args, **kwargs):
        from ..nets.mamba_nd2net import get_mamband2net_from_plans
        return _legacy_or_live(lambda *a, **k: get_mamband2net_from_plans(*a, small_mode=True, **k), args, kwargs)


class nnUNetTrainerUNETR2Net(_X2Trainer):
    """reference: training/nnUNetTrainer/nnUNetTrainerUNETR2Net.py:15-118 (inherits the base autocast train_step;
    num_epochs 1000; AdamW 1e-4 / wd 5e-2, cosine; the 7-entry deep-supervision scale list per axis)"""

    def __init__(self, plans: dict, configuration: str, fold: int, dataset_json: dict, unpack_dataset: bool = True,
                 device: torch.device = torch.device('cuda'), num_epochs: int = 1000):
        super().__init__(plans, configuration, fold, dataset_json, unpack_dataset, device, num_epochs=num_epochs)

    _no_miopen = True                          # same small-channel UNETR conv blocks as MambaND2Net (see there)

    @staticmethod
    def build_network_architecture(*args, **kwargs):
        from ..nets.unetr2net import get_unetr2net_from_plans
        return _legacy_or_live(lambda *a, **k: get_unetr2net_from_plans(*a, small_mode=False, **k), args, kwargs)


class nnUNetTrainerUNETR(_X2Trainer):
    """reference: training/nnUNetTrainer/nnUNetTrainerUNETR.py:13-150 (monai's UNETR, 2-D or 3-D; patch size rounded up to
    multiples of the ViT's 16-voxel patches :18-28; fp32 step without autocast / GradScaler, clip 12 :61-77; ONE output - deep
    supervision off; AdamW 1e-4 / wd 0.01 / eps 1e-5 with PolyLR exponent 1.0 :138-141)"""
    _fp32_step = True
    _fp32_validation = True                    # the class overrides validation_step without autocast (:79-136)
    _no_miopen = True                          # same small-channel UNETR conv blocks as MambaND2Net (see there)

    def __init__(self, plans: dict, configuration: str, fold: int, dataset_json: dict, unpack_dataset: bool = True,
                 device: torch.device = torch.device('cuda'), num_epochs: int = 250):
        super().__init__(plans, configuration, fold, dataset_json, unpack_dataset, device, num_epochs=num_epochs)
        self.enable_deep_supervision = False
        old = list(self.configuration_manager.patch_size)
        new = [round(v / 16 + 0.5) * 16 if (v / 16) < 1 or (v / 16) % 1 != 0 else v for v in old]
        self.configuration_manager.configuration['patch_size'] = new
        self.plans['configurations'][self.configuration_name]['patch_size'] = new
        self.initial_lr = 1e-4
        self.grad_scaler = None
        self.weight_decay = 0.01

    def _get_deep_supervision_scales(self):
        return None

    def set_deep_supervision_enabled(self, enabled: bool):
        pass

    def configure_optimizers(self):
        from .lr_scheduler import PolyLRScheduler
        fused = self.device.type == 'cuda' and os.environ.get("NNZ_FUSED_ADAMW", "1") != "0"
        optimizer = AdamW(self.network.parameters(), lr=self.initial_lr, weight_decay=self.weight_decay, eps=1e-5, fused=fused)
        return optimizer, PolyLRScheduler(optimizer, self.initial_lr, self.num_epochs, exponent=1.0)

    @staticmethod
    def build_network_architecture(*args, **kwargs):
        from ..nets.unetr2net import MonaiUNETR
        num_in, num_out, _ = _live_num_in_out(args, kwargs)
        cm = next((a for a in list(args) + list(kwargs.values()) if hasattr(a, "patch_size")), None)
        if cm is None:
            raise ValueError("nnUNetTrainerUNETR.build_network_architecture needs the configuration manager (patch size)")
        return MonaiUNETR(in_channels=num_in, out_channels=num_out, img_size=list(cm.patch_size), feature_size=16,
                          hidden_size=768, mlp_dim=3072, num_heads=12, proj_type="conv", norm_name="instance", res_block=True,
                          dropout_rate=0.0, spatial_dims=len(cm.patch_size), qkv_bias=False, save_attn=False)


class nnUNetTrainerLightMamba2Net(_X2Trainer):
    """reference: training/nnUNetTrainer/nnUNetTrainerLightMamba2Net.py:18-131 (N-D; fp32 step without autocast /
    GradScaler :30-48; AdamW 1e-4 / wd 5e-2, cosine; deep-supervision scales from get_scales(min_size=8) :70-94)"""
    _fp32_step = True
    _small_model = False

    def __init__(self, plans: dict, configuration: str, fold: int, dataset_json: dict, unpack_dataset: bool = True,
                 device: torch.device = torch.device('cuda'), num_epochs: int = 250):
        super().__init__(plans, configuration, fold, dataset_json, unpack_dataset, device, num_epochs=num_epochs)
        self.early_stop_epoch = 350

    _no_miopen = True   # fp32 convolutions with 16..256 channels, depthwise 3x3 and 1x1 (see MambaND2Net above)

    def _get_deep_supervision_scales(self):
        if not self.enable_deep_supervision:
            return None
        from ..nets.ssnd2net import get_scales
        ps = self.configuration_manager.patch_size
        cum, out = np.ones(len(ps)), [[1.0] * len(ps), [1.0] * len(ps)]
        for s in get_scales(len(ps), ps, n_layers=5, patch_size=None, min_size=8):
            cum = cum / np.array(s)
            out.append([float(v) for v in cum])
        return out

    @classmethod
    def _build(cls, args, kwargs, small):
        from ..nets.light_mamba2net import get_light_mamba2net_from_plans
        return _legacy_or_live(lambda *a, **k: get_light_mamba2net_from_plans(*a, small_model=small, **k), args, kwargs)

    @staticmethod
    def build_network_architecture(*args, **kwargs):
        return nnUNetTrainerLightMamba2Net._build(args, kwargs, False)


class nnUNetTrainerLightMamba2NetP(nnUNetTrainerLightMamba2Net):
    """reference :134-156: the small model (LightMamba2NetP)"""

    @staticmethod
    def build_network_architecture(*args, **kwargs):
        return nnUNetTrainerLightMamba2Net._build(args, kwargs, True)


class nnUNetTrainerLM2Net(nnUNetTrainerLightMamba2Net):
    """reference: training/nnUNetTrainer/nnUNetTrainerLM2Net.py:16-133 (LM2Net, 2-D; the base trainer's fp16-autocast
    train_step - the class does not override it; AdamW 1e-4 / wd 5e-2 / eps 1e-5, cosine to 1e-6; deep-supervision scales
    from get_scales(n_layers=5, min_size=8) :53-74, the same rule as LightMamba2Net's)"""
    _fp32_step = False

    def __init__(self, plans: dict, configuration: str, fold: int, dataset_json: dict, unpack_dataset: bool = True,
                 device: torch.device = torch.device('cuda'), num_epochs: int = 250):
        super().__init__(plans, configuration, fold, dataset_json, unpack_dataset, device, num_epochs=num_epochs)
        self.early_stop_epoch = 25

    @classmethod
    def _build(cls, args, kwargs, small):
        from ..nets.lm2net import get_lm2net_from_plans
        return _legacy_or_live(lambda *a, **k: get_lm2net_from_plans(*a, small=small, **k), args, kwargs)

    @staticmethod
    def build_network_architecture(*args, **kwargs):
        return nnUNetTrainerLM2Net._build(args, kwargs, False)


class nnUNetTrainerLM2NetP(nnUNetTrainerLM2Net):
    """reference :136-158: the small model (LM2NetP)"""

    @staticmethod
    def build_network_architecture(*args, **kwargs):
        return nnUNetTrainerLM2Net._build(args, kwargs, True)


class nnUNetTrainerLightMUNet(_X2Trainer):
    """reference: training/nnUNetTrainer/nnUNetTrainerLightMUNet.py:14-129 (stand-alone LightMUNet, 2-D or 3-D; fp32 step
    without autocast / GradScaler :45-63; ONE output, deep supervision off :29, :127-128; Adam lr 1e-4 / wd 1e-5 / eps 1e-5,
    PolyLR exponent 0.9 :120-124; gradient clipping 12)"""
    _fp32_step = True
    _fp32_validation = True
    _no_miopen = True   # small-channel fp32 convolutions, depthwise 3x3 / 1x1 (see MambaND2Net above)

    def __init__(self, plans: dict, configuration: str, fold: int, dataset_json: dict, unpack_dataset: bool = True,
                 device: torch.device = torch.device('cuda'), num_epochs: int = 250):
        super().__init__(plans, configuration, fold, dataset_json, unpack_dataset, device, num_epochs=num_epochs)
        self.grad_scaler = None
        self.initial_lr = 1e-4
        self.weight_decay = 1e-5
        self.enable_deep_supervision = False

    def _get_deep_supervision_scales(self):
        return None

    def set_deep_supervision_enabled(self, enabled: bool):
        pass

    def configure_optimizers(self):
        from torch.optim import Adam
        from .lr_scheduler import PolyLRScheduler
        fused = self.device.type == 'cuda' and os.environ.get("NNZ_FUSED_ADAMW", "1") != "0"
        optimizer = Adam(self.network.parameters(), lr=self.initial_lr, weight_decay=self.weight_decay, eps=1e-5, fused=fused)
        return optimizer, PolyLRScheduler(optimizer, self.initial_lr, self.num_epochs, exponent=0.9)

    @staticmethod
    def build_network_architecture(*args, **kwargs):
        from ..nets.lightmunet import LightMUNet
        num_in, num_out, _ = _live_num_in_out(args, kwargs)
        cm = next((a for a in list(args) + list(kwargs.values()) if hasattr(a, "patch_size")), None)
        if cm is None:
            raise ValueError("nnUNetTrainerLightMUNet.build_network_architecture needs the configuration manager (patch size "
                             "-> spatial dims)")
        return LightMUNet(spatial_dims=len(cm.patch_size), init_filters=32, in_channels=num_in, out_channels=num_out,
                          blocks_down=[1, 2, 2, 4], blocks_up=[1, 1, 1])


class nnUNetTrainerLightSS2DMambaUNet(nnUNetTrainerLightMUNet):
    """reference: training/nnUNetTrainer/nnUNetTrainerLightSS2DMambaUNet.py:17-140 (LightSS2DMambaUNet, 2-D: its Mamba layers
    unpack (B, C, H, W); fp32 step without autocast / GradScaler, clip 12 :60-78; Adam 1e-4 / wd 1e-5 / eps 1e-5 + PolyLR 0.9
    :128-133; fp32 validation_step :80-126).  The network returns ONE tensor; the reference's class leaves
    `enable_deep_supervision` at the base default, with which its loss wrapper rejects that output - deep supervision is off
    here, as in nnUNetTrainerLightMUNet."""

    @staticmethod
    def build_network_architecture(*args, **kwargs):
        from ..nets.light_ss2d_mamba_unet import LightSS2DMambaUNet
        num_in, num_out, _ = _live_num_in_out(args, kwargs)
        cm = next((a for a in list(args) + list(kwargs.values()) if hasattr(a, "patch_size")), None)
        if cm is None:
            raise ValueError("nnUNetTrainerLightSS2DMambaUNet.build_network_architecture needs the configuration manager "
                             "(patch size -> spatial dims)")
        return LightSS2DMambaUNet(spatial_dims=len(cm.patch_size), in_channels=num_in, out_channels=num_out)


def _live_num_in_out(args, kwargs):
    """(num_input_channels, num_output_channels, deep_supervision) from the live calling convention
    (architecture_class_name, arch_init_kwargs, req_import, num_input_channels, num_output_channels, ds) or, for callers
    that still use the legacy one, from (plans_manager, dataset_json, configuration_manager, num_input_channels, ds)"""
    if args and isinstance(args[0], str):
        ds = args[5] if len(args) > 5 else kwargs.get("enable_deep_supervision", True)
        return args[3], args[4], ds
    names = ["plans_manager", "dataset_json", "configuration_manager", "num_input_channels", "enable_deep_supervision"]
    a = dict(zip(names, args))
    a.update(kwargs)
    from .nnUNetTrainer import _num_segmentation_heads
    return a["num_input_channels"], _num_segmentation_heads(a["dataset_json"]), a.get("enable_deep_supervision", True)


class nnUNetTrainerSegMamba(_X2Trainer):
    """reference: training/nnUNetTrainer/nnUNetTrainerSegMamba.py:15-110 (SegMamba, 2-D or 3-D; the base trainer's fp16-autocast
    train_step - the class does not override it; deep supervision OFF - one output; AdamW 1e-4 / wd 5e-2 / eps 1e-5, cosine to
    1e-6; `small_mode` of the factory raises there as well)"""
    _no_miopen = True   # small-channel convolutions around the Mamba layers (see MambaND2Net above)

    def __init__(self, plans: dict, configuration: str, fold: int, dataset_json: dict, unpack_dataset: bool = True,
                 device: torch.device = torch.device('cuda'), num_epochs: int = 250):
        super().__init__(plans, configuration, fold, dataset_json, unpack_dataset, device, num_epochs=num_epochs)
        self.enable_deep_supervision = False

    def _get_deep_supervision_scales(self):
        if not self.enable_deep_supervision:
            return None
        return super()._get_deep_supervision_scales()

    @staticmethod
    def build_network_architecture(*args, **kwargs):
        from ..nets.segmamba import get_seg_mamba_from_plans
        return _legacy_or_live(lambda *a, **k: get_seg_mamba_from_plans(*a, use_pretrain=False, small_mode=False, **k),
                               args, kwargs)


class _FrozenEncoderEpochs:
    """`on_train_epoch_start` of the Swin-UMamba plugins (nnUNetTrainerSwinUMamba.py:79-94): the VSSM encoder (except its patch
    embedding) is frozen for the first `freeze_encoder_epochs` epochs.  A change of the trainable set invalidates the captured
    step (the graph holds the autograd graph of the frozen configuration) and the optimizer's chunk table."""

    def on_train_epoch_start(self):
        freeze = self.current_epoch < self.freeze_encoder_epochs
        net = self.network.module if hasattr(self.network, "module") else self.network
        before = [p.requires_grad for p in net.parameters()]
        if freeze:
            net.freeze_encoder()
        else:
            net.unfreeze_encoder()
        if before != [p.requires_grad for p in net.parameters()]:
            self._graphed = None
            if hasattr(getattr(self, "optimizer", None), "_token"):
                self.optimizer._token = None      # the chunk table of the old trainable set must not survive (FusedAdamW)
            for p in net.parameters():
                if not p.requires_grad:
                    p.grad = None
        parent = getattr(super(), "on_train_epoch_start", None)   # the epoch loop itself is outside the hot path
        if parent is not None:
            parent()


class nnUNetTrainerSwinUMamba(_FrozenEncoderEpochs, _X2Trainer):
    """reference: training/nnUNetTrainer/nnUNetTrainerSwinUMamba.py:18-113 (Swin-UMamba, 2-D; the base trainer's fp16-autocast
    train_step; AdamW 1e-4 / wd 5e-2 / eps 1e-5, cosine to 1e-6; encoder frozen for 10 epochs; four deep-supervision outputs at
    1, 1/2, 1/4, 1/8; built with use_pretrain=False as there)"""
    _no_miopen = True

    def __init__(self, plans: dict, configuration: str, fold: int, dataset_json: dict, unpack_dataset: bool = True,
                 device: torch.device = torch.device('cuda'), num_epochs: int = 250, **kwargs):
        super().__init__(plans, configuration, fold, dataset_json, unpack_dataset, device, num_epochs=num_epochs)
        self.freeze_encoder_epochs = 10
        self.early_stop_epoch = 350

    def _get_deep_supervision_scales(self):
        return [[1.0, 1.0], [0.5, 0.5], [0.25, 0.25], [0.125, 0.125]] if self.enable_deep_supervision else None

    @staticmethod
    def build_network_architecture(*args, **kwargs):
        from ..nets.swin_umamba import get_swin_umamba_from_plans
        num_in, num_out, ds = _live_num_in_out(args, kwargs)
        return get_swin_umamba_from_plans(num_out, num_in, deep_supervision=ds, use_pretrain=False)


class nnUNetTrainerSwinUMambaD(_FrozenEncoderEpochs, _X2Trainer):
    """reference: training/nnUNetTrainer/nnUNetTrainerSwinUMambaD.py:17-124 (Swin-UMamba-D: Mamba decoder; same optimizer and
    freezing schedule; deep-supervision outputs at 1, 1/4, 1/8, 1/16 - the decoder's last stage expands by 4).  `load_checkpoint`
    of a path containing "vmamba" loads VMamba ImageNet weights into the encoder (:30-57), anything else is a trainer
    checkpoint."""
    _no_miopen = True

    def __init__(self, plans: dict, configuration: str, fold: int, dataset_json: dict, unpack_dataset: bool = True,
                 device: torch.device = torch.device('cuda'), num_epochs: int = 250):
        super().__init__(plans, configuration, fold, dataset_json, unpack_dataset, device, num_epochs=num_epochs)
        self.freeze_encoder_epochs = 10

    def _get_deep_supervision_scales(self):
        return [[1.0, 1.0], [0.25, 0.25], [0.125, 0.125], [0.0625, 0.0625]] if self.enable_deep_supervision else None

    def load_checkpoint(self, filename_or_checkpoint) -> None:
        if isinstance(filename_or_checkpoint, str) and "vmamba" in filename_or_checkpoint:
            from ..nets.swin_umamba import load_pretrained_ckpt
            net = self.network.module if hasattr(self.network, "module") else self.network
            load_pretrained_ckpt(net, filename_or_checkpoint, num_input_channels=next(net.parameters()).shape[1])
        else:
            super().load_checkpoint(filename_or_checkpoint)

    @staticmethod
    def build_network_architecture(*args, **kwargs):
        from ..nets.swin_umamba import get_swin_umamba_d_from_plans
        return _legacy_or_live(lambda *a, **k: get_swin_umamba_d_from_plans(*a, use_pretrain=True, **k), args, kwargs)


class nnUNetTrainerU2Net(_X2Trainer):
    """reference: training/nnUNetTrainer/nnUNetTrainerU2Net.py:14-99 (U2NET; the base trainer's autocast train_step; AdamW
    1e-4 / wd 5e-2 / eps 1e-5, cosine to 1e-6; seven deep-supervision outputs, ALL at full resolution: scales [[1, 1]] * 7)"""

    def __init__(self, plans: dict, configuration: str, fold: int, dataset_json: dict, unpack_dataset: bool = True,
                 device: torch.device = torch.device('cuda'), num_epochs: int = 250, **kwargs):
        super().__init__(plans, configuration, fold, dataset_json, unpack_dataset, device, num_epochs=num_epochs)
        self.early_stop_epoch = 10

    def _get_deep_supervision_scales(self):
        return [[1.0, 1.0]] * 7 if self.enable_deep_supervision else None

    @staticmethod
    def build_network_architecture(*args, **kwargs):
        from ..nets.u2net import get_u2net_from_plans
        num_in, num_out, ds = _live_num_in_out(args, kwargs)
        return get_u2net_from_plans(num_out, num_in, deep_supervision=ds, use_pretrain=False)


class nnUNetTrainerU2NetP(nnUNetTrainerU2Net):
    """reference :102-124 (U2NETP)"""

    @staticmethod
    def build_network_architecture(*args, **kwargs):
        from ..nets.u2net import get_u2netp_from_plans
        num_in, num_out, ds = _live_num_in_out(args, kwargs)
        return get_u2netp_from_plans(num_out, num_in, deep_supervision=ds, use_pretrain=False)


class nnUNetTrainerU2NetMulti(_X2Trainer):
    """reference: training/nnUNetTrainer/nnUNetTrainerU2NetMulti.py:14-103 (the N-D U^2-Net of nets/u2net_multi.py on monai's
    Convolution unit; the base trainer's autocast train_step; AdamW 1e-4 / wd 5e-2 / eps 1e-5, cosine to 1e-6; seven
    deep-supervision outputs, ALL at full resolution: [[1.0] * dims] * 7, :48-56).  The reference's build_network_architecture
    passes (plans_manager, dataset_json, configuration_manager, num_input_channels, deep_supervision=...) to a factory declared as
    (spatial_dims, num_segmentation_heads, num_input_channels, deep_supervision, ...) - a TypeError as written (:37-44 vs
    u2net_multi.py:699-705); here the factory receives what it is declared with."""

    def __init__(self, plans: dict, configuration: str, fold: int, dataset_json: dict, unpack_dataset: bool = True,
                 device: torch.device = torch.device('cuda'), num_epochs: int = 250):
        super().__init__(plans, configuration, fold, dataset_json, unpack_dataset, device, num_epochs=num_epochs)
        self.early_stop_epoch = 10
        self.spatial_dims = len(self.configuration_manager.patch_size)

    def _get_deep_supervision_scales(self):
        return [[1.0] * self.spatial_dims] * 7 if self.enable_deep_supervision else None

    @staticmethod
    def _dims(args, kwargs) -> int:
        cm = kwargs.get("configuration_manager")
        if cm is None and len(args) > 2 and not isinstance(args[0], str):
            cm = args[2]
        if cm is None:
            raise ValueError("nnUNetTrainerU2NetMulti.build_network_architecture needs the configuration_manager (patch size -> 2-D / 3-D)")
        return len(cm.patch_size)

    @staticmethod
    def build_network_architecture(*args, **kwargs):
        from ..nets.u2net_multi import get_u2net_from_plans
        num_in, num_out, ds = _live_num_in_out(args, kwargs)
        return get_u2net_from_plans(nnUNetTrainerU2NetMulti._dims(args, kwargs), num_out, num_in, deep_supervision=ds,
                                    use_pretrain=False)


class nnUNetTrainerU2NetMultiP(nnUNetTrainerU2NetMulti):
    """reference :106-194 (U2NETP of nets/u2net_multi.py through get_u2netp_from_plans, whose signature IS the plans-style one)"""

    @staticmethod
    def build_network_architecture(*args, **kwargs):
        from ..nets.u2net_multi import get_u2netp_from_plans
        return _legacy_or_live(get_u2netp_from_plans, args, kwargs)


class nnUNetTrainerSwUNETR(_X2Trainer):
    """reference: training/nnUNetTrainer/nnUNetTrainerSwUNETR.py:13-99 - monai's SwinUNETR (feature_size 48) behind the base
    trainer's autocast step, deep supervision off, AdamW 1e-4 / wd 5e-2 / eps 1e-5, cosine.  NOTHING of the network is defined in
    the reference: the class is `monai.networks.nets.SwinUNETR`, imported at module level (:4), and monai is absent from
    /root/reference and from this image.  The plugin therefore exists under its name with the reference's
    hyper-parameters and fails where the reference fails without monai - at the import - with the same exception type; with monai
    installed it builds monai's network exactly as the reference does (library kernels: no HIP path is claimed for it)."""

    def __init__(self, plans: dict, configuration: str, fold: int, dataset_json: dict, unpack_dataset: bool = True,
                 device: torch.device = torch.device('cuda'), num_epochs: int = 250, **kwargs):
        super().__init__(plans, configuration, fold, dataset_json, unpack_dataset, device, num_epochs=num_epochs)
        self.enable_deep_supervision = False

    def _get_deep_supervision_scales(self):
        return [[1.0, 1.0]] * 7 if self.enable_deep_supervision else None

    @staticmethod
    def build_network_architecture(*args, **kwargs):
        from monai.networks.nets import SwinUNETR      # ModuleNotFoundError without monai, like the reference's module import
        num_in, num_out, _ = _live_num_in_out(args, kwargs)
        cm = kwargs.get("configuration_manager")
        if cm is None and len(args) > 2 and not isinstance(args[0], str):
            cm = args[2]
        return SwinUNETR(img_size=cm.patch_size[0], in_channels=num_in, out_channels=num_out,
                         spatial_dims=len(cm.patch_size), feature_size=48, drop_rate=0.0, attn_drop_rate=0.0)


class nnUNetTrainerSwinTransformerUnet(_X2Trainer):
    """reference: training/nnUNetTrainer/nnUNetTrainerSwinTransformerUnet.py:17-108 (the single Swin U-net of nets/swt.py; deep
    supervision OFF - one output -, AdamW 1e-4 / wd 5e-2, cosine; the base trainer's autocast train_step).  The reference's
    build_network_architecture passes (plans_manager, dataset_json, configuration_manager, ...) to a factory that takes
    (num_segmentation_heads, num_input_channels, ...) - it cannot run as written (:36-43 vs swt.py:505-510); here the factory
    receives the counts it is declared with."""

    def __init__(self, plans: dict, configuration: str, fold: int, dataset_json: dict, unpack_dataset: bool = True,
                 device: torch.device = torch.device('cuda'), num_epochs: int = 250):
        super().__init__(plans, configuration, fold, dataset_json, unpack_dataset, device, num_epochs=num_epochs)
        self.enable_deep_supervision = False
        self.early_stop_epoch = 10

    def _get_deep_supervision_scales(self):
        return [[1.0, 1.0]] * 7 if self.enable_deep_supervision else None

    def set_deep_supervision_enabled(self, enabled: bool):
        pass                                   # single-output network

    @staticmethod
    def build_network_architecture(*args, **kwargs):
        from ..nets.swt import get_swin_transformer_unet
        num_in, num_out, ds = _live_num_in_out(args, kwargs)
        return get_swin_transformer_unet(num_out, num_in, deep_supervision=ds, use_pretrain=False)
