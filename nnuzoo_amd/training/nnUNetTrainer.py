"""nnUNetTrainer - the hot-path subset of the reference's trainer plugin surface, MI355X-native.

Mirrors /root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainer.py for everything the per-patch
forward/backward step touches, with the same names, argument meaning and numerics:
  __init__(plans, configuration, fold, dataset_json, unpack_dataset, device, num_epochs, initial_lr, ...)   :80-239
  initialize()                                         :242-294   (network, optimizer, DDP, loss)
  build_network_architecture(...) (staticmethod, live AND legacy calling convention)   :360-399, SURVEY.md §8b
  _get_deep_supervision_scales()                       :401-408
  _build_loss()                                        :455-489
  configure_optimizers()                               :571-575   SGD(lr, wd 3e-5, momentum .99, nesterov) + PolyLR
  set_deep_supervision_enabled()                       :1010-1022
  train_step(batch) / validation_step(batch)           :1112-1144 / :1161-1226
  on_validation_epoch_end pseudo-Dice formula          :1255-1259
  save_checkpoint / load_checkpoint (same dict keys)   :1291-1352
Out of scope: data loading/augmentation, planning, sliding-window validation, logging/plots.
Data parallelism: one process per GPU; gradients are exchanged by nnuzoo_amd.ddp.BucketedAllReduce (RCCL over
xGMI, overlapped with the explicit backward schedule) instead of torch DDP's autograd hooks.
"""
from __future__ import annotations

import contextlib
import os
import inspect
from typing import List, Union

import numpy as np
import torch
import torch.distributed as dist

from .. import hip_ops as ops
from ..ddp import attach_bucketed_allreduce
from ..utilities.get_network_from_plans import get_network_from_plans
from .fused_sgd import FusedSGD
from .loss import DC_and_BCE_loss, DC_and_CE_loss, DeepSupervisionWrapper, MemoryEfficientSoftDiceLoss
from .lr_scheduler import PolyLRScheduler


class ConfigurationView:
    """The few ConfigurationManager properties the step needs (plans_handler.py:30-230)."""

    def __init__(self, cfg: dict):
        self.configuration = cfg

    @property
    def patch_size(self) -> List[int]:
        return self.configuration['patch_size']

    @property
    def batch_size(self) -> int:
        return self.configuration['batch_size']

    @property
    def batch_dice(self) -> bool:
        return self.configuration['batch_dice']

    @property
    def network_arch_class_name(self) -> str:
        return self.configuration['architecture']['network_class_name']

    @property
    def network_arch_init_kwargs(self) -> dict:
        return self.configuration['architecture']['arch_kwargs']

    @property
    def network_arch_init_kwargs_req_import(self):
        return self.configuration['architecture']['_kw_requires_import']

    @property
    def pool_op_kernel_sizes(self):
        return self.network_arch_init_kwargs['strides']


def _has_regions(dataset_json: dict) -> bool:
    """LabelManager.has_regions (label_handling.py:46-48): any label declared as a list / tuple of label values"""
    return any(isinstance(v, (list, tuple)) and len(v) > 1 for v in dataset_json['labels'].values())


def _num_segmentation_heads(dataset_json: dict) -> int:
    """LabelManager.num_segmentation_heads (label_handling.py:58-100): one head per label, or per foreground region"""
    labels = dataset_json['labels']
    if _has_regions(dataset_json):
        n = 0
        for k, r in labels.items():
            if k == 'ignore':
                continue
            vals = set(r) if isinstance(r, (list, tuple)) else {r}
            if vals == {0}:
                continue                      # regions that are background
            n += 1
        return n
    return len([k for k in labels if k != 'ignore'])


def _ignore_label(dataset_json: dict):
    """LabelManager.ignore_label (label_handling.py:102-127): the integer under the key 'ignore' (must be the highest)"""
    ig = dataset_json['labels'].get('ignore')
    if ig is not None:
        others = []
        for k, v in dataset_json['labels'].items():
            if k != 'ignore':
                others += [int(i) for i in v] if isinstance(v, (list, tuple)) else [int(v)]
        assert isinstance(ig, int) and ig == max(others) + 1, \
            'If you use the ignore label it must have the highest label value! It cannot be 0 or in between other labels.'
    return ig


def _num_input_channels(dataset_json: dict) -> int:
    names = dataset_json.get('channel_names', dataset_json.get('modality', {'0': 'x'}))
    return len(names)


def _scaler_internals_ok(scaler) -> bool:
    """the fused tail drives torch.amp.GradScaler's state directly (what GradScaler.update() does internally)"""
    return all(hasattr(scaler, a) for a in ("_scale", "_growth_tracker", "_growth_factor", "_backoff_factor",
                                             "_growth_interval")) and scaler._scale is not None


class nnUNetTrainer:
    def __init__(self, plans: dict, configuration: str, fold: Union[int, str], dataset_json: dict,
                 unpack_dataset: bool = True, device: torch.device = torch.device('cuda'), num_epochs: int = 1000,
                 initial_lr: float = 1e-2, up_sample_type: str = 'convtranspose', batch_size: int = None, **kwargs):
        plans["configurations"][configuration]["batch_size"] = batch_size or \
            plans["configurations"][configuration]["batch_size"]
        self.up_sample_type = up_sample_type
        self.is_ddp = dist.is_available() and dist.is_initialized()
        self.local_rank = 0 if not self.is_ddp else dist.get_rank()
        self.device = device
        if self.device.type == 'cuda':
            idx = int(torch.cuda.current_device()) if self.is_ddp else 0
            self.device = torch.device(type='cuda', index=idx)
        self.my_init_kwargs = {}
        for k in inspect.signature(self.__init__).parameters.keys():
            if k in locals():
                self.my_init_kwargs[k] = locals()[k]
        self.plans = plans
        self.configuration_name = configuration
        self.configuration_manager = ConfigurationView(plans["configurations"][configuration])
        self.dataset_json = dataset_json
        self.fold = fold
        # hyper-parameters (nnUNetTrainer.py:178-188)
        self.initial_lr = initial_lr
        self.weight_decay = 3e-5
        self.oversample_foreground_percent = 0.33
        self.num_iterations_per_epoch = 250
        self.num_val_iterations_per_epoch = 50
        self.num_epochs = num_epochs
        self.current_epoch = 0
        self.enable_deep_supervision = True
        self.num_input_channels = None
        self.network = None
        self.optimizer = self.lr_scheduler = None
        self.grad_scaler = torch.amp.GradScaler("cuda") if self.device.type == 'cuda' else None
        self.use_fused_optimizer = True  # unscale + clip + SGD as the fused HIP tail when the network has a gradient arena
        # forward + loss + backward replayed as ONE hipGraph (training/graph_step.py): the explicit schedule issues ~300
        # launches per step; the GPU is the bottleneck, but every launch boundary costs it 2-4 us: replay is worth 3.6 %
        # (13.53 -> 13.07 ms per step, same-box A/B).  Every kernel of this path is deterministic and capture-safe (no
        # memsets, no host syncs, self-resetting scratch), so a replayed step is BIT-identical to an eager one
        # (tests/test_trainer_gpu.py).  Single-process training only: under DDP the step stays eager, because there the
        # gradient all-reduce is overlapped with the backward schedule bucket by bucket.  NNZ_UNET_GRAPH=0 selects eager.
        import os
        self.use_hip_graph = self.device.type == 'cuda' and os.environ.get("NNZ_UNET_GRAPH", "1") != "0"
        self._graphed = None
        self._graphed_ddp = None
        self.loss = None
        self._best_ema = None
        self.inference_allowed_mirroring_axes = None
        self.was_initialized = False
        self.batch_size = self.configuration_manager.batch_size

    # ---- set-up ------------------------------------------------------------------------------------------------
    def initialize(self):
        if self.was_initialized:
            raise RuntimeError("You have called self.initialize even though the trainer was already initialized.")
        self._set_batch_size_and_oversample()
        self.num_input_channels = _num_input_channels(self.dataset_json)
        cm = self.configuration_manager
        self.network = self.build_network_architecture(
            cm.network_arch_class_name, cm.network_arch_init_kwargs, cm.network_arch_init_kwargs_req_import,
            self.num_input_channels, _num_segmentation_heads(self.dataset_json), self.enable_deep_supervision,
            up_sample_type=self.up_sample_type, configuration_manager=cm).to(self.device)
        self.optimizer, self.lr_scheduler = self.configure_optimizers()
        if self.is_ddp:
            attach_bucketed_allreduce(self.network)
        self.loss = self._build_loss()
        self.was_initialized = True

    def _set_batch_size_and_oversample(self):
        """global batch split evenly over ranks, remainder to the low ranks (nnUNetTrainer.py:410-453)."""
        if not self.is_ddp:
            self.batch_size = self.configuration_manager.batch_size
            return
        world, rank = dist.get_world_size(), dist.get_rank()
        gbs = self.configuration_manager.batch_size
        assert gbs >= world, 'Cannot run DDP if the batch size is smaller than the number of GPUs'
        per = gbs // world
        self.batch_size = per + (1 if rank < gbs % world else 0)

    @staticmethod
    def build_network_architecture(*args, **kwargs) -> torch.nn.Module:
        """Accepts BOTH calling conventions found in the reference:
        live   (architecture_class_name, arch_init_kwargs, arch_init_kwargs_req_import, num_input_channels,
                num_output_channels, enable_deep_supervision=True, *, up_sample_type=..., configuration_manager=None)
        legacy (plans_manager, dataset_json, configuration_manager, num_input_channels, enable_deep_supervision=True)
        """
        if len(args) >= 1 and isinstance(args[0], str):
            names = ["architecture_class_name", "arch_init_kwargs", "arch_init_kwargs_req_import",
                     "num_input_channels", "num_output_channels", "enable_deep_supervision"]
            a = dict(zip(names, args))
            a.update(kwargs)
            return get_network_from_plans(a["architecture_class_name"], a["arch_init_kwargs"],
                                          a["arch_init_kwargs_req_import"], a["num_input_channels"],
                                          a["num_output_channels"], allow_init=True,
                                          deep_supervision=a.get("enable_deep_supervision", True),
                                          up_sample_type=a.get("up_sample_type", "convtranspose"))
        names = ["plans_manager", "dataset_json", "configuration_manager", "num_input_channels",
                 "enable_deep_supervision"]
        a = dict(zip(names, args))
        a.update(kwargs)
        cm = a["configuration_manager"]
        return get_network_from_plans(cm.network_arch_class_name, cm.network_arch_init_kwargs,
                                      cm.network_arch_init_kwargs_req_import, a["num_input_channels"],
                                      _num_segmentation_heads(a["dataset_json"]), allow_init=True,
                                      deep_supervision=a.get("enable_deep_supervision", True))

    def _get_deep_supervision_scales(self):
        if not self.enable_deep_supervision:
            return None
        return list(list(i) for i in 1 / np.cumprod(np.vstack(self.configuration_manager.pool_op_kernel_sizes),
                                                     axis=0))[:-1]

    def _do_i_compile(self) -> bool:
        return False  # the explicit HIP schedule replaces torch.compile (nnUNetTrainer.py:296-322)

    def _build_loss(self):
        ig = _ignore_label(self.dataset_json)
        if _has_regions(self.dataset_json):
            loss = DC_and_BCE_loss({}, {'batch_dice': self.configuration_manager.batch_dice, 'do_bg': True, 'smooth': 1e-5,
                                        'ddp': self.is_ddp}, use_ignore_label=ig is not None,
                                   dice_class=MemoryEfficientSoftDiceLoss)
        else:
            loss = DC_and_CE_loss({'batch_dice': self.configuration_manager.batch_dice, 'smooth': 1e-5, 'do_bg': False,
                                   'ddp': self.is_ddp}, {}, weight_ce=1, weight_dice=1, ignore_label=ig,
                                  dice_class=MemoryEfficientSoftDiceLoss)
        if self.enable_deep_supervision:
            scales = self._get_deep_supervision_scales()
            weights = np.array([1 / (2 ** i) for i in range(len(scales))])
            # the reference needs 1e-6 under DDP because torch DDP rejects unused parameters (:476-482); our
            # reducer zero-fills the unused head instead, so the mathematically intended 0 is kept on every rank
            weights[-1] = 0
            weights = weights / weights.sum()
            loss = DeepSupervisionWrapper(loss, weights)
        return loss

    def configure_optimizers(self):
        if hasattr(self.network, "grad_arena"):
            # same optimizer (torch.optim.SGD subclass, identical state_dict); its step can run as the fused HIP tail
            optimizer = FusedSGD(self.network, self.initial_lr, weight_decay=self.weight_decay, momentum=0.99,
                                 nesterov=True)
        else:
            optimizer = torch.optim.SGD(self.network.parameters(), self.initial_lr, weight_decay=self.weight_decay,
                                        momentum=0.99, nesterov=True)
        lr_scheduler = PolyLRScheduler(optimizer, self.initial_lr, self.num_epochs)
        return optimizer, lr_scheduler

    def set_deep_supervision_enabled(self, enabled: bool):
        self.network.decoder.deep_supervision = enabled

    # ---- the hot loop ------------------------------------------------------------------------------------------
    def _autocast_context(self):
        """The native HIP schedule (a network with a gradient arena) already has the numerics of the reference's autocast
        region - fp16 operands, fp32 accumulate - so nothing is wrapped around it.  Any other class resolved from
        plans.json (get_network_from_plans locates arbitrary names) gets `torch.autocast` on cuda exactly as
        nnUNetTrainer.py:1128 / :1178 do, instead of silently training in fp32 under an active GradScaler."""
        if self.device.type != 'cuda' or hasattr(self.network, "grad_arena"):
            return contextlib.nullcontext()
        return torch.autocast('cuda', enabled=True)

    def train_step(self, batch: dict) -> dict:
        data = batch['data'].to(self.device, non_blocking=True)
        target = batch['target']
        if isinstance(target, list):
            target = [i.to(self.device, non_blocking=True) for i in target]
        else:
            target = target.to(self.device, non_blocking=True)
        fused = isinstance(self.optimizer, FusedSGD) and self.use_fused_optimizer
        can_graph = self.use_hip_graph and fused and self.grad_scaler is not None \
            and hasattr(self.network, "grad_arena") and isinstance(target, list)
        graphed = can_graph and not self.is_ddp
        # data-parallel: graph SEGMENTS with the RCCL collectives between them (graph_step.GraphedDDPStep); needs a loss without
        # collectives (batch Dice gathers statistics across ranks) and the fused optimizer (gradients stay in the arena).
        # OPT-IN (NNZ_DDP_GRAPH=1): measured at world size 1 on RCCL (NNZ_BENCH_FORCE_DDP=1, same box) plain graph 13.43 ms,
        # eager DDP 13.95 ms, 8 segments 13.88 ms - each segment boundary (graph launch + collective enqueue) costs ~55 us, so
        # the segmented step recovers 0.06 of the 0.5 ms; the eager step stays the default at N > 1.
        ddp_graphed = can_graph and self.is_ddp and not self.configuration_manager.batch_dice \
            and getattr(self.network, "grad_reducer", None) is not None and self.optimizer.fused_available_static() \
            and _scaler_internals_ok(self.grad_scaler) and os.environ.get("NNZ_DDP_GRAPH", "0") == "1"
        world_div = 1.0
        if ddp_graphed:
            from .graph_step import GraphedDDPStep
            if self._graphed_ddp is None:
                self._graphed_ddp = GraphedDDPStep(self.network, self.loss, self.grad_scaler)
            l = self._graphed_ddp(data, target)
            world_div = float(dist.get_world_size())      # the replay all-reduces SUMs: the mean is taken in the unscale factor
            graphed = True
        elif graphed:
            from .graph_step import GraphedForwardBackward
            if self._graphed is None:
                self._graphed = GraphedForwardBackward(self.network, self.loss, self.grad_scaler, autocast=False)
            l = self._graphed(data, target)
        else:
            self.optimizer.zero_grad(set_to_none=True)
            with self._autocast_context():
                output = self.network(data)
                l = self.loss(output, target)
        if self.grad_scaler is not None:
            if not graphed:
                self.grad_scaler.scale(l).backward()
            if fused and self.optimizer.fused_available() and _scaler_internals_ok(self.grad_scaler):
                # unscale_ + clip_grad_norm_(12) + step + update (nnUNetTrainer.py:1133-1138) without leaving the
                # device: two kernels over the gradient arena, then torch's own scale-update op on the found_inf flag
                sc = self.grad_scaler
                inv_scale = (sc._scale.double() * world_div).reciprocal().float()
                found_inf = self.optimizer.fused_step(inv_scale, 12)
                torch._amp_update_scale_(sc._scale, sc._growth_tracker, found_inf, sc._growth_factor,
                                         sc._backoff_factor, sc._growth_interval)
            else:
                self.grad_scaler.unscale_(self.optimizer)
                torch.nn.utils.clip_grad_norm_(self.network.parameters(), 12)
                self.grad_scaler.step(self.optimizer)
                self.grad_scaler.update()
        else:
            l.backward()
            if fused and self.optimizer.fused_available():
                self.optimizer.fused_step(None, 12)
            else:
                torch.nn.utils.clip_grad_norm_(self.network.parameters(), 12)
                self.optimizer.step()
        return {'loss': l.detach().cpu().numpy()}

    def validation_step(self, batch: dict) -> dict:
        data = batch['data'].to(self.device, non_blocking=True)
        target = batch['target']
        if isinstance(target, list):
            target = [i.to(self.device, non_blocking=True) for i in target]
        else:
            target = target.to(self.device, non_blocking=True)
        with torch.no_grad(), self._autocast_context():
            output = self.network(data)
            l = self.loss(output, target)
        if self.enable_deep_supervision:
            output, target = output[0], target[0]
        # online-Dice statistics in one HIP pass over logits and labels (reference: argmax / sigmoid -> one-hot ->
        # get_tp_fp_fn_tn with the ignore mask, nnUNetTrainer.py:1188-1226); float32 arrays like the reference's
        ig = _ignore_label(self.dataset_json)
        if _has_regions(self.dataset_json):
            tp, fp, fn = ops.region_tp_fp_fn(output, target)      # the ignore mask is the target's last channel
            first = 0                                             # every head predicts some foreground region
        else:
            tgt = target if target.dtype == torch.int16 else target.to(torch.int16)
            tp, fp, fn = ops.argmax_tp_fp_fn(output, tgt, -32768 if ig is None else ig)
            first = 1                                             # drop the background class
        stats = torch.stack([tp, fp, fn]).to(torch.float32).cpu().numpy()
        return {'loss': l.detach().cpu().numpy(), 'tp_hard': stats[0][first:], 'fp_hard': stats[1][first:],
                'fn_hard': stats[2][first:]}

    @staticmethod
    def pseudo_dice(val_outputs: List[dict]) -> List[float]:
        """2TP / (2TP + FP + FN) per foreground class over the validation batches (nnUNetTrainer.py:1229-1259)."""
        tp = np.sum([o['tp_hard'] for o in val_outputs], 0)
        fp = np.sum([o['fp_hard'] for o in val_outputs], 0)
        fn = np.sum([o['fn_hard'] for o in val_outputs], 0)
        return [float(i) for i in [2 * i / (2 * i + j + k) for i, j, k in zip(tp, fp, fn)]]

    # ---- checkpoints (same dictionary layout as the reference) -------------------------------------------------
    def save_checkpoint(self, filename: str) -> None:
        if self.local_rank != 0:
            return
        checkpoint = {
            'network_weights': self.network.state_dict(),
            'optimizer_state': self.optimizer.state_dict(),
            'grad_scaler_state': self.grad_scaler.state_dict() if self.grad_scaler is not None else None,
            'logging': {},
            '_best_ema': self._best_ema,
            'current_epoch': self.current_epoch + 1,
            'init_args': self.my_init_kwargs,
            'trainer_name': self.__class__.__name__,
            'inference_allowed_mirroring_axes': self.inference_allowed_mirroring_axes,
        }
        torch.save(checkpoint, filename)

    def load_checkpoint(self, filename_or_checkpoint: Union[dict, str]) -> None:
        if not self.was_initialized:
            self.initialize()
        ckpt = filename_or_checkpoint
        if isinstance(ckpt, str):
            ckpt = torch.load(ckpt, map_location=self.device, weights_only=False)
        new_state_dict = {}
        own = self.network.state_dict().keys()
        for k, value in ckpt['network_weights'].items():
            key = k
            if key not in own and key.startswith('module.'):
                key = key[7:]
            new_state_dict[key] = value
        self.my_init_kwargs = ckpt['init_args']
        self.current_epoch = ckpt['current_epoch']
        self._best_ema = ckpt['_best_ema']
        self.inference_allowed_mirroring_axes = ckpt.get('inference_allowed_mirroring_axes',
                                                         self.inference_allowed_mirroring_axes)
        self.network.load_state_dict(new_state_dict)
        self.optimizer.load_state_dict(ckpt['optimizer_state'])
        if self.grad_scaler is not None and ckpt['grad_scaler_state'] is not None:
            self.grad_scaler.load_state_dict(ckpt['grad_scaler_state'])
