"""Loss stack of the nnU-Net training step, same class names / constructor arguments / semantics as the reference:
  DC_and_CE_loss                /root/reference/nnunetv2/training/loss/compound_losses.py:8-56
  MemoryEfficientSoftDiceLoss   /root/reference/nnunetv2/training/loss/dice.py:58-119
  RobustCrossEntropyLoss        /root/reference/nnunetv2/training/loss/robust_ce_loss.py:6-16
  DeepSupervisionWrapper        /root/reference/nnunetv2/training/loss/deep_supervision.py:5-30
  AllGatherGrad                 /root/reference/nnunetv2/utilities/ddp_allgather.py:25-48

Device arithmetic: DC_and_CE_loss runs ONE fused HIP kernel per deep-supervision output (nnz_dc_ce_loss_* in
csrc/loss.hip: softmax, CE sum, Dice sums in one read of the logits; the logit gradient in one more); only the
tiny (B, C) Dice algebra and the DDP all-gather of those statistics stay in torch.  There is no CPU path: CPU
tensors raise (the CPU restatement of the reference formulas is oracle/losses.py, test-only).
"""
from __future__ import annotations

from typing import Any, Callable, Tuple

import torch
from torch import nn


class AllGatherGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx: Any, tensor: torch.Tensor, group=None) -> torch.Tensor:
        ctx.group = group
        gathered = [torch.zeros_like(tensor) for _ in range(torch.distributed.get_world_size())]
        torch.distributed.all_gather(gathered, tensor, group=group)
        return torch.stack(gathered, dim=0)

    @staticmethod
    def backward(ctx: Any, *grad_output: torch.Tensor) -> Tuple[torch.Tensor, None]:
        g = torch.cat(grad_output)
        torch.distributed.all_reduce(g, op=torch.distributed.ReduceOp.SUM, async_op=False, group=ctx.group)
        return g[torch.distributed.get_rank()], None


def softmax_helper_dim1(x: torch.Tensor) -> torch.Tensor:
    return torch.softmax(x, 1)


class RobustCrossEntropyLoss(nn.CrossEntropyLoss):
    def forward(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        if target.ndim == input.ndim:
            assert target.shape[1] == 1
            target = target[:, 0]
        return super().forward(input, target.long())


class MemoryEfficientSoftDiceLoss(nn.Module):
    def __init__(self, apply_nonlin: Callable = None, batch_dice: bool = False, do_bg: bool = True,
                 smooth: float = 1., ddp: bool = True):
        super().__init__()
        self.do_bg, self.batch_dice, self.apply_nonlin, self.smooth, self.ddp = do_bg, batch_dice, apply_nonlin, smooth, ddp

    def dice_from_sums(self, intersect, sum_pred, sum_gt):
        """(b, c) statistics -> -mean dice; shared by the torch path and the fused-kernel path."""
        if self.batch_dice:
            if self.ddp:
                intersect = AllGatherGrad.apply(intersect).sum(0)
                sum_pred = AllGatherGrad.apply(sum_pred).sum(0)
                sum_gt = AllGatherGrad.apply(sum_gt).sum(0)
            intersect, sum_pred, sum_gt = intersect.sum(0), sum_pred.sum(0), sum_gt.sum(0)
        dc = (2 * intersect + self.smooth) / (torch.clip(sum_gt + sum_pred + self.smooth, 1e-8))
        return -dc.mean()

    def forward(self, x, y, loss_mask=None, ignore_label=None):
        """loss_mask is supported in the form the reference produces it (DC_and_CE_loss: `target != ignore_label`):
        pass `ignore_label` and the kernel leaves those voxels out; arbitrary mask tensors raise."""
        if loss_mask is not None:
            raise NotImplementedError("pass ignore_label (the mask DC_and_CE_loss derives); arbitrary loss_mask tensors "
                                      "are not supported by the fused kernel")
        if self.apply_nonlin is not softmax_helper_dim1:
            raise NotImplementedError("the fused HIP Dice statistics assume apply_nonlin=softmax_helper_dim1")
        intersect, sum_pred, sum_gt, _ = _fused_stats(x, y, ignore_label)
        if not self.do_bg:
            intersect, sum_pred, sum_gt = intersect[:, 1:], sum_pred[:, 1:], sum_gt[:, 1:]
        return self.dice_from_sums(intersect, sum_pred, sum_gt.detach())


def _fused_stats(net_output: torch.Tensor, target: torch.Tensor, ignore_label=None):
    if not net_output.is_cuda:
        raise RuntimeError("nnuzoo_amd losses run on MI355X through libnnuzoo_hip.so only (no CPU fallback); "
                           "the CPU restatement is oracle/losses.py (test-only)")
    if net_output.shape[1] > 32:
        raise NotImplementedError("fused Dice+CE kernel supports up to 32 classes (register-resident softmax)")
    if target.ndim == net_output.ndim:
        assert target.shape[1] == 1, "target must be a label map (b, 1, ...)"
    tgt = target if target.dtype == torch.int16 else target.to(torch.int16)
    from ..hip_ops import NO_IGNORE
    return _FusedDiceCE.apply(net_output.contiguous(), tgt.contiguous(),
                              NO_IGNORE if ignore_label is None else int(ignore_label))


class _FusedDiceCE(torch.autograd.Function):
    """softmax + CE-sum + Dice sums in one pass over the logits; backward is one more pass (csrc/loss.hip)."""

    @staticmethod
    def forward(ctx, logits: torch.Tensor, target: torch.Tensor, ignore: int):
        from .. import hip_ops as ops
        B, C = logits.shape[:2]
        V = logits[0, 0].numel()
        sums = torch.empty((B, 3 * C + 1), dtype=torch.float32, device=logits.device)
        ops.dc_ce_forward(logits, target, sums, B, C, V, ignore)
        ctx.save_for_backward(logits, target)
        ctx.dims = (B, C, V, ignore)
        intersect, sum_pred, sum_gt = (sums[:, :C].clone(), sums[:, C:2 * C].clone(), sums[:, 2 * C:3 * C].clone())
        ce_sum = sums[:, 3 * C].clone()
        ctx.mark_non_differentiable(sum_gt)
        return intersect, sum_pred, sum_gt, ce_sum

    @staticmethod
    def backward(ctx, g_int, g_pred, g_gt, g_ce):
        from .. import hip_ops as ops
        logits, target = ctx.saved_tensors
        B, C, V, ignore = ctx.dims
        dev = logits.device
        coef = torch.zeros((B, 2 * C + 1), dtype=torch.float32, device=dev)
        if g_int is not None:
            coef[:, :C] = g_int
        if g_pred is not None:
            coef[:, C:2 * C] = g_pred
        if g_ce is not None:
            coef[:, 2 * C] = g_ce
        dlogits = torch.empty_like(logits)
        ops.dc_ce_backward(logits, target, coef, dlogits, B, C, V, ignore)
        return dlogits, None, None


def _prep_target(net_output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    if target.ndim == net_output.ndim:
        assert target.shape[1] == 1, "target must be a label map (b, 1, ...)"
    return (target if target.dtype == torch.int16 else target.to(torch.int16)).contiguous()


class _FusedDSDiceCE(torch.autograd.Function):
    """The whole (deep-supervision) DC_and_CE_loss on the device: per output one statistics pass over the logits and one
    tiny launch that turns the sums into the weighted loss value and the gradient coefficients (csrc/loss.hip
    dc_ce_finalize_kernel); backward is one pass per output, scaled by the upstream gradient read from device memory
    (the GradScaler's loss scale), so nothing synchronises.  Sum over outputs as deep_supervision.py:30 does."""

    @staticmethod
    def forward(ctx, cfg, weights, targets, *logits):
        from .. import hip_ops as ops
        batch_dice, do_bg, smooth, w_ce, w_dice, ignore = cfg
        dev = logits[0].device
        loss = torch.zeros(1, dtype=torch.float32, device=dev)
        saved, meta = [], []
        for w, lg, tg in zip(weights, logits, targets):
            if w == 0:
                meta.append(None)
                continue
            B, C = lg.shape[:2]
            V = lg[0, 0].numel()
            lg = lg.contiguous()
            sums = torch.empty((B, 3 * C + 1), dtype=torch.float32, device=dev)
            coef = torch.empty((B, 2 * C + 1), dtype=torch.float32, device=dev)
            ops.dc_ce_forward(lg, tg, sums, B, C, V, ignore)
            ops.dc_ce_finalize(sums, loss, coef, B, C, V, batch_dice, do_bg, smooth, w_ce, w_dice, w,
                               ignore != ops.NO_IGNORE)
            meta.append((B, C, V, len(saved)))
            saved += [lg, tg, coef]
        ctx.save_for_backward(*saved)
        ctx.meta, ctx.ignore = meta, ignore
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        from .. import hip_ops as ops
        saved = ctx.saved_tensors
        gmul = g.reshape(1).float().contiguous()
        grads = []
        for m in ctx.meta:
            if m is None:
                grads.append(None)
                continue
            B, C, V, i = m
            lg, tg, coef = saved[i:i + 3]
            dlogits = torch.empty_like(lg)
            ops.dc_ce_backward_scaled(lg, tg, coef, gmul, dlogits, B, C, V, ctx.ignore)
            grads.append(dlogits)
        return (None, None, None, *grads)


class DC_and_CE_loss(nn.Module):
    def __init__(self, soft_dice_kwargs, ce_kwargs, weight_ce=1, weight_dice=1, ignore_label=None,
                 dice_class=MemoryEfficientSoftDiceLoss):
        super().__init__()
        if ignore_label is not None:
            ce_kwargs = dict(ce_kwargs)
            ce_kwargs['ignore_index'] = ignore_label    # as the reference does (compound_losses.py:21-22)
        self.weight_dice, self.weight_ce, self.ignore_label = weight_dice, weight_ce, ignore_label
        self.ce = RobustCrossEntropyLoss(**ce_kwargs)
        self.dc = dice_class(apply_nonlin=softmax_helper_dim1, **soft_dice_kwargs)
        self._plain_ce = not [k for k in ce_kwargs if k != 'ignore_index']

    def finalize_on_device(self) -> bool:
        """batch Dice under DDP sums the statistics over ranks (AllGatherGrad, dice.py:96-103) between the two kernels, so
        that one configuration keeps the element-wise torch arithmetic; everything else is finalised by one launch."""
        if not self._plain_ce or type(self.dc) is not MemoryEfficientSoftDiceLoss:
            return False
        if self.dc.batch_dice and self.dc.ddp and torch.distributed.is_available() \
                and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            return False
        return True

    def fused_sum(self, outputs, targets, weights) -> torch.Tensor:
        """sum_i weights[i] * loss(outputs[i], targets[i]) - the deep-supervision sum - without leaving the device"""
        from ..hip_ops import NO_IGNORE
        for o in outputs:
            if not o.is_cuda:
                raise RuntimeError("nnuzoo_amd losses run on MI355X through libnnuzoo_hip.so only (no CPU fallback); "
                                   "the CPU restatement is oracle/losses.py (test-only)")
            if o.shape[1] > 32:
                raise NotImplementedError("fused Dice+CE kernel supports up to 32 classes (register-resident softmax)")
        if self.ignore_label is not None:
            assert all(t.shape[1] == 1 for t in targets), 'ignore label is not implemented for one hot encoded ' \
                                                          'target variables (DC_and_CE_loss)'
        cfg = (bool(self.dc.batch_dice), bool(self.dc.do_bg), float(self.dc.smooth), float(self.weight_ce),
               float(self.weight_dice), NO_IGNORE if self.ignore_label is None else int(self.ignore_label))
        tg = tuple(_prep_target(o, t) for o, t in zip(outputs, targets))
        return _FusedDSDiceCE.apply(cfg, tuple(float(w) for w in weights), tg, *outputs)

    def forward(self, net_output: torch.Tensor, target: torch.Tensor):
        if not self._plain_ce or not isinstance(self.dc, MemoryEfficientSoftDiceLoss):
            raise NotImplementedError("fused HIP loss: ce_kwargs must be {} and dice_class MemoryEfficientSoftDiceLoss")
        if self.ignore_label is not None:
            assert target.shape[1] == 1, 'ignore label is not implemented for one hot encoded target variables ' \
                                         '(DC_and_CE_loss)'
        if self.finalize_on_device():
            return self.fused_sum((net_output,), (target,), (1.0,))
        intersect, sum_pred, sum_gt_all, ce_sum = _fused_stats(net_output, target, self.ignore_label)
        sum_gt = sum_gt_all
        if not self.dc.do_bg:
            intersect, sum_pred, sum_gt = intersect[:, 1:], sum_pred[:, 1:], sum_gt[:, 1:]
        dc_loss = self.dc.dice_from_sums(intersect, sum_pred, sum_gt.detach()) if self.weight_dice != 0 else 0
        if self.ignore_label is None:
            n_vox = net_output.shape[0] * net_output[0, 0].numel()
        else:
            # mean over the voxels that are not ignored (CrossEntropyLoss(ignore_index)); an all-ignored batch gives CE 0
            # (compound_losses.py:51-52: `num_fg > 0`) - the sum is 0 then, the clamp only avoids 0/0
            n_vox = sum_gt_all.detach().sum().clamp_min(1.0)
        ce_loss = ce_sum.sum() / n_vox if self.weight_ce != 0 else 0
        return self.weight_ce * ce_loss + self.weight_dice * dc_loss


class _FusedDiceBCE(torch.autograd.Function):
    """sigmoid + Dice sums + BCE-with-logits sum in one pass over the logits; backward one more pass (csrc/loss.hip)"""

    @staticmethod
    def forward(ctx, logits: torch.Tensor, target: torch.Tensor):
        from .. import hip_ops as ops
        B, C = logits.shape[:2]
        Ct = target.shape[1]
        V = logits[0, 0].numel()
        sums = torch.empty((B, 3 * C + 2), dtype=torch.float32, device=logits.device)
        ops.dc_bce_forward(logits, target, sums, B, C, Ct, V)
        ctx.save_for_backward(logits, target)
        ctx.dims = (B, C, Ct, V)
        intersect, sum_pred, sum_gt = sums[:, :C].clone(), sums[:, C:2 * C].clone(), sums[:, 2 * C:3 * C].clone()
        bce_sum, mask_sum = sums[:, 3 * C].clone(), sums[:, 3 * C + 1].clone()
        ctx.mark_non_differentiable(sum_gt, mask_sum)
        return intersect, sum_pred, sum_gt, bce_sum, mask_sum

    @staticmethod
    def backward(ctx, g_int, g_pred, g_gt, g_bce, g_mask):
        from .. import hip_ops as ops
        logits, target = ctx.saved_tensors
        B, C, Ct, V = ctx.dims
        coef = torch.zeros((B, 2 * C + 1), dtype=torch.float32, device=logits.device)
        if g_int is not None:
            coef[:, :C] = g_int
        if g_pred is not None:
            coef[:, C:2 * C] = g_pred
        if g_bce is not None:
            coef[:, 2 * C] = g_bce
        dlogits = torch.empty_like(logits)
        ops.dc_bce_backward(logits, target, coef, dlogits, B, C, Ct, V)
        return dlogits, None


class DC_and_BCE_loss(nn.Module):
    """Region-based training loss (reference: training/loss/compound_losses.py:59-109): sigmoid soft Dice over the region
    channels (all of them: do_bg=True is the trainer's setting) + BCE-with-logits, optional ignore mask in the target's
    last channel.  One fused HIP pass; bce_kwargs must be {} (what nnUNetTrainer._build_loss passes)."""

    def __init__(self, bce_kwargs, soft_dice_kwargs, weight_ce=1, weight_dice=1, use_ignore_label: bool = False,
                 dice_class=MemoryEfficientSoftDiceLoss):
        super().__init__()
        if bce_kwargs:
            raise NotImplementedError("fused HIP loss: bce_kwargs must be {}")
        if dice_class is not MemoryEfficientSoftDiceLoss:
            raise NotImplementedError("fused HIP loss: dice_class must be MemoryEfficientSoftDiceLoss")
        self.weight_dice, self.weight_ce, self.use_ignore_label = weight_dice, weight_ce, use_ignore_label
        self.dc = dice_class(apply_nonlin=torch.sigmoid, **soft_dice_kwargs)

    def forward(self, net_output: torch.Tensor, target: torch.Tensor):
        if not net_output.is_cuda:
            raise RuntimeError("nnuzoo_amd losses run on MI355X through libnnuzoo_hip.so only (no CPU fallback); "
                               "the CPU restatement is oracle/losses.py (test-only)")
        C = net_output.shape[1]
        if C > 32:
            raise NotImplementedError("fused Dice+BCE kernel supports up to 32 regions")
        want = C + 1 if self.use_ignore_label else C
        assert target.shape[1] == want, f"target must hold {want} channels (regions{' + ignore mask' if self.use_ignore_label else ''})"
        from ..hip_ops import _regions_i16
        intersect, sum_pred, sum_gt, bce_sum, mask_sum = _FusedDiceBCE.apply(net_output.contiguous(), _regions_i16(target))
        if not self.dc.do_bg:
            intersect, sum_pred, sum_gt = intersect[:, 1:], sum_pred[:, 1:], sum_gt[:, 1:]
        dc_loss = self.dc.dice_from_sums(intersect, sum_pred, sum_gt.detach())
        if self.use_ignore_label:
            # (bce * mask).sum() / clip(mask.sum(), 1e-8): summed over regions, averaged over the unmasked voxels
            ce_loss = bce_sum.sum() / torch.clip(mask_sum.sum(), min=1e-8)
        else:
            ce_loss = bce_sum.sum() / (net_output.shape[0] * C * net_output[0, 0].numel())
        return self.weight_ce * ce_loss + self.weight_dice * dc_loss


class DeepSupervisionWrapper(nn.Module):
    def __init__(self, loss, weight_factors=None):
        super().__init__()
        assert any([x != 0 for x in weight_factors]), "At least one weight factor should be != 0.0"
        self.weight_factors = tuple(weight_factors)
        self.loss = loss

    def forward(self, *args):
        assert all([isinstance(i, (tuple, list)) for i in args]), \
            f"all args must be either tuple or list, got {[type(i) for i in args]}"
        weights = (1,) * len(args[0]) if self.weight_factors is None else self.weight_factors
        if isinstance(self.loss, DC_and_CE_loss) and len(args) == 2 and self.loss.finalize_on_device():
            return self.loss.fused_sum(args[0], args[1], weights[:len(args[0])])
        return sum([weights[i] * self.loss(*inputs) for i, inputs in enumerate(zip(*args)) if weights[i] != 0.0])
