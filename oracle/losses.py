"""TEST INFRASTRUCTURE ONLY - CPU restatement (plain torch) of the reference's loss stack.

Follows, line by line in meaning:
  DC_and_CE_loss.forward               /root/reference/nnunetv2/training/loss/compound_losses.py:31-56
  MemoryEfficientSoftDiceLoss.forward  /root/reference/nnunetv2/training/loss/dice.py:72-119 (ddp=False branch)
  RobustCrossEntropyLoss.forward       /root/reference/nnunetv2/training/loss/robust_ce_loss.py:12-16
  DeepSupervisionWrapper.forward       /root/reference/nnunetv2/training/loss/deep_supervision.py:19-30
  deep-supervision weights             /root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainer.py:473-487
Pinned by KAT-2 / KAT-3 of SURVEY.md §8c and by tests/golden/loss_*.npz, which were produced by importing the
reference's own modules in the build container (tools/make_golden.py).
"""
import numpy as np
import torch
import torch.nn.functional as F


def soft_dice(x_logits, y, batch_dice, do_bg=False, smooth=1e-5, loss_mask=None):
    x = torch.softmax(x_logits.float(), 1)
    axes = tuple(range(2, x.ndim))
    with torch.no_grad():
        y_onehot = torch.zeros(x.shape, dtype=torch.bool)
        y_onehot.scatter_(1, y.long(), 1)
        if not do_bg:
            y_onehot = y_onehot[:, 1:]
        sum_gt = y_onehot.sum(axes) if loss_mask is None else (y_onehot * loss_mask).sum(axes)
    if not do_bg:
        x = x[:, 1:]
    if loss_mask is None:
        intersect = (x * y_onehot).sum(axes)
        sum_pred = x.sum(axes)
    else:  # dice.py:98-103
        intersect = (x * y_onehot * loss_mask).sum(axes)
        sum_pred = (x * loss_mask).sum(axes)
    if batch_dice:
        intersect, sum_pred, sum_gt = intersect.sum(0), sum_pred.sum(0), sum_gt.sum(0)
    dc = (2 * intersect + smooth) / torch.clip(sum_gt + sum_pred + smooth, 1e-8)
    return -dc.mean()


def dc_and_ce(x_logits, y, batch_dice, weight_ce=1.0, weight_dice=1.0, ignore_label=None):
    if ignore_label is None:
        ce = F.cross_entropy(x_logits.float(), y[:, 0].long())
        return weight_ce * ce + weight_dice * soft_dice(x_logits, y, batch_dice)
    # compound_losses.py:38-55: mask = target != ignore; Dice on the masked sums; CE with ignore_index (0 if all ignored)
    mask = y != ignore_label
    y_dice = torch.where(mask, y, torch.zeros_like(y))
    dc = soft_dice(x_logits, y_dice, batch_dice, loss_mask=mask)
    ce = F.cross_entropy(x_logits.float(), y[:, 0].long(), ignore_index=ignore_label) if mask.sum() > 0 else 0
    return weight_ce * ce + weight_dice * dc


def dc_and_bce(x_logits, y_regions, batch_dice, use_ignore_label=False, smooth=1e-5):
    """DC_and_BCE_loss.forward (compound_losses.py:82-109) with MemoryEfficientSoftDiceLoss(sigmoid, do_bg=True)"""
    if use_ignore_label:
        mask = (1 - y_regions[:, -1:]).bool()
        yr = y_regions[:, :-1]
    else:
        mask, yr = None, y_regions
    x = torch.sigmoid(x_logits.float())
    axes = tuple(range(2, x.ndim))
    yf = yr.float()
    if mask is None:
        intersect, sum_pred, sum_gt = (x * yf).sum(axes), x.sum(axes), yf.sum(axes)
    else:
        intersect, sum_pred, sum_gt = (x * yf * mask).sum(axes), (x * mask).sum(axes), (yf * mask).sum(axes)
    if batch_dice:
        intersect, sum_pred, sum_gt = intersect.sum(0), sum_pred.sum(0), sum_gt.sum(0)
    dc = -((2 * intersect + smooth) / torch.clip(sum_gt + sum_pred + smooth, 1e-8)).mean()
    if mask is not None:
        bce = (F.binary_cross_entropy_with_logits(x_logits.float(), yf, reduction='none') * mask).sum() / \
            torch.clip(mask.sum(), min=1e-8)
    else:
        bce = F.binary_cross_entropy_with_logits(x_logits.float(), yf)
    return bce + dc


def region_tp_fp_fn(logits, target_regions, use_ignore_label=False):
    """validation_step for regions (nnUNetTrainer.py:1188-1216): prediction sigmoid > 0.5, get_tp_fp_fn_tn with mask"""
    axes = [0] + list(range(2, logits.ndim))
    pred = (torch.sigmoid(logits.float()) > 0.5).float()
    if use_ignore_label:
        mask = (1 - target_regions[:, -1:]).float()
        tgt = target_regions[:, :-1].float()
    else:
        mask, tgt = torch.ones_like(pred[:, :1]), target_regions.float()
    tp = (pred * tgt * mask).sum(axes)
    fp = (pred * (1 - tgt) * mask).sum(axes)
    fn = ((1 - pred) * tgt * mask).sum(axes)
    return tp, fp, fn


def ds_weights(n_outputs, ddp_no_compile=False):
    w = np.array([1 / (2 ** i) for i in range(n_outputs)])
    w[-1] = 1e-6 if ddp_no_compile else 0
    return w / w.sum()


def deep_supervision_loss(outputs, targets, batch_dice, weights=None):
    if weights is None:
        weights = ds_weights(len(outputs))
    return sum(w * dc_and_ce(o, t, batch_dice) for w, o, t in zip(weights, outputs, targets) if w != 0)


def tp_fp_fn_hard(logits: torch.Tensor, target: torch.Tensor, ignore_label=None):
    """Online-Dice statistics of validation_step (nnUNetTrainer.py:1188-1221: argmax -> one-hot scatter ->
    get_tp_fp_fn_tn, dice.py:122-180, label-map target; with an ignore label the mask `target != ignore` multiplies
    every term and ignored voxels are relabelled 0 first, :1199-1203): float32 tp / fp / fn per class.
    Pinned by tests/golden/tp_fp_fn.npz (outputs of the reference's own get_tp_fp_fn_tn)."""
    axes = [0] + list(range(2, logits.ndim))
    seg = logits.argmax(1)[:, None]
    onehot = torch.zeros(logits.shape, dtype=torch.float32)
    onehot.scatter_(1, seg, 1)
    target = target.clone()
    mask = torch.ones(target.shape, dtype=torch.float32)
    if ignore_label is not None:
        mask = (target != ignore_label).float()
        target[target == ignore_label] = 0
    y = torch.zeros(logits.shape, dtype=torch.bool)
    y.scatter_(1, target.long(), 1)
    tp = (onehot * y * mask).sum(axes)
    fp = (onehot * (~y) * mask).sum(axes)
    fn = ((1 - onehot) * y * mask).sum(axes)
    return tp, fp, fn
