"""Polynomial learning-rate decay behind the constructor the reference's trainer uses
(/root/reference/nnunetv2/training/lr_scheduler/polylr.py:7-26, called at nnUNetTrainer.py:574 and, once per epoch with
the epoch index, at :1068): lr(step) = initial_lr * (1 - step / max_steps) ** exponent.

Built on torch's LambdaLR (the schedule is a closed-form factor of one base rate), so `state_dict()`,
`get_last_lr()` and chaining behave like any torch scheduler.  `step()` without an argument advances an internal
counter; `step(k)` jumps to step k - both forms occur in the reference's plugins.
"""
from torch.optim.lr_scheduler import LambdaLR


class PolyLRScheduler(LambdaLR):
    def __init__(self, optimizer, initial_lr: float, max_steps: int, exponent: float = 0.9, current_step: int = None,
                 verbose: bool = True):
        del verbose  # accepted for signature compatibility; nothing is printed
        self.initial_lr, self.max_steps, self.exponent = float(initial_lr), int(max_steps), float(exponent)
        self._next = 0
        for group in optimizer.param_groups:
            group["initial_lr"] = self.initial_lr  # LambdaLR's base rate: the schedule ignores the optimizer's own lr
        super().__init__(optimizer, self._factor, last_epoch=-1 if current_step is None else current_step)

    def _factor(self, step: int) -> float:
        return (1.0 - step / self.max_steps) ** self.exponent

    def step(self, current_step=None):
        if current_step is None or current_step == -1:
            current_step, self._next = self._next, self._next + 1
        self.last_epoch = current_step
        lr = self.initial_lr * self._factor(current_step)
        for group in self.optimizer.param_groups:
            group["lr"] = lr
        self._last_lr = [lr] * len(self.optimizer.param_groups)
        return lr
