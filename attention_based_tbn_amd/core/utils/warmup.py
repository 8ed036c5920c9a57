"""`GradualWarmupScheduler` + the optimiser / scheduler construction of the reference's trainer.

Reference: core/tools/train.py:190-217 builds `optim.SGD` + `MultiStepLR` (or `optim.Adam`, no scheduler) and, when
`cfg.train.warmup.enable`, wraps the scheduler in `warmup_scheduler.GradualWarmupScheduler(optimizer, multiplier,
total_epoch, after_scheduler)`, stepped as `scheduler_warmup.step(epoch + 1)` (train.py:291-295).

`warmup_scheduler` is a third-party package (ildoonet/pytorch-gradual-warmup-lr, un-pinned in install/requirements.txt)
that is NOT in this image: the class below restates its published algorithm -- **parity unpinned** (no reference-held
fixture exists; known-answer tests in tests/test_host_cpu.py):
  * epochs 1 .. total_epoch: lr = base_lr * epoch / total_epoch                     (multiplier == 1: ramp from 0)
                            lr = base_lr * ((multiplier - 1) * epoch / total_epoch + 1)   (multiplier > 1)
  * afterwards: the wrapped scheduler, whose base learning rates become base_lr * multiplier and which is stepped with
    (epoch - total_epoch).
Host arithmetic only: it drives `FusedSGD` (a torch.optim.Optimizer) unchanged.
"""
import torch
from torch.optim.lr_scheduler import _LRScheduler

from .optim import FusedSGD


class GradualWarmupScheduler(_LRScheduler):
    def __init__(self, optimizer, multiplier, total_epoch, after_scheduler=None):
        self.multiplier = multiplier
        if self.multiplier < 1.0:
            raise ValueError("multiplier should be greater thant or equal to 1.")
        self.total_epoch = total_epoch
        self.after_scheduler = after_scheduler
        self.finished = False
        super().__init__(optimizer)

    def get_lr(self):
        if self.last_epoch > self.total_epoch:
            if self.after_scheduler:
                if not self.finished:
                    self.after_scheduler.base_lrs = [base_lr * self.multiplier for base_lr in self.base_lrs]
                    self.finished = True
                return self.after_scheduler.get_last_lr()
            return [base_lr * self.multiplier for base_lr in self.base_lrs]
        if self.multiplier == 1.0:
            return [base_lr * (float(self.last_epoch) / self.total_epoch) for base_lr in self.base_lrs]
        return [base_lr * ((self.multiplier - 1.0) * self.last_epoch / self.total_epoch + 1.0) for base_lr in self.base_lrs]

    def step(self, epoch=None):
        if self.finished and self.after_scheduler:
            if epoch is None:
                self.after_scheduler.step(None)
            else:
                self.after_scheduler.step(epoch - self.total_epoch)
            self._last_lr = self.after_scheduler.get_last_lr()
        else:
            super().step(epoch)


def build_optimizer(cfg, model):
    """core/tools/train.py:190-217 -> (optimizer, lr_scheduler or None, scheduler_warmup or None).  "sgd": the fused
    multi-tensor HIP optimiser (`FusedSGD`: same update rule and state layout as `optim.SGD`) + `MultiStepLR`; "adam":
    `torch.optim.Adam` (torch-ROCm; the reference's Adam branch has no scheduler)."""
    kind = cfg.train.optim.type.lower()
    lr_scheduler = scheduler_warmup = None
    if kind == "sgd":
        optimizer = FusedSGD(model.parameters(), cfg.train.optim.lr, momentum=cfg.train.optim.momentum,
                             weight_decay=cfg.train.optim.weight_decay)
        lr_scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=list(cfg.train.scheduler.lr_steps),
                                                            gamma=cfg.train.scheduler.lr_decay)
    elif kind == "adam":
        optimizer = torch.optim.Adam(model.parameters(), cfg.train.optim.lr, betas=(0.9, 0.999),
                                     weight_decay=cfg.train.optim.weight_decay)
    else:
        raise ValueError(f"unknown optimiser '{cfg.train.optim.type}' (the reference knows 'sgd' and 'adam')")
    if lr_scheduler and cfg.train.warmup.enable:
        scheduler_warmup = GradualWarmupScheduler(optimizer, multiplier=cfg.train.warmup.multiplier,
                                                  total_epoch=cfg.train.warmup.epochs, after_scheduler=lr_scheduler)
    return optimizer, lr_scheduler, scheduler_warmup
