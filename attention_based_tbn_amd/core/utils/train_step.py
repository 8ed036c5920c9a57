"""The body of the reference's training loop for ONE iteration, gradient accumulation included.

Reference: core/tools/train.py:66-94.  With `accumulator_step = k` (cfg.train.optim.accumulator_step) iteration
`it` of an epoch does, in this order,

    (it + 1) % k == 0        -> optimizer.zero_grad()                      (:71-72, BEFORE the forward)
    out = model(data); loss, B = model.get_loss(criterion, target, out, epoch)
    loss["total"] /= k                                                      (:79)
    loss["total"].backward()                                                (:81)
    cfg.train.clip_grad      -> clip_grad_norm_(model.parameters(), clip)   (:84-91, EVERY iteration, in place, on
                                                                             whatever has ACCUMULATED in .grad so far)
    (it + 1) % k == k - 1    -> optimizer.step()                            (:93-94)

so for k = 2 the optimiser steps after iterations 0, 2, 4, ... and the gradients are cleared in front of iterations
1, 3, 5, ... -- the very first step of an epoch is taken on one micro-batch, and a gradient that was clipped in
iteration i is clipped AGAIN (together with what iteration i + 1 added) before the step.  `TrainStep` reproduces exactly
that schedule; nothing is "fixed".

MI355X side: for k == 1 the clip is folded into the fused SGD launch (`FusedSGD.step(clip_grad=)`: the gradients are read
once and not rescaled in memory -- legal because the next thing that happens to them is `zero_grad`); for k > 1 the
clipped gradients must persist into the next iteration, so the in-place multi-tensor `clip_grad_norm_` runs every
iteration and `step()` on the stepping ones.

Data parallel (`core.models.DataParallel`, one process per GPU): the reference's nn.DataParallel reduces the replicas'
gradients into the master copy on EVERY backward, and its per-iteration clip sees that global accumulated gradient.
  * clip_grad off: accumulation is linear, so the non-stepping iterations run under `DataParallel.no_sync()` and the
    stepping iteration's all-reduce averages the locally accumulated sums -- one gradient exchange per optimiser step.
    Parameters that a replica may leave without a gradient in some iteration (the audio branch under
    `data.audio.dropout > 0`, reference model.py:215-222) are part of that exchange with whatever each rank accumulated:
    `DataParallel.finish_gradient_sync` sends a rank's existing .grad (zeros where it has none) and only reports "no
    gradient" when NO rank holds one (tests/test_host_cpu.py::test_accumulation_with_optional_branch_gloo_world2);
  * clip_grad on: the clip coefficient is a function of the norm of the GLOBAL accumulated gradient, which no rank can
    form from local data, so every iteration synchronises (avg_r(G + g_r) = G + avg_r(g_r): the previously synchronised
    part G is identical on all ranks) -- identical to the reference, at the reference's own communication volume.
"""
import collections
import contextlib

import torch

from .optim import FusedSGD, clip_grad_norm_


class TrainStep:
    def __init__(self, model, optimizer, criterion, accumulator_step=1, clip_grad=None):
        if int(accumulator_step) < 1:
            raise ValueError("TrainStep: accumulator_step must be >= 1")
        self.model = model
        self.optimizer = optimizer
        self.criterion = criterion
        self.k = int(accumulator_step)
        self.clip_grad = clip_grad if clip_grad else None    # reference: `if cfg.train.clip_grad:`
        self.last_total_norm = None     # 0-dim device tensor of the last clip (None when clipping is off)
        # did the backward of the last iterations exchange gradients (data parallel only)?  Bounded: a run has millions
        self.synced = collections.deque(maxlen=64)

    @classmethod
    def from_config(cls, cfg, model, optimizer, criterion):
        return cls(model, optimizer, criterion, cfg.train.optim.accumulator_step, cfg.train.clip_grad)

    def zeroes(self, iter_no):
        return (iter_no + 1) % self.k == 0

    def steps(self, iter_no):
        return (iter_no + 1) % self.k == self.k - 1

    def _sync_context(self, iter_no):
        no_sync = getattr(self.model, "no_sync", None)
        if no_sync is None or self.clip_grad is not None or self.steps(iter_no):
            self.synced.append(True)
            return contextlib.nullcontext()
        self.synced.append(False)
        return no_sync()

    def __call__(self, iter_no, data, target, epoch=0):
        """one iteration of the reference loop -> (loss dict, batch_size); loss["total"] is already divided by k, as the
        reference's metric bookkeeping sees it (train.py:79-80)"""
        if self.zeroes(iter_no):
            self.optimizer.zero_grad()
        with self._sync_context(iter_no):
            out = self.model(data)
            loss, batch_size = self.model.get_loss(self.criterion, target, out, epoch)
            loss["total"] = loss["total"] / self.k
            loss["total"].backward()
        fused = isinstance(self.optimizer, FusedSGD) and self.k == 1 and self.clip_grad is not None
        if self.clip_grad is not None and not fused:
            params = [p for p in self.model.parameters() if p.grad is not None]
            if params and params[0].grad.is_cuda:
                self.last_total_norm = clip_grad_norm_(params, self.clip_grad)
            else:       # CPU rehearsals of the schedule (gloo tests): torch's own
                self.last_total_norm = torch.nn.utils.clip_grad_norm_(params, self.clip_grad)
        if self.steps(iter_no):
            if fused:
                self.optimizer.step(clip_grad=self.clip_grad, grads_consumed=True)
                self.last_total_norm = self.optimizer.last_total_norm
            else:
                self.optimizer.step()
        return loss, batch_size
