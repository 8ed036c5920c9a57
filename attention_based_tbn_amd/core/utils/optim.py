"""Train-step shell on the MI355X: multi-tensor gradient clipping + SGD(momentum) in three launches.

Mirrors the reference loop body (core/tools/train.py:82-94) and its optimiser construction
(core/tools/train.py:190-202): `torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.train.clip_grad)`
followed by `optim.SGD(lr, momentum, weight_decay).step()`.  `FusedSGD` is a `torch.optim.Optimizer`, so
`MultiStepLR` / the warm-up scheduler of the reference drive it unchanged, and its `state_dict()` has torch's
layout (`momentum_buffer` per parameter).  No host synchronisation: the total norm and the clipping
coefficient stay on the device (`clip_grad_norm_` returns the norm as a 0-dim device tensor, like torch).
There is no CPU fallback: CPU tensors raise.
"""
import ctypes as C

import torch

from ..._lib import OptTensor, TbnHipError, call, lib, ptr, stream_ptr

MAX_TENSORS = 48   # TBN_OPT_MAX_TENSORS


def _entries(params, with_param, bufs=None):
    ents = []
    for i, p in enumerate(params):
        g = p.grad
        if g is None:
            continue
        if not (p.is_cuda and g.is_cuda):
            raise TbnHipError("fused optimiser: parameters and gradients must live on the GPU (no CPU fallback)")
        if p.dtype != torch.float32 or g.dtype != torch.float32 or not p.is_contiguous() or not g.is_contiguous():
            raise TbnHipError("fused optimiser: contiguous fp32 parameters / gradients only")
        if g.is_sparse:
            raise TbnHipError("fused optimiser: sparse gradients are not supported")
        ents.append(OptTensor(ptr(p) if with_param else 0, ptr(g), ptr(bufs[i]) if bufs is not None else 0, p.numel()))
    return ents


def _chunks(ents):
    for i in range(0, len(ents), MAX_TENSORS):
        part = ents[i:i + MAX_TENSORS]
        yield (OptTensor * len(part))(*part), len(part)


def _sqnorm_coef(ents, max_norm, device):
    """-> (total_norm, coef) 1-element device tensors"""
    L = lib()
    arrs = list(_chunks(ents))
    counts = [L.tbn_opt_num_partials(a, n) for a, n in arrs]
    partials = torch.empty(max(1, sum(counts)), dtype=torch.float32, device=device)
    st, off = stream_ptr(), 0
    for (a, n), c in zip(arrs, counts):
        call("tbn_opt_sqnorm_partials", a, n, partials.data_ptr() + 4 * off, st)
        off += c
    out = torch.empty(2, dtype=torch.float32, device=device)
    call("tbn_opt_clip_coef", ptr(partials), off, float(max_norm), out.data_ptr(), out.data_ptr() + 4, st)
    return out[0], out[1:2]


def clip_grad_norm_(parameters, max_norm, norm_type=2.0):
    """Drop-in for `torch.nn.utils.clip_grad_norm_` (L2 only) on GPU gradients: scales them in place by
    min(1, max_norm / (total_norm + 1e-6)) and returns the total norm (0-dim device tensor, not synchronised)."""
    if float(norm_type) != 2.0:
        raise TbnHipError("clip_grad_norm_: only norm_type=2 is implemented on the HIP path")
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    params = [p for p in parameters if p.grad is not None]
    if not params:
        return torch.tensor(0.0)
    ents = _entries(params, False)
    total, coef = _sqnorm_coef(ents, max_norm, params[0].device)
    for a, n in _chunks(ents):
        call("tbn_opt_scale_grads", a, n, ptr(coef), stream_ptr())
    return total


class FusedSGD(torch.optim.Optimizer):
    """`torch.optim.SGD(params, lr, momentum, weight_decay)` (dampening 0, no Nesterov) as one multi-tensor HIP
    launch per step.  `step(clip_grad=max_norm)` additionally runs `clip_grad_norm_` (norm in `self.last_total_norm`,
    a device tensor) and is equivalent to the reference's `clip_grad_norm_(...)` + `optimizer.step()`
    (core/tools/train.py:84-94) in every schedule: the gradients are rescaled IN MEMORY, so what a later backward
    accumulates onto them (accumulator_step > 1, train.py:71-72) is what the reference accumulates onto.
    `step(clip_grad=max_norm, grads_consumed=True)` folds the clip coefficient into the update instead (gradients read
    once, NOT rescaled in memory: one launch and one pass over the gradients fewer) -- only for callers that clear the
    gradients before the next backward (accumulator_step == 1: `TrainStep`, `bench.py`); under accumulation it would
    silently differ from the reference, which is why it is opt-in."""

    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0):
        if lr < 0.0 or momentum < 0.0 or weight_decay < 0.0:
            raise ValueError("FusedSGD: lr, momentum and weight_decay must be non-negative")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, dampening=0,
                                      nesterov=False, maximize=False, foreach=None, differentiable=False,
                                      fused=None))
        self.last_total_norm = None

    @torch.no_grad()
    def step(self, closure=None, clip_grad=None, grads_consumed=False):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        gscale = None
        if clip_grad:
            allp = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
            if allp:
                ents = _entries(allp, False)
                self.last_total_norm, gscale = _sqnorm_coef(ents, clip_grad, allp[0].device)
                if not grads_consumed:
                    for a, n in _chunks(ents):
                        call("tbn_opt_scale_grads", a, n, ptr(gscale), stream_ptr())
                    gscale = None
        for group in self.param_groups:
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            mom = float(group["momentum"])
            bufs = None
            if mom != 0.0:
                bufs = []
                for p in params:
                    st = self.state[p]
                    if st.get("momentum_buffer") is None:
                        st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    bufs.append(st["momentum_buffer"])
            ents = _entries(params, True, bufs)
            for a, n in _chunks(ents):
                call("tbn_opt_sgd_step", a, n, float(group["lr"]), mom, float(group["weight_decay"]), ptr(gscale),
                     stream_ptr())
            # the kernel wrote the parameters (and momentum buffers) through raw pointers: tell autograd, as an in-place
            # torch op would have -- saved-tensor checks and every cache keyed on `_version` (the classifier's stacked
            # weight copy, core/models/model.py) rely on the counter
            torch.autograd.graph.increment_version(params)
            if bufs is not None:
                torch.autograd.graph.increment_version(bufs)
        return loss
