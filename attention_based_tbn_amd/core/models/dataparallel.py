"""Data parallelism for the TBN hot path: one process per GPU, RCCL all-reduce over xGMI.

Replaces reference core/models/dataparallel.py:4-6 (single-process nn.DataParallel: per-step
parameter broadcast + scatter/gather + reduce to GPU 0).  Here every rank holds a replica, runs
the full forward/backward on its own shard of the clips (per-replica BatchNorm statistics, as
nn.DataParallel's chunks had) and gradients are averaged with one collective per LARGE parameter
tensor -- the backbones keep their parameters in a few flat tensors (~41 MB per backbone) -- plus ONE
for all small tensors (heads, biases, BN affine) packed into a flat buffer: four all-reduces per step.
A backbone's all-reduce is issued as soon as autograd has accumulated its flat gradient, i.e. while the
backbones whose backward was enqueued later are still running on their own HIP streams; the packed
small-tensor collective goes last.  No collective on the data path.

`DataParallel(model)` keeps the reference surface: `.module`, `forward`, `get_loss(...)`,
`state_dict()` of the wrapped model under the `module.` prefix.
"""
import torch
import torch.distributed as dist
import torch.nn as nn


class DataParallel(nn.Module):
    def __init__(self, module, device_ids=None, process_group=None, broadcast_parameters=True, overlap=True):
        super().__init__()
        self.module = module
        self.device_ids = device_ids
        self.process_group = process_group
        self.overlap = overlap          # False: every collective at the end of backward (diagnostic)
        self._pending = []
        self._ready = []
        self._callback_queued = False
        self._hooks = []
        self.world_size = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        if self.world_size > 1:
            if broadcast_parameters:
                self.broadcast_parameters()
            for p in self.module.parameters():
                if p.requires_grad:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad_ready))

    def broadcast_parameters(self, src=0):
        """rank `src`'s parameters and buffers become everyone's (once, at wrap time)"""
        with torch.no_grad():
            for t in list(self.module.parameters()) + list(self.module.buffers()):
                dist.broadcast(t, src, group=self.process_group)

    def _reduce_op(self):
        avg = dist.get_backend(self.process_group) == "nccl"
        return avg, (dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM)

    def _on_grad_ready(self, p):
        """post-accumulate-grad hook.  A large (flat backbone) gradient is all-reduced right here: the hook
        runs inside the AccumulateGrad node, whose stream the autograd engine has already made wait for the
        producing backbone's side stream (the stream semantics torch's own DDP reducer relies on), and the
        process group orders its collective after the current stream -- so the transfer starts when THAT
        backbone's backward has finished on the GPU, under the backward of the others.  Hooks fire in the
        same (reverse-forward) order on every rank, so the collectives match up.  Small tensors are only
        remembered; one callback at the end of backward packs and reduces them and waits for everything."""
        if self.overlap and p.grad is not None and p.grad.numel() >= self.SMALL:
            _, op = self._reduce_op()
            self._pending.append((p.grad, dist.all_reduce(p.grad, op=op, group=self.process_group, async_op=True)))
        else:
            self._ready.append(p)
        if not self._callback_queued:
            self._callback_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self.finish_gradient_sync)

    SMALL = 1 << 18   # elements: gradients below 1 MB travel together in one flat buffer (one collective)

    def finish_gradient_sync(self):
        """average every gradient produced by this backward across ranks (RCCL all-reduce): one collective per
        large (flat backbone) tensor -- already in flight when `overlap` -- and ONE for all the small head /
        bias / BN tensors together"""
        avg, op = self._reduce_op()
        grads = [p.grad for p in self._ready if p.grad is not None]
        small = [g for g in grads if g.numel() < self.SMALL]
        large = [g for g in grads if g.numel() >= self.SMALL]
        works = [w for _, w in self._pending]
        works += [dist.all_reduce(g, op=op, group=self.process_group, async_op=True) for g in large]
        large = [g for g, _ in self._pending] + large
        flat = None
        if small:
            flat = torch.cat([g.reshape(-1) for g in small])
            works.append(dist.all_reduce(flat, op=op, group=self.process_group, async_op=True))
        for work in works:
            work.wait()          # stream-level wait on RCCL; host-blocking only on gloo
        if not avg:
            for g in large:
                g.div_(self.world_size)
            if flat is not None:
                flat.div_(self.world_size)
        if flat is not None:
            off = 0
            for g in small:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
        self._ready = []
        self._pending = []
        self._callback_queued = False

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def get_loss(self, criterion, target, preds, epoch=0):
        return self.module.get_loss(criterion, target, preds, epoch)
