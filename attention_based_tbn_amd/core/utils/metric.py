"""Evaluation metrics of the reference (`core/utils/metric.py:4-157`) with the per-batch top-k correctness and
confusion matrix computed by one HIP launch per class head (`tbn_topk_correct`) instead of `topk` + a Python loop
over the batch that indexes the device matrix element by element."""
import torch

from ..._lib import TbnHipError, call, ptr, stream_ptr


def get_correct_score(out, target, topk, device=None):
    """`Metric._get_correct_score` (reference metric.py:137-157): (correct (maxk, B) bool, conf_mat (C, C) float)"""
    if not out.is_cuda:
        raise TbnHipError("metrics: class scores must be on the GPU (no CPU fallback)")
    out = out.detach().float().contiguous()
    target = target.to(device=out.device, dtype=torch.long).contiguous().view(-1)
    B, Cn = out.shape
    maxk = max(topk)
    correct = torch.empty((maxk, B), dtype=torch.uint8, device=out.device)
    conf = torch.zeros((Cn, Cn), dtype=torch.float32, device=out.device)
    call("tbn_topk_correct", ptr(out), Cn, ptr(target), B, Cn, maxk, ptr(correct), 0, ptr(conf), stream_ptr())
    return correct.bool(), conf


class Metric(object):
    """Same constructor, `set_metrics` / `get_metrics` and rounding as the reference class."""

    def __init__(self, cfg, no_batches, device=torch.device("cuda")):
        self.cfg = cfg
        self.topk = list(cfg.val.topk)
        self.device = device
        self.no_batches = no_batches
        self.multi_class = len(list(cfg.model.num_classes.keys())) > 1
        self.loss, self.accuracy, self.conf_mat = {}, {}, {}
        for key, no_cls in cfg.model.num_classes.items():
            self.accuracy[key] = [0] * len(self.topk)
            self.conf_mat[key] = torch.zeros((no_cls, no_cls), device=device)
            self.loss[key] = 0
        if self.multi_class:
            self.loss["all_class"] = 0
            self.accuracy["all_class"] = [0] * len(self.topk)
        att = cfg.model.attention
        if att.enable and not att.use_fixed:
            if att.use_prior:
                self.loss["prior"] = 0
            if att.use_contrast:
                self.loss["contrast"] = 0
            if att.use_entropy:
                self.loss["entropy"] = 0
        self.loss["total"] = 0

    def set_metrics(self, out, target, batch_size, batch_loss):
        correct = {}
        if self.multi_class:
            correct["all_class"] = []
        for key in out.keys():
            if key == "weights":
                continue
            corr, cm = get_correct_score(out[key], target["class"][key], self.topk, self.device)
            self.conf_mat[key] += cm
            correct[key] = corr
            if self.multi_class:
                correct["all_class"].append(corr)
            self.loss[key] += batch_loss[key].item()
        if self.multi_class:
            self.loss["all_class"] += batch_loss["all_class"].item()
        att = self.cfg.model.attention
        if att.enable and not att.use_fixed:
            for k, on in (("prior", att.use_prior), ("contrast", att.use_contrast), ("entropy", att.use_entropy)):
                if on:
                    self.loss[k] += batch_loss[k].item()
        self.loss["total"] += batch_loss["total"].item()
        for key in self.accuracy.keys():
            for i, k in enumerate(self.topk):
                if key == "all_class":
                    c = correct[key][0][:k].sum(0)
                    for c2 in correct[key][1:]:
                        c = c * c2[:k].sum(0)   # a clip counts when every class head is right within its top k
                    acc = float(c.to(torch.float32).sum().mul_(100.0 / batch_size))
                else:
                    acc = float(correct[key][:k].reshape(-1).to(torch.float32).sum().mul_(100.0 / batch_size))
                self.accuracy[key][i] += acc

    def get_metrics(self):
        for key in self.accuracy.keys():
            self.accuracy[key] = [round(x / self.no_batches, 2) for x in self.accuracy[key]]
        for key in self.loss.keys():
            self.loss[key] = round(self.loss[key] / self.no_batches, 5)
        return self.loss, self.accuracy, self.conf_mat
