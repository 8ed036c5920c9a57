"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement, in NumPy, of the train-step shell and the metrics the reference runs right after the hot path:

  clip_grad_norm   torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.train.clip_grad)   core/tools/train.py:82-85
  sgd_step         torch.optim.SGD(lr, momentum, weight_decay).step()                        core/tools/train.py:93,190-196
  multistep_lr     optim.lr_scheduler.MultiStepLR(milestones, gamma)                          core/tools/train.py:197-201
  topk_correct     Metric._get_correct_score                                                  core/utils/metric.py:137-157
  Metric           Metric.set_metrics / get_metrics                                           core/utils/metric.py:18-135

The algorithms live in torch (a dependency of the reference, any 1.x/2.x release computes the same formulas):
total_norm = || [ ||g_i||_2 ] ||_2, coef = clamp(max_norm / (total_norm + 1e-6), max=1), g *= coef;
d = g + wd * p, buf = momentum * buf + d (first step buf = d), p -= lr * buf.
Pinned by tests/golden/trainstep.npz, metric.json (tests/golden/make_golden_trainstep.py runs torch's own
functions and the unmodified reference classes).
"""
import numpy as np


def clip_grad_norm(grads, max_norm):
    """-> (total_norm, [clipped grads]); fp32 arithmetic like torch, accumulation in fp64 then rounded"""
    total = np.float32(np.sqrt(sum(float(np.sum(g.astype(np.float64) ** 2)) for g in grads)))
    coef = np.float32(max_norm) / (total + np.float32(1e-6))
    coef = np.float32(min(coef, np.float32(1.0)))
    return total, [(g * coef).astype(np.float32) for g in grads]


def sgd_step(p, g, buf, lr, momentum, weight_decay):
    """one parameter tensor; buf None on the first step -> (p_new, buf_new)"""
    lr, momentum, weight_decay = np.float32(lr), np.float32(momentum), np.float32(weight_decay)
    d = g.astype(np.float32)
    if weight_decay != 0:
        d = d + weight_decay * p
    if momentum != 0:
        buf = d.copy() if buf is None else momentum * buf + d
        d = buf
    return (p - lr * d).astype(np.float32), buf


def multistep_lr(base_lr, epoch, milestones, gamma):
    return base_lr * gamma ** sum(1 for m in milestones if epoch >= m)


def topk_correct(scores, target, maxk):
    """-> (correct (maxk, B) bool, conf_mat (C, C)); ranking = value descending, lower class index first on ties"""
    scores = np.asarray(scores, dtype=np.float32)
    target = np.asarray(target, dtype=np.int64)
    B, C = scores.shape
    order = np.lexsort((np.arange(C)[None, :].repeat(B, 0), -scores), axis=1)[:, :maxk]    # (B, maxk)
    correct = (order.T == target[None, :])
    conf = np.zeros((C, C), dtype=np.float32)
    for t, p in zip(target, order[:, 0]):
        conf[t, p] += 1
    return correct, conf


class Metric:
    """reference core/utils/metric.py:4-135 on NumPy arrays (losses are plain floats)"""

    def __init__(self, num_classes, topk, no_batches, extra_losses=()):
        self.topk, self.no_batches = list(topk), no_batches
        self.multi = len(num_classes) > 1
        self.accuracy = {k: [0] * len(self.topk) for k in num_classes}
        self.conf_mat = {k: np.zeros((n, n), dtype=np.float32) for k, n in num_classes.items()}
        self.loss = {k: 0 for k in num_classes}
        if self.multi:
            self.loss["all_class"] = 0
            self.accuracy["all_class"] = [0] * len(self.topk)
        for k in extra_losses:
            self.loss[k] = 0
        self.loss["total"] = 0

    def set_metrics(self, out, target, batch_size, batch_loss):
        correct = {}
        for key, sc in out.items():
            corr, cm = topk_correct(sc, target[key], max(self.topk))
            self.conf_mat[key] += cm
            correct[key] = corr
        for k in self.loss:
            self.loss[k] += batch_loss[k]
        for key in self.accuracy:
            for i, k in enumerate(self.topk):
                if key == "all_class":
                    c = None
                    for ck in out:
                        hit = correct[ck][:k].sum(0)
                        c = hit if c is None else c * hit
                    acc = float(np.float32(c.astype(np.float32).sum()) * np.float32(100.0 / batch_size))
                else:
                    acc = float(np.float32(correct[key][:k].astype(np.float32).sum()) * np.float32(100.0 / batch_size))
                self.accuracy[key][i] += acc

    def get_metrics(self):
        acc = {k: [round(x / self.no_batches, 2) for x in v] for k, v in self.accuracy.items()}
        loss = {k: round(v / self.no_batches, 5) for k, v in self.loss.items()}
        return loss, acc, self.conf_mat
