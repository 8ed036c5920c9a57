"""Oracle (TEST INFRASTRUCTURE): TBN model, attention fusion, consensus and losses.

torch-CPU fp32 restatement of
  * `TBNModel`            reference core/models/model.py:21-334
  * `Fusion`/`Classifier` reference core/models/model.py:337-386
  * `PositionalEncoding`, `MultiheadedAttention`, `UniModalAttention`,
    `PrototypeAttention`  reference core/models/attention.py:8-145
  * `ContrastLoss`        reference core/models/contrast_loss.py:4-25
  * `build_model`         reference core/models/model_builder.py:25-81 (single device)
Weights for the backbones come from a caller-supplied dict of state dicts
(`{"imagenet": sd, "kinetics": sd}`), standing in for the `.pth` files.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.distributions import Categorical

from .bninception import bninception


def gaussian_kernel(n, sigma):
    """cv2.getGaussianKernel(n, sigma) for sigma > 0 (reference attention.py:122, dataset.py:545).

    Note: for n <= 7 and sigma <= 0 OpenCV uses fixed tables; the reference always
    passes sigma=1 so the closed form applies.
    """
    i = np.arange(n, dtype=np.float64)
    k = np.exp(-((i - (n - 1) / 2.0) ** 2) / (2.0 * sigma * sigma))
    return (k / k.sum()).reshape(n, 1)


class PositionalEncoding(nn.Module):
    # reference attention.py:8-45 ("concat" encoding; the dropout attribute ends up a float)
    def __init__(self, dim_size, dropout=0.0, max_len=25, encoding_type="concat", device=None):
        super().__init__()
        self.encoding_type = encoding_type
        self.dim_size = dim_size
        self.max_len = max_len
        self.dropout = dropout
        half = dim_size // 2
        t = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1).expand(-1, half)
        ang = t * torch.arange(1, half + 1)
        pe = torch.zeros(max_len, dim_size)
        pe[:, 0::2] = torch.sin(ang)
        pe[:, 1::2] = torch.cos(ang)
        self.register_buffer("pe", pe.unsqueeze(0).transpose(1, 2))

    def forward(self, x):
        x = x.squeeze(2)
        b = x.shape[0]
        if self.encoding_type == "add":
            x = x + self.pe[: x.size(0), :]
        elif self.encoding_type == "concat":
            x = torch.cat((x, self.pe.expand(b, self.dim_size, self.max_len)), dim=1)
        if self.dropout > 0:
            return F.dropout(x, self.dropout, self.training)
        return x


class MultiheadedAttention(nn.Module):
    # reference attention.py:48-57
    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        self.attention_layer = nn.MultiheadAttention(embed_dim, num_heads, dropout=dropout, bias=True)

    def forward(self, query, key, value):
        return self.attention_layer(query, key, value)


class UniModalAttention(nn.Module):
    # reference attention.py:60-91
    def __init__(self, in_size, out_size, hidden_size=256, use_gumbel=True, temperature=1, one_hot=True):
        super().__init__()
        self.seq = nn.Sequential(nn.Linear(in_size, hidden_size), nn.ReLU(), nn.Linear(hidden_size, out_size))
        self.use_gumbel, self.temperature, self.one_hot = use_gumbel, temperature, one_hot

    def forward(self, vis, aud):
        logits = self.seq(vis)
        if self.training and self.use_gumbel:
            w = F.gumbel_softmax(logits, tau=self.temperature, hard=self.one_hot)
        else:
            w = F.softmax(logits, dim=1)
        return (aud * w.unsqueeze(1)).sum(dim=2), w


class PrototypeAttention(nn.Module):
    # reference attention.py:94-145
    def __init__(self, in_size, win_size, hidden_size=256, use_gumbel=True, temperature=1, device=None):
        super().__init__()
        self.use_gumbel, self.temperature = use_gumbel, temperature
        g = gaussian_kernel(win_size, 1)
        shift = win_size // 2 - 2
        protos = np.concatenate((g, np.roll(g, -shift), np.roll(g, shift)), axis=1).T
        self.register_buffer("prototype_wts", torch.from_numpy(protos).float())
        self.seq = nn.Sequential(nn.Linear(in_size, hidden_size), nn.ReLU(),
                                 nn.Linear(hidden_size, self.prototype_wts.shape[0]))

    def forward(self, vis, aud):
        logits = self.seq(vis)
        if self.training and self.use_gumbel:
            m = F.gumbel_softmax(logits, tau=self.temperature, hard=True)
        else:
            m = F.softmax(logits, dim=1)
        w = torch.matmul(m, self.prototype_wts)
        return (aud * w.unsqueeze(1)).sum(dim=2), w


class ContrastLoss(nn.Module):
    # reference contrast_loss.py:4-25
    def __init__(self, threshold=0.5, reduction=None):
        super().__init__()
        if reduction not in ("mean", "batchmean", "sum"):
            raise Exception(f"{reduction} type reduction not supported for Contrast Loss")
        self.threshold, self.reduction = threshold, reduction

    def forward(self, w):
        mask = (w.detach() >= self.threshold).to(w.dtype)
        loss = (w * (1 - mask) - w * mask).sum(dim=1)
        if self.reduction in ("mean", "batchmean"):
            loss = loss.mean()
        return loss


class Fusion(nn.Module):
    # reference model.py:337-362
    def __init__(self, in_size, out_size, dropout=0):
        super().__init__()
        self.dropout = dropout
        self.fusion_layer = nn.Sequential(nn.Linear(in_size, out_size), nn.ReLU())
        nn.init.normal_(self.fusion_layer[0].weight, 0, 1e-3)
        nn.init.constant_(self.fusion_layer[0].bias, 0)
        if dropout > 0:
            self.dropout_layer = nn.Dropout(p=dropout)

    def forward(self, x):
        x = self.fusion_layer(x)
        return self.dropout_layer(x) if self.dropout > 0 else x


class Classifier(nn.Module):
    # reference model.py:365-386
    def __init__(self, num_classes, in_features):
        super().__init__()
        self.num_classes = num_classes
        for k in num_classes.keys():
            lin = nn.Linear(in_features, num_classes[k])
            nn.init.normal_(lin.weight, 0, 1e-3)
            nn.init.constant_(lin.bias, 0)
            self.add_module(k, lin)

    def forward(self, x):
        return OrderedDict((k, getattr(self, k)(x)) for k in self.num_classes)


class TBNModel(nn.Module):
    # reference model.py:21-334 (bninception arch only)
    IN_CH = {"RGB": 3, "Flow": 10, "Audio": 1}

    def __init__(self, cfg, modality, pretrained):
        super().__init__()
        self.cfg, self.modality = cfg, list(modality)
        att = cfg.model.attention
        self.use_attention, self.attention_type = att.enable, att.type
        feat = 0
        for m in self.modality:
            src = "kinetics" if m == "Flow" else "imagenet"
            base = bninception(self.IN_CH[m], m, pretrained[src], is_audio=(m == "Audio"),
                               attend=self.use_attention, num_classes=400 if m == "Flow" else 1000)
            self.add_module("Base_" + m, base)
            feat += base.feature_size
            if cfg.model.freeze_base:
                self._freeze(m, cfg.model.freeze_mode)
        if len(self.modality) > 1:
            if self.use_attention and not att.use_fixed:
                win = round(cfg.data.audio.audio_length * (25 / 4))
                if att.use_pe:
                    self.pe = nn.Sequential(PositionalEncoding(10, max_len=win),
                                            nn.Conv1d(1034, 1024, kernel_size=1), nn.GroupNorm(64, 1024))
                if self.attention_type == "mha":
                    self.attention_layer = MultiheadedAttention(1024, att.attn_heads, att.attn_dropout)
                elif self.attention_type == "unimodal":
                    self.attention_layer = UniModalAttention(1024, win, 256, att.use_gumbel, 1, True)
                elif self.attention_type == "proto":
                    self.attention_layer = PrototypeAttention(1024, win, 256, att.use_gumbel, 1)
            self.fusion = Fusion(feat, 512, dropout=cfg.model.fusion_dropout)
            self.classifier = Classifier(cfg.model.num_classes, 512)
        else:
            self.classifier = Classifier(cfg.model.num_classes, feat)

    def _freeze(self, m, mode):
        base = getattr(self, "Base_" + m)
        if mode == "all":
            for p in base.parameters():
                p.requires_grad = False
        elif mode == "partialbn":
            for i, mod in enumerate(base.children()):
                if isinstance(mod, nn.BatchNorm2d) and ((m == "Audio" and i > 6) or i > 1):
                    mod.weight.requires_grad = False
                    mod.bias.requires_grad = False

    def forward(self, inp):
        feats, att_wts = [], None
        for i, m in enumerate(self.modality):
            b, n, c, h, w = inp[m].shape
            f = getattr(self, "Base_" + m)(inp[m].view(b * n, c, h, w))
            if m == "Audio":
                if (self.training and len(self.modality) > 1 and self.cfg.data.audio.dropout > 0
                        and np.random.uniform() > self.cfg.data.audio.dropout):
                    f = torch.zeros_like(feats[0])
                elif self.use_attention:
                    if self.cfg.model.attention.use_fixed:
                        f = (f.squeeze(2) * inp["weights"].view(b * n, -1).unsqueeze(1)).sum(2)
                    elif self.attention_type == "mha":
                        f = self.pe(f).transpose(1, 2).transpose(0, 1)
                        f, att_wts = self.attention_layer(feats[0].unsqueeze(0), f, f)
                        f = f.squeeze(0)
                    else:
                        f, att_wts = self.attention_layer(feats[0], f.squeeze(2))
                if i > 0 and feats[0].shape[0] > f.shape[0]:
                    k = feats[0].shape[0] // f.shape[0]
                    f = f.repeat(k, 1)
                    n *= k
            feats.append(f)
        x = torch.cat(feats, dim=1)
        if len(self.modality) > 1:
            x = self.fusion(x)
        out = self.classifier(x)
        for k in out:
            out[k] = out[k].view(b, n, -1).mean(dim=1)
        if self.use_attention and not self.cfg.model.attention.use_fixed:
            out["weights"] = att_wts
        return out

    def get_loss(self, criterion, target, preds, epoch=0):
        att = self.cfg.model.attention
        loss = {"total": 0, "all_class": 0}
        for k in target["class"].keys():
            labels = target["class"][k]
            bs = labels.shape[0]
            loss[k] = criterion["crossentropy"](preds[k], labels)
            loss["all_class"] += loss[k]
        loss["total"] += loss["all_class"]
        if self.use_attention and not att.use_fixed:
            if self.training and epoch + 1 < att.decay_step:
                pm = cm = em = 0
            else:
                pm, cm, em = att.wt_decay, att.contrast_decay, att.entropy_decay
            w = preds["weights"].squeeze(1)
            if att.use_prior:
                b, n, _, _ = target["weights"].shape
                prior = target["weights"].reshape(b * n, -1)
                if att.wt_loss == "kl":
                    w = torch.log(w + 1e-7)
                loss["prior"] = criterion["prior"](w, prior)
                loss["total"] += pm * loss["prior"]
            if att.use_contrast:
                loss["contrast"] = criterion["contrast"](w)
                loss["total"] += cm * loss["contrast"]
            if att.use_entropy:
                loss["entropy"] = Categorical(probs=w + 1e-6).entropy().mean()
                if self.training and em > 0 and loss["entropy"] < att.entropy_thresh:
                    em = 0
                loss["total"] += em * loss["entropy"]
        return loss, bs


_LOSSES = {"crossentropy": nn.CrossEntropyLoss, "nll": nn.NLLLoss, "kl": nn.KLDivLoss,
           "mse": nn.MSELoss, "smoothl1": nn.SmoothL1Loss}


def build_model(cfg, modality, pretrained):
    """reference model_builder.py:25-81 restricted to one CPU device."""
    assert cfg.model.arch == "bninception"
    assert cfg.model.loss_fn in _LOSSES
    model = TBNModel(cfg, modality, pretrained)
    crit = OrderedDict()
    crit[cfg.model.loss_fn] = _LOSSES[cfg.model.loss_fn]()
    if cfg.model.attention.enable:
        if cfg.model.attention.use_prior:
            crit["prior"] = _LOSSES[cfg.model.attention.wt_loss](reduction=cfg.model.attention.loss_reduction)
        if cfg.model.attention.use_contrast:
            crit["contrast"] = ContrastLoss(cfg.model.attention.contrast_thresh,
                                            cfg.model.attention.loss_reduction)
    return model, crit, 1
