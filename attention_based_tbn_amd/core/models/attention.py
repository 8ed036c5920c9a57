"""Attention modules of the mid-level fusion -- API mirror of reference core/models/attention.py
(`PositionalEncoding` :8-45, `MultiheadedAttention` :48-57, `UniModalAttention` :60-91,
`PrototypeAttention` :94-145) with the math on the HIP kernels (ops.py).

Native layout for the audio sequence is (R, T, C) (channels fastest).  The modules accept the
reference's layouts as well -- (R, C, 1, T) for the positional encoding, (T, R, C) key/value for
the attention -- so existing callers keep working; `TBNModel` uses the native fast paths.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops


def gaussian_kernel(n, sigma=1.0):
    """cv2.getGaussianKernel(n, sigma) for sigma > 0 (reference attention.py:122; cv2 not required)"""
    i = np.arange(n, dtype=np.float64)
    k = np.exp(-((i - (n - 1) / 2.0) ** 2) / (2.0 * sigma * sigma))
    return (k / k.sum()).reshape(n, 1)


class PositionalEncoding(nn.Module):
    def __init__(self, dim_size, dropout=0.0, max_len=25, encoding_type="concat", device=None):
        super().__init__()
        self.encoding_type = encoding_type
        self.dim_size = dim_size
        self.max_len = max_len
        self.dropout = dropout
        half = dim_size // 2
        pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1).expand(-1, half) * torch.arange(1, half + 1)
        pe = torch.zeros(max_len, dim_size)
        pe[:, 0::2] = torch.sin(pos)
        pe[:, 1::2] = torch.cos(pos)
        self.register_buffer("pe", pe.unsqueeze(0).transpose(1, 2))       # (1, dim, max_len) like the reference

    def forward_sequence(self, seq, out_ld):
        """native path: (R, T, C) -> (R, T, out_ld) = [feat | pe | 0-pad]"""
        if self.encoding_type != "concat":
            raise NotImplementedError("only the 'concat' encoding the reference uses is on the HIP path")
        assert seq.shape[1] == self.max_len, "attention window must equal max_len (reference attention.py:41)"
        return ops.pe_concat(seq, self.pe[0], out_ld)

    def forward(self, x):
        """reference layout: (R, C, 1, T) -> (R, C + dim, T); `encoding_type="add"` (unused by TBNModel, reference
        attention.py:38-39): x + pe with the reference's own broadcasting rule -- it needs C == dim_size and T == max_len and
        fails like the reference otherwise -- one elementwise torch-ROCm add, there is nothing to accelerate"""
        if self.encoding_type == "add":
            x = x.squeeze(2)
            if not x.is_cuda:
                raise ops.TbnHipError("PositionalEncoding: tensor is on the CPU; the TBN hot path only runs on an MI355X")
            out = x + self.pe[: x.size(0), :]
            return ops.dropout(out, self.dropout, self.training) if self.dropout > 0 else out
        seq = x.squeeze(2).transpose(1, 2)
        c = seq.shape[2] + self.dim_size
        out = self.forward_sequence(seq, (c + 31) // 32 * 32)[:, :, :c]
        out = out.transpose(1, 2)
        return ops.dropout(out, self.dropout, self.training) if self.dropout > 0 else out


class _MHAParams(nn.Module):
    """parameter container with torch.nn.MultiheadAttention's names/shapes/init"""

    def __init__(self, embed_dim, num_heads, dropout):
        super().__init__()
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=True)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.0)


class MultiheadedAttention(nn.Module):
    """torch.nn.MultiheadAttention(embed_dim, heads, dropout, bias=True) restricted to what the TBN
    uses: ONE query per sample, key == value (reference model.py:231-237)."""

    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        assert embed_dim % num_heads == 0
        self.attention_layer = _MHAParams(embed_dim, num_heads, dropout)

    def attend(self, query, seq):
        """native path: query (R, E), seq (R, T, E) -> (out (R, E), weights (R, T))"""
        a = self.attention_layer
        E, H = a.embed_dim, a.num_heads
        R, T, _ = seq.shape
        q = ops.linear(query, a.in_proj_weight[:E], a.in_proj_bias[:E])
        kv = ops.linear(seq.reshape(R * T, E), a.in_proj_weight[E:], a.in_proj_bias[E:]).view(R, T, 2 * E)
        mask = ops.dropout_mask((R, H, T), a.dropout, self.training, seq.device)
        ctx, w = ops.mha_q1(q, kv, mask, H)
        return ops.linear(ctx, a.out_proj.weight, a.out_proj.bias), w

    def forward(self, query, key, value):
        """reference layout: query (L, R, E), key / value (T, R, E) -> ((L, R, E), (R, L, T)) like
        torch.nn.MultiheadAttention (reference attention.py:48-57).  The TBN's own call -- ONE query per sample, key is value
        (model.py:231-237) -- runs on the wavefront-reduction kernel; any other call shape takes the general path: the three
        projections and the output projection on the HIP GEMM, the (tiny) score / softmax / weighted-sum core as batched
        torch-ROCm ops."""
        if key is value and query.shape[0] == 1:
            out, w = self.attend(query[0], key.transpose(0, 1).contiguous())
            return out.unsqueeze(0), w.unsqueeze(1)
        a = self.attention_layer
        E, H = a.embed_dim, a.num_heads
        L, R, _ = query.shape
        T = key.shape[0]
        assert key.shape[1] == R and value.shape[:2] == key.shape[:2] and query.shape[2] == E
        d = E // H
        q = ops.linear(query.reshape(L * R, E), a.in_proj_weight[:E], a.in_proj_bias[:E]) * (float(d) ** -0.5)
        k = ops.linear(key.reshape(T * R, E), a.in_proj_weight[E:2 * E], a.in_proj_bias[E:2 * E])
        v = ops.linear(value.reshape(T * R, E), a.in_proj_weight[2 * E:], a.in_proj_bias[2 * E:])
        q = q.view(L, R * H, d).transpose(0, 1)              # (R H, L, d): torch's head layout
        k = k.view(T, R * H, d).transpose(0, 1)
        v = v.view(T, R * H, d).transpose(0, 1)
        p = torch.softmax(torch.bmm(q, k.transpose(1, 2)), dim=-1)      # (R H, L, T)
        if self.training and a.dropout > 0:
            p = F.dropout(p, p=a.dropout)
        ctx = torch.bmm(p, v).transpose(0, 1).reshape(L * R, E)
        out = ops.linear(ctx, a.out_proj.weight, a.out_proj.bias).view(L, R, E)
        return out, p.view(R, H, L, T).mean(dim=1)


class UniModalAttention(nn.Module):
    def __init__(self, in_size, out_size, hidden_size=256, use_gumbel=True, temperature=1, one_hot=True):
        super().__init__()
        self.seq = nn.Sequential(nn.Linear(in_size, hidden_size), nn.ReLU(), nn.Linear(hidden_size, out_size))
        self.use_gumbel, self.temperature, self.one_hot = use_gumbel, temperature, one_hot

    def _logits(self, x):
        h = ops.linear(x, self.seq[0].weight, self.seq[0].bias, relu=True)
        return ops.linear(h, self.seq[2].weight, self.seq[2].bias)

    def attend(self, vis, seq):
        """vis (R, C); seq (R, T, C) native layout"""
        logits = self._logits(vis)
        if self.training and self.use_gumbel:
            w = F.gumbel_softmax(logits, tau=self.temperature, hard=self.one_hot)
        else:
            w = F.softmax(logits, dim=1)
        return ops.weighted_sum(seq, w), w

    def forward(self, input1, input2):
        """reference layout: input2 (R, C, T)"""
        return self.attend(input1, input2.transpose(1, 2).contiguous())


class PrototypeAttention(nn.Module):
    def __init__(self, in_size, win_size, hidden_size=256, use_gumbel=True, temperature=1, device=None):
        super().__init__()
        self.in_size, self.win_size = in_size, win_size
        self.use_gumbel, self.temperature = use_gumbel, temperature
        g = gaussian_kernel(win_size, 1)
        shift = win_size // 2 - 2
        protos = np.concatenate((g, np.roll(g, -shift), np.roll(g, shift)), axis=1).T
        self.register_buffer("prototype_wts", torch.from_numpy(protos).float())
        self.seq = nn.Sequential(nn.Linear(in_size, hidden_size), nn.ReLU(),
                                 nn.Linear(hidden_size, self.prototype_wts.shape[0]))

    def attend(self, vis, seq):
        h = ops.linear(vis, self.seq[0].weight, self.seq[0].bias, relu=True)
        logits = ops.linear(h, self.seq[2].weight, self.seq[2].bias)
        if self.training and self.use_gumbel:
            m = F.gumbel_softmax(logits, tau=self.temperature, hard=True)
        else:
            m = F.softmax(logits, dim=1)
        w = torch.matmul(m, self.prototype_wts)
        return ops.weighted_sum(seq, w), w

    def forward(self, input1, input2):
        return self.attend(input1, input2.transpose(1, 2).contiguous())
