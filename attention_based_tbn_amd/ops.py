"""torch.autograd wrappers over the C-ABI operators (HIP only -- no CPU fallback).

PyTorch is plumbing here: it owns device memory, the stream and the autograd tape; every
numerical op of the path runs in libtbn_hip.so.
"""
import torch

from ._lib import call, lib, ptr, stream_ptr, TbnHipError


def _need_cuda(t, what):
    if not t.is_cuda:
        raise TbnHipError(f"{what}: tensor is on {t.device}; the TBN hot path only runs on an MI355X "
                          "(HIP) device -- there is no CPU fallback")


def _f32c(t):
    t = t.contiguous()
    if t.dtype != torch.float32:
        t = t.float()
    return t


def _pad_rows(w, b, mult):
    """pad the output dim of a linear layer (rows of w) up to a multiple of `mult` with zeros"""
    n = w.shape[0]
    npad = (n + mult - 1) // mult * mult
    if npad == n:
        return w, b, n
    w = torch.cat([w, w.new_zeros(npad - n, w.shape[1])], 0)
    if b is not None:
        b = torch.cat([b, b.new_zeros(npad - n)], 0)
    return w, b, n


class _LinearFn(torch.autograd.Function):
    """out = x @ w.T + b (optional ReLU);  x (M,K) K%32==0, w (N,K) N%32==0."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        _need_cuda(x, "linear")
        x, w = _f32c(x), _f32c(w)
        M, K = x.shape
        N = w.shape[0]
        out = torch.empty(M, N, device=x.device, dtype=torch.float32)
        call("tbn_linear_fwd", ptr(x), K, ptr(w), ptr(b.contiguous()) if b is not None else 0, ptr(out), N, M, K, N,
             int(relu), stream_ptr())
        ctx.relu = relu
        ctx.has_bias = b is not None
        ctx.save_for_backward(x, w, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w, out = ctx.saved_tensors
        dout = _f32c(dout)
        M, K = x.shape
        N = w.shape[0]
        st = stream_ptr()
        if ctx.relu:
            g = torch.empty_like(dout)
            call("tbn_relu_mask_bwd", ptr(dout), ptr(out), 0, ptr(g), dout.numel(), st)
            dout = g
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, device=x.device, dtype=torch.float32)
            ws = torch.empty(N * K, device=x.device, dtype=torch.float32)
            call("tbn_linear_dgrad", ptr(dout), N, ptr(w), ptr(dx), K, M, K, N, 0, ptr(ws), st)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty(N, K, device=x.device, dtype=torch.float32)
            db = torch.empty(N, device=x.device, dtype=torch.float32) if ctx.has_bias else None
            nws = lib().tbn_linear_wgrad_workspace_floats(M, K, N)
            ws = torch.empty(max(nws, 1), device=x.device, dtype=torch.float32)
            call("tbn_linear_wgrad", ptr(dout), N, ptr(x), K, ptr(dw), ptr(db), M, K, N, ptr(ws), st)
        return dx, dw, db, None


def linear(x, weight, bias=None, relu=False):
    """nn.Linear / Conv1d(k=1) on the HIP GEMM.  Pads N to a multiple of 32 and K to a multiple of
    32 with zeros when needed (the padding is sliced off again; autograd sees through it)."""
    K = x.shape[1]
    if K % 32:
        kp = (K + 31) // 32 * 32
        x = torch.cat([x, x.new_zeros(x.shape[0], kp - K)], 1)
        weight = torch.cat([weight, weight.new_zeros(weight.shape[0], kp - K)], 1)
    w, b, n = _pad_rows(weight, bias, 32)
    out = _LinearFn.apply(x, w, b, relu)
    return out if n == w.shape[0] else out[:, :n]


class _SpatialMeanFn(torch.autograd.Function):
    """x NHWC (N,H,W,C) -> (N,C) or, freq_only, (N,W,C)   [reference bn_inception.py:16-35]"""

    @staticmethod
    def forward(ctx, x, freq_only):
        _need_cuda(x, "spatial_mean")
        x = _f32c(x)
        N, H, W, Cc = x.shape
        out = torch.empty((N, W, Cc) if freq_only else (N, Cc), device=x.device, dtype=torch.float32)
        call("tbn_spatial_mean_fwd", ptr(x), Cc, ptr(out), Cc, N, H, W, Cc, int(freq_only), stream_ptr())
        ctx.shape, ctx.freq_only = (N, H, W, Cc), freq_only
        return out

    @staticmethod
    def backward(ctx, dout):
        N, H, W, Cc = ctx.shape
        dout = _f32c(dout)
        dx = torch.empty(ctx.shape, device=dout.device, dtype=torch.float32)
        call("tbn_spatial_mean_bwd", ptr(dout), Cc, ptr(dx), Cc, N, H, W, Cc, int(ctx.freq_only), stream_ptr())
        return dx, None


def spatial_mean(x_nhwc, freq_only=False):
    return _SpatialMeanFn.apply(x_nhwc, freq_only)


class _PEConcatFn(torch.autograd.Function):
    """feat (R,T,C) + pe (PD,T) -> (R,T,C+PD padded to out_ld)   [reference attention.py:36-41]"""

    @staticmethod
    def forward(ctx, feat, pe, out_ld):
        _need_cuda(feat, "pe_concat")
        feat = _f32c(feat)
        R, T, Cc = feat.shape
        out = torch.empty(R, T, out_ld, device=feat.device, dtype=torch.float32)
        call("tbn_pe_concat_fwd", ptr(feat), Cc, ptr(_f32c(pe)), ptr(out), out_ld, R, T, Cc, pe.shape[0], stream_ptr())
        ctx.C = Cc
        return out

    @staticmethod
    def backward(ctx, dout):
        return dout[:, :, :ctx.C].contiguous(), None, None


def pe_concat(feat, pe, out_ld):
    return _PEConcatFn.apply(feat, pe, out_ld)


class _GroupNormFn(torch.autograd.Function):
    """nn.GroupNorm(groups, C) on (R,T,C) rows  [reference model.py:62-67 pe.2]"""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps):
        _need_cuda(x, "groupnorm")
        x = _f32c(x)
        R, T, Cc = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(R * groups, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        call("tbn_groupnorm_fwd", ptr(x), ptr(y), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), R, T, Cc, groups,
             float(eps), stream_ptr())
        ctx.groups = groups
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        dy = _f32c(dy)
        R, T, Cc = x.shape
        st = stream_ptr()
        dx = torch.empty_like(x)
        dgp = torch.empty(R, Cc, device=x.device, dtype=torch.float32)
        dbp = torch.empty_like(dgp)
        call("tbn_groupnorm_bwd", ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), ptr(dgp), ptr(dbp),
             R, T, Cc, ctx.groups, st)
        dg = torch.empty(Cc, device=x.device, dtype=torch.float32)
        db = torch.empty_like(dg)
        call("tbn_colsum", ptr(dgp), Cc, ptr(dg), R, Cc, st)
        call("tbn_colsum", ptr(dbp), Cc, ptr(db), R, Cc, st)
        return dx, dg, db, None, None


def group_norm(x, gamma, beta, groups, eps=1e-5):
    return _GroupNormFn.apply(x, gamma, beta, groups, eps)


class _MHAq1Fn(torch.autograd.Function):
    """attention core for one query per sample: q (R,E), kv (R,T,2E) -> ctx (R,E), avg weights (R,T)"""

    @staticmethod
    def forward(ctx, q, kv, drop_mask, heads):
        _need_cuda(q, "mha_q1")
        q, kv = _f32c(q), _f32c(kv)
        R, E = q.shape
        T = kv.shape[1]
        scale = float(E // heads) ** -0.5
        out = torch.empty(R, E, device=q.device, dtype=torch.float32)
        probs = torch.empty(2, R, heads, T, device=q.device, dtype=torch.float32)
        avg = torch.empty(R, T, device=q.device, dtype=torch.float32)
        call("tbn_mha_q1_fwd", ptr(q), ptr(kv), ptr(drop_mask), ptr(out), ptr(probs), ptr(avg), R, T, E, heads, scale,
             stream_ptr())
        ctx.heads, ctx.scale = heads, scale
        ctx.save_for_backward(q, kv, probs, drop_mask)
        return out, avg

    @staticmethod
    def backward(ctx, dctx, davg):
        q, kv, probs, drop = ctx.saved_tensors
        R, E = q.shape
        T = kv.shape[1]
        dctx = _f32c(dctx) if dctx is not None else torch.zeros_like(q)
        davg = _f32c(davg) if davg is not None else None
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        call("tbn_mha_q1_bwd", ptr(dctx), ptr(davg), ptr(q), ptr(kv), ptr(probs), ptr(drop), ptr(dq), ptr(dkv), R, T, E,
             ctx.heads, ctx.scale, stream_ptr())
        return dq, dkv, None, None


def mha_q1(q, kv, drop_mask, heads):
    return _MHAq1Fn.apply(q, kv, drop_mask, heads)


class _WeightedSumFn(torch.autograd.Function):
    """fixed attention: out[r] = sum_t feat[r,t] * w[r,t]   [reference model.py:224-228]"""

    @staticmethod
    def forward(ctx, feat, w):
        _need_cuda(feat, "weighted_sum")
        feat, w = _f32c(feat), _f32c(w)
        R, T, Cc = feat.shape
        out = torch.empty(R, Cc, device=feat.device, dtype=torch.float32)
        call("tbn_weighted_sum_fwd", ptr(feat), ptr(w), ptr(out), Cc, R, T, Cc, stream_ptr())
        ctx.save_for_backward(w, feat if w.requires_grad else None)
        ctx.shape = (R, T, Cc)
        return out

    @staticmethod
    def backward(ctx, dout):
        w, feat = ctx.saved_tensors
        R, T, Cc = ctx.shape
        dout = _f32c(dout)
        df = dw = None
        if ctx.needs_input_grad[0]:
            df = torch.empty(R, T, Cc, device=dout.device, dtype=torch.float32)
            call("tbn_weighted_sum_bwd", ptr(dout), Cc, ptr(w), ptr(df), R, T, Cc, stream_ptr())
        if ctx.needs_input_grad[1]:
            # (R,T) weights of the learnt unimodal / prototype variants: tiny, elementwise torch
            dw = (feat * dout.unsqueeze(1)).sum(2)
        return df, dw


def weighted_sum(feat, w):
    return _WeightedSumFn.apply(feat, w)


class _SegmentMeanFn(torch.autograd.Function):
    """temporal consensus: (B*n, C) -> (B, C)   [reference model.py:178-203]"""

    @staticmethod
    def forward(ctx, x, b, n):
        _need_cuda(x, "segment_mean")
        x = _f32c(x)
        Cc = x.shape[1]
        out = torch.empty(b, Cc, device=x.device, dtype=torch.float32)
        call("tbn_segment_mean_fwd", ptr(x), ptr(out), b, n, Cc, stream_ptr())
        ctx.dims = (b, n, Cc)
        return out

    @staticmethod
    def backward(ctx, dout):
        b, n, Cc = ctx.dims
        dout = _f32c(dout)
        dx = torch.empty(b * n, Cc, device=dout.device, dtype=torch.float32)
        call("tbn_segment_mean_bwd", ptr(dout), ptr(dx), b, n, Cc, stream_ptr())
        return dx, None, None


def segment_mean(x, b, n):
    return _SegmentMeanFn.apply(x, b, n)


class _MulMaskFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask):
        _need_cuda(x, "mul_mask")
        x = _f32c(x)
        y = torch.empty_like(x)
        call("tbn_mul_mask", ptr(x), ptr(mask), ptr(y), x.numel(), stream_ptr())
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = _f32c(dy)
        dx = torch.empty_like(dy)
        call("tbn_mul_mask", ptr(dy), ptr(mask), ptr(dx), dy.numel(), stream_ptr())
        return dx, None


class _DropoutFn(torch.autograd.Function):
    """y = x * mask, mask = rnd >= p ? 1 / (1 - p) : 0 in ONE launch (was: compare, cast, scale, multiply)"""

    @staticmethod
    def forward(ctx, x, rnd, p):
        _need_cuda(x, "dropout")
        x = _f32c(x)
        y = torch.empty_like(x)
        mask = torch.empty_like(x)
        call("tbn_dropout_fwd", ptr(x), ptr(rnd), float(p), ptr(y), ptr(mask), x.numel(), stream_ptr())
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = _f32c(dy)
        dx = torch.empty_like(dy)
        call("tbn_mul_mask", ptr(dy), ptr(mask), ptr(dx), dy.numel(), stream_ptr())
        return dx, None, None


def dropout(x, p, training):
    """nn.Dropout: the uniform draw comes from torch's RNG (plumbing), mask and multiply are one HIP launch."""
    if not training or p <= 0:
        return x
    return _DropoutFn.apply(x, torch.rand_like(x), p)


class _CatPadFn(torch.autograd.Function):
    """rows of several parameters stacked and zero-padded to a multiple of `mult` rows -- the classifier's per-key
    nn.Linear weights / biases as ONE GEMM operand (reference model.py:365-386 keeps one nn.Linear per key).  The stacked
    copy lives in `cache` and is rebuilt only when a parameter changed (its autograd version counter: the optimiser's
    in-place update bumps it); backward hands each parameter its rows of the gradient."""

    @staticmethod
    def forward(ctx, cache, mult, *params):
        key = tuple((p.data_ptr(), p._version) for p in params)
        rows = [p.shape[0] for p in params]
        if cache.get("key") != key:
            n = sum(rows)
            npad = (n + mult - 1) // mult * mult
            buf = cache.get("buf")
            shape = (npad,) + tuple(params[0].shape[1:])
            if buf is None or tuple(buf.shape) != shape or buf.device != params[0].device:
                buf = cache["buf"] = torch.zeros(shape, device=params[0].device, dtype=params[0].dtype)
            o = 0
            for p, r in zip(params, rows):
                buf[o:o + r].copy_(p.detach())
                o += r
            cache["key"] = key
        ctx.rows = rows
        return cache["buf"].detach()     # a fresh tensor object per call (shares storage and version counter with the cache)

    @staticmethod
    def backward(ctx, g):
        out, o = [], 0
        for r in ctx.rows:
            out.append(g[o:o + r])
            o += r
        return (None, None) + tuple(out)


def cat_pad_rows(cache, mult, *params):
    return _CatPadFn.apply(cache, mult, *params)


class _CeHeadsFn(torch.autograd.Function):
    """mean cross entropy of H heads over one (B, ld) score matrix -> H scalar losses (tbn_ce_heads_fwd / _bwd)"""

    @staticmethod
    def forward(ctx, scores, heads, *labels):
        _need_cuda(scores, "cross_entropy_heads")
        import ctypes as C
        B, ld = scores.shape
        H = len(heads)
        col0 = (C.c_int * H)(*[h[0] for h in heads])
        ncls = (C.c_int * H)(*[h[1] for h in heads])
        labs = [l.contiguous() for l in labels]
        for l in labs:
            if l.dtype != torch.int64 or not l.is_cuda or l.numel() != B:
                raise TbnHipError("cross_entropy_heads: labels must be int64 GPU tensors of the batch size")
        lptr = (C.c_void_p * H)(*[l.data_ptr() for l in labs])
        rowloss = torch.empty(H * B, device=scores.device, dtype=torch.float32)
        loss = torch.empty(H, device=scores.device, dtype=torch.float32)
        dsc = torch.zeros_like(scores)
        call("tbn_ce_heads_fwd", ptr(scores), ld, B, H, col0, ncls, lptr, ptr(rowloss), ptr(loss), ptr(dsc), stream_ptr())
        ctx.heads = heads
        ctx.save_for_backward(dsc)
        return tuple(loss[h] for h in range(H))

    @staticmethod
    def backward(ctx, *grads):
        import ctypes as C
        (dsc,) = ctx.saved_tensors
        B, ld = dsc.shape
        H = len(ctx.heads)
        up = torch.stack([g if g is not None else dsc.new_zeros(()) for g in grads]).float().contiguous()
        col0 = (C.c_int * H)(*[h[0] for h in ctx.heads])
        ncls = (C.c_int * H)(*[h[1] for h in ctx.heads])
        out = torch.zeros_like(dsc)
        call("tbn_ce_heads_bwd", ptr(dsc), ld, B, H, col0, ncls, ptr(up), ptr(out), stream_ptr())
        return (out, None) + (None,) * H


def cross_entropy_heads(scores, heads, labels):
    """scores (B, ld) contiguous; heads = [(first column, classes)]; labels = [int64 (B,)] -> tuple of 0-dim losses
    (nn.CrossEntropyLoss() with its defaults, per head: a label of -100 is ignored -- no loss, no gradient, the mean runs
    over the other rows).  Any other label outside [0, classes) gives NaN, not an error."""
    return _CeHeadsFn.apply(scores, tuple(heads), *labels)


def dropout_mask(shape, p, training, device):
    if not training or p <= 0:
        return None
    return (torch.rand(shape, device=device) >= p).float() / (1.0 - p)
