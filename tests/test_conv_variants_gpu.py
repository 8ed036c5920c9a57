"""GPU parity of the kernel forms that only the backbone engine's autotuner reaches -- called directly through the
C-ABI descriptor entry points (tbn_conv_launch / tbn_conv_launch_pair, include/tbn_hip.h) against fp64 torch
references of the same op (reference layers: nn.Conv2d forward / backward of core/models/bn_inception_audio.py:24-401,
BatchNorm2d.train() + ReLU + MaxPool2d of :21-23,28-34):

  * the data-gradient and eval epilogues of the LDS-halo / LDS-DMA / split-K tile kernels,
  * the BN-backward reduce fused into a data-gradient epilogue (partials vs sum g, sum g * xhat), two producer layers
    with different pitches, every kernel variant and tile,
  * every tile of the parity-phase launch of a stride-2 data gradient (conv_igemm_phases_kernel), with the reduce,
  * two sibling convolutions in ONE launch (conv_pair_igemm_kernel / conv_pair_halo_kernel): statistics, eval and
    reduce epilogues, every variant and tile,
  * the fused BN + ReLU + max-pool kernels of the stem (bn_apply_maxpool, bn_bwd_{reduce,apply}_pooled2x2) on odd and
    even maps against F.max_pool2d(F.relu(bn(x))) autograd in fp64 routed through the product's arg-max.
Tolerance 1e-4 relative to the tensor's maximum (the north star allows 1e-3 end to end); arg-max bytes exact.
"""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from attention_based_tbn_amd._lib import ConvDesc, call, lib, ptr  # noqa: E402

DEV = "cuda"
TOL = 1e-4
HALO, DMA, SK4 = 4, 8, 16
ACCUM, RELU = 1, 2


def st():
    return torch.cuda.current_stream().cuda_stream


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def g(seed):
    return torch.Generator().manual_seed(seed)


def variants_for(k, s, p, w, dgrad):
    """(flags, stages) pairs a launch of this geometry may take: generic 1 / 2 stages always; LDS-halo for 3x3 / stride 1
    / pad 1 maps up to 64 wide; LDS-DMA and the split-K tile kernel for unit-stride launches (a stride-2 DATA gradient is
    the parity-phase launch of the generic kernel)"""
    v = [(0, 1), (0, 2)]
    unit = (s == 1) or not dgrad
    if k == 3 and s == 1 and p == 1 and w <= 64:
        v.append((HALO, 0))
    if unit:
        v += [(DMA, 0), (SK4, 0)]
    return v


def tiles_for(flags):
    mts, nts = (1, 2), ((1, 2) if flags & SK4 else (1, 2, 3, 4))
    return [(mt, nt) for mt in mts for nt in nts]


class Problem:
    """one conv layer with fp64 references of forward, data gradient and the BN-backward reduce of its input side"""

    def __init__(self, n, h, w, cin, cout, k, s, p, seed, split=None):
        self.geom = (n, h, w, cin, cout, k, s, p)
        self.x = torch.randn(n, cin, h, w, generator=g(seed))
        self.wt = torch.randn(cout, cin, k, k, generator=g(seed + 1)) / (cin * k * k) ** 0.5
        xr = self.x.double().requires_grad_(True)
        self.y_ref = F.conv2d(xr, self.wt.double(), None, stride=s, padding=p)
        self.oh, self.ow = self.y_ref.shape[2:]
        self.dy = torch.randn(self.y_ref.shape, generator=g(seed + 2))
        self.y_ref.backward(self.dy.double())
        self.dx_ref = xr.grad                                     # (n, cin, h, w)
        self.xd = nhwc(self.x).to(DEV)
        self.wd = self.wt.permute(0, 2, 3, 1).contiguous().to(DEV)
        self.dyd = nhwc(self.dy).to(DEV)
        # producer side of the data gradient: the cin input channels are the outputs of one or two BN layers whose
        # BN INPUTS were yb (pitches differ: segment 0 sits in a wider buffer) with batch statistics `stats`
        self.split = split if split is not None else cin
        self.yb = torch.randn(n * h * w, cin, generator=g(seed + 3)) * 1.5 + 0.2
        mean = self.yb.double().mean(0)
        rstd = 1.0 / (self.yb.double().var(0, unbiased=False) + 1e-5).sqrt()
        gamma = torch.rand(cin, generator=g(seed + 4)).double() + 0.5
        beta = torch.randn(cin, generator=g(seed + 5)).double() * 0.3
        self.stats = torch.stack([mean, rstd, gamma * rstd, beta - mean * gamma * rstd]).float()     # 4 x cin
        st64 = self.stats.double()
        dz = nhwc(self.dx_ref).reshape(n * h * w, cin)
        mask = (self.yb.double() * st64[2] + st64[3]) > 0          # exact products / sums of fp32 values: the kernel's fmaf sign
        gq = dz * mask
        xhat = (self.yb.double() - st64[0]) * st64[1]
        self.s1_ref, self.s2_ref = gq.sum(0), (gq * xhat).sum(0)
        self.statsd = self.stats.to(DEV).contiguous()
        pad0 = 8
        self.y0 = torch.zeros(n * h * w, self.split + pad0, device=DEV)
        self.y0[:, :self.split] = self.yb[:, :self.split].to(DEV)
        self.y1 = self.yb[:, self.split:].contiguous().to(DEV) if self.split < cin else None
        self.ws = torch.empty(cout * k * k * cin, device=DEV)

    def desc(self, dgrad, out, out_ld, epilogue=0, flags=0, stages=0, stat_partial=None, scale=None, shift=None,
             red_partials=None, bias=None):
        n, h, w, cin, cout, k, s, p = self.geom
        d = ConvDesc()
        d.inp, d.in_ld = (ptr(self.dyd), cout) if dgrad else (ptr(self.xd), cin)
        d.weight, d.bias, d.out, d.out_ld = ptr(self.wd), ptr(bias), out, out_ld
        d.n, d.h, d.w, d.cin, d.cout, d.ksize, d.stride, d.pad = n, h, w, cin, cout, k, s, p
        d.dgrad, d.epilogue, d.flags, d.stages = int(dgrad), epilogue, flags, stages
        d.scale, d.shift, d.stat_partial = ptr(scale), ptr(shift), ptr(stat_partial)
        if red_partials is not None:
            segs = [(self.y0, self.y0.shape[1], 0, self.split)]
            if self.y1 is not None:
                segs.append((self.y1, self.y1.shape[1], self.split, cin - self.split))
            d.nred = len(segs)
            for i, (yt, ld, c0, cc) in enumerate(segs):
                d.red[i].y, d.red[i].y_ld, d.red[i].col_begin, d.red[i].channels = ptr(yt), ld, c0, cc
                d.red[i].stat_offset, d.red[i].partial = c0, ptr(red_partials[i])
            d.red_stats, d.red_stats_stride = ptr(self.statsd), cin
        return d

    def red_buffers(self, d_probe, mt, pair=0):
        rows = lib().tbn_conv_partial_rows(C.byref(d_probe), mt, pair)
        cin = self.geom[3]
        bufs = [torch.full((rows, 2, self.split), float("nan"), device=DEV)]
        if self.split < cin:
            bufs.append(torch.full((rows, 2, cin - self.split), float("nan"), device=DEV))
        return bufs

    def check_reduce(self, bufs, tag):
        s1 = torch.cat([b[:, 0].double().sum(0).cpu() for b in bufs])
        s2 = torch.cat([b[:, 1].double().sum(0).cpu() for b in bufs])
        assert relerr(s1, self.s1_ref) < TOL, tag
        assert relerr(s2, self.s2_ref) < TOL, tag


DGRAD_CASES = [
    # n, h, w, cin, cout, k, s, p, split (column where the second producer layer starts; None = one layer)
    (2, 14, 14, 64, 96, 3, 1, 1, 32),
    (3, 7, 7, 192, 320, 3, 1, 1, 64),
    (2, 9, 11, 96, 160, 3, 1, 1, None),
    (2, 14, 14, 320, 192, 1, 1, 0, 96),          # short-K wide-N fused 1x1 group (inception_3c shape)
    (3, 7, 7, 1056, 352, 1, 1, 0, 448),
    (2, 28, 28, 128, 160, 3, 2, 1, 64),          # stride 2: parity phases (14x14 output)
    (2, 15, 15, 96, 96, 3, 2, 1, None),          # odd map: the four phases have different sizes
    (3, 4, 16, 192, 256, 3, 2, 1, 96),
]


@pytest.mark.parametrize("case", DGRAD_CASES)
def test_dgrad_every_variant_tile_with_fused_bn_backward_reduce(case):
    """data gradient through every kernel variant / tile, plain and accumulating, with and without the fused
    BN-backward reduce (stride-2 cases = every tile and stage count of conv_igemm_phases_kernel)"""
    n, h, w, cin, cout, k, s, p, split = case
    P = Problem(n, h, w, cin, cout, k, s, p, seed=41, split=split)
    for flags, stages in variants_for(k, s, p, P.ow, dgrad=True):
        if flags & HALO and P.ow != w:
            continue
        for mt, nt in tiles_for(flags):
            if 32 * (nt - 1) >= cin:
                continue
            tag = (case, flags, stages, mt, nt)
            # plain data gradient into a channel slice of a wider buffer
            wide = torch.full((n, h, w, cin + 32), 3.0, device=DEV)
            d = P.desc(True, wide.data_ptr() + 16 * 4, cin + 32, flags=flags, stages=stages)
            call("tbn_conv_launch", C.byref(d), mt, nt, ptr(P.ws), st())
            assert relerr(nchw(wide[..., 16:16 + cin]), P.dx_ref) < TOL, tag
            assert float((wide[..., :16] - 3).abs().max()) == 0 and float((wide[..., 16 + cin:] - 3).abs().max()) == 0
            # with the reduce epilogue: same dz, partials = what a pass over (dz, y) would have summed
            dx = torch.empty(n, h, w, cin, device=DEV)
            probe = P.desc(True, ptr(dx), cin, flags=flags, stages=stages)
            bufs = P.red_buffers(probe, mt)
            d = P.desc(True, ptr(dx), cin, flags=flags, stages=stages, red_partials=bufs)
            call("tbn_conv_launch", C.byref(d), mt, nt, ptr(P.ws), st())
            assert relerr(nchw(dx), P.dx_ref) < TOL, tag
            assert all(bool(torch.isfinite(b).all()) for b in bufs), tag      # every partial row was written
            P.check_reduce(bufs, tag)
            # accumulate on top (the reduce then sees the SUM, as it does for a block input with several consumers)
            d = P.desc(True, ptr(dx), cin, flags=flags | ACCUM, stages=stages)
            call("tbn_conv_launch", C.byref(d), mt, nt, ptr(P.ws), st())
            assert relerr(nchw(dx), 2 * P.dx_ref) < TOL, tag


@pytest.mark.parametrize("case", [(2, 14, 14, 64, 96, 3, 1, 1), (3, 7, 7, 192, 320, 3, 1, 1), (2, 9, 11, 32, 160, 3, 1, 1),
                                  (5, 7, 7, 192, 352, 1, 1, 0), (2, 28, 28, 128, 160, 3, 2, 1)])
def test_eval_epilogue_every_variant_tile(case):
    """relu(conv * scale + shift) (running-stat BN folded into the epilogue) through the LDS-halo / LDS-DMA / split-K tile /
    generic kernels, every tile"""
    n, h, w, cin, cout, k, s, p = case
    P = Problem(n, h, w, cin, cout, k, s, p, seed=51)
    sc = torch.rand(cout, generator=g(5)) + 0.5
    sh = torch.randn(cout, generator=g(6))
    ref = F.relu(P.y_ref.detach() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    scd, shd = sc.to(DEV), sh.to(DEV)
    for flags, stages in variants_for(k, s, p, w, dgrad=False):
        for mt, nt in tiles_for(flags):
            if 32 * (nt - 1) >= cout:
                continue
            y = torch.full((n, P.oh, P.ow, cout + 32), 3.0, device=DEV)
            d = P.desc(False, y.data_ptr() + 16 * 4, cout + 32, epilogue=2, flags=flags, stages=stages, scale=scd, shift=shd)
            call("tbn_conv_launch", C.byref(d), mt, nt, 0, st())
            assert relerr(nchw(y[..., 16:16 + cout]), ref) < TOL, (case, flags, stages, mt, nt)
            assert float((y[..., :16] - 3).abs().max()) == 0 and float((y[..., 16 + cout:] - 3).abs().max()) == 0


PAIR_CASES = [
    # two sibling 3x3 layers of a block: (n, h, w) shared; (cin_a, cout_a), (cin_b, cout_b)
    (2, 14, 14, (64, 96), (96, 128)),
    (3, 7, 7, (192, 320), (160, 224)),
    (2, 9, 11, (32, 160), (64, 32)),
    (1, 28, 28, (64, 64), (64, 96)),
]


@pytest.mark.parametrize("case", PAIR_CASES)
def test_pair_launch_every_variant_tile(case):
    """tbn_conv_launch_pair: both members against their own fp64 references; BN-statistics, eval and
    data-gradient-with-reduce epilogues; LDS-halo and generic (1 / 2 stages) members; tiles (1,1) (1,2) (2,1) (2,2)"""
    n, h, w, (ca, oa), (cb, ob) = case
    A = Problem(n, h, w, ca, oa, 3, 1, 1, seed=61, split=32 if ca > 32 else None)
    B = Problem(n, h, w, cb, ob, 3, 1, 1, seed=71, split=None)
    sca, sha = (torch.rand(oa, generator=g(5)) + 0.5).to(DEV), torch.randn(oa, generator=g(6)).to(DEV)
    scb, shb = (torch.rand(ob, generator=g(7)) + 0.5).to(DEV), torch.randn(ob, generator=g(8)).to(DEV)
    for variant in (0, 1, 2):
        for mt in (1, 2):
            for nt in (1, 2):
                tag = (case, variant, mt, nt)
                # training forward: raw conv output + statistics partials of both members
                ya, yb = torch.empty(n, h, w, oa, device=DEV), torch.empty(n, h, w, ob, device=DEV)
                rows = (n * h * w + 128 * mt - 1) // (128 * mt)
                pa, pb = torch.full((rows, 2, oa), float("nan"), device=DEV), torch.full((rows, 2, ob), float("nan"), device=DEV)
                da = A.desc(False, ptr(ya), oa, epilogue=1, stat_partial=pa)
                db = B.desc(False, ptr(yb), ob, epilogue=1, stat_partial=pb)
                assert lib().tbn_conv_partial_rows(C.byref(da), mt, 1) == rows
                call("tbn_conv_launch_pair", C.byref(da), C.byref(db), variant, mt, nt, 0, 0, st())
                for Pm, yy, pp in ((A, ya, pa), (B, yb, pb)):
                    yr = Pm.y_ref.detach()
                    assert relerr(nchw(yy), yr) < TOL, tag
                    assert relerr(pp[:, 0].double().sum(0).cpu(), yr.sum((0, 2, 3))) < TOL, tag
                    assert relerr(pp[:, 1].double().sum(0).cpu(), (yr * yr).sum((0, 2, 3))) < TOL, tag
                # eval forward
                da = A.desc(False, ptr(ya), oa, epilogue=2, scale=sca, shift=sha)
                db = B.desc(False, ptr(yb), ob, epilogue=2, scale=scb, shift=shb)
                call("tbn_conv_launch_pair", C.byref(da), C.byref(db), variant, mt, nt, 0, 0, st())
                for Pm, yy, sc, sh in ((A, ya, sca, sha), (B, yb, scb, shb)):
                    ref = F.relu(Pm.y_ref.detach() * sc.double().cpu().view(1, -1, 1, 1) + sh.double().cpu().view(1, -1, 1, 1))
                    assert relerr(nchw(yy), ref) < TOL, tag
                # data gradients with the fused reduce
                dxa, dxb = torch.empty(n, h, w, ca, device=DEV), torch.empty(n, h, w, cb, device=DEV)
                ba = A.red_buffers(A.desc(True, ptr(dxa), ca), mt, pair=1)
                bb = B.red_buffers(B.desc(True, ptr(dxb), cb), mt, pair=1)
                da = A.desc(True, ptr(dxa), ca, red_partials=ba)
                db = B.desc(True, ptr(dxb), cb, red_partials=bb)
                call("tbn_conv_launch_pair", C.byref(da), C.byref(db), variant, mt, nt, ptr(A.ws), ptr(B.ws), st())
                for Pm, dd, bufs in ((A, dxa, ba), (B, dxb, bb)):
                    assert relerr(nchw(dd), Pm.dx_ref) < TOL, tag
                    Pm.check_reduce(bufs, tag)


def test_pair_launch_rejects_mismatched_members():
    A = Problem(1, 8, 8, 32, 32, 3, 1, 1, seed=1)
    B = Problem(1, 8, 8, 32, 32, 3, 2, 1, seed=2)                  # stride 2: a parity-phase launch, not pairable as dgrad
    ya, yb = torch.empty(1, 8, 8, 32, device=DEV), torch.empty(1, 8, 8, 32, device=DEV)
    da, db = A.desc(True, ptr(ya), 32), B.desc(True, ptr(yb), 32)
    rc = lib().tbn_conv_launch_pair(C.byref(da), C.byref(db), 1, 1, 1, ptr(A.ws), ptr(B.ws), st())
    assert rc < 0 and b"conv_pair" in lib().tbn_last_error()
    sp = torch.empty(1, 2, 32, device=DEV)
    da, db = A.desc(False, ptr(ya), 32, epilogue=1, stat_partial=sp), A.desc(False, ptr(yb), 32, epilogue=0)
    rc = lib().tbn_conv_launch_pair(C.byref(da), C.byref(db), 1, 1, 1, 0, 0, st())
    assert rc < 0 and b"share the epilogue" in lib().tbn_last_error()
    # the split-K tile variant with the statistics epilogue needs an explicit tile (its partial rows are per 32*mt rows)
    rc = lib().tbn_conv2d_fwd(ptr(A.xd), 32, ptr(A.wd), None, ptr(ya), 32, 1, 8, 8, 32, 32, 3, 1, 1, 1, SK4, None, None,
                              ptr(sp), st())
    assert rc < 0 and b"explicit tile" in lib().tbn_last_error()


def _forced_pool(z, argmax, stride, pad, oh, ow):
    """gathers z (n, c, h, w) at the window entry `argmax` (n, oh, ow, c; k = r * 3 + s) selected: a max pool routed
    through ANOTHER run's decisions (differentiable)"""
    n, c, h, w = z.shape
    k = argmax.permute(0, 3, 1, 2).long()                                             # n, c, oh, ow
    oy = torch.arange(oh).view(1, 1, oh, 1)
    ox = torch.arange(ow).view(1, 1, 1, ow)
    iy, ix = oy * stride - pad + k // 3, ox * stride - pad + k % 3
    assert bool(((iy >= 0) & (iy < h) & (ix >= 0) & (ix < w)).all())
    return z.flatten(2).gather(2, (iy * w + ix).flatten(2)).view(n, c, oh, ow)


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 2, 0), (2, 15, 13, 64, 2, 0), (1, 112, 5, 32, 2, 0), (3, 9, 9, 192, 2, 0),
                                  (2, 7, 9, 64, 1, 1), (1, 56, 56, 192, 2, 0)])
def test_bn_relu_maxpool_fused_fwd_bwd(case):
    """the stem's fused BN apply + ReLU + 3x3 max pool (z never written) and its backward on 2x2 input blocks, odd and
    even maps, against F.max_pool2d(F.relu(batch_norm(y))) in fp64: pooled values, running statistics, arg-max (must
    select a window maximum of the fp64 z up to the fp32 error), and dy / dgamma / dbeta with the fp64 autograd routed
    through the product's arg-max"""
    n, h, w, c, s, p = case
    oh = -(-(h + 2 * p - 3) // s) + 1
    ow = -(-(w + 2 * p - 3) // s) + 1
    if (oh - 1) * s >= h + p:
        oh -= 1
    if (ow - 1) * s >= w + p:
        ow -= 1
    y = torch.randn(n, h, w, c, generator=g(1)) * 2 + 0.5
    gamma = torch.rand(c, generator=g(2)) + 0.5
    beta = torch.randn(c, generator=g(3)) * 0.3
    rm, rv = torch.randn(c, generator=g(4)), torch.rand(c, generator=g(5)) + 0.5
    dpool = torch.randn(n, oh, ow, c, generator=g(6))
    P = n * h * w
    ws = torch.empty(lib().tbn_bn_workspace_floats(P, c), device=DEV)
    yd, gd, bd, rmd, rvd = y.to(DEV), gamma.to(DEV), beta.to(DEV), rm.to(DEV), rv.to(DEV)
    mean, rstd, scale, shift = (torch.empty(c, device=DEV) for _ in range(4))
    pooled = torch.full((n, oh, ow, c + 8), 3.0, device=DEV)
    am = torch.full((n, oh, ow, c), 255, dtype=torch.uint8, device=DEV)
    call("tbn_bn_relu_maxpool_train_fwd", ptr(yd), n, h, w, c, ptr(gd), ptr(bd), ptr(rmd), ptr(rvd), 0.1, 1e-5, ptr(mean),
         ptr(rstd), ptr(scale), ptr(shift), pooled.data_ptr() + 16, c + 8, ptr(am), oh, ow, s, p, ptr(ws), st())
    # fp64 reference
    yr = nchw(y).double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rmr, rvr = rm.double().clone(), rv.double().clone()
    z = F.relu(F.batch_norm(yr, rmr, rvr, gr, br, True, 0.1, 1e-5))
    pool_ref = F.max_pool2d(z, 3, s, p, ceil_mode=True)
    assert pool_ref.shape[2:] == (oh, ow)
    assert relerr(nchw(pooled[..., 4:4 + c]), pool_ref.detach()) < TOL
    assert float((pooled[..., :4] - 3).abs().max()) == 0 and float((pooled[..., 4 + c:] - 3).abs().max()) == 0
    assert relerr(rmd, rmr) < TOL and relerr(rvd, rvr) < TOL
    amc = am.cpu()
    assert int(amc.max()) < 9
    routed = _forced_pool(z, amc, s, p, oh, ow)
    # the selected entry IS a window maximum (up to fp32 rounding of z)
    assert float((routed.detach() - pool_ref.detach()).abs().max()) <= 1e-5 * float(pool_ref.detach().abs().max())
    routed.backward(nchw(dpool).double())
    dpd = dpool.to(DEV)
    dy, dg, db = torch.empty(n, h, w, c, device=DEV), torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    call("tbn_bn_relu_maxpool_train_bwd", ptr(dpd), c, ptr(am), ptr(yd), n, h, w, c, oh, ow, s, p, ptr(mean), ptr(rstd),
         ptr(scale), ptr(shift), ptr(dy), ptr(dg), ptr(db), ptr(ws), st())
    assert relerr(nchw(dy), yr.grad) < TOL
    assert relerr(dg, gr.grad) < TOL and relerr(db, br.grad) < TOL
    # in place (the engine converts y to dy in place)
    y2 = yd.clone()
    call("tbn_bn_relu_maxpool_train_bwd", ptr(dpd), c, ptr(am), ptr(y2), n, h, w, c, oh, ow, s, p, ptr(mean), ptr(rstd),
         ptr(scale), ptr(shift), ptr(y2), ptr(dg), ptr(db), ptr(ws), st())
    assert torch.equal(y2, dy)
