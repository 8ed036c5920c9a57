"""GPU parity of the backbone engine and of the whole TBN path.

* vs the golden fixtures written by the UNMODIFIED reference (tests/golden/model_*.npz): logits,
  attention weights, losses, gradients, BN running statistics;
* vs the CPU oracle on seeded inputs (per-layer gradients of a backbone, full-resolution inputs);
* size-independent properties at BASELINE.json's full input sizes (224x224 / 256x256).
Tolerance: the north star's 1e-3 relative (fp32); most checks are far inside it.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.util import build_oracle, load_case, prior_target, assert_close, rel_err  # noqa: E402

DEV = torch.device("cuda")
EVAL_CASES = ["cfg1_audio_only", "cfg2_rgb_only", "cfg3_rgb_audio_mha_T8", "cfg3_rgb_audio_mha_T13",
              "cfg4_all_noattn", "cfg5_all_mha_eval", "fixed_attn", "unimodal_attn", "proto_attn"]
TRAIN_CASES = ["train_cfg4_all_noattn", "train_cfg3_mha"]


def build_product(cfg, modality, meta):
    from oracle.fill import fill_state_dict
    from attention_based_tbn_amd.core.models import build_model
    model, crit, _ = build_model(cfg, modality, DEV)
    sd = model.state_dict()
    assert [[k, list(v.shape)] for k, v in sd.items()] == meta["keys"]      # checkpoint-key contract
    model.load_state_dict(fill_state_dict(sd, meta["fill_seed"]))
    return model, crit


def to_dev(d):
    return {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in d.items()}


@pytest.mark.parametrize("name", EVAL_CASES)
def test_eval_forward_matches_reference_golden(name):
    cfg, modality, meta, data, inp, target = load_case(name)
    model, crit = build_product(cfg, modality, meta)
    model.eval()
    with torch.no_grad():
        out = model(to_dev(inp))
    for k, v in out.items():
        want = data["out_" + k]
        assert tuple(v.shape) == want.shape, k
        assert_close(v, want, k)
    tgt = {"class": to_dev(target["class"])}
    for ep in (0, 20):
        loss, bs = model.get_loss(crit, tgt, out, epoch=ep)
        for k, v in loss.items():
            want = float(data[f"loss_ep{ep}_{k}"])
            assert abs(float(torch.as_tensor(v).detach()) - want) < 1e-3 * max(1.0, abs(want)), (ep, k)


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_train_step_matches_reference_golden(name):
    cfg, modality, meta, data, inp, target = load_case(name)
    model, crit = build_product(cfg, modality, meta)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    B, n = inp[modality[0]].shape[:2]
    tgt = {"class": to_dev(target["class"])}
    if cfg.model.attention.enable and cfg.model.attention.use_prior:
        tgt["weights"] = prior_target(cfg, B, n).to(DEV)
    model.train()
    for ep in (0, 20):
        model.load_state_dict(sd)
        model.zero_grad()
        out = model(to_dev(inp))
        loss, bs = model.get_loss(crit, tgt, out, epoch=ep)
        loss["total"].backward()
        for k, v in out.items():
            assert_close(v, data[f"ep{ep}_out_{k}"], (ep, k))
        for k, v in loss.items():
            want = float(data[f"ep{ep}_loss_{k}"])
            assert abs(float(torch.as_tensor(v).detach()) - want) < 1e-3 * max(1.0, abs(want)), (ep, k)
        grads = reference_named_grads(model)
        truth = fp64_truth(cfg, modality, meta, inp, target, tgt, ep)
        checked = 0
        for k in data:
            if not k.startswith(f"ep{ep}_grad_"):
                continue
            name_ = k[len(f"ep{ep}_grad_"):]
            want = torch.from_numpy(data[k])
            got = grads[name_].cpu()
            if float(truth[name_].abs().max()) < 1e-9:
                # conv bias in front of a batch-stat BN: analytically zero (fp64: ~1e-16), the reference
                # holds fp32 rounding noise, the HIP path returns exact zeros
                assert float(got.abs().max()) < 1e-5 and float(want.abs().max()) < 1e-4, name_
            else:
                # training-mode backward is ill-conditioned in fp32 (ReLU / max-pool decisions flip on
                # rounding): the reference's own fp32 gradients sit ~1e-2 from the fp64 truth.  The HIP
                # path must be as close to the truth as the reference is (x4), or within 1e-3 of it.
                e_hip, e_ref = l2_err(got, truth[name_]), l2_err(want, truth[name_])
                assert e_hip < max(2e-2, 4 * e_ref), (ep, name_, e_hip, e_ref)
                assert cosine(got, truth[name_]) > 0.999, (ep, name_)
            checked += 1
        assert checked >= 1
        gn = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values()))
        want = float(data[f"ep{ep}_gradnorm"])
        assert abs(float(gn) - want) < 2e-2 * want
    st = model.state_dict()
    for k in data:
        if k.startswith("post_"):
            assert rel_err(st[k[5:]].double().cpu(), data[k]) < 1e-3, k


def test_audio_dropout_branches_match_reference_golden():
    """row a11 (reference model.py:215-222): both branches of the host-side audio-dropout draw, training mode.
    Dropped: zero audio feature, no gradient reaches the audio backbone, its BN running statistics still advance."""
    cfg, modality, meta, data, inp, target = load_case("train_audio_dropout")
    model, crit = build_product(cfg, modality, meta)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    tgt = {"class": to_dev(target["class"])}
    model.train()
    for branch, seed in meta["np_seeds"].items():
        model.load_state_dict(sd)
        model.zero_grad()
        np.random.seed(seed)
        out = model(to_dev(inp))
        loss, _ = model.get_loss(crit, tgt, out, epoch=0)
        loss["total"].backward()
        for k, v in out.items():
            assert_close(v, data[f"{branch}_out_{k}"], (branch, k))
        for k, v in loss.items():
            want = float(data[f"{branch}_loss_{k}"])
            assert abs(float(torch.as_tensor(v).detach()) - want) < 1e-3 * max(1.0, abs(want)), (branch, k)
        grads = reference_named_grads(model)
        for k in data:
            if not k.startswith(f"{branch}_hasgrad_"):
                continue
            name_ = k[len(f"{branch}_hasgrad_"):]
            assert (name_ in grads) == bool(data[k]), (branch, name_)
            if bool(data[k]):
                want = torch.from_numpy(data[f"{branch}_grad_{name_}"])
                assert l2_err(grads[name_], want) < 2e-2 and cosine(grads[name_].cpu(), want) > 0.999, (branch, name_)
        key = "Base_Audio.conv1_7x7_s2_bn.running_mean"
        assert rel_err(model.state_dict()[key].cpu(), data[f"{branch}_post_{key}"]) < 1e-3, branch
        assert (model.Base_Audio.flat_weight.grad is None) == (branch == "drop")


def test_crop_repeat_eval_matches_reference_golden():
    """row a11 (reference model.py:243-248): RGB rows = 2 x audio rows (multi-crop testing) -> tiled audio feature"""
    cfg, modality, meta, data, inp, target = load_case("crop_repeat_eval")
    model, crit = build_product(cfg, modality, meta)
    model.eval()
    with torch.no_grad():
        out = model(to_dev(inp))
    for k, v in out.items():
        assert tuple(v.shape) == data["out_" + k].shape, k
        assert_close(v, data["out_" + k], k)
    loss, bs = model.get_loss(crit, {"class": to_dev(target["class"])}, out, epoch=0)
    for k, v in loss.items():
        want = float(data[f"loss_ep0_{k}"])
        assert abs(float(torch.as_tensor(v).detach()) - want) < 1e-3 * max(1.0, abs(want)), k


def test_product_bninception_factory_matches_reference():
    """row a2 (reference bn_inception.py:38-107) on the PRODUCT factory: Audio conv1 = channel mean of the RGB filter,
    key reconciliation, the kinetics 10-channel conv1, `last_linear` removal -- against the reference-written fixture"""
    import os
    from oracle.fill import pretrained_pair
    from tests.util import GOLDEN
    from attention_based_tbn_amd.core.models import bninception
    d = np.load(os.path.join(GOLDEN, "factory_audio.npz"))
    pre = pretrained_pair(int(d["seed"]))
    m = bninception(1, "Audio", pretrained="imagenet", is_audio=True, attend=True, state_dict=pre["imagenet"]).to(DEV)
    assert not hasattr(m, "last_linear") and not bool(d["has_last_linear"])
    assert m.is_audio and m.attend and m.feature_size == 1024
    assert torch.equal(m.state_dict()["conv1_7x7_s2.weight"].cpu(), torch.from_numpy(d["conv1_w"]))
    m.eval()
    with torch.no_grad():
        y = m(torch.from_numpy(d["x"].astype(np.float32)).to(DEV))
    assert tuple(y.shape) == d["y"].shape == (1, 1024, 1, 8)
    assert rel_err(y.cpu(), d["y"]) < 1e-3
    mf = bninception(10, "Flow", pretrained="kinetics", state_dict=pre["kinetics"])
    w = mf.state_dict()["conv1_7x7_s2.weight"]
    assert tuple(w.shape) == (64, 10, 7, 7)
    assert abs(float(w.double().sum()) - float(d["flow_conv1_w_sum"])) < 1e-9
    mr = bninception(3, "RGB", pretrained="imagenet", state_dict=pre["imagenet"])
    assert torch.equal(mr.state_dict()["conv1_7x7_s2.weight"], pre["imagenet"]["conv1_7x7_s2.weight"])
    assert not hasattr(mr, "last_linear") and not hasattr(mf, "last_linear")
    with pytest.raises(RuntimeError):       # strict load (reference :101): a 3-channel filter does not fit the flow stem
        bninception(10, "Flow", pretrained="kinetics", state_dict=pre["imagenet"])


# Per-layer bound of the operating-point gradient checks.  Two fp32 evaluations of this training graph differ in the
# last-bit ReLU / max-pool decisions they take, and one flipped decision moves every upstream gradient (DESIGN.md section 2:
# torch's own fp32 CPU gradients sit 1e-3 ... 7e-2 from an fp64 run of the small cases; with the decisions pinned the
# product's gradients agree with fp64 to 1e-4 on every layer, test_backbone_all_layer_grads_forced_decisions).  A layer
# whose kernel is WRONG (a dropped split-K slab, a transposed tile, a mis-ordered tap) shows relative L2 of order 1 and a
# cosine far below 0.99, so the bound separates the two by an order of magnitude.
# Observed (round 5, R = 96 and R = 192): the noise spreads evenly -- every layer sits at 1.1e-2 ... 1.6e-2 / cosine 0.99988 ...
# 0.99994, the same figures as the per-backbone aggregates.
PER_LAYER_L2 = 3e-2
PER_LAYER_COS = 0.999


def l2_err(a, b):
    """relative L2 error: robust to the single-element ReLU / max-pool decision flips of fp32"""
    a, b = torch.as_tensor(a).double().flatten().cpu(), torch.as_tensor(b).double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def per_layer_grad_parity(grads, ograds, modality, tag, l2_bound, cos_bound):
    """EVERY conv / BN parameter gradient of every backbone against the oracle's own (round-4 verdict: the aggregate
    check lets a layer whose gradient is wrong pass as long as it carries < 3 % of the norm).  Per tensor: relative L2
    and cosine; the worst five are printed.  Conv biases sit in front of a batch-statistics BN: their gradient is
    analytically zero, the product returns exact zeros and the oracle fp32 rounding noise -- compared as absolute values
    against the scale of the layer's BN-bias gradient.  Returns the number of tensors compared."""
    rows, bias_bad = [], []
    for m in modality:
        keys = [k for k in sorted(grads) if k.startswith(f"Base_{m}.")]
        assert keys and set(keys) <= set(ograds), (m, set(keys) - set(ograds))
        for k in keys:
            got, want = grads[k].detach().cpu(), ograds[k].detach()
            assert got.shape == want.shape, k
            if k.endswith(".bias") and "_bn." not in k:
                # exact zeros here; the oracle holds the rounding noise of its sums (observed <= 6e-6 on conv1): judged
                # against the size of the same layer's weight gradient
                ref = float(ograds[k.replace(".bias", ".weight")].abs().max())
                if not (float(got.abs().max()) == 0.0 and float(want.abs().max()) <= 1e-3 * max(ref, 1e-6)):
                    bias_bad.append((k, float(got.abs().max()), float(want.abs().max()), ref))
                continue
            rows.append((l2_err(got, want), cosine(got, want), k))
    rows.sort(reverse=True)
    for e, c, k in rows[:5]:
        print("%s per-layer parity: %-55s relative L2 %.2e  cosine %.6f" % (tag, k, e, c))
    bad = [(k, e, c) for e, c, k in rows if not (e < l2_bound and c > cos_bound)]
    assert not bad, bad[:8]
    assert not bias_bad, bias_bad[:8]
    return len(rows)


def fp64_truth(cfg, modality, meta, inp, target, tgt, ep):
    """gradients of the same step from the oracle run in float64 (named like the reference)"""
    oracle, ocrit = build_oracle(cfg, modality, meta)
    oracle = oracle.double().train()
    t64 = {"class": target["class"]}
    if "weights" in tgt:
        t64["weights"] = tgt["weights"].double().cpu()
    out = oracle({k: v.double() for k, v in inp.items()})
    loss, _ = oracle.get_loss(ocrit, t64, out, epoch=ep)
    loss["total"].backward()
    return {k: p.grad for k, p in oracle.named_parameters() if p.grad is not None}


def reference_named_grads(model):
    """gradients under the reference's parameter names (backbone grads live in flat tensors)"""
    out = {}
    for name, p in model.named_parameters():
        if p.grad is None or ".flat_" in name or ".bn_weight_" in name or ".bn_bias_" in name:
            continue
        out[name] = p.grad
    for m in model.modality:
        base = getattr(model, "Base_" + m)
        n0 = base.first_bn_channels
        for lname, L in base._layers.items():
            a, b_ = L["c_off"], L["c_off"] + L["cout"]
            if base.flat_weight.grad is not None:
                nw = L["cout"] * L["k"] * L["k"] * L["cin"]
                out[f"Base_{m}.{lname}.weight"] = base.flat_weight.grad[L["w_off"]:L["w_off"] + nw].view(
                    L["cout"], L["k"], L["k"], L["cin"]).permute(0, 3, 1, 2)
                out[f"Base_{m}.{lname}.bias"] = base.flat_bias.grad[a:b_]
            first = b_ <= n0
            gw = base.bn_weight_first.grad if first else base.bn_weight_rest.grad
            gb = base.bn_bias_first.grad if first else base.bn_bias_rest.grad
            if gw is not None:
                o = 0 if first else n0
                out[f"Base_{m}.{lname}_bn.weight"] = gw[a - o:b_ - o]
                out[f"Base_{m}.{lname}_bn.bias"] = gb[a - o:b_ - o]
    return out


@pytest.mark.parametrize("cin_hw", [(3, 64, 64), (10, 96, 64), (1, 128, 256), (3, 224, 224), (3, 97, 97), (10, 70, 129)])
def test_backbone_all_layer_grads_vs_oracle(cin_hw):
    """every conv / BN parameter gradient of one backbone + running stats, train mode.
    (audio uses H=128: with H=64 the last maps are 2 rows high, the 3x3/pad-1 average pool then
    yields bit-identical rows and the following max pool ties EXACTLY -- gradient routing through an
    exact tie is decided by last-bit noise in any fp32/fp64 reference, so it cannot be compared.)"""
    from oracle.bninception import BNInception as OBN
    from oracle.fill import fill_state_dict
    from attention_based_tbn_amd.core.models.bn_inception import BNInception
    cin, H, W = cin_hw
    N = 2 if H >= 224 else 3
    import copy
    ora = OBN(1000, cin)
    sd = fill_state_dict(ora.state_dict(), 42)
    ora.load_state_dict(sd)
    o64 = copy.deepcopy(ora).double()
    net = BNInception(1000, cin).to(DEV)
    net.load_state_dict(sd)
    x = torch.randn(N, cin, H, W, generator=torch.Generator().manual_seed(1))
    ora.train(), net.train(), o64.train()
    yo = ora(x)
    dy = torch.randn(yo.shape, generator=torch.Generator().manual_seed(2))
    yo.backward(dy)
    y64 = o64(x.double())
    y64.backward(dy.double())
    y = net(x.to(DEV))
    y.backward(dy.to(DEV))
    assert rel_err(y.detach().cpu(), yo.detach()) < 1e-3
    assert rel_err(y.detach().cpu(), y64.detach()) < max(1e-4, 4 * rel_err(yo.detach(), y64.detach()))
    op, p64 = dict(ora.named_parameters()), dict(o64.named_parameters())
    n0 = net.first_bn_channels
    for lname, L in net._layers.items():
        nw = L["cout"] * L["k"] * L["k"] * L["cin"]
        gw = net.flat_weight.grad[L["w_off"]:L["w_off"] + nw].view(L["cout"], L["k"], L["k"], L["cin"]).permute(
            0, 3, 1, 2).cpu()
        a, b_ = L["c_off"], L["c_off"] + L["cout"]
        gg = (net.bn_weight_first.grad[a:b_] if b_ <= n0 else net.bn_weight_rest.grad[a - n0:b_ - n0]).cpu()
        gb = (net.bn_bias_first.grad[a:b_] if b_ <= n0 else net.bn_bias_rest.grad[a - n0:b_ - n0]).cpu()
        for got, key in ((gw, lname + ".weight"), (gg, lname + "_bn.weight"), (gb, lname + "_bn.bias")):
            truth = p64[key].grad
            # yardstick: how far torch's own fp32 CPU backward is from the fp64 truth for this tensor
            e_hip, e_cpu = l2_err(got, truth), l2_err(op[key].grad, truth)
            # floor: ONE ReLU / max-pool decision that flips on a last-bit difference (either in the HIP or in
            # the CPU fp32 run) moves the gradient of its layer AND of every layer before it: by ~1e-2 in relative
            # L2 on the larger inputs, by up to ~4e-2 on the 64x64 case whose 14x14-stage maps have only 48 pixels
            # (seen at inception_4d_double_3x3_1 while the layers after it sat at 2e-4).  Genuine indexing /
            # accumulation bugs showed up as >= 5e-2 on every input size and fail the cosine test too.
            floor = 4.4e-2 if H * W <= 64 * 64 else 2e-2
            cos_min = 0.999
            if H % 2 or W % 2:
                # odd sizes exercise the partial ceil-mode pool windows (fused stem pools, gather backward); their
                # 4x4 / 3x3 late maps make a single flipped decision worth up to ~5e-2, an indexing bug >> 1e-1
                floor, cos_min = 8e-2, 0.995
            assert e_hip < max(floor, 4 * e_cpu), (key, e_hip, e_cpu)
            assert cosine(got, truth) > cos_min, key
    so, sn = ora.state_dict(), net.state_dict()
    for k in so:
        if "running" in k or "num_batches" in k:
            assert rel_err(sn[k].double().cpu(), so[k].double()) < 1e-3, k
    # eval mode too (running-stat BN folded into the conv epilogue), incl. attended-audio pooling
    ora.eval(), net.eval()
    with torch.no_grad():
        assert rel_err(net(x.to(DEV)).cpu(), ora(x)) < 1e-3
        if cin == 1:
            ora.is_audio = ora.attend = net.is_audio = net.attend = True
            a, b2 = net(x.to(DEV)).cpu(), ora(x)
            assert a.shape == b2.shape and rel_err(a, b2) < 1e-3
            f = net.features(x.to(DEV))
            assert rel_err(f.cpu(), ora.features(x)) < 1e-3 and rel_err(net.logits(f).cpu(), b2) < 1e-3


class _ForcedReLU(torch.nn.Module):
    """ReLU whose on/off decisions come from another run (here: the HIP product's z > 0)"""

    def __init__(self, mask_rows):
        super().__init__()
        self.mask_rows = mask_rows            # (N*H*W, C) bool, NHWC pixel order

    def forward(self, x):
        n, c, h, w = x.shape
        return x * self.mask_rows.view(n, h, w, c).permute(0, 3, 1, 2).to(x.dtype)


class _ForcedMaxPool(torch.nn.Module):
    """3x3 max pool that routes through the window entry ANOTHER run selected: the entry of the product's own fp32
    input that equals the product's pooled value (first in scan order, the engine's tie rule)"""

    def __init__(self, pool, x32, out32):
        super().__init__()
        import torch.nn.functional as F
        s = pool.stride if isinstance(pool.stride, int) else pool.stride[0]
        p = pool.padding if isinstance(pool.padding, int) else pool.padding[0]
        n, c, h, w = x32.shape
        oh, ow = out32.shape[2:]
        pb, pr = max(0, (oh - 1) * s + 3 - h - p), max(0, (ow - 1) * s + 3 - w - p)
        xp = F.pad(x32, (p, pr, p, pb), value=float("-inf"))
        win = F.unfold(xp, 3, stride=s).view(n, c, 9, oh * ow)
        eq = win == out32.reshape(n, c, 1, oh * ow)
        assert bool(eq.any(2).all()), "pooled value not found in its window"
        k = eq.float().argmax(2)                                          # first match
        oy = torch.arange(oh).repeat_interleave(ow).view(1, 1, -1)
        ox = torch.arange(ow).repeat(oh).view(1, 1, -1)
        self.idx = (oy * s - p + k // 3) * w + (ox * s - p + k % 3)
        self.out_hw = (oh, ow)

    def forward(self, x):
        n, c = x.shape[:2]
        return x.flatten(2).gather(2, self.idx).view(n, c, *self.out_hw)


@pytest.mark.parametrize("cin_hw", [(3, 96, 96), (10, 96, 96), (1, 96, 96), (10, 70, 129)])
def test_backbone_all_layer_grads_forced_decisions(cin_hw):
    """see _forced_decision_grads; N = 2 frames (224 x 224 runs at the benchmarked R = 96 below)"""
    _forced_decision_grads(cin_hw, 2)


@pytest.mark.parametrize("cin_hw", [(10, 224, 224), (1, 256, 256)])
def test_flow_audio_backbones_R96_tuned_plan_forced_decision_grads(cin_hw):
    """the same check for the other two backbones of the benchmarked step: Flow (row-run stem, K = 512) and Audio (256 x 256
    spectrograms: 8 x 8 final maps, other tiles / split-K plans than the 224 x 224 backbones) at R = 96 frames"""
    _forced_decision_grads(cin_hw, 96, expect_tuned_variants="halo+pairs")


def test_rgb_backbone_R96_tuned_plan_forced_decision_grads():
    """Round-5 verdict item 3(i): the forced-decision fp64 gradient check AT THE OPERATING POINT bench.py times -- the RGB
    backbone at R = 96 frames of 224 x 224 with the plan the autotuner picks for that size (split-K slab counts of the
    weight gradients, sibling-pair launches, 32-row split-K tiles and parity-phase tiles only appear there): every one of
    the 207 conv / BN parameter gradients within 5e-4 relative L2 of the fp64 oracle that takes the product's ReLU /
    max-pool decisions.  The operating-point tests compare against the oracle's own fp32 step at 3e-2 (decision noise);
    this closes the gap to the 1e-4 the small cases show."""
    import time
    t0 = time.time()
    _forced_decision_grads((3, 224, 224), 96, expect_tuned_variants=True)
    print("R = 96 forced-decision check: %.1f s in all" % (time.time() - t0))


def _forced_decision_grads(cin_hw, N, expect_tuned_variants=False):
    """EVERY conv / BN parameter gradient of a backbone within 1e-3 relative L2 of an fp64 reference (observed: 1e-5 to
    1e-4).  The fp32 training-mode backward is ill conditioned: some ReLU input or max-pool runner-up always lies
    within the fp32 error of its decision boundary (tests/golden/make_tight_seeds.py: best safety over 300 seeds 1.3
    fp32-error units), and ONE flipped decision in a 3x3 map moves every upstream gradient by percents -- in torch's own
    fp32 CPU path exactly as here.  So the fp64 oracle is made to take the PRODUCT's decisions: its ReLUs multiply by the
    product's (z > 0) masks, its max pools route through the entry the product selected (both read back from the
    engine workspace), and everything else -- convolutions, batch statistics, the whole backward -- is the oracle's own
    fp64 arithmetic.  What remains is the arithmetic error of the 69 weight-gradient / data-gradient / BN-backward
    steps: merged 1x1 groups, reduce-in-epilogue, parity-phase strided gradients, LDS-halo 3x3 kernels AND the shipped
    fused stem kernels (conv1 / conv2_3x3: BN apply + ReLU + max pool in one pass, BN backward on 2x2 blocks) included:
    their z is never written, so the test rebuilds it bit-exactly as relu(fma(y, scale, shift)) from the engine's y and
    batch-statistics coefficients (tbn_backbone_tensor_info kinds 1 and 4) and finds the pool route from the pooled values."""
    import copy
    import ctypes as C
    from oracle.bninception import BNInception as OBN
    from oracle.fill import fill_state_dict
    from attention_based_tbn_amd._lib import call
    from attention_based_tbn_amd.core.models.bn_inception import BNInception
    cin, H, W = cin_hw
    ora = OBN(1000, cin)
    sd = fill_state_dict(ora.state_dict(), 42)
    ora.load_state_dict(sd)
    o64 = copy.deepcopy(ora).double().train()
    net = BNInception(1000, cin).to(DEV)
    net.load_state_dict(sd)
    net.train()
    x = torch.randn(N, cin, H, W, generator=torch.Generator().manual_seed(5))
    y = net(x.to(DEV))
    plan = net._plans[(N, H, W)]
    if expect_tuned_variants:
        # what the tuned plan launches at this size (tbn_backbone_launch_info: [halo, mt, nt, stages, pair, ...] forward and
        # data gradient per conv): the variants that only win at R = 96 must really be in the plan that is checked
        info = (C.c_int * 16)()
        variants, pairs = set(), 0
        for lname in net._layers:
            call("tbn_backbone_launch_info", plan.handle, lname.encode(), 1, info)
            variants.update({("fwd", info[0]), ("dgrad", info[8])})
            pairs += info[4] + info[12]
        print("R = %d plan %s: kernel variants %s, %d pair decisions" % (N, plan.fingerprint(), sorted(variants), pairs))
        need = {1, 3} if expect_tuned_variants is True else {1}
        assert {v for _, v in variants} >= need and pairs > 0, (variants, pairs)      # LDS-halo, (split-K tiles,) sibling pairs
    ws = plan.pool[0][0].view(torch.float32)

    def tensor(name, kind):
        off, rows, cols, ld = C.c_long(), C.c_int(), C.c_int(), C.c_int()
        call("tbn_backbone_tensor_info", plan.handle, name.encode(), kind, C.byref(off), C.byref(rows), C.byref(cols),
             C.byref(ld))
        return torch.as_strided(ws, (rows.value, cols.value), (ld.value, 1), off.value).cpu()

    def nchw(rows, like_hw):
        h, w = like_hw
        return rows.view(N, h, w, -1).permute(0, 3, 1, 2).contiguous()

    from attention_based_tbn_amd._lib import TbnHipError
    fused = []

    def z_of(name):
        """z = relu(bn(conv)) as the product formed it.  A conv whose max pool runs inside its BN apply never writes z:
        the engine computes fmaxf(fmaf(y, scale, shift), 0) per element in registers -- rebuilt here from its y and its
        coefficients (the products and sums of fp32 values are exact in fp64; one rounding to fp32, like the fma)"""
        try:
            return tensor(name, 0)
        except TbnHipError:
            fused.append(name)
            yv, co = tensor(name, 1).double(), tensor(name, 4).double()
            return torch.clamp_min(yv * co[2] + co[3], 0.0).float()

    # the product's decisions -> the fp64 oracle
    names = [n for n, m in o64.named_children() if isinstance(m, torch.nn.Conv2d)]
    relus = [n for n, m in o64.named_children() if isinstance(m, torch.nn.ReLU)]
    assert len(names) == len(relus) == 69
    for cname, rname in zip(names, relus):
        setattr(o64, rname, _ForcedReLU(z_of(cname) > 0))
    assert fused == ["conv1_7x7_s2", "conv2_3x3"], fused      # the shipped plan fuses exactly the two stem pools
    shapes = {}
    hooks = [m.register_forward_pre_hook(lambda mod, inp, k=k: shapes.__setitem__(k, inp[0].shape[2:]))
             for k, m in ora.named_children() if isinstance(m, torch.nn.MaxPool2d)]
    hooks += [m.register_forward_hook(lambda mod, inp, out, k=k: shapes.__setitem__(k + "_out", out.shape[2:]))
              for k, m in ora.named_children() if isinstance(m, torch.nn.MaxPool2d)]
    with torch.no_grad():
        ora.train()(x)
    for h in hooks:
        h.remove()
    pools = {
        "pool1_3x3_s2": (z_of("conv1_7x7_s2"), tensor("conv2_3x3_reduce", 3)),
        "pool2_3x3_s2": (z_of("conv2_3x3"), tensor("inception_3a_1x1", 3)),
        "inception_3c_pool": (tensor("inception_3c_3x3_reduce", 3), tensor("inception_4a_1x1", 3)[:, 256:576]),
        "inception_4e_pool": (tensor("inception_4e_3x3_reduce", 3), tensor("inception_5a_1x1", 3)[:, 448:1056]),
        "inception_5b_pool": (tensor("inception_5b_1x1", 3), tensor("inception_5b_pool_proj", 3)),
    }
    for pname, (xin, xout) in pools.items():
        setattr(o64, pname, _ForcedMaxPool(getattr(ora, pname), nchw(xin, shapes[pname]), nchw(xout, shapes[pname + "_out"])))
    y64 = o64(x.double())
    dy = torch.randn(y64.shape, generator=torch.Generator().manual_seed(2))
    y64.backward(dy.double())
    y.backward(dy.to(DEV))
    assert rel_err(y.detach().cpu(), y64.detach()) < 1e-3
    p64 = dict(o64.named_parameters())
    n0 = net.first_bn_channels
    worst = (0.0, None)
    for lname, L in net._layers.items():
        nw = L["cout"] * L["k"] * L["k"] * L["cin"]
        gw = net.flat_weight.grad[L["w_off"]:L["w_off"] + nw].view(L["cout"], L["k"], L["k"], L["cin"]).permute(0, 3, 1, 2)
        a, b_ = L["c_off"], L["c_off"] + L["cout"]
        gg = net.bn_weight_first.grad[a:b_] if b_ <= n0 else net.bn_weight_rest.grad[a - n0:b_ - n0]
        gb = net.bn_bias_first.grad[a:b_] if b_ <= n0 else net.bn_bias_rest.grad[a - n0:b_ - n0]
        for got, key in ((gw, lname + ".weight"), (gg, lname + "_bn.weight"), (gb, lname + "_bn.bias")):
            e = l2_err(got, p64[key].grad)
            if e > worst[0]:
                worst = (e, key)
            assert e < 5e-4, (key, e)         # the north star asks 1e-3; observed worst over all five cases: 1.05e-4
    print(f"forced-decision gradient case {cin_hw}: worst relative L2 error {worst[0]:.2e} at {worst[1]}")


def test_config4_full_batch_train_step_vs_oracle():
    """Parity AT THE BENCHMARKED OPERATING POINT (round-2 verdict): BASELINE config 4 at B = 32 clips x 3 segments, i.e.
    R = 96 frames per backbone -- the plans, per-layer autotune choices and split-K plans `bench.py` times (LDS-DMA,
    sibling pairs, split-K tiles, 160-wide weight-gradient tiles only win at this size) -- one training step (reference
    loop body core/tools/train.py:76-81) against the CPU oracle with the same weights and full-size synthetic inputs:
    logits, losses and every BN running statistic within 1e-3; the gradient norm and each backbone's conv weight
    gradients against the oracle's own fp32 step (relative L2 < 3e-2, cosine > 0.999; observed 0.6e-2 ... 1.2e-2: two
    fp32 runs of this graph differ by their last-bit ReLU / max-pool decisions -- with the decisions pinned the same
    gradients agree to 1e-4, test_backbone_all_layer_grads_forced_decisions -- and the reference's own fp32 gradients sit
    1e-3 ... 7e-2 from an fp64 run on the small golden cases).  The oracle step takes ~85 s on the GPU box's host cores."""
    import ctypes as C
    import time
    from attention_based_tbn_amd._lib import lib
    cfg, modality, meta, _, _, _ = load_case("train_cfg4_all_noattn")
    assert modality == ["RGB", "Flow", "Audio"]
    B, n = 32, 3
    g = torch.Generator().manual_seed(0)
    mean = torch.tensor([0.408, 0.459, 0.502]).view(1, 1, 3, 1, 1)
    inp = {"RGB": torch.rand(B, n, 3, 224, 224, generator=g) - mean,
           "Flow": torch.rand(B, n, 10, 224, 224, generator=g) - 0.502,
           "Audio": (torch.randn(B, n, 1, 256, 256, generator=g) * 3 - 6).clamp_(-13.8155, 8.0)}
    target = {"class": {"verb": torch.randint(0, 125, (B,), generator=g), "noun": torch.randint(0, 352, (B,), generator=g)}}
    model, crit = build_product(cfg, modality, meta)
    model.train()
    L = lib()
    L.tbn_profile_reset()
    model.zero_grad()
    dinp = to_dev(inp)
    tgt = {"class": to_dev(target["class"])}
    out = model(dinp)                       # first use of the shape: per-layer autotune, as in bench.py's priming step
    loss, _ = model.get_loss(crit, tgt, out, epoch=0)
    loss["total"].backward()
    # second step with the launches bracketed: which kernel families the tuned plan runs (weights unchanged: same step)
    sd_post = {k: v.clone() for k, v in model.state_dict().items()}
    model.load_state_dict(fill_sd(model, meta))
    model.zero_grad()
    L.tbn_profile_enable(1)
    out = model(dinp)
    loss, _ = model.get_loss(crit, tgt, out, epoch=0)
    loss["total"].backward()
    torch.cuda.synchronize()
    L.tbn_profile_enable(0)
    fam = {}
    name = C.create_string_buffer(160)
    for i in range(L.tbn_profile_num_entries()):
        cnt, ms, fl = C.c_long(), C.c_double(), C.c_double()
        L.tbn_profile_entry(i, name, 160, C.byref(cnt), C.byref(ms), C.byref(fl))
        f_ = name.value.decode().split("<")[0]
        fam[f_] = fam.get(f_, 0) + cnt.value
    L.tbn_profile_reset()
    print("kernel families at R = 96:", fam)
    for need in ("conv_igemm_kernel", "conv_halo_kernel", "conv_igemm_phases_kernel", "conv_sk4_kernel", "conv_wgrad_kernel"):
        assert fam.get(need, 0) > 0, (need, fam)
    assert fam.get("conv_pair_igemm_kernel", 0) + fam.get("conv_pair_halo_kernel", 0) > 0, fam
    # the two product steps started from the same weights: deterministic kernels -> identical statistics updates
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd_post[k]), k

    t0 = time.time()
    oracle, ocrit = build_oracle(cfg, modality, meta)
    oracle.train()
    oout = oracle(inp)
    oloss, _ = oracle.get_loss(ocrit, target, oout, epoch=0)
    oloss["total"].backward()
    print("oracle step at B = 32: %.1f s on %d threads" % (time.time() - t0, torch.get_num_threads()))
    for k in ("verb", "noun"):
        assert_close(out[k], oout[k], k)
    for k, v in oloss.items():
        want = float(torch.as_tensor(v).detach())
        assert abs(float(torch.as_tensor(loss[k]).detach()) - want) < 1e-3 * max(1.0, abs(want)), k
    osd = oracle.state_dict()
    checked = 0
    for k, v in model.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert rel_err(v.double().cpu(), osd[k].double()) < 1e-3, k
            checked += 1
    assert checked == 2 * 69 * 3
    grads = reference_named_grads(model)
    ograds = {k: p.grad for k, p in oracle.named_parameters() if p.grad is not None}
    worst = (0.0, None)
    for m in modality:
        got = torch.cat([grads[k].reshape(-1).cpu() for k in sorted(grads) if k.startswith(f"Base_{m}.") and k.endswith(".weight")
                         and "_bn." not in k])
        want = torch.cat([ograds[k].reshape(-1) for k in sorted(grads) if k.startswith(f"Base_{m}.") and k.endswith(".weight")
                          and "_bn." not in k])
        e = l2_err(got, want)
        print("R = 96 parity: %s conv weight gradients relative L2 %.2e, cosine %.6f" % (m, e, cosine(got, want)))
        worst = max(worst, (e, m))
        assert e < 3e-2 and cosine(got, want) > 0.999, (m, e)
    # ... and layer by layer: 69 conv weights + the 2 trainable first-BN tensors per backbone (partialbn), the bound set from
    # the fp32-vs-fp32 decision noise this graph shows per layer (PER_LAYER_L2 below)
    n = per_layer_grad_parity(grads, ograds, modality, "R = 96", PER_LAYER_L2, PER_LAYER_COS)
    assert n == 3 * (69 + 2), n
    gn = torch.sqrt(sum((v.double() ** 2).sum() for v in grads.values()))
    ogn = torch.sqrt(sum((v.double() ** 2).sum() for k, v in ograds.items() if k in grads))
    assert abs(float(gn) - float(ogn)) < 2e-2 * float(ogn), (float(gn), float(ogn))
    print("R = 96 parity: worst backbone weight-gradient relative L2 %.2e (%s), grad norm %.5f vs %.5f" %
          (worst[0], worst[1], float(gn), float(ogn)))


def test_config2_full_batch_train_step_vs_oracle_lone_backbone_streams():
    """Round-5 verdict item 3(ii): BASELINE config 2 (RGB only, attention off) AT ITS OPERATING POINT -- B = 32 clips x 3
    segments = 96 frames -- with the stream policy the model ships for a LONE backbone (weight gradients on the aux
    stream, the 3x3 / pool_proj chain of every inception block on the side stream: reference dataflow
    core/models/bn_inception_audio.py:437-1003), one training step (core/tools/train.py:76-81) against the CPU oracle:
    logits, losses and all 138 BN running statistics within 1e-3, every conv / BN gradient by the config-4 rule (oracle's
    own fp32 step, relative L2 < 3e-2, cosine > 0.999 per layer).  Until now that policy was oracle-checked only by
    transitivity (bit-identity with the one-chain program at N = 6, tests/test_branch_gpu.py)."""
    import time
    cfg, modality, meta, _, _, _ = load_case("cfg2_rgb_only")
    assert modality == ["RGB"]
    B, n = 32, 3
    g = torch.Generator().manual_seed(2)
    mean = torch.tensor([0.408, 0.459, 0.502]).view(1, 1, 3, 1, 1)
    inp = {"RGB": torch.rand(B, n, 3, 224, 224, generator=g) - mean}
    target = {"class": {"verb": torch.randint(0, 125, (B,), generator=g), "noun": torch.randint(0, 352, (B,), generator=g)}}
    model, crit = build_product(cfg, modality, meta)
    model.train()
    base = model.Base_RGB
    assert base.use_aux_stream and base.use_branch_streams          # what TBNModel sets for one modality
    dinp, tgt = to_dev(inp), {"class": to_dev(target["class"])}
    model.zero_grad()
    out = model(dinp)                                               # first use: autotune
    loss, _ = model.get_loss(crit, tgt, out, epoch=0)
    loss["total"].backward()
    torch.cuda.synchronize()
    from attention_based_tbn_amd._lib import lib
    plan = base._plans[(B * n, 224, 224)]
    assert lib().tbn_backbone_num_streams(plan.handle) == 2 and base._side_streams and base._aux_streams   # both streams were in use
    first = {k: v.clone() for k, v in reference_named_grads(model).items()}
    sd_post = {k: v.clone() for k, v in model.state_dict().items()}
    # the same step again from the same weights: two chains + the weight-gradient stream must reproduce themselves bit for bit
    model.load_state_dict(fill_sd(model, meta))
    model.zero_grad()
    out = model(dinp)
    loss, _ = model.get_loss(crit, tgt, out, epoch=0)
    loss["total"].backward()
    torch.cuda.synchronize()
    grads = reference_named_grads(model)
    assert all(torch.equal(grads[k], first[k]) for k in first) and all(torch.equal(v, sd_post[k]) for k, v in model.state_dict().items())
    t0 = time.time()
    oracle, ocrit = build_oracle(cfg, modality, meta)
    oracle.train()
    oout = oracle(inp)
    oloss, _ = oracle.get_loss(ocrit, target, oout, epoch=0)
    oloss["total"].backward()
    print("oracle step of config 2 at B = 32: %.1f s on %d threads" % (time.time() - t0, torch.get_num_threads()))
    for k in ("verb", "noun"):
        assert_close(out[k], oout[k], k)
    for k, v in oloss.items():
        want = float(torch.as_tensor(v).detach())
        assert abs(float(torch.as_tensor(loss[k]).detach()) - want) < 1e-3 * max(1.0, abs(want)), k
    osd = oracle.state_dict()
    checked = 0
    for k, v in model.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert rel_err(v.double().cpu(), osd[k].double()) < 1e-3, k
            checked += 1
    assert checked == 2 * 69
    ograds = {k: p.grad for k, p in oracle.named_parameters() if p.grad is not None}
    nl = per_layer_grad_parity(grads, ograds, modality, "config 2, R = 96, branch + aux streams", PER_LAYER_L2, PER_LAYER_COS)
    assert nl == 69 + 2, nl
    for k in ("classifier.verb.weight", "classifier.noun.weight", "classifier.noun.bias"):
        assert l2_err(grads[k], ograds[k]) < 1e-3, (k, l2_err(grads[k], ograds[k]))


def fill_sd(model, meta):
    from oracle.fill import fill_state_dict
    return fill_state_dict(model.state_dict(), meta["fill_seed"])


def test_autotune_choices_are_kept_per_mode():
    """a validation pass at the training shape tunes the EVAL epilogue of the same plan: it must not overwrite the
    launch choices (tile, kernel variant, stages, sibling pairing; forward and data gradient) training was tuned to
    (round-2 advisor finding: shared plan fields).  Checked through tbn_backbone_launch_info, and by the training step
    reproducing its own output bit for bit afterwards (every kernel is deterministic for a given choice)."""
    import ctypes as C
    from attention_based_tbn_amd._lib import call
    from attention_based_tbn_amd.core.models.bn_inception import BNInception
    torch.manual_seed(3)
    net = BNInception(1000, 3).to(DEV).train()
    N, H, W = 6, 96, 96
    x = torch.randn(N, 3, H, W, device=DEV)

    def train_step():
        net.zero_grad()
        rm = net.running_mean.clone()
        out = net(x)
        out.square().mean().backward()
        net.running_mean.copy_(rm)            # same statistics state for the repeat
        return out.detach().clone(), net.flat_weight.grad.clone()

    def choices(training):
        plan = net._plans[(N, H, W)]
        out = {}
        for name in net._layers:
            buf = (C.c_int * 16)()
            call("tbn_backbone_launch_info", plan.handle, name.encode(), training, buf)
            out[name] = tuple(buf)
        return out

    y0, g0 = train_step()
    before = choices(1)
    assert any(v[8:] != (0,) * 8 for v in before.values())          # data-gradient choices exist in training mode
    net.eval()
    with torch.no_grad():
        net(x)                                                       # tunes the eval epilogue of the same plan
    ev = choices(0)
    assert all(v[8:] == (0,) * 8 for v in ev.values())              # no data gradient in eval mode
    net.train()
    assert choices(1) == before
    y1, g1 = train_step()
    assert torch.equal(y0, y1) and torch.equal(g0, g1)


def test_full_size_properties_config4_shapes():
    """BASELINE.json full input sizes (3x224x224 RGB, 10x224x224 flow, 256x256 spectrogram):
    eval logits of a clip do not depend on its batch neighbours, equal-segment clips reproduce
    the single-segment logits (temporal consensus is a mean), train step yields finite grads."""
    from attention_based_tbn_amd.config import load_config, get_modality
    from attention_based_tbn_amd.core.models import build_model
    cfg = load_config(["model.attention.enable=False", "data.audio.audio_length=1.279", "data.sampling=async"])
    modality = get_modality(cfg)
    torch.manual_seed(0)
    model, crit, _ = build_model(cfg, modality, DEV)
    with torch.no_grad():   # non-trivial BN statistics
        for m in modality:
            b = getattr(model, "Base_" + m)
            b.running_var.uniform_(0.5, 1.5)
            b.running_mean.normal_(0, 0.1)
    g = torch.Generator(device="cuda").manual_seed(0)
    B, n = 4, 3
    inp = {"RGB": torch.rand(B, n, 3, 224, 224, device=DEV, generator=g) - 0.45,
           "Flow": torch.rand(B, n, 10, 224, 224, device=DEV, generator=g) - 0.5,
           "Audio": torch.randn(B, n, 1, 256, 256, device=DEV, generator=g) * 3 - 6}
    model.eval()
    with torch.no_grad():
        full = model(inp)
        one = model({k: v[1:2] for k, v in inp.items()})
        rep = model({k: v[1:2, :1].expand(-1, n, -1, -1, -1).contiguous() for k, v in inp.items()})
        single = model({k: v[1:2, :1].contiguous() for k, v in inp.items()})
    for k in ("verb", "noun"):
        assert full[k].shape == (B, cfg.model.num_classes[k])
        assert rel_err(full[k][1:2].cpu(), one[k].cpu()) < 1e-5
        assert rel_err(rep[k].cpu(), single[k].cpu()) < 1e-5
    model.train()
    out = model(inp)
    tgt = {"class": {"verb": torch.randint(0, 125, (B,), device=DEV), "noun": torch.randint(0, 352, (B,), device=DEV)}}
    loss, bs = model.get_loss(crit, tgt, out, 0)
    loss["total"].backward()
    assert bs == B and torch.isfinite(loss["total"])
    for name, p in model.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), name
    # partialbn: only the first BN of each backbone trains
    assert model.Base_RGB.bn_weight_rest.grad is None and model.Base_RGB.bn_weight_first.grad is not None


@pytest.mark.parametrize("ov,sizes", [
    (["model.attention.enable=False", "data.audio.audio_length=1.279", "data.sampling=async",
      "model.fusion_dropout=0"], {"RGB": (3, 224, 224), "Flow": (10, 224, 224), "Audio": (1, 256, 256)}),
    (["data.flow.enable=False", "model.attention.use_entropy=True", "model.fusion_dropout=0",
      "model.attention.attn_dropout=0.0"], {"RGB": (3, 224, 224), "Audio": (1, 256, 420)}),
])
def test_full_size_train_and_eval_vs_oracle(ov, sizes):
    """the whole path at BASELINE.json's full input sizes (config 4: three modalities, 1.279 s audio; config 3 as the
    reference README runs it: RGB + 2.1 s audio = 256x420 spectrogram, T = 13, MHA + entropy loss) against the CPU
    oracle carrying the same weights: training-mode logits / attention weights / losses (batch-stat BN, dropout off),
    the BN running statistics after the step, then eval logits.  1e-3 relative, the north star's tolerance."""
    from attention_based_tbn_amd.config import load_config, get_modality
    cfg = load_config(ov)
    modality = get_modality(cfg)
    from attention_based_tbn_amd.core.models import build_model
    probe, _, _ = build_model(cfg, modality, DEV)
    meta = {"keys": [[k, list(v.shape)] for k, v in probe.state_dict().items()], "fill_seed": 11}
    del probe
    model, crit = build_product(cfg, modality, meta)
    oracle, ocrit = build_oracle(cfg, modality, meta)
    g = torch.Generator().manual_seed(5)
    B, n = 2, 3
    inp = {m: (torch.randn(B, n, *sizes[m], generator=g) * 3 - 6) if m == "Audio"
           else (torch.rand(B, n, *sizes[m], generator=g) - 0.45) for m in modality}
    target = {"class": {"verb": torch.randint(0, 125, (B,), generator=g), "noun": torch.randint(0, 352, (B,), generator=g)}}
    model.train()
    oracle.train()
    out = model(to_dev(inp))
    want = oracle(inp)
    assert set(out) == set(want)
    for k in want:
        assert_close(out[k], want[k], k)
    for ep in (0, 20):
        loss, bs = model.get_loss(crit, {"class": to_dev(target["class"])}, out, epoch=ep)
        wl, wbs = oracle.get_loss(ocrit, target, want, epoch=ep)
        assert bs == wbs and set(loss) == set(wl)
        for k in wl:
            a, b = float(torch.as_tensor(loss[k]).detach()), float(torch.as_tensor(wl[k]).detach())
            assert abs(a - b) < 1e-3 * max(1.0, abs(b)), (ep, k)
    st, wst = model.state_dict(), oracle.state_dict()
    for k in wst:
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert rel_err(st[k].cpu(), wst[k]) < 1e-3, k
    model.eval()
    oracle.eval()
    with torch.no_grad():
        out, want = model(to_dev(inp)), oracle(inp)
    for k in want:
        assert_close(out[k], want[k], k)


def test_state_dict_roundtrip_with_oracle_checkpoint():
    """a reference-format checkpoint (here: the oracle's state_dict) loads and round-trips"""
    cfg, modality, meta, data, inp, target = load_case("cfg5_all_mha_eval")
    oracle, _ = build_oracle(cfg, modality, meta)
    from attention_based_tbn_amd.core.models import build_model
    model, _, _ = build_model(cfg, modality, DEV)
    model.load_state_dict(oracle.state_dict())
    back = model.state_dict()
    for k, v in oracle.state_dict().items():
        assert torch.equal(back[k].cpu(), v), k


def test_eval_chunking_config5_shape():
    """config-5 style eval (25 segments): more frames than one engine call takes (eval_chunk) must give
    the same logits as a single call, and the MHA weights keep the reference shape (B*n, 1, T)"""
    from attention_based_tbn_amd.config import load_config, get_modality
    from attention_based_tbn_amd.core.models import build_model
    cfg = load_config(["data.audio.audio_length=1.279", "data.flow.enable=False"])
    modality = get_modality(cfg)
    torch.manual_seed(1)
    model, crit, _ = build_model(cfg, modality, DEV)
    with torch.no_grad():
        for m in modality:
            b = getattr(model, "Base_" + m)
            b.running_var.uniform_(0.5, 1.5)
            b.running_mean.normal_(0, 0.1)
    g = torch.Generator(device="cuda").manual_seed(3)
    B, n = 3, 25                                  # 75 frames per modality
    inp = {"RGB": torch.rand(B, n, 3, 224, 224, device=DEV, generator=g) - 0.45,
           "Audio": torch.randn(B, n, 1, 256, 256, device=DEV, generator=g) * 3 - 6}
    model.eval()
    with torch.no_grad():
        one = model(inp)
        for m in modality:
            getattr(model, "Base_" + m).eval_chunk = 32     # 75 frames -> 32 + 32 + 11
        chunked = model(inp)
    assert one["weights"].shape == (B * n, 1, 8)
    assert abs(float(one["weights"].sum(-1).mean()) - 1.0) < 1e-5
    for k in ("verb", "noun", "weights"):
        assert rel_err(chunked[k].cpu(), one[k].cpu()) < 1e-5, k


def _lone_backbone_for_capture():
    from attention_based_tbn_amd.core.models.bn_inception import BNInception
    torch.manual_seed(21)
    net = BNInception(1000, 3).to(DEV).train()
    net.use_aux_stream = True                # the default of a lone backbone: weight gradients on a second stream
    x = torch.randn(4, 3, 64, 64, device=DEV)

    def step():
        net.zero_grad(set_to_none=True)
        out = net(x)
        out.square().mean().backward()
        return out

    return net, x, step


def test_stem_weight_gradients_last_is_bit_identical():
    """TBN_BACKBONE_STEM_WGRAD_LAST (include/tbn_hip.h; what TBNModel sets for multi-modality models): the weight gradients of
    conv2_3x3 / conv2_3x3_reduce issued after conv1's pooled BN backward instead of before it -- a pure reordering of
    launches on one stream (the deferred launches read dy and z buffers nothing overwrites in between): outputs, every
    gradient and the running statistics must be bit-identical, for every stem form (RGB / audio space-to-depth, flow row runs)"""
    from attention_based_tbn_amd.core.models.bn_inception import BNInception
    for cin, N, H, W in ((3, 4, 224, 224), (10, 3, 96, 96), (1, 3, 128, 256)):
        torch.manual_seed(cin)
        net = BNInception(1000, cin).to(DEV)
        net.set_bn_trainable(True, True)
        net.use_aux_stream = net.use_branch_streams = False
        x = torch.randn(N, cin, H, W, device=DEV)
        rm0, rv0 = net.running_mean.clone(), net.running_var.clone()

        def step(flag):
            net.train()
            net.stem_wgrad_last = flag
            net.running_mean.copy_(rm0)
            net.running_var.copy_(rv0)
            net.zero_grad(set_to_none=True)
            out = net(x)
            (out.square().mean() + out.sum() * 1e-3).backward()
            torch.cuda.synchronize()
            return [out.detach().clone(), net.flat_weight.grad.clone(), net.flat_bias.grad.clone(), net.bn_weight_first.grad.clone(),
                    net.bn_weight_rest.grad.clone(), net.bn_bias_rest.grad.clone(), net.running_mean.clone(), net.running_var.clone()]

        ref = step(False)
        assert float(ref[1].abs().max()) > 0
        for rep in range(2):
            got = step(True)
            assert all(torch.equal(a, b) for a, b in zip(got, ref)), (cin, rep)


def test_early_weight_flip_is_bit_identical():
    """TBNModel.flip_weights_early / BNInception.flip_weights_early (tbn_backbone_flip_weights + TBN_BACKBONE_WEIGHTS_FLIPPED,
    include/tbn_hip.h): the data-gradient weight copy launched right after the modality streams are joined instead of at the
    start of each backward pass.  Same kernel on the same weights: every gradient of a config-4 train step is bit-identical
    with the switch off; the library timeline shows one copy per backbone and step either way, before the heads' kernels when
    early; a lone backbone: the copy is launched at most once per training forward and never for a no-grad forward."""
    import ctypes as C
    from attention_based_tbn_amd import _lib
    from attention_based_tbn_amd.config import load_config, get_modality
    from attention_based_tbn_amd.core.models import build_model
    from attention_based_tbn_amd.core.models.bn_inception import BNInception
    cfg = load_config(["model.attention.enable=False", "data.audio.audio_length=1.279", "data.sampling=async", "model.fusion_dropout=0"])
    modality = get_modality(cfg)
    torch.manual_seed(0)
    model, crit, _ = build_model(cfg, modality, DEV)
    core = model.module if hasattr(model, "module") else model
    assert core.flip_weights_early is True
    g = torch.Generator(device="cuda").manual_seed(1)
    B, n = 2, 3
    inp = {"RGB": torch.rand(B, n, 3, 96, 96, device=DEV, generator=g) - 0.45,
           "Flow": torch.rand(B, n, 10, 96, 96, device=DEV, generator=g) - 0.5,
           "Audio": torch.randn(B, n, 1, 96, 128, device=DEV, generator=g) * 3 - 6}
    tgt = {"class": {"verb": torch.randint(0, 125, (B,), device=DEV), "noun": torch.randint(0, 352, (B,), device=DEV)}}
    state = {k: v.clone() for k, v in model.state_dict().items()}
    lib = _lib.lib()

    def step(early, trace=None):
        model.load_state_dict(state)
        model.train()
        core.flip_weights_early = early
        model.zero_grad(set_to_none=True)
        if trace:
            lib.tbn_timeline_enable(1)
        out = model(inp)
        loss, _ = model.get_loss(crit, tgt, out, 0)
        loss["total"].backward()
        torch.cuda.synchronize()
        if trace:
            assert lib.tbn_timeline_dump(trace.encode()) == 0
            lib.tbn_timeline_enable(0)
        return {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

    step(True)                       # tunes the plans
    ref = step(False)
    import csv, tempfile, os
    for early in (True, False):
        path = os.path.join(tempfile.mkdtemp(), "tl.csv")
        got = step(early, path)
        assert got.keys() == ref.keys()
        for k in ref:
            assert torch.equal(got[k], ref[k]), (early, k)
        rows = sorted((int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path)))
        names = [nm for _, nm in rows]
        flips = [i for i, nm in enumerate(names) if "weight_flip_transpose_all" in nm]
        ce = [i for i, nm in enumerate(names) if "ce_heads_bwd" in nm]
        assert len(flips) == 3 and len(ce) == 1, (early, len(flips), len(ce))
        if not early:
            assert min(flips) > ce[0]        # the backward passes' own first launch
    # one backbone on its own
    torch.manual_seed(2)
    net = BNInception(1000, 3).to(DEV)
    net.set_bn_trainable(True, True)
    net.use_aux_stream = net.use_branch_streams = False
    x = torch.randn(3, 3, 96, 96, device=DEV)

    def lone(early):
        net.train()
        net.zero_grad(set_to_none=True)
        out = net(x)
        if early:
            assert net.flip_weights_early() is True and net.flip_weights_early() is False      # once per training forward
        (out.square().mean()).backward()
        torch.cuda.synchronize()
        return net.flat_weight.grad.clone(), net.flat_bias.grad.clone()

    a = lone(False)
    for _ in range(2):
        b = lone(True)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    with torch.no_grad():
        assert net(x) is not None and net.flip_weights_early() is False    # no backward to prepare


def test_capture_guard_refuses_aux_stream_and_leaves_the_plan_intact(monkeypatch):
    """`tbn_backbone_backward` refuses an aux stream while its launch stream is capturing (TBN_ERR_UNSUPPORTED: the fix
    for the SIGSEGV of nested capture forks in ROCm 7.x's hipStreamEndCapture, profiles/r03_graph_capture_multi_aux_rocgdb.log).
    The Python host passes no aux stream while capturing, so the guard is reached here by hiding the capture from the host
    check: the C side must raise its message, the capture must end cleanly, and the same backward must then run eagerly
    with the aux stream and reproduce the earlier gradients bit for bit (plan and event pool intact)."""
    from attention_based_tbn_amd._lib import TbnHipError
    net, x, step = _lone_backbone_for_capture()
    step()                                    # eager: plan creation + autotune (synchronises: not allowed in a capture)
    torch.cuda.synchronize()
    g0 = net.flat_weight.grad.clone()
    rm0 = net.running_mean.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    raised = None
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        graph.capture_begin()
        try:
            out = net(x)
            loss = out.square().mean()
            monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
            try:
                loss.backward()
            except TbnHipError as e:
                raised = str(e)
            finally:
                monkeypatch.undo()
        finally:
            graph.capture_end()               # must not crash: nothing was forked
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert raised is not None and "aux_stream inside a stream capture is not supported" in raised, raised
    assert "rc=-3" in raised, raised          # TBN_ERR_UNSUPPORTED
    del graph, out, loss
    net.running_mean.copy_(rm0)
    step()                                    # eager again, aux stream in use
    torch.cuda.synchronize()
    assert torch.equal(net.flat_weight.grad, g0)


def test_captured_lone_backbone_train_step_replays_like_eager():
    """a lone backbone (use_aux_stream = True, the model's policy for one modality) under torch.cuda.graph: the host
    passes no aux stream while capturing, so the step captures with serial weight-gradient launches (it
    used to raise from inside autograd) and a replay on NEW input data reproduces the eager step bit for bit."""
    net, x, step = _lone_backbone_for_capture()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()                            # warm-up on the side stream (autotune happens here)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    net.zero_grad(set_to_none=True)
    # the library's kernel timeline may be ON while a step is captured (round-5 advisor): launches inside the capture must not
    # carry timeline events (an event-carrying dispatch invalidates a capture) -- they are simply absent from the timeline
    from attention_based_tbn_amd._lib import lib
    lib().tbn_timeline_enable(1)
    with torch.cuda.graph(graph):
        static_out = step()
    lib().tbn_timeline_enable(0)
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".csv") as tf:
        assert lib().tbn_timeline_dump(tf.name.encode()) == 0
        assert len(open(tf.name).read().splitlines()) == 1          # the header only: nothing was recorded inside the capture
    static_grad = net.flat_weight.grad        # allocated inside the capture: rewritten by every replay
    x.copy_(torch.randn(x.shape, generator=torch.Generator().manual_seed(5)).to(DEV))
    graph.replay()
    torch.cuda.synchronize()
    got_out, got_grad = static_out.detach().clone(), static_grad.clone()
    del graph
    # a captured step runs the one-chain program (no aux / side stream inside a capture): the eager step it must match bit
    # for bit is the same program -- branch mode off, whose two-chain program replaces sibling-pair launches by single ones
    # (other tiles, last-bit differences); the weight gradients may still take the aux stream (same kernels)
    net.use_branch_streams = False
    eager_out = step()
    torch.cuda.synchronize()
    assert torch.equal(got_out, eager_out.detach()) and torch.equal(got_grad, net.flat_weight.grad)
    assert float(got_grad.abs().max()) > 0
    net.use_branch_streams = True             # ... and the default (two-chain) eager step agrees to rounding
    branch_out = step()
    torch.cuda.synchronize()
    assert rel_err(branch_out.detach().cpu(), got_out.cpu()) < 2e-4    # (observed 4e-5: other GEMM variants for 3x3 | double_3x3_1, 69 layers deep)
