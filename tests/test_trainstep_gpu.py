"""GPU parity of the train-step shell and metrics (through the C-ABI) against the oracle and the fixtures made by
torch's own optimiser / the reference's Metric (tests/golden/make_golden_trainstep.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import trainstep as ot

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"


def _load():
    z = np.load(os.path.join(GOLD, "trainstep.npz"))
    return z, [torch.nn.Parameter(torch.from_numpy(z["p0_%d" % i].astype(np.float32)).to(DEV)) for i in range(3)]


@pytest.mark.parametrize("fused_clip", [False, True, "consumed"])
def test_fused_sgd_and_clip_match_torch_golden(fused_clip):
    from attention_based_tbn_amd.core.utils import FusedSGD, clip_grad_norm_
    z, params = _load()
    frozen = torch.nn.Parameter(torch.ones(7, device=DEV))     # no gradient: must stay untouched
    opt = FusedSGD(params + [frozen], 0.01, momentum=0.9, weight_decay=0.0005)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[2], gamma=0.1)
    for s in range(3):
        opt.zero_grad()
        for i, p in enumerate(params):
            p.grad = torch.from_numpy(z["g%d_%d" % (s, i)].astype(np.float32)).to(DEV)
        if fused_clip:
            # default: gradients rescaled in memory like the reference's clip_grad_norm_; "consumed": coefficient folded
            # into the update, gradients left as they were (callers that zero them before the next backward)
            before = [p.grad.clone() for p in params]
            opt.step(clip_grad=20, grads_consumed=(fused_clip == "consumed"))
            total = opt.last_total_norm
            if s == 1:
                for i, p in enumerate(params):
                    if fused_clip == "consumed":
                        assert torch.equal(p.grad, before[i])
                    else:
                        np.testing.assert_allclose(p.grad.cpu().numpy(), z["gclip1_%d" % i], rtol=2e-6, atol=0)
        else:
            total = clip_grad_norm_(params + [frozen], 20)
            if s == 1:
                for i, p in enumerate(params):
                    np.testing.assert_allclose(p.grad.cpu().numpy(), z["gclip1_%d" % i], rtol=2e-6, atol=0)
            opt.step()
        assert abs(total.item() - z["norm%d" % s]) <= 2e-6 * z["norm%d" % s]
        sched.step()
        assert abs(opt.param_groups[0]["lr"] - z["lr%d" % s]) < 1e-9
        for i, p in enumerate(params):
            np.testing.assert_allclose(p.detach().cpu().numpy(), z["p%d_%d" % (s + 1, i)], rtol=2e-6, atol=2e-7)
    for i, p in enumerate(params):
        np.testing.assert_allclose(opt.state[p]["momentum_buffer"].cpu().numpy(), z["m3_%d" % i], rtol=4e-6, atol=2e-7)
    assert torch.equal(frozen.detach().cpu(), torch.ones(7))
    # torch's state_dict layout, so torch.optim.SGD can resume from it
    ref = torch.optim.SGD(params + [frozen], 0.01, momentum=0.9, weight_decay=0.0005)
    ref.load_state_dict(opt.state_dict())
    assert torch.equal(ref.state[params[2]]["momentum_buffer"], opt.state[params[2]]["momentum_buffer"])


def test_fused_sgd_many_tensors_vs_oracle():
    """more tensors than one launch table holds (48), odd sizes, no momentum / no weight decay variants"""
    from attention_based_tbn_amd.core.utils import FusedSGD
    g = torch.Generator().manual_seed(3)
    sizes = [1 + (i * 977) % 5003 for i in range(61)]
    for mom, wd in ((0.0, 0.0), (0.9, 0.0), (0.5, 0.01)):
        ps = [torch.randn(n, generator=g) for n in sizes]
        params = [torch.nn.Parameter(p.clone().to(DEV)) for p in ps]
        opt = FusedSGD(params, 0.05, momentum=mom, weight_decay=wd)
        want = [p.numpy().copy() for p in ps]
        bufs = [None] * len(ps)
        for step in range(2):
            gs = [torch.randn(n, generator=g) for n in sizes]
            for p, gr in zip(params, gs):
                p.grad = gr.to(DEV)
            opt.step()
            for i in range(len(ps)):
                want[i], bufs[i] = ot.sgd_step(want[i], gs[i].numpy(), bufs[i], 0.05, mom, wd)
        for i, p in enumerate(params):
            np.testing.assert_allclose(p.detach().cpu().numpy(), want[i], rtol=2e-6, atol=2e-7)


def test_optimizer_rejects_cpu_tensors():
    from attention_based_tbn_amd._lib import TbnHipError
    from attention_based_tbn_amd.core.utils import FusedSGD, clip_grad_norm_
    p = torch.nn.Parameter(torch.ones(8))
    p.grad = torch.ones(8)
    with pytest.raises(TbnHipError):
        clip_grad_norm_([p], 1.0)
    with pytest.raises(TbnHipError):
        FusedSGD([p], 0.1).step()


def test_metric_matches_reference_golden():
    from attention_based_tbn_amd.config import load_config
    from attention_based_tbn_amd.core.utils import Metric
    doc = json.load(open(os.path.join(GOLD, "metric.json")))
    cfg = load_config(doc["overrides"])
    m = Metric(cfg, 2, device=torch.device(DEV))
    for b in doc["batches"]:
        out = {"verb": torch.tensor(b["verb"], device=DEV), "noun": torch.tensor(b["noun"], device=DEV),
               "weights": torch.zeros(b["B"] * 3, 1, 8, device=DEV)}
        tgt = {"class": {"verb": torch.tensor(b["t_verb"], device=DEV), "noun": torch.tensor(b["t_noun"], device=DEV)}}
        m.set_metrics(out, tgt, b["B"], {k: torch.tensor(v) for k, v in b["loss"].items()})
    loss, acc, cm = m.get_metrics()
    exp = doc["expected"]
    assert acc == exp["accuracy"]
    for k, v in exp["loss"].items():
        assert abs(loss[k] - v) <= 1.1e-5
    for k in ("verb", "noun"):
        assert torch.equal(cm[k].cpu(), torch.tensor(exp["conf_mat"][k]))


@pytest.mark.parametrize("B,C,k", [(1, 2, 1), (33, 125, 5), (512, 352, 5), (9, 1000, 10)])
def test_topk_correct_vs_oracle_with_ties(B, C, k):
    from attention_based_tbn_amd.core.utils.metric import get_correct_score
    g = torch.Generator().manual_seed(B * 7 + C)
    scores = torch.randn(B, C, generator=g)
    scores[:, ::3] = scores[:, :1].clone()    # many exact ties: lower class index ranks first
    target = torch.randint(0, C, (B,), generator=g)
    corr, cm = get_correct_score(scores.to(DEV), target.to(DEV), [1, k])
    wc, wm = ot.topk_correct(scores.numpy(), target.numpy(), k)
    assert np.array_equal(corr.cpu().numpy(), wc)
    assert np.array_equal(cm.cpu().numpy(), wm)
    assert cm.sum().item() == B               # every sample lands in exactly one cell (checksum property)


def test_checkpoint_roundtrip_reference_layout(tmp_path):
    """save_checkpoint writes the reference's file layout (per-layer keys, per-layer optimizer slots); loading it
    back restores parameters and the flat momentum buffers bit for bit"""
    from attention_based_tbn_amd.config import get_modality, load_config
    from attention_based_tbn_amd.core.models import build_model
    from attention_based_tbn_amd.core.utils import FusedSGD, load_checkpoint, save_checkpoint
    from attention_based_tbn_amd.core.utils.misc import reference_parameter_names
    cfg = load_config(["data.flow.enable=False", "data.audio.enable=False", "model.attention.enable=False"])
    mod = get_modality(cfg)
    torch.manual_seed(0)
    model, crit, _ = build_model(cfg, mod, torch.device(DEV))
    model.train()
    opt = FusedSGD(model.parameters(), 0.01, momentum=0.9, weight_decay=0.0005)
    x = {"RGB": torch.rand(2, 3, 3, 64, 64, device=DEV) - 0.45}
    tgt = {"class": {"verb": torch.tensor([1, 2], device=DEV), "noun": torch.tensor([3, 4], device=DEV)}}
    loss, _ = model.get_loss(crit, tgt, model(x), 0)
    loss["total"].backward()
    opt.step(clip_grad=20)
    fn = str(tmp_path / "ck.pth")
    save_checkpoint(model, opt, 3, [1.0], [2.0], [3.0], None, 1, filename=fn)
    data = torch.load(fn, map_location="cpu")
    names = reference_parameter_names(model)
    assert list(data.keys()) == ["epoch", "train_loss", "validation_loss", "validation_accuracy", "optimizer", "model"]
    assert data["optimizer"]["param_groups"][0]["params"] == list(range(len(names)))
    assert data["model"]["Base_RGB.conv1_7x7_s2.weight"].shape == (64, 3, 7, 7)
    assert data["optimizer"]["state"][0]["momentum_buffer"].shape == (64, 3, 7, 7)
    torch.manual_seed(1)
    model2, _, _ = build_model(cfg, mod, torch.device(DEV))
    opt2 = FusedSGD(model2.parameters(), 0.5, momentum=0.1)
    load_checkpoint(fn, model2, opt2)
    for (n1, p1), (n2, p2) in zip(model.named_parameters(), model2.named_parameters()):
        assert torch.equal(p1, p2), n1
        b1, b2 = opt.state.get(p1, {}).get("momentum_buffer"), opt2.state.get(p2, {}).get("momentum_buffer")
        assert (b1 is None and b2 is None) or torch.equal(b1, b2), n1
    assert opt2.param_groups[0]["lr"] == 0.01 and opt2.param_groups[0]["momentum"] == 0.9


def _reference_loop(params, micro_grads, k, clip, lr, mom, wd):
    """core/tools/train.py:66-94 driven by torch's own clip_grad_norm_ + optim.SGD on CPU copies: `micro_grads[it]` is what
    iteration it's backward adds to .grad.  -> per iteration (parameters, momentum buffers, total norm).  Run in float64:
    torch's fp32 CPU norm of a 10-M-element gradient sums sequentially per thread and comes out 0.2 % LOW (71.574 for a
    true 71.721 -- what the HIP kernel's fp64-finalised partials and torch's own fp64 norm both give), so an fp32 CPU
    replay is the less accurate side; fp64 is the exact statement of the reference's arithmetic."""
    ps = [torch.nn.Parameter(p.double().clone()) for p in params]
    opt = torch.optim.SGD(ps, lr, momentum=mom, weight_decay=wd)
    trace = []
    for it, gs in enumerate(micro_grads):
        if (it + 1) % k == 0:
            opt.zero_grad()
        for p, g in zip(ps, gs):
            if g is not None:
                p.grad = g.double().clone() if p.grad is None else p.grad + g.double()
        tn = torch.nn.utils.clip_grad_norm_(ps, clip) if clip else None
        if (it + 1) % k == k - 1:
            opt.step()
        bufs = [opt.state[p].get("momentum_buffer") for p in ps]
        trace.append(([p.detach().clone() for p in ps], [None if b is None else b.clone() for b in bufs],
                      None if tn is None else float(tn)))
    return trace


@pytest.mark.parametrize("k,clip", [(2, 2.0), (3, None), (1, 2.0)])
def test_accumulation_schedule_on_the_real_model(k, clip):
    """SURVEY 8 f2 "accumulation semantics" (reference core/tools/train.py:66-94: zero_grad when (it+1) % k == 0 BEFORE the
    forward, loss / k, clip_grad_norm_ on the ACCUMULATED gradients every iteration, step when (it+1) % k == k-1).
    `TrainStep` + `FusedSGD` + the multi-tensor clip run four or five iterations of the small three-modality TBN model on the GPU;
    the micro-batch gradient of every iteration is captured by tensor hooks and the reference loop is REPLAYED on the CPU
    with torch's own `clip_grad_norm_` + `optim.SGD` in float64 on exactly those gradients: parameters, momentum buffers and the
    clip norms agree after every iteration within 2e-6 (the schedule, the re-clipping of accumulated gradients and the
    fused / in-place clip modes -- independent of the backbone's fp32 gradient conditioning)."""
    from tests.util import load_case
    from tests.test_model_gpu import build_product, to_dev
    from attention_based_tbn_amd.core.utils import FusedSGD, TrainStep
    cfg, modality, meta, data, inp, target = load_case("train_cfg4_all_noattn")
    model, crit = build_product(cfg, modality, meta)
    model.train()
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    params = [p for _, p in named]
    p0 = [p.detach().cpu().clone() for p in params]
    lr, mom, wd = 0.01, 0.9, 0.0005
    opt = FusedSGD(params, lr, momentum=mom, weight_decay=wd)
    step = TrainStep(model, opt, crit, accumulator_step=k, clip_grad=clip)
    micro = []
    # (device-side clone, moved to the host after the step: stream-ordered behind the kernels that produce the gradient)
    hooks = [p.register_hook(lambda g, i=i: micro[-1].__setitem__(i, g.detach().clone())) for i, p in enumerate(params)]
    g = torch.Generator().manual_seed(k)
    got = []
    nit = 5 if k == 3 else 4       # k = 3: steps after iterations 1 and 4, zero_grad in front of iteration 2
    for it in range(nit):
        micro.append([None] * len(params))
        x = {m: v + 0.05 * torch.randn(v.shape, generator=g) for m, v in inp.items()}
        loss, bs = step(it, to_dev(x), {"class": to_dev(target["class"])}, epoch=0)
        assert bs == 2
        torch.cuda.synchronize()
        micro[-1] = [None if g_ is None else g_.cpu() for g_ in micro[-1]]
        got.append(([p.detach().cpu().clone() for p in params],
                    [opt.state[p].get("momentum_buffer") for p in params],
                    None if step.last_total_norm is None else float(step.last_total_norm)))
        got[-1] = (got[-1][0], [None if b is None else b.cpu().clone() for b in got[-1][1]], got[-1][2])
    for h in hooks:
        h.remove()
    want = _reference_loop(p0, micro, k, clip, lr, mom, wd)
    nclipped = 0
    for it in range(nit):
        gp, gm, gn = got[it]
        wp, wm, wn = want[it]
        if clip:
            assert abs(gn - wn) <= 4e-6 * wn, (it, gn, wn)
            nclipped += int(wn > clip)
        for (n, _), a, b, ma, mb in zip(named, gp, wp, gm, wm):
            np.testing.assert_allclose(a.double().numpy(), b.numpy(), rtol=2e-6, atol=2e-7, err_msg=f"it {it} {n}")
            assert (ma is None) == (mb is None), (it, n)
            if ma is not None:
                np.testing.assert_allclose(ma.double().numpy(), mb.numpy(), rtol=4e-6, atol=2e-7, err_msg=f"it {it} momentum {n}")
    assert not clip or nclipped >= 2, nclipped          # the clip bit, also on re-clipped accumulated gradients
    if k > 1:   # the first optimiser step of the reference schedule comes after iteration k - 2, not k - 1
        first = min(it for it in range(nit) if (it + 1) % k == k - 1)
        assert all(torch.equal(a, b) for a, b in zip(got[first - 1][0], p0)) if first > 0 else True
        assert not all(torch.equal(a, b) for a, b in zip(got[first][0], p0))


def test_accumulation_k2_against_the_cpu_oracle_in_the_loop():
    """the same schedule with the CPU ORACLE model in the loop (its own forward / backward, torch's clip_grad_norm_ and
    optim.SGD, the reference loop body written out): four iterations at accumulator_step = 2 on the small TBN model.
    Losses within 1e-3 every iteration, the clip norms within 2e-2; the accumulated parameter UPDATE (four iterations,
    momentum 0.9) agrees with the oracle's as the gradients of this B = 2 case do: all tensors together relative L2 < 0.1 /
    cosine > 0.995, each tensor on its own < 0.25 / > 0.96 (two fp32 runs of this graph differ by last-bit ReLU / max-pool
    decisions, and on a 6-frame batch one flipped decision moves a gradient by percents: torch's own fp32 CPU gradients sit
    up to 7e-2 from an fp64 run of the small cases; observed 4.8e-2 / 0.9989 over all tensors, worst
    tensor 8.7e-2).
    A wrong schedule (a missed zero_grad, a step on the wrong iteration, an un-persisted clip) changes the update by
    tens of percent."""
    from tests.util import build_oracle, load_case
    from tests.test_model_gpu import build_product, to_dev
    from attention_based_tbn_amd.core.utils import FusedSGD, TrainStep
    cfg, modality, meta, data, inp, target = load_case("train_cfg4_all_noattn")
    model, crit = build_product(cfg, modality, meta)
    oracle, ocrit = build_oracle(cfg, modality, meta)
    model.train()
    oracle.train()
    k, clip, lr, mom, wd = 2, 2.0, 0.01, 0.9, 0.0005
    opt = FusedSGD([p for p in model.parameters() if p.requires_grad], lr, momentum=mom, weight_decay=wd)
    oopt = torch.optim.SGD([p for p in oracle.parameters() if p.requires_grad], lr, momentum=mom, weight_decay=wd)
    step = TrainStep(model, opt, crit, accumulator_step=k, clip_grad=clip)
    p0 = {n: p.detach().clone() for n, p in oracle.named_parameters()}
    g = torch.Generator().manual_seed(3)
    for it in range(4):
        x = {m: v + 0.05 * torch.randn(v.shape, generator=g) for m, v in inp.items()}
        loss, _ = step(it, to_dev(x), {"class": to_dev(target["class"])}, epoch=0)
        if (it + 1) % k == 0:
            oopt.zero_grad()
        oloss, _ = oracle.get_loss(ocrit, target, oracle({m: v.clone() for m, v in x.items()}), 0)
        oloss["total"] = oloss["total"] / k
        oloss["total"].backward()
        otn = float(torch.nn.utils.clip_grad_norm_(oracle.parameters(), clip))
        if (it + 1) % k == k - 1:
            oopt.step()
        a, b = float(loss["total"].detach()), float(oloss["total"].detach())
        assert abs(a - b) <= 1e-3 * max(1.0, abs(b)), (it, a, b)
        assert abs(float(step.last_total_norm) - otn) <= 2e-2 * otn, (it, float(step.last_total_norm), otn)
    sd = model.state_dict()
    worst = (0.0, "")
    all_u, all_w = [], []
    for n, q in oracle.named_parameters():
        if not q.requires_grad:
            continue
        du = (sd[n].detach().cpu().double() - p0[n].double()).flatten()
        dw = (q.detach().double() - p0[n].double()).flatten()
        if float(dw.norm()) < 1e-12:
            assert float(du.norm()) < 1e-9, n
            continue
        e = float((du - dw).norm() / dw.norm())
        c = float(torch.dot(du, dw) / (du.norm() * dw.norm()))
        if e > worst[0]:
            worst = (e, n)
        assert e < 0.25 and c > 0.96, (n, e, c)
        all_u.append(du)
        all_w.append(dw)
    du, dw = torch.cat(all_u), torch.cat(all_w)
    e = float((du - dw).norm() / dw.norm())
    c = float(torch.dot(du, dw) / (du.norm() * dw.norm()))
    print("accumulated update vs oracle: all tensors relative L2 %.2e cosine %.6f; worst tensor %.2e at %s" % (e, c, worst[0], worst[1]))
    assert e < 0.1 and c > 0.995, (e, c)
