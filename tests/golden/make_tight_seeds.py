#!/usr/bin/env python
"""Seed search for WELL-CONDITIONED whole-backbone gradient cases -- the evidence behind the forced-decision design
of tests/test_model_gpu.py::test_backbone_all_layer_grads_forced_decisions (result of 300 seeds per modality at one
96x96 frame: the best instance keeps its closest decision only 0.7-1.3 fp32-error units from the boundary, i.e. NO
instance of this network is safe from a decision flip in fp32; hence the oracle takes the product's decisions instead).

The fp32 training-mode backward of BN-Inception is ill conditioned in general: a ReLU input within rounding distance of
zero, or two near-equal maxima in a max-pool window, flips a routing decision and moves every upstream gradient by
1e-3...1e-2 -- in ANY fp32 implementation (torch's CPU path included).  Such flips say nothing about the kernels.  This
script looks, per modality, for the input seed whose fp64 forward keeps the LARGEST distance from every such decision:
  relu margin = min |BatchNorm output| over all 69 layers (the ReLU input), relative to that tensor's rms,
  pool margin = min gap between the two largest entries of every max-pool window whose maximum is positive.
and, because the fp32 error of an activation grows with depth (1e-6 at the stem, 1e-4 in inception_5x), ALSO measures
the margins in units of the observed fp32 error: the oracle's fp32 forward runs beside the fp64 one and
  safety = min over every decision of (distance from the boundary) / |fp32 value - fp64 value|.
A seed would be flip-proof at safety >> 1 (a perturbation several times the torch-fp32 error still flips nothing); the
search reports the best safety it finds.
Weights are the fixed seeded fill (seed 42) of the other backbone tests.  Writes tests/golden/tight_grad_seeds.json.
Usage: python tests/golden/make_tight_seeds.py [n_seeds]
"""
import json
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.bninception import BNInception  # noqa: E402
from oracle.fill import fill_state_dict  # noqa: E402

N, H, W = 1, 96, 96   # >= 3 x 3 final maps: on 2-row maps the 3x3 average pool yields identical rows = exact max-pool ties


def record(net, x):
    """BN outputs (= ReLU inputs) and max-pool inputs of one forward, in module order"""
    bn, pl = [], []
    hs = []
    for mod in net.modules():
        if isinstance(mod, nn.BatchNorm2d):
            hs.append(mod.register_forward_hook(lambda m, i, o: bn.append(o.detach().clone())))
        elif isinstance(mod, nn.MaxPool2d):
            hs.append(mod.register_forward_pre_hook(lambda m, i: pl.append((m, i[0].detach().clone()))))
    with torch.no_grad():
        net(x)
    for h in hs:
        h.remove()
    return bn, pl


def margins(net64, net32, x):
    m = {"relu": float("inf"), "pool": float("inf"), "safety": float("inf")}
    bn64, pl64 = record(net64, x.double())
    bn32, pl32 = record(net32, x.float())
    for a, b in zip(bn64, bn32):
        rms = float(a.pow(2).mean().sqrt())
        m["relu"] = min(m["relu"], float(a.abs().min()) / max(rms, 1e-30))
        err = (b.double() - a).abs().clamp_min(1e-9 * rms)
        m["safety"] = min(m["safety"], float((a.abs() / err).min()))
    for (mod, a), (_, b) in zip(pl64, pl32):
        kw = dict(kernel_size=mod.kernel_size, stride=mod.stride, padding=mod.padding, ceil_mode=mod.ceil_mode)
        top, idx = F.max_pool2d(a, return_indices=True, **kw)
        a2 = a.clone().flatten(2)
        a2.scatter_(2, idx.flatten(2), float("-inf"))
        second = F.max_pool2d(a2.view_as(a), **kw)
        live = (top > 0) & torch.isfinite(second)
        if live.any():
            rms = float(a.pow(2).mean().sqrt())
            gap = (top - second)[live]
            m["pool"] = min(m["pool"], float(gap.min()) / max(rms, 1e-30))
            err = 2.0 * max(float((b.double() - a).abs().max()), 1e-9 * rms)   # both window entries may move
            m["safety"] = min(m["safety"], float(gap.min()) / err)
    return m


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    torch.set_num_threads(8)
    out = {"frames": N, "height": H, "width": W, "weight_fill_seed": 42, "searched_seeds": n_seeds}
    for cin in (3, 10, 1):
        net = BNInception(1000, cin)
        net.load_state_dict(fill_state_dict(net.state_dict(), 42))
        import copy
        net32 = copy.deepcopy(net).float().train()
        net = net.double().train()
        best = None
        for seed in range(n_seeds):
            x = torch.randn(N, cin, H, W, generator=torch.Generator().manual_seed(10_000 + seed))
            mg = margins(net, net32, x)
            if best is None or mg["safety"] > best[0]:
                best = (mg["safety"], 10_000 + seed, mg)
        out[str(cin)] = {"input_seed": best[1], "relu_margin": best[2]["relu"], "pool_margin": best[2]["pool"],
                         "safety_vs_fp32_error": best[0]}
        print(cin, out[str(cin)], flush=True)
    with open(os.path.join(HERE, "tight_grad_seeds.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
