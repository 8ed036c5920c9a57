"""debug: tensor-hook gradients vs p.grad on the small config-4 model (why the accumulation test's replay norm is 0.2 % low)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.util import load_case
from tests.test_model_gpu import build_product, to_dev
cfg, modality, meta, data, inp, target = load_case("train_cfg4_all_noattn")
model, crit = build_product(cfg, modality, meta)
model.train()
named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
cap, fired = {}, {}
def mk(n):
    def h(g):
        fired[n] = fired.get(n, 0) + 1
        cap[n] = g.detach().clone()
    return h
for n, p in named:
    p.register_hook(mk(n))
for sync_first in (False, True):
    cap.clear(); fired.clear()
    model.zero_grad(set_to_none=True)
    out = model(to_dev(inp))
    loss, _ = model.get_loss(crit, {"class": to_dev(target["class"])}, out, 0)
    loss["total"].backward()
    torch.cuda.synchronize()
    print("---- pass", sync_first)
    for n, p in named:
        g = p.grad
        c = cap.get(n)
        if g is None or c is None:
            print("%-40s grad %s hook %s fired %d" % (n, g is not None, c is not None, fired.get(n, 0)))
            continue
        d = float((g - c).norm()); gn = float(g.norm())
        if d > 1e-6 * max(gn, 1e-20) or fired.get(n, 0) != 1:
            print("%-40s |grad| %.5f |hook| %.5f |diff| %.5f fired %d" % (n, gn, float(c.norm()), d, fired.get(n, 0)))
    print("total |grad| %.5f  |hook| %.5f" % (float(torch.sqrt(sum((p.grad.double() ** 2).sum() for _, p in named if p.grad is not None))),
                                                float(torch.sqrt(sum((c.double() ** 2).sum() for c in cap.values())))))
