"""debug: why the accumulation test's replay norm is 0.2 % below the HIP clip norm"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.util import load_case
from tests.test_model_gpu import build_product, to_dev
from attention_based_tbn_amd.core.utils import FusedSGD, TrainStep, clip_grad_norm_
cfg, modality, meta, data, inp, target = load_case("train_cfg4_all_noattn")
model, crit = build_product(cfg, modality, meta)
model.train()
named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
params = [p for _, p in named]
print("params with requires_grad:", len(named), " all params:", len(list(model.parameters())))
cap, fired = {}, {}
def mk(n):
    def h(g):
        fired[n] = fired.get(n, 0) + 1
        cap[n] = g.detach().clone()
    return h
for n, p in named:
    p.register_hook(mk(n))
opt = FusedSGD(params, 0.01, momentum=0.9, weight_decay=0.0005)
step = TrainStep(model, opt, crit, accumulator_step=2, clip_grad=1e9)     # clip never bites: grads stay as produced
g = torch.Generator().manual_seed(2)
x = {m: v + 0.05 * torch.randn(v.shape, generator=g) for m, v in inp.items()}
loss, bs = step(0, to_dev(x), {"class": to_dev(target["class"])}, epoch=0)
torch.cuda.synchronize()
hip = float(step.last_total_norm)
withgrad = [(n, p) for n, p in model.named_parameters() if p.grad is not None]
tn_all = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for _, p in withgrad)))
tn_hook = float(torch.sqrt(sum((c.double() ** 2).sum() for c in cap.values())))
print("HIP clip norm %.6f | torch norm of p.grad (all with grad: %d tensors) %.6f | torch norm of hook captures (%d) %.6f" %
      (hip, len(withgrad), tn_all, len(cap), tn_hook))
names_req = {n for n, _ in named}
for n, p in withgrad:
    c = cap.get(n)
    if n not in names_req:
        print("GRAD WITHOUT requires_grad:", n, float(p.grad.norm()))
    elif c is None:
        print("no hook capture:", n, float(p.grad.norm()))
    else:
        d = float((p.grad - c).norm())
        if d > 1e-6 * float(p.grad.norm()) or fired.get(n, 0) != 1:
            print("%-40s |grad| %.5f |hook| %.5f |diff| %.5f fired %d" % (n, float(p.grad.norm()), float(c.norm()), d, fired.get(n, 0)))
# per-tensor HIP norm vs torch norm
for n, p in withgrad:
    t = float(clip_grad_norm_([p], 1e9))
    r = float(p.grad.norm())
    if abs(t - r) > 1e-5 * max(r, 1e-12):
        print("HIP norm differs on %-40s hip %.6f torch %.6f numel %d ptr%%16 %d contiguous %s" % (n, t, r, p.grad.numel(), p.grad.data_ptr() % 16, p.grad.is_contiguous()))
