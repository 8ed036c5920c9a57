"""per-layer gradient error of one backbone vs the fp64 oracle (the numbers test_backbone_all_layer_grads_vs_oracle checks)"""
import sys, os, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.bninception import BNInception as OBN
from oracle.fill import fill_state_dict
from attention_based_tbn_amd.core.models.bn_inception import BNInception
cin, H, W, N = [int(v) for v in sys.argv[1:5]]
def l2(a, b): return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
ora = OBN(1000, cin); sd = fill_state_dict(ora.state_dict(), 42); ora.load_state_dict(sd)
o64 = copy.deepcopy(ora).double(); net = BNInception(1000, cin).cuda(); net.load_state_dict(sd)
x = torch.randn(N, cin, H, W, generator=torch.Generator().manual_seed(1))
ora.train(), net.train(), o64.train()
yo = ora(x); dy = torch.randn(yo.shape, generator=torch.Generator().manual_seed(2)); yo.backward(dy)
y64 = o64(x.double()); y64.backward(dy.double())
y = net(x.cuda()); y.backward(dy.cuda())
print("fwd", l2(y.cpu(), y64), l2(yo, y64))
op, p64 = dict(ora.named_parameters()), dict(o64.named_parameters())
rows = []
for lname, L in net._layers.items():
    nw = L["cout"] * L["k"] * L["k"] * L["cin"]
    gw = net.flat_weight.grad[L["w_off"]:L["w_off"] + nw].view(L["cout"], L["k"], L["k"], L["cin"]).permute(0, 3, 1, 2).cpu()
    rows.append((l2(gw, p64[lname + ".weight"].grad), l2(op[lname + ".weight"].grad, p64[lname + ".weight"].grad), lname))
for r in sorted(rows, reverse=True)[:12]: print("%.4f (cpu fp32 %.4f) %s" % r)
print("pool_proj layers:")
for r in rows:
    if "pool_proj" in r[2]: print("   %.5f (cpu %.5f) %s" % r)
