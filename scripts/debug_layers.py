"""Per-layer gradient error of the HIP backbone vs an fp64 oracle, next to the fp32 CPU oracle's own
error vs fp64 (how ill-conditioned the training-mode backward is at this batch size)."""
import sys, os, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.bninception import BNInception as OBN
from oracle.fill import fill_state_dict
from attention_based_tbn_amd.core.models.bn_inception import BNInception
from tests.util import rel_err
cin, H, W, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
DEV = "cuda"
ora = OBN(1000, cin); sd = fill_state_dict(ora.state_dict(), 42); ora.load_state_dict(sd)
o64 = copy.deepcopy(ora).double()
net = BNInception(1000, cin).to(DEV); net.load_state_dict(sd)
x = torch.randn(N, cin, H, W, generator=torch.Generator().manual_seed(1))
ora.train(); net.train(); o64.train()
yo = ora(x); dy = torch.randn(yo.shape, generator=torch.Generator().manual_seed(2)); yo.backward(dy)
y64 = o64(x.double()); y64.backward(dy.double())
y = net(x.to(DEV)); y.backward(dy.to(DEV))
print("fwd  hip-vs-64 %.2e   cpu32-vs-64 %.2e" % (rel_err(y.detach().cpu(), y64.detach()), rel_err(yo.detach(), y64.detach())))
op = dict(ora.named_parameters()); p64 = dict(o64.named_parameters())
for lname, L in net._layers.items():
    nw = L["cout"] * L["k"] * L["k"] * L["cin"]
    gw = net.flat_weight.grad[L["w_off"]:L["w_off"] + nw].view(L["cout"], L["k"], L["k"], L["cin"]).permute(0, 3, 1, 2).cpu()
    print(f"{lname:34s} dW hip-vs-64 {rel_err(gw, p64[lname + '.weight'].grad):.2e}   cpu32-vs-64 {rel_err(op[lname + '.weight'].grad, p64[lname + '.weight'].grad):.2e}")
