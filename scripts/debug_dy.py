"""Compare every conv's output gradient dy (and forward z) inside the engine workspace with the oracle."""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.bninception import BNInception as OBN
from oracle.fill import fill_state_dict
from attention_based_tbn_amd.core.models.bn_inception import BNInception
from attention_based_tbn_amd._lib import call
from tests.util import rel_err
cin, H, W, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
DEV = "cuda"
ora = OBN(1000, cin).double(); sd = fill_state_dict(OBN(1000, cin).state_dict(), 42); ora.load_state_dict(sd)
net = BNInception(1000, cin).to(DEV); net.load_state_dict(sd)
x = torch.randn(N, cin, H, W, generator=torch.Generator().manual_seed(1))
ora.train(); net.train()
dys, zs = {}, {}
for name in net._order:
    getattr(ora, name).register_full_backward_hook(lambda m, gi, go, name=name: dys.__setitem__(name, go[0].detach()))
    getattr(ora, name + "_bn").register_forward_hook(lambda m, i, o, name=name: zs.__setitem__(name, torch.relu(o.detach())))
yo = ora(x.double()); dy = torch.randn(yo.shape, generator=torch.Generator().manual_seed(2)); yo.backward(dy.double())
y = net(x.to(DEV)); y.backward(dy.to(DEV))
plan = net._plan(N, H, W); ws = plan.pool[0][0].view(torch.float32)
def fetch(name, kind):
    off, rows, cols, ld = C.c_long(), C.c_int(), C.c_int(), C.c_int()
    call("tbn_backbone_tensor_info", plan.handle, name.encode(), kind, C.byref(off), C.byref(rows), C.byref(cols), C.byref(ld))
    if off.value < 0: return None
    t = torch.as_strided(ws, (rows.value, cols.value), (ld.value, 1), off.value)
    return t.cpu()
for name in net._order:
    L = net._layers[name]
    d = fetch(name, 1); z = fetch(name, 0)
    ref = dys[name]; n_, c_, h_, w_ = ref.shape
    refm = ref.permute(0, 2, 3, 1).reshape(-1, c_); zref = zs[name].permute(0, 2, 3, 1).reshape(-1, c_)
    print(f"{name:34s} z {rel_err(z, zref):.2e}  dy {rel_err(d, refm):.2e}  shape {tuple(ref.shape)}")

# ---- gradient wrt z (relu output) per conv, column-profile of the error
def relu_name(n):
    if n == "conv1_7x7_s2": return "conv1_relu_7x7"
    if n.startswith("conv2_"): return "conv2_relu_" + n[len("conv2_"):]
    pre = n[:len("inception_3a_")]; return pre + "relu_" + n[len(pre):]
ora.zero_grad(); dzs = {}
hooks = []
for name in net._order:
    def fh(m, i, o, name=name):
        o.register_hook(lambda g, name=name: dzs.__setitem__(name, g.detach()))
    hooks.append(getattr(ora, relu_name(name)).register_forward_hook(fh))
yo = ora(x.double()); yo.backward(dy.double())
for name in sys.argv[5:]:
    d = fetch(name, 2)
    ref = dzs[name]; c_ = ref.shape[1]
    refm = ref.permute(0, 2, 3, 1).reshape(-1, c_)
    err = (d.double() - refm).abs()
    print(name, "dz overall", rel_err(d, refm), "rows", d.shape[0])
    print("  per-col-block(32) max err:", [f"{float(err[:, i:i+32].max()):.1e}" for i in range(0, c_, 32)])
    print("  per-row max err:", [f"{float(v):.1e}" for v in err.max(1).values[:48]])
    print("  ref max", float(refm.abs().max()))

print("==== dy profiles")
for name in ["inception_5a_3x3_reduce", "inception_4d_double_3x3_1", "inception_5a_pool_proj"]:
    d = fetch(name, 1)
    ref = dys[name]; c_ = ref.shape[1]
    refm = ref.permute(0, 2, 3, 1).reshape(-1, c_)
    err = (d.double() - refm).abs()
    print(name, "dy overall", rel_err(d, refm), "rows", d.shape[0], "refmax", float(refm.abs().max()))
    print("  per-col-block(32) max err:", [f"{float(err[:, i:i+32].max()):.1e}" for i in range(0, c_, 32)])
    print("  per-row max err:", [f"{float(v):.1e}" for v in err.max(1).values[:64]])

print("==== values")
name = "inception_5a_pool_proj"
d = fetch(name, 2); ref = dzs[name]; c_ = ref.shape[1]
refm = ref.permute(0, 2, 3, 1).reshape(-1, c_)
torch.set_printoptions(precision=4, linewidth=200)
print("hip  dz ch0 n0:", d[:16, 0].view(2, 8))
print("ref  dz ch0 n0:", refm[:16, 0].view(2, 8))
z = fetch(name, 0); zr = zs[name].permute(0, 2, 3, 1).reshape(-1, c_)
print("hip  z ch0 n0:", z[:16, 0].view(2, 8))
print("ref  z ch0 n0:", zr[:16, 0].view(2, 8))
zz = fetch("inception_5a_pool_proj", 0)
print("bitwise row-tie mismatches (hip z):", int((zz[0:8] != zz[8:16]).sum()), "of", zz[0:8].numel())
print("max |z0-z1|:", float((zz[0:8]-zz[8:16]).abs().max()))
yy = fetch("inception_5a_pool_proj", 1)
