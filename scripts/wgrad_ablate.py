"""Ablation timing of the weight-gradient kernel on one conv shape (needs a -DTBN_ABLATE=1 build:
`TBN_ABLATE=1 python -m attention_based_tbn_amd.build --force`; never the shipped library).
flags (env TBN_WGRAD_ABLATE, read per launch): 1 no global loads in the loop, 2 no LDS stores after the first
step, 4 no fragment reads / MFMAs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attention_based_tbn_amd._lib import call, ptr, lib
n, h, w, cin, cout, k, s, p = [int(v) for v in sys.argv[1:9]]
DEV = "cuda"
oh = (h + 2 * p - k) // s + 1; ow = (w + 2 * p - k) // s + 1
x = torch.randn(n, h, w, cin, device=DEV); dy = torch.randn(n, oh, ow, cout, device=DEV)
dw = torch.empty(cout, k, k, cin, device=DEV)
ws = torch.empty(max(1, lib().tbn_conv2d_wgrad_workspace_floats(n, h, w, cin, cout, k, s, p)), device=DEV)
st = torch.cuda.current_stream().cuda_stream
flops = 2.0 * n * oh * ow * cout * k * k * cin
def run(flags, reps=10):
    os.environ["TBN_WGRAD_ABLATE"] = str(flags)
    args = (ptr(dy), cout, ptr(x), cin, ptr(dw), n, h, w, cin, cout, k, s, p, ptr(ws), st)
    for _ in range(3): call("tbn_conv2d_wgrad", *args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call("tbn_conv2d_wgrad", *args)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print(f"shape {sys.argv[1:9]}  ideal {flops/157.3e12*1e6:.1f} us (incl. split-K reduce launch in every column)")
if os.environ.get("TILES"):          # weight-gradient time per forced tile (0,0 = the heuristic)
    ws = torch.empty(max(1, lib().tbn_conv2d_wgrad_workspace_floats(n, h, w, cin, cout, k, s, p)) * 4, device=DEV)
    for mt, nt in [(0, 0), (1, 1), (1, 2), (2, 1), (2, 2), (1, 3), (3, 1), (2, 3), (3, 2), (3, 3)]:
        os.environ["TBN_WGRAD_MT"], os.environ["TBN_WGRAD_NT"] = str(mt), str(nt)
        t = run(0)
        print(f"tile <{mt},{nt}>  {t:8.1f} us  {flops / t / 1e6:6.1f} TF/s")
    sys.exit(0)
if os.environ.get("SWEEP"):
    ws = torch.empty(max(1, lib().tbn_conv2d_wgrad_workspace_floats(n, h, w, cin, cout, k, s, p)) * 16, device=DEV)
    print("target\\minrows " + " ".join(f"{m:8d}" for m in (256, 640, 1280, 2560)))
    for target in (512, 768, 1024, 1536, 2048, 3072):
        os.environ["TBN_WGRAD_TARGET"] = str(target)
        row = []
        for mr in (256, 640, 1280, 2560):
            os.environ["TBN_WGRAD_MINROWS"] = str(mr)
            row.append(run(0))
        print(f"{target:14d} " + " ".join(f"{t:8.1f}" for t in row))
    sys.exit(0)
for name, fl in [("full", 0), ("no-loads", 1), ("no-loads,no-stores", 3), ("no-mfma", 4), ("no-mfma,no-stores", 6), ("nothing (prologue+epilogue)", 7)]:
    t = run(fl)
    print(f"{name:30s} {t:8.1f} us  {flops / t / 1e6:6.1f} TF/s-equivalent")
