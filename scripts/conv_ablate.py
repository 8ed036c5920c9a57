"""Ablation timing of the implicit-GEMM kernel on one conv shape, per tile
(flags: 4 no loop loads, 8 no MFMA, 16 no stores, 32 no LDS fragment reads, 64 no barrier)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attention_based_tbn_amd._lib import call, ptr, lib
n, h, w, cin, cout, k, s, p = [int(v) for v in sys.argv[1:9]]
DEV = "cuda"
x = torch.randn(n, h, w, cin, device=DEV); wt = torch.randn(cout, k, k, cin, device=DEV) * 0.05; b = torch.zeros(cout, device=DEV)
oh = (h + 2 * p - k) // s + 1; ow = (w + 2 * p - k) // s + 1
y = torch.empty(n, oh, ow, cout, device=DEV)
st = torch.cuda.current_stream().cuda_stream
flops = 2.0 * n * oh * ow * cout * k * k * cin
def run(flags, mt, nt, reps=10):
    args = (ptr(x), cin, ptr(wt), ptr(b), ptr(y), cout, n, h, w, cin, cout, k, s, p, 0, flags, 0, mt, nt, st)
    for _ in range(3): call("tbn_conv2d_fwd_tile", *args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call("tbn_conv2d_fwd_tile", *args)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
cfgs = [("setup-only", 128), ("setup+prologue", 256), ("no-epilogue", 512), ("full", 0), ("no-loads", 4), ("no-mfma", 8), ("no-loads,no-mfma", 12), ("no-ld/mfma/lds/bar + no-epi", 108 + 512)]
print(f"shape {sys.argv[1:9]}  ideal {flops/157.3e12*1e6:.1f} us")
print("tile   " + "  ".join(f"{c[0]:>24s}" for c in cfgs))
for mt in (1, 2):
    for nt in (1, 2, 3, 4):
        if 32 * (nt - 1) >= cout: continue
        print(f"<{mt},{nt}>  " + "  ".join(f"{run(fl, mt, nt)*1e3:21.1f} us" for _, fl in cfgs))
