"""Launches the forward implicit-GEMM kernel on one conv shape with a few fixed tiles (production library), for
`rocprofv3 --pmc ... -- python3 scripts/conv_pmc.py n h w cin cout k s p` counter passes (per-kernel SQ counters)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attention_based_tbn_amd._lib import call, ptr
n, h, w, cin, cout, k, s, p = [int(v) for v in sys.argv[1:9]]
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 3
x = torch.randn(n, h, w, cin, device="cuda"); wt = torch.randn(cout, k, k, cin, device="cuda") * 0.05
b = torch.zeros(cout, device="cuda")
oh = (h + 2 * p - k) // s + 1; ow = (w + 2 * p - k) // s + 1
y = torch.empty(n, oh, ow, cout, device="cuda")
part = torch.empty((n * oh * ow // 128 + 8) * 2 * cout, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for mt in (1, 2):
    for nt in (1, 2, 3, 4):
        if 32 * (nt - 1) >= cout: continue
        for _ in range(reps):
            call("tbn_conv2d_fwd_tile", ptr(x), cin, ptr(wt), ptr(b), ptr(y), cout, n, h, w, cin, cout, k, s, p, 1, 0, ptr(part), mt, nt, st)
torch.cuda.synchronize()
print("done")
