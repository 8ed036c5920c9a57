"""Per-tile timing of the three forward implicit-GEMM variants (register-staged generic kernel, LDS-halo 3x3 kernel,
LDS-DMA staging) on one conv shape, BN-statistics epilogue (the training forward): us and TFLOP/s per (MT, NT).
REPS / WARM (environment): timed / untimed launches per cell, default 10 / 2 = a cold 1-ms burst; REPS=100 WARM=150 times
with the shader clock warm (profiles/HISTORY.md finding 13: ~12 % faster, and the only setting in which 1 % differences resolve)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attention_based_tbn_amd._lib import call, ptr, lib
n, h, w, cin, cout, k, s, p = [int(v) for v in sys.argv[1:9]]
x = torch.randn(n, h, w, cin, device="cuda"); wt = torch.randn(cout, k, k, cin, device="cuda") * 0.05; b = torch.zeros(cout, device="cuda")
oh = (h + 2 * p - k) // s + 1; ow = (w + 2 * p - k) // s + 1
y = torch.empty(n, oh, ow, cout, device="cuda")
part = torch.empty((n * oh * ow // 128 + 8) * 2 * cout, device="cuda")
st = torch.cuda.current_stream().cuda_stream
flops = 2.0 * n * oh * ow * cout * k * k * cin
def run(flags, mt, nt, reps=int(os.environ.get('REPS', '10'))):
    args = (ptr(x), cin, ptr(wt), ptr(b), ptr(y), cout, n, h, w, cin, cout, k, s, p, 1, flags, ptr(part), mt, nt, st)
    if lib().tbn_conv2d_fwd_tile(*args) != 0: return float("nan")
    for _ in range(int(os.environ.get('WARM', '2'))): call("tbn_conv2d_fwd_tile", *args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call("tbn_conv2d_fwd_tile", *args)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print(f"shape {sys.argv[1:9]}  ideal {flops/157.3e12*1e6:.1f} us")
print("tile    generic        halo           dma")
for mt in (1, 2):
    for nt in (1, 2, 3, 4):
        if 32 * (nt - 1) >= cout: continue
        t = [run(f, mt, nt) for f in (0, 4, 8)]
        print(f"<{mt},{nt}>  " + "  ".join(f"{v:7.1f} us {flops/v/1e6:5.1f}" for v in t))
