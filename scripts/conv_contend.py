"""How much does an implicit-GEMM conv slow down while an HBM-bound kernel runs on another stream?"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attention_based_tbn_amd._lib import call, ptr
n, h, w, cin, cout, k, s, p = [int(v) for v in sys.argv[1:9]]
mt, nt = int(sys.argv[9]), int(sys.argv[10])
DEV = "cuda"
x = torch.randn(n, h, w, cin, device=DEV); wt = torch.randn(cout, k, k, cin, device=DEV) * 0.05; b = torch.zeros(cout, device=DEV)
oh = (h + 2 * p - k) // s + 1; ow = (w + 2 * p - k) // s + 1
y = torch.empty(n, oh, ow, cout, device=DEV)
big_a = torch.empty(1 << 28, device=DEV); big_b = torch.empty(1 << 28, device=DEV)   # 1 GiB each
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
flops = 2.0 * n * oh * ow * cout * k * k * cin
def gemm(reps):
    args = (ptr(x), cin, ptr(wt), ptr(b), ptr(y), cout, n, h, w, cin, cout, k, s, p, 0, 0, 0, mt, nt, s1.cuda_stream)
    for _ in range(reps): call("tbn_conv2d_fwd_tile", *args)
def timed(hog):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if hog:
        with torch.cuda.stream(s2):
            for _ in range(6): big_b.copy_(big_a)        # ~6 x 2 GiB of traffic
    with torch.cuda.stream(s1):
        gemm(2)
        e0.record(s1); gemm(20); e1.record(s1)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20
gemm(3); torch.cuda.synchronize()
a, c = timed(False), timed(True)
print(f"shape {sys.argv[1:9]} tile <{mt},{nt}>: alone {a*1e3:.1f} us ({flops/a/1e9:.1f} TF/s)   beside an HBM-bound copy {c*1e3:.1f} us ({flops/c/1e9:.1f} TF/s)  x{c/a:.2f}")
