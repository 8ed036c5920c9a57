"""Does the steady-state GEMM loop get better with MORE independent waves per SIMD?  One large 3x3 layer (1024 frames of
28x28, 96 -> 96: every CU holds its full complement of workgroups for almost the whole launch) run alone, and as 2 / 3
identical launches on 2 / 3 streams at once.  If the concurrent copies raise the MFMA rate well above the single
launch, the loop is limited by correlated waits (a ping-pong schedule would help); if not, by the shared VALU / LDS
issue itself."""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from attention_based_tbn_amd._lib import call, ptr
N, H, W, CIN, COUT = 1024, 28, 28, 96, 96
def mk():
    x = torch.randn(N, H, W, CIN, device="cuda"); wt = torch.randn(COUT, 3, 3, CIN, device="cuda") * 0.05
    b = torch.zeros(COUT, device="cuda"); y = torch.empty(N, H, W, COUT, device="cuda")
    part = torch.empty((N * H * W // 128 + 8) * 2 * COUT, device="cuda")
    return x, wt, b, y, part
sets = [mk() for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]
flops = 2.0 * N * H * W * COUT * 9 * CIN
def launch(s, st, flags, mt, nt):
    x, wt, b, y, part = s
    call("tbn_conv2d_fwd_tile", ptr(x), CIN, ptr(wt), ptr(b), ptr(y), COUT, N, H, W, CIN, COUT, 3, 1, 1, 1, flags, ptr(part), mt, nt, st)
for flags, name in ((0, "generic"), (4, "halo"), (8, "dma")):
    for mt, nt in ((1, 1), (1, 3), (2, 1)):
        res = []
        for k in (1, 2, 3):
            def run():
                e = torch.cuda.Event(); e.record()
                for i in range(k):
                    streams[i].wait_event(e); launch(sets[i], streams[i].cuda_stream, flags, mt, nt)
                for i in range(k):
                    torch.cuda.current_stream().wait_stream(streams[i])
            for _ in range(2): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(5): run()
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 5
            res.append(f"{k} concurrent: {t*1e3:8.1f} us = {k*flops/t/1e9:6.1f} TF/s")
        print(f"{name:8s} <{mt},{nt}>  " + "   ".join(res))
