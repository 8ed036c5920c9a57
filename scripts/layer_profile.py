"""Per-layer conv-GEMM efficiency of one backbone (fwd + bwd, training mode) on the GPU.
layer_profile.py cin H W N [burst]: with `burst`, three launches of the pure-MFMA register loop (tbn_diag_mfma_burst, 1024
workgroups x 1500 iterations: known FLOPs, ~99 % of the pipe) follow the profiled iteration -- the calibration launch
scripts/pmc_sq.py scales SQ_VALU_MFMA_BUSY_CYCLES by (round-5 verdict item 4)."""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attention_based_tbn_amd.core.models.bn_inception import BNInception
from attention_based_tbn_amd._lib import lib
cin, H, W, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
net = BNInception(1000, cin).cuda().train()
net.use_aux_stream = False   # one stream: an event bracket times exactly one kernel
net.use_branch_streams = False
x = torch.randn(N, cin, H, W, device="cuda")
L = lib()
for it in range(3):
    if it == 2:
        L.tbn_profile_reset(); L.tbn_profile_enable(2)
    y = net(x); y.sum().backward()
L.tbn_profile_enable(0)
if len(sys.argv) > 5 and sys.argv[5] == "burst":
    from attention_based_tbn_amd._lib import call, ptr, stream_ptr
    sink, fl = torch.zeros(16, device="cuda"), C.c_double()
    for _ in range(3):
        call("tbn_diag_mfma_burst", ptr(sink), 1024, 1500, C.byref(fl), stream_ptr())
    torch.cuda.synchronize()
name = C.create_string_buffer(160); rows = []
for i in range(L.tbn_profile_num_entries()):
    n, ms, fl = C.c_long(), C.c_double(), C.c_double()
    L.tbn_profile_entry(i, name, 160, C.byref(n), C.byref(ms), C.byref(fl))
    rows.append((name.value.decode(), n.value, ms.value, fl.value))
tot_ms = sum(r[2] for r in rows); tot_fl = sum(r[3] for r in rows)
print(f"total conv-GEMM {tot_ms:.2f} ms, {tot_fl/1e9:.1f} GFLOP, {tot_fl/tot_ms/1e9:.1f} TFLOP/s")
rows.sort(key=lambda r: -r[2])
for k, n, ms, fl in rows:
    print(f"{k:78s} {ms*1e3:8.1f} us {fl/1e9:7.2f} GF {fl/ms/1e9:6.1f} TF/s")
