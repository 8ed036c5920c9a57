"""Calibration of the L2 -> fabric read-request counters on KNOWN byte counts, in the access patterns of this repo
(MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern before trusting an
absolute").  Run under
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum -d DIR -o t --output-format csv \
            -- python3 scripts/pmc_calibrate.py
and feed DIR/.../t_counter_collection.csv to scripts/pmc_traffic.py --calibration.  Every launch below reads each input byte
exactly ONCE from a buffer larger than the 256-MiB Infinity Cache, so requests x bytes-per-request must equal the byte count:
  wide   torch elementwise add (16 B per lane, 1 KiB per wave instruction)              -> reads numel * 4 B
  bn     tbn_bn_relu_train_fwd's apply pass (the bn_apply kernels of the step)           -> reads P * C * 4 B (+ statistics pass)
  wgrad  conv_wgrad_kernel<2,2,2> on a pointwise layer with ONE 64 x 64 tile: four lanes x 16 B per pixel row of dy and
         of x, i.e. 64-B segments at a 256-B pitch -- the access shape of the step's dominant kernel
         -> reads M * (64 + 64) * 4 B
The script prints the byte counts it expects per kernel name (JSON on the last line)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attention_based_tbn_amd._lib import call, lib, ptr, stream_ptr  # noqa: E402

dev = torch.device("cuda")
expect = {}
# wide streaming read: 1 GiB
a = torch.empty(256 * 1024 * 1024, device=dev).normal_()
b = torch.empty_like(a)
torch.cuda.synchronize()
for _ in range(3):
    torch.add(a, 1.0, out=b)          # (a same-dtype copy_ would go through the copy engine, not a kernel)
torch.cuda.synchronize()
expect["wide_copy"] = {"kernel_contains": "elementwise", "read_bytes_per_launch": a.numel() * 4, "launches": 3}
del a, b
# weight gradient, pointwise, one 64 x 64 tile per split: dy (M, 64), x (M, 64), both read once
M, C = 2 * 1024 * 1024, 64
dy = torch.randn(M, C, device=dev)
x = torch.randn(M, C, device=dev)
dw = torch.empty(C, 1, 1, C, device=dev)
nws = lib().tbn_conv2d_wgrad_workspace_floats(1, M, 1, C, C, 1, 1, 0)
ws = torch.empty(max(1, nws), device=dev)
torch.cuda.synchronize()
for _ in range(3):
    call("tbn_conv2d_wgrad", ptr(dy), C, ptr(x), C, ptr(dw), 1, M, 1, C, C, 1, 1, 0, ptr(ws), stream_ptr())
torch.cuda.synchronize()
expect["wgrad_pointwise_64x64"] = {"kernel_contains": "conv_wgrad_kernel", "read_bytes_per_launch": 2 * M * C * 4, "launches": 3}
del dy, x
# the step's BN apply kernels on a 75264 x 96 layer are Infinity-Cache sized; here a 2^21 x 96 tensor (805 MB)
P, Cb = 2 * 1024 * 1024, 96
y = torch.randn(P, Cb, device=dev)
z = torch.empty_like(y)
gamma, beta = torch.ones(Cb, device=dev), torch.zeros(Cb, device=dev)
rm, rv = torch.zeros(Cb, device=dev), torch.ones(Cb, device=dev)
sm, sr, sc, sh = (torch.empty(Cb, device=dev) for _ in range(4))
wsb = torch.empty(lib().tbn_bn_workspace_floats(P, Cb), device=dev)
torch.cuda.synchronize()
for _ in range(3):
    call("tbn_bn_relu_train_fwd", ptr(y), P, Cb, ptr(gamma), ptr(beta), ptr(rm), ptr(rv), 0.1, 1e-5, ptr(sm), ptr(sr), ptr(sc),
         ptr(sh), ptr(z), Cb, ptr(wsb), stream_ptr())
torch.cuda.synchronize()
expect["bn_apply"] = {"kernel_contains": "bn_apply_kernel", "read_bytes_per_launch": P * Cb * 4, "launches": 3}
expect["bn_stats"] = {"kernel_contains": "bn_stats", "read_bytes_per_launch": P * Cb * 4, "launches": 3}
print(json.dumps(expect))
