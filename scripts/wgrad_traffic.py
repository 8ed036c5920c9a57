"""Per-layer fabric traffic of the weight-gradient kernel: launches tbn_conv2d_wgrad on the 3x3 shapes of one backbone at R = 96
frames (the launches that run `conv_wgrad_kernel<2,2,0>`, the dominant kernel of bench.py's roofline), three times each, for
    rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum -- python3 scripts/wgrad_traffic.py
and, given the counter CSV as argument, prints bytes read per launch against the operands' size (dispatch order = shape order).
The aggregate over these launches is what roofline.traffic reports (1.95 x algorithmic); this table says which layers carry it."""
import csv, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [  # name, H = W, cin, cout, stride   (reference graph: core/models/bn_inception_audio.py:58-404)
    ("conv2_3x3", 56, 64, 192, 1), ("3a_3x3", 28, 64, 64, 1), ("3a_double_3x3_2", 28, 96, 96, 1), ("3b_3x3", 28, 64, 96, 1),
    ("4a_double_3x3_2", 14, 128, 128, 1), ("4d_3x3", 14, 128, 192, 1), ("4d_double_3x3_2", 14, 192, 192, 1),
    ("4e_double_3x3_1", 14, 192, 256, 1), ("5a_3x3", 7, 192, 320, 1), ("5b_double_3x3_1", 7, 192, 224, 1)]
R, REPS = 96, 3
if len(sys.argv) > 1:
    rows = list(csv.DictReader(open(sys.argv[1])))
    disp = {}
    for r in rows:
        if "conv_wgrad_kernel" not in r["Kernel_Name"]:
            continue
        d = disp.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"].split("(")[0].replace("void ", ""), "grid": int(r["Grid_Size"]),
                                                   "t": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(disp)
    assert len(ids) == len(SHAPES) * REPS, (len(ids), len(SHAPES) * REPS)
    print("%-18s %-28s %6s %9s %9s %7s %8s" % ("layer", "kernel", "wgs", "read MB", "operands", "ratio", "us"))
    tot_r = tot_a = 0.0
    for i, (name, hw, cin, cout, s) in enumerate(SHAPES):
        ds = [disp[j] for j in ids[i * REPS:(i + 1) * REPS]][1:]            # first launch of a shape: cold L2 / MALL, dropped
        rd = sum(128.0 * (d["TCC_EA0_RDREQ_sum"] - d.get("TCC_EA0_RDREQ_32B_sum", 0.0)) + 32.0 * d.get("TCC_EA0_RDREQ_32B_sum", 0.0) for d in ds) / len(ds)
        alg = R * hw * hw * (cin + cout) * 4.0
        tot_r += rd; tot_a += alg
        print("%-18s %-28s %6d %9.1f %9.1f %7.2f %8.1f" % (name, ds[0]["k"], ds[0]["grid"] // 256, rd / 1e6, alg / 1e6, rd / alg, sum(d["t"] for d in ds) / len(ds) / 1e3))
    print("sum: read %.1f MB for %.1f MB of operands = %.2f x" % (tot_r / 1e6, tot_a / 1e6, tot_r / tot_a))
    sys.exit(0)
import torch
from attention_based_tbn_amd._lib import call, ptr, lib
st = torch.cuda.current_stream().cuda_stream
for name, hw, cin, cout, s in SHAPES:
    x = torch.randn(R, hw, hw, cin, device="cuda")
    dy = torch.randn(R, hw, hw, cout, device="cuda")
    dw = torch.empty(cout, 3, 3, cin, device="cuda")
    ws = torch.empty(max(1, lib().tbn_conv2d_wgrad_workspace_floats(R, hw, hw, cin, cout, 3, s, 1)), device="cuda")
    for _ in range(REPS):
        call("tbn_conv2d_wgrad", ptr(dy), cout, ptr(x), cin, ptr(dw), R, hw, hw, cin, cout, 3, s, 1, ptr(ws), st)
    torch.cuda.synchronize()
print("done")
