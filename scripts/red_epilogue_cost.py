"""What the fused BN-backward reduce costs a data-gradient launch (round-5 verdict item 1B): the SAME launch -- layer shape
at R frames, kernel variant, tile -- through the descriptor entry point tbn_conv_launch with the reduce epilogue off and on
(RED: per-column sums of g and g * xhat of the producer BN layer while the final dz is stored; reads y once more).
  python scripts/red_epilogue_cost.py [R=96]
Timed warm (150 untimed launches first: a cold burst runs ~12 % slow, profiles/HISTORY.md finding 13), 100 launches per cell.
Columns: us and algorithmic TFLOP/s without / with the reduce, the difference in us and in % of the launch."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attention_based_tbn_amd._lib import ConvDesc, call, lib, ptr
R = int(sys.argv[1]) if len(sys.argv) > 1 else 96
HALO, DMA, SK4 = 4, 8, 16
st = lambda: torch.cuda.current_stream().cuda_stream
# (name, H = W, layer Cin [= columns of the data gradient], layer Cout, k, variant flags, (mt, nt)): the <1,1> class of
# profiles/r05_layer_profile_rgb_R96_single_stream.txt and its <2,x> neighbours
CASES = [("conv2_3x3 dgrad", 56, 64, 192, 3, HALO, (1, 1)), ("conv2_3x3 dgrad", 56, 64, 192, 3, HALO, (1, 2)),
         ("3b_double_3x3_2 dgrad", 28, 96, 96, 3, HALO, (1, 1)), ("3b_double_3x3_2 dgrad", 28, 96, 96, 3, HALO, (2, 1)),
         ("4d_double_3x3_2 dgrad", 14, 192, 192, 3, HALO, (1, 1)), ("4d_double_3x3_2 dgrad", 14, 192, 192, 3, HALO, (2, 2)),
         ("4c pool_proj group dgrad", 14, 576, 576, 1, 0, (1, 1)), ("4c pool_proj group dgrad", 14, 576, 576, 1, 0, (1, 2)),
         ("5a_double_3x3_2 dgrad", 7, 224, 224, 3, SK4, (1, 1)), ("5b_double_3x3_reduce group dgrad", 7, 1024, 864, 1, 0, (1, 1))]
print(f"R = {R} frames; data gradient of a layer (Cin <- Cout), variant flags 4 LDS-halo / 16 split-K tile / 0 generic")
print("%-34s %-8s %-6s %9s %7s %9s %7s %8s %6s" % ("layer", "variant", "tile", "off us", "TF/s", "RED us", "TF/s", "delta us", "%"))
for name, hw, cin, cout, k, flags, (mt, nt) in CASES:
    p = (k - 1) // 2
    dy = torch.randn(R, hw, hw, cout, device="cuda")
    wt = torch.randn(cout, k, k, cin, device="cuda") * 0.05
    dx = torch.empty(R, hw, hw, cin, device="cuda")
    y = torch.randn(R * hw * hw, cin, device="cuda")
    stats = torch.stack([torch.zeros(cin), torch.ones(cin), torch.ones(cin), torch.zeros(cin)]).cuda().contiguous()
    ws = torch.empty(cout * k * k * cin, device="cuda")

    def desc(red):
        d = ConvDesc()
        d.inp, d.in_ld, d.weight, d.bias, d.out, d.out_ld = ptr(dy), cout, ptr(wt), 0, ptr(dx), cin
        d.n, d.h, d.w, d.cin, d.cout, d.ksize, d.stride, d.pad = R, hw, hw, cin, cout, k, 1, p
        d.dgrad, d.epilogue, d.flags, d.stages = 1, 0, flags, 1 if not flags else 0
        if red is not None:
            d.nred = 1
            d.red[0].y, d.red[0].y_ld, d.red[0].col_begin, d.red[0].channels = ptr(y), cin, 0, cin
            d.red[0].stat_offset, d.red[0].partial = 0, ptr(red)
            d.red_stats, d.red_stats_stride = ptr(stats), cin
        return d
    rows = lib().tbn_conv_partial_rows(C.byref(desc(None)), mt, 0)
    part = torch.empty(rows, 2, cin, device="cuda")
    flops = 2.0 * R * hw * hw * cin * cout * k * k
    out = []
    for red in (None, part):
        d = desc(red)
        if lib().tbn_conv_launch(C.byref(d), mt, nt, ptr(ws), st()) != 0:
            out.append(float("nan"))
            continue
        for _ in range(150):
            call("tbn_conv_launch", C.byref(d), mt, nt, ptr(ws), st())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            call("tbn_conv_launch", C.byref(d), mt, nt, ptr(ws), st())
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 10.0)      # us per launch
    off, on = out
    print("%-34s %-8d <%d,%d>  %9.1f %7.1f %9.1f %7.1f %8.1f %6.1f" % (name, flags, mt, nt, off, flops / off / 1e6, on, flops / on / 1e6,
                                                                      on - off, 100.0 * (on - off) / off))
