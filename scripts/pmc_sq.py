#!/usr/bin/env python
"""Aggregate a rocprofv3 --pmc counter_collection.csv of scripts/layer_profile.py by kernel name
(last training iteration only: from the last input-layout (nchw_to_*_pad) dispatch on)."""
import csv, re, sys
from collections import defaultdict, OrderedDict

def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*$", "", n); return n

rows = list(csv.DictReader(open(sys.argv[1])))
disp = OrderedDict()
for r in rows:
    d = int(r["Dispatch_Id"])
    e = disp.setdefault(d, {"k": short(r["Kernel_Name"]), "grid": int(r["Grid_Size"]), "vgpr": r["VGPR_Count"], "agpr": r["Accum_VGPR_Count"], "lds": r["LDS_Block_Size"],
                            "t": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    e[r["Counter_Name"]] = float(r["Counter_Value"])
ids = list(disp)
start = max(i for i in ids if disp[i]["k"].startswith(("nchw_to_s2d_pad", "nchw_to_nhwc_pad")))
agg = defaultdict(lambda: defaultdict(float))
for i in ids:
    if i < start: continue
    e = disp[i]
    a = agg[(e["k"], e["vgpr"], e["agpr"], e["lds"])]
    a["n"] += 1
    for k, v in e.items():
        if isinstance(v, (int, float)) and k not in ("grid",): a[k] += v
names = sorted({k for a in agg.values() for k in a} - {"n", "t"})
# MFMA pipe utilisation: SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 SIMDs (= 64 cycles per
# v_mfma_f32_32x32x2_f32, checked against SQ_INSTS_MFMA); GRBM_GUI_ACTIVE sums the active cycles of the 8 XCDs
util = "SQ_VALU_MFMA_BUSY_CYCLES" in names and "GRBM_GUI_ACTIVE" in names
raw_busy = lambda a: 100.0 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["GRBM_GUI_ACTIVE"] / 8 * 1024)
# Calibration (round-5 verdict item 4: a kernel's raw MFMA_busy came out BELOW its own FLOP fraction, which cannot be if both
# are right): scripts/layer_profile.py ... burst appends launches of the pure-MFMA register loop whose FLOPs are known
# (1024 workgroups x 4 waves x 1500 iterations x 16 MFMAs of 4096 FLOP) -- its FLOP fraction of the 157.3-TFLOP/s peak from
# the dispatch timestamps is what its busy figure SHOULD read; every kernel's raw figure is scaled by (that / burst's raw).
cal = clk = None
burst = [a for k, a in agg.items() if k[0].startswith("mfma_burst_kernel")]
if util and burst and burst[0]["GRBM_GUI_ACTIVE"]:
    b = burst[0]
    flop_frac = 100.0 * (b["n"] * 1024 * 4 * 1500 * 16 * 4096.0) / (b["t"] * 1e-9) / 157.3e12
    cal = flop_frac / raw_busy(b)
    clk = b["GRBM_GUI_ACTIVE"] / 8 / (b["t"] * 1e-9)        # shader clock while the burst ran (cycles per second)
    print(f"# calibration: mfma_burst_kernel raw MFMA_busy {raw_busy(b):.1f} %, its FLOP fraction of 157.3 TFLOP/s {flop_frac:.1f} % "
          f"-> MFMA_busy_cal = raw x {cal:.3f}; shader clock {clk / 1e9:.3f} GHz")
    # Round 6: the counter itself is right (the burst reads what its FLOPs say) -- what made a kernel's busy figure fall BELOW
    # its own FLOP fraction (round-5 verdict) is the DENOMINATOR: GRBM_GUI_ACTIVE of a dispatch covers 3 - 12 % more time than
    # the kernel's own [start, end] for launches of 40 - 100 us (dispatch / drain inside the counter window).  MFMA_busy_kt_% =
    # busy cycles / (kernel time x the burst's clock x 1024 SIMDs) is the figure that compares with a FLOP fraction from kernel
    # times; MFMA_exec_% = executed MFMA FLOPs (SQ_INSTS_MFMA x 4096, padding included) / kernel time / 157.3 TFLOP/s.
print("kernel vgpr agpr lds | n us " + " ".join(names) + (" | MFMA_busy_%" if util else "") +
      (" MFMA_busy_cal_% MFMA_busy_kt_% MFMA_exec_% window/kernel_time" if cal else ""))
for key, a in sorted(agg.items(), key=lambda kv: -kv[1]["t"])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    extra = f" | {raw_busy(a):5.1f}" if util and a["GRBM_GUI_ACTIVE"] else ""
    if cal and extra:
        kt = a["t"] * 1e-9
        extra += (f" {raw_busy(a) * cal:5.1f} {100.0 * a['SQ_VALU_MFMA_BUSY_CYCLES'] / (kt * clk * 1024):5.1f}"
                  f" {100.0 * a.get('SQ_INSTS_MFMA', 0.0) * 4096.0 / kt / 157.3e12:5.1f} {a['GRBM_GUI_ACTIVE'] / 8 / clk / kt:5.3f}")
    print(f"{key[0][:44]:44s} v{key[1]:>3s} a{key[2]:>3s} l{key[3]:>6s} | {int(a['n']):4d} {a['t']/1e3:9.1f} " + " ".join(f"{a[k]:.4g}" for k in names) + extra)
