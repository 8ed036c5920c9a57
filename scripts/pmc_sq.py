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
print("kernel vgpr agpr lds | n us " + " ".join(names) + (" | MFMA_busy_%" if util else ""))
for key, a in sorted(agg.items(), key=lambda kv: -kv[1]["t"])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    extra = f" | {100.0 * a['SQ_VALU_MFMA_BUSY_CYCLES'] / (a['GRBM_GUI_ACTIVE'] / 8 * 1024):5.1f}" if util and a["GRBM_GUI_ACTIVE"] else ""
    print(f"{key[0][:44]:44s} v{key[1]:>3s} a{key[2]:>3s} l{key[3]:>6s} | {int(a['n']):4d} {a['t']/1e3:9.1f} " + " ".join(f"{a[k]:.4g}" for k in names) + extra)
