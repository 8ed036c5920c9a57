#!/usr/bin/env python
"""Summarise the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, see
MI355X_MICROARCH.md HBM section) of `bench.py --profile-every 1` into HBM bytes per launch per
kernel.  FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for
wide coalesced reads, so the read side is doubled (upper bound for narrow reads).

Only the TIMED steps count: a step ends with one `opt_sgd_kernel` dispatch, so everything up to the SKIP-th such
dispatch (priming step with the per-layer tile autotune + warm-up steps) is dropped.

  python scripts/pmc_traffic.py gpurun_out/pmc_fetch/fetch_counter_collection.csv \
         gpurun_out/pmc_write/write_counter_collection.csv profiles/r02_pmc_traffic.json [SKIP=3] [note]
"""
import csv
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the kernel sources a traffic figure belongs to: bench.py refuses the file when any of them changed since
KERNEL_SOURCES = ["attention_based_tbn_amd/csrc/conv_igemm.hip", "attention_based_tbn_amd/csrc/engine.hip",
                  "attention_based_tbn_amd/csrc/tbn_kernels.h", "attention_based_tbn_amd/build.py"]


def source_hash(root=ROOT):
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(root, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("(anonymous namespace)::", "")


def load(path, counter, skip_steps):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        rows = [r for r in csv.DictReader(f) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    steps_seen, first = 0, 0
    for r in rows:
        if steps_seen >= skip_steps:
            break
        first = int(r["Dispatch_Id"])
        if "opt_sgd_kernel" in r["Kernel_Name"]:
            steps_seen += 1
    for row in rows:
        if skip_steps and int(row["Dispatch_Id"]) <= first:
            continue
        k = short(row["Kernel_Name"])
        acc[k][0] += 1
        acc[k][1] += float(row["Counter_Value"])
    return acc


def load_raw(path, skip_steps):
    """per kernel: launches and summed TCC_EA0_RDREQ / TCC_EA0_RDREQ_32B / TCC_BUBBLE (128-B requests) of the timed steps"""
    with open(path) as f:
        rows = [r for r in csv.DictReader(f)]
    names = {"TCC_EA0_RDREQ_sum": "req", "TCC_EA0_RDREQ_32B_sum": "r32", "TCC_BUBBLE_sum": "r128"}
    rows = [r for r in rows if r["Counter_Name"] in names]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    first, steps_seen, seen_disp = 0, 0, set()
    for r in rows:
        if steps_seen >= skip_steps:
            break
        d = int(r["Dispatch_Id"])
        first = d
        if "opt_sgd_kernel" in r["Kernel_Name"] and d not in seen_disp:
            steps_seen += 1
        seen_disp.add(d)
    acc = defaultdict(lambda: {"disp": set(), "req": 0.0, "r32": 0.0, "r128": 0.0})
    for r in rows:
        d = int(r["Dispatch_Id"])
        if skip_steps and d <= first:
            continue
        a = acc[short(r["Kernel_Name"])]
        a["disp"].add(d)
        a[names[r["Counter_Name"]]] += float(r["Counter_Value"])
    return {k: {"launches": len(v["disp"]), "req": v["req"], "r32": v["r32"], "r128": v["r128"]} for k, v in acc.items()}


def calibrate(calib_csv, expect_json):
    """bytes per read request of the class the box's FETCH_SIZE formula prices at 64 B (requests that are neither 32-B nor
    TCC_BUBBLE's 128-B ones), from launches whose read bytes are KNOWN (scripts/pmc_calibrate.py), per access pattern"""
    raw = load_raw(calib_csv, 0)
    with open(expect_json) as f:
        expect = json.load(f)
    out = {}
    for tag, e in expect.items():
        hits = [(k, v) for k, v in raw.items() if e["kernel_contains"] in k]
        if not hits:
            continue
        k, v = max(hits, key=lambda kv: kv[1]["req"])
        n = v["launches"]
        req, r32, r128 = v["req"] / n, v["r32"] / n, v["r128"] / n
        mid = req - r32 - r128
        out[tag] = {"kernel": k, "launches": n, "known_read_bytes_per_launch": e["read_bytes_per_launch"],
                    "rdreq_per_launch": round(req), "rdreq_32B_per_launch": round(r32), "bubble_128B_per_launch": round(r128),
                    "fetch_size_formula_bytes": round(128 * r128 + 64 * mid + 32 * r32),
                    "bytes_per_request_of_the_64B_class": round((e["read_bytes_per_launch"] - 128 * r128 - 32 * r32) / max(1.0, mid), 2)}
    return out


def main_raw(argv):
    """pmc_traffic.py --raw READS.csv WRITES.csv OUT.json SKIP NOTE CALIB.csv CALIB_EXPECT.json
    READS.csv: a --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum pass of the bench command; WRITES.csv: the
    WRITE_SIZE pass; CALIB*: the same three read counters over scripts/pmc_calibrate.py and the JSON line it printed."""
    reads, writes_csv, out_path, skip, note, calib_csv, expect_json = argv[0], argv[1], argv[2], int(argv[3]), argv[4], argv[5], argv[6]
    cal = calibrate(calib_csv, expect_json)
    def factor(tag, default):
        v = cal.get(tag, {}).get("bytes_per_request_of_the_64B_class", default)
        return v if 30.0 <= v <= 130.0 else default      # a calibration launch that was not found / matched a no-read kernel
    wide = factor("wide_copy", factor("bn_apply", 128.0))
    wg = factor("wgrad_pointwise_64x64", wide)
    raw = load_raw(reads, skip)
    write = load(writes_csv, "WRITE_SIZE", skip)
    out = {}
    for k, v in sorted(raw.items()):
        nw, w = write.get(k, [0, 0.0])
        if not v["launches"] or not nw:
            continue
        n = v["launches"]
        req, r32, r128 = v["req"] / n, v["r32"] / n, v["r128"] / n
        f = wg if "conv_wgrad_kernel" in k else wide       # bytes per request of the middle class: by calibrated access pattern
        rd = 128 * r128 + f * (req - r32 - r128) + 32 * r32
        wr = 1024.0 * w / nw
        out[k] = {"launches": n, "rdreq_per_launch": round(req), "rdreq_32B_per_launch": round(r32),
                  "bubble_128B_per_launch": round(r128), "bytes_per_request_used": f,
                  "fetch_kib_raw_per_launch": round((128 * r128 + 64 * (req - r32 - r128) + 32 * r32) / 1024.0, 2),
                  "write_kib_per_launch": round(w / nw, 2), "hbm_read_bytes_per_launch": round(rd),
                  "hbm_write_bytes_per_launch": round(wr), "hbm_bytes_per_launch": round(rd + wr)}
    tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in out.values())
    res = {"note": "reads = L2 -> fabric requests (TCC_EA0_RDREQ_sum, _32B_sum, TCC_BUBBLE_sum: one --pmc pass) priced per access "
                   "pattern with the bytes per request CALIBRATED on launches of known byte counts (`calibration`, "
                   "scripts/pmc_calibrate.py); writes = WRITE_SIZE (separate pass); per launch over the TIMED steps of bench.py "
                   f"--steps 3 --warmup 2 --profile-every 1 (B=32; the first {skip} steps are filtered out)" + (" | " + note if note else ""),
           "source_sha16": source_hash(), "workload": os.environ.get("TBN_PMC_WORKLOAD", "config4_B32"),
           "calibration": cal, "total_hbm_bytes_all_launches": tot, "kernels": out}
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(cal, indent=1))
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:14]:
        print(f"{k[:70]:70s} n={v['launches']:5d} rd={v['hbm_read_bytes_per_launch']/1e6:9.2f} MB wr={v['hbm_write_bytes_per_launch']/1e6:9.2f} MB")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--raw":
        return main_raw(sys.argv[2:])
    skip = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    fetch, write = load(sys.argv[1], "FETCH_SIZE", skip), load(sys.argv[2], "WRITE_SIZE", skip)
    out = {}
    for k in sorted(set(fetch) | set(write)):
        nf, f = fetch.get(k, [0, 0.0])
        nw, w = write.get(k, [0, 0.0])
        if not nf or not nw:
            continue
        rd = 2.0 * 1024.0 * f / nf
        wr = 1024.0 * w / nw
        out[k] = {"launches": nf, "fetch_kib_raw_per_launch": round(f / nf, 2), "write_kib_per_launch": round(w / nw, 2),
                  "hbm_read_bytes_per_launch": round(rd), "hbm_write_bytes_per_launch": round(wr),
                  "hbm_bytes_per_launch": round(rd + wr)}
    tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in out.values())
    res = {"note": "FETCH_SIZE x2 gfx950 correction applied to reads; separate --pmc passes; bytes per launch averaged over "
                   "the launches of that kernel name in the TIMED steps of bench.py --steps 3 --warmup 2 --profile-every 1 "
                   f"(B=32; the first {skip} steps -- priming with autotune + warm-up -- are filtered out)"
                   + (" | " + sys.argv[5] if len(sys.argv) > 5 else ""),
           "source_sha16": source_hash(),   # sha256 over KERNEL_SOURCES of the tree the counters were collected from
           "workload": os.environ.get("TBN_PMC_WORKLOAD", "config4_B32"),   # bench.py only quotes these figures on that workload
           "total_hbm_bytes_all_launches": tot, "kernels": out}
    with open(sys.argv[3], "w") as f:
        json.dump(res, f, indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:14]:
        print(f"{k[:70]:70s} n={v['launches']:5d} rd={v['hbm_read_bytes_per_launch']/1e6:9.2f} MB wr={v['hbm_write_bytes_per_launch']/1e6:9.2f} MB")


if __name__ == "__main__":
    main()
