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


def main():
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
           "total_hbm_bytes_all_launches": tot, "kernels": out}
    with open(sys.argv[3], "w") as f:
        json.dump(res, f, indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:14]:
        print(f"{k[:70]:70s} n={v['launches']:5d} rd={v['hbm_read_bytes_per_launch']/1e6:9.2f} MB wr={v['hbm_write_bytes_per_launch']/1e6:9.2f} MB")


if __name__ == "__main__":
    main()
