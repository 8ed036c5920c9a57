"""Per-kernel instruction mix of the main (MFMA) loop of every kernel in a gfx950 assembly listing.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude --cuda-device-only -S -o /tmp/conv_igemm.s \
        attention_based_tbn_amd/csrc/conv_igemm.hip
  python scripts/isa_scan.py /tmp/conv_igemm.s [substring]

The fp32 MFMA shares its issue port with the VALU on gfx950 (DESIGN.md hardware finding 1): every VALU instruction in a
K loop costs ~4 of the MFMA's 64 cycles, and a v_accvgpr_read / v_accvgpr_write pair per accumulator register per
iteration (a register-class copy the compiler inserts around some loop shapes) additionally drains the MFMA pipe.
The scan needs no GPU: it is how the K loops are kept lean between GPU runs.  Columns: MFMA / VALU / of which
accumulator copies / LDS / VMEM / barriers inside the outermost loop that contains the MFMAs, and VALU per MFMA
(code that only runs on a tap change is counted as if it ran every iteration: an upper bound).
"""
import collections
import re
import sys


def scan(path, want=""):
    lines = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\S+:", l)]
    rows = []
    for idx, (s, name) in enumerate(starts):
        if want and want not in name:
            continue
        e = starts[idx + 1][0] if idx + 1 < len(starts) else len(lines)
        body = lines[s:e]
        labels = {}
        for i, l in enumerate(body):
            m = re.match(r"^(\.LBB[0-9_]+):", l)
            if m:
                labels[m.group(1)] = i
        loops = []
        for i, l in enumerate(body):
            m = re.match(r"^\s+s_c?branch\w*\s+(\.LBB[0-9_]+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        if not loops:
            continue

        def count(a, b):
            c = collections.Counter()
            for s_ in body[a:b + 1]:
                t = s_.strip()
                if not t or t.startswith((".", ";")) or t.endswith(":"):
                    continue
                op = t.split()[0]
                if op.startswith("v_mfma"):
                    c["mfma"] += 1
                elif op.startswith("v_accvgpr"):
                    c["acc"] += 1
                    c["valu"] += 1
                elif op.startswith("v_"):
                    c["valu"] += 1
                elif op.startswith("ds_"):
                    c["ds"] += 1
                elif op.startswith(("buffer", "global")):
                    c["vmem"] += 1
                elif op.startswith("s_barrier"):
                    c["bar"] += 1
            return c

        best = max(loops, key=lambda ab: count(*ab)["mfma"])
        enc = [ab for ab in loops if ab[0] <= best[0] and ab[1] >= best[1]]
        big = (min(ab[0] for ab in enc), max(ab[1] for ab in enc))
        c = count(*big)
        if c["mfma"] == 0:
            continue
        regs = {}
        for l in body:
            m = re.match(r"^;\s*(NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy):\s*(\d+)", l.strip())
            if m:
                regs[m.group(1)] = int(m.group(2))
        short = re.sub(r"^_Z\d+", "", name)
        short = re.sub(r"Ev\d+\w+$", "", short)
        rows.append((short, c["mfma"], c["valu"], c["acc"], c["ds"], c["vmem"], c["bar"], c["valu"] / c["mfma"],
                     regs.get("NumVgprs", -1), regs.get("NumAgprs", -1), regs.get("ScratchSize", -1), regs.get("Occupancy", -1)))
    return rows


if __name__ == "__main__":
    for r in scan(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""):
        print("%-52s mfma %4d valu %4d acc %3d ds %3d vmem %3d bar %2d  valu/mfma %.2f  vgpr %3d agpr %3d scratch %d occ %d" % r)
