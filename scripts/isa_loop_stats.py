"""Instruction mix of the hottest loop of a kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only).

Usage: python scripts/isa_loop_stats.py conv_igemm.s 'conv_halo_kernelILi1ELi1ELi1ELb0E' [--dump]
Finds every backward branch (loop), keeps the loop body with the most v_mfma instructions and prints the count of
MFMA / VALU / SALU / DS / VMEM / waitcnt / barrier instructions in it (VALU per MFMA is what the fp32 MFMA pipe pays).
"""
import re
import sys


def kernel_body(lines, pat):
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\S+:", l) and pat in l.split(":")[0]:
            start = i
            break
    if start is None:
        raise SystemExit("kernel not found: " + pat)
    for j in range(start + 1, len(lines)):
        if lines[j].strip().startswith("s_endpgm"):
            return lines[start:j + 1]
    return lines[start:]


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "ds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    dump = "--dump" in sys.argv
    lines = open(path).read().split("\n")
    body = kernel_body(lines, pat)
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            labels[m.group(1)] = i
    best = None
    for i, l in enumerate(body):
        m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB[0-9_]+)", l) or re.match(r"^\s+s_branch\s+(\.LBB[0-9_]+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            seg = body[labels[m.group(1)]:i + 1]
            n = sum(1 for s in seg if s.strip().startswith("v_mfma"))
            if best is None or n > best[0]:
                best = (n, labels[m.group(1)], i)
    print(body[0])
    for l in body:
        if any(k in l for k in (".vgpr_count", ".sgpr_count", "ScratchSize", "Occupancy", "NumVgprs", "NumAgprs", "LDSByteSize", "TotalNumVgprs")):
            print("  ", l.strip())
    if best is None:
        print("no loop")
        return
    seg = body[best[1]:best[2] + 1]
    cnt = {}
    ops = {}
    for s in seg:
        t = s.strip()
        if not t or t.startswith((".", ";")) or t.endswith(":"):
            continue
        op = t.split()[0]
        c = classify(op)
        cnt[c] = cnt.get(c, 0) + 1
        if c in ("valu", "salu"):
            ops[op] = ops.get(op, 0) + 1
    print("hottest loop: %d instructions" % sum(cnt.values()), cnt)
    if cnt.get("mfma"):
        print("VALU per MFMA: %.2f" % (cnt.get("valu", 0) / cnt["mfma"]))
    print("  ", sorted(ops.items(), key=lambda kv: -kv[1])[:25])
    if dump:
        print("\n".join(seg))


if __name__ == "__main__":
    main()
