"""How far in front of the wait that needs it is every global load of a kernel's hottest loop issued?

Usage: python scripts/isa_wait_scan.py conv_igemm.s conv_halo_kernel [more name fragments ...]
       (assembly from `hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only [-mllvm -amdgpu-mfma-vgpr-form] file.hip`)

For every kernel whose mangled name contains one of the fragments: finds the loop with the most MFMAs, replays two trips
of it with an in-order queue of outstanding vector-memory operations (vmcnt counts loads and stores in issue order) and
reports, per `s_waitcnt vmcnt(N)`, how many MFMAs were issued between the youngest operation that wait forces to have
completed and the wait itself.  A small distance = a memory round trip in the open.  This is how the LDS-halo kernel's
`vmcnt(1)` right behind the 13 halo loads of the NEXT channel chunk was found (a prefetch under a run-time condition
makes the compiler's wait-count pass assume the worst where registers are reused), and the STFT kernel's `vmcnt(0)` behind
its conditional twiddle prefetch.  No GPU needed.
"""
import re
import sys


def main():
    path, want = sys.argv[1], sys.argv[2:]
    lines = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\S+:", l)]
    for si, (i0, name) in enumerate(starts):
        if not any(w in name for w in want):
            continue
        i1 = starts[si + 1][0] if si + 1 < len(starts) else len(lines)
        body = lines[i0:i1]
        labels = {}
        for i, l in enumerate(body):
            m = re.match(r"^(\.LBB[0-9_]+):", l)
            if m:
                labels[m.group(1)] = i
        best = None
        for i, l in enumerate(body):
            m = re.match(r"^\s+s_c?branch\w*\s+(\.LBB[0-9_]+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                n = sum(1 for s in body[labels[m.group(1)]:i + 1] if s.strip().startswith("v_mfma"))
                if best is None or n > best[0]:
                    best = (n, labels[m.group(1)], i)
        if not best or best[0] < 8:
            continue
        seg = [s.strip() for s in body[best[1]:best[2] + 1] if s.strip() and not s.strip().startswith(";")]
        queue, mfma, waits = [], 0, []
        for trip in range(2):
            for s in seg:
                op = s.split()[0]
                if op.startswith("v_mfma"):
                    mfma += 1
                elif op.startswith(("buffer_load", "global_load", "buffer_store", "global_store", "buffer_atomic", "global_atomic")):
                    queue.append(mfma)
                elif op == "s_waitcnt":
                    m = re.search(r"vmcnt\((\d+)\)", s)
                    if m and len(queue) > int(m.group(1)):
                        n = int(m.group(1))
                        if trip == 1:
                            waits.append((mfma - queue[len(queue) - n - 1], n, len(queue)))
                        queue = queue[len(queue) - n:] if n else []
        if waits:
            w = min(waits)
            tight = sum(1 for x in waits if x[0] < 12)
            print(f"{name[:72]:72s} loop of {best[0]:4d} MFMAs: tightest wait {w[0]:3d} MFMAs behind the load it needs "
                  f"(vmcnt({w[1]}), {w[2]} in flight); {tight} of {len(waits)} waits closer than 12 MFMAs")


if __name__ == "__main__":
    main()
