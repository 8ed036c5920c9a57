"""Timeline of ONE training step from a rocprofv3 --kernel-trace CSV of `bench.py --profile-every 0`:
where the wall time of the multi-stream step goes.  Usage: step_timeline.py <kernel_trace.csv> [step_from_end=2]

Steps are delimited by the fused optimiser launches (`opt_sgd_kernel`) that end each step.  Reported: wall time,
device-idle gaps, the phases (forward / heads + loss / backward / optimiser) by first / last kernel of each kind, and
per kernel family the summed duration and the time during which ONLY that family was running (exposed time).
"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
opt = [i for i, r in enumerate(rows) if "opt_sgd" in r[2]]
groups = []           # the optimiser launches that end a step (several multi-tensor launches, back to back)
for i in opt:
    if groups and rows[i][0] - rows[groups[-1][-1]][1] < 1_000_000:
        groups[-1].append(i)
    else:
        groups.append([i])
assert len(groups) > back + 1, "not enough steps in the trace"
a, b = groups[-back - 1][-1] + 1, groups[-back][-1] + 1
step = rows[a:b]
t0, t1 = step[0][0], max(r[1] for r in step)
t_next = rows[b][0] if b < len(rows) else t1


def family(name):
    n = name.replace("void ", "")
    for key, fam in (("conv_wgrad", "gemm"), ("conv_igemm", "gemm"), ("conv_halo", "gemm"), ("conv_dma", "gemm"),
                     ("conv_pair", "gemm"), ("conv_sk4", "gemm"), ("splitk_reduce", "splitk_reduce"), ("weight_flip", "weight_flip"),
                     ("bn_bwd", "bn_backward"), ("bn_", "bn_forward"), ("maxpool", "pool"), ("avgpool", "pool"),
                     ("pool", "pool"), ("opt_sgd", "optimizer"), ("clip", "optimizer"), ("sqnorm", "optimizer"),
                     ("nchw_to_s2d", "layout"), ("pack_stem", "layout"), ("unpack_stem", "layout"),
                     ("spatial_mean", "heads"), ("Cijk", "heads"), ("gemm", "heads"), ("elementwise", "torch_ew"),
                     ("reduce_kernel", "torch_ew"), ("ncclDevKernel", "rccl")):
        if key in n:
            return fam
    return "other:" + n.split("<")[0].split("(")[0][:40]


events = []
for s, e, n, q in step:
    fam = family(n)
    events.append((s, 1, fam))
    events.append((e, -1, fam))
events.sort()
active = defaultdict(int)
busy = 0
excl = defaultdict(int)
gaps = 0
last = t0
for t, d, fam in events:
    live = [k for k, v in active.items() if v > 0]
    if live:
        busy += t - last
        if len(live) == 1:
            excl[live[0]] += t - last
    else:
        gaps += t - last
    active[fam] += d
    last = t
tot = defaultdict(int)
cnt = defaultdict(int)
for s, e, n, q in step:
    tot[family(n)] += e - s
    cnt[family(n)] += 1
print("step wall %.3f ms (first kernel -> last kernel end), to next step start %.3f ms, %d kernels" %
      ((t1 - t0) / 1e6, (t_next - t0) / 1e6, len(step)))
print("device idle inside the step %.3f ms, gap to the next step %.3f ms" % (gaps / 1e6, (t_next - t1) / 1e6))
first_bwd = min((r[0] for r in step if "wgrad" in r[2] or "bn_bwd" in r[2]), default=t1)
last_fwd_gemm = max((r[1] for r in step if r[0] < first_bwd and family(r[2]) == "gemm"), default=t0)
last_bwd = max((r[1] for r in step if family(r[2]) in ("gemm", "bn_backward", "splitk_reduce", "layout")), default=t1)
print("phases: backbone forward %.3f ms | heads + loss (fwd+bwd) %.3f ms | backbone backward %.3f ms | tail (optimiser ...) %.3f ms"
      % ((last_fwd_gemm - t0) / 1e6, (first_bwd - last_fwd_gemm) / 1e6, (last_bwd - first_bwd) / 1e6, (t1 - last_bwd) / 1e6))
print("%-28s %8s %10s %12s" % ("family", "kernels", "sum ms", "exposed ms"))
for fam in sorted(tot, key=lambda k: -tot[k]):
    print("%-28s %8d %10.3f %12.3f" % (fam, cnt[fam], tot[fam] / 1e6, excl[fam] / 1e6))
