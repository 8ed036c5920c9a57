"""Timeline of ONE training step from a rocprofv3 --kernel-trace CSV of `bench.py --profile-steps 0`:
where the wall time of the (multi-stream) step goes, per kernel family ON THE CRITICAL PATH vs HIDDEN.

Usage: step_timeline.py <kernel_trace.csv> [step_from_end=2]

Steps are delimited by the fused optimiser launches (`opt_sgd_kernel`) that end each step.  The step's wall time is
partitioned into
    (a) time during which at least one conv GEMM runs                      -> the MFMA-bound floor is being worked on
    (b) time during which kernels run but NO conv GEMM                     -> exposed non-GEMM time, charged to the
        families running then (split evenly when several run)
    (c) device idle (no kernel of the step running: launch gaps, dependency waits)
and per family: launches, summed duration, the part that ran UNDER a GEMM ("hidden"), the part that ran with no GEMM
beside it ("exposed" = its share of (b)), and the time it was the ONLY family running.  A family's exposed time is what
the step would lose at best if that family were free; hidden time costs only through contention.
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
                     ("bn_bwd_reduce_pooled", "bn_bwd_stem_pooled"), ("bn_bwd_apply_pooled", "bn_bwd_stem_pooled"),
                     ("bn_apply_maxpool", "bn_fwd_stem_pooled"),
                     ("bn_bwd_finalize", "bn_bwd_finalize"), ("bn_bwd_reduce", "bn_bwd_reduce"), ("bn_bwd_apply", "bn_bwd_apply"),
                     ("bn_finalize", "bn_fwd_finalize"), ("bn_apply", "bn_fwd_apply"), ("bn_stats", "bn_fwd_stats"),
                     ("bn_", "bn_other"), ("maxpool", "pool"), ("avgpool", "pool"),
                     ("pool", "pool"), ("opt_sgd", "optimizer"), ("opt_", "optimizer"), ("clip", "optimizer"), ("sqnorm", "optimizer"),
                     ("nchw_to_", "layout"), ("pack_stem", "layout"), ("unpack_stem", "layout"),
                     ("spatial_mean", "heads"), ("segment_mean", "heads"), ("mul_mask", "heads"), ("dropout_fwd", "heads"), ("ce_heads", "heads"),
                     ("relu_mask_bwd", "heads"), ("colsum", "heads"), ("Cijk", "heads"),
                     ("gemm", "heads"), ("elementwise", "torch_ew"), ("CatArray", "torch_ew"), ("softmax", "torch_ew"),
                     ("nll_loss", "torch_ew"), ("reduce_kernel", "torch_ew"), ("ncclDevKernel", "rccl")):
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
gemm_time = nogemm_time = gaps = 0
alone = defaultdict(float)
hidden = defaultdict(float)
exposed = defaultdict(float)
last = t0
for t, d, fam in events:
    dt = t - last
    live = [k for k, v in active.items() if v > 0]
    if dt > 0:
        if not live:
            gaps += dt
        elif "gemm" in live:
            gemm_time += dt
            for k in live:
                if k != "gemm":
                    hidden[k] += dt
        else:
            nogemm_time += dt
            for k in live:
                exposed[k] += dt / len(live)
        if len(live) == 1:
            alone[live[0]] += dt
    active[fam] += d
    last = t
tot = defaultdict(int)
cnt = defaultdict(int)
for s, e, n, q in step:
    tot[family(n)] += e - s
    cnt[family(n)] += 1
# how many conv GEMMs / how many kernels of any kind are in flight, by share of the wall time
ev2 = []
for s, e, n, q in step:
    g = 1 if family(n) == "gemm" else 0
    ev2.append((s, 1, g))
    ev2.append((e, -1, -g))
ev2.sort()
ng = nk = 0
by_g = defaultdict(int)
by_k = defaultdict(int)
last2 = t0
for t, dk, dg in ev2:
    by_g[min(ng, 4)] += t - last2
    by_k[min(nk, 6)] += t - last2
    nk += dk
    ng += dg
    last2 = t
wall = t1 - t0
print("step wall %.3f ms (first kernel -> last kernel end), to next step start %.3f ms, %d kernels" %
      (wall / 1e6, (t_next - t0) / 1e6, len(step)))
print("partition of the wall time: >= 1 conv GEMM running %.3f ms (%.1f %%) | kernels but no GEMM %.3f ms (%.1f %%) | "
      "device idle %.3f ms (%.1f %%); gap to the next step %.3f ms" %
      (gemm_time / 1e6, 100.0 * gemm_time / wall, nogemm_time / 1e6, 100.0 * nogemm_time / wall, gaps / 1e6,
       100.0 * gaps / wall, (t_next - t1) / 1e6))
first_bwd = min((r[0] for r in step if "wgrad" in r[2] or "bn_bwd" in r[2]), default=t1)
last_fwd_gemm = max((r[1] for r in step if r[0] < first_bwd and family(r[2]) == "gemm"), default=t0)
last_bwd = max((r[1] for r in step if family(r[2]) in ("gemm", "bn_bwd_apply", "bn_bwd_stem_pooled", "splitk_reduce", "layout")), default=t1)
print("phases: backbone forward %.3f ms | heads + loss (fwd+bwd) %.3f ms | backbone backward %.3f ms | tail (optimiser ...) %.3f ms"
      % ((last_fwd_gemm - t0) / 1e6, (first_bwd - last_fwd_gemm) / 1e6, (last_bwd - first_bwd) / 1e6, (t1 - last_bwd) / 1e6))
print("conv GEMMs in flight (share of the wall time): " + "  ".join("%d%s: %.1f %%" % (k, "+" if k == 4 else "", 100.0 * v / (t1 - t0)) for k, v in sorted(by_g.items())))
print("kernels in flight, any kind:                    " + "  ".join("%d%s: %.1f %%" % (k, "+" if k == 6 else "", 100.0 * v / (t1 - t0)) for k, v in sorted(by_k.items())))
print("%-22s %8s %10s %12s %12s %10s" % ("family", "kernels", "sum ms", "hidden ms", "exposed ms", "alone ms"))
for fam in sorted(tot, key=lambda k: -tot[k]):
    print("%-22s %8d %10.3f %12.3f %12.3f %10.3f" % (fam, cnt[fam], tot[fam] / 1e6, hidden[fam] / 1e6, exposed[fam] / 1e6,
                                                     alone[fam] / 1e6))
# optional third argument "bins": the same partition per millisecond of the step -- where the periods with 0 or 1 GEMM in flight lie
if len(sys.argv) > 3 and sys.argv[3] == "bins":
    nb = int(wall // 1_000_000) + 1
    share = [[0.0] * 5 for _ in range(nb)]
    names = [defaultdict(float) for _ in range(nb)]
    ng = 0
    last3 = t0
    for t, dk, dg in ev2:
        s = last3
        while s < t:                      # spread [last3, t) over the bins it crosses
            b = int((s - t0) // 1_000_000)
            e = min(t, t0 + (b + 1) * 1_000_000)
            share[min(b, nb - 1)][min(ng, 4)] += e - s
            s = e
        ng += dg
        last3 = t
    for s, e, n, q in step:
        if family(n) == "gemm":
            continue
        b = min(int((s - t0) // 1_000_000), nb - 1)
        names[b][family(n)] += e - s
    print("per millisecond: share of the bin with 0 / 1 / 2 / 3 / 4+ conv GEMMs in flight | largest non-GEMM families started in the bin (kernel ms)")
    for b in range(nb):
        tot_b = sum(share[b]) or 1.0
        top = sorted(names[b].items(), key=lambda kv: -kv[1])[:3]
        print("%3d ms  " % b + " ".join("%5.1f" % (100.0 * v / tot_b) for v in share[b]) + "  | " +
              ", ".join("%s %.2f" % (k, v / 1e6) for k, v in top))
# optional third argument "turn": the serial turn between forward and backward of every complete step in the trace -- from the end of
# the last forward backbone GEMM to the start of the first backward backbone GEMM (backbone GEMMs: conv kernels on a modality
# stream, i.e. not on the null stream the heads run on) -- and the kernels of the selected step's turn
if len(sys.argv) > 3 and sys.argv[3] == "turn":
    def backbone_gemm(r):
        return family(r[2]) == "gemm" and r[3] not in ("(nil)", "0x0", "")
    turns = []
    for gi in range(1, len(groups)):
        sa, sb = groups[gi - 1][-1] + 1, groups[gi][-1] + 1
        st_rows = rows[sa:sb]
        ce = [r for r in st_rows if "ce_heads_bwd" in r[2]]
        big = [r for r in st_rows if backbone_gemm(r)]
        if not ce or not big:
            continue
        lf = max(r[1] for r in big if r[0] < ce[0][0])
        fb = min(r[0] for r in big if r[0] > ce[0][0])
        turns.append((fb - lf) / 1e3)
    ce = [r for r in step if "ce_heads_bwd" in r[2]][0]
    big = [r for r in step if backbone_gemm(r)]
    lf = max(r[1] for r in big if r[0] < ce[0])
    fb = min(r[0] for r in big if r[0] > ce[0])
    print("turn (last forward backbone GEMM end -> first backward backbone GEMM start), us per step: " + " ".join("%.0f" % t for t in turns))
    for s, e, n, q in step:
        if lf - 50_000 <= s <= fb + 50_000:
            print("%9.3f %7.1f us q=%s %s" % ((s - t0) / 1e6, (e - s) / 1e3, q[-6:], n.replace("void ", "")[:70]))
