#!/bin/bash
set -eo pipefail
for arm in early late; do
  flag=""; [ $arm = late ] && flag="--no-early-flip"
  timeout -k 10 300 python bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 --timeline-steps 0 --timeline /tmp/tl_$arm.csv $flag > /dev/null 2>&1
  python - $arm <<'PY' > gpurun_out/r06r_turn_$arm.txt
import csv, sys
arm = sys.argv[1]
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"],r.get("Queue_Id","")) for r in csv.DictReader(open('/tmp/tl_%s.csv' % arm))]
rows.sort()
opt=[i for i,r in enumerate(rows) if "opt_sgd" in r[2]]
groups=[]
for i in opt:
    if groups and rows[i][0]-rows[groups[-1][-1]][1]<1_000_000: groups[-1].append(i)
    else: groups.append([i])
turns=[]
for gi in range(1, len(groups)):
    a,b=groups[gi-1][-1]+1,groups[gi][-1]+1
    step=rows[a:b]
    ce=[r for r in step if "ce_heads_bwd" in r[2]][0]
    gemm=lambda n: any(k in n for k in ("conv_wgrad","conv_igemm","conv_halo","conv_dma","conv_pair","conv_sk4"))
    big=[r for r in step if gemm(r[2]) and r[3] not in ("(nil)","0x0","")]
    last_fwd=max(r[1] for r in big if r[0]<ce[0])
    first_bwd=min(r[0] for r in big if r[0]>ce[0])
    turns.append((first_bwd-last_fwd)/1e3)
    wall=(max(r[1] for r in step)-step[0][0])/1e6
    if gi==len(groups)-1:
        t0=step[0][0]
        for s,e,n,q in step:
            if last_fwd-50_000 <= s <= first_bwd+50_000:
                print("%9.3f %7.1f us q=%s %s" % ((s-t0)/1e6,(e-s)/1e3,q[-6:],n.replace("void ","")[:70]))
print("turn (last forward backbone GEMM end -> first backward backbone GEMM start), us per step:", " ".join("%.0f"%t for t in turns))
PY
  tail -1 gpurun_out/r06r_turn_$arm.txt
done
