#!/bin/bash
set -eo pipefail
timeout -k 10 300 python bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 --timeline-steps 0 --timeline /tmp/tl_4.csv > /dev/null 2>&1
python scripts/step_timeline.py /tmp/tl_4.csv 1 bins > gpurun_out/r06p_timeline_bins_config4.txt 2>&1
python - <<'PY' > gpurun_out/r06p_timeline_streams.txt
# per stream (queue): the kernels of the last full step in order with start offsets, for the first and last 3 ms of the backward
import csv
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"],r.get("Queue_Id","")) for r in csv.DictReader(open('/tmp/tl_4.csv'))]
rows.sort()
opt=[i for i,r in enumerate(rows) if "opt_sgd" in r[2]]
groups=[]
for i in opt:
    if groups and rows[i][0]-rows[groups[-1][-1]][1]<1_000_000: groups[-1].append(i)
    else: groups.append([i])
a,b=groups[-2][-1]+1,groups[-1][-1]+1
step=rows[a:b]; t0=step[0][0]
for s,e,n,q in step:
    print("%9.3f %8.1f us q=%s %s" % ((s-t0)/1e6,(e-s)/1e3,q,n.replace("void ","")[:90]))
PY
gzip -f gpurun_out/r06p_timeline_streams.txt
