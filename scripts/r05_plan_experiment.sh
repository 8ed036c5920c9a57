#!/bin/bash
# does the per-process autotune's choice matter?  N tuning runs (each writes its plans to its own TBN_PLAN_CACHE directory), then
# the best and the worst directory are re-run WITHOUT tuning: if their difference reproduces, it is the plan; if not, it is the box
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%s clips/s %.1f ms %.3f conv_stage %.4f e2e %.4f plans %s' % (sys.argv[1], d['value'], d['ms_per_step'], r['all_conv_gemm']['frac'], r['end_to_end_frac'], ' '.join(v[:6] for v in d['box']['plans'].values())))" "$1"; }
{
for i in 1 2 3 4 5 6; do
  TBN_PLAN_CACHE=/tmp/plans_$i python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "tune_$i"
done
} | tee $O/r05_plan_experiment.txt
best=$(sort -k8 -n -r $O/r05_plan_experiment.txt | head -1 | cut -d' ' -f1 | sed 's/tune_//')
worst=$(sort -k8 -n $O/r05_plan_experiment.txt | head -1 | cut -d' ' -f1 | sed 's/tune_//')
echo "best plan set: $best, worst: $worst (by conv stage)" | tee -a $O/r05_plan_experiment.txt
{
for rep in 1 2 3; do
  TBN_PLAN_CACHE=/tmp/plans_$best python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "reuse_best_$best"
  TBN_PLAN_CACHE=/tmp/plans_$worst python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "reuse_worst_$worst"
done
} | tee -a $O/r05_plan_experiment.txt
mkdir -p $O/r05_best_plans && cp /tmp/plans_$best/* $O/r05_best_plans/
echo done
