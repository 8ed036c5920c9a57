#!/bin/bash
mkdir -p gpurun_out
for rep in 1 2 3; do
  for b in 0 8; do
    echo "halo_bias=$b $(TBN_TUNE_HALO_BIAS=$b python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-every 0 2>/dev/null | grep -o 'ms_per_step.: [0-9.]*')"
  done
done
TBN_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 3 --warmup 1 --batch-per-gpu 8 > gpurun_out/c10_gloo2.json 2> gpurun_out/c10_gloo2.err
echo "gloo rehearsal rc=$?"; cat gpurun_out/c10_gloo2.json | cut -c1-1500; tail -3 gpurun_out/c10_gloo2.err
