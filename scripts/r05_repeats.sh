#!/bin/bash
for i in 1 2 3 4 5; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('run %s: %.2f clips/s  %.3f ms/step  end to end %.4f  conv stage %.4f [%.4f, %.4f]  dominant %.4f  step_gpu_ms median %.3f min %.3f max %.3f (step %d)  plans %s' % (sys.argv[1], d['value'], d['ms_per_step'], r['end_to_end_frac'], r['all_conv_gemm']['frac'], r['all_conv_gemm']['frac_min'], r['all_conv_gemm']['frac_max'], r['frac'], d['step_gpu_ms']['median'], d['step_gpu_ms']['min'], d['step_gpu_ms']['max'], d['step_gpu_ms']['max_at_step'], ' '.join(v[:6] for v in d['box']['plans'].values())))" $i
done
