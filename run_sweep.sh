#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; rm -f gpurun_out/sweep.log
for cfg in "" "--eloc-chunk 2048" "--eloc-chunk 1024" "--eloc-chunk 512" "--eloc-chunk 256" "--molecule N2 --ansatz ferminet --n-sub 10" "--molecule N2 --ansatz ferminet --n-sub 10 --eloc-chunk 512" "--molecule N2 --ansatz ferminet --n-sub 10 --eloc-chunk 128"; do
  echo "== $cfg" >> gpurun_out/sweep.log
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $cfg 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    print('ms/step', round(d['ms_per_step'], 3), 'evals/s', round(d['value']), 'eloc-only/s', round(d['eloc_only_evals_per_s']), 'linTF', round(d['roofline']['per_kernel_tflops'].get('linear',0),1))
" >> gpurun_out/sweep.log
done
cat gpurun_out/sweep.log
