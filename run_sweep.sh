#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; rm -f gpurun_out/sweep.log
python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3 >> gpurun_out/sweep.log
for cfg in "" "--molecule benzene --ansatz psiformer --ecp --walkers 64 --steps 1 --warmup 1 --n-sub 2" "--molecule N2 --ansatz ferminet --n-sub 10 --steps 3 --warmup 1"; do
  echo "== $cfg" >> gpurun_out/sweep.log
  timeout 900 python bench.py --no-cpu-baseline $cfg 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    print('ms/step', round(d['ms_per_step'], 3), 'evals/s', round(d['value'],1), 'eloc-only/s', round(d['eloc_only_evals_per_s'],1), 'TF', d['roofline']['per_kernel_tflops'], d['roofline']['kernel_ms_per_step'])
" >> gpurun_out/sweep.log
done
cat gpurun_out/sweep.log
