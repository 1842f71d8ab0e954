#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; rm -f gpurun_out/sweep.log
python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3 >> gpurun_out/sweep.log
for cfg in "" "--molecule LiH --ansatz ferminet" "--molecule LiH --ansatz psiformer --steps 3"; do
  echo "== $cfg" >> gpurun_out/sweep.log
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $cfg 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    print('ms/step', round(d['ms_per_step'], 3), 'evals/s', round(d['value']), 'dom us', round(d['roofline']['avg_launch_us'], 1), 'TF', round(d['roofline']['achieved'],1), d['roofline']['kernel'][:30], d['roofline']['kernel_ms_per_step'])
" >> gpurun_out/sweep.log
done
cat gpurun_out/sweep.log
