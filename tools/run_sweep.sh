#!/bin/bash
# Scratch sweep used during tuning: GPU parity tests, then bench.py over a list of option sets.
#   gpurun -- ./run_sweep.sh "" "--fused-sched 1" "--molecule N2 --ansatz ferminet --n-sub 10"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; rm -f gpurun_out/sweep.log
python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3 >> gpurun_out/sweep.log
[ $# -eq 0 ] && set -- ""
for cfg in "$@"; do
  echo "== $cfg" >> gpurun_out/sweep.log
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $cfg 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    print('ms/step', round(d['ms_per_step'], 3), 'evals/s', round(d['value']), 'eloc/s', round(d['eloc_only_evals_per_s']), {k: round(v, 2) for k, v in d['roofline']['kernel_ms_per_step'].items()})
" >> gpurun_out/sweep.log
done
cat gpurun_out/sweep.log
