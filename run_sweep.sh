#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; rm -f gpurun_out/sweep.log
for cfg in "--fused-wt 4" "--fused-wt 8" "--fused-wt 12" "--fused-wt 4 --fused-sched 1" "--fused-wt 8 --fused-sched 1"; do
  echo "== $cfg" >> gpurun_out/sweep.log
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $cfg >> gpurun_out/sweep.log 2>&1
done
python tools/fused_profile.py --wt 8 > gpurun_out/fprof8.log 2>&1
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "rng or metropolis or full_size" 2>&1 | tail -3
