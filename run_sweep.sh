#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; rm -f gpurun_out/sweep.log
for cfg in "--fused-occ 2" "--fused-occ 3 --fused-sched 0 --fused-wt 4" "--fused-occ 4 --fused-sched 0 --fused-wt 4" "--fused-occ 3 --fused-wt 2" "--fused-occ 4 --fused-wt 2" "--fused-occ 4 --fused-sched 0 --fused-wt 2" "--fused-occ 3 --fused-wt 3"; do
  echo "== $cfg" >> gpurun_out/sweep.log
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $cfg >> gpurun_out/sweep.log 2>&1
done
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "rng or metropolis or full_size" 2>&1 | tail -3
