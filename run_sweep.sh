#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; rm -f gpurun_out/sweep.log
python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3 >> gpurun_out/sweep.log
for cfg in "" "--fused 0"; do
  echo "== $cfg" >> gpurun_out/sweep.log
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $cfg >> gpurun_out/sweep.log 2>&1
done
tail -40 gpurun_out/sweep.log
