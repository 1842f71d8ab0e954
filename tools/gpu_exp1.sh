#!/bin/bash
# Experiment call: new GPU tests (pseudo-Hamiltonian) + occupancy scan of the sub-step kernel + N2 after the k_edge_sum change
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pseudo_hamiltonian.py -q > gpurun_out/exp1_ph.log 2>&1; echo "rc=$?" >> gpurun_out/exp1_ph.log
: > gpurun_out/exp1_scan.log
for w in 1024 2048 3072 4096 8192; do
  timeout 200 python bench.py --walkers $w --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --refine 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('walkers', $w, 'ms/step', round(d['ms_per_step'], 3), 'substep_us', round(d['roofline']['avg_launch_us'], 1), 'kernel_ms', {k: round(v, 3) for k, v in d['roofline']['kernel_ms_per_step'].items()})
" >> gpurun_out/exp1_scan.log 2>&1
done
timeout 300 python bench.py --molecule N2 --ansatz ferminet --n-sub 10 --steps 3 --warmup 1 --repeats 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('N2 ms/step', round(d['ms_per_step'], 2), {k: round(v, 2) for k, v in d['roofline']['kernel_ms_per_step'].items()})
" >> gpurun_out/exp1_scan.log 2>&1
tail -5 gpurun_out/exp1_ph.log; cat gpurun_out/exp1_scan.log
