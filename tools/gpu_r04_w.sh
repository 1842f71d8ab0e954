#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in "--molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 --steps 2 --warmup 1 --repeats 3" \
           "--molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 --steps 2 --warmup 1 --repeats 3 --equilibrate 800" \
           "--molecule benzene --ansatz psiformer --walkers 2048 --n-sub 10 --steps 1 --warmup 1 --repeats 1"; do
  timeout 900 python bench.py --no-cpu-baseline $cfg 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:60], 'ms/step %.1f (%.1f..%.1f)'%(d['ms_per_step'],d['ms_per_step_min'],d['ms_per_step_max']), 'refine off', d['ms_per_step_refine_off'], d['config']['refine_engaged'])"
done
