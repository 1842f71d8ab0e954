#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/ecp_pass.py 256 2 2>&1 | grep "ms per"
timeout 900 python tools/value_vs_lap.py benzene psiformer 2048 2>&1 | tail -4
for cfg in "--molecule cyclobutadiene_square --ansatz transpsiformer --walkers 512 --steps 2 --warmup 1 --repeats 2"; do
  timeout 900 python bench.py --no-cpu-baseline $cfg 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:60], 'ms/step %.1f'%d['ms_per_step'], 'refine off', d['ms_per_step_refine_off'], 'attention ms', d['roofline']['kernel_ms_per_step'].get('attention'))"
done
timeout 1200 python -m pytest tests/test_gpu_parity_full.py tests/test_gpu_samplers.py -x -q -m gpu 2>&1 | tail -2
