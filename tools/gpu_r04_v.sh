#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/ecp_pass.py 256 2 2>&1 | grep "ms per"
DQMC_OPTS=slogdet_mfma=3 timeout 600 python tools/ecp_pass.py 256 2 2>&1 | grep "ms per"
for cfg in "--molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 --steps 2 --warmup 1 --repeats 2" \
           "--molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 --steps 2 --warmup 1 --repeats 2 --opt slogdet_mfma=3" \
           "--molecule cyclobutadiene_square --ansatz transpsiformer --walkers 512 --steps 2 --warmup 1 --repeats 2" \
           "--molecule cyclobutadiene_square --ansatz transpsiformer --walkers 512 --steps 2 --warmup 1 --repeats 2 --opt slogdet_mfma=3"; do
  timeout 900 python bench.py --no-cpu-baseline $cfg 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:60], 'ms/step %.1f'%d['ms_per_step'], 'refine off', d['ms_per_step_refine_off'], 'slogdet ms', d['roofline']['kernel_ms_per_step'].get('slogdet'))"
done
timeout 1200 python -m pytest tests/test_gpu_samplers.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
