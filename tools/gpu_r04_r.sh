#!/bin/bash
# epilogues that fetch their residuals in batches + the unrolled value-row epilogue: before/after numbers
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/ecp_pass.py 256 2 2>&1 | grep "ms per"
for i in 1 2; do timeout 300 python tools/eloc_only.py 1 2>&1 | grep "ms per"; done
timeout 300 python tools/eloc_only.py 0 2>&1 | grep "ms per"
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --min-seconds 4 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('LiH ms/step', d['ms_per_step'], 'eloc-only', d['eloc_only_evals_per_s'], 'refine off', d['ms_per_step_refine_off'])"
for cfg in "--molecule N2 --ansatz ferminet --n-sub 10 --steps 3 --warmup 2 --repeats 2" \
           "--molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 --steps 2 --warmup 1 --repeats 2" \
           "--molecule cyclobutadiene_square --ansatz transpsiformer --walkers 512 --steps 2 --warmup 1 --repeats 2"; do
  timeout 900 python bench.py --no-cpu-baseline $cfg 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:60], 'ms/step %.1f'%d['ms_per_step'], 'refine off', d['ms_per_step_refine_off'], d['config'].get('refine_engaged',{}).get('fraction_refined'))"
done
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py -x -q -m gpu 2>&1 | tail -3
