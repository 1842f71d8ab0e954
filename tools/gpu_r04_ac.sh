#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2; do timeout 100 python tools/eloc_only.py 1 2>&1 | grep "ms per"; done
timeout 100 python tools/eloc_only.py 0 2>&1 | grep "ms per"
timeout 100 python bench.py --no-cpu-baseline --molecule N2 --ansatz ferminet --n-sub 10 --steps 3 --warmup 2 --repeats 2 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:40], 'ms/step %.1f'%d['ms_per_step'], 'refine off', d['ms_per_step_refine_off'])"
timeout 100 python tools/ecp_pass.py 256 2 2>&1 | grep "ms per"
