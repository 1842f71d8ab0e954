#!/bin/bash
# Round 4, fourth GPU call: -m gpu suite; headline bench with the deferred float64 pass (default) and the synchronous one.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); mkdir -p gpurun_out; rm -f gpurun_out/parity_report.json
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_defer.log 2> gpurun_out/bench_defer.err; echo "rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_defer.log').read().strip().splitlines()[-1])
print('ms/step %.3f'%d['ms_per_step'], 'sync', d.get('ms_per_step_sync_refine'), 'refine-off', d.get('ms_per_step_refine_off'), 'eloc-only', d.get('eloc_only_evals_per_s'), 'frac', d['roofline']['frac'], d['config'].get('refine_engaged'))
print(d['energy'])
PY
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --molecule N2 --ansatz ferminet --n-sub 10 --steps 3 --warmup 2 --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N2 ms/step %.2f'%d['ms_per_step'], 'sync', d.get('ms_per_step_sync_refine'), 'off', d.get('ms_per_step_refine_off'))"
