#!/bin/bash
# Round 4: -m gpu suite; dual chained MLPs A/B on the E_loc pass and on the headline step.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); mkdir -p gpurun_out; rm -f gpurun_out/parity_report.json
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
for o in "mlp_dual=1,twin.mlp_dual=1" "mlp_dual=0,twin.mlp_dual=0" "mlp_dual=1,twin.mlp_dual=1" "mlp_dual=0,twin.mlp_dual=0"; do
  echo -n "$o  refine=1: "; DQMC_OPTS=$o timeout 300 python tools/eloc_only.py 1 2>&1 | grep "ms per"
  echo -n "$o  refine=0: "; DQMC_OPTS=$o timeout 300 python tools/eloc_only.py 0 2>&1 | grep "ms per"
done
for o in "" "--opt mlp_dual=0 --opt twin.mlp_dual=0"; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $o 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$o ms/step %.3f'%d['ms_per_step'], 'off', d.get('ms_per_step_refine_off'), 'eloc-only', d.get('eloc_only_evals_per_s'))"
done
