#!/bin/bash
# Round 4, third GPU call: -m gpu suite; mixed-precision ECP sweep on the parity fixture; ECP bench A/B at 64 walkers.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); mkdir -p gpurun_out; rm -f gpurun_out/parity_report.json gpurun_out/ab_ecp.log
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -6 gpurun_out/pytest_gpu.log
timeout 900 python tools/ecp_sweep.py > gpurun_out/ecp_sweep.log 2>&1; tail -8 gpurun_out/ecp_sweep.log | cut -c1-400
for opt in "" "--opt ecp_mixed=0"; do
  echo "## ecp 64 $opt" >> gpurun_out/ab_ecp.log
  timeout 900 python bench.py --no-cpu-baseline --molecule benzene --ansatz psiformer --ecp --walkers 64 --n-sub 2 --steps 1 --warmup 1 --repeats 2 $opt 2>/dev/null | grep '^{' >> gpurun_out/ab_ecp.log
done
python - <<'PY'
import json
for l in open('gpurun_out/ab_ecp.log'):
    if l.startswith('#'): print(l.strip()); continue
    d=json.loads(l); print('  ms/step %.1f'%d['ms_per_step'], '| %.0f /s'%d['value'], '| refine-off', d.get('ms_per_step_refine_off'), '|', d['config'].get('refine_engaged',{}).get('fraction_refined'))
PY
