#!/bin/bash
# Round 4: 32-wide K chunks for value-only rows (A/B on the three larger configurations).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; rm -f gpurun_out/ab_bkxval.log
for cfg in "--molecule N2 --ansatz ferminet --n-sub 10 --steps 3 --warmup 2 --repeats 3" \
           "--molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 --steps 2 --warmup 1 --repeats 2" \
           "--molecule cyclobutadiene_square --ansatz transpsiformer --walkers 512 --steps 2 --warmup 1 --repeats 2"; do
  for opt in "" "--opt linear_bkx_val=1 --opt twin.linear_bkx_val=1"; do
    echo "## $cfg $opt" >> gpurun_out/ab_bkxval.log
    timeout 900 python bench.py --no-cpu-baseline $cfg $opt 2>/dev/null | grep '^{' >> gpurun_out/ab_bkxval.log
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/ab_bkxval.log'):
    if l.startswith('#'): print(l.strip()[:150]); continue
    d=json.loads(l); print('  ms/step %.2f'%d['ms_per_step'], '| refine-off', d.get('ms_per_step_refine_off'))
PY
