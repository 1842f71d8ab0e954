#!/bin/bash
# Round 4, second GPU call: the -m gpu suite again (three-state test reworked), A/B of the split-group float64 linear tiles,
# the other BASELINE configurations at HEAD.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); mkdir -p gpurun_out; rm -f gpurun_out/parity_report.json gpurun_out/ab_split.log
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
for cfg in "--molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 --steps 2 --warmup 1 --repeats 2" \
           "--molecule cyclobutadiene_square --ansatz transpsiformer --walkers 512 --steps 2 --warmup 1 --repeats 2"; do
  for opt in "" "--opt twin.linear_f64_split=0"; do
    echo "## $cfg $opt" >> gpurun_out/ab_split.log
    timeout 900 python bench.py --no-cpu-baseline $cfg $opt 2>/dev/null | grep '^{' >> gpurun_out/ab_split.log
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/ab_split.log'):
    if l.startswith('#'): print(l.strip()); continue
    d=json.loads(l); print('  ms/step %.1f'%d['ms_per_step'], '| %.0f /s'%d['value'], '| refine-off', d.get('ms_per_step_refine_off'), '|', d['config'].get('refine_engaged',{}).get('fraction_refined'))
PY
tools/prof_cfg.sh benzene --molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 --steps 1 --warmup 1 --repeats 1 > gpurun_out/prof_benzene.txt 2>&1
head -8 gpurun_out/kernel_stats_benzene.csv | cut -c1-160
tools/prof_cfg.sh c4h4 --molecule cyclobutadiene_square --ansatz transpsiformer --walkers 512 --steps 1 --warmup 1 --repeats 1 > gpurun_out/prof_c4h4.txt 2>&1
head -8 gpurun_out/kernel_stats_c4h4.csv | cut -c1-160
