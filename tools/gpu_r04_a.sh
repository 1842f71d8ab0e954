#!/bin/bash
# Round 4, first GPU call: the whole -m gpu suite (new: chunked benzene, C4H4 three states, 2-rank in-library reduction when
# 2 GPUs are visible), then the float64 MFMA attention A/B on the two attention configurations + kernel stats of benzene.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); mkdir -p gpurun_out; rm -f gpurun_out/parity_report.json gpurun_out/ab_attn.log
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
for cfg in "--molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 --steps 2 --warmup 1 --repeats 2" \
           "--molecule cyclobutadiene_square --ansatz transpsiformer --walkers 512 --steps 2 --warmup 1 --repeats 2"; do
  for opt in "" "--opt twin.attention_mfma=0"; do
    echo "## $cfg $opt" >> gpurun_out/ab_attn.log
    timeout 900 python bench.py --no-cpu-baseline $cfg $opt 2>/dev/null | grep '^{' >> gpurun_out/ab_attn.log
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/ab_attn.log'):
    if l.startswith('#'): print(l.strip()); continue
    d=json.loads(l); print('  ms/step %.1f'%d['ms_per_step'], '| %.0f /s'%d['value'], '| refine-off', d.get('ms_per_step_refine_off'), '|', d['config'].get('refine_engaged',{}).get('fraction_refined'), {k: round(v,1) for k,v in d['roofline']['kernel_ms_per_step'].items()} if d.get('roofline') else None)
PY
tools/prof_cfg.sh benzene --molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 --steps 1 --warmup 1 --repeats 1 > gpurun_out/prof_benzene.txt 2>&1
head -14 gpurun_out/kernel_stats_benzene.csv | cut -c1-200
