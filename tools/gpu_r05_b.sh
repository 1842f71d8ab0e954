#!/bin/bash
# Round-5 GPU call B: the parity tests under the exponential-tail threshold rule, and what the miss rate costs (headline, N2).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; rm -f gpurun_out/parity_report.json gpurun_out/gpu_mem.log
timeout 1500 python -m pytest tests/test_gpu_parity_full.py -m gpu -q > gpurun_out/pytest_gpu_b.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_b.log
: > gpurun_out/ab_miss.jsonl
for o in "" "--opt refine_miss_e9=1000" "--opt refine_miss_e9=10000" "--refine 0"; do
  echo "# LiH $o" >> gpurun_out/ab_miss.jsonl
  timeout 200 python bench.py --steps 20 --warmup 5 --min-seconds 3 --no-cpu-baseline $o 2>/dev/null >> gpurun_out/ab_miss.jsonl
  echo "# N2 $o" >> gpurun_out/ab_miss.jsonl
  timeout 300 python bench.py --molecule N2 --ansatz ferminet --n-sub 10 --steps 5 --warmup 2 --min-seconds 3 --no-cpu-baseline $o 2>/dev/null >> gpurun_out/ab_miss.jsonl
done
echo "# benzene 256" >> gpurun_out/ab_miss.jsonl
timeout 400 python bench.py --molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 --steps 2 --warmup 1 --min-seconds 3 --no-cpu-baseline 2>/dev/null >> gpurun_out/ab_miss.jsonl
tail -6 gpurun_out/pytest_gpu_b.log
python - <<'P'
import json
for l in open('gpurun_out/ab_miss.jsonl'):
    if l.startswith('#'): print(l.strip()); continue
    if not l.startswith('{'): continue
    d=json.loads(l); c=d['config']['refine_engaged']
    print('   ms/step %.3f  value %.0f  eloc_only %.0f  refined frac %s  thr %.1f  refine_off %s' % (d['ms_per_step'], d['value'], d['eloc_only_evals_per_s'] or 0, c.get('fraction_refined'), c.get('score_threshold', 0), d.get('ms_per_step_refine_off')))
P
