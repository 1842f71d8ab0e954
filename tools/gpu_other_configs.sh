#!/bin/bash
# The other BASELINE configurations under library defaults -> gpurun_out/other_configs.log (one JSON line each)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; rm -f gpurun_out/other_configs.log
for cfg in "--molecule N2 --ansatz ferminet --n-sub 10 --steps 3 --warmup 2 --repeats 3" \
           "--molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 --steps 2 --warmup 1 --repeats 2" \
           "--molecule benzene --ansatz psiformer --walkers 2048 --n-sub 10 --steps 1 --warmup 1 --repeats 1" \
           "--molecule benzene --ansatz psiformer --ecp --walkers 64 --n-sub 2 --steps 1 --warmup 1 --repeats 2" \
           "--molecule benzene --ansatz psiformer --ecp --walkers 2048 --n-sub 10 --steps 1 --warmup 1 --repeats 1 --equilibrate 100" \
           "--molecule cyclobutadiene_square --ansatz transpsiformer --walkers 512 --steps 2 --warmup 1 --repeats 2" \
           "--molecule cyclobutadiene_square --ansatz transpsiformer --states 3 --walkers 2048 --steps 1 --warmup 1 --repeats 1"; do
  timeout 1200 python bench.py --no-cpu-baseline $cfg 2>/dev/null | grep '^{' >> gpurun_out/other_configs.log
done
python - <<'PY'
import json
for l in open('gpurun_out/other_configs.log'):
    d=json.loads(l); rf=d.get('roofline') or {}
    print(d['config']['workload'][:96], '| ms/step %.1f'%d['ms_per_step'], '| %.0f /s'%d['value'], '| refine-off', d.get('ms_per_step_refine_off'), '|', d['config'].get('refine_engaged',{}).get('fraction_refined'), '| roofline %s %.3f of %s (f64 share %s)'%(rf.get('kernel','')[:24], rf.get('frac',0), rf.get('peak'), rf.get('f64_share_of_kernel_time')))
PY
