#!/bin/bash
# Round-5 GPU call E: float64 linear kernels without the scratch round trip of the weight prefetch -- headline, N2, benzene 256, C4H4 512
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
: > gpurun_out/ab_scratch.jsonl
run() { echo "# $1" >> gpurun_out/ab_scratch.jsonl; shift; timeout 600 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '^{' >> gpurun_out/ab_scratch.jsonl; }
run LiH --steps 20 --warmup 5 --min-seconds 4
run N2 --molecule N2 --ansatz ferminet --n-sub 10 --steps 5 --warmup 2 --min-seconds 3
run benzene256 --molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 --steps 2 --warmup 1 --repeats 2
run c4h4_512 --molecule cyclobutadiene_square --ansatz transpsiformer --walkers 512 --steps 2 --warmup 1 --repeats 2
run ecp64 --molecule benzene --ansatz psiformer --ecp --walkers 64 --n-sub 2 --steps 1 --warmup 1 --repeats 2
python - <<'P'
import json
for l in open('gpurun_out/ab_scratch.jsonl'):
    if l.startswith('#'): print(l.strip()); continue
    d=json.loads(l); c=d['config']['refine_engaged']; rf=d['roofline']
    t=rf.get('float64_twin') or (rf if rf.get('peak')==78.6 else {})
    print('   ms/step %.3f  value %.0f  eloc_only %.0f  refined %s  refine_off %s | f64 linear %.1f TF/s exec %s' % (d['ms_per_step'], d['value'], d['eloc_only_evals_per_s'] or 0, c.get('fraction_refined'), d.get('ms_per_step_refine_off'), (t.get('per_kernel_tflops') or {}).get('linear',0), (t.get('executed') or {}).get('linear')))
P
