#!/bin/bash
# A/B of library options inside one GPU call: tools/gpu_ab_opt.sh "<bench args>" opt1=v opt2=v ...   (each option set is one run)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
args=$1; shift
for o in "$@"; do
  opts=""; for kv in ${o//,/ }; do opts="$opts --opt $kv"; done
  timeout 900 python bench.py $args --no-cpu-baseline $opts > gpurun_out/bench_ab.log 2> gpurun_out/bench_ab.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/bench_ab.log').read().strip().splitlines()[-1])
    print('$o', {k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ('ms_per_step','ms_per_step_refine_off','eloc_only_evals_per_s')}, d['config']['refine_engaged']['fraction_refined'], {k:round(v,3) for k,v in d['roofline']['kernel_ms_per_step'].items()}, round(d['roofline']['avg_launch_us'],1))
except Exception as e:
    print('$o', 'FAILED', e); print(open('gpurun_out/bench_ab.err').read()[-600:])
PY
done
