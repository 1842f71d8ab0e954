#!/bin/bash
# A/B of library builds on the whole bench inside one gpurun call: tools/ab_bench.sh "<bench args>" a.so b.so ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
args=$1; shift
for rep in 1 2; do
for L in "$@"; do
  cp $L deepqmc_amd/csrc/libdqmc_hip.so
  timeout 900 python bench.py $args --no-cpu-baseline > gpurun_out/bench_ab.log 2> gpurun_out/bench_ab.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/bench_ab.log').read().strip().splitlines()[-1])
    print('$L', {k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ('ms_per_step','ms_per_step_refine_off','eloc_only_evals_per_s')}, {k:round(v,3) for k,v in d['roofline']['kernel_ms_per_step'].items()})
except Exception as e:
    print('$L', 'FAILED', e); print(open('gpurun_out/bench_ab.err').read()[-600:])
PY
done
done
