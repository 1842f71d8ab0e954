#!/bin/bash
# Round 4: pipelined k_slogdet_mfma -- parity on the 28- / 42-electron fixtures, then benzene / C4H4 / N2 step times.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; rm -f gpurun_out/parity_report.json
timeout 1800 python -m pytest tests/test_gpu_parity_full.py tests/test_gpu_parity.py -m gpu -q -k "benzene or c4h4 or f64_every_buffer or three_states or N2 or n2" > gpurun_out/pytest_slogdet.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_slogdet.log
for cfg in "--molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 --steps 2 --warmup 1 --repeats 2" \
           "--molecule cyclobutadiene_square --ansatz transpsiformer --walkers 512 --steps 2 --warmup 1 --repeats 2"; do
  timeout 900 python bench.py --no-cpu-baseline $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); rf=d['roofline']
print(d['config']['workload'][:40], 'ms/step %.1f'%d['ms_per_step'], 'off', d.get('ms_per_step_refine_off'), {k: round(v,1) for k,v in (rf.get('float32_kernels') or rf)['kernel_ms_per_step'].items()})"
done
