#!/bin/bash
# Round 4: whole -m gpu suite (memory log), then benzene / C4H4 step times with the NCB-templated attention.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; rm -f gpurun_out/parity_report.json gpurun_out/gpu_mem.log
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
sort -n gpurun_out/gpu_mem.log | head -3
for cfg in "--molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 --steps 2 --warmup 1 --repeats 2" \
           "--molecule cyclobutadiene_square --ansatz transpsiformer --walkers 512 --steps 2 --warmup 1 --repeats 2"; do
  timeout 900 python bench.py --no-cpu-baseline $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); rf=d['roofline']
print(d['config']['workload'][:40], 'ms/step %.1f'%d['ms_per_step'], 'off', d.get('ms_per_step_refine_off'), 'f64', {k: round(v,1) for k,v in (rf if 'float32_kernels' in rf else rf.get('float64_twin',{})).get('kernel_ms_per_step',{}).items()})"
done
