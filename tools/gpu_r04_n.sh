#!/bin/bash
# Round 4: eight-wave (split) attention kernel -- parity, then A/B against the four-wave kernel on the float64 twin and in float32.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity_full.py tests/test_gpu_parity.py -m gpu -q -k "benzene_psiformer_256 or c4h4_transpsiformer_512 or benzene_ecp or f64_every_buffer or attention" > gpurun_out/pytest_n.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_n.log
for cfg in "--molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 --steps 2 --warmup 1 --repeats 2" \
           "--molecule cyclobutadiene_square --ansatz transpsiformer --walkers 512 --steps 2 --warmup 1 --repeats 2"; do
  for opt in "" "--opt twin.attention_split=0" "--opt attention_split=1" ""; do
    timeout 900 python bench.py --no-cpu-baseline $cfg $opt 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); rf=d['roofline']
f32=(rf.get('float32_kernels') or rf)['kernel_ms_per_step']; f64=(rf if 'float32_kernels' in rf else rf.get('float64_twin',{})).get('kernel_ms_per_step',{})
print(d['config']['workload'][:30], '$opt'[:28], 'ms/step %.1f'%d['ms_per_step'], 'off %.1f'%d.get('ms_per_step_refine_off'), 'f32 att %.1f'%(f32.get('attention',0)), 'f64 att %.1f lin %.1f'%(f64.get('attention',0), f64.get('linear',0)))"
  done
done
