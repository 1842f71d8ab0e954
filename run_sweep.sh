#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; rm -f gpurun_out/sweep.log
for cfg in "--fused-occ 2" "--fused-occ 64 --fused-sched 0 --fused-wt 4" "--fused-occ 128 --fused-sched 0 --fused-wt 4" "--fused-occ 128 --fused-sched 0 --fused-wt 8" "--fused-occ 64 --fused-sched 0 --fused-wt 2" "--fused-occ 256 --fused-sched 0 --fused-wt 4"; do
  echo "== $cfg" >> gpurun_out/sweep.log
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $cfg 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    print('ms/step', round(d['ms_per_step'], 3), 'fused us', round(d['roofline']['avg_launch_us'], 1), d['roofline']['kernel_ms_per_step'])
" >> gpurun_out/sweep.log
done
cat gpurun_out/sweep.log
