#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/exp5.log
python tools/substep_time.py fused=1 2>/dev/null | tail -1 >> gpurun_out/exp5.log
python tools/fused_profile.py --substep 1 --walkers 1024 --quiet 1 2>/dev/null | grep "^wave 0" >> gpurun_out/exp5.log
python tools/wg_timeline.py --walkers 4096 2>/dev/null | grep -v "running at" >> gpurun_out/exp5.log
python tools/wg_timeline.py --walkers 1024 2>/dev/null | head -3 >> gpurun_out/exp5.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3 >> gpurun_out/exp5.log
cat gpurun_out/exp5.log | cut -c1-600
