#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for w in 4 1024 4096; do
  python tools/fused_profile.py --substep 1 --walkers $w > gpurun_out/exp2_prof_$w.log 2>&1
done
python tools/substep_time.py > gpurun_out/exp2_substep_time.log 2>&1
tail -3 gpurun_out/exp2_substep_time.log
