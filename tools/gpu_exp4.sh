#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/exp4.log
for ab in 0 8; do
  echo "== ablate $ab" >> gpurun_out/exp4.log
  python tools/fused_profile.py --substep 1 --walkers 1024 --quiet 1 --opt fused_ablate=$ab 2>/dev/null | grep "^wave 0" >> gpurun_out/exp4.log
  python tools/substep_time.py fused_ablate=$ab 2>/dev/null | tail -1 >> gpurun_out/exp4.log
done
cat gpurun_out/exp4.log | cut -c1-700
