#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python tools/substep_time.py fused_wt=4 fused_wt=8 fused_wt=16 fused_wt=2 fused_wt=4 2>&1 | grep -v amdgpu.ids > gpurun_out/exp6.log
cat gpurun_out/exp6.log
