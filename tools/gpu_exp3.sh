#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for w in 1024 2048 4096; do python tools/wg_timeline.py --walkers $w 2>&1 | grep -v amdgpu.ids; done > gpurun_out/exp3_timeline.log 2>&1
cat gpurun_out/exp3_timeline.log
