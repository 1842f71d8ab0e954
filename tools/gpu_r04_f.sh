#!/bin/bash
# HIP streams share a few hardware queues (in order per queue): does the deferred float64 pass overlap with more queues?
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for q in 4 8 16 24; do
  for o in "refine_defer=0" "refine_defer=1" "refine_defer=1,twin.multi_stream=0,twin.dual_stream=0" "refine_defer=1,twin.pass_graph=0,twin.multi_stream=0,twin.dual_stream=0"; do
    echo -n "GPU_MAX_HW_QUEUES=$q  "; GPU_MAX_HW_QUEUES=$q DQMC_OPTS=$o timeout 300 python tools/trace_defer.py 2>&1 | grep "ms per step"
  done
done
