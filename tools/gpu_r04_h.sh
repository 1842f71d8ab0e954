#!/bin/bash
# Round 4: 32-wide K chunks for the 128 x 128 g-layer tiles (A/B), then the other BASELINE configurations at HEAD.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for o in "linear_bkx_big=0" "linear_bkx_big=1" "linear_bkx_big=0" "linear_bkx_big=1"; do
  echo -n "$o  refine=0: "; DQMC_OPTS=$o timeout 300 python tools/eloc_only.py 0 2>&1 | grep "ms per"
done
bash tools/gpu_other_configs.sh
