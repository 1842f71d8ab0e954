#!/bin/bash
# Round-5 GPU call C: options of the float64 twin at the batch sizes the miss-rate rule gives it (~1100 LiH / ~1400 N2 walkers)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
: > gpurun_out/ab_twin.txt
for o in "" "twin.linear_bkx=1" "twin.linear_bkx=4" "twin.mlp_fuse=0" "twin.multi_stream=0" "twin.split_bcast=0" "twin.pass_graph=0" ""; do
  echo "# LiH DQMC_OPTS=$o" >> gpurun_out/ab_twin.txt
  DQMC_OPTS=$o timeout 120 python tools/eloc_only.py 1 2>/dev/null | tail -1 >> gpurun_out/ab_twin.txt
done
cat gpurun_out/ab_twin.txt
