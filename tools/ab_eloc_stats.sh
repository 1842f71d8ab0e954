#!/bin/bash
# Per-kernel statistics of E_loc-only passes (serial streams) for several library builds: tools/ab_eloc_stats.sh a.so b.so ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for L in "$@"; do
  cp $L deepqmc_amd/csrc/libdqmc_hip.so
  echo "#### $L"
  tools/gpu_stats_eloc_ab.sh "multi_stream=0,pass_graph=0,$OPTS" 2>&1 | grep -E "^==|k_linear"
done
