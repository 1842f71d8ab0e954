#!/bin/bash
# A/B of two builds of the library inside one gpurun call: tools/ab.sh old.so new.so [substep_time args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
A=$1; B=$2; shift 2
for rep in 1 2; do
  for L in $A $B; do
    cp $L deepqmc_amd/csrc/libdqmc_hip.so
    echo "$L: $(python tools/substep_time.py "$@" 2>/dev/null | tail -1)"
  done
done
