#!/bin/bash
# A/B of two builds of the library inside one gpurun call on the E_loc pass: tools/ab_eloc.sh old.so new.so
cd "${GRAFT_REPO_ROOT:-/root/repo}"
A=$1; B=$2; shift 2
cp deepqmc_amd/csrc/libdqmc_hip.so /tmp/keep.so
for rep in 1 2; do
  for L in $A $B; do
    cp $L deepqmc_amd/csrc/libdqmc_hip.so
    echo "$L: $(python tools/eloc_only.py 0 2>/dev/null | tail -1) | $(python tools/eloc_only.py 1 2>/dev/null | tail -1)"
  done
done
cp /tmp/keep.so deepqmc_amd/csrc/libdqmc_hip.so
