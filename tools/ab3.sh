#!/bin/bash
# A/B/C... of several builds of the library inside one gpurun call: tools/ab3.sh a.so b.so c.so -- [substep_time args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
libs=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do libs+=("$1"); shift; done; shift
for rep in 1 2; do
  for L in "${libs[@]}"; do
    cp $L deepqmc_amd/csrc/libdqmc_hip.so
    echo "$L: $(python tools/substep_time.py "$@" 2>/dev/null | tail -1)"
  done
done
