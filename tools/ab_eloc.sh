#!/bin/bash
# A/B of two builds of the library inside one gpurun call (same box): tools/ab_eloc.sh old.so new.so -> E_loc-only pass time and the headline step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
A=$1; B=$2
cp deepqmc_amd/csrc/libdqmc_hip.so /tmp/keep.so
for rep in 1 2; do
  for L in $A $B; do
    cp $L deepqmc_amd/csrc/libdqmc_hip.so
    echo "$L: $(python tools/eloc_only.py 2>/dev/null | tail -1) | refine 0: $(python tools/eloc_only.py 0 2>/dev/null | tail -1)"
    echo "$L: $(python bench.py --steps 20 --warmup 5 --repeats 8 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print("ms_per_step", d["ms_per_step"], "value", d["value"], {k:v for k,v in d["config"].items() if "refin" in k})')"
  done
done
cp /tmp/keep.so deepqmc_amd/csrc/libdqmc_hip.so
