#!/bin/bash
# A/B of two library builds on N2 / FermiNet (4096 walkers, 10 sub-steps) and benzene / Psiformer (256 walkers): tools/ab_n2.sh old.so new.so
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp deepqmc_amd/csrc/libdqmc_hip.so /tmp/keep.so
for rep in 1 2; do for L in "$@"; do
  cp $L deepqmc_amd/csrc/libdqmc_hip.so
  echo "$L N2: $(python bench.py --molecule N2 --ansatz ferminet --n-sub 10 --steps 3 --warmup 2 --repeats 3 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d["config"]["refine_engaged"]; print("ms_per_step", round(d["ms_per_step"],2), "refine_off", round(d["ms_per_step_refine_off"],2), "refined", round(r["fraction_refined"],4), "energy", d["energy"])' | cut -c1-260)"
done; done
for L in "$@"; do
  cp $L deepqmc_amd/csrc/libdqmc_hip.so
  echo "$L benzene: $(python bench.py --molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 --steps 2 --warmup 1 --repeats 2 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print("ms_per_step", round(d["ms_per_step"],2))')"
done
cp /tmp/keep.so deepqmc_amd/csrc/libdqmc_hip.so
