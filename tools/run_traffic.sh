#!/bin/bash
# HBM traffic of every kernel of one VMC step: two counter passes (FETCH_SIZE / WRITE_SIZE do not fit together).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$ROOT/gpurun_out/pmc_$c"
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$ROOT/gpurun_out/pmc_$c" -o t -- python "$ROOT/bench.py" --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline "$@" > "$ROOT/gpurun_out/pmc_$c.log" 2>&1
done
cd "$ROOT"
python tools/pmc_traffic.py $(find gpurun_out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) gpurun_out/pmc_hbm_traffic.json "${WORKLOAD:-LiH/paulinet/4096/f32}"
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE   # keep the merge small
