#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf "$ROOT/gpurun_out/prof_benz"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_benz" -o b -- python "$ROOT/bench.py" --molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 --steps 1 --warmup 1 --repeats 1 --no-cpu-baseline > "$ROOT/gpurun_out/prof_benz.log" 2>&1
f=$(find "$ROOT/gpurun_out/prof_benz" -name "*kernel_stats.csv" | head -1); head -14 "$f" | cut -c1-170
rm -f $(find "$ROOT/gpurun_out/prof_benz" -name "*kernel_trace.csv")
