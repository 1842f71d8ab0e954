#!/bin/bash
# rocprofv3 kernel stats of the N2 / FermiNet configuration -> gpurun_out/prof_n2 (copy the summary to profiles/)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf "$ROOT/gpurun_out/prof_n2"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_n2" -o n2 -- python "$ROOT/bench.py" --molecule N2 --ansatz ferminet --n-sub 10 --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline > "$ROOT/gpurun_out/prof_n2.log" 2>&1
cd "$ROOT"
rm -f $(find gpurun_out/prof_n2 -name "*kernel_trace.csv")
f=$(find gpurun_out/prof_n2 -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-160
