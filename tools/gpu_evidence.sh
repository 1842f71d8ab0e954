#!/bin/bash
# Evidence run (no tests): HBM traffic counters, headline bench, rocprofv3 kernel stats, SQ counters -> gpurun_out/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
mkdir -p gpurun_out
nproc > gpurun_out/device.log
tools/run_traffic.sh > gpurun_out/traffic.log 2>&1
cp gpurun_out/pmc_hbm_traffic.json profiles/r02_pmc_hbm_traffic.json
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf "$ROOT/gpurun_out/prof"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof" -o r02 -- python "$ROOT/bench.py" --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline > "$ROOT/gpurun_out/prof.log" 2>&1
echo "prof rc=$?" >> "$ROOT/gpurun_out/prof.log"
rm -f $(find "$ROOT/gpurun_out/prof" -name "*kernel_trace.csv")
cd "$ROOT"
[ "$1" = "pmc" ] && tools/run_pmc.sh > gpurun_out/pmc.log 2>&1
tail -2 gpurun_out/bench.log | cut -c1-900; tail -3 gpurun_out/bench.err; tail -3 gpurun_out/traffic.log; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-160
