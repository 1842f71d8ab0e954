#!/bin/bash
# rocprofv3 kernel stats of one bench configuration: tools/prof_cfg.sh <tag> <bench.py args...> -> gpurun_out/prof_<tag>/
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf "$ROOT/gpurun_out/prof_$tag"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_$tag" -o "$tag" -- python "$ROOT/bench.py" --no-cpu-baseline "$@" > "$ROOT/gpurun_out/prof_$tag.log" 2>&1
echo "rc=$?" >> "$ROOT/gpurun_out/prof_$tag.log"
f=$(find "$ROOT/gpurun_out/prof_$tag" -name "*kernel_stats.csv" | head -1)
cp "$f" "$ROOT/gpurun_out/kernel_stats_$tag.csv"
rm -rf "$ROOT/gpurun_out/prof_$tag"
head -12 "$ROOT/gpurun_out/kernel_stats_$tag.csv" | cut -c1-170
