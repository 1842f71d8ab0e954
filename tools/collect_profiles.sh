#!/bin/bash
# Copy the judged summaries of the last run_gpu_checks.sh call from gpurun_out/ (scratch) to profiles/ (tracked).
cd "$(dirname "$0")/.."
R=${1:-r01}
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" profiles/${R}_kernel_stats.csv
cp gpurun_out/parity_report.json profiles/${R}_parity_report.json
grep '^{' gpurun_out/bench.log | tail -1 > profiles/${R}_bench_1gpu.json
cp gpurun_out/other_configs.log profiles/${R}_bench_other_configs.jsonl
cp gpurun_out/pmc_hbm_traffic.json profiles/${R}_pmc_hbm_traffic.json
tail -4 gpurun_out/pytest_gpu.log > profiles/${R}_pytest_gpu_tail.txt
[ -f gpurun_out/pmc_sq.json ] && cp gpurun_out/pmc_sq.json profiles/${R}_pmc_sq_counters.json
[ -f gpurun_out/fused_profile.txt ] && grep -v amdgpu.ids gpurun_out/fused_profile.txt > profiles/${R}_fused_substep_clock_profile.txt
[ -f gpurun_out/other_configs2.log ] && cat gpurun_out/other_configs2.log >> profiles/${R}_bench_other_configs.jsonl
ls -la profiles/
