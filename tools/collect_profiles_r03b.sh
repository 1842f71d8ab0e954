#!/bin/bash
# Copy the judged summaries of the last tools/gpu_evidence_r03b.sh call from gpurun_out/ (scratch) to profiles/ (tracked);
# the E_loc-pass and other-configuration summaries stay those of tools/gpu_evidence_r03.sh.
cd "$(dirname "$0")/.."
R=r03
cp gpurun_out/kernel_stats_lih.csv profiles/${R}_kernel_stats.csv
cp gpurun_out/parity_report.json profiles/${R}_parity_report.json
grep '^{' gpurun_out/bench.log | tail -1 > profiles/${R}_bench_1gpu.json
cp gpurun_out/pmc_hbm_traffic.json profiles/${R}_pmc_hbm_traffic.json
tail -4 gpurun_out/pytest_gpu.log > profiles/${R}_pytest_gpu_tail.txt
grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -1 >> profiles/${R}_pytest_gpu_tail.txt
cp gpurun_out/pmc_sq.json profiles/${R}_pmc_sq_counters.json
ls -la profiles/ | grep r03
