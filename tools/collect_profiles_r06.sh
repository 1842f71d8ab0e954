#!/bin/bash
# Copy the judged summaries of the last tools/gpu_evidence_r06.sh call from gpurun_out/ (scratch) to profiles/ (tracked).
cd "$(dirname "$0")/.."
R=r06
cp gpurun_out/kernel_stats_lih.csv profiles/${R}_kernel_stats.csv
for t in n2 benzene c4h4 ecp; do cp gpurun_out/kernel_stats_$t.csv profiles/${R}_kernel_stats_$t.csv; done
cp gpurun_out/parity_report.json profiles/${R}_parity_report.json
grep '^{' gpurun_out/bench.log | tail -1 > profiles/${R}_bench_1gpu.json
cp gpurun_out/other_configs.log profiles/${R}_bench_other_configs.jsonl
cp gpurun_out/pmc_hbm_traffic.json profiles/${R}_pmc_hbm_traffic.json
cp gpurun_out/pmc_hbm_traffic_eloc.json profiles/${R}_pmc_hbm_traffic_eloc_pass.json
tail -4 gpurun_out/pytest_gpu.log > profiles/${R}_pytest_gpu_tail.txt
cp gpurun_out/pmc_sq.json profiles/${R}_pmc_sq_counters.json
cp gpurun_out/trace_eloc.txt profiles/${R}_eloc_pass_timeline.txt
cp gpurun_out/mfma_peak.txt profiles/${R}_mfma_peak.txt
ls -la profiles/ | grep r06
for t in n2 benzene c4h4; do cp gpurun_out/pmc_sq_$t.json profiles/${R}_pmc_sq_counters_$t.json; done
