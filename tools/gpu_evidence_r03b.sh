#!/bin/bash
# Round-3 evidence after the sub-step kernel moved to the bf16 matrix pipe (the E_loc pass and the other configurations
# are as in tools/gpu_evidence_r03.sh): smoke, the whole -m gpu suite, HBM traffic of the VMC step, headline bench with
# the CPU baseline, rocprofv3 kernel stats and SQ counters of the headline.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); mkdir -p gpurun_out; rm -f gpurun_out/parity_report.json
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tools/run_traffic.sh > gpurun_out/traffic.log 2>&1
cp gpurun_out/pmc_hbm_traffic.json profiles/r03_pmc_hbm_traffic.json           # bench.py reports roofline.traffic from here
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
tools/prof_cfg.sh lih --steps 5 --warmup 5 --repeats 1 > gpurun_out/prof_lih.txt 2>&1
tools/run_pmc.sh > gpurun_out/pmc.log 2>&1
tail -3 gpurun_out/smoke.log; tail -4 gpurun_out/pytest_gpu.log; tail -1 gpurun_out/bench.log | cut -c1-900; tail -3 gpurun_out/traffic.log
