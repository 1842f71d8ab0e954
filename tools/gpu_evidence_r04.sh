#!/bin/bash
# Round-4 evidence in one GPU call: smoke, the whole -m gpu suite, HBM traffic (VMC step + E_loc pass), headline bench with
# the CPU baseline, rocprofv3 kernel stats of the headline and of configs 3-5, SQ counters (headline + benzene), the other
# configurations, the E_loc timeline.  Everything lands in gpurun_out/; tools/collect_profiles_r04.sh copies the summaries.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); mkdir -p gpurun_out; rm -f gpurun_out/parity_report.json gpurun_out/gpu_mem.log
nproc > gpurun_out/device.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tools/run_traffic.sh > gpurun_out/traffic.log 2>&1
cp gpurun_out/pmc_hbm_traffic.json profiles/r04_pmc_hbm_traffic.json           # bench.py reports roofline.traffic from here
tools/run_traffic_eloc.sh 1 > gpurun_out/traffic_eloc.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
tools/prof_cfg.sh lih --steps 5 --warmup 5 --repeats 1 > gpurun_out/prof_lih.txt 2>&1
tools/run_pmc.sh > gpurun_out/pmc.log 2>&1
tools/gpu_other_configs.sh > gpurun_out/other_configs.txt 2>&1
tools/prof_cfg.sh n2 --molecule N2 --ansatz ferminet --n-sub 10 --steps 2 --warmup 1 --repeats 1 > gpurun_out/prof_n2.txt 2>&1
tools/prof_cfg.sh benzene --molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 --steps 1 --warmup 1 --repeats 1 > gpurun_out/prof_benzene.txt 2>&1
tools/prof_cfg.sh c4h4 --molecule cyclobutadiene_square --ansatz transpsiformer --walkers 512 --steps 1 --warmup 1 --repeats 1 > gpurun_out/prof_c4h4.txt 2>&1
tools/gpu_trace_eloc.sh 1 > /dev/null 2>&1
tools/prof_cfg.sh ecp --molecule benzene --ansatz psiformer --ecp --walkers 256 --n-sub 10 --steps 1 --warmup 1 --repeats 1 --equilibrate 100 > gpurun_out/prof_ecp.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak > gpurun_out/mfma_peak.txt 2>&1
tail -3 gpurun_out/smoke.log; tail -4 gpurun_out/pytest_gpu.log; grep '^{' gpurun_out/bench.log | tail -1 | cut -c1-700; tail -3 gpurun_out/traffic_eloc.log; cat gpurun_out/other_configs.txt | cut -c1-260
