#!/bin/bash
# Round-6 evidence in one GPU call: smoke, the whole -m gpu suite, HBM traffic (VMC step + E_loc pass), headline bench with
# the CPU baseline, rocprofv3 kernel stats of the headline and of configs 3-5, SQ counters (headline per family; N2 / benzene /
# C4H4 per kernel instantiation, i.e. the float64 kernels that make up 65 % of configs 4-5), the other configurations, the
# E_loc timeline, the MFMA rates with the clock they ran at.  Everything lands in gpurun_out/; tools/collect_profiles_r06.sh
# copies the summaries into profiles/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); mkdir -p gpurun_out; [ -z "$SKIP_PYTEST" ] && rm -f gpurun_out/parity_report.json gpurun_out/gpu_mem.log
nproc > gpurun_out/device.log; rocm-smi --showclocks >> gpurun_out/device.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
if [ -z "$SKIP_PYTEST" ]; then timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; fi    # (SKIP_PYTEST=1: the suite ran in a call of its own at the same commit)
tools/run_traffic.sh > gpurun_out/traffic.log 2>&1
cp gpurun_out/pmc_hbm_traffic.json profiles/r06_pmc_hbm_traffic.json           # bench.py reports roofline.traffic_from_profile from here
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
PMC_FULLNAME=1 PMC_OUT=pmc_sq_n2 tools/run_pmc.sh --molecule N2 --ansatz ferminet --n-sub 10 > gpurun_out/pmc_n2.log 2>&1
PMC_FULLNAME=1 PMC_OUT=pmc_sq_benzene tools/run_pmc.sh --molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 > gpurun_out/pmc_benzene.log 2>&1
PMC_FULLNAME=1 PMC_OUT=pmc_sq_c4h4 tools/run_pmc.sh --molecule cyclobutadiene_square --ansatz transpsiformer --walkers 512 > gpurun_out/pmc_c4h4.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak > gpurun_out/mfma_peak.txt 2>&1; rocm-smi --showclocks >> gpurun_out/mfma_peak.txt 2>&1
find gpurun_out -type f -size +8M -delete; du -sh gpurun_out            # (gpurun merges at most 64 MiB back -- and nothing at all above that)
tail -3 gpurun_out/smoke.log; tail -4 gpurun_out/pytest_gpu.log; grep '^{' gpurun_out/bench.log | tail -1 | cut -c1-700; tail -3 gpurun_out/traffic_eloc.log; cat gpurun_out/other_configs.txt | cut -c1-260; cat gpurun_out/mfma_peak.txt | cut -c1-220
