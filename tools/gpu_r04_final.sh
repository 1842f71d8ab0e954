#!/bin/bash
# Refresh of the evidence the last two k_linear commits touch (LiH and N2: chained MLPs and the small Laplacian tiles)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log; tail -1 gpurun_out/smoke.log
timeout 100 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"
grep '^{' gpurun_out/bench.log | tail -1 | cut -c1-400
tools/prof_cfg.sh lih --steps 5 --warmup 5 --repeats 1 > gpurun_out/prof_lih.txt 2>&1
timeout 60 python bench.py --no-cpu-baseline --molecule N2 --ansatz ferminet --n-sub 10 --steps 3 --warmup 2 --repeats 3 2>/dev/null | grep '^{' > gpurun_out/n2_line.json; cut -c1-300 gpurun_out/n2_line.json
tools/prof_cfg.sh n2 --molecule N2 --ansatz ferminet --n-sub 10 --steps 2 --warmup 1 --repeats 1 > gpurun_out/prof_n2.txt 2>&1
tools/gpu_trace_eloc.sh 1 > /dev/null 2>&1
timeout 120 python -m pytest tests/test_gpu_parity_full.py -q -m gpu -k "lih or n2" 2>&1 | tail -2
