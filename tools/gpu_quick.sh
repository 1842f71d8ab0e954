#!/bin/bash
# Quick GPU check: smoke + GPU tests + headline bench (-> gpurun_out/)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.json
nproc > gpurun_out/device.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 420 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
tail -3 gpurun_out/smoke.log; tail -15 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/bench.log | cut -c1-1500
