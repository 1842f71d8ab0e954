#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/ecp_pass.py 256 2 2>&1 | grep "ms per"
DQMC_OPTS=linear_bf=3 timeout 600 python tools/ecp_pass.py 256 2 2>&1 | grep "ms per"
rm -rf gpurun_out/ecp_trace
timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ecp_trace -- python tools/ecp_pass.py 256 1 > gpurun_out/ecp_trace.log 2>&1
python tools/trace_busy.py gpurun_out/ecp_trace 0.45 | tee gpurun_out/ecp_trace_summary.txt
find gpurun_out/ecp_trace -name "*.csv" -size +20M -delete
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py tests/test_gpu_samplers.py -x -q -m gpu 2>&1 | tail -3
