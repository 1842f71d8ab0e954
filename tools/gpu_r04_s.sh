#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/ecp_trace
timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ecp_trace -- python tools/ecp_pass.py 256 1 > gpurun_out/ecp_trace.log 2>&1
python tools/trace_busy.py gpurun_out/ecp_trace 0.35 | tee gpurun_out/ecp_trace_summary.txt
find gpurun_out/ecp_trace -name "*.csv" -size +20M -delete
