#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak | tee gpurun_out/mfma_peak.txt
timeout 600 python tools/ecp_pass.py 256 2 2>&1 | grep "ms per"
