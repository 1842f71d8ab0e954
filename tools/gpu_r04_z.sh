#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for o in ecp_max_cfg=16384 ecp_max_cfg=32768 ecp_max_cfg=65536 ecp_max_cfg=131072 ecp_max_cfg=262144; do echo -n "$o  "; DQMC_OPTS=$o timeout 600 python tools/ecp_pass.py 256 2 2>&1 | grep "ms per"; done
