#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/calib_spread.py benzene psiformer 256 2>&1 | grep seed | tee gpurun_out/calib_spread.txt
