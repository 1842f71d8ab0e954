#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/value_vs_lap.py benzene psiformer 2048 2>&1 | tail -6
timeout 900 python tools/value_vs_lap.py cyclobutadiene_square transpsiformer 1024 2>&1 | tail -5
timeout 900 python tools/value_vs_lap.py N2 ferminet 4096 2>&1 | tail -5
