#!/bin/bash
# E_loc-only pass time and refined walkers per call for a list of library option sets: tools/ab_opts.sh "" "linear_bf=1" "mlp_fuse=0,linear_bf=1"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for o in "$@"; do
  echo "opts [$o]: $(DQMC_OPTS=$o python tools/eloc_only.py 2>/dev/null | tail -1) | refine 0: $(DQMC_OPTS=$o python tools/eloc_only.py 0 2>/dev/null | tail -1)"
done
