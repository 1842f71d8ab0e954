cd "${GRAFT_REPO_ROOT:-/root/repo}"
for o in "" "linear_bf=1" "linear_bf=0" "linear_bkx=2" "linear_bkx=4" "linear_bf=1,linear_bkx=4"; do
  echo "opts [$o]: $(DQMC_OPTS=$o python tools/eloc_only.py 2>/dev/null | tail -1) | refine 0: $(DQMC_OPTS=$o python tools/eloc_only.py 0 2>/dev/null | tail -1)"
done
