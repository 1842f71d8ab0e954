#!/bin/bash
# Per-kernel statistics of E_loc-only passes for several option sets: tools/gpu_stats_eloc_ab.sh "opts1" "opts2" ...  -> gpurun_out/stats_eloc_<n>.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
n=0
for o in "$@"; do
  n=$((n+1)); rm -rf /tmp/stats_eloc
  DQMC_OPTS="$o" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats_eloc -o t -- python "$ROOT/tools/eloc_only.py" 0 > "$ROOT/gpurun_out/stats_eloc_$n.log" 2>&1
  f=$(find /tmp/stats_eloc -name "*kernel_stats.csv" | head -1)
  echo "== $o: $(grep 'ms per' $ROOT/gpurun_out/stats_eloc_$n.log)"
  python - "$f" <<'PY' | tee "$ROOT/gpurun_out/stats_eloc_$n.txt"
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print('%6d calls %9.1f us avg %6.2f %%  %s' % (int(r['Calls']), float(r['AverageNs'])/1e3, float(r['Percentage']), r['Name'][:90]))
PY
done
