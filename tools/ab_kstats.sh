#!/bin/bash
# per-kernel average durations of the E_loc pass alone, one stream (DQMC_SERIAL=1), for each library given: tools/ab_kstats.sh a.so b.so
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); mkdir -p gpurun_out
cp deepqmc_amd/csrc/libdqmc_hip.so /tmp/keep.so
for L in "$@"; do
  cp $L deepqmc_amd/csrc/libdqmc_hip.so
  n=$(basename $L .so)
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks_$n && DQMC_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$n -o s -- python "$ROOT/tools/eloc_only.py" ${REFINE:-0} > /tmp/ks_$n.log 2>&1)
  f=$(find /tmp/ks_$n -name "*kernel_stats.csv" | head -1)
  echo "== $L $(tail -1 /tmp/ks_$n.log)"
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(__import__("os").environ.get("NTOP","14"))]: print('%9.1f us x %4s  %5.1f%%  %s' % (float(r['AverageNs']) / 1e3, r['Calls'], float(r['Percentage']), r['Name'][:100]))
PY
done
cp /tmp/keep.so deepqmc_amd/csrc/libdqmc_hip.so
