#!/bin/bash
# (investigation aid) ablation timings of k_linear with the PROBE build of the library (tools/build_probe_lib.sh: -DDQMC_LIN_PROBE hooks of kernel_linear.hip:
# 1 no MFMAs, 2 no epilogue, 4 no A-tile loads, 8 no barriers): tools/probe_lin.sh probe.so "0 1 2 4 8 ..."
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd)
cp deepqmc_amd/csrc/libdqmc_hip.so /tmp/keep.so; cp $1 deepqmc_amd/csrc/libdqmc_hip.so
for P in $2; do
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pl && DQMC_LIN_PROBE=$P DQMC_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pl -o s -- python "$ROOT/tools/eloc_only.py" 0 > /tmp/pl.log 2>&1)
  f=$(find /tmp/pl -name "*kernel_trace.csv" | head -1)
  python - "$f" $P <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if 'k_linear' in n: d[n.split('k_linear')[1][:34]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
out = []
for k, v in d.items():
    v = v[len(v) // 2:]
    out.append('%s max %.1f mean %.1f' % (k, max(v), sum(v) / len(v)))
print('probe', sys.argv[2], ' | '.join(sorted(out)))
PY
done
cp /tmp/keep.so deepqmc_amd/csrc/libdqmc_hip.so
