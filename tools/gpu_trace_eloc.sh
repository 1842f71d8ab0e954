#!/bin/bash
# Kernel timeline of a few E_loc-only calls: gpurun_out/trace_eloc.csv (start / end / stream per kernel)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf "$ROOT/gpurun_out/trace_eloc"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$ROOT/gpurun_out/trace_eloc" -o t -- python "$ROOT/tools/eloc_only.py" ${1:-1} > "$ROOT/gpurun_out/trace_eloc.log" 2>&1
f=$(find "$ROOT/gpurun_out/trace_eloc" -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# keep the last E_loc pass: find the last k_feat_en<float> start
idx=[i for i,r in enumerate(rows) if 'k_feat_en<float' in r['Kernel_Name']]
i0=idx[-1]
t0=int(rows[i0]['Start_Timestamp'])
out=open('/root/repo/gpurun_out/trace_eloc.txt','w')
for r in rows[i0:]:
    s=(int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
    line='%8.1f %8.1f %6.1f q%s %s'%(s,e,e-s,r.get('Queue_Id','?'),r['Kernel_Name'][:80])
    out.write(line+'\n')
print(open('/root/repo/gpurun_out/trace_eloc.txt').read()[:6000])
PY
rm -rf "$ROOT/gpurun_out/trace_eloc"
