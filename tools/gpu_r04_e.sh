#!/bin/bash
# Does the deferred float64 pass overlap the next float32 pass?  Step times for a few option sets, then a kernel trace.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); mkdir -p gpurun_out
for o in "refine_defer=0" "refine_defer=1" "refine_defer=1,twin.pass_graph=0" "refine_defer=1,pass_graph=0" "refine_defer=1,twin.multi_stream=0,twin.dual_stream=0" "refine=0"; do
  DQMC_OPTS=$o timeout 300 python tools/trace_defer.py 2>&1 | grep "ms per step"
done
cd /tmp && export TMPDIR=/tmp; rm -rf "$ROOT/gpurun_out/trace_defer"
DQMC_OPTS=refine_defer=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$ROOT/gpurun_out/trace_defer" -o t -- python "$ROOT/tools/trace_defer.py" > "$ROOT/gpurun_out/trace_defer.log" 2>&1
f=$(find "$ROOT/gpurun_out/trace_defer" -name "*kernel_trace.csv" | head -1)
python - "$f" "$ROOT" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_feat_en<float' in r['Kernel_Name']]
i0=idx[-3]; i1=idx[-2]
# back up to the start of the sub-steps before pass idx[-3]
j=i0
while j>0 and 'k_fused2' in rows[j-1]['Kernel_Name'] or 'k_rng' in rows[j-1]['Kernel_Name']: j-=1
t0=int(rows[j]['Start_Timestamp'])
out=open(sys.argv[2]+'/gpurun_out/trace_defer.txt','w')
n_sub=0
for r in rows[j:i1+45]:
    s=(int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
    nm=r['Kernel_Name']
    if 'k_fused2' in nm:
        n_sub+=1
        if n_sub>2 and n_sub<29: continue
    out.write('%8.1f %8.1f %6.1f q%s %s\n'%(s,e,e-s,r.get('Queue_Id','?'),nm[:70]))
out.close()
print(open(sys.argv[2]+'/gpurun_out/trace_defer.txt').read()[:9000])
PY
rm -rf "$ROOT/gpurun_out/trace_defer"
