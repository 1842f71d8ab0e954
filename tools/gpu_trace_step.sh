#!/bin/bash
# Kernel + copy timeline of ONE whole VMC step of the headline bench (from a k_rng launch to the next): gpurun_out/trace_step.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf "$ROOT/gpurun_out/trace_step"
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$ROOT/gpurun_out/trace_step" -o t -- python "$ROOT/bench.py" --steps 4 --warmup 4 --repeats 1 --no-cpu-baseline > "$ROOT/gpurun_out/trace_step.log" 2>&1
python - "$ROOT/gpurun_out/trace_step" <<'PY'
import csv, sys, glob
d = sys.argv[1]
k = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), 'q' + r.get('Queue_Id', '?'), r['Kernel_Name'][:90]) for r in csv.DictReader(open(k))]
for m in glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(m)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'cp', 'COPY ' + r.get('Direction', '') + ' ' + r.get('Bytes', r.get('Size', ''))))
rows.sort()
idx = [i for i, r in enumerate(rows) if 'k_rng' in r[3]]
i0, i1 = idx[-3], idx[-2]
t0 = rows[i0][0]
out = open('/root/repo/gpurun_out/trace_step.txt', 'w')
prev_e = t0
for s, e, q, n in rows[i0:i1 + 1]:
    if 'k_substep' in n and prev_e and (s - prev_e) < 3000: prev_e = max(prev_e, e); continue     # back-to-back sub-steps: not listed
    out.write('%8.1f %8.1f %6.1f gap %6.1f %s %s\n' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, (s - prev_e) / 1e3, q, n))
    prev_e = max(prev_e, e)
out.close()
print(open('/root/repo/gpurun_out/trace_step.txt').read()[:9000])
PY
rm -rf "$ROOT/gpurun_out/trace_step"
