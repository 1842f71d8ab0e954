cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
tools/prof_cfg.sh lih --steps 5 --warmup 40 --repeats 1 > gpurun_out/prof_lih.txt 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/kernel_stats_lih.csv')))
for r in rows[:32]:
    print(r['Name'][:100].ljust(100), r['Calls'].rjust(6), ('%.2f ms'%(float(r['TotalDurationNs'])/1e6)).rjust(11), ('%.1f us'%(float(r['AverageNs'])/1e3)).rjust(11), r['Percentage'])
PY
