cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
tools/prof_cfg.sh benzene --molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 --steps 1 --warmup 1 --repeats 1 > gpurun_out/prof_benzene.txt 2>&1
tools/prof_cfg.sh c4h4 --molecule cyclobutadiene_square --ansatz transpsiformer --walkers 512 --steps 1 --warmup 1 --repeats 1 > gpurun_out/prof_c4h4.txt 2>&1
tools/prof_cfg.sh n2 --molecule N2 --ansatz ferminet --n-sub 10 --steps 2 --warmup 12 --repeats 1 > gpurun_out/prof_n2.txt 2>&1
for t in benzene c4h4 n2; do echo "== $t"; python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/kernel_stats_$t.csv')))
print('total ms', sum(float(r['TotalDurationNs']) for r in rows)/1e6)
for r in rows[:12]:
    print(r['Name'][:84].ljust(84), r['Calls'].rjust(6), ('%.1f ms'%(float(r['TotalDurationNs'])/1e6)).rjust(10), ('%.1f us'%(float(r['AverageNs'])/1e3)).rjust(11), r['Percentage'])
PY
tail -2 gpurun_out/prof_$t.log | head -1 | cut -c1-400
done
