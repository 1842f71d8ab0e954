cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
O=gpurun_out/ab_tail2.txt; : > $O
run() { echo "# $1" >> $O; shift; timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   ms/step %.3f  value %.0f  eloc_only %.0f  refined %s  m %s' % (d['ms_per_step'], d['value'], d['eloc_only_evals_per_s'] or 0, d['config']['refine_engaged'].get('fraction_refined'), d['config']['refine_engaged'].get('error_per_score')))" >> $O; }
run "LiH default" --steps 20 --warmup 5 --min-seconds 3
run "LiH tail_f64=0" --steps 20 --warmup 5 --min-seconds 3 --opt tail_f64=0
run "N2 default" --molecule N2 --ansatz ferminet --n-sub 10 --steps 5 --warmup 2 --min-seconds 3
run "N2 tail_f64=0" --molecule N2 --ansatz ferminet --n-sub 10 --steps 5 --warmup 2 --min-seconds 3 --opt tail_f64=0
run "LiH default again" --steps 20 --warmup 5 --min-seconds 3
cat $O
