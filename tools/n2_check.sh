cd "${GRAFT_REPO_ROOT:-/root/repo}"
for o in linear_bkx=3 linear_bkx=4; do
python bench.py --molecule N2 --ansatz ferminet --n-sub 10 --steps 3 --warmup 2 --repeats 3 --no-cpu-baseline --opt $o 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d["config"]["refine_engaged"]; print("N2 '$o' ms_per_step", d["ms_per_step"], r["error_per_score"], r["fraction_refined"])'
done
