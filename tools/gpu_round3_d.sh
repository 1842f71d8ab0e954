cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; rm -f gpurun_out/parity_report.json
timeout 1500 python -m pytest tests/test_gpu_parity_full.py -m gpu -q -x > gpurun_out/pytest_full.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_full.log
tail -5 gpurun_out/pytest_full.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/parity_report.json'))
for k,v in d.items():
    if 'refine_on' in v:
        print(k, 'first', {a:v['refine_on'][a] for a in ('frac_within_1e-5','p99','max','n_refined')}, 'second', {a:v['refine_on_second_call'][a] for a in ('frac_within_1e-5','p99','max','n_refined')}, v.get('ecp'))
PY
for o in "refine_ahead=1" "refine_ahead=0"; do
timeout 600 python bench.py --steps 20 --warmup 60 --min-seconds 4 --no-cpu-baseline --opt $o > gpurun_out/bench_d.log 2> gpurun_out/bench_d.err
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_d.log').read().strip().splitlines()[-1])
print('$o', {k:d[k] for k in ('value','ms_per_step','ms_per_step_refine_off','eloc_only_evals_per_s')}); print(d['config']['refine_engaged'])
PY
done
timeout 900 python bench.py --molecule benzene --ansatz psiformer --ecp --walkers 64 --n-sub 2 --steps 1 --warmup 1 --repeats 2 --no-cpu-baseline > gpurun_out/bench_ecp.log 2>&1; tail -1 gpurun_out/bench_ecp.log | cut -c1-700
