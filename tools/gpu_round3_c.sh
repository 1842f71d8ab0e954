cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; rm -f gpurun_out/parity_report.json
timeout 1500 python -m pytest tests/test_gpu_parity_full.py -m gpu -q > gpurun_out/pytest_full.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_full.log
tail -30 gpurun_out/pytest_full.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/parity_report.json'))
for k,v in d.items():
    if 'refine_on' in v:
        print(k, 'first', {a:v['refine_on'][a] for a in ('frac_within_1e-5','p99','max','n_refined')}, 'second', {a:v['refine_on_second_call'][a] for a in ('frac_within_1e-5','p99','max','n_refined')}, v.get('ecp'))
PY
timeout 600 python bench.py --steps 20 --warmup 3 --min-seconds 3 --no-cpu-baseline > gpurun_out/bench_c.log 2> gpurun_out/bench_c.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_c.log').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_refine_off','eloc_only_evals_per_s')}); print(d['config']['refine_engaged']); print(d['roofline']['kernel_ms_per_step'], d['roofline']['avg_launch_us'])
PY
timeout 600 python bench.py --steps 20 --warmup 3 --min-seconds 3 --no-cpu-baseline --opt refine_ahead=0 > gpurun_out/bench_c2.log 2> gpurun_out/bench_c2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_c2.log').read().strip().splitlines()[-1])
print('ahead off', {k:d[k] for k in ('value','ms_per_step','ms_per_step_refine_off','eloc_only_evals_per_s')})
PY
