cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; rm -f gpurun_out/parity_report.json
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_gpu.log
tail -6 gpurun_out/pytest_gpu.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/parity_report.json'))
for k,v in d.items():
    if 'refine_on' in v:
        print(k, {a:v['refine_on'][a] for a in ('frac_within_1e-5','p99','max','n_refined')})
PY
for o in "mlp_fuse=1" "mlp_fuse=0"; do
timeout 600 python bench.py --steps 20 --warmup 40 --min-seconds 4 --no-cpu-baseline --opt $o > gpurun_out/bench_f.log 2> gpurun_out/bench_f.err
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_f.log').read().strip().splitlines()[-1])
print('$o', {k:d[k] for k in ('value','ms_per_step','ms_per_step_refine_off','eloc_only_evals_per_s')}); print(d['config']['refine_engaged']['fraction_refined'], d['roofline']['kernel_ms_per_step'])
PY
done
