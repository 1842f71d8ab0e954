cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; rm -f gpurun_out/parity_report.json
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/smoke.log; tail -12 gpurun_out/pytest_gpu.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/parity_report.json'))
for k,v in d.items():
    if 'refine_on' in v:
        print(k, 'first', {a:v['refine_on'][a] for a in ('frac_within_1e-5','p99','max','n_refined')}, 'second', {a:v['refine_on_second_call'][a] for a in ('frac_within_1e-5','p99','max','n_refined')}, v.get('ecp'), v.get('logpsi_rel_err_p99'))
PY
