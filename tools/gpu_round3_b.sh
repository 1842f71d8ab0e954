cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; rm -f gpurun_out/parity_report.json
timeout 1200 python -m pytest tests/test_gpu_parity_full.py -m gpu -q > gpurun_out/pytest_full.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_full.log
tail -40 gpurun_out/pytest_full.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/parity_report.json'))
for k,v in d.items():
    if 'refine_on' in v: print(k, v['refine_on'])
PY
