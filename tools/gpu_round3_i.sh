cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; rm -f gpurun_out/parity_report.json
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py tests/test_gpu_samplers.py -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_gpu.log
tail -6 gpurun_out/pytest_gpu.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/parity_report.json'))
for k,v in d.items():
    if 'refine_on' in v:
        print(k, {a:v['refine_on'][a] for a in ('frac_within_1e-5','p99','max','n_refined')}, v['refine_on']['refine_info']['score_threshold'])
PY
tools/gpu_ab_opt.sh "--molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 --steps 2 --warmup 1 --repeats 2" refine=1
tools/gpu_ab_opt.sh "--molecule cyclobutadiene_square --ansatz transpsiformer --walkers 512 --steps 2 --warmup 1 --repeats 2" refine=1
