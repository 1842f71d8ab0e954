#!/bin/bash
# One gpurun call: smoke + GPU parity tests + bench (headline + the other BASELINE configs) + rocprofv3 kernel
# stats + HBM traffic counters (-> gpurun_out/; tools/collect_profiles.sh copies the summaries to profiles/)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.json gpurun_out/other_configs.log
nproc > gpurun_out/device.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tools/run_traffic.sh > gpurun_out/traffic.log 2>&1          # HBM counters first: bench.py reports them as roofline.traffic
cp gpurun_out/pmc_hbm_traffic.json profiles/r02_pmc_hbm_traffic.json
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.log
for cfg in "--refine 0 --steps 20 --warmup 3" \
           "--refine 2 --steps 10 --warmup 2" \
           "--overlap 1 --steps 20 --warmup 3" \
           "--molecule N2 --ansatz ferminet --n-sub 10 --steps 3 --warmup 1 --repeats 3" \
           "--molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 --steps 2 --warmup 1 --repeats 2" \
           "--molecule benzene --ansatz psiformer --ecp --walkers 64 --n-sub 2 --steps 1 --warmup 1 --repeats 1" \
           "--molecule cyclobutadiene_square --ansatz transpsiformer --walkers 512 --steps 2 --warmup 1 --repeats 2" \
           "--molecule cyclobutadiene_square --ansatz transpsiformer --states 3 --walkers 2048 --steps 1 --warmup 1 --repeats 1"; do
  timeout 900 python bench.py --no-cpu-baseline $cfg 2>/dev/null | grep '^{' >> gpurun_out/other_configs.log
done
cd /tmp && export TMPDIR=/tmp
rm -rf "$ROOT/gpurun_out/prof"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof" -o r02 -- python "$ROOT/bench.py" --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline > "$ROOT/gpurun_out/prof.log" 2>&1
echo "prof rc=$?" >> "$ROOT/gpurun_out/prof.log"
cd "$ROOT"
tail -3 gpurun_out/smoke.log; tail -5 gpurun_out/pytest_gpu.log; tail -1 gpurun_out/bench.log | cut -c1-600; wc -l gpurun_out/other_configs.log; tail -5 gpurun_out/traffic.log
