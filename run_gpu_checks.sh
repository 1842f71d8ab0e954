#!/bin/bash
# One gpurun call: smoke + GPU parity tests + bench sweeps + rocprofv3 kernel stats (-> gpurun_out/)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.json
nproc > gpurun_out/device.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
for cfg in "--fused 0" "--fused-wt 4" "--fused-wt 8" "--fused-wt 16" "--fused-wt 32"; do
  echo "== $cfg" >> gpurun_out/sweep.log
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $cfg >> gpurun_out/sweep.log 2>&1
done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof" -o r01 -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$ROOT/gpurun_out/prof.log" 2>&1
echo "prof rc=$?" >> "$ROOT/gpurun_out/prof.log"
cd "$ROOT"
tail -3 gpurun_out/smoke.log; tail -15 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/bench.log
