cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
tools/gpu_ab_opt.sh "--steps 20 --warmup 10 --min-seconds 3" multi_stream=1 multi_stream=0 multi_stream=1 multi_stream=0
tools/gpu_trace_eloc.sh 1 > /dev/null 2>&1; tail -3 gpurun_out/trace_eloc.log | head -1; cut -c1-130 gpurun_out/trace_eloc.txt | head -60
