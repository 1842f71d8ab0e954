#!/bin/bash
# Round-5 GPU call A: the whole -m gpu suite (incl. the new regime / trajectory tests), per-walker calibration data of every
# fixture and of bench-like trajectories (tools/calib_data.py -> offline analysis), the headline bench line, and a handful of
# tile-shape options of the sub-step kernel.  Everything lands in gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; rm -f gpurun_out/parity_report.json gpurun_out/gpu_mem.log
nproc > gpurun_out/device.log; rocm-smi --showclocks >> gpurun_out/device.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
FIX="lih_paulinet_4096 n2_ferminet_4096 benzene_psiformer_256 c4h4_transpsiformer_512 benzene_ecp_psiformer_32 lih_psiformer_256 n2_ferminet_512 benzene_psiformer_8 c4h4_transpsiformer_64 lih_paulinet_raw_1024"
timeout 900 python tools/calib_data.py $FIX traj:LiH:paulinet:4096:30 traj:N2:ferminet:4096:10 traj:benzene:psiformer:256:10 > gpurun_out/calib_data.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
: > gpurun_out/ab_substep.jsonl
for o in "" "--fused-wt 8 --fused-occ 2" "--fused-wt 8" "--fused-occ 3" "--fused-wt 2"; do
  echo "# $o" >> gpurun_out/ab_substep.jsonl
  timeout 200 python bench.py --steps 20 --warmup 3 --min-seconds 2 --no-cpu-baseline --refine 0 $o 2>/dev/null | cut -c1-400 >> gpurun_out/ab_substep.jsonl
done
timeout 400 python bench.py --molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 --steps 2 --warmup 1 --min-seconds 3 --no-cpu-baseline > gpurun_out/bench_benzene256.log 2>&1
tail -5 gpurun_out/pytest_gpu.log; tail -14 gpurun_out/calib_data.log; grep '^{' gpurun_out/bench.log | cut -c1-300; cat gpurun_out/ab_substep.jsonl | cut -c1-200; grep '^{' gpurun_out/bench_benzene256.log | cut -c1-300
