#!/bin/bash
# Round-5 GPU call D: the float64 tail -- error model with / without it along the bench trajectories, cost on the headline and N2
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; rm -f gpurun_out/parity_report.json
timeout 600 python tools/calib_data.py traj:LiH:paulinet:4096:30 traj:N2:ferminet:4096:10 lih_paulinet_4096 n2_ferminet_4096 > gpurun_out/calib_data_tail.log 2>&1
python tools/calib_sim.py gpurun_out/calib_traj_LiH_paulinet_4096.npz gpurun_out/calib_traj_N2_ferminet_4096.npz gpurun_out/calib_lih_paulinet_4096.npz gpurun_out/calib_n2_ferminet_4096.npz > gpurun_out/calib_model_tail.txt 2>&1
: > gpurun_out/ab_tail.jsonl
for o in "" "--opt tail_f64=0"; do
  echo "# LiH $o" >> gpurun_out/ab_tail.jsonl
  timeout 200 python bench.py --steps 20 --warmup 5 --min-seconds 3 --no-cpu-baseline $o 2>/dev/null >> gpurun_out/ab_tail.jsonl
  echo "# N2 $o" >> gpurun_out/ab_tail.jsonl
  timeout 300 python bench.py --molecule N2 --ansatz ferminet --n-sub 10 --steps 5 --warmup 2 --min-seconds 3 --no-cpu-baseline $o 2>/dev/null >> gpurun_out/ab_tail.jsonl
done
timeout 1500 python -m pytest tests/test_gpu_parity_full.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/pytest_gpu_d.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_d.log
tail -6 gpurun_out/pytest_gpu_d.log; cat gpurun_out/calib_data_tail.log | tail -5; grep -E "^   (traj|lih|n2)|m from" gpurun_out/calib_model_tail.txt | cut -c1-220 | head; grep -E "round|traj|lih|n2" gpurun_out/calib_model_tail.txt | tail -24 | cut -c1-200
python - <<'P'
import json
for l in open('gpurun_out/ab_tail.jsonl'):
    if l.startswith('#'): print(l.strip()); continue
    if not l.startswith('{'): continue
    d=json.loads(l); c=d['config']['refine_engaged']
    print('   ms/step %.3f  value %.0f  eloc_only %.0f  refined frac %s  thr %.1f  m %.2e refine_off %s' % (d['ms_per_step'], d['value'], d['eloc_only_evals_per_s'] or 0, c.get('fraction_refined'), c.get('score_threshold', 0), c.get('error_per_score', 0), d.get('ms_per_step_refine_off')))
P
