cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
python tools/calib_refine.py > gpurun_out/calib.log 2>&1; echo "calib rc=$?" >> gpurun_out/calib.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ref_ansatz_hip.py -m gpu -x -q > gpurun_out/pytest_quick.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_quick.log
tools/prof_cfg.sh n2 --molecule N2 --ansatz ferminet --n-sub 10 --steps 2 --warmup 1 --repeats 1 > gpurun_out/prof_n2.txt 2>&1
tools/prof_cfg.sh benzene --molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 --steps 1 --warmup 1 --repeats 1 > gpurun_out/prof_benzene.txt 2>&1
tools/prof_cfg.sh c4h4 --molecule cyclobutadiene_square --ansatz transpsiformer --walkers 512 --steps 1 --warmup 1 --repeats 1 > gpurun_out/prof_c4h4.txt 2>&1
tail -8 gpurun_out/calib.log; tail -3 gpurun_out/pytest_quick.log
