#!/bin/bash
# round 6, call a: streaming-skeleton microbenchmark + clock profile of the current sub-step kernel + a baseline bench line
mkdir -p gpurun_out/r06a
./tools/ubench/_gen/skel > gpurun_out/r06a/skel.txt 2>&1
python tools/fused_profile.py --substep 1 --quiet 1 > gpurun_out/r06a/profile_4096.txt 2>&1
python tools/fused_profile.py --substep 1 --quiet 1 --walkers 1024 > gpurun_out/r06a/profile_1024.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r06a/bench.json 2> gpurun_out/r06a/bench.err
