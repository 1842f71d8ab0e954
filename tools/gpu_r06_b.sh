#!/bin/bash
mkdir -p gpurun_out/r06b
python tools/spec_check.py > gpurun_out/r06b/spec_check.txt 2>&1
python tools/spec_check.py --walkers 1000 --reps 5 > gpurun_out/r06b/spec_check_1000.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r06b/bench.json 2> gpurun_out/r06b/bench.err
