#!/bin/bash
mkdir -p gpurun_out/r06c
python tools/spec_profile.py > gpurun_out/r06c/spec_profile.txt 2>&1
python tools/spec_profile.py --walkers 16 > gpurun_out/r06c/spec_profile_16.txt 2>&1
