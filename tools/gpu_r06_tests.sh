#!/bin/bash
mkdir -p gpurun_out/r06t
python -m pytest tests -m gpu -x -q > gpurun_out/r06t/pytest_gpu.txt 2>&1
tail -5 gpurun_out/r06t/pytest_gpu.txt
