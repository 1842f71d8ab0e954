#!/bin/bash
mkdir -p gpurun_out/r06d
rm -f gpurun_out/r06d/variants.txt
for v in $VARIANTS; do
  echo "== $v" >> gpurun_out/r06d/variants.txt
  DQMC_SPEC_VARIANT=$v python tools/spec_check.py 2>&1 | grep -E "spec 1: |accept bits|maxdiff" >> gpurun_out/r06d/variants.txt
done
