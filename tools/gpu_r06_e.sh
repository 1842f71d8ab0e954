#!/bin/bash
# rocprofv3 kernel durations of the sub-step kernel variants (spec_check.py runs), + the 16-walker profile
ROOT=$(pwd); mkdir -p gpurun_out/r06e; cd /tmp && export TMPDIR=/tmp
for v in $VARIANTS; do
  rm -rf /tmp/prof_$v
  DQMC_SPEC_VARIANT=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o $v -- python $ROOT/tools/spec_check.py --reps 10 > $ROOT/gpurun_out/r06e/log_$v.txt 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v" >> $ROOT/gpurun_out/r06e/stats.txt
  head -6 "$f" | cut -c1-200 >> $ROOT/gpurun_out/r06e/stats.txt
done
