#!/bin/bash
# time the generated sub-step kernel variants named in $VARIANTS (DQMC_CODEGEN_VARIANTS -> python -m deepqmc_amd.codegen -> make), one spec_check.py run each
mkdir -p gpurun_out/spec_variants
rm -f gpurun_out/spec_variants/variants.txt
for v in $VARIANTS; do
  echo "== $v" >> gpurun_out/spec_variants/variants.txt
  DQMC_SPEC_VARIANT=$v python tools/spec_check.py 2>&1 | grep -E "spec 1: |accept bits|maxdiff" >> gpurun_out/spec_variants/variants.txt
done
