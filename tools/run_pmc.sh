#!/bin/bash
# SQ counters of one VMC step per kernel (three passes of <= 8 SQ counters) -> gpurun_out/${PMC_OUT:-pmc_sq}.json
# (PMC_FULLNAME=1: one entry per kernel instantiation instead of per kernel family)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P3="SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"
k=1
for P in "$P1" "$P2" "$P3"; do
  rm -rf "$ROOT/gpurun_out/pmc$k"
  timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$ROOT/gpurun_out/pmc$k" -o p$k -- python "$ROOT/bench.py" --steps 1 --warmup 1 --repeats 1 --no-cpu-baseline "$@" > "$ROOT/gpurun_out/pmc$k.log" 2>&1
  k=$((k+1))
done
cd "$ROOT"
python tools/pmc_sq.py gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3 > gpurun_out/${PMC_OUT:-pmc_sq}.json
rm -rf gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3      # (the raw counter tables are tens of MB: gpurun merges at most 64 MiB back)
python - <<'PY'
import json
import os
d = json.load(open('gpurun_out/%s.json' % os.environ.get('PMC_OUT', 'pmc_sq')))
for k in d:
    if k.startswith(('k_fused2_value', 'k_substep', 'k_linear', 'k_attention')): print(k, json.dumps({a: (round(b, 3) if isinstance(b, float) else b) for a, b in d[k].items() if a.startswith(('frac_', 'valu_per', 'mfma_busy', 'launches'))}))
PY
