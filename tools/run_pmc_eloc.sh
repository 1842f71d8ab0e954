#!/bin/bash
# SQ counters of the E_loc pass per kernel instantiation (serial streams, refinement off) -> gpurun_out/pmc_sq_eloc.json
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P3="SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"
k=1
for P in "$P1" "$P2" "$P3"; do
  rm -rf "$ROOT/gpurun_out/pmce$k"
  DQMC_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$ROOT/gpurun_out/pmce$k" -o p$k -- python "$ROOT/tools/eloc_only.py" 0 > "$ROOT/gpurun_out/pmce$k.log" 2>&1
  k=$((k+1))
done
cd "$ROOT"
PMC_FULLNAME=1 python tools/pmc_sq.py gpurun_out/pmce1 gpurun_out/pmce2 gpurun_out/pmce3 > gpurun_out/pmc_sq_eloc.json
rm -rf gpurun_out/pmce[123]
python - <<'PY'
import json
d = json.load(open('gpurun_out/pmc_sq_eloc.json'))
for k,v in sorted(d.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CYCLES',0)*kv[1].get('launches',0)):
    if not k.startswith('k_'): continue
    g=lambda c: v.get(c,0)
    print('%-44s n=%4d busy=%9.0f mfma_busy/busy=%.2f  wait_any=%.2f wait_inst=%.2f active=%.2f  valu/mfma=%.1f lds_conf/lds_active=%.2f  vmem_rd/wave=%.0f' % (k[:44], g('launches'), g('SQ_BUSY_CYCLES'), g('mfma_busy_over_sq_busy'), g('frac_WAIT_ANY'), g('frac_WAIT_INST_ANY'), g('frac_ACTIVE_INST_ANY'), g('valu_per_mfma'), g('SQ_LDS_BANK_CONFLICT')/max(g('SQ_LDS_IDX_ACTIVE'),1), g('SQ_INSTS_VMEM_RD')/max(g('SQ_WAVES'),1)))
PY
