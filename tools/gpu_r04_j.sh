#!/bin/bash
# SQ counters per kernel instantiation for benzene / Psiformer (256 walkers): what bounds k_slogdet_mfma, the attention and the float64 tiles
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
PMC_FULLNAME=1 bash tools/run_pmc.sh --molecule benzene --ansatz psiformer --walkers 256 --n-sub 10 --equilibrate 50 > gpurun_out/pmc_benzene.log 2>&1
cp gpurun_out/pmc_sq.json gpurun_out/pmc_sq_benzene.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_sq_benzene.json'))
for k,v in d.items():
    if any(s in k for s in ('slogdet_mfma','attention_mfma','k_linear<double, 4','k_linear<float, 8','k_orbitals')):
        keys=['launches','SQ_WAVES','SQ_BUSY_CYCLES','SQ_WAVE_CYCLES','SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS','SQ_LDS_BANK_CONFLICT','SQ_INSTS_MFMA','SQ_VALU_MFMA_BUSY_CYCLES','SQ_INSTS_VMEM_RD','frac_WAIT_ANY','frac_WAIT_INST_ANY','frac_ACTIVE_INST_ANY','frac_WAIT_INST_LDS','frac_ACTIVE_INST_LDS','frac_ACTIVE_INST_VALU','SQ_LDS_IDX_ACTIVE']
        print(k[:60], {x:(round(v[x],3) if v.get(x,0)<10 else int(v[x])) for x in keys if x in v})
PY
