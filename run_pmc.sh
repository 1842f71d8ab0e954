#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$ROOT/gpurun_out/counters.txt" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d "$ROOT/gpurun_out/pmc1" -o p1 -- python "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --fused-wt 4 > "$ROOT/gpurun_out/pmc1.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d "$ROOT/gpurun_out/pmc2" -o p2 -- python "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --fused-wt 4 > "$ROOT/gpurun_out/pmc2.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d "$ROOT/gpurun_out/pmc3" -o p3 -- python "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --fused-wt 4 > "$ROOT/gpurun_out/pmc3.log" 2>&1
ls -la "$ROOT"/gpurun_out/pmc*/ | head -30
tail -3 "$ROOT"/gpurun_out/pmc1.log "$ROOT"/gpurun_out/pmc3.log
