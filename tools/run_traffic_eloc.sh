#!/bin/bash
# HBM traffic of ONE local-energy pass (tools/eloc_only.py: 23 passes of 4096 LiH walkers): FETCH_SIZE / WRITE_SIZE in two
# counter passes -> gpurun_out/pmc_hbm_traffic_eloc.json with the sum over the kernels of a pass.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$ROOT/gpurun_out/pmce_$c"
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$ROOT/gpurun_out/pmce_$c" -o t -- python "$ROOT/tools/eloc_only.py" ${1:-1} > "$ROOT/gpurun_out/pmce_$c.log" 2>&1
done
cd "$ROOT"
PMC_FULLNAME=1 python tools/pmc_traffic.py $(find gpurun_out/pmce_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmce_WRITE_SIZE -name "*counter_collection.csv" | head -1) gpurun_out/pmc_hbm_traffic_eloc.json "LiH/paulinet/4096/f32 E_loc passes (tools/eloc_only.py, refine ${1:-1})"
python - <<'PY'
import json
p='gpurun_out/pmc_hbm_traffic_eloc.json'
d=json.load(open(p))
K=d['kernels']
skip=('k_fused2_value','k_rng','k_sampler_stats','k_tau_finalize')
is64=lambda k: '<double' in k or k.startswith('k_refine')
n32=sum(v['launches'] for k,v in K.items() if k.startswith('k_final<float'))
n64=sum(v['launches'] for k,v in K.items() if k.startswith('k_final<double'))
tot32=sum(v['hbm_bytes']*v['launches'] for k,v in K.items() if k.startswith('k_') and not k.startswith(skip) and not is64(k))
tot64=sum(v['hbm_bytes']*v['launches'] for k,v in K.items() if k.startswith('k_') and not k.startswith(skip) and is64(k))
d['float32_passes']=n32; d['float64_twin_passes']=n64
d['hbm_bytes_per_float32_pass']=tot32/max(n32,1)
d['hbm_bytes_per_twin_pass']=tot64/max(n64,1)
d['hbm_bytes_per_eloc_call']=(tot32+tot64)/max(n32,1)
d['note_eloc']='sums over every kernel of the forward-Laplacian pass of 4096 walkers (float32 instantiations) and of its float64 refinement twin (double instantiations + gather / scatter), each divided by the number of k_final launches of that precision'
json.dump(d,open(p,'w'),indent=1)
print('HBM bytes per float32 E_loc pass: %.3f GB (%d passes); twin: %.3f GB (%d); per call %.3f GB' % (tot32/max(n32,1)/1e9, n32, tot64/max(n64,1)/1e9, n64, (tot32+tot64)/max(n32,1)/1e9))
for k,v in sorted(K.items(), key=lambda kv:-kv[1]['hbm_bytes']*kv[1]['launches'])[:14]: print('  %-50s n=%4d %8.1f MB/launch  %5.1f us' % (k[:50], v['launches'], v['hbm_bytes']/1e6, v['avg_us'] or 0))
PY
rm -rf gpurun_out/pmce_FETCH_SIZE gpurun_out/pmce_WRITE_SIZE
