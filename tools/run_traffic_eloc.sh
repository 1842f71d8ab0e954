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
python tools/pmc_traffic.py $(find gpurun_out/pmce_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmce_WRITE_SIZE -name "*counter_collection.csv" | head -1) gpurun_out/pmc_hbm_traffic_eloc.json "LiH/paulinet/4096/f32 E_loc passes (tools/eloc_only.py, refine ${1:-1})"
python - <<'PY'
import json
p='gpurun_out/pmc_hbm_traffic_eloc.json'
d=json.load(open(p))
n_pass=d['kernels'].get('k_final',{}).get('launches',0) or 1
tot=sum(v['hbm_bytes']*v['launches'] for k,v in d['kernels'].items() if k.startswith('k_') and k not in ('k_fused2_value','k_rng','k_sampler_stats','k_tau_finalize'))
d['eloc_passes']=n_pass
d['hbm_bytes_per_eloc_pass']=tot/n_pass
d['note_eloc']='sum over every kernel of the forward-Laplacian pass (float32 pass + float64 refinement twin: k_final launches count both), divided by the number of k_final launches of the float32 build counted as passes'
json.dump(d,open(p,'w'),indent=1)
print('HBM bytes per E_loc pass: %.3f GB over %d k_final launches' % (tot/n_pass/1e9, n_pass))
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['hbm_bytes']*kv[1]['launches'])[:10]: print(' ', k, v['launches'], '%.1f MB'%(v['hbm_bytes']/1e6))
PY
rm -rf gpurun_out/pmce_FETCH_SIZE gpurun_out/pmce_WRITE_SIZE
