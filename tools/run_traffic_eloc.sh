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
# passes are counted by their FIRST kernel (k_feat_en of the precision); since round 5 the float32 pass ends in float64 kernels
# (the float64 tail: backflow head, orbitals, determinants, k_final on the twin for every walker), so the double instantiations
# belong partly to the float32 pass and partly to the refinement twin's pass -- the split by precision below is by kernel
# instantiation, the per-call total is what the call moves
is64=lambda k: '<double' in k or k.startswith(('k_refine','k_widen','k_tail'))
n32=sum(v['launches'] for k,v in K.items() if k.startswith('k_feat_en<float'))
n64=sum(v['launches'] for k,v in K.items() if k.startswith('k_feat_en<double'))
tot32=sum(v['hbm_bytes']*v['launches'] for k,v in K.items() if k.startswith('k_') and not k.startswith(skip) and not is64(k))
tot64=sum(v['hbm_bytes']*v['launches'] for k,v in K.items() if k.startswith('k_') and not k.startswith(skip) and is64(k))
d['eloc_calls']=n32; d['float64_twin_passes']=n64
d['hbm_bytes_per_call_float32_kernels']=tot32/max(n32,1)
d['hbm_bytes_per_call_float64_kernels']=tot64/max(n32,1)
d['hbm_bytes_per_eloc_call']=(tot32+tot64)/max(n32,1)
for k in ('float32_passes','hbm_bytes_per_float32_pass','hbm_bytes_per_twin_pass'): d.pop(k, None)
d['note_eloc']='sums over every kernel of a local-energy call on 4096 walkers, divided by the number of calls (= k_feat_en<float> launches): float32 instantiations (the head of the forward-Laplacian pass) and float64 instantiations (its float64 tail for every walker + the refinement twin over the flagged walkers + gather / scatter / widen / narrow)'
json.dump(d,open(p,'w'),indent=1)
print('HBM bytes per E_loc call: %.3f GB = %.3f (float32 kernels) + %.3f (float64 kernels: tail + twin); %d calls, %d twin passes' % ((tot32+tot64)/max(n32,1)/1e9, tot32/max(n32,1)/1e9, tot64/max(n32,1)/1e9, n32, n64))
for k,v in sorted(K.items(), key=lambda kv:-kv[1]['hbm_bytes']*kv[1]['launches'])[:14]: print('  %-50s n=%4d %8.1f MB/launch  %5.1f us' % (k[:50], v['launches'], v['hbm_bytes']/1e6, v['avg_us'] or 0))
PY
rm -rf gpurun_out/pmce_FETCH_SIZE gpurun_out/pmce_WRITE_SIZE
