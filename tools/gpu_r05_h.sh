#!/bin/bash
# Round-5 GPU call H: does the float32 error scale m depend on HOW k_linear accumulates?  A fresh accumulator per 16- / 32-wide
# k chunk (kernel_linear.hip: FRESH; libraries built by hand with -DDQMC_FRESH_MR_MAX=2 (the tree) / 3 and from the previous commit): m of the calibration probe,
# refined share and step time on the headline configuration; same call.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
O=gpurun_out/ab_fresh.txt; : > $O
probe() {
timeout 300 python - >> $O 2>&1 <<'P'
import numpy as np, torch
from deepqmc_amd import MolecularHamiltonian, Molecule
from deepqmc_amd.engine import Engine
from deepqmc_amd.params import init_params
from deepqmc_amd.spec import paulinet, ferminet
from deepqmc_amd.sampling import synthetic_walkers
eps = float(np.finfo(np.float32).eps)
for molname, spec_fn, n_sub in (('LiH', paulinet, 30), ('N2', ferminet, 10)):
    h = MolecularHamiltonian(mol=Molecule.from_name(molname)); spec = spec_fn()
    params = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=5, perturb_envelopes=0.1)
    B = 4096
    eng = Engine(spec, h, params, dtype=torch.float32, device='cuda', norm_eps=eps)
    r0 = torch.as_tensor(synthetic_walkers(h, B, seed=3).astype(np.float32), device='cuda')
    sign0, log0 = eng.wf_eval(r0)
    st = {'r': r0.clone(), 'log': log0.clone(), 'sign': sign0.clone(), 'age': torch.zeros(B, dtype=torch.int32, device='cuda'),
          'tau': torch.full((1,), 0.3, dtype=torch.float32, device='cuda')}
    for k in range(20):
        eng.mcmc_steps(st, n_sub, seed=100 + k)
    ms, fr = [], []
    eng.set_option('refine_probe', 1)            # every call probes: six estimates of m (smoothed geometrically by the library)
    for k in range(6):
        eng.mcmc_steps(st, n_sub, seed=200 + k)
        e, _ = eng.local_energy(st['r'])
        info = eng.refine_info(); ms.append(info['error_per_score']); fr.append(eng.last_refined() / B)
    eng.set_option('refine', 2); e64, _ = eng.local_energy(st['r']); eng.set_option('refine', 0); e32, _ = eng.local_energy(st['r'])
    rel = ((e32.double() - e64.double()).abs() / e64.double().abs().clamp(min=1.0)).cpu().numpy()
    print(molname, 'm', ['%.3e' % x for x in ms], 'refined', ['%.3f' % x for x in fr],
          'plain f32 error p50 %.3e p90 %.3e p99 %.3e' % tuple(np.quantile(rel, [0.5, 0.9, 0.99])), flush=True)
P
}
run() { echo "# $1" >> $O; shift; timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   ms/step %.3f  value %.0f  eloc_only %.0f  refined %s' % (d['ms_per_step'], d['value'], d['eloc_only_evals_per_s'] or 0, d['config']['refine_engaged'].get('fraction_refined')))" >> $O; }
suite() {
  echo "== $1" >> $O; probe
  run "$1 N2" --molecule N2 --ansatz ferminet --n-sub 10 --steps 5 --warmup 2 --min-seconds 3
  run "$1 LiH" --steps 20 --warmup 5 --min-seconds 3
}
cp deepqmc_amd/csrc/libdqmc_hip.so /tmp/new.so
suite "tree: fresh accumulators in k_linear (<= 3 row blocks) and in k_linear_bf"
cp deepqmc_amd/csrc/libdqmc_hip_nobf.so deepqmc_amd/csrc/libdqmc_hip.so
suite "fresh accumulators in k_linear only"
cp deepqmc_amd/csrc/libdqmc_hip_old.so deepqmc_amd/csrc/libdqmc_hip.so
suite "previous commit"
cp /tmp/new.so deepqmc_amd/csrc/libdqmc_hip.so
cat $O
