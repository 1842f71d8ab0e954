#!/bin/bash
# 64-wide K chunks for the float64 twin's small tiles (twin.linear_bkx=5) vs the default (32-wide)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for o in "twin.linear_bkx=3" "twin.linear_bkx=5" "twin.linear_bkx=3" "twin.linear_bkx=5"; do
  echo -n "$o  "; DQMC_OPTS=$o timeout 300 python tools/eloc_only.py 1 2>&1 | grep "ms per"
done
python - <<'PY'
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from deepqmc_amd import MolecularHamiltonian, Molecule
from deepqmc_amd.sampling import synthetic_walkers
from deepqmc_amd.wf import NeuralNetworkWaveFunction
h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
wf = NeuralNetworkWaveFunction(h, 'paulinet', dtype=torch.float32, device='cuda:0')
params = wf.init(0, perturb_envelopes=0.05)
eng = wf.engine(params)
r = torch.as_tensor(synthetic_walkers(h, 1024, seed=3).astype(np.float32), device='cuda:0')
eng.set_option('refine', 2)
e0 = eng.local_energy(r)[0].clone()
eng.set_option('twin.linear_bkx', 5)
e1 = eng.local_energy(r)[0]
print('refine=2 energies: bkx 3 vs 5 max abs diff', float((e0 - e1).abs().max()))
PY
