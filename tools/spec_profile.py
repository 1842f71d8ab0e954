"""Shader-clock profile of the plan-specialised sub-step kernel (PROF instance, workgroup 0): cycles between the op
boundaries of the generated source, per wave.  Usage: python tools/spec_profile.py [--walkers 4096]"""
import argparse, ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepqmc_amd import MolecularHamiltonian, Molecule
from deepqmc_amd.sampling import synthetic_walkers
from deepqmc_amd.wf import NeuralNetworkWaveFunction

ap = argparse.ArgumentParser(); ap.add_argument('--walkers', type=int, default=4096)
args = ap.parse_args()
h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
wf = NeuralNetworkWaveFunction(h, 'paulinet', dtype=torch.float32, device='cuda:0')
params = wf.init(0, perturb_envelopes=0.05)
eng = wf.engine(params)
eng.set_option('fused_dbg', 1)
B = args.walkers
r = torch.as_tensor(synthetic_walkers(h, B).astype(np.float32), device='cuda:0')
sg, lg = eng.wf_eval(r)
st = {'r': r.clone(), 'log': lg, 'sign': sg, 'age': torch.zeros(B, dtype=torch.int32, device='cuda:0'),
      'tau': torch.full((1,), 0.3, dtype=torch.float32, device='cuda:0')}
eng.mcmc_steps(st, 3, seed=1)
torch.cuda.synchronize()
out = np.zeros(1024)
eng._check(eng.lib.dqmc_debug_read(eng._ctx, -3, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), out.size))
stp = out.reshape(4, 256)
ops = eng.program.ops
from deepqmc_amd.codegen.substep import Gen
g_ = Gen('x', eng.program.n_up, eng.program.n_down, eng.program.n_nuc, eng.program.spec.n_determinants, eng.program.bufs, eng.program.ops, eng.program.itable); g_.source()
labels = ['prologue'] + g_.stamp_labels
t0 = stp[:, 0].min()
for w in range(4):
    row = stp[w]; n = int((row > 0).sum())
    print('wave', w, 'start', int(row[0] - t0), 'total', int(row[n - 1] - row[0]))
row = stp[0]; n = int((row > 0).sum()); d = np.diff(row[:n])
for k in range(n - 1):
    print(f'{int(d[k]):8d}  {labels[k] if k < len(labels) else "?"}')
