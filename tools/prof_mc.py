import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from deepqmc_amd import MolecularHamiltonian, Molecule
from deepqmc_amd.sampling import DecorrSampler
from deepqmc_amd.wf import NeuralNetworkWaveFunction
h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
wf = NeuralNetworkWaveFunction(h, 'paulinet', dtype=torch.float32, device='cuda:0')
params = wf.init(0, perturb_envelopes=0.05)
eng = wf.engine(params)
eng.set_option('fused_dbg', 1)
s = DecorrSampler(h, wf, length=5)
st = s.init(1, params, 4096)
for k in range(3): st, pc, _ = s.sample(k, st, params)
torch.cuda.synchronize()
out = np.zeros(1024 + 16)
eng._check(eng.lib.dqmc_debug_read(eng._ctx, -3, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), out.size))
t = out[1000:1009]
print('prologue', t[7]-t[8], 'main', t[6]-t[7], 'mats', t[1]-t[0], 'lu', t[2]-t[1], 'ci', t[3]-t[2], 'walker', t[4]-t[3], 'atomics', t[5]-t[4], 'total', t[5]-t[8])
