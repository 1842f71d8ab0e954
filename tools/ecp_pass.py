"""Local-energy passes of benzene + (synthetic-coefficient) ECP / Psiformer: B walkers (argv[1], default 256), library options from
DQMC_OPTS=name=value,...  Prints ms per pass, the quadrature class counts and the walkers refined (for rocprofv3 --kernel-trace)."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepqmc_amd import MolecularHamiltonian, Molecule
from deepqmc_amd.ecp import ELEMENTS
from deepqmc_amd.sampling import DecorrSampler
from deepqmc_amd.wf import NeuralNetworkWaveFunction
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
mol = Molecule.from_name('benzene')
tab = lambda z: [2 if z > 2 else 0, [[-1, [[], [[5.4, float(z - 2)]], [[4.6, -4.6]], [[2.7, 5.4]]]],
                                      [0, [[], [], [[1.33, 6.75]]]], [1, [[], [], [[1.25, 0.45]]]]]]
h = MolecularHamiltonian(mol=mol, ecp_type='synthetic', ecp_tables={ELEMENTS[int(z)]: tab(int(z)) for z in set(mol.charges) if z > 2})
wf = NeuralNetworkWaveFunction(h, 'psiformer', dtype=torch.float32, device='cuda:0')
params = wf.init(0, perturb_envelopes=0.05)
eng = wf.engine(params)
for kv in filter(None, os.environ.get('DQMC_OPTS', '').split(',')):
    eng.set_option(kv.split('=')[0], int(kv.split('=')[1]))
smp = DecorrSampler(h, wf, length=10); st = smp.init(1, params, B)
for k in range(10): st, pc, stats = smp.sample(k, st, params)
r = st['r']
eng.local_energy(r, rng=0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(reps): e, _ = eng.local_energy(r, rng=k + 1)
torch.cuda.synchronize()
print('B %d  ms per E_loc pass %.1f' % (B, (time.perf_counter() - t0) / reps * 1e3), 'counts', eng.ecp_counts(), 'refined', eng.last_refined(),
      'E mean %.6f' % float(e.double().mean()), flush=True)
