"""20 local-energy passes on the headline workload (for rocprofv3 --kernel-trace --stats of the E_loc path alone)."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepqmc_amd import MolecularHamiltonian, Molecule
from deepqmc_amd.sampling import DecorrSampler
from deepqmc_amd.wf import NeuralNetworkWaveFunction
refine = int(sys.argv[1]) if len(sys.argv) > 1 else 1
h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
wf = NeuralNetworkWaveFunction(h, 'paulinet', dtype=torch.float32, device='cuda:0')
params = wf.init(0, perturb_envelopes=0.05)
eng = wf.engine(params); eng.set_option('refine', refine)
for kv in filter(None, os.environ.get('DQMC_OPTS', '').split(',')):      # extra library options: DQMC_OPTS=linear_bf=0,multi_stream=0
    eng.set_option(kv.split('=')[0], int(kv.split('=')[1]))
if os.environ.get('DQMC_SERIAL'):      # one stream: clean per-kernel counters
    eng.set_option('dual_stream', 0)
smp = DecorrSampler(h, wf, length=30); st = smp.init(1, params, 4096)
for k in range(5): st, pc, stats = smp.sample(k, st, params)
r = st['r']
for _ in range(3): eng.local_energy(r, rng=0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(20): eng.local_energy(r, rng=k)
torch.cuda.synchronize(); print('ms per E_loc pass %.3f' % ((time.perf_counter() - t0) / 20 * 1e3), 'refined', eng.last_refined())
