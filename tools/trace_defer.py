"""A few VMC steps (30 sub-steps + local energy) with the deferred float64 pass, for rocprofv3 --kernel-trace: does the twin's
pass overlap the next float32 pass?  DQMC_OPTS=name=value,... sets library options."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepqmc_amd import MolecularHamiltonian, Molecule
from deepqmc_amd.sampling import DecorrSampler
from deepqmc_amd.wf import NeuralNetworkWaveFunction
h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
wf = NeuralNetworkWaveFunction(h, 'paulinet', dtype=torch.float32, device='cuda:0')
params = wf.init(0, perturb_envelopes=0.05)
eng = wf.engine(params)
for kv in filter(None, os.environ.get('DQMC_OPTS', '').split(',')):
    eng.set_option(kv.split('=')[0], int(kv.split('=')[1]))
smp = DecorrSampler(h, wf, length=30, in_place=True); st = smp.init(1, params, 4096)
burn = DecorrSampler(h, wf, length=50, in_place=True)
for k in range(8): st, pc, stats = burn.sample(k, st, params)
held = None
for k in range(6):
    st, pc, stats = smp.sample(100 + k, st, params); e, _ = eng.local_energy(st['r']); held = e
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 40
for k in range(n):
    st, pc, stats = smp.sample(200 + k, st, params); e, _ = eng.local_energy(st['r']); held = e
eng.refine_finish(); torch.cuda.synchronize()
print('ms per step %.3f' % ((time.perf_counter() - t0) / n * 1e3), 'refined', eng.last_refined(), os.environ.get('DQMC_OPTS', ''))
