"""Time per Metropolis sub-step of the headline workload for a list of (option, value) settings: python tools/substep_time.py fused_stagger=0 fused_stagger=1 ..."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepqmc_amd import MolecularHamiltonian, Molecule
from deepqmc_amd.sampling import DecorrSampler
from deepqmc_amd.wf import NeuralNetworkWaveFunction
h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
wf = NeuralNetworkWaveFunction(h, 'paulinet', dtype=torch.float32, device='cuda:0')
params = wf.init(0, perturb_envelopes=0.05)
eng = wf.engine(params)
smp = DecorrSampler(h, wf, length=30); st = smp.init(1, params, 4096)
for k in range(5): st, pc, stats = smp.sample(k, st, params)
for setting in sys.argv[1:] or ['fused=1']:
    name, val = setting.split('=')
    eng.set_option(name, int(val))
    for k in range(3): st, pc, stats = smp.sample(k, st, params)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(40): st, pc, stats = smp.sample(10 + k, st, params)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40 / 30
    print(setting, 'us per sub-step %.1f' % (dt * 1e6), 'acc %.3f' % stats['sampling/acceptance'])
