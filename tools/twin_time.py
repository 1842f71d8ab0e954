import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from deepqmc_amd import MolecularHamiltonian, Molecule
from deepqmc_amd.sampling import synthetic_walkers
from deepqmc_amd.wf import NeuralNetworkWaveFunction
h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
wf = NeuralNetworkWaveFunction(h, 'paulinet', dtype=torch.float64, device='cuda:0')
eng = wf.engine(wf.init(0, perturb_envelopes=0.05))
for B in (32, 130, 512):
    r = torch.as_tensor(synthetic_walkers(h, B), device='cuda:0')
    for _ in range(3): eng.local_energy(r)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): eng.local_energy(r)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    eng.timing(True); eng.timing_reset()
    for _ in range(5): eng.local_energy(r)
    rep = eng.timing_report(); eng.timing(False)
    print(B, 'wall ms %.3f' % (dt * 1e3), {k: (round(v['ms'] / 5, 3), v['launches'] // 5) for k, v in rep.items()})
