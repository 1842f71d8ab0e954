"""Experiment: the 4096-walker Metropolis loop as two half-batches on two HIP streams (two contexts) vs one batch."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepqmc_amd import MolecularHamiltonian, Molecule
from deepqmc_amd.engine import Engine
from deepqmc_amd.sampling import DecorrSampler, synthetic_walkers
from deepqmc_amd.wf import NeuralNetworkWaveFunction
h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
wf = NeuralNetworkWaveFunction(h, 'paulinet', dtype=torch.float32, device='cuda:0')
params = wf.init(0, perturb_envelopes=0.05)
dev = torch.device('cuda:0')
def make_state(eng, n, seed):
    r = torch.as_tensor(synthetic_walkers(h, n, seed=seed).astype(np.float32), device=dev)
    sg, lg = eng.wf_eval(r)
    return {'r': r, 'log': lg, 'sign': sg, 'age': torch.zeros(n, dtype=torch.int32, device=dev), 'tau': torch.full((1,), 0.3, dtype=torch.float32, device=dev)}
n_sub, reps = 30, 40
for parts in (1, 2, 4):
    streams = [torch.cuda.Stream(dev) for _ in range(parts)]
    engs, states = [], []
    for k, s in enumerate(streams):
        with torch.cuda.stream(s):
            e = Engine(wf.spec, h, params, dtype=torch.float32, device=dev)
            engs.append(e); states.append(make_state(e, 4096 // parts, 10 + k))
    torch.cuda.synchronize()
    def run():
        for k, s in enumerate(streams):
            with torch.cuda.stream(s):
                engs[k].mcmc_steps(states[k], n_sub, seed=1, want_stats=False)
    for _ in range(5): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f'{parts} stream(s) x {4096 // parts} walkers: {dt * 1e3:.3f} ms per 30 sub-steps = {dt / n_sub * 1e6:.1f} us per sub-step of 4096 walkers')
