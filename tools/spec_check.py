"""Plan-specialised sub-step kernel (csrc/gen) against the descriptor-driven one on the device: same noise, same walkers ->
accept bits, positions, ages, step size; then time both.  Usage: python tools/spec_check.py [--walkers 4096] [--nsub 30]"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepqmc_amd import MolecularHamiltonian, Molecule
from deepqmc_amd.sampling import synthetic_walkers
from deepqmc_amd.wf import NeuralNetworkWaveFunction

ap = argparse.ArgumentParser(); ap.add_argument('--walkers', type=int, default=4096); ap.add_argument('--nsub', type=int, default=30)
ap.add_argument('--reps', type=int, default=20)
args = ap.parse_args()
dev = 'cuda:0'
h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
wf = NeuralNetworkWaveFunction(h, 'paulinet', dtype=torch.float32, device=dev)
params = wf.init(0, perturb_envelopes=0.05)
B, n_sub = args.walkers, args.nsub
r0 = torch.as_tensor(synthetic_walkers(h, B).astype(np.float32), device=dev)
g = torch.Generator(device='cpu'); g.manual_seed(1)
noise = torch.randn(n_sub, B, 4, 3, generator=g).to(dev)
unif = torch.rand(n_sub, B, generator=g).to(dev)
res = {}
for spec_on in (0, 1):
    eng = wf.engine(params) if spec_on == 0 else eng
    eng.set_option('fused_spec', spec_on)
    sg, lg = eng.wf_eval(r0)
    st = {'r': r0.clone(), 'log': lg.clone(), 'sign': sg.clone(), 'age': torch.zeros(B, dtype=torch.int32, device=dev),
          'tau': torch.full((1,), 0.3, dtype=torch.float32, device=dev)}
    out, acc = eng.mcmc_steps(st, n_sub, noise=noise, unif=unif, return_accept=True)
    res[spec_on] = {k: v.cpu().numpy().copy() for k, v in st.items()}
    res[spec_on]['acc'] = acc.cpu().numpy().copy()
    print('spec', spec_on, out)
    # timing: n_sub sub-steps per call with library noise
    for _ in range(3): eng.mcmc_steps(st, n_sub, seed=5, want_stats=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.reps): eng.mcmc_steps(st, n_sub, seed=7, want_stats=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'spec {spec_on}: {dt / args.reps / n_sub * 1e6:.1f} us per sub-step (wall, {B} walkers)')
a, b = res[0], res[1]
nd = int((a['acc'] != b['acc']).sum())
print('accept bits differing:', nd, 'of', a['acc'].size, '| first sub-step:', int((a['acc'][0] != b['acc'][0]).sum()))
same = (a['acc'] == b['acc']).all(axis=0)
print('walkers with identical accept history:', int(same.sum()), 'of', B)
print('on those: r maxdiff', float(np.abs(a['r'][same] - b['r'][same]).max()), 'log maxdiff', float(np.abs(a['log'][same] - b['log'][same]).max()),
      'sign eq', bool((a['sign'][same] == b['sign'][same]).all()), 'age eq', bool((a['age'][same] == b['age'][same]).all()), 'tau', a['tau'], b['tau'])
