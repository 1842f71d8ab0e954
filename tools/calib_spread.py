"""Run-to-run spread of the self-calibrated refinement threshold: benzene / Psiformer, B walkers, several sampler seeds; prints the
measured float32 error per unit of score (90th percentile of the calibration sample), the threshold and the walkers refined."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepqmc_amd import MolecularHamiltonian, Molecule
from deepqmc_amd.sampling import DecorrSampler
from deepqmc_amd.wf import NeuralNetworkWaveFunction
mol = sys.argv[1] if len(sys.argv) > 1 else 'benzene'
ans = sys.argv[2] if len(sys.argv) > 2 else 'psiformer'
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
h = MolecularHamiltonian(mol=Molecule.from_name(mol))
for seed in range(1, 7):
    wf = NeuralNetworkWaveFunction(h, ans, dtype=torch.float32, device='cuda:0')
    params = wf.init(0, perturb_envelopes=0.05)
    eng = wf.engine(params)
    for kv in filter(None, os.environ.get('DQMC_OPTS', '').split(',')):
        eng.set_option(kv.split('=')[0], int(kv.split('=')[1]))
    smp = DecorrSampler(h, wf, length=10); st = smp.init(seed, params, B)
    for k in range(40): st, pc, stats = smp.sample(seed * 1000 + k, st, params)
    out = []
    for k in range(3):
        e, _ = eng.local_energy(st['r'], rng=k)
        out.append((eng.refine_info()['error_per_score'], eng.refine_info()['score_threshold'], eng.last_refined()))
        st, pc, stats = smp.sample(seed * 1000 + 50 + k, st, params)
    print('seed', seed, ' '.join('c %.3e thr %.0f refined %d |' % o for o in out), flush=True)
    wf.release() if hasattr(wf, 'release') else None
    del eng, wf
