"""Per-walker calibration data of the float64 refinement, for offline analysis (tools/calib_sim.py): for every parity
fixture and for bench-like VMC trajectories, the error-predictor score of each walker, its plain float32 local energy and
its float64 local energy -> gpurun_out/calib_<name>.npz.  Nothing here is asserted; the numbers decide the library's
defaults ("refine_target_e7", the percentile of the calibration, "refine_direct_pct").

    python tools/calib_data.py [fixture names ... | traj:LiH:paulinet:4096:30 ...]
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
OUT = os.path.join(ROOT, 'gpurun_out')
DEV = 'cuda:0'


def triple(eng, r, phi=None):
    """(score, plain float32 E_loc, float64 E_loc) of the walkers r on one engine."""
    eng.set_option('refine', 1)
    eng.set_option('refine_probe', 0)
    eng.set_option('refine_thresh', 10 ** 9)           # the float32 pass computes the scores, nobody is refined
    e32, _ = eng.local_energy(r, rng=0, ecp_phi=phi)
    score = eng.refine_scores(r.shape[0])
    eng.set_option('refine', 2)
    e64, _ = eng.local_energy(r, rng=0, ecp_phi=phi)
    eng.set_option('refine', 1)
    return score, e32.double().cpu().numpy(), e64.double().cpu().numpy()


def fixture(name):
    from test_gpu_parity_full import load
    d, meta, h, eng = load(name)
    r = torch.as_tensor(d['r'], device=DEV)
    phi = torch.as_tensor(d['ecp_phi'], dtype=torch.float32, device=DEV) if 'ecp_phi' in d.files else None
    if phi is not None:
        eng.set_option('ecp_mixed', 0)                 # kinetic + quadrature of one precision per pass
    score, e32, e64 = triple(eng, r, phi)
    np.savez_compressed(os.path.join(OUT, f'calib_{name}.npz'), score=score, e32=e32, e64=e64, e_ref=d['e_loc'])
    rel = np.abs(e32 - e64) / np.maximum(1, np.abs(e64))
    print(name, 'walkers', len(score), 'plain f32 within 1e-5: %.4f' % (rel < 1e-5).mean(), 'max %.2e' % rel.max(), flush=True)


def trajectory(molname, ansatz, B, n_sub, steps=20):
    from deepqmc_amd import MolecularHamiltonian, Molecule
    from deepqmc_amd.sampling import DecorrSampler
    from deepqmc_amd.wf import NeuralNetworkWaveFunction
    h = MolecularHamiltonian(mol=Molecule.from_name(molname))
    wf = NeuralNetworkWaveFunction(h, ansatz, dtype=torch.float32, device=DEV)
    params = wf.init(0, perturb_envelopes=0.05)        # bench.py's parameters
    eng = wf.engine(params)
    smp = DecorrSampler(h, wf, length=n_sub, in_place=True)
    st = smp.init(1000, params, B)
    burn = DecorrSampler(h, wf, length=50)
    for k in range(8):
        st = burn.sample(900_000 + k, st, params)[0]
    sc, a32, a64 = [], [], []
    for s in range(steps):
        st, pc, _ = smp.sample(s, st, params)
        score, e32, e64 = triple(eng, st['r'])
        sc.append(score); a32.append(e32); a64.append(e64)
    name = f'traj_{molname}_{ansatz}_{B}'
    np.savez_compressed(os.path.join(OUT, f'calib_{name}.npz'), score=np.stack(sc), e32=np.stack(a32), e64=np.stack(a64))
    rel = np.abs(np.stack(a32) - np.stack(a64)) / np.maximum(1, np.abs(np.stack(a64)))
    print(name, 'evaluations', rel.size, 'plain f32 within 1e-5: %.5f' % (rel < 1e-5).mean(), 'max %.2e' % rel.max(), flush=True)


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    for a in sys.argv[1:]:
        if a.startswith('traj:'):
            _, m, an, b, ns = a.split(':')
            trajectory(m, an, int(b), int(ns))
        else:
            fixture(a)
