"""Oracle results on |psi|^2-distributed walkers at the BASELINE batch sizes -> tests/golden/parity_*.npz.

Run in the build container (CPU, float64 NumPy oracle; minutes per configuration):

    python tests/golden/make_parity_fixtures.py [config ...]

For each BASELINE.json configuration the same synthetic parameters the GPU tests use (init_params seed 5,
envelope jitter 0.1) are compiled into the layer program, walkers are drawn atom-centred Gaussian and then
EQUILIBRATED with Metropolis sub-steps of the float64 oracle interpreter (the distribution the VMC loop of
bench.py actually evaluates -- Gaussian walkers sit near the nodes of a random-init psi far more often than
|psi|^2-distributed ones do), rounded to float32 (the dtype the HIP path reads), and the oracle's
E_loc / log|psi| / sign / statistics at those rounded positions are stored together with two conditioning
diagnostics per walker: the CI cancellation sum_k |c_k det_k| / |psi| and max_k cond(A_k).
The `-m gpu` tests (tests/test_gpu_parity_full.py) feed the stored float32 walkers to the HIP path and
compare.  A second, un-equilibrated ("raw") set is stored for LiH to document the near-node tail.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from deepqmc_amd.hamil import MolecularHamiltonian  # noqa: E402
from deepqmc_amd.molecule import Molecule  # noqa: E402
from deepqmc_amd.params import init_params  # noqa: E402
from deepqmc_amd.program import compile_program  # noqa: E402
from deepqmc_amd.sampling import synthetic_walkers  # noqa: E402
from deepqmc_amd.spec import ANSATZES  # noqa: E402
from oracle import geom  # noqa: E402
from oracle.program_interp import Interp  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
PARAM_SEED, PERTURB = 5, 0.1

# name: (molecule, ansatz, walkers, equilibration sub-steps, oracle chunk)
CONFIGS = {
    'lih_paulinet_4096': ('LiH', 'paulinet', 4096, 200, 1024),
    'n2_ferminet_512': ('N2', 'ferminet', 512, 200, 64),
    'benzene_psiformer_8': ('benzene', 'psiformer', 8, 200, 2),
    'c4h4_transpsiformer_64': ('cyclobutadiene_square', 'transpsiformer', 64, 200, 4),
    'lih_paulinet_raw_1024': ('LiH', 'paulinet', 1024, 0, 1024),
    'lih_psiformer_256': ('LiH', 'psiformer', 256, 200, 256),
}


def setup(molname, ansatz):
    mol = Molecule.from_name(molname)
    spec = ANSATZES[ansatz](mol.charges) if ansatz == 'transpsiformer' else ANSATZES[ansatz]()
    h = MolecularHamiltonian(mol=mol)
    tree = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=PARAM_SEED, perturb_envelopes=PERTURB)
    prog = compile_program(spec, tree, h.n_up, h.n_down, h.n_nuc, R=mol.coords, eps=geom.F32_EPS)
    return mol, spec, h, prog


def chunks(n, c):
    return [(a, min(a + c, n)) for a in range(0, n, c)]


def make(name):
    molname, ansatz, B, n_eq, chunk = CONFIGS[name]
    mol, spec, h, prog = setup(molname, ansatz)
    R = mol.coords.astype(np.float32).astype(np.float64)        # the f32 build sees the rounded geometry
    it = Interp(prog, mol.charges, geom.F32_EPS)
    r = synthetic_walkers(h, B, seed=11)
    rng = np.random.default_rng(12)
    t0 = time.time()
    acc_hist = []
    if n_eq:
        tau = 0.3 if h.n_elec <= 4 else 0.1
        lp = np.concatenate([it.run(r[a:b], R, False)['log'] for a, b in chunks(B, max(chunk * 8, 64))])
        for s in range(n_eq):
            rp = r + tau * rng.standard_normal(r.shape)
            lpp = np.concatenate([it.run(rp[a:b], R, False)['log'] for a, b in chunks(B, max(chunk * 8, 64))])
            acc = 2 * (lpp - lp) > np.log(rng.random(B))
            r[acc], lp[acc] = rp[acc], lpp[acc]
            a_ = float(acc.mean())
            acc_hist.append(a_)
            tau = tau / (0.57 / max(a_, 0.05))
    r = r.astype(np.float32)
    t_eq = time.time() - t0
    N, K = h.n_elec, spec.n_determinants
    fin = [op for op in prog.ops if op.kind == 10][0]
    c = prog.weights[fin.i[1]:fin.i[1] + K] if fin.i[1] >= 0 else np.ones(K)
    out = {k: [] for k in ('e_loc', 'log', 'sign', 'stats', 'kappa', 'cond', 'grad')}
    for a, b in chunks(B, chunk):
        res = it.run(r[a:b].astype(np.float64), R, True)
        out['e_loc'].append(res['e_loc']); out['log'].append(res['log']); out['sign'].append(res['sign'])
        out['stats'].append(res['stats']); out['grad'].append(res['grad'])
        x = it.logdet[:, :, 0]
        sh = x.max(1)
        pt = c[None] * it.sign_k * np.exp(x - sh[:, None])
        out['kappa'].append(np.abs(pt).sum(1) / np.abs(pt.sum(1)))
        A = it.bufs[prog.buf_names['orbitals']][:, :, 0, :N * N].reshape(b - a, K, N, N)
        out['cond'].append(np.linalg.cond(A).max(1))
        print(f'  {name}: oracle {b}/{B}  ({time.time() - t0:.0f} s)', flush=True)
    fix = {'r': r, 'e_loc': np.concatenate(out['e_loc']), 'log': np.concatenate(out['log']),
           'sign': np.concatenate(out['sign']).astype(np.int32), 'stats': np.concatenate(out['stats'], axis=1),
           'kappa': np.concatenate(out['kappa']), 'cond': np.concatenate(out['cond']),
           'grad': np.concatenate(out['grad']).astype(np.float64)}
    if B > 1024:       # keep the committed fixture small: the gradient of every 8th walker
        fix['grad'] = fix['grad'][::8]
    meta = {'molecule': molname, 'ansatz': ansatz, 'walkers': B, 'equilibration_sub_steps': n_eq,
            'param_seed': PARAM_SEED, 'perturb_envelopes': PERTURB, 'norm_eps': geom.F32_EPS,
            'acceptance_last': acc_hist[-1] if acc_hist else None, 'seconds': round(time.time() - t0, 1),
            'seconds_equilibration': round(t_eq, 1)}
    np.savez_compressed(os.path.join(HERE, f'parity_{name}.npz'), meta=json.dumps(meta), **fix)
    print(name, meta, 'kappa q50/q99/max', np.quantile(fix['kappa'], [.5, .99, 1.0]), flush=True)


if __name__ == '__main__':
    for nm in (sys.argv[1:] or list(CONFIGS)):
        make(nm)
