"""Oracle results on |psi|^2-distributed walkers at the BASELINE batch sizes -> tests/golden/parity_*.npz.

Run in the build container (CPU, float64 NumPy oracle; the large configurations take tens of CPU-minutes and are
dealt to worker processes in independent walker blocks):

    python tests/golden/make_parity_fixtures.py [--procs 6] [config ...]

For each BASELINE.json configuration the same synthetic parameters the GPU tests use (init_params seed 5,
envelope jitter 0.1) are compiled into the layer program, walkers are drawn atom-centred Gaussian and then
EQUILIBRATED with Metropolis sub-steps of the float64 oracle interpreter (the distribution the VMC loop of
bench.py actually evaluates -- Gaussian walkers sit near the nodes of a random-init psi far more often than
|psi|^2-distributed ones do), rounded to float32 (the dtype the HIP path reads), and the oracle's
E_loc / log|psi| / sign / statistics at those rounded positions are stored together with two conditioning
diagnostics per walker: the CI cancellation sum_k |c_k det_k| / |psi| and max_k cond(A_k).
The `-m gpu` tests (tests/test_gpu_parity_full.py) feed the stored float32 walkers to the HIP path and
compare.  A second, un-equilibrated ("raw") set is stored for LiH to document the near-node tail.

The `ecp` configuration (BASELINE configs[3]: benzene with an ECP on the carbons, Psiformer) adds the Gaussian-type
ECP of oracle/ecp.py with SYNTHETIC coefficients (pyscf's tables exist nowhere offline; the table is the one
bench.py --ecp uses): the local terms replace the bare Coulomb attraction, the non-local term runs its 12-point
quadrature -- 12 x 30 electrons x 6 carbons = 2 160 psi ratios per walker -- through the interpreter's value path
with explicit rotation angles that are stored in the fixture.

Every block of walkers is an independent Markov chain ensemble with its own seed, so a fixture does not depend on
the number of worker processes.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
PARAM_SEED, PERTURB = 5, 0.1

# name: (molecule, ansatz, walkers, equilibration sub-steps, walkers per block, Laplacian-mode chunk, ecp)
CONFIGS = {
    'lih_paulinet_4096': ('LiH', 'paulinet', 4096, 200, 1024, 1024, False),
    'n2_ferminet_512': ('N2', 'ferminet', 512, 200, 512, 64, False),
    'benzene_psiformer_8': ('benzene', 'psiformer', 8, 200, 8, 2, False),
    'c4h4_transpsiformer_64': ('cyclobutadiene_square', 'transpsiformer', 64, 200, 64, 4, False),
    'lih_paulinet_raw_1024': ('LiH', 'paulinet', 1024, 0, 1024, 1024, False),
    'lih_psiformer_256': ('LiH', 'psiformer', 256, 200, 256, 256, False),
    # round 3: the BASELINE batch sizes of configs[2..4]
    'n2_ferminet_4096': ('N2', 'ferminet', 4096, 200, 64, 32, False),
    'benzene_psiformer_256': ('benzene', 'psiformer', 256, 200, 8, 2, False),
    'c4h4_transpsiformer_512': ('cyclobutadiene_square', 'transpsiformer', 512, 200, 16, 4, False),
    'benzene_ecp_psiformer_32': ('benzene', 'psiformer', 32, 200, 2, 2, True),
    # round 5: a SECOND synthetic table at the BASELINE-shaped batch -- what the ECP cut-offs of the library (tuned on the
    # 32 walkers above) have never seen: exponents x 4 and / 4, a d channel
    'benzene_ecpB_psiformer_256': ('benzene', 'psiformer', 256, 200, 2, 2, 'B'),
    # round 6: a HOLD-OUT table -- generated after every ECP option of the library was frozen at the round-5 defaults, run once,
    # reported as it came out (tests/test_gpu_parity_full.py::test_ecp_hold_out_table_c_128)
    'benzene_ecpC_psiformer_128': ('benzene', 'psiformer', 128, 200, 2, 2, 'C'),
}
LEGACY = ('lih_paulinet_4096', 'n2_ferminet_512', 'benzene_psiformer_8', 'c4h4_transpsiformer_64', 'lih_paulinet_raw_1024',
          'lih_psiformer_256')      # round-2 fixtures: one block, the original random streams


def ecp_table(z: int):
    """The synthetic table of bench.py --ecp (pyscf ECP format: [n_core, [[l, [r^-2.., r^-1, r^0, r^1 terms]] ...]])."""
    return [2 if z > 2 else 0, [[-1, [[], [[5.4, float(z - 2)]], [[4.6, -4.6]], [[2.7, 5.4]]]],
                                [0, [[], [], [[1.33, 6.75]]]], [1, [[], [], [[1.25, 0.45]]]]]]


def ecp_table_b(z: int):
    """Set B (round 5; the library's ECP thresholds were chosen on set A): the local exponents x 4 / : 4 / x 4, a broad s
    channel (exponent : 4: its non-local weight reaches across the whole ring, so almost no (nucleus, electron) pair falls
    below the drop threshold), a tight p channel (x 4) and an l = 2 channel that set A does not have."""
    return [2 if z > 2 else 0, [[-1, [[], [[21.6, float(z - 2)]], [[1.15, -4.6]], [[10.8, 5.4]]]],
                                [0, [[], [], [[0.3325, 6.75]]]], [1, [[], [], [[5.0, 0.45]]]], [2, [[], [], [[0.8, 0.9]]]]]]


def ecp_table_c(z: int):
    """Set C (round 6, hold-out): exponents unlike A and B -- local terms 9.1 / 2.2 / 6.3, an s channel of exponent 0.71 and a
    p channel of exponent 2.9 (two l channels per atom), other coefficient magnitudes."""
    return [2 if z > 2 else 0, [[-1, [[], [[9.1, float(z - 2)]], [[2.2, -3.1]], [[6.3, 4.0]]]],
                                [0, [[], [], [[0.71, 4.2]]]], [1, [[], [], [[2.9, 1.1]]]]]]


ECP_TABLES = {True: ecp_table, 'A': ecp_table, 'B': ecp_table_b, 'C': ecp_table_c}


def setup(molname, ansatz, ecp=False):
    from deepqmc_amd.hamil import MolecularHamiltonian
    from deepqmc_amd.molecule import Molecule
    from deepqmc_amd.params import init_params
    from deepqmc_amd.program import compile_program
    from deepqmc_amd.spec import ANSATZES
    from oracle import geom
    mol = Molecule.from_name(molname)
    spec = ANSATZES[ansatz](mol.charges) if ansatz == 'transpsiformer' else ANSATZES[ansatz]()
    if ecp:
        from deepqmc_amd.ecp import ELEMENTS
        h = MolecularHamiltonian(mol=mol, ecp_type='synthetic',
                                 ecp_tables={ELEMENTS[int(z)]: ECP_TABLES[ecp](int(z)) for z in set(mol.charges) if z > 2})
    else:
        h = MolecularHamiltonian(mol=mol)
    tree = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=PARAM_SEED, perturb_envelopes=PERTURB)
    prog = compile_program(spec, tree, h.n_up, h.n_down, h.n_nuc, R=mol.coords, eps=geom.F32_EPS)
    return mol, spec, h, prog


def chunks(n, c):
    return [(a, min(a + c, n)) for a in range(0, n, c)]


def block(task):
    """One independent block of walkers: equilibrate, round to float32, evaluate.  -> dict of arrays."""
    name, a0, a1 = task
    os.environ.setdefault('OMP_NUM_THREADS', '1')
    from deepqmc_amd.sampling import synthetic_walkers
    from oracle import geom
    from oracle.program_interp import Interp
    molname, ansatz, B, n_eq, blk, chunk, ecp = CONFIGS[name]
    mol, spec, h, prog = setup(molname, ansatz, ecp)
    R = mol.coords.astype(np.float32).astype(np.float64)        # the f32 build sees the rounded geometry
    charges = h.ns_valence if ecp else mol.charges
    it = Interp(prog, charges, geom.F32_EPS)
    legacy = name in LEGACY
    nb = a1 - a0
    if legacy:
        r = synthetic_walkers(h, B, seed=11)[a0:a1]
        rng = np.random.default_rng(12)
    else:
        r = synthetic_walkers(h, nb, seed=1000 + a0)
        rng = np.random.default_rng(2000 + a0)
    t0 = time.time()
    acc_last = None
    vchunk = max(chunk * 8, 64)
    if n_eq:
        tau = 0.3 if h.n_elec <= 4 else 0.1
        lp = np.concatenate([it.run(r[a:b], R, False)['log'] for a, b in chunks(nb, vchunk)])
        for s in range(n_eq):
            rp = r + tau * rng.standard_normal(r.shape)
            lpp = np.concatenate([it.run(rp[a:b], R, False)['log'] for a, b in chunks(nb, vchunk)])
            acc = 2 * (lpp - lp) > np.log(rng.random(nb))
            r[acc], lp[acc] = rp[acc], lpp[acc]
            acc_last = float(acc.mean())
            tau = tau / (0.57 / max(acc_last, 0.05))
    r = r.astype(np.float32)
    t_eq = time.time() - t0
    N, K = h.n_elec, spec.n_determinants
    fin = [op for op in prog.ops if op.kind == 10][0]
    c = prog.weights[fin.i[1]:fin.i[1] + K] if fin.i[1] >= 0 else np.ones(K)
    out = {k: [] for k in ('e_loc', 'log', 'sign', 'stats', 'kappa', 'cond', 'grad')}
    for a, b in chunks(nb, chunk):
        res = it.run(r[a:b].astype(np.float64), R, True)
        out['e_loc'].append(res['e_loc']); out['log'].append(res['log']); out['sign'].append(res['sign'])
        out['stats'].append(res['stats']); out['grad'].append(res['grad'])
        x = it.logdet[:, :, 0]
        sh = x.max(1)
        pt = c[None] * it.sign_k * np.exp(x - sh[:, None])
        out['kappa'].append(np.abs(pt).sum(1) / np.abs(pt.sum(1)))
        A = it.bufs[prog.buf_names['orbitals']][:, :, 0, :N * N].reshape(b - a, K, N, N)
        out['cond'].append(np.linalg.cond(A).max(1))
    fix = {'r': r, 'e_loc': np.concatenate(out['e_loc']), 'log': np.concatenate(out['log']),
           'sign': np.concatenate(out['sign']).astype(np.int32), 'stats': np.concatenate(out['stats'], axis=1),
           'kappa': np.concatenate(out['kappa']), 'cond': np.concatenate(out['cond']),
           'grad': np.concatenate(out['grad']).astype(np.float64)}
    if ecp:
        # ecp/gaussian_type_ecp.py:127-255 through oracle/ecp.py: local terms instead of the bare attraction, plus the
        # non-local quadrature with psi ratios from the interpreter's value path and explicit rotation angles
        import torch
        from oracle import ecp as oecp
        T = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float64)

        def psi_fn(rq):
            res = it.run(rq.numpy(), R, False)
            return T(res['sign']), T(res['log'])
        n_nl = int(sum(bool((h.pot.nl_params[a] != 0).any()) for a in range(h.n_nuc)))
        phi = rng.uniform(0, np.pi / 5, (nb, n_nl, N))
        v_loc = np.array([float(oecp.local_potential(T(r[b]), T(R), T(h.ns_valence), T(h.pot.loc_params), h.ecp_mask)) for b in range(nb)])
        v_nl = np.array([float(oecp.nonloc_potential(T(r[b]), T(R), T(h.pot.nl_params), psi_fn, T(phi[b]))) for b in range(nb)])
        fix['e_loc'] = fix['e_loc'] - fix['stats'][2] + v_loc + v_nl
        fix['stats'][2], fix['stats'][3] = v_loc, v_nl
        fix['ecp_phi'] = phi
    return {'task': task, 'fix': fix, 'acc_last': acc_last, 'seconds': time.time() - t0, 'seconds_equilibration': t_eq}


def assemble(name, results):
    from oracle import geom
    molname, ansatz, B, n_eq, blk, chunk, ecp = CONFIGS[name]
    results = sorted(results, key=lambda x: x['task'][1])
    fix = {}
    for k in results[0]['fix']:
        ax = 1 if k == 'stats' else 0
        fix[k] = np.concatenate([x['fix'][k] for x in results], axis=ax)
    if B > 1024:       # keep the committed fixture small: the gradient of every 8th walker
        fix['grad'] = fix['grad'][::8]
    elif fix['grad'].size > 2_000_000:
        fix['grad'] = fix['grad'][::8]
    meta = {'molecule': molname, 'ansatz': ansatz, 'walkers': B, 'equilibration_sub_steps': n_eq,
            'param_seed': PARAM_SEED, 'perturb_envelopes': PERTURB, 'norm_eps': geom.F32_EPS, 'ecp': bool(ecp), 'ecp_table': (ecp if ecp in ('B', 'C') else 'A') if ecp else None,
            'blocks': len(results), 'acceptance_last': float(np.mean([x['acc_last'] for x in results])) if n_eq else None,
            'cpu_seconds': round(sum(x['seconds'] for x in results), 1),
            'cpu_seconds_equilibration': round(sum(x['seconds_equilibration'] for x in results), 1)}
    np.savez_compressed(os.path.join(HERE, f'parity_{name}.npz'), meta=json.dumps(meta), **fix)
    print(name, meta, 'kappa q50/q99/max', np.quantile(fix['kappa'], [.5, .99, 1.0]), flush=True)


if __name__ == '__main__':
    import multiprocessing as mp
    args = sys.argv[1:]
    procs = 6
    if '--procs' in args:
        i = args.index('--procs')
        procs = int(args[i + 1])
        del args[i:i + 2]
    names = args or list(CONFIGS)
    tasks = []
    for nm in names:
        B, blk = CONFIGS[nm][2], CONFIGS[nm][4]
        tasks += [(nm, a, b) for a, b in chunks(B, blk)]
    os.environ['OMP_NUM_THREADS'] = '1'
    t0 = time.time()
    done = {nm: [] for nm in names}
    need = {nm: len(chunks(CONFIGS[nm][2], CONFIGS[nm][4])) for nm in names}
    with mp.get_context('spawn').Pool(procs) as pool:
        for res in pool.imap_unordered(block, tasks):
            nm = res['task'][0]
            done[nm].append(res)
            print(f'  {nm}: block {len(done[nm])}/{need[nm]} ({time.time() - t0:.0f} s wall)', flush=True)
            if len(done[nm]) == need[nm]:
                assemble(nm, done[nm])
