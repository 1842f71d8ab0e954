"""float32 parity at the BASELINE batch sizes: the HIP path (through the C ABI, LIBRARY DEFAULTS: fused sub-step
kernel, MFMA attention / MFMA slogdet where they are the default, self-calibrated float64 refinement) against the
oracle's float64 results on the SAME float32 walkers, committed as fixtures by tests/golden/make_parity_fixtures.py
(|psi|^2-equilibrated walkers = what the VMC loop evaluates; one raw Gaussian set documents the near-node tail).
Round 3 adds the BASELINE batch sizes of configs[2..4]: N2 4096, benzene 256 all-electron, benzene + ECP (synthetic
coefficients) 32 through the Psiformer / MFMA value path with its 2 160 psi ratios per walker, C4H4 512.

Asserted, per configuration (north star: "within 1e-5 Ha relative", sign bit-exact), AT LIBRARY DEFAULTS:
  * every psi sign equals the oracle's;
  * EVERY walker of the BASELINE-size sets with |E - E_ref| / max(1, |E_ref|) < 1e-5 (>= 99 % for the auxiliary sets),
    and explicit p50 / p99 / max bounds (table BOUNDS below -- plain percentiles), on the first call of a fresh context (the call that runs the
    calibration probe) AND on the following call (the calibrated steady state);
  * log|psi| absolute error percentiles.
With the refinement off the same numbers are recorded (not asserted) so that the report shows what it buys, and
with option "refine" = 2 (the whole E_loc pass in float64, sampling stays float32) every configuration must agree
with the oracle to float32 OUTPUT rounding (< 2e-7) for 100 % of the walkers.

How the default gets there (engine.hip: lap_refined): the float32 error of E_loc is predicted per
walker by score = (|lap| + |grad|^2) / max(1, |E_loc|) x conditioning record of the Slater matrices; on the first call
(and every 32nd) <= 256 extra walkers are evaluated in float64; measured, err = m x score x (exponential factor), the sample
gives m, and the threshold is the largest one whose kept walkers miss 1e-5 at an expected rate <= 1e-8 (the shipped default,
`refine_miss_e9` = 10; 1e-7 until the fresh accumulators of round 5); walkers above it are
re-evaluated by the float64 twin.  LiH / PauliNet refines ~8 % of the fixture's walkers (~18 % along the bench trajectory),
N2 / FermiNet ~12 %, the Psiformers and the random-init TransPsiformer nearly all -- those run in the direct float64 mode.
Everything lands in gpurun_out/parity_report.json -> profiles/r0N_parity_report.json.
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

from deepqmc_amd.engine import Engine
from deepqmc_amd.hamil import MolecularHamiltonian
from deepqmc_amd.molecule import Molecule
from deepqmc_amd.params import init_params
from deepqmc_amd.spec import ANSATZES
from oracle import geom

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out')

# name: (min fraction within 1e-5, p50 bound, p99 bound, max bound, log|psi| p99 bound)
# Every set asserts the north-star tolerance itself -- EVERY walker within 1e-5 -- with ONE documented exception.  Round 5: the
# score threshold promises a miss rate of 1e-8 (`refine_miss_e9` 10) under the measured exponential error model (engine.hip, above refine_thresh),
# which refines 2-3 x more walkers than the 90th-percentile rule of rounds 3-4 did (profiles/r05_calibration_model.txt).
BOUNDS = {
    'lih_paulinet_4096': (1.0, 1e-6, 1e-5, 1e-5, 1e-5),
    'n2_ferminet_4096': (1.0, 2e-6, 1e-5, 1e-5, 1e-4),
    'benzene_psiformer_256': (1.0, 3e-6, 1e-5, 1e-5, 1e-3),
    'c4h4_transpsiformer_512': (1.0, 1e-6, 1e-5, 1e-5, 1e-3),
    'benzene_ecp_psiformer_32': (1.0, 2e-6, 1e-5, 1e-5, 1e-3),    # + V_nl: 2 160 psi ratios per walker (float64 for refined walkers)
    # auxiliary sets (small batches of the same systems, raw Gaussian walkers)
    'lih_psiformer_256': (1.0, 2e-6, 1e-5, 1e-5, 2e-5),
    # ONE walker of this set (index 468: score 86, float32 error 1.40e-5 -- 17 x the scale m x score, i.e. a 2e-8 event under the
    # exponential model that fits the other 170 k evaluations we hold; nothing singular about it: cond(A) 3e4, nearest
    # nucleus 0.24 bohr) is a model outlier that only a threshold below ITS score catches.  Kept as the documented
    # exception instead of tuning the default to a single walker: >= 99.8 % of the set and max < 2e-5 are asserted.
    'n2_ferminet_512': (0.998, 2e-6, 1e-5, 2e-5, 1e-4),
    'benzene_psiformer_8': (1.0, 3e-6, 1e-5, 1e-5, 5e-4),
    'c4h4_transpsiformer_64': (1.0, 1e-6, 1e-5, 1e-5, 1e-3),      # the probe sends this system to the direct float64 pass
    'lih_paulinet_raw_1024': (1.0, 1e-6, 1e-5, 1e-5, 5e-4),       # raw Gaussian walkers (not what the VMC loop evaluates): ~50 % refined
}


def load(name, dtype=torch.float32):
    path = os.path.join(ROOT, 'tests', 'golden', f'parity_{name}.npz')
    if not os.path.exists(path):
        pytest.skip(f'fixture {path} not generated')
    d = np.load(path)
    meta = json.loads(str(d['meta']))
    mol = Molecule.from_name(meta['molecule'])
    spec = ANSATZES[meta['ansatz']](mol.charges) if meta['ansatz'] == 'transpsiformer' else ANSATZES[meta['ansatz']]()
    if meta.get('ecp'):
        from deepqmc_amd.ecp import ELEMENTS
        sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
        from make_parity_fixtures import ECP_TABLES
        table = ECP_TABLES[meta.get('ecp_table') or 'A']
        h = MolecularHamiltonian(mol=mol, ecp_type='synthetic',
                                 ecp_tables={ELEMENTS[int(z)]: table(int(z)) for z in set(mol.charges) if z > 2})
    else:
        h = MolecularHamiltonian(mol=mol)
    tree = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=meta['param_seed'], perturb_envelopes=meta['perturb_envelopes'])
    # (a float64 context is built at the float32-rounded geometry the fixtures were generated at -- the one a float32 context sees)
    R = None if dtype == torch.float32 else mol.coords.astype(np.float32).astype(np.float64)
    eng = Engine(spec, h, tree, dtype=dtype, device=DEV, norm_eps=meta['norm_eps'], R=R)
    return d, meta, h, eng


def profile(e, ref):
    rel = np.abs(e - ref) / np.maximum(1.0, np.abs(ref))
    return rel, {'frac_within_1e-5': float((rel < 1e-5).mean()), 'p50': float(np.quantile(rel, 0.5)),
                 'p90': float(np.quantile(rel, 0.9)), 'p99': float(np.quantile(rel, 0.99)), 'max': float(rel.max())}


def report(name, payload):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, 'parity_report.json')
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[name] = payload
    json.dump(data, open(path, 'w'), indent=1)


@pytest.mark.parametrize('name', list(BOUNDS))
def test_f32_parity_at_baseline_size(name):
    d, meta, h, eng = load(name)
    r = torch.as_tensor(d['r'], device=DEV)
    B = r.shape[0]
    frac_min, p50_max, p99_max, max_max, lp99_max = BOUNDS[name]
    phi = torch.as_tensor(d['ecp_phi'], dtype=torch.float32, device=DEV) if 'ecp_phi' in d.files else None
    out = {}
    # refine 0: plain float32; then the library default twice: the probing first call and the calibrated steady state
    for key, refine in (('refine_off', 0), ('refine_on', 1), ('refine_on_second_call', 1)):
        eng.set_option('refine', refine)
        e, stats, grad = eng.local_energy(r, rng=0, return_grad=True, ecp_phi=phi)
        n_ref = eng.last_refined()
        rel, prof = profile(e.double().cpu().numpy(), d['e_loc'])
        prof['n_refined'] = n_ref
        if refine:
            prof['refine_info'] = eng.refine_info()
        out[key] = prof
    if phi is not None:       # the potential terms of the ECP Hamiltonian against oracle/ecp.py (gaussian_type_ecp.py:127-255)
        v_nl, v_loc = stats['hamil/V_nl'].double().cpu().numpy(), stats['hamil/V_loc'].double().cpu().numpy()
        out['ecp'] = {'V_nl_abs_err_max': float(np.abs(v_nl - d['stats'][3]).max()), 'V_nl_abs_mean': float(np.abs(d['stats'][3]).mean()),
                      'V_loc_rel_err_max': float((np.abs(v_loc - d['stats'][2]) / np.abs(d['stats'][2])).max())}
    has_f64 = True                      # (the scalar float64 attention splits its queries over workgroups for 42 electrons)
    if has_f64:
        eng.set_option('refine', 2)                                   # E_loc pass entirely in float64 (sampling stays f32)
        e2, _ = eng.local_energy(r, rng=0, ecp_phi=phi)
        _, out['eloc_f64'] = profile(e2.double().cpu().numpy(), d['e_loc'])
        eng.set_option('refine', 1)
    sign, logpsi = eng.wf_eval(r)                                     # value path (fused kernel where it exists)
    lp = np.abs(logpsi.double().cpu().numpy() - d['log'])
    g = grad.double().cpu().numpy()
    g_ref = d['grad'] if d['grad'].shape[0] == B else None
    payload = {'walkers': B, 'equilibration_sub_steps': meta['equilibration_sub_steps'], **out,
               'logpsi_abs_err': {'p50': float(np.median(lp)), 'p99': float(np.quantile(lp, 0.99)), 'max': float(lp.max())},
               'sign_mismatches': int((sign.cpu().numpy() != d['sign']).sum()),
               'kappa_ci_cancellation': {'p50': float(np.median(d['kappa'])), 'p99': float(np.quantile(d['kappa'], 0.99)), 'max': float(d['kappa'].max())},
               'cond_slater': {'p50': float(np.median(d['cond'])), 'p99': float(np.quantile(d['cond'], 0.99))},
               'corr_log_err_log_kappa': float(np.corrcoef(np.log(rel + 1e-12), np.log(d['kappa']))[0, 1]),
               'corr_log_err_log_cond': float(np.corrcoef(np.log(rel + 1e-12), np.log(d['cond']))[0, 1]),
               'bounds_asserted': dict(zip(('frac_within_1e-5_min', 'p50_max', 'p99_max', 'max_max', 'logpsi_p99_max'), BOUNDS[name]))}
    if g_ref is not None:
        gs = np.abs(g - g_ref) / np.maximum(1.0, np.abs(g_ref))
        payload['grad_rel_err_p99'] = float(np.quantile(gs, 0.99))
    # value path (what the Metropolis acceptance sees; plain float32, no refinement): absolute bound per configuration,
    # relative to |log|psi|| in the report (42-electron Psiformer: |log|psi|| ~ 1e2, p99 6e-4 absolute = 8e-6 relative)
    lp_rel = lp / np.maximum(1.0, np.abs(d['log']))
    payload['logpsi_rel_err_p99'] = float(np.quantile(lp_rel, 0.99))
    report(f'f32_full_{name}', payload)
    np.testing.assert_array_equal(sign.cpu().numpy(), d['sign'])       # bit-exact item
    for key in ('refine_on', 'refine_on_second_call'):
        prof = out[key]
        assert prof['frac_within_1e-5'] >= frac_min, (key, prof)
        assert prof['p50'] < p50_max and prof['p99'] < p99_max and prof['max'] < max_max, (key, prof)
    if phi is not None:
        # (V_nl enters E_loc ~ 40 Ha: its absolute error is held to 1e-6 of that, ten times inside the tolerance of E_loc itself;
        # observed 5.9e-7 Ha on set A, 1.3e-5 on B, 2.3e-5 on the hold-out C)
        assert out['ecp']['V_loc_rel_err_max'] < 1e-5 and out['ecp']['V_nl_abs_err_max'] < 1e-6 * np.abs(d['e_loc']).max(), out['ecp']
    assert np.quantile(lp, 0.99) < lp99_max and np.quantile(lp_rel, 0.99) < 5e-5, payload['logpsi_abs_err']
    if has_f64:
        assert out['eloc_f64']['max'] < 2e-7, out['eloc_f64']          # float32 output rounding of a float64 evaluation
    # the sampler state path: psi of the same walkers through dqmc_mcmc_steps' own evaluation must agree with wf_eval
    st = {'r': r.clone(), 'log': logpsi.clone(), 'sign': sign.clone(), 'age': torch.zeros(B, dtype=torch.int32, device=DEV),
          'tau': torch.full((1,), 1e-12, dtype=torch.float32, device=DEV)}
    eng.mcmc_steps(st, 1, seed=3, target_acceptance=None)             # a zero-length move: psi' == psi up to round-off
    assert torch.equal(st['sign'], sign)
    # (round 6: the sub-step of LiH / PauliNet runs on the plan-specialised kernel, wf_eval on the descriptor-driven one -- two
    # float32 kernels with different summation orders.  Each is held to the float64 fixture by the SAME bound; between
    # themselves they agree to round-off except on walkers next to a node of psi, where either carries ~1e-4)
    lp_sub = np.abs(st['log'].cpu().numpy().astype(np.float64) - d['log'])
    assert np.quantile(lp_sub, 0.99) < lp99_max and np.quantile(lp_sub / np.maximum(1.0, np.abs(d['log'])), 0.99) < 5e-5
    rel = ((st['log'] - logpsi).abs() / logpsi.abs().clamp(min=1.0)).cpu().numpy()
    assert np.quantile(rel, 0.99) < 3e-5 and rel.max() < 1e-3, (np.quantile(rel, 0.99), rel.max())


def test_staged_metropolis_n2_bit_exact_f64():
    """The staged sub-step path (N > 4: k_propose -> layered / fused psi -> k_slogdet_lu -> k_final -> k_accept ->
    k_tau_update) on N2 / FermiNet in float64 against oracle/sampling.py on the same noise: accept bits, ages,
    positions, tau, and all seven sampler statistics."""
    from deepqmc_amd.sampling import synthetic_walkers
    from deepqmc_amd.spec import ferminet
    from oracle import sampling as osamp
    from oracle import wf as owf
    mol = Molecule.from_name('N2')
    spec = ferminet()
    h = MolecularHamiltonian(mol=mol)
    tree = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=5, perturb_envelopes=0.1)
    eng = Engine(spec, h, tree, dtype=torch.float64, device=DEV, norm_eps=geom.F32_EPS)
    B, n_sub = 24, 5
    rng = np.random.default_rng(0)
    r0 = synthetic_walkers(h, B, seed=2)
    noise = rng.standard_normal((n_sub, B, h.n_elec, 3))
    unif = rng.random((n_sub, B))
    sign0, log0 = eng.wf_eval(torch.as_tensor(r0, device=DEV))
    st = {'r': torch.as_tensor(r0, device=DEV).clone(), 'log': log0.clone(), 'sign': sign0.clone(),
          'age': torch.zeros(B, dtype=torch.int32, device=DEV), 'tau': torch.full((1,), 0.1, dtype=torch.float64, device=DEV)}
    stats, acc = eng.mcmc_steps(st, n_sub, max_age=3, target_acceptance=0.57, noise=noise, unif=unif, return_accept=True)
    p = owf.to_torch(tree)
    T = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
    ost = {'r': T(r0), 'sign': T(sign0.cpu().numpy()), 'log': T(log0.cpu().numpy()), 'age': torch.zeros(B, dtype=torch.int64), 'tau': 0.1}
    ost, ostats, oacc = osamp.decorr_sample(p, spec, ost, T(mol.coords), h.n_up, geom.F32_EPS, T(noise), T(unif), max_age=3,
                                            target_acceptance=0.57)
    np.testing.assert_array_equal(acc.cpu().numpy().astype(bool), oacc.numpy())
    np.testing.assert_array_equal(st['age'].cpu().numpy(), ost['age'].numpy())
    np.testing.assert_allclose(st['r'].cpu().numpy(), ost['r'].numpy(), rtol=0, atol=1e-12)
    np.testing.assert_allclose(float(st['tau'][0]), ost['tau'], rtol=1e-12)
    np.testing.assert_allclose(st['log'].cpu().numpy(), ost['log'].numpy(), rtol=1e-10, atol=1e-10)
    for k in ostats:
        np.testing.assert_allclose(stats[k], ostats[k], rtol=1e-9, atol=1e-9, err_msg=k)
    report('staged_mcmc_n2_f64', {'walkers': B, 'sub_steps': n_sub, 'accept_bits_equal': True,
                                  'acceptance_last': stats['sampling/acceptance']})


def test_chunked_workspace_benzene_256_bit_equal():
    """BASELINE configs[3] runs 2048 benzene walkers per GPU, which only ever fit as workspace CHUNKS (Engine::run:
    ~0.2 GB of Laplacian-mode activations per walker).  Here the 256-walker fixture with a budget that forces >= 4
    chunks in the float32 engine AND in its float64 twin: bit-equal to the unchunked evaluation (energies, statistics,
    gradient, psi), and within the fixture's bounds."""
    d, meta, h, eng = load('benzene_psiformer_256')
    r = torch.as_tensor(d['r'], device=DEV)
    e0, st0, g0 = eng.local_energy(r, return_grad=True)
    s0, l0 = eng.wf_eval(r)
    n_ref0 = eng.last_refined()
    for name in ('ws_budget_mb', 'twin.ws_budget_mb'):
        eng.set_option(name, 10 * 1024)           # 10 GiB: ~50 float32 / ~25 float64 walkers per chunk
    e1, st1, g1 = eng.local_energy(r, return_grad=True)
    info = eng.last_chunks()
    s1, l1 = eng.wf_eval(r)
    assert eng.last_refined() == n_ref0
    assert max(info['own'], info['twin']) >= 4, info                 # (the whole batch may go to the float64 twin)
    assert torch.equal(e1, e0) and torch.equal(g1, g0) and torch.equal(s1, s0) and torch.equal(l1, l0)
    for k in st0:
        assert torch.equal(st1[k], st0[k]), k
    rel, prof = profile(e1.double().cpu().numpy(), d['e_loc'])
    report('chunked_benzene_256', {'walkers': int(r.shape[0]), 'workspace': info, **prof})
    assert prof['frac_within_1e-5'] == 1.0 and prof['max'] < 1e-5, prof


def _three_state_reference(fx, S, Bp):
    logs = fx['log'][:, :S * Bp].reshape(S, S, Bp)              # [i, j, b]: psi_i on the walkers of state j
    signs = fx['sign'][:, :S * Bp].reshape(S, S, Bp).astype(np.float64)
    shifted = logs - logs.mean(axis=(1, 2))[:, None, None]
    dg = np.stack([shifted[j, j] for j in range(S)])
    sdg = np.stack([signs[j, j] for j in range(S)])
    ref = signs * sdg[None] * np.exp(shifted - dg[None])
    mean = ref.mean(-1)
    sym = np.sign(mean) * np.sqrt(np.clip(mean * mean.T, 0, None))
    return ref, sym, sum(sym[i, j] ** 2 for i in range(S) for j in range(i + 1, S))


@pytest.mark.parametrize('dtype', ['f32', 'f64'])
def test_three_states_c4h4_local_energy_psi_ratio_overlap(dtype):
    """BASELINE configs[4] on its own ansatz: `compute_local_energy [1, 3, B]` (loss/energy.py:19-60), the psi-ratio matrix
    and the overlap penalty (loss/overlap.py:40-149) for three parameter sets on the walkers of the c4h4_512 fixture, against
    the oracle's float64 values (tests/golden/states_c4h4_transpsiformer.npz, made by make_states_fixture.py).
      * f32: LIBRARY DEFAULTS (float32 contexts, MFMA-attention value path, self-calibrated refinement of E_loc).  E_loc of
        every walker within 1e-5; psi signs bit-exact.  The ratios are exp(differences of float32 log|psi| values) of a
        random-init TransPsiformer whose Slater matrices have cond ~ 1e6, evaluated on walkers that are |psi|^2-distributed
        for ONE of the three parameter sets only: float32 puts ~1e-7 x cond on log|psi| whoever computes it (the reference
        runs this in float32 too), so the ratio bounds are the float32 bounds of this value path, not 1e-5;
      * f64: the same quantities from float64 contexts, to 1e-7 -- the algorithm itself (log-shifted ratios, weighted mean,
        clipped geometric mean) is pinned tightly, what is loose above is arithmetic."""
    from deepqmc_amd import loss
    from deepqmc_amd.wf import NeuralNetworkWaveFunction
    path = os.path.join(ROOT, 'tests', 'golden', 'states_c4h4_transpsiformer.npz')
    fx = np.load(path)
    meta = json.loads(str(fx['meta']))
    r_all = np.load(os.path.join(ROOT, 'tests', 'golden', meta['walkers_from']))['r']
    mol = Molecule.from_name(meta['molecule'])
    h = MolecularHamiltonian(mol=mol)
    tdt = torch.float32 if dtype == 'f32' else torch.float64
    wf = NeuralNetworkWaveFunction(h, 'transpsiformer', dtype=tdt, device=DEV, norm_eps=geom.F32_EPS)
    params = [wf.init(s, perturb_envelopes=meta['perturb_envelopes']) for s in meta['param_seeds']]
    S, Be = 3, meta['b_eloc']
    # --- local energies [1, 3, B]: state s on walkers [B s, B s + B) ---
    # The fixture is what a float32 context sees: the nuclear stream folded on the host at the Hamiltonian's geometry, the
    # electron-nucleus features at the float32-ROUNDED geometry.  With cond(A) ~ 1e6 that rounding moves log|psi| by ~1e-3,
    # so the float64 contexts are given exactly the same view (their run-time R replaced by the rounded one).
    if dtype == 'f64':
        for p_ in params:
            eng_ = wf.engine(p_)
            eng_.R = torch.as_tensor(mol.coords.astype(np.float32).astype(np.float64), device=DEV)
    Rr = None
    pc = lambda r: r
    r_e = torch.as_tensor(r_all[:S * Be].reshape(1, S, Be, h.n_elec, 3), dtype=tdt, device=DEV)
    E, stats = loss.compute_local_energy(0, h, wf, params, pc(r_e))
    assert E.shape == (1, S, Be) and stats['hamil/E_kin'].shape == (1, S)
    rel = np.abs(E[0].double().cpu().numpy() - fx['e_loc']) / np.maximum(1.0, np.abs(fx['e_loc']))
    # --- psi ratios: three states x 170 walkers each ---
    Bp = 170 if dtype == 'f32' else 48
    r_p = torch.as_tensor(r_all[:S * Bp].reshape(1, S, Bp, h.n_elec, 3), dtype=tdt, device=DEV)
    ratio, ov_stats = loss.compute_psi_ratio(wf, params, pc(r_p))
    assert ov_stats == {}                                                           # (overlap.py:50: the reference returns an empty Stats)
    assert ratio.shape == (1, S, S, Bp)
    ref, sym, pen_ref = _three_state_reference(fx, S, Bp)
    got = ratio[0].cpu().numpy()
    np.testing.assert_array_equal(np.sign(got), np.sign(ref))                     # sign work: bit-exact
    rr = np.abs(got - ref) / np.abs(ref)
    w = torch.ones(1, S, Bp, dtype=torch.float64, device=DEV)
    pen, info = loss.compute_mean_overlap(ratio, w)
    ov = info['overlap/pairwise/mean'][0].cpu().numpy()
    # the value path behind the ratios in numbers: log|psi| of every (state, walker)
    lg = np.stack([wf.engine(params[i], Rr).wf_eval(r_p[0].reshape(S * Bp, h.n_elec, 3), Rr)[1].double().cpu().numpy() for i in range(S)])
    lerr = np.abs(lg - fx['log'][:, :S * Bp])
    ov_rel = float(np.abs(ov - sym).max() / np.abs(sym).max())
    report(f'three_states_c4h4_{dtype}', {
        'e_loc_rel_err': {'p50': float(np.median(rel)), 'max': float(rel.max()), 'frac_within_1e-5': float((rel < 1e-5).mean())},
        'psi_ratio_rel_err': {'p50': float(np.median(rr)), 'p99': float(np.quantile(rr, 0.99)), 'max': float(rr.max())},
        'logpsi_abs_err': {'p50': float(np.median(lerr)), 'p99': float(np.quantile(lerr, 0.99)), 'max': float(lerr.max())},
        'overlap_rel_err_max': ov_rel, 'overlap_ref': sym.tolist(), 'penalty': float(pen), 'penalty_ref': float(pen_ref)})
    if dtype == 'f64':
        assert rel.max() < 1e-9 and rr.max() < 1e-7 and lerr.max() < 1e-8, (rel.max(), rr.max(), lerr.max())
        np.testing.assert_allclose(ov, sym, rtol=1e-7, atol=1e-9 * np.abs(sym).max())
        np.testing.assert_allclose(float(pen), pen_ref, rtol=1e-7)
        return
    assert (rel < 1e-5).all(), rel.max()                                          # the north-star tolerance on E_loc
    # observed on the MI355X (profiles/r04_parity_report.json): |d log psi| p50 1.2e-4, p99 7.0e-3, max 0.20; ratio p50 1.6e-4,
    # p99 8.4e-3; overlap matrix 3.3e-4 of its largest entry, penalty 6.7e-4 -- bounds = those with a margin of ~3
    assert np.median(lerr) < 5e-4 and np.quantile(lerr, 0.99) < 2e-2, (np.median(lerr), np.quantile(lerr, 0.99))
    assert np.median(rr) < 5e-4 and np.quantile(rr, 0.99) < 3e-2, (np.median(rr), np.quantile(rr, 0.99))
    assert ov_rel < 2e-3, ov_rel
    np.testing.assert_allclose(float(pen), pen_ref, rtol=3e-3)


# ---- round 5: parity in the regime, and along the trajectory, that bench.py actually runs -------------------------------

@pytest.mark.parametrize('chunked', [False, True])
def test_benzene_regimes_256(chunked):
    """BASELINE configs[3] on the machine.  Round 4's bench.py ran benzene / Psiformer with 28-40 % of the walkers refined in
    float64 and the rest left in float32 -- a regime the 256-walker fixture never exercised, because a majority above the
    threshold sent that batch to float64 whole.  Two parts:
      (1) LIBRARY DEFAULTS (round 5: the threshold promises a miss RATE under the measured exponential error model, engine.hip
          above refine_thresh): nearly every benzene walker lies above it, the context enters the whole-batch float64 mode on
          the calibrating call and stays there; EVERY walker within 1e-5 of the oracle, first and second call.
      (2) THE MIXED REGIME, PINNED (probe off, threshold at the 60th percentile of the batch's scores, mode switch disabled):
          ~40 % refined, 60 % kept in float32 -- what round 4 ran.  Asserted: the mechanics (refined walkers carry the float64
          result, kept ones the plain float32 one, bit for bit; counts).  RECORDED, because it is the finding: the kept
          walkers do NOT all meet 1e-5 (5 k evaluations along the bench trajectory at 49 % refined: 21 beyond it, max 2.1e-5,
          profiles/r05_calibration_model.txt) -- which is why (1) is the default.
    `chunked`: part (2) again under a workspace budget that splits the float32 pass AND the twin's pass into >= 4 chunks each
    (what the 2048-walker BASELINE batch does): bit-equal to the unchunked evaluation."""
    d, meta, h, eng = load('benzene_psiformer_256')
    r = torch.as_tensor(d['r'], device=DEV)
    B = r.shape[0]
    out = {}
    for key in ('first_call', 'second_call'):
        e, _ = eng.local_energy(r)
        rel, prof = profile(e.double().cpu().numpy(), d['e_loc'])
        prof.update({'last_refined': eng.last_refined(), **eng.refine_info(), 'chunks': eng.last_chunks()})
        out[key] = prof
    out['counters'] = eng.refine_counters()
    del eng
    # ---- (2) the mixed regime, pinned ----
    d, meta, h, mix = load('benzene_psiformer_256')
    for name, val in (('refine_probe', 0), ('refine_direct_pct', 100), ('refine_thresh', 10 ** 9)):
        mix.set_option(name, val)
    e_plain, _ = mix.local_energy(r)
    score = mix.refine_scores(B)
    thr = int(np.quantile(score, 0.6))
    mix.set_option('refine_thresh', thr)
    e_mix, st_mix, g_mix = mix.local_energy(r, return_grad=True)
    kept = score <= thr
    assert mix.last_refined() == int((~kept).sum()) and 0.3 * B <= (~kept).sum() <= 0.5 * B
    assert mix.refine_info()['direct_f64_calls_left'] == 0      # (the 32 GB default budget already splits this pass in two)
    if chunked:
        for name in ('ws_budget_mb', 'twin.ws_budget_mb'):
            mix.set_option(name, 10 * 1024)
        e_c, st_c, g_c = mix.local_energy(r, return_grad=True)
        info = mix.last_chunks()
        assert info['own'] >= 4 and info['twin'] >= 4, info
        assert torch.equal(e_c, e_mix) and torch.equal(g_c, g_mix)
        for k in st_mix:
            assert torch.equal(st_c[k], st_mix[k]), k
        out['mixed_chunks'] = info
    mix.set_option('refine', 2)
    e_f64, _ = mix.local_energy(r)
    em, ep, ef = (x.cpu().numpy() for x in (e_mix, e_plain, e_f64))
    np.testing.assert_array_equal(em[kept], ep[kept])                      # kept walkers: the plain float32 result, untouched
    np.testing.assert_array_equal(em[~kept], ef[~kept])                    # refined walkers: the float64 pass's result
    rel = np.abs(em.astype(np.float64) - d['e_loc']) / np.maximum(1.0, np.abs(d['e_loc']))
    out['mixed_regime_pinned'] = {'score_threshold': thr, 'n_refined': int((~kept).sum()), 'n_kept_f32': int(kept.sum()),
                                  'kept_p50': float(np.median(rel[kept])), 'kept_p99': float(np.quantile(rel[kept], 0.99)),
                                  'kept_max': float(rel[kept].max()), 'kept_beyond_1e-5': int((rel[kept] >= 1e-5).sum()),
                                  'refined_max': float(rel[~kept].max())}
    report('benzene_regimes_256' + ('_chunked' if chunked else ''), out)
    assert out['mixed_regime_pinned']['refined_max'] < 2e-7 and out['mixed_regime_pinned']['kept_max'] < 1e-4
    for key in ('first_call', 'second_call'):
        p = out[key]
        assert p['frac_within_1e-5'] == 1.0 and p['max'] < 1e-5, (key, p)      # the north-star tolerance, every walker, at defaults
    assert out['second_call']['direct_f64_calls_left'] > 0, out['second_call']  # (the mode the default runs this system in)


@pytest.mark.parametrize('molname,ansatz,n_sub', [('LiH', 'paulinet', 30), ('N2', 'ferminet', 10)])
def test_trajectory_parity_at_bench_settings(molname, ansatz, n_sub):
    """Parity ALONG THE TRAJECTORY bench.py runs (BASELINE configs[1] / [2]: 4096 walkers, bench's parameters, 400 burn-in
    sub-steps), not on one frozen ensemble: 20 VMC steps; at every step the local energies the library returns at its
    DEFAULTS (self-calibrated refinement, captured passes) against the float64 pass of a second context ("refine" 2: oracle-
    checked to 2e-7 by the fixtures) on the same walkers.  >= 80 k evaluations per system; asserted: EVERY one within 1e-5
    (reference semantics: loss/energy.py:50-57 -- the reference evaluates every walker in one precision).  The tail goes to
    the report, and so does what two LOOSER miss rates ("refine_miss_e9" 1e-7 / 1e-6 instead of the default 1e-8) would save
    and cost on the same walkers."""
    from deepqmc_amd import MolecularHamiltonian as MH, Molecule as Mol
    from deepqmc_amd.sampling import DecorrSampler
    from deepqmc_amd.wf import NeuralNetworkWaveFunction
    B, steps = 4096, 20
    h = MH(mol=Mol.from_name(molname))
    wf = NeuralNetworkWaveFunction(h, ansatz, dtype=torch.float32, device=DEV)
    params = wf.init(0, perturb_envelopes=0.05)
    eng = wf.engine(params)
    ref = Engine(wf.spec, h, params, dtype=torch.float32, device=DEV)
    ref.set_option('refine', 2)
    alt = {}
    for t7 in (100, 1000):
        alt[t7] = Engine(wf.spec, h, params, dtype=torch.float32, device=DEV)
        alt[t7].set_option('refine_miss_e9', t7)
    smp = DecorrSampler(h, wf, length=n_sub, in_place=True)
    st = smp.init(1000, params, B)
    burn = DecorrSampler(h, wf, length=50)
    for k in range(8):
        st = burn.sample(900_000 + k, st, params)[0]
    rels, n_ref, alt_rel, alt_ref = [], [], {t: [] for t in alt}, {t: [] for t in alt}
    for s in range(steps):
        st, pc, _ = smp.sample(s, st, params)
        e, _ = eng.local_energy(st['r'])
        n_ref.append(eng.last_refined())
        e64 = ref.local_energy(st['r'])[0].double().cpu().numpy()
        rels.append(np.abs(e.double().cpu().numpy() - e64) / np.maximum(1.0, np.abs(e64)))
        for t7, a in alt.items():
            ea = a.local_energy(st['r'])[0].double().cpu().numpy()
            alt_rel[t7].append(np.abs(ea - e64) / np.maximum(1.0, np.abs(e64)))
            alt_ref[t7].append(a.last_refined())
    rel = np.concatenate(rels)
    tail = lambda x: {'frac_within_1e-5': float((x < 1e-5).mean()), 'n_above_1e-5': int((x >= 1e-5).sum()), 'n_above_5e-6': int((x >= 5e-6).sum()),
                      'p99': float(np.quantile(x, 0.99)), 'p99.9': float(np.quantile(x, 0.999)), 'max': float(x.max())}
    payload = {'evaluations': int(rel.size), 'steps': steps, 'default': {**tail(rel), 'refined_per_step_mean': float(np.mean(n_ref[1:])),
                                                                        'refined_fraction': float(np.mean(n_ref[1:])) / B},
               'refine_info': eng.refine_info(), 'counters': eng.refine_counters(),
               'max_per_step': [float(x.max()) for x in rels]}
    for t7 in alt:
        x = np.concatenate(alt_rel[t7])
        payload[f'miss_rate_{t7}e-9'] = {**tail(x), 'refined_fraction': float(np.mean(alt_ref[t7][1:])) / B}
    report(f'trajectory_{molname}_{ansatz}_{B}', payload)
    assert rel.size >= 80_000
    assert payload['default']['frac_within_1e-5'] == 1.0, payload['default']


def test_ecp_thresholds_hold_on_a_second_table_256():
    """The cut-offs of the mixed-precision ECP quadrature ("ecp_skip_e12", "ecp_heavy_e6": chosen in round 4 on the 32 walkers
    of set A, profiles/r04_ecp_mixed_precision_sweep.json) on what they were NOT tuned on: a second synthetic table (local
    exponents x 4 / : 4, a broad s channel whose weight reaches across the ring, a tight p channel, an l = 2 channel A does
    not have) at 256 |psi|^2-equilibrated benzene walkers, oracle values from tests/golden/make_parity_fixtures.py
    (oracle/ecp.py <- ecp/gaussian_type_ecp.py:161-255; the reference has ONE precision and no cut-off, :239-244).
    Library defaults.  Asserted: every walker's E_loc within 1e-5 relative, first and second call; V_nl within 1e-5 |E|;
    the share of pairs per class and the V_nl error go to the report."""
    d, meta, h, eng = load('benzene_ecpB_psiformer_256')
    assert meta['ecp_table'] == 'B'
    r = torch.as_tensor(d['r'], device=DEV)
    phi = torch.as_tensor(d['ecp_phi'], dtype=torch.float32, device=DEV)
    out = {}
    for key in ('first_call', 'second_call'):
        e, stats = eng.local_energy(r, rng=0, ecp_phi=phi)
        rel, prof = profile(e.double().cpu().numpy(), d['e_loc'])
        v_nl = stats['hamil/V_nl'].double().cpu().numpy()
        prof.update({'n_refined': eng.last_refined(), 'pairs': eng.ecp_counts(), 'refine_info': eng.refine_info(),
                     'V_nl_abs_err_max': float(np.abs(v_nl - d['stats'][3]).max()), 'V_nl_abs_err_p50': float(np.median(np.abs(v_nl - d['stats'][3]))),
                     'V_nl_abs_mean': float(np.abs(d['stats'][3]).mean()), 'E_abs_mean': float(np.abs(d['e_loc']).mean())})
        out[key] = prof
    eng.set_option('ecp_dlog_floor_e6', 0)             # the round-4 rule (weights alone) on the same walkers: recorded
    e0, _ = eng.local_energy(r, rng=0, ecp_phi=phi)
    _, out['weights_only_rule'] = profile(e0.double().cpu().numpy(), d['e_loc'])
    out['weights_only_rule']['pairs'] = eng.ecp_counts()
    report('ecp_set_b_benzene_256', out)
    for key in ('first_call', 'second_call'):
        p = out[key]
        assert p['frac_within_1e-5'] == 1.0 and p['max'] < 1e-5, (key, p)
        assert p['V_nl_abs_err_max'] < 1e-6 * np.abs(d['e_loc']).max(), (key, p)      # (1.3e-5 Ha of |E| ~ 42 Ha: 3e-7 relative)


def test_ecp_hold_out_table_c_128():
    """Round 6, the hold-out asked for by the round-5 review: EVERY option of the ECP path frozen at the round-5 defaults
    ("ecp_heavy_e6" 10000, "ecp_skip_e12" 100, "ecp_dlog_floor_e6" 30 -- the last one was re-tuned after looking at set B),
    THEN a third synthetic table generated (tests/golden/make_parity_fixtures.py: ecp_table_c -- local exponents 9.1 / 2.2 /
    6.3, an s channel of exponent 0.71 and a p channel of exponent 2.9, other coefficient magnitudes) at 128
    |psi|^2-equilibrated benzene walkers, run ONCE at library defaults and reported as it came out
    (profiles/r06_parity_report.json: ecp_set_c_hold_out_128).  Nothing is tuned on this set; the assertion is a
    sanity band only -- a failure of the 1e-5 tolerance on a hold-out is a finding to report, not a threshold to move
    (reference: ecp/gaussian_type_ecp.py:161-255, one precision and no cut-off there)."""
    d, meta, h, eng = load('benzene_ecpC_psiformer_128')
    assert meta['ecp_table'] == 'C'
    r = torch.as_tensor(d['r'], device=DEV)
    phi = torch.as_tensor(d['ecp_phi'], dtype=torch.float32, device=DEV)
    out = {}
    for key in ('first_call', 'second_call'):
        e, stats = eng.local_energy(r, rng=0, ecp_phi=phi)
        rel, prof = profile(e.double().cpu().numpy(), d['e_loc'])
        v_nl = stats['hamil/V_nl'].double().cpu().numpy()
        prof.update({'n_refined': eng.last_refined(), 'pairs': eng.ecp_counts(), 'refine_info': eng.refine_info(),
                     'n_beyond_1e-5': int((rel >= 1e-5).sum()),
                     'V_nl_abs_err_max': float(np.abs(v_nl - d['stats'][3]).max()), 'V_nl_abs_err_p50': float(np.median(np.abs(v_nl - d['stats'][3]))),
                     'V_nl_abs_mean': float(np.abs(d['stats'][3]).mean()), 'E_abs_max': float(np.abs(d['e_loc']).max())})
        out[key] = prof
    report('ecp_set_c_hold_out_128', out)
    for key in ('first_call', 'second_call'):
        p = out[key]
        assert p['frac_within_1e-5'] >= 0.95 and p['max'] < 1e-4, (key, p)       # sanity band; the recorded numbers are the result
        assert p['V_nl_abs_err_max'] < 1e-6 * np.abs(d['e_loc']).max(), (key, p)


@pytest.mark.parametrize('name', ['benzene_psiformer_256', 'c4h4_transpsiformer_512'])
def test_split_f64_attention_kernel_equals_the_one_wave_kernel_on_device(name):
    """The eight-wave float64 attention kernel of the Laplacian pass (kernel_attention_mfma.hip: k_attention_mfma_split<double>, a PAIR of
    waves per query row block; the default of every float64 pass, i.e. of the whole local energy of the attention ansatzes) against the
    four-wave kernel it replaced (option "attention_split" 0) on the SAME walkers on the device, in a float64 context so that nothing is
    narrowed on the way out: the float32 instance of the split kernel once misbehaved on this hardware while agreeing in the emulation
    (engine_pass.inl, DQMC_OP_ATTENTION), so the float64 instance is pinned here directly -- not only through the oracle fixtures, where
    a 1e-8 defect of the reference energies would pass."""
    d, meta, h, eng = load(name, dtype=torch.float64)
    r = torch.as_tensor(d['r'][:128].astype(np.float64), device=DEV)
    out = {}
    for split in (1, 0):
        eng.set_option('attention_split', split)
        e, stats, grad = eng.local_energy(r, return_grad=True)
        out[split] = (e.cpu().numpy(), grad.cpu().numpy(), stats['hamil/lap'].cpu().numpy())
    ref = d['e_loc'][:128]
    rel = np.abs(out[1][0] - out[0][0]) / np.maximum(1.0, np.abs(out[0][0]))
    rel_o = np.abs(out[1][0] - ref) / np.maximum(1.0, np.abs(ref))
    report(f'{name}_attention_split_vs_one_wave', {'max_rel_between_kernels': float(rel.max()), 'max_rel_to_oracle_split': float(rel_o.max()),
                                                   'identical': bool(np.array_equal(out[1][0], out[0][0]))})
    # (two summation orders of the same float64 arithmetic; the oracle fixture holds float32-rounded geometry: ~5e-8)
    assert rel.max() < 1e-10, rel.max()
    assert not np.array_equal(out[1][0], out[0][0])          # (the switch did select another kernel)
    assert np.abs(out[1][1] - out[0][1]).max() <= 1e-9 * max(1.0, np.abs(out[0][1]).max())
    assert (np.abs(out[1][2] - out[0][2]) / np.maximum(1.0, np.abs(out[0][2]))).max() < 1e-10
    # against the oracle: benzene 5e-12.  (The TransPsiformer folds its nuclear stream on the host at the context's geometry; the float64
    # context built here does not reproduce the mixed float32 / float64 geometry of the fixture -- 3e-5, recorded; its oracle parity is
    # asserted through the float32 context's float64 pass in test_f32_parity_at_baseline_size: 5.8e-8.)
    if not meta['ansatz'] == 'transpsiformer':
        assert rel_o.max() < 1e-9, rel_o.max()


@pytest.mark.parametrize('name,ratio_max', [('lih_paulinet_4096', 0.95), ('n2_ferminet_4096', 0.90)])
def test_float64_tail_lowers_the_float32_error_on_device(name, ratio_max):
    """The float64 tail of a float32 pass (engine.hip above tail_f64: the backflow head, envelopes x backflow, determinants and
    E_loc on the float64 twin for every walker, the float32 head's activations read in place) ON the MI355X, refinement off:
    the mean float32 error of E_loc against the oracle drops (measured: x 0.79 LiH / PauliNet, x 0.81 N2 / FermiNet; x 0.81 and
    x 0.63 before the linear kernels got their fresh per-chunk accumulators -- the head no longer stands out as much); the captured pass (eager first call, capture on the
    second, replays afterwards) stays bit-identical; psi signs equal the oracle's."""
    d, meta, h, eng = load(name)
    r = torch.as_tensor(d['r'], device=DEV)
    eng.set_option('refine', 0)
    mean = {}
    for tail in (1, 0):
        eng.set_option('tail_f64', tail)
        prev = None
        for _ in range(4):          # (outputs released before the next call: the allocator hands the same blocks back, the captured pass replays)
            e, st_ = eng.local_energy(r)
            cur = e.clone()
            del e, st_
            assert prev is None or torch.equal(prev, cur)
            prev = cur
        rel, prof = profile(prev.double().cpu().numpy(), d['e_loc'])
        mean[tail] = float(rel.mean())
        prof['mean'] = mean[tail]
        report(f'f64_tail_{name}_{"on" if tail else "off"}', prof)
    assert mean[1] < ratio_max * mean[0], mean
    eng.set_option('tail_f64', 1)
    sign, logpsi, grad = eng.psi_and_grad(r)
    np.testing.assert_array_equal(sign.cpu().numpy(), d['sign'])
