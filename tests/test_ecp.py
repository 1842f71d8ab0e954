"""Gaussian-type ECPs (reference ecp/gaussian_type_ecp.py, ecp/ecp_utils.py): quadrature properties of the
oracle restatement, the pyscf-format table parser, and the HIP path (through the SIMT emulator, float64)
against the oracle on the same walkers and the same rotation angles.  The coefficient tables are synthetic
(pyscf's are not available offline -- oracle/ecp.py header)."""
import math

import numpy as np
import pytest
import torch

from deepqmc_amd import MolecularHamiltonian, Molecule
from deepqmc_amd.ecp import GaussianTypeECP
from deepqmc_amd.sampling import synthetic_walkers
from deepqmc_amd.wf import NeuralNetworkWaveFunction
from oracle import ecp as oecp
from oracle import geom, physics
from oracle import wf as owf
from simt_util import emu_lib

# pyscf ECP format: [n_core, [[-1, [r^-2, r^-1, r^0, r^1 terms]], [0, [.., .., r^0 terms]], [1, ...]]]
TABLES = {
    'Li': [2, [[-1, [[], [[5.4104, 1.0]], [[4.6015, -4.6015]], [[2.7052, 5.4104]]]],
               [0, [[], [], [[1.3302, 6.7529], [0.9, -0.8]]]],
               [1, [[], [], [[1.25, 0.45]]]]]],
    'H': [0, [[-1, [[], [[21.24, 1.0]], [[21.78, -10.85]], [[21.24, 21.24]]]],
              [0, [[], [], [[1.0, 0.0]]]]]],
}


def test_icosahedron_quadrature():
    pts, th = oecp.unit_icosahedron()
    assert pts.shape == (12, 3)
    np.testing.assert_allclose(np.linalg.norm(pts, axis=-1), 1.0, atol=1e-15)
    # exact for spherical polynomials up to degree 5: <z^2> = 1/3, <x^4> = 1/5, <x^2 y^2> = 1/15, odd -> 0
    np.testing.assert_allclose((pts[:, 2] ** 2).mean(), 1 / 3, atol=1e-14)
    np.testing.assert_allclose((pts[:, 0] ** 4).mean(), 1 / 5, atol=1e-14)
    np.testing.assert_allclose((pts[:, 0] ** 2 * pts[:, 1] ** 2).mean(), 1 / 15, atol=1e-14)
    np.testing.assert_allclose((pts[:, 0] ** 3 * pts[:, 2] ** 2).mean(), 0.0, atol=1e-14)
    # Legendre orthogonality under the rule (what makes the l-projection work)
    P = oecp.legendre_table(3, th)
    gram = P.T @ P / 12
    np.testing.assert_allclose(gram, np.diag([1, 1 / 3, 1 / 5]), atol=1e-14)
    # the first vertex is the electron itself, every vertex keeps the electron-nucleus distance
    r_i, R_a = torch.tensor([0.3, -1.1, 0.7], dtype=torch.float64), torch.tensor([0.1, 0.2, -0.4], dtype=torch.float64)
    q = oecp.quadrature_points(r_i, R_a, torch.tensor(0.37, dtype=torch.float64), torch.as_tensor(pts))
    np.testing.assert_allclose(q[0].numpy(), r_i.numpy(), atol=1e-14)
    np.testing.assert_allclose(torch.linalg.norm(q - R_a, dim=-1).numpy(), float(torch.linalg.norm(r_i - R_a)), rtol=1e-14)
    np.testing.assert_allclose(q[1].numpy(), (2 * R_a - r_i).numpy(), atol=1e-14)     # antipode


def test_table_parser_and_valence_counts():
    mol = Molecule.from_name('LiH')
    pot = GaussianTypeECP.from_tables(mol.charges, [True, False], TABLES)
    np.testing.assert_array_equal(pot.ns_valence, [1.0, 1.0])
    assert pot.loc_params.shape == (2, 3, 2, 1) and pot.nl_params.shape == (2, 2, 2, 2)
    np.testing.assert_allclose(pot.loc_params[0, :, 0, 0], [5.4104, 4.6015, 2.7052])     # exponents r^-1, r^0, r^1
    np.testing.assert_allclose(pot.loc_params[0, :, 1, 0], [1.0, -4.6015, 5.4104])
    np.testing.assert_allclose(pot.nl_params[0, 0], [[1.3302, 0.9], [6.7529, -0.8]])
    np.testing.assert_allclose(pot.nl_params[0, 1], [[1.25, 0.0], [0.45, 0.0]])
    assert not pot.loc_params[1].any() and not pot.nl_params[1].any()
    np.testing.assert_array_equal(pot.nuc_with_nl_pot, [0])
    h = MolecularHamiltonian(mol=mol, ecp_type='synthetic', ecp_tables=TABLES)        # default mask: charges > 2
    assert (h.n_up, h.n_down) == (1, 1) and list(h.ecp_mask) == [True, False]           # tests/test_hamil PP golden: 1, 1
    assert h.mol_ecp_shells == [1, 0]


def test_s_wave_sees_only_l0():
    """psi depending on |r_i - R_a| only: every quadrature point has ratio 1, so V_nl = sum_i V_0(r_i)."""
    R = torch.tensor([[0.0, 0.0, 0.0]], dtype=torch.float64)
    r = torch.tensor([[0.4, 0.1, -0.3], [-0.2, 0.9, 0.5]], dtype=torch.float64)
    nl = torch.tensor([[[[1.3, 0.9], [2.0, -0.8]], [[1.25, 0.0], [0.45, 0.0]]]], dtype=torch.float64)
    psi = lambda rr: (torch.ones(rr.shape[0], dtype=torch.float64), -(rr.norm(dim=-1) ** 2).sum(-1))
    v = oecp.nonloc_potential(r, R, nl, psi, torch.tensor([[0.1, 0.5]], dtype=torch.float64))
    d2 = (r ** 2).sum(-1)
    expect = (2.0 * torch.exp(-1.3 * d2) - 0.8 * torch.exp(-0.9 * d2)).sum()
    np.testing.assert_allclose(float(v), float(expect), rtol=1e-13)


def _hip_vs_oracle(mask, tables, B=2, seed=5):
    mol = Molecule.from_name('LiH')
    h = MolecularHamiltonian(mol=mol, ecp_type='synthetic', ecp_mask=mask, ecp_tables=tables)
    wf = NeuralNetworkWaveFunction(h, 'paulinet', dtype=torch.float64, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    params = wf.init(0, perturb_envelopes=0.1)
    N = h.n_elec
    r = torch.as_tensor(synthetic_walkers(h, B, seed=seed))
    n_nl = len(h.pot.nuc_with_nl_pot)
    phi = torch.as_tensor(np.random.default_rng(1).uniform(0, math.pi / 5, (B, n_nl, N)))
    e, stats = wf.engine(params).local_energy(r, ecp_phi=phi)
    T = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
    R, p = T(mol.coords), owf.to_torch(params)
    loc, nl, zv = T(h.pot.loc_params), T(h.pot.nl_params), T(h.ns_valence)
    e_ref, st_ref, _ = physics.batch_local_energy(p, wf.spec, r, R, zv, h.n_up, geom.F32_EPS)
    psi = lambda rr: physics.batch_wave_function(p, wf.spec, rr, R, h.n_up, geom.F32_EPS)
    for b in range(B):
        v_loc = oecp.local_potential(r[b], R, zv, loc, h.ecp_mask)
        v_nl = oecp.nonloc_potential(r[b], R, nl, psi, phi[b])
        np.testing.assert_allclose(float(stats['hamil/V_loc'][b]), float(v_loc), rtol=1e-11)
        np.testing.assert_allclose(float(stats['hamil/V_nl'][b]), float(v_nl), rtol=1e-8, atol=1e-10)
        e_b = float(e_ref[b]) - float(st_ref['hamil/V_loc'][b]) + float(v_loc) + float(v_nl)
        np.testing.assert_allclose(float(e[b]), e_b, rtol=1e-8, atol=1e-8)
    return h, stats


def test_hip_ecp_matches_oracle_valence_only():
    """Li core removed (n_core = 2): 2 valence electrons, non-local s and p channels on Li."""
    h, stats = _hip_vs_oracle([True, False], TABLES)
    assert h.n_elec == 2 and float(stats['hamil/V_nl'].abs().max()) > 1e-6


def test_hip_ecp_matches_oracle_two_centres():
    """ECP tables on both nuclei (Li with n_core = 0 keeps 4 electrons; H local-only plus a zero s channel):
    exercises the compaction to nuclei with a non-local part and the per-nucleus loop."""
    t = dict(TABLES)
    t['Li'] = [0, TABLES['Li'][1]]
    t['H'] = [0, [TABLES['H'][1][0], [0, [[], [], [[0.7, 0.3]]]]]]
    h, stats = _hip_vs_oracle([True, True], t, B=1)
    assert h.n_elec == 4 and len(h.pot.nuc_with_nl_pot) == 2


def test_mixed_precision_quadrature_classes():
    """float32 context, library defaults ("refine" 1, "ecp_mixed" 1): the non-local term of every walker is assembled from
    psi ratios whose precision follows the radial weight w = max_l (2l+1)|V_l(r_ia)| of the (nucleus, electron) pair
    (kernels_ecp.hip).  Through the emulator: all pairs float64 / all float32 / the default split, against the float64
    engine on the same walkers and angles; the classes partition the pairs; repeated calls are bit-identical."""
    mol = Molecule.from_name('LiH')
    t = dict(TABLES)
    t['Li'] = [0, TABLES['Li'][1]]                          # 4 electrons, non-local channels on Li
    h = MolecularHamiltonian(mol=mol, ecp_type='synthetic', ecp_mask=[True, False], ecp_tables=t)
    import dataclasses
    from deepqmc_amd.spec import paulinet
    small = dataclasses.replace(paulinet(), embedding_dim=32, n_interactions=1, n_determinants=4)      # (keeps the emulation short)
    mk = lambda dt: NeuralNetworkWaveFunction(h, small, dtype=dt, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    wf32, wf64 = mk(torch.float32), mk(torch.float64)
    params = wf32.init(0, perturb_envelopes=0.1)
    B, N = 3, h.n_elec
    r = synthetic_walkers(h, B, seed=5).astype(np.float32)
    r[0, 0] = mol.coords[0] + np.array([6.5, 0.3, -0.2])     # one electron far from the Li core: a negligible pair
    n_nl = len(h.pot.nuc_with_nl_pot)
    phi = np.random.default_rng(1).uniform(0, math.pi / 5, (B, n_nl, N)).astype(np.float32)
    # reference: the float64 engine at the float32-rounded geometry a float32 context sees
    from deepqmc_amd.types import PhysicalConfiguration
    R32 = torch.as_tensor(mol.coords.astype(np.float32).astype(np.float64))
    e64, st64 = wf64.engine(params, R32).local_energy(PhysicalConfiguration(R32, torch.as_tensor(r.astype(np.float64)), None),
                                                      ecp_phi=torch.as_tensor(phi.astype(np.float64)))
    v64 = st64['hamil/V_nl'].numpy()
    eng = wf32.engine(params)
    n_pairs = B * n_nl * N
    out = {}
    for name, opts in (('all_f64', {'ecp_heavy_e6': 0, 'ecp_skip_e12': 0}), ('all_f32', {'ecp_heavy_e6': 2_000_000_000, 'ecp_skip_e12': 0}),
                       ('default', {'ecp_heavy_e6': 10_000, 'ecp_skip_e12': 100})):
        for k, v in opts.items():
            eng.set_option(k, v)
        e, st = eng.local_energy(torch.as_tensor(r), ecp_phi=torch.as_tensor(phi))
        c = eng.ecp_counts()
        assert c['f32'] + c['f64'] + c['dropped'] == n_pairs, c
        out[name] = (st['hamil/V_nl'].numpy().astype(np.float64), c, e)
    e2, st2 = eng.local_energy(torch.as_tensor(r), ecp_phi=torch.as_tensor(phi))
    assert torch.equal(out['default'][2], e2) and np.array_equal(out['default'][0], st2['hamil/V_nl'].numpy().astype(np.float64))
    assert out['all_f64'][1]['f64'] == n_pairs and out['all_f32'][1]['f32'] == n_pairs
    d = out['default'][1]
    assert d['f64'] > 0 and d['dropped'] >= 1, d            # the far electron's pair is dropped, the core-near ones are float64
    np.testing.assert_allclose(out['all_f64'][0], v64, rtol=2e-6, atol=2e-7)          # float32 output rounding of float64 ratios
    np.testing.assert_allclose(out['all_f32'][0], v64, rtol=2e-3, atol=2e-4)          # the float32 value path's ratios
    np.testing.assert_allclose(out['default'][0], v64, rtol=2e-3, atol=2e-4)
    # what the classes are for: the default is at least as close to float64 as all-float32 on the batch
    assert np.abs(out['default'][0] - v64).max() <= np.abs(out['all_f32'][0] - v64).max() + 1e-7


def test_near_node_walker_and_no_twin_fallback():
    """Two advisor findings of round 4 on the mixed-precision quadrature (engine_ecp.inl: ecp_mixed):
      (1) near a node of psi the float32 psi(r) in the denominator of EVERY ratio of a walker is what float32 cannot resolve,
          and the ratios themselves grow like 1 / psi(r) -- pairs classified by their radial weight alone then carry a large
          error.  Since round 5 both value paths evaluate the walkers themselves first and a walker whose float32 log|psi| is
          off sends its kept pairs to float64 ("ecp_dlog_floor_e6").  A walker is driven onto a node here (bisection between two
          walkers of opposite sign), and V_nl with the rule must be as close to float64 as an all-float64 quadrature, while the
          weights-only rule of round 4 is recorded to be worse (or no better).
      (2) a context without a float64 twin ("no_twin": what a program without a float64 kernel set looks like) evaluates the
          non-local term in float32 instead of failing, and every later call works too."""
    import dataclasses
    from deepqmc_amd.spec import paulinet
    from deepqmc_amd.types import PhysicalConfiguration
    mol = Molecule.from_name('LiH')
    t = dict(TABLES)
    t['Li'] = [0, TABLES['Li'][1]]
    h = MolecularHamiltonian(mol=mol, ecp_type='synthetic', ecp_mask=[True, False], ecp_tables=t)
    small = dataclasses.replace(paulinet(), embedding_dim=32, n_interactions=1, n_determinants=4)
    mk = lambda dt: NeuralNetworkWaveFunction(h, small, dtype=dt, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    wf32, wf64 = mk(torch.float32), mk(torch.float64)
    params = wf32.init(0, perturb_envelopes=0.1)
    N = h.n_elec
    R32 = torch.as_tensor(mol.coords.astype(np.float32).astype(np.float64))
    e64 = wf64.engine(params, R32)
    cand = synthetic_walkers(h, 64, seed=7)
    sg, lg = e64.wf_eval(torch.as_tensor(cand), R32)
    sg = sg.numpy()
    ia, ib = int(np.argmax(sg > 0)), int(np.argmax(sg < 0))
    assert sg[ia] > 0 > sg[ib], 'need walkers of both signs'
    lo, hi = cand[ia].copy(), cand[ib].copy()
    for _ in range(40):                                       # bisection onto the node between them
        mid = 0.5 * (lo + hi)
        s_mid = int(e64.wf_eval(torch.as_tensor(mid[None]), R32)[0][0])
        if s_mid > 0:
            lo = mid
        else:
            hi = mid
    near = (0.5 * (lo + hi)).astype(np.float32)               # float32 rounding leaves |psi| ~ 1e-6 of its usual size
    r = np.stack([near, cand[2].astype(np.float32), cand[3].astype(np.float32)])
    B = r.shape[0]
    n_nl = len(h.pot.nuc_with_nl_pot)
    phi = np.random.default_rng(3).uniform(0, math.pi / 5, (B, n_nl, N)).astype(np.float32)
    lp64 = e64.wf_eval(torch.as_tensor(r.astype(np.float64)), R32)[1].numpy()
    assert lp64[0] < lp64[1:].min() - 8.0, lp64               # walker 0 really sits on a node
    _, st64 = e64.local_energy(PhysicalConfiguration(R32, torch.as_tensor(r.astype(np.float64)), None), ecp_phi=torch.as_tensor(phi.astype(np.float64)))
    v64 = st64['hamil/V_nl'].numpy()
    eng = wf32.engine(params)
    eng.set_option('refine', 1); eng.set_option('refine_probe', 0); eng.set_option('refine_thresh', 10 ** 9)      # kinetic part plain float32: V_nl alone is looked at
    res = {}
    for name, opts in (('rule', {'ecp_dlog_floor_e6': 30}), ('weights_only', {'ecp_dlog_floor_e6': 0}), ('all_f64', {'ecp_dlog_floor_e6': 0, 'ecp_heavy_e6': 0, 'ecp_skip_e12': 0})):
        for k, v in opts.items():
            eng.set_option(k, v)
        _, st = eng.local_energy(torch.as_tensor(r), ecp_phi=torch.as_tensor(phi))
        res[name] = (st['hamil/V_nl'].numpy().astype(np.float64), eng.ecp_counts())
    err = {k: np.abs(v[0] - v64) / np.maximum(1.0, np.abs(v64)) for k, v in res.items()}
    assert res['rule'][1]['f64'] > res['weights_only'][1]['f64'], (res['rule'][1], res['weights_only'][1])     # the near-node walker's light pairs went to float64
    assert err['rule'][0] <= err['weights_only'][0] + 1e-7, err
    assert err['rule'][0] <= 10 * err['all_f64'][0] + 1e-6, err            # as good as the all-float64 quadrature on the near-node walker
    # (2) no float64 twin: float32 quadrature, no failure, on the first and on later calls
    lone = NeuralNetworkWaveFunction(h, small, dtype=torch.float32, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS).engine(params)
    lone.set_option('no_twin', 1)
    lone.set_option('refine_thresh', 1)                       # every walker would be flagged: the twin is asked for on the first call
    e_a, st_a = lone.local_energy(torch.as_tensor(r), ecp_phi=torch.as_tensor(phi))
    e_b, st_b = lone.local_energy(torch.as_tensor(r), ecp_phi=torch.as_tensor(phi))
    c = lone.ecp_counts() if lone.refine_info()['mode'] == 1 else None
    assert torch.isfinite(e_a[1:]).all() and lone.last_refined() == 0 and lone.refine_info()['mode'] == 0      # refinement switched itself off
    plain = NeuralNetworkWaveFunction(h, small, dtype=torch.float32, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS).engine(params)
    plain.set_option('refine', 0); plain.set_option('tail_f64', 0)
    e_p, st_p = plain.local_energy(torch.as_tensor(r), ecp_phi=torch.as_tensor(phi))
    np.testing.assert_allclose(st_a['hamil/V_nl'].numpy()[1:], st_p['hamil/V_nl'].numpy()[1:], rtol=1e-5, atol=1e-6)   # float32 ratios, only the negligible pairs dropped
    np.testing.assert_array_equal(e_b.numpy()[1:], e_p.numpy()[1:])         # later calls: the plain float32 path (refine 0)
    with pytest.raises(Exception):
        eng.set_option('no_twin', 1)                          # (too late once the twin exists)
