"""Pseudo-Hamiltonians (reference ecp/pseudo_hamiltonian.py): properties of the oracle restatement, the XML table
parser, and the HIP path (through the SIMT emulator) against the oracle on the same walkers.  The reference has no
test of this module, so parity is unpinned (oracle/pseudo_hamiltonian.py header); the tables used against the HIP
path are synthetic smooth functions on the reference's kind of grid."""
import os

import numpy as np
import pytest
import torch

from deepqmc_amd import MolecularHamiltonian, Molecule
from deepqmc_amd.ecp import ELEMENTS_WITH_EXISTING_PH, PseudoHamiltonian, parse_ph_xml
from deepqmc_amd.sampling import synthetic_walkers
from deepqmc_amd.wf import NeuralNetworkWaveFunction
from oracle import geom, physics
from oracle import pseudo_hamiltonian as oph
from oracle import wf as owf
from simt_util import emu_lib

REF_PH_DATA = '/root/reference/src/deepqmc/ecp/ph_data'
N_GRID, R_MAX = 2001, 10.0


def synthetic_tables():
    """{element: (rV_loc, rV_L2, n_valence)}: smooth, rV_L2 * r > -1/2 everywhere (A stays positive definite)."""
    x = np.linspace(0.0, R_MAX, N_GRID)
    li = (3.0 * np.exp(-1.5 * x ** 2) - 0.5 * x * np.exp(-x), 0.25 * x * np.exp(-0.8 * x ** 2) * (1 - 0.5 * x), 3.0)
    h = (1.0 * np.exp(-2.0 * x ** 2) + 0.2 * x * np.exp(-1.3 * x), -0.1 * x * np.exp(-0.5 * x ** 2), 1.0)
    return {'Li': li, 'H': h}


def T(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float64)


def test_interpolation_matches_numpy_and_fills_zero():
    x = np.linspace(0.0, R_MAX, N_GRID)
    tab = np.sin(x) * np.exp(-0.1 * x)
    q = np.array([0.0, 1e-4, 0.37, 4.99999, 9.9999, 10.0, 10.0001, 25.0])
    got = oph.interp(T(tab), R_MAX, T(q)).numpy()
    want = np.where(q <= R_MAX, np.interp(q, x, tab), 0.0)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-15)


def _logpsi_toy(R):
    """A smooth, non-separable log|psi| of 3 electrons."""
    def f(flat):
        r = flat.reshape(-1, 3)
        d = (r[:, None, :] - R[None]).norm(dim=-1)
        rij = (r[0] - r[1]).norm() + (r[1] - r[2]).norm() * 0.5
        return -(d.sum(1) * torch.tensor([1.1, 0.7, 0.9], dtype=torch.float64)).sum() + 0.3 * torch.log1p(rij) + 0.1 * r[0, 0] * r[2, 1]
    return f


def test_coordinate_change_equals_direct_operator_and_reduces_to_laplacian():
    tabs = synthetic_tables()
    R = T([[0.0, 0.0, 0.0], [0.0, 0.0, 3.0]])
    r = T([[0.3, -0.2, 0.5], [0.1, 0.4, 2.2], [-0.6, 0.2, 1.1]])
    rv_l2 = T(np.stack([tabs['Li'][1], tabs['H'][1]]))
    f = _logpsi_toy(R)
    for mask in ([True, False], [True, True]):
        e_kin, lap, qf, jac_v = oph.kinetic_term(f, r, R, mask, rv_l2, R_MAX)
        direct = oph.kinetic_term_direct(f, r, R, mask, rv_l2, R_MAX)
        np.testing.assert_allclose(float(e_kin), float(direct), rtol=1e-12)
    # a vanishing L^2 function: A = I/2, b = 0 -> the ordinary -1/2 (lap + |grad|^2), physics.py:108
    e_kin, lap, qf, _ = oph.kinetic_term(f, r, R, [True, True], torch.zeros_like(rv_l2), R_MAX)
    lap0, g0 = physics.laplacian_hessian(f, r.reshape(-1))
    np.testing.assert_allclose(float(e_kin), float(-0.5 * (lap0 + (g0 ** 2).sum())), rtol=1e-12)
    np.testing.assert_allclose(float(lap), 0.5 * float(lap0), rtol=1e-12)


def test_table_container_and_reference_lookup_rules():
    tabs = synthetic_tables()
    ph = PseudoHamiltonian.from_tables(np.array([3, 1]), [True, False], tabs, r_max=R_MAX)
    assert ph.rv_loc.shape == (2, N_GRID) and not ph.rv_loc[1].any() and ph.ns_valence.tolist() == [3.0, 1.0]
    assert ELEMENTS_WITH_EXISTING_PH[17] == ('Cl', 'cc') and ELEMENTS_WITH_EXISTING_PH[29] == ('Cu', 'hf')
    with pytest.raises(AssertionError, match='not found'):
        PseudoHamiltonian.from_xml_dir(np.array([3, 1]), 'PHcc', [True, False], '/nonexistent')
    with pytest.raises(RuntimeError, match='ph_tables'):
        MolecularHamiltonian(mol=Molecule.from_name('LiH'), ecp_type='PHcc')


@pytest.mark.skipif(not os.path.isdir(REF_PH_DATA), reason='reference pseudo-Hamiltonian tables not present')
def test_reference_xml_tables_parse_and_satisfy_the_ph_relation():
    """pseudo_hamiltonian.py:59-62: the files are built such that 2 (s - d) = 3 (p - d); the parser relies on it."""
    from xml.etree import ElementTree
    for z, (name, suffix) in ELEMENTS_WITH_EXISTING_PH.items():
        path = os.path.join(REF_PH_DATA, f'{name}.{suffix}.xml')
        if not os.path.exists(path):
            continue
        loc, l2, zval = parse_ph_xml(path)
        assert loc.shape == l2.shape == (10001,) and zval > 0
        root = ElementTree.parse(path).getroot()
        chan = {v.attrib['l']: np.array(v.find('radfunc').find('data').text.split(), float) for v in root.find('semilocal').findall('vps')}
        v0, v1 = chan['s'] - chan['d'], chan['p'] - chan['d']
        np.testing.assert_allclose(2 * v0, 3 * v1, atol=2e-6 * max(1.0, np.abs(v0).max()))
        np.testing.assert_allclose(loc, chan['s'] + zval)           # local_nl + v0_nl + n_valence = s + zval
        assert (l2 * np.linspace(0, 10, 10001)).min() > -0.5        # A = 1/2 + rV_L2 r (perpendicular part) stays positive
        assert abs(loc[-1]) < 1e-6 and abs(l2[-1]) < 1e-8           # the tables decay to the bare -Z_eff/r tail


def _hip_vs_oracle(ansatz, mask, dtype=torch.float64, B=2, refine=None, tol=1e-8):
    mol = Molecule.from_name('LiH')
    h = MolecularHamiltonian(mol=mol, ecp_type='PH', ecp_mask=mask, ph_tables=synthetic_tables())
    h.pot.r_max = R_MAX
    wf = NeuralNetworkWaveFunction(h, ansatz, dtype=dtype, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    params = wf.init(0, perturb_envelopes=0.1)
    eng = wf.engine(params)
    if refine is not None:
        eng.set_option('refine', refine)
    r = synthetic_walkers(h, B, seed=7).astype(np.float32 if dtype == torch.float32 else np.float64)
    e, stats, grad = eng.local_energy(torch.as_tensor(r), return_grad=True)
    R = T(mol.coords.astype(np.float32)) if dtype == torch.float32 else T(mol.coords)
    p = owf.to_torch(params)
    for b in range(B):
        e_ref, st, jac_v = oph.local_energy(p, wf.spec, T(r[b]), R, T(h.ns_valence), h.n_up, mask, T(h.pot.rv_loc), T(h.pot.rv_l2),
                                            R_MAX, geom.F32_EPS)
        np.testing.assert_allclose(float(e[b]), float(e_ref), rtol=tol, atol=tol)
        for key in ('hamil/V_el', 'hamil/E_kin', 'hamil/V_loc', 'hamil/lap', 'hamil/quantum_force'):
            np.testing.assert_allclose(float(stats[key][b]), float(st[key]), rtol=tol, atol=tol, err_msg=key)
        np.testing.assert_allclose(grad[b].double().numpy().reshape(-1, 3), jac_v.detach().numpy(), rtol=tol, atol=tol)
    # the plain gradient entry point is untouched by the pseudo-Hamiltonian
    sign, logpsi, g = eng.psi_and_grad(torch.as_tensor(r))
    _, _, qf = physics.batch_local_energy(p, wf.spec, T(r), R, T(h.ns_valence), h.n_up, geom.F32_EPS)
    np.testing.assert_allclose(g.double().numpy().reshape(B, -1), qf.numpy(), rtol=max(tol, 1e-9), atol=max(tol, 1e-9))
    return eng


@pytest.mark.parametrize('ansatz,mask', [('paulinet', [True, False]), ('paulinet', [True, True]), ('ferminet', [True, True]),
                                         ('psiformer', [True, False])])
def test_hip_pseudo_hamiltonian_matches_oracle_f64(ansatz, mask):
    """Pair features (plain and log-rescaled), envelopes and cusps seeded with the Cholesky factors; k_final's first-order
    term and local PH potential."""
    _hip_vs_oracle(ansatz, mask)


def test_hip_pseudo_hamiltonian_f32_and_refinement_twin():
    """float32 build: plain float32 to float32 accuracy; with the E_loc pass handed to the float64 twin (refine = 2)
    the twin must carry the same tables and agree to output rounding."""
    _hip_vs_oracle('paulinet', [True, True], dtype=torch.float32, refine=0, tol=2e-4)
    _hip_vs_oracle('paulinet', [True, True], dtype=torch.float32, refine=2, tol=2e-6)


@pytest.mark.skipif(not os.path.isdir(REF_PH_DATA), reason='reference pseudo-Hamiltonian tables not present')
def test_reference_tables_through_the_facade(monkeypatch):
    """`ecp_type='PHcc'` with the reference's own XML directory (environment variable, as a maintainer without the
    deepqmc package on the path would set it): HCl keeps 7 + 1 valence electrons, the tables land on the reference's grid."""
    monkeypatch.setenv('DEEPQMC_PH_DATA', REF_PH_DATA)
    mol = Molecule(coords=np.array([[0.0, 0.0, 0.0], [0.0, 0.0, 2.4]]), charges=np.array([17, 1]), charge=0, spin=0)
    h = MolecularHamiltonian(mol=mol, ecp_type='PHcc')
    assert h.ecp_mask.tolist() == [True, False] and h.ns_valence.tolist() == [7.0, 1.0] and (h.n_up, h.n_down) == (4, 4)
    assert h.pot.rv_loc.shape == (2, 10001) and h.pot.r_max == 10.0 and not h.pot.rv_l2[1].any()


@pytest.mark.skipif(not os.path.isdir(REF_PH_DATA), reason='reference pseudo-Hamiltonian tables not present')
def test_hip_hcl_with_the_reference_chlorine_table(monkeypatch):
    """End to end on a molecule the reference's tables cover: HCl with the OPH23 chlorine pseudo-Hamiltonian (7 valence
    electrons on Cl), 8 electrons, float64, HIP path (emulated) against the oracle restatement on the reference's grid."""
    monkeypatch.setenv('DEEPQMC_PH_DATA', REF_PH_DATA)
    mol = Molecule(coords=np.array([[0.0, 0.0, 0.0], [0.0, 0.0, 2.4]]), charges=np.array([17, 1]), charge=0, spin=0)
    h = MolecularHamiltonian(mol=mol, ecp_type='PHcc')
    wf = NeuralNetworkWaveFunction(h, 'paulinet', dtype=torch.float64, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    params = wf.init(0, perturb_envelopes=0.1)
    r = synthetic_walkers(h, 1, seed=3)
    e, stats = wf.engine(params).local_energy(torch.as_tensor(r))
    e_ref, st, _ = oph.local_energy(owf.to_torch(params), wf.spec, T(r[0]), T(mol.coords), T(h.ns_valence), h.n_up, [True, False],
                                    T(h.pot.rv_loc), T(h.pot.rv_l2), h.pot.r_max, geom.F32_EPS)
    np.testing.assert_allclose(float(e[0]), float(e_ref), rtol=1e-8)
    for key in ('hamil/E_kin', 'hamil/V_loc', 'hamil/lap', 'hamil/quantum_force'):
        np.testing.assert_allclose(float(stats[key][0]), float(st[key]), rtol=1e-8, atol=1e-8, err_msg=key)
