"""Pin the CPU oracle against the reference's parameter-free golden vectors
(tests/golden/reference_kats.npz, produced by tests/golden/make_golden.py) and check its
three Laplacian evaluations against each other."""
import numpy as np
import pytest
import torch

from deepqmc_amd.hamil import MolecularHamiltonian, get_shell
from deepqmc_amd.molecule import Molecule
from deepqmc_amd.params import init_params
from deepqmc_amd.spec import ferminet, paulinet, psiformer
from oracle import geom, physics
from oracle import wf as owf

T = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)


def test_graph_edge_builder_golden(kats):
    # reference tests/test_gnn.py:18-29
    nodes = T([[0.0, 0.0, 0.0], [0.0, 0.0, 2.0], [0.0, 0.0, 6.0]])
    for ms in (True, False):
        got = geom.compute_edges(nodes, nodes, ms).numpy()
        np.testing.assert_array_equal(got, kats[f'graph_edges_mask_self_{ms}'])


def test_molecular_edges_golden(kats, lih_walker):
    # reference tests/test_gnn.py:30-44
    mol = Molecule.from_name('LiH')
    e = geom.molecular_edges(T(lih_walker), T(mol.coords), 2, ('ne', 'same', 'anti'), False)
    np.testing.assert_allclose(geom.single_array(e['same']).numpy(), kats['lih_edges_same'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(geom.single_array(e['anti']).numpy(), kats['lih_edges_anti'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(e['ne']['ne'].numpy(), kats['lih_edges_ne'], rtol=0, atol=1e-7)


def test_molecule_golden(kats):
    for name in ('LiH', 'H2O'):
        mol = Molecule.from_name(name)
        np.testing.assert_allclose(mol.coords, kats[f'molecule_{name}_coords'], rtol=1e-9)
        np.testing.assert_array_equal(mol.charges, kats[f'molecule_{name}_charges'])
        assert mol.charge == int(kats[f'molecule_{name}_charge']) and mol.spin == int(kats[f'molecule_{name}_spin'])
    with pytest.raises(ValueError):
        Molecule.from_name('unobtainium')


def test_hamil_init_golden(kats):
    # reference tests/test_hamil.py:19-28
    h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
    assert h.n_up == int(kats['hamil_init_Molecular_n_up']) and h.n_down == int(kats['hamil_init_Molecular_n_down'])
    np.testing.assert_array_equal(h.ns_valence, kats['hamil_init_Molecular_ns_valence'])
    np.testing.assert_array_equal(h.ecp_mask, kats['hamil_init_Molecular_pp_mask'])
    assert [get_shell(z) for z in (1, 2, 3, 10, 11)] == [1, 1, 2, 2, 3]
    assert h.mol_ecp_shells == [0, 0]


def test_hamil_init_pp_golden(kats):
    """reference tests/test_hamil.py:19-28 with `ecp_type='bfd'` (test_init_Molecular_PP_.npz): the integer logic of
    hamil.py:119-142 -- default mask = charges > 2, valence counts = Z - n_core, electron / spin counts from them.  The
    bfd table itself lives in pyscf (not available offline); what the golden fixes is independent of its coefficients:
    lithium keeps one valence electron (n_core = 2), hydrogen is untouched."""
    from deepqmc_amd.ecp import GaussianTypeECP
    mol = Molecule.from_name('LiH')
    li = [2, [[-1, [[], [[5.4, 1.0]], [[4.6, -4.6]], [[2.7, 5.4]]]], [0, [[], [], [[1.33, 6.75]]]]]]      # pyscf layout, n_core 2
    h = MolecularHamiltonian(mol=mol, ecp_type='bfd', ecp_tables={'Li': li})
    assert h.n_up == int(kats['hamil_init_Molecular_PP_n_up']) and h.n_down == int(kats['hamil_init_Molecular_PP_n_down'])
    np.testing.assert_array_equal(h.ns_valence, kats['hamil_init_Molecular_PP_ns_valence'])
    np.testing.assert_array_equal(h.ecp_mask, kats['hamil_init_Molecular_PP_pp_mask'])
    assert isinstance(h.pot, GaussianTypeECP) and h.n_elec == 2
    # an explicit mask overrides the default (hamil.py:122-125); no ECP atoms -> bare Coulomb, whatever ecp_type says
    h2 = MolecularHamiltonian(mol=mol, ecp_type='bfd', ecp_mask=[False, False])
    assert h2.pot is None and (h2.n_up, h2.n_down) == (2, 2)
    with pytest.raises(AssertionError):
        MolecularHamiltonian(mol=mol, ecp_type='bfd', ecp_mask=[True])


def test_local_potential_golden(kats, lih_walker):
    # reference tests/test_potential.py (LiH, ecp None): -93.0144804569
    mol = Molecule.from_name('LiH')
    v = physics.local_potential(T(lih_walker), T(mol.coords), T(mol.charges))
    np.testing.assert_allclose(float(v), float(kats['lih_potential_local_potential']), rtol=1e-9)


def test_coulomb_kat():
    # reference tests/test_physics.py:7-17
    R = T([[0.0, 0.0, 0.0], [0.0, 0.0, 1.4]])
    r = T([[0.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    np.testing.assert_allclose(float(physics.nuclear_energy(R, T([1.0, 1.0]))), 1 / 1.4, rtol=1e-12)
    np.testing.assert_allclose(float(physics.electronic_potential(r, geom.F64_EPS)), 1.0, rtol=1e-12)


@pytest.mark.parametrize('spec_fn,molname', [(paulinet, 'LiH'), (ferminet, 'LiH'), (psiformer, 'LiH')])
def test_laplacians_agree(spec_fn, molname, lih_walker):
    """Hessian trace == literal reverse-forward loop (physics.py:144-156) == finite
    differences, at the reference's canonical walker."""
    spec = spec_fn()
    mol = Molecule.from_name(molname)
    h = MolecularHamiltonian(mol=mol)
    params = owf.to_torch(init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=3, perturb_envelopes=0.1))
    R, r = T(mol.coords), T(lih_walker)

    def f(flat):
        return owf.wave_function(params, spec, flat.reshape(-1, 3), R, h.n_up, geom.F32_EPS)[1]

    x = r.reshape(-1)
    l1, g1 = physics.laplacian_hessian(f, x)
    l2, g2 = physics.laplacian_loop(f, x)
    l3, g3 = physics.laplacian_fd(f, x, h=2e-4)
    assert torch.isfinite(l1)
    np.testing.assert_allclose(float(l1), float(l2), rtol=1e-10)
    np.testing.assert_allclose(g1.numpy(), g2.numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(float(l1), float(l3), rtol=2e-4, atol=1e-3)
    np.testing.assert_allclose(g1.numpy(), g3.numpy(), rtol=1e-4, atol=2e-5)  # FD truncation near the nuclear cusp


def test_local_energy_parts(lih_walker):
    spec = paulinet()
    mol = Molecule.from_name('LiH')
    h = MolecularHamiltonian(mol=mol)
    params = owf.to_torch(init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=0))
    e, st, _ = physics.local_energy(params, spec, T(lih_walker), T(mol.coords), T(mol.charges), h.n_up, geom.F32_EPS)
    assert set(st) == {'hamil/V_el', 'hamil/E_kin', 'hamil/V_loc', 'hamil/V_nl', 'hamil/lap', 'hamil/quantum_force'}
    total = st['hamil/E_kin'] + st['hamil/V_loc'] + st['hamil/V_nl'] + st['hamil/V_el'] + 3.0 / 3.01411317
    np.testing.assert_allclose(float(e), float(total), rtol=1e-8)
