"""The layer-program compiler + forward-Laplacian rules (NumPy interpreter) against the
autograd oracle: same walkers, same parameters, float64."""
import numpy as np
import pytest
import torch

from deepqmc_amd.hamil import MolecularHamiltonian
from deepqmc_amd.molecule import Molecule
from deepqmc_amd.params import init_params
from deepqmc_amd.program import compile_program
from deepqmc_amd.spec import ferminet, paulinet, psiformer, transpsiformer
from oracle import geom, physics
from oracle import wf as owf
from oracle.program_interp import Interp

T = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)


def make_walkers(mol, n_elec, B, seed=1):
    """Synthetic walkers: electrons placed on nuclei in proportion to charge + N(0,1)."""
    rng = np.random.default_rng(seed)
    centers = np.repeat(np.arange(len(mol.charges)), mol.charges.astype(int))[:n_elec]
    return mol.coords[centers][None] + rng.standard_normal((B, n_elec, 3))


@pytest.mark.parametrize('spec_fn,molname', [(paulinet, 'LiH'), (ferminet, 'LiH'), (psiformer, 'LiH'), (paulinet, 'Be'),
                                             (transpsiformer, 'LiH')])
def test_interp_matches_autograd(spec_fn, molname, lih_walker):
    mol = Molecule.from_name(molname)
    spec = spec_fn(mol.charges) if spec_fn is transpsiformer else spec_fn()
    h = MolecularHamiltonian(mol=mol)
    tree = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=5, perturb_envelopes=0.1)
    # transpsiformer: the nuclear stream is folded at compile time (deepqmc_amd/nuclear_stream.py); the oracle
    # evaluates the full masked attention over [nuclei; electrons] instead
    prog = compile_program(spec, tree, h.n_up, h.n_down, h.n_nuc, R=mol.coords, eps=geom.F32_EPS)
    r = make_walkers(mol, h.n_elec, 3)
    if molname == 'LiH':
        r[0] = lih_walker
    eps = geom.F32_EPS
    it = Interp(prog, mol.charges, eps)
    val = it.run(r, mol.coords, laplacian=False)
    lap = it.run(r, mol.coords, laplacian=True)
    params = owf.to_torch(tree)
    e_ref, st_ref, qf_ref = physics.batch_local_energy(params, spec, T(r), T(mol.coords), T(mol.charges), h.n_up, eps)
    s_ref, l_ref = physics.batch_wave_function(params, spec, T(r), T(mol.coords), h.n_up, eps)
    np.testing.assert_array_equal(val['sign'], s_ref.numpy().astype(np.int32))
    tol = 1e-10 if spec.nuclei_tokens else 1e-12      # folded nuclear stream: different summation order
    np.testing.assert_allclose(val['log'], l_ref.numpy(), rtol=tol, atol=tol)
    np.testing.assert_allclose(lap['log'], l_ref.numpy(), rtol=tol, atol=tol)
    np.testing.assert_allclose(lap['grad'], qf_ref.numpy(), rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(lap['e_loc'], e_ref.numpy(), rtol=1e-9, atol=1e-9)
    order = ['hamil/V_el', 'hamil/E_kin', 'hamil/V_loc', 'hamil/V_nl', 'hamil/lap', 'hamil/quantum_force']
    for k, name in enumerate(order):
        np.testing.assert_allclose(lap['stats'][k], st_ref[name].numpy(), rtol=1e-9, atol=1e-9)


def test_flops_count():
    # F_lin(LiH, PauliNet) ~= 1.58 MFLOP per walker
    mol = Molecule.from_name('LiH')
    h = MolecularHamiltonian(mol=mol)
    spec = paulinet()
    prog = compile_program(spec, init_params(spec, 2, 2, 2), 2, 2, 2)
    assert abs(prog.flops_per_walker - 1.58e6) / 1.58e6 < 0.02
