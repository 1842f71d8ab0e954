"""Pseudo-Hamiltonian on the device (through the C ABI): E_loc, its parts and the transformed gradient against the oracle
restatement (oracle/pseudo_hamiltonian.py) on synthetic radial tables; float64 to round-off, float32 to float32 accuracy,
and the float64 refinement twin carrying the tables."""
import numpy as np
import pytest
import torch

from deepqmc_amd import MolecularHamiltonian, Molecule
from deepqmc_amd.sampling import synthetic_walkers
from deepqmc_amd.wf import NeuralNetworkWaveFunction
from oracle import geom
from oracle import pseudo_hamiltonian as oph
from oracle import wf as owf

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
N_GRID, R_MAX = 2001, 10.0


def tables():
    x = np.linspace(0.0, R_MAX, N_GRID)
    li = (3.0 * np.exp(-1.5 * x ** 2) - 0.5 * x * np.exp(-x), 0.25 * x * np.exp(-0.8 * x ** 2) * (1 - 0.5 * x), 3.0)
    h = (1.0 * np.exp(-2.0 * x ** 2) + 0.2 * x * np.exp(-1.3 * x), -0.1 * x * np.exp(-0.5 * x ** 2), 1.0)
    n = (7.0 * np.exp(-1.1 * x ** 2) - 0.8 * x * np.exp(-0.9 * x), 0.2 * x * np.exp(-0.6 * x ** 2), 7.0)
    return {'Li': li, 'H': h, 'N': n}


def T(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float64)


@pytest.mark.parametrize('molname,ansatz,mask,dtype,refine,B,tol', [
    ('LiH', 'paulinet', [True, True], torch.float64, None, 4, 1e-9),
    ('LiH', 'psiformer', [True, False], torch.float64, None, 3, 1e-9),
    ('N2', 'ferminet', [True, True], torch.float64, None, 2, 1e-8),
    ('LiH', 'paulinet', [True, True], torch.float32, 0, 4, 3e-4),
    ('LiH', 'paulinet', [True, True], torch.float32, 2, 4, 2e-6),
])
def test_pseudo_hamiltonian_local_energy(molname, ansatz, mask, dtype, refine, B, tol):
    mol = Molecule.from_name(molname)
    h = MolecularHamiltonian(mol=mol, ecp_type='PH', ecp_mask=mask, ph_tables=tables())
    wf = NeuralNetworkWaveFunction(h, ansatz, dtype=dtype, device=DEV, norm_eps=geom.F32_EPS)
    params = wf.init(0, perturb_envelopes=0.1)
    eng = wf.engine(params)
    if refine is not None:
        eng.set_option('refine', refine)
    r = synthetic_walkers(h, B, seed=7).astype(np.float32 if dtype == torch.float32 else np.float64)
    e, stats, grad = eng.local_energy(torch.as_tensor(r, device=DEV), return_grad=True)
    R = T(mol.coords.astype(np.float32)) if dtype == torch.float32 else T(mol.coords)
    p = owf.to_torch(params)
    for b in range(B):
        e_ref, st, jac_v = oph.local_energy(p, wf.spec, T(r[b]), R, T(h.ns_valence), h.n_up, mask, T(h.pot.rv_loc), T(h.pot.rv_l2),
                                            R_MAX, geom.F32_EPS)
        scale = max(1.0, abs(float(e_ref)))
        assert abs(float(e[b]) - float(e_ref)) < tol * scale, (b, float(e[b]), float(e_ref))
        for key in ('hamil/V_el', 'hamil/E_kin', 'hamil/V_loc', 'hamil/lap', 'hamil/quantum_force'):
            assert abs(float(stats[key][b]) - float(st[key])) < tol * max(1.0, abs(float(st[key]))), key
        np.testing.assert_allclose(grad[b].double().cpu().numpy().reshape(-1, 3), jac_v.detach().numpy(), rtol=10 * tol, atol=10 * tol)
    # switching the tables off restores the ordinary Hamiltonian (V_loc = bare Coulomb of the valence charges)
    eng._check(eng.lib.dqmc_set_pseudo_hamiltonian(eng._ctx, 0, 0.0, None, None, None))
    e0, st0 = eng.local_energy(torch.as_tensor(r, device=DEV))
    d = torch.as_tensor(r[:, :, None, :] - mol.coords[None, None]).norm(dim=-1)
    v_bare = -(torch.as_tensor(h.ns_valence) / d).sum((-1, -2))
    np.testing.assert_allclose(st0['hamil/V_loc'].double().cpu().numpy(), v_bare.numpy(), rtol=1e-5)
