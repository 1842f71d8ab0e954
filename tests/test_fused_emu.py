"""The fused LDS-resident psi kernel (kernel_fused2.hip, descriptor driven) against the one-launch-per-op path
and the oracle, through the SIMT emulator: same program, same weights, ragged last tile."""
import numpy as np
import pytest
import torch

from deepqmc_amd.engine import Engine
from deepqmc_amd.hamil import MolecularHamiltonian
from deepqmc_amd.molecule import Molecule
from deepqmc_amd.params import init_params
from deepqmc_amd.spec import ferminet, paulinet
from oracle import geom
from oracle.program_interp import Interp
from simt_util import emu_lib
from test_program_interp import make_walkers


@pytest.mark.parametrize('spec_fn,dtype,wt', [(paulinet, torch.float64, 0), (paulinet, torch.float32, 4), (ferminet, torch.float64, 2)])
def test_fused_matches_layered_and_oracle(spec_fn, dtype, wt):
    spec = spec_fn()
    mol = Molecule.from_name('LiH')
    h = MolecularHamiltonian(mol=mol)
    tree = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=5, perturb_envelopes=0.1)
    eng = Engine(spec, h, tree, dtype=dtype, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    B = 11                                     # not a multiple of any tile size
    r = make_walkers(mol, h.n_elec, B).astype(np.float32 if dtype == torch.float32 else np.float64)
    rt = torch.as_tensor(r)
    if wt:
        eng.set_option('fused_wt', wt)
    eng.set_option('fused', 1)
    s1, l1 = eng.wf_eval(rt)
    eng.set_option('fused', 0)
    s0, l0 = eng.wf_eval(rt)
    np.testing.assert_array_equal(s1.numpy(), s0.numpy())
    tol = 1e-12 if dtype == torch.float64 else 2e-5
    np.testing.assert_allclose(l1.numpy(), l0.numpy(), rtol=tol, atol=tol)
    ref = Interp(eng.program, mol.charges, geom.F32_EPS).run(r.astype(np.float64), mol.coords.astype(r.dtype).astype(np.float64), laplacian=False)
    np.testing.assert_array_equal(s1.numpy(), ref['sign'])
    tol = 1e-11 if dtype == torch.float64 else 2e-5
    np.testing.assert_allclose(l1.numpy(), ref['log'], rtol=tol, atol=tol)


def test_set_params_repacks_fused_weights():
    """Engine.set_params (new parameters after an optimiser step, fit.py:75-92 -> dqmc_set_weights) must refresh
    the plain weights AND the fragment-major copy the fused kernel reads: same result as a fresh engine."""
    spec = paulinet()
    mol = Molecule.from_name('LiH')
    h = MolecularHamiltonian(mol=mol)
    t0 = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=1, perturb_envelopes=0.1)
    t1 = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=2, perturb_envelopes=0.1)
    rt = torch.as_tensor(make_walkers(mol, h.n_elec, 5))
    eng = Engine(spec, h, t0, dtype=torch.float64, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    s_a, l_a = eng.wf_eval(rt)
    eng.set_params(t1)
    s_b, l_b = eng.wf_eval(rt)
    fresh = Engine(spec, h, t1, dtype=torch.float64, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    s_c, l_c = fresh.wf_eval(rt)
    assert not np.allclose(l_a.numpy(), l_b.numpy())
    np.testing.assert_array_equal(s_b.numpy(), s_c.numpy())
    np.testing.assert_array_equal(l_b.numpy(), l_c.numpy())
    e_b, _ = eng.local_energy(rt)
    e_c, _ = fresh.local_energy(rt)
    np.testing.assert_array_equal(e_b.numpy(), e_c.numpy())


def test_plan_variants_agree():
    """The lean unit body for small layers (default) and the general body (fused_lean 0) are different instruction
    sequences for the same arithmetic: identical psi in float64."""
    spec = paulinet()
    mol = Molecule.from_name('LiH')
    h = MolecularHamiltonian(mol=mol)
    tree = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=5, perturb_envelopes=0.1)
    eng = Engine(spec, h, tree, dtype=torch.float64, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    rt = torch.as_tensor(make_walkers(mol, h.n_elec, 6))
    s0, l0 = eng.wf_eval(rt)
    for opt, val in (('fused_lean', 0), ('fused_prio', 0), ('fused_prio', 2), ('fused', 2)):
        eng.set_option(opt, val)
        s1, l1 = eng.wf_eval(rt)
        np.testing.assert_array_equal(s1.numpy(), s0.numpy())
        np.testing.assert_allclose(l1.numpy(), l0.numpy(), rtol=1e-13, atol=1e-13)
    eng.set_option('fused_lean', 1)
    eng.set_option('fused_prio', 1)
    eng.set_option('fused', 1)
    # "fused" = 1 picks the LDS-resident kernel only where it is the faster value path: not for larger systems at large batch
    # (the layered kernels run then: same psi to float64 round-off)
    eng.set_option('fused', 0)
    s2, l2 = eng.wf_eval(rt)
    np.testing.assert_array_equal(s2.numpy(), s0.numpy())
    np.testing.assert_allclose(l2.numpy(), l0.numpy(), rtol=1e-12, atol=1e-12)
    eng.set_option('fused', 1)


@pytest.mark.parametrize('spec_fn,molname,wt', [(paulinet, 'LiH', 4), (ferminet, 'LiH', 2), (paulinet, 'H2O', 2)])
def test_bf16_pipe_units_match_float64(spec_fn, molname, wt):
    """Option 'fused_bf' (float32 contexts): the linear units of the fused kernel split their float32 operands into three
    bf16 pieces and multiply them with six bf16 MFMAs per K = 32 chunk (kernel_fused2.hip: FusedBfUnit).  Against the
    float64 oracle the result must be in the same accuracy class as the f32-MFMA units (option value 0), with identical
    signs; a batch that leaves the last tile ragged, tiles with several row blocks per unit (FermiNet / H2O)."""
    spec = spec_fn()
    mol = Molecule.from_name(molname)
    h = MolecularHamiltonian(mol=mol)
    tree = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=7, perturb_envelopes=0.1)
    eng = Engine(spec, h, tree, dtype=torch.float32, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    B = 13
    r = make_walkers(mol, h.n_elec, B).astype(np.float32)
    ref = Interp(eng.program, mol.charges, geom.F32_EPS).run(r.astype(np.float64), mol.coords.astype(np.float32).astype(np.float64), laplacian=False)
    eng.set_option('fused_wt', wt)
    eng.set_option('fused', 2)
    err = {}
    for bf in (0, 1, 2):
        eng.set_option('fused_bf', bf)
        s, l = eng.wf_eval(torch.as_tensor(r))
        np.testing.assert_array_equal(s.numpy(), ref['sign'])
        err[bf] = np.abs(l.numpy().astype(np.float64) - ref['log'])
    # (a walker next to a node of psi carries 1e-4 .. 1e-3 in float32 whichever instructions multiply: compare the classes)
    for bf in (1, 2):
        assert np.median(err[bf]) < 3 * np.median(err[0]) + 2e-7, (bf, np.median(err[0]), np.median(err[bf]))
        assert err[bf].max() < 5 * err[0].max() + 3e-5, (bf, err[0].max(), err[bf].max())
    assert np.median(err[0]) < 1e-5
