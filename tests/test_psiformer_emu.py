"""Psiformer (attention layers) through the SIMT emulator: forward-Laplacian attention kernel,
log-rescaled input features, spin feature, projection -- against the NumPy interpreter."""
import numpy as np
import torch

from deepqmc_amd.engine import Engine
from deepqmc_amd.hamil import MolecularHamiltonian
from deepqmc_amd.molecule import Molecule
from deepqmc_amd.params import init_params
from deepqmc_amd.spec import AnsatzSpec, MLPSpec, psiformer, transpsiformer
from oracle import geom
from oracle.program_interp import Interp
from simt_util import emu_lib
from test_program_interp import make_walkers
import dataclasses
import pytest


def small_psiformer():
    """Psiformer with a narrow embedding so the emulated run stays short (same op set)."""
    return dataclasses.replace(psiformer(), embedding_dim=32, n_interactions=2, n_determinants=4)


def test_psiformer_emu_f64():
    spec = small_psiformer()
    mol = Molecule.from_name('LiH')
    h = MolecularHamiltonian(mol=mol)
    tree = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=5, perturb_envelopes=0.1)
    eng = Engine(spec, h, tree, dtype=torch.float64, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    B = 2
    r = make_walkers(mol, h.n_elec, B)
    it = Interp(eng.program, mol.charges, geom.F32_EPS)
    ref = it.run(r, mol.coords, laplacian=True)
    from buffers_util import check_every_buffer
    (e, stats, grad), _ = check_every_buffer(eng, it, B, lambda: eng.local_energy(torch.as_tensor(r), return_grad=True))
    np.testing.assert_allclose(e.numpy(), ref['e_loc'], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(grad.numpy(), ref['grad'], rtol=1e-8, atol=1e-8)
    val = it.run(r, mol.coords, laplacian=False)
    sign, logpsi = eng.wf_eval(torch.as_tensor(r))
    np.testing.assert_array_equal(sign.numpy(), val['sign'])
    np.testing.assert_allclose(logpsi.numpy(), val['log'], rtol=1e-11, atol=1e-11)


def test_transpsiformer_emu_f64():
    """TransPsiformer (conf/ansatz/transpsiformer.yaml): attention of the electron queries over
    [nuclear tokens; electrons] with the nuclear stream folded on the host, envelope exponents read out of
    the nuclear embeddings (3 per nucleus, pi = 1).  HIP kernels (emulated) vs the NumPy interpreter on every
    buffer, and vs the un-folded torch oracle (full masked attention over all tokens) on psi and E_loc."""
    from oracle import physics
    from oracle import wf as owf
    mol = Molecule.from_name('LiH')
    spec = dataclasses.replace(transpsiformer(mol.charges), embedding_dim=32, n_interactions=2, n_determinants=4)
    h = MolecularHamiltonian(mol=mol)
    tree = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=5, perturb_envelopes=0.1)
    eng = Engine(spec, h, tree, dtype=torch.float64, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    B = 2
    r = make_walkers(mol, h.n_elec, B)
    it = Interp(eng.program, mol.charges, geom.F32_EPS)
    ref = it.run(r, mol.coords, laplacian=True)
    from buffers_util import check_every_buffer
    (e, stats, grad), _ = check_every_buffer(eng, it, B, lambda: eng.local_energy(torch.as_tensor(r), return_grad=True))
    np.testing.assert_allclose(e.numpy(), ref['e_loc'], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(grad.numpy(), ref['grad'], rtol=1e-8, atol=1e-8)
    T = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
    e_ref, _, _ = physics.batch_local_energy(owf.to_torch(tree), spec, T(r), T(mol.coords), T(mol.charges), h.n_up, geom.F32_EPS)
    np.testing.assert_allclose(e.numpy(), e_ref.numpy(), rtol=1e-8, atol=1e-8)
    sign, logpsi = eng.wf_eval(torch.as_tensor(r))
    s_ref, l_ref = physics.batch_wave_function(owf.to_torch(tree), spec, T(r), T(mol.coords), h.n_up, geom.F32_EPS)
    np.testing.assert_array_equal(sign.numpy(), s_ref.numpy().astype(np.int32))
    np.testing.assert_allclose(logpsi.numpy(), l_ref.numpy(), rtol=1e-10, atol=1e-10)
    # a different geometry needs a new engine (the nuclear stream is folded at the Hamiltonian's R)
    import pytest
    from deepqmc_amd.engine import DqmcError
    with pytest.raises(DqmcError):
        eng.wf_eval(torch.as_tensor(r), R=torch.as_tensor(mol.coords + 0.1))


@pytest.mark.parametrize('ansatz', ['psiformer', 'transpsiformer'])
def test_attention_mfma_f32(ansatz):
    """The MFMA attention kernel (float32 build, head_dim = 16: kernel_attention_mfma.hip) against the scalar
    kernel on the attention output buffers, value and Laplacian mode, and against the float64 interpreter on
    E_loc; with nuclear-token keys (constant rows) for the TransPsiformer."""
    mol = Molecule.from_name('LiH')
    base = transpsiformer(mol.charges) if ansatz == 'transpsiformer' else psiformer()
    spec = dataclasses.replace(base, embedding_dim=64, n_interactions=2, n_determinants=4)
    h = MolecularHamiltonian(mol=mol)
    tree = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=5, perturb_envelopes=0.1)
    eng = Engine(spec, h, tree, dtype=torch.float32, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    B = 2
    r = make_walkers(mol, h.n_elec, B).astype(np.float32)
    names = [n for n in eng.program.buf_names if n.endswith('/att')]
    assert names
    out = {}
    for flag in (1, 0):
        eng.set_option('attention_mfma', 2 if flag else 0)       # 2: force the MFMA kernel on this small system
        e, _ = eng.local_energy(torch.as_tensor(r))
        out[flag] = (e.numpy().copy(), {n: eng.debug_read(n, B) for n in names})
        s, l = eng.wf_eval(torch.as_tensor(r))
        out[flag] += (l.numpy().copy(),)
    for n in names:
        a, b = out[1][1][n], out[0][1][n]
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-5 * max(1.0, np.abs(b).max()), err_msg=n)
    np.testing.assert_allclose(out[1][0], out[0][0], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(out[1][2], out[0][2], rtol=1e-5, atol=1e-5)
    it = Interp(eng.program, mol.charges, geom.F32_EPS)
    ref = it.run(r.astype(np.float64), mol.coords.astype(np.float32).astype(np.float64), laplacian=True)
    np.testing.assert_allclose(out[1][0], ref['e_loc'], rtol=5e-4, atol=5e-4)


def test_attention_query_blocks_are_independent():
    """The scalar attention kernel splits the queries over workgroups when its tile set would not fit the LDS (float64,
    42 electrons); forcing the split on a small system must not change a single bit (rows of the attention matrix are
    independent given all keys)."""
    import os
    from deepqmc_amd.engine import Engine
    from deepqmc_amd.sampling import synthetic_walkers
    mol = Molecule.from_name('LiH')
    h = MolecularHamiltonian(mol=mol)
    spec = psiformer()
    tree = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=3, perturb_envelopes=0.1)
    eng = Engine(spec, h, tree, dtype=torch.float64, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    r = torch.as_tensor(synthetic_walkers(h, 2, seed=3))
    e0, st0, g0 = eng.local_energy(r, return_grad=True)
    os.environ['DQMC_ATTN_QSPLIT'] = '3'
    try:
        e1, st1, g1 = eng.local_energy(r, return_grad=True)
    finally:
        del os.environ['DQMC_ATTN_QSPLIT']
    np.testing.assert_array_equal(e0.numpy(), e1.numpy())
    np.testing.assert_array_equal(g0.numpy(), g1.numpy())


@pytest.mark.parametrize('ansatz', ['psiformer', 'transpsiformer'])
def test_attention_mfma_f64_two_row_blocks(ansatz):
    """The float64 instance of the MFMA attention kernel (round 4: q0 and P held as A fragments in registers, dP scratch per
    active wave) on 18 electrons -- two query row blocks, ragged second block, 56 of 64 lanes used, nuclear-token keys for
    the TransPsiformer -- against the NumPy interpreter on every buffer, E_loc, gradient and psi."""
    mol = Molecule(coords=np.array([[-1.3, 0.0, 0.0], [1.3, 0.0, 0.0]]), charges=np.array([9, 9]), charge=0, spin=0)
    base = transpsiformer(mol.charges) if ansatz == 'transpsiformer' else psiformer()
    spec = dataclasses.replace(base, embedding_dim=64, n_interactions=1, n_determinants=2)
    h = MolecularHamiltonian(mol=mol)
    assert h.n_elec == 18
    tree = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=5, perturb_envelopes=0.1)
    eng = Engine(spec, h, tree, dtype=torch.float64, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    B = 1
    r = make_walkers(mol, h.n_elec, B)
    it = Interp(eng.program, mol.charges, geom.F32_EPS)
    ref = it.run(r, mol.coords, laplacian=True)
    eng.timing(True); eng.timing_reset()
    from buffers_util import check_every_buffer
    (e, stats, grad), _ = check_every_buffer(eng, it, B, lambda: eng.local_energy(torch.as_tensor(r), return_grad=True))
    rep = eng.timing_report(); eng.timing(False)
    assert 'attention' in rep
    np.testing.assert_allclose(e.numpy(), ref['e_loc'], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(grad.numpy(), ref['grad'], rtol=1e-8, atol=1e-8)
    # ... and bit-for-bit the same energies as the scalar kernel is NOT expected (different summation order): 1e-10
    eng.set_option('attention_mfma', 0)
    e0 = eng.local_energy(torch.as_tensor(r))[0]
    np.testing.assert_allclose(e.numpy(), e0.numpy(), rtol=1e-10, atol=1e-10)
    eng.set_option('attention_mfma', 1)
    eng.set_option('fused', 0)
    val = it.run(r, mol.coords, laplacian=False)
    sign, logpsi = eng.wf_eval(torch.as_tensor(r))
    np.testing.assert_array_equal(sign.numpy(), val['sign'])
    np.testing.assert_allclose(logpsi.numpy(), val['log'], rtol=1e-11, atol=1e-11)


def test_attention_mfma_split_three_row_blocks():
    """The eight-wave attention kernel (a pair of waves per query row block: kernel_attention_mfma.hip, split variant) on 38
    electrons: three row blocks, three key tiles (the pair owns two / one of them), 116 of 128 lanes -- against the four-wave
    kernel (option "attention_split" 0) and the interpreter (float64: the only instance the product builds)."""
    mol = Molecule(coords=np.array([[-1.4, 0.0, 0.0], [1.4, 0.0, 0.0]]), charges=np.array([19, 19]), charge=0, spin=0)
    spec = dataclasses.replace(psiformer(), embedding_dim=64, n_interactions=1, n_determinants=1)
    h = MolecularHamiltonian(mol=mol)
    assert h.n_elec == 38
    tree = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=5, perturb_envelopes=0.1)
    eng = Engine(spec, h, tree, dtype=torch.float64, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    B = 1
    r = make_walkers(mol, h.n_elec, B)
    it = Interp(eng.program, mol.charges, geom.F32_EPS)
    ref = it.run(r, mol.coords, laplacian=True)
    names = [n for n in eng.program.buf_names if n.endswith('/att')]
    e1, _, g1 = eng.local_energy(torch.as_tensor(r), return_grad=True)          # float64 default: split
    att1 = {n: eng.debug_read(n, B) for n in names}
    eng.set_option('attention_split', 0)
    e0, _, g0 = eng.local_energy(torch.as_tensor(r), return_grad=True)
    for n in names:
        np.testing.assert_allclose(att1[n], eng.debug_read(n, B), rtol=1e-11, atol=1e-12, err_msg=n)
        np.testing.assert_allclose(att1[n], it.bufs[eng.program.buf_names[n]], rtol=1e-9, atol=1e-10, err_msg=n)
    np.testing.assert_allclose(e1.numpy(), e0.numpy(), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(e1.numpy(), ref['e_loc'], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(g1.numpy(), ref['grad'], rtol=1e-8, atol=1e-8)
