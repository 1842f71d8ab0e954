"""Kernel-logic tests on the CPU: the UNMODIFIED HIP kernel sources, compiled against the SIMT
emulation shim (tests/simt), executed through the same C ABI and Python engine as on the GPU,
and compared buffer by buffer with the NumPy forward-Laplacian interpreter and with the
autograd oracle.  These check index arithmetic (MFMA fragment maps, lane shuffles, LDS tiling,
barriers); the `-m gpu` tests repeat the comparison on the real device."""
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


def _setup(spec_fn, molname, dtype, B, seed=5):
    spec = spec_fn()
    mol = Molecule.from_name(molname)
    h = MolecularHamiltonian(mol=mol)
    tree = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=seed, perturb_envelopes=0.1)
    eng = Engine(spec, h, tree, dtype=dtype, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    r = make_walkers(mol, h.n_elec, B)
    it = Interp(eng.program, mol.charges, geom.F32_EPS)
    return spec, mol, h, eng, r, it


@pytest.mark.parametrize('spec_fn,molname', [(paulinet, 'LiH'), (ferminet, 'LiH')])
def test_emu_f64_buffers_and_energy(spec_fn, molname):
    B = 3
    spec, mol, h, eng, r, it = _setup(spec_fn, molname, torch.float64, B)
    ref = it.run(r, mol.coords, laplacian=True)
    e, stats, grad = eng.local_energy(torch.as_tensor(r), return_grad=True)
    for name, idx in eng.program.buf_names.items():
        got = eng.debug_read(name, B)
        np.testing.assert_allclose(got, it.bufs[idx], rtol=1e-10, atol=1e-10, err_msg=f'buffer {name}')
    np.testing.assert_allclose(eng.debug_read('logdet', B), it.logdet, rtol=1e-9, atol=1e-9)
    np.testing.assert_array_equal(eng.debug_read('sign_k', B), it.sign_k)
    np.testing.assert_allclose(e.numpy(), ref['e_loc'], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(grad.numpy(), ref['grad'], rtol=1e-9, atol=1e-9)
    for k, key in enumerate(['hamil/V_el', 'hamil/E_kin', 'hamil/V_loc', 'hamil/V_nl', 'hamil/lap', 'hamil/quantum_force']):
        np.testing.assert_allclose(stats[key].numpy(), ref['stats'][k], rtol=1e-9, atol=1e-9)
    # value-only path
    val = it.run(r, mol.coords, laplacian=False)
    sign, logpsi = eng.wf_eval(torch.as_tensor(r))
    np.testing.assert_array_equal(sign.numpy(), val['sign'])
    np.testing.assert_allclose(logpsi.numpy(), val['log'], rtol=1e-11, atol=1e-11)


def test_emu_f32_energy():
    B = 2
    spec, mol, h, eng, r, it = _setup(paulinet, 'LiH', torch.float32, B)
    r32 = r.astype(np.float32)
    ref = it.run(r32.astype(np.float64), mol.coords.astype(np.float32).astype(np.float64), laplacian=True)
    e, stats = eng.local_energy(torch.as_tensor(r32))
    # float32 tolerance of the path (north star: 1e-5 relative on E_loc)
    np.testing.assert_allclose(e.numpy(), ref['e_loc'], rtol=2e-5, atol=2e-5)
    sign, logpsi = eng.wf_eval(torch.as_tensor(r32))
    np.testing.assert_array_equal(sign.numpy(), ref['sign'])
    np.testing.assert_allclose(logpsi.numpy(), ref['log'], rtol=1e-5, atol=1e-5)
