"""The plan-specialised Metropolis sub-step kernel (deepqmc_amd/codegen/substep.py -> deepqmc_amd/csrc/gen/*.hip): the
committed source is what the generator writes today, the library binds it to the program it was generated from (same
structural hash in Python and in C++), and -- through the SIMT emulation of the generated source -- it reproduces the
descriptor-driven kernel and the oracle's Metropolis step: accept bits, ages, step size, positions exactly, log|psi| to
float32 round-off.  Reference: sampling/electron_samplers.py:76-138."""
import numpy as np
import pytest
import torch

from deepqmc_amd.codegen import TARGETS, Unsupported, generate, program_hash
from deepqmc_amd.codegen.__main__ import main as codegen_main, target_program
from deepqmc_amd.engine import Engine
from deepqmc_amd.hamil import MolecularHamiltonian
from deepqmc_amd.molecule import Molecule
from deepqmc_amd.params import init_params
from deepqmc_amd.spec import ferminet, paulinet
from oracle import geom
from oracle import sampling as osamp
from oracle.program_interp import Interp
from simt_util import emu_lib
from test_program_interp import make_walkers


def test_committed_sources_are_current():
    assert codegen_main(['--check']) == 0


def test_unsupported_programs_are_refused():
    """N != 4 (H2O: 10 electrons) and spin patterns whose edge rows are not whole xor blocks (3 up + 1 down) keep the generic kernel."""
    with pytest.raises(Unsupported):
        generate('x', target_program('H2O', 'paulinet'))
    from deepqmc_amd.program import compile_program
    sp = paulinet()
    tree = init_params(sp, 3, 1, 2, seed=0)
    with pytest.raises(Unsupported):
        generate('x', compile_program(sp, tree, 3, 1, 2, R=np.zeros((2, 3)), eps=geom.F32_EPS))


def _engines(seed, B, n_sub):
    spec = paulinet()
    mol = Molecule.from_name('LiH')
    h = MolecularHamiltonian(mol=mol)
    tree = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=seed, perturb_envelopes=0.1)
    r0 = make_walkers(mol, h.n_elec, B).astype(np.float32)
    rng = np.random.default_rng(seed)
    noise = rng.standard_normal((n_sub, B, 4, 3)).astype(np.float32)
    unif = rng.random((n_sub, B)).astype(np.float32)
    return spec, mol, h, tree, r0, noise, unif


def _run(spec, h, tree, r0, noise, unif, spec_on, max_age=None):
    eng = Engine(spec, h, tree, dtype=torch.float32, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    eng.set_option('fused_spec', spec_on)
    B, n_sub = r0.shape[0], noise.shape[0]
    rt = torch.as_tensor(r0.copy())
    sg, lg = eng.wf_eval(rt)
    st = {'r': rt, 'log': lg.clone(), 'sign': sg.clone(), 'age': torch.zeros(B, dtype=torch.int32), 'tau': torch.full((1,), 0.3, dtype=torch.float32)}
    out, acc = eng.mcmc_steps(st, n_sub, noise=noise, unif=unif, return_accept=True, max_age=max_age)
    return eng, {k: v.numpy().copy() for k, v in st.items()}, acc.numpy().copy(), out


@pytest.mark.parametrize('B,max_age', [(21, None), (37, 1)])     # ragged last tile / last workgroup; forced moves
def test_specialised_kernel_matches_generic_and_oracle(B, max_age):
    spec, mol, h, tree, r0, noise, unif = _engines(5, B, 3)
    eng1, s1, a1, o1 = _run(spec, h, tree, r0, noise, unif, 1, max_age)
    assert eng1.substep_kernel() == 'k_substep_' + TARGETS[0][0]
    # the hash the generator wrote into the source is the hash the library computes from the program it is handed
    p = eng1.program
    assert program_hash(p.n_up, p.n_down, p.n_nuc, p.spec.n_determinants, p.bufs, p.ops, p.itable) == \
        program_hash(*(lambda q: (q.n_up, q.n_down, q.n_nuc, q.spec.n_determinants, q.bufs, q.ops, q.itable))(target_program('LiH', 'paulinet')))
    eng0, s0, a0, o0 = _run(spec, h, tree, r0, noise, unif, 0, max_age)
    assert eng0.substep_kernel() == ''
    np.testing.assert_array_equal(a1, a0)
    np.testing.assert_array_equal(s1['r'], s0['r'])
    np.testing.assert_array_equal(s1['age'], s0['age'])
    np.testing.assert_array_equal(s1['sign'], s0['sign'])
    np.testing.assert_array_equal(s1['tau'], s0['tau'])
    # (a walker next to a node of psi carries 1e-4 in float32 whichever kernel multiplies: bound the worst, pin the typical)
    np.testing.assert_allclose(s1['log'], s0['log'], rtol=0, atol=3e-4)
    assert np.median(np.abs(s1['log'] - s0['log'])) < 3e-6
    assert o1['sampling/acceptance'] == o0['sampling/acceptance']
    # the final log|psi| of every walker against the float64 interpreter of the same program at the final positions
    ref = Interp(eng1.program, mol.charges, geom.F32_EPS).run(s1['r'].astype(np.float64), mol.coords.astype(np.float32).astype(np.float64), laplacian=False)
    np.testing.assert_array_equal(s1['sign'], ref['sign'])
    np.testing.assert_allclose(s1['log'], ref['log'], rtol=0, atol=3e-4)
    assert np.median(np.abs(s1['log'] - ref['log'])) < 3e-6


def test_new_parameters_repack_the_tape():
    """dqmc_set_weights refreshes the weight tape of the specialised kernel: same sub-steps as a fresh context."""
    spec, mol, h, tree, r0, noise, unif = _engines(7, 9, 2)
    tree2 = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=8, perturb_envelopes=0.1)
    eng = Engine(spec, h, tree, dtype=torch.float32, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    eng.set_params(tree2)
    fresh = Engine(spec, h, tree2, dtype=torch.float32, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    res = []
    for e in (eng, fresh):
        rt = torch.as_tensor(r0.copy())
        sg, lg = e.wf_eval(rt)
        st = {'r': rt, 'log': lg.clone(), 'sign': sg.clone(), 'age': torch.zeros(9, dtype=torch.int32), 'tau': torch.full((1,), 0.3, dtype=torch.float32)}
        _, acc = e.mcmc_steps(st, 2, noise=noise, unif=unif, return_accept=True)
        res.append((st['r'].numpy().copy(), st['log'].numpy().copy(), acc.numpy().copy()))
    for a, b in zip(res[0], res[1]):
        np.testing.assert_array_equal(a, b)


def test_other_programs_keep_the_generic_kernel():
    """LiH / FermiNet has no specialised kernel in the library (not in codegen.TARGETS): nothing is bound, sub-steps run as before."""
    spec = ferminet()
    h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
    tree = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=1, perturb_envelopes=0.1)
    eng = Engine(spec, h, tree, dtype=torch.float32, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    assert eng.substep_kernel() == ''
    eng64 = Engine(paulinet(), h, init_params(paulinet(), h.n_up, h.n_down, h.n_nuc, seed=1), dtype=torch.float64, device='cpu', lib=emu_lib())
    assert eng64.substep_kernel() == ''        # float64 contexts: the descriptor-driven kernel (parity build)
