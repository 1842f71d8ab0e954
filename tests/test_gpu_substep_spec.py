"""`-m gpu`: the plan-specialised Metropolis sub-step kernel (deepqmc_amd/csrc/gen, written by deepqmc_amd/codegen) on the
MI355X through the C ABI -- against the oracle's float64 Metropolis restatement (oracle/sampling.py; reference
sampling/electron_samplers.py:102-163) on the same noise, and against the library's descriptor-driven kernel at the
BASELINE batch size.  float32 decisions can differ from float64 ones only where 2 (log|psi'| - log|psi|) sits within
round-off of log u; walkers whose accept history agrees must agree in position bit for bit (the proposal arithmetic is the same)."""
import numpy as np
import pytest
import torch

from deepqmc_amd.engine import Engine
from deepqmc_amd.hamil import MolecularHamiltonian
from deepqmc_amd.molecule import Molecule
from deepqmc_amd.params import init_params
from deepqmc_amd.sampling import synthetic_walkers
from deepqmc_amd.spec import paulinet
from oracle import geom
from oracle import sampling as osamp
from oracle import wf as owf

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def make(seed=5):
    spec = paulinet()
    mol = Molecule.from_name('LiH')
    h = MolecularHamiltonian(mol=mol)
    tree = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=seed, perturb_envelopes=0.1)
    eng = Engine(spec, h, tree, dtype=torch.float32, device=DEV, norm_eps=geom.F32_EPS)
    return spec, mol, h, tree, eng


def run(eng, r0, noise, unif, **kw):
    B = r0.shape[0]
    rt = torch.as_tensor(r0, device=DEV)
    sg, lg = eng.wf_eval(rt)
    st = {'r': rt.clone(), 'log': lg.clone(), 'sign': sg.clone(), 'age': torch.zeros(B, dtype=torch.int32, device=DEV),
          'tau': torch.full((1,), 0.3, dtype=torch.float32, device=DEV)}
    out, acc = eng.mcmc_steps(st, noise.shape[0], noise=noise, unif=unif, return_accept=True, **kw)
    return {k: v.cpu().numpy() for k, v in st.items()}, acc.cpu().numpy().astype(bool), out, (sg.cpu().numpy(), lg.cpu().numpy())


def test_specialised_sub_steps_against_the_oracle():
    spec, mol, h, tree, eng = make()
    assert eng.substep_kernel() == 'k_substep_lih_paulinet'
    B, n_sub = 72, 5                                     # 4.5 workgroups: a ragged last one
    rng = np.random.default_rng(0)
    r0 = synthetic_walkers(h, B, seed=2).astype(np.float32)
    noise = rng.standard_normal((n_sub, B, h.n_elec, 3)).astype(np.float32)
    unif = rng.random((n_sub, B)).astype(np.float32)
    st, acc, stats, (sg0, lg0) = run(eng, r0, noise, unif, max_age=3, target_acceptance=0.57)
    T = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
    ost = {'r': T(r0), 'sign': T(sg0), 'log': T(lg0), 'age': torch.zeros(B, dtype=torch.int64), 'tau': 0.3}
    ost, ostats, oacc = osamp.decorr_sample(owf.to_torch(tree), spec, ost, T(mol.coords), h.n_up, geom.F32_EPS, T(noise), T(unif),
                                            max_age=3, target_acceptance=0.57)
    oacc = oacc.numpy()
    same = (acc == oacc).all(axis=0)
    assert same.sum() >= B - 2, (B - same.sum(), 'walkers differ in an accept decision')
    np.testing.assert_array_equal(st['age'][same], ost['age'].numpy()[same])
    np.testing.assert_allclose(st['r'][same], ost['r'].numpy()[same], rtol=0, atol=2e-6)       # float32 r + tau xi against float64
    np.testing.assert_array_equal(st['sign'][same], ost['sign'].numpy()[same])
    err = np.abs(st['log'][same].astype(np.float64) - ost['log'].numpy()[same])
    assert np.median(err) < 5e-6 and err.max() < 1e-3, (np.median(err), err.max())
    if same.all():
        np.testing.assert_allclose(float(st['tau'][0]), ost['tau'], rtol=1e-6)
        np.testing.assert_allclose(stats['sampling/acceptance'], ostats['sampling/acceptance'], rtol=1e-12)


def test_specialised_against_descriptor_driven_kernel_4096():
    """BASELINE configs[1] batch: 4096 walkers, 10 sub-steps with the step-size adaptation on.  Both kernels multiply float32
    operands as three bf16 pieces; they sum in different orders, so a decision may flip where the ratio ties with u."""
    spec, mol, h, tree, eng = make(seed=3)
    B, n_sub = 4096, 10
    gen = torch.Generator(device='cpu').manual_seed(5)
    noise = torch.randn(n_sub, B, h.n_elec, 3, generator=gen).to(DEV)
    unif = torch.rand(n_sub, B, generator=gen).to(DEV)
    r0 = synthetic_walkers(h, B, seed=11).astype(np.float32)
    res = {}
    try:
        for on in (1, 0):
            eng.set_option('fused_spec', on)
            assert (eng.substep_kernel() != '') == bool(on)
            res[on] = run(eng, r0, noise, unif, target_acceptance=0.57)
    finally:
        eng.set_option('fused_spec', 1)
    (s1, a1, o1, _), (s0, a0, o0, _) = res[1], res[0]
    assert (a1[0] != a0[0]).mean() < 1e-3                 # first sub-step: identical state, decisions differ only at ties
    same = (a1 == a0).all(axis=0)
    assert same.mean() > 0.995, same.mean()
    np.testing.assert_array_equal(s1['age'][same], s0['age'][same])
    np.testing.assert_array_equal(s1['sign'][same], s0['sign'][same])
    if (a1 == a0).all():
        np.testing.assert_array_equal(s1['r'], s0['r'])
        np.testing.assert_array_equal(s1['tau'], s0['tau'])
    err = np.abs(s1['log'][same] - s0['log'][same])
    assert np.median(err) < 3e-6 and np.quantile(err, 0.999) < 3e-4, (np.median(err), np.quantile(err, 0.999))
    # psi of the final state is what a fresh evaluation gives (the sampler state stays consistent)
    sg, lg = eng.wf_eval(torch.as_tensor(s1['r'], device=DEV))
    np.testing.assert_array_equal(sg.cpu().numpy(), s1['sign'])
    np.testing.assert_allclose(lg.cpu().numpy(), s1['log'], rtol=0, atol=3e-4)
