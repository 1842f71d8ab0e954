"""The reference's OWN golden vectors compared DIRECTLY with the HIP path (no oracle in between): the reference's test
ansatz (tests/conf/ansatz.yaml; compiled by deepqmc_amd/program_featurewise.py) with the parameters its
`init(PRNGKey(0))` produces, at the reference's canonical LiH test walker:

    tests/test_wf/test_psi.npz                       sign = +1, log|psi| = -2.0604105875
    tests/test_wf/test_laplace_psi.npz               lap log|psi| = -168.7613557797 and the 12 quantum-force components
    tests/test_hamil/test_local_energy_Molecular_.npz  E_loc = -14.0194888479 (the reference asserts rtol 2e-4)
    tests/test_sampling/test_sampler_*_{Metropolis,DecorrMetropolis}_.npz   (GPU variant only: 4 x sample(PRNGKey(step)))

The same function runs in the CPU suite through the SIMT emulation of the kernels (`device = 'cpu'`) and, marked
`gpu`, on the MI355X.  Residual differences (1e-7 level) are the float32 erfinv of haiku's truncated-normal
initialiser, which the parameter emulation evaluates in float64 (see tests/test_reference_goldens.py)."""
import numpy as np
import pytest
import torch

from deepqmc_amd.types import PhysicalConfiguration
from oracle import geom
from oracle.jaxrng import JaxRNG
from ref_ansatz_util import reference_test_engine


def check_wf_goldens(kats, device, lib=None):
    h, P, tree, eng = reference_test_engine(device, lib=lib, norm_eps=geom.F64_EPS)
    r = torch.as_tensor(kats['lih_edges_ne'][0][None], dtype=torch.float64, device=device)
    sign, log = eng.wf_eval(r)
    assert int(sign[0]) == int(kats['wf_psi_sign'])
    np.testing.assert_allclose(float(log[0]), float(kats['wf_psi_log']), rtol=0, atol=2e-6)
    e, stats, grad = eng.local_energy(r, return_grad=True)
    np.testing.assert_allclose(float(stats['hamil/lap'][0]), float(kats['wf_lap_log_psis']), rtol=1e-6)
    np.testing.assert_allclose(grad[0].cpu().numpy(), kats['wf_quantum_force'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(float(e[0]), float(kats['hamil_E_loc']), rtol=1e-6)       # reference tolerance: 2e-4
    return h, eng


def test_reference_goldens_through_emulated_hip_path(kats):
    from simt_util import emu_lib
    check_wf_goldens(kats, 'cpu', emu_lib())


@pytest.mark.gpu
def test_reference_goldens_through_hip_path(kats):
    check_wf_goldens(kats, 'cuda:0')


def test_reference_metropolis_golden_through_emulated_hip_path(kats):
    from simt_util import emu_lib
    check_sampler_golden(kats, 'cpu', emu_lib(), 'Metropolis', 1, None)


@pytest.mark.gpu
@pytest.mark.parametrize('tag,length,max_age', [('Metropolis', 1, None), ('DecorrMetropolis', 20, 20)])
def test_reference_sampler_goldens_through_hip_path(kats, tag, length, max_age):
    check_sampler_golden(kats, 'cuda:0', None, tag, length, max_age)


def check_sampler_golden(kats, dev, lib, tag, length, max_age):
    """reference tests/test_sampling.py:30-83 through dqmc_mcmc_steps: psi of the 10 golden initial walkers, then
    4 x sample(PRNGKey(step)) driven by the emulated jax.random.normal / uniform streams; the final sampler state and
    all seven statistics against the golden (the reference asserts rtol 5e-4)."""
    h, P, tree, eng = reference_test_engine(dev, lib=lib, norm_eps=geom.F64_EPS)
    rng = JaxRNG(True)
    r0 = torch.as_tensor(kats[f'sampler_init_{tag}_r'], dtype=torch.float64, device=dev)
    sign0, log0 = eng.wf_eval(r0)
    np.testing.assert_array_equal(sign0.cpu().numpy(), kats[f'sampler_init_{tag}_psi:sign'])
    np.testing.assert_allclose(log0.cpu().numpy(), kats[f'sampler_init_{tag}_psi:log'], rtol=0, atol=2e-5)
    st = {'r': r0.clone(), 'log': log0.clone(), 'sign': sign0.clone(), 'age': torch.zeros(10, dtype=torch.int32, device=dev),
          'tau': torch.full((1,), 0.1, dtype=torch.float64, device=dev)}
    for step in range(4):
        keys = rng.split(rng.key(step), length) if length > 1 else [rng.key(step)]
        noise, unif = [], []
        for k in keys:
            kp, ka = rng.split(k, 2)
            noise.append(rng.normal(kp, (10, 4, 3)))
            unif.append(rng.uniform(ka, (10,), np.float64, 0.0, 1.0))
        stats = eng.mcmc_steps(st, length, max_age=max_age, target_acceptance=0.57, noise=np.stack(noise), unif=np.stack(unif))
    g = lambda k: kats[f'sampler_sample_{tag}_{k}']
    np.testing.assert_array_equal(st['age'].cpu().numpy(), g('smpl_state:age'))
    np.testing.assert_allclose(float(st['tau'][0]), float(g('smpl_state:tau')), rtol=1e-12)
    np.testing.assert_allclose(st['r'].cpu().numpy(), g('smpl_state:r'), rtol=0, atol=1e-12)
    np.testing.assert_allclose(st['log'].cpu().numpy(), g('smpl_state:psi:log'), rtol=0, atol=2e-5)
    np.testing.assert_array_equal(st['sign'].cpu().numpy(), g('smpl_state:psi:sign'))
    for key in ('acceptance', 'tau', 'age/mean', 'age/max', 'log_psi/mean', 'log_psi/std', 'dists/mean'):
        np.testing.assert_allclose(stats[f'sampling/{key}'], float(g(f'stats:sampling/{key}')), rtol=1e-5, err_msg=key)


def check_langevin_golden(kats, device, lib=None):
    """reference tests/test_sampling.py (Langevin, tau = 0.1) through dqmc_langevin_update / dqmc_langevin_steps:
    the cleaned initial drift of the 10 golden walkers, then 4 x sample(PRNGKey(step))."""
    h, P, tree, eng = reference_test_engine(device, lib=lib, norm_eps=geom.F64_EPS)
    rng = JaxRNG(True)
    Z = h.mol.charges
    r0 = torch.as_tensor(kats['sampler_init_Langevin_r'], dtype=torch.float64, device=device)
    tau = torch.full((1,), 0.1, dtype=torch.float64, device=device)
    sign0, log0, f0 = eng.langevin_update(r0, tau, Z)
    np.testing.assert_allclose(f0.cpu().numpy(), kats['sampler_init_Langevin_force'], rtol=0, atol=5e-5)
    st = {'r': r0.clone(), 'log': log0, 'sign': sign0, 'force': f0, 'age': torch.zeros(10, dtype=torch.int32, device=device), 'tau': tau}
    for step in range(4):
        kp, ka = rng.split(rng.key(step), 2)
        stats = eng.langevin_steps(st, 1, Z, target_acceptance=0.57, noise=rng.normal(kp, (10, 4, 3))[None],
                                   unif=rng.uniform(ka, (10,), np.float64, 0.0, 1.0)[None])
    g = lambda k: kats[f'sampler_sample_Langevin_{k}']
    np.testing.assert_array_equal(st['age'].cpu().numpy(), g('smpl_state:age'))
    np.testing.assert_allclose(float(st['tau'][0]), float(g('smpl_state:tau')), rtol=1e-12)
    np.testing.assert_allclose(st['r'].cpu().numpy(), g('smpl_state:r'), rtol=0, atol=2e-5)
    np.testing.assert_allclose(st['force'].cpu().numpy(), g('smpl_state:force'), rtol=0, atol=1e-4)
    np.testing.assert_allclose(st['log'].cpu().numpy(), g('smpl_state:psi:log'), rtol=0, atol=2e-5)
    assert stats['sampling/acceptance'] == float(g('stats:sampling/acceptance'))


def test_reference_langevin_golden_through_emulated_hip_path(kats):
    from simt_util import emu_lib
    check_langevin_golden(kats, 'cpu', emu_lib())


@pytest.mark.gpu
def test_reference_langevin_golden_through_hip_path(kats):
    check_langevin_golden(kats, 'cuda:0')
