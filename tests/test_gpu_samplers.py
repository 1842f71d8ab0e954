"""`-m gpu` tests of the callers either side of the hot path, on the device through the C ABI against the oracle:
Langevin (MALA) sub-steps and drift cleaning (dqmc_langevin_*), opposite-spin exchange steps (dqmc_exchange_step),
the batch-level `compute_local_energy [M,S,B]` and the psi-ratio matrix of the overlap penalty."""
import json
import os

import numpy as np
import pytest
import torch

from deepqmc_amd import MolecularHamiltonian, Molecule, loss
from deepqmc_amd.sampling import LangevinSampler, MetropolisSampler, OppositeSpinExchangeSampler, synthetic_walkers
from deepqmc_amd.wf import NeuralNetworkWaveFunction
from oracle import geom, physics
from oracle import sampling as osamp
from oracle import wf as owf

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
T = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)


def make(molname='LiH', ansatz='paulinet', dtype=torch.float64):
    h = MolecularHamiltonian(mol=Molecule.from_name(molname))
    wf = NeuralNetworkWaveFunction(h, ansatz, dtype=dtype, device=DEV, norm_eps=geom.F32_EPS)
    return h, wf


def oracle_psi_force(p, spec, h, R):
    def f(rb):
        sg, lg, gr = [], [], []
        for x in rb:
            x = x.clone().requires_grad_(True)
            s, l = owf.wave_function(p, spec, x, R, h.n_up, geom.F32_EPS)
            g, = torch.autograd.grad(l, x)
            sg.append(s); lg.append(l.detach()); gr.append(g)
        return torch.stack(sg), torch.stack(lg), torch.stack(gr)
    return f


@pytest.mark.parametrize('molname,ansatz', [('LiH', 'paulinet'), ('Be', 'ferminet')])
def test_langevin_steps_match_oracle_f64(molname, ansatz):
    h, wf = make(molname, ansatz)
    params = wf.init(3, perturb_envelopes=0.1)
    B, n = 16, 4
    smp = LangevinSampler(h, wf, tau=0.1, max_age=3)
    smp.length = n
    state = smp.init(5, params, B)
    rng = np.random.default_rng(0)
    noise, unif = rng.standard_normal((n, B, h.n_elec, 3)), rng.random((n, B))
    p, R, Z = owf.to_torch(params), T(h.mol.coords), T(h.mol.charges)
    psi_force = oracle_psi_force(p, wf.spec, h, R)
    r0 = state['r'].cpu()
    s0, l0, g0 = psi_force(r0)
    ost = {'r': r0.clone(), 'sign': s0, 'log': l0, 'force': osamp.clean_force(g0, r0, R, Z, 0.1),
           'age': torch.zeros(B, dtype=torch.int64), 'tau': 0.1}
    np.testing.assert_allclose(state['force'].cpu().numpy(), ost['force'].numpy(), rtol=1e-9, atol=1e-11)
    state, _, stats = smp.sample(0, state, params, noise=noise, unif=unif)
    accs = []
    for k in range(n):
        ost, acc, a = osamp.langevin_step(psi_force, ost, R, Z, T(noise[k]), T(unif[k]), max_age=3)
        accs.append(acc)
    assert torch.stack(accs).any() and not torch.stack(accs).all()
    np.testing.assert_array_equal(state['age'].cpu().numpy(), ost['age'].numpy())
    np.testing.assert_allclose(state['r'].cpu().numpy(), ost['r'].numpy(), rtol=0, atol=1e-10)
    np.testing.assert_allclose(state['force'].cpu().numpy(), ost['force'].numpy(), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(float(state['tau'][0]), ost['tau'], rtol=1e-11)
    np.testing.assert_allclose(state['psi'].log.cpu().numpy(), ost['log'].numpy(), rtol=0, atol=1e-10)
    np.testing.assert_allclose(stats['sampling/acceptance'], a, rtol=1e-12)


def test_langevin_device_rng_f32():
    """float32, device Philox noise: the chain stays self-consistent (carried psi / drift = psi / drift of the carried
    positions) and the acceptance sits in a sane range."""
    h, wf = make(dtype=torch.float32)
    params = wf.init(3, perturb_envelopes=0.1)
    smp = LangevinSampler(h, wf, tau=0.05)
    smp.length = 10
    state = smp.init(5, params, 2048)
    state, pc, stats = smp.sample(7, state, params)
    assert 0.2 < stats['sampling/acceptance'] <= 1.0
    chk = smp.update(state, params)
    np.testing.assert_array_equal(chk['psi'].sign.cpu().numpy(), state['psi'].sign.cpu().numpy())
    np.testing.assert_allclose(chk['psi'].log.cpu().numpy(), state['psi'].log.cpu().numpy(), rtol=0, atol=1e-4)


def test_exchange_step_matches_oracle_f64():
    h, wf = make()
    params = wf.init(3, perturb_envelopes=0.1)
    B = 64
    smp = OppositeSpinExchangeSampler(MetropolisSampler(h, wf, tau=0.3), exchange_step_probability=0.5)
    state = smp.init(0, params, B)
    rng = np.random.default_rng(4)
    up, dn, u = rng.integers(0, h.n_up, B), rng.integers(0, h.n_down, B), rng.random(B)
    new, pc, stats = smp.sample(1, state, params, choices=(True, up, dn, u))
    p = owf.to_torch(params)
    psi = lambda rr: physics.batch_wave_function(p, wf.spec, rr, T(h.mol.coords), h.n_up, geom.F32_EPS)
    ost = {'r': state['r'].cpu().clone(), 'sign': state['psi'].sign.cpu().to(torch.float64), 'log': state['psi'].log.cpu().clone(),
           'age': state['age'].cpu().to(torch.int64), 'tau': 0.3}
    onew, oacc = osamp.spin_exchange_step(psi, ost, h.n_up, torch.as_tensor(up), torch.as_tensor(dn), T(u))
    assert oacc.any() and not oacc.all()
    np.testing.assert_array_equal(new['age'].cpu().numpy(), onew['age'].numpy())
    np.testing.assert_array_equal(new['r'].cpu().numpy(), onew['r'].numpy())
    np.testing.assert_allclose(new['psi'].log.cpu().numpy(), onew['log'].numpy(), rtol=1e-11, atol=1e-11)
    np.testing.assert_array_equal(new['psi'].sign.cpu().numpy(), onew['sign'].numpy().astype(np.int32))
    np.testing.assert_allclose(stats['sampling/acceptance'], float(oacc.double().mean()), rtol=1e-12)
    assert float(new['tau'][0]) == 0.3


def test_batched_local_energy_and_psi_ratio_three_states():
    """compute_local_energy [M=1, S=3, B] (loss/energy.py:19-60) and the psi-ratio matrix (loss/overlap.py:40-99) for
    three electronic states (three parameter sets = three HIP contexts) against the oracle, float64."""
    h, wf = make()
    S, B = 3, 8
    params = [wf.init(s, perturb_envelopes=0.1) for s in range(S)]
    r = torch.as_tensor(np.stack([synthetic_walkers(h, B, seed=10 + s) for s in range(S)]), device=DEV)[None]      # [1,S,B,N,3]
    E, stats = loss.compute_local_energy(None, h, wf, params, r)
    assert E.shape == (1, S, B) and stats['hamil/E_kin'].shape == (1, S)
    Rt, Zt = T(h.mol.coords), T(h.mol.charges)
    logs, signs = np.zeros((S, S, B)), np.zeros((S, S, B))
    for s in range(S):
        e_ref, st_ref, _ = physics.batch_local_energy(owf.to_torch(params[s]), wf.spec, r[0, s].cpu(), Rt, Zt, h.n_up, geom.F32_EPS)
        np.testing.assert_allclose(E[0, s].cpu().numpy(), e_ref.numpy(), rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(float(stats['hamil/E_kin'][0, s]), float(st_ref['hamil/E_kin'].mean()), rtol=1e-8)
        for j in range(S):
            sg, lg = physics.batch_wave_function(owf.to_torch(params[s]), wf.spec, r[0, j].cpu(), Rt, h.n_up, geom.F32_EPS)
            logs[s, j], signs[s, j] = lg.numpy(), sg.numpy()
    ratio, _ = loss.compute_psi_ratio(wf, params, r)
    assert ratio.shape == (1, S, S, B)
    shifted = logs - logs.mean(axis=(1, 2))[:, None, None]
    for i in range(S):
        for j in range(S):
            ref = signs[i, j] * signs[j, j] * np.exp(shifted[i, j] - shifted[j, j])
            np.testing.assert_allclose(ratio[0, i, j].cpu().numpy(), ref, rtol=1e-9)
    w = torch.ones(1, S, B, dtype=torch.float64, device=DEV)
    pen, info = loss.compute_mean_overlap(ratio, w)
    mean = np.zeros((S, S))
    for i in range(S):
        for j in range(S):
            mean[i, j] = (signs[i, j] * signs[j, j] * np.exp(shifted[i, j] - shifted[j, j])).mean()
    sym = np.sign(mean) * np.sqrt(np.clip(mean * mean.T, 0, None))
    np.testing.assert_allclose(info['overlap/pairwise/mean'][0].cpu().numpy(), sym, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(float(pen), sum(sym[i, j] ** 2 for i in range(S) for j in range(i + 1, S)), rtol=1e-9)


def test_multi_geometry_batches_incl_nuclear_tokens():
    """MultiNuclearGeometrySampler on the device; for the TransPsiformer (nuclear stream folded into the program) the
    facade builds one context per geometry, for PauliNet the geometry is an argument of each call."""
    from deepqmc_amd.engine import Engine
    from deepqmc_amd.sampling import DecorrSampler, MultiElectronicStateSampler, MultiNuclearGeometrySampler
    from deepqmc_amd.types import PhysicalConfiguration
    for ansatz in ('paulinet', 'transpsiformer'):
        h, wf = make('LiH', ansatz)
        S, B = 2, 16
        params = [wf.init(s, perturb_envelopes=0.1) for s in range(S)]
        Rs = torch.as_tensor(np.stack([h.mol.coords, h.mol.coords * 1.25]), device=DEV)
        ms = MultiNuclearGeometrySampler(MultiElectronicStateSampler(DecorrSampler(h, wf, length=3, tau=0.3), S))
        state = ms.init(0, params, B, Rs)
        state, pc, stats = ms.sample(1, state, params, [1, 0])
        E, st = loss.compute_local_energy(None, h, wf, params, pc)
        assert E.shape == (2, S, B) and torch.isfinite(E).all()
        # against the ORACLE at the geometry each sample belongs to (not against another HIP context)
        from oracle import physics
        from oracle import wf as owf
        T = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
        for k, m in enumerate([1, 0]):
            Rm = T(Rs[m].cpu().numpy())
            for s in range(S):
                p = owf.to_torch(params[s])
                rk = T(pc.r[k, s].cpu().numpy())
                e_ref, _, _ = physics.batch_local_energy(p, wf.spec, rk, Rm, T(h.mol.charges), h.n_up, geom.F32_EPS)
                np.testing.assert_allclose(E[k, s].cpu().numpy(), e_ref.numpy(), rtol=1e-8, atol=1e-8)
                _, lg = physics.batch_wave_function(p, wf.spec, T(state['elec'][m][s]['r'].cpu().numpy()), Rm, h.n_up, geom.F32_EPS)
                np.testing.assert_allclose(state['elec'][m][s]['psi'].log.cpu().numpy(), lg.numpy(), rtol=1e-9, atol=1e-9)


def test_inlib_rccl_allgather_single_rank():
    """dqmc_energy_stats_allgather: the per-rank record, ONE ncclAllGather over an RCCL communicator and the Chan merge
    inside the library (here a 1-rank communicator created with ncclCommInitRank; the multi-rank protocol is the same
    call) against the host-side path and NumPy."""
    from deepqmc_amd import parallel
    h, wf = make(dtype=torch.float32)
    eng = wf.engine(wf.init(0))
    comm = parallel.RcclCommunicator(0, 1)
    try:
        e = torch.randn(4096, dtype=torch.float32, device=DEV) * 3 - 7
        out = parallel.energy_stats_inlib(eng, e, comm)
        ref = parallel.energy_stats(eng, e)
        x = e.double().cpu().numpy()
        for k in ref:
            np.testing.assert_allclose(out[k], ref[k], rtol=1e-14)
        np.testing.assert_allclose(out['local_energy/mean'], x.mean(), rtol=1e-12)
        np.testing.assert_allclose(out['local_energy/std'], x.std(), rtol=1e-10)
    finally:
        comm.close()


def test_inlib_rccl_allgather_two_ranks_through_bench():
    """The multi-rank protocol on hardware when the node has it: `bench.py --gpus 2` (one process per GPU under
    torch.distributed.run) reduces the energy statistics with ONE ncclAllGather inside the library on each context's stream
    (`dqmc_energy_stats_allgather` over an `RcclCommunicator` whose unique id travels by one torch.distributed broadcast).
    Skipped on a 1-GPU box (the gloo world-size-2 tests cover the host path there)."""
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip('needs >= 2 visible GPUs')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--walkers', '512', '--steps', '2', '--warmup', '1',
                        '--repeats', '2', '--equilibrate', '50', '--no-cpu-baseline'], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1])
    assert out['n_gpus'] == 2 and out['n_ranks_seen'] == 2
    assert out['config']['reduction'].startswith('in-library')
    assert np.isfinite(out['energy']['local_energy/mean']) and out['energy']['local_energy/std'] > 0


def test_evaluate_spin_on_device():
    """`evaluate_spin` (reference physics.py:159-226) through the HIP value path on the device: float64 against the oracle's
    literal loop; float32 (the fused LDS-resident kernel, 4096 x 4 swapped configurations in one launch) against float64."""
    from deepqmc_amd.physics import evaluate_spin
    from oracle import physics as ophys
    h, wf = make('LiH', 'paulinet')
    params = wf.init(2, perturb_envelopes=0.1)
    B = 6
    r = torch.as_tensor(synthetic_walkers(h, B, seed=6), device=DEV)
    s2 = evaluate_spin(h, wf)(params, r)
    p = owf.to_torch(params)
    ref = torch.stack([ophys.evaluate_spin(p, wf.spec, r[b].cpu(), T(h.mol.coords), h.n_up, h.n_down, geom.F32_EPS) for b in range(B)])
    np.testing.assert_allclose(s2.cpu().numpy(), ref.numpy(), rtol=1e-9, atol=1e-10)
    h32, wf32 = make('LiH', 'paulinet', dtype=torch.float32)
    big = torch.as_tensor(synthetic_walkers(h, 4096, seed=7), device=DEV)
    a = evaluate_spin(h32, wf32)(params, big.float())
    b = evaluate_spin(h, wf)(params, big.float().double())
    err = ((a - b).abs() / b.abs().clamp(min=1.0)).cpu().numpy()
    assert np.quantile(err, 0.99) < 1e-3 and np.median(err) < 1e-5, (np.median(err), np.quantile(err, 0.99))
