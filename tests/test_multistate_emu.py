"""Multi-state wrappers (reference sampling/combined_samplers.py:58-90, loss/energy.py:19-60,
loss/overlap.py:40-99) through the SIMT emulator against closed forms / the oracle."""
import numpy as np
import torch

from deepqmc_amd import MolecularHamiltonian, Molecule
from deepqmc_amd import loss
from deepqmc_amd.sampling import DecorrSampler, MultiElectronicStateSampler, synthetic_walkers
from deepqmc_amd.wf import NeuralNetworkWaveFunction
from oracle import geom, physics
from oracle import wf as owf
from simt_util import emu_lib


def test_two_states_energy_and_psi_ratio():
    h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
    wf = NeuralNetworkWaveFunction(h, 'paulinet', dtype=torch.float64, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    params = [wf.init(s, perturb_envelopes=0.1) for s in range(2)]
    B = 3
    r = torch.as_tensor(np.stack([synthetic_walkers(h, B, seed=10 + s) for s in range(2)]))[None]      # [1,S,B,N,3]
    E, stats = loss.compute_local_energy(None, h, wf, params, r)
    assert E.shape == (1, 2, B) and stats['hamil/E_kin'].shape == (1, 2)
    T = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
    for s in range(2):
        e_ref, _, _ = physics.batch_local_energy(owf.to_torch(params[s]), wf.spec, r[0, s], T(h.mol.coords), T(h.mol.charges),
                                                 h.n_up, geom.F32_EPS)
        np.testing.assert_allclose(E[0, s].numpy(), e_ref.numpy(), rtol=1e-8, atol=1e-8)
    R, ratio_stats = loss.compute_psi_ratio(wf, params, r)
    assert ratio_stats == {}
    assert R.shape == (1, 2, 2, B)
    np.testing.assert_allclose(R[0, 0, 0].numpy(), 1.0, rtol=1e-12)            # psi_j / psi_j on its own samples
    np.testing.assert_allclose(R[0, 1, 1].numpy(), 1.0, rtol=1e-12)
    # oracle: psi_0 on the samples of state 1, log-shifted as in overlap.py:70-75
    logs = np.zeros((2, 2, B)); signs = np.zeros((2, 2, B))
    for i in range(2):
        for j in range(2):
            sg, lg = physics.batch_wave_function(owf.to_torch(params[i]), wf.spec, r[0, j], T(h.mol.coords), h.n_up, geom.F32_EPS)
            logs[i, j], signs[i, j] = lg.numpy(), sg.numpy()
    shifted = logs - logs.mean(axis=(1, 2))[:, None, None]
    ref01 = signs[0, 1] * signs[1, 1] * np.exp(shifted[0, 1] - shifted[1, 1])
    np.testing.assert_allclose(R[0, 0, 1].numpy(), ref01, rtol=1e-9)


def test_multi_state_sampler_runs():
    h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
    wf = NeuralNetworkWaveFunction(h, 'paulinet', dtype=torch.float64, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    params = [wf.init(s, perturb_envelopes=0.1) for s in range(2)]
    ms = MultiElectronicStateSampler(DecorrSampler(h, wf, length=2, tau=0.3), 2)
    state = ms.init(0, params, 4)
    state, pc, stats = ms.sample(1, state, params)
    assert pc.r.shape == (2, 4, 4, 3) and len(stats['sampling/acceptance']) == 2
    for s in range(2):       # the carried psi is psi of the carried positions, with that state's parameters
        sg, lg = wf.apply(params[s], state[s]['r'])
        np.testing.assert_allclose(lg.numpy(), state[s]['psi'].log.numpy(), rtol=1e-12, atol=1e-12)


def test_langevin_sampler_matches_oracle():
    """LangevinSampler (HIP psi + gradient through the emulator, torch bookkeeping) vs the oracle's
    langevin_step (itself pinned to the reference's Langevin golden) on the same noise."""
    from deepqmc_amd.sampling import LangevinSampler
    from oracle import sampling as osamp
    h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
    wf = NeuralNetworkWaveFunction(h, 'paulinet', dtype=torch.float64, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    params = wf.init(3, perturb_envelopes=0.1)
    B, n = 4, 2
    smp = LangevinSampler(h, wf, tau=0.1)
    state = smp.init(5, params, B)
    rng = np.random.default_rng(0)
    noise, unif = rng.standard_normal((n, B, 4, 3)), rng.random((n, B))
    T = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
    p, R, Z = owf.to_torch(params), T(h.mol.coords), T(h.mol.charges)

    def psi_force(rb):
        sg, lg, gr = [], [], []
        for x in rb:
            x = x.clone().requires_grad_(True)
            s, l = owf.wave_function(p, wf.spec, x, R, h.n_up, geom.F32_EPS)
            g, = torch.autograd.grad(l, x)
            sg.append(s); lg.append(l.detach()); gr.append(g)
        return torch.stack(sg), torch.stack(lg), torch.stack(gr)

    s0, l0, g0 = psi_force(state['r'])
    ost = {'r': state['r'].clone(), 'sign': s0, 'log': l0, 'force': osamp.clean_force(g0, state['r'], R, Z, 0.1),
           'age': torch.zeros(B, dtype=torch.int64), 'tau': 0.1}
    np.testing.assert_allclose(state['force'].numpy(), ost['force'].numpy(), rtol=1e-8, atol=1e-10)
    for k in range(n):
        state, _, stats = smp.sample(k, state, params, noise=noise[k:k + 1], unif=unif[k:k + 1])
        ost, acc, a = osamp.langevin_step(psi_force, ost, R, Z, T(noise[k]), T(unif[k]))
    np.testing.assert_array_equal(state['age'].numpy(), ost['age'].numpy())
    np.testing.assert_allclose(state['r'].numpy(), ost['r'].numpy(), rtol=0, atol=1e-9)
    np.testing.assert_allclose(float(state['tau']), ost['tau'], rtol=1e-10)
    np.testing.assert_allclose(state['psi'].log.numpy(), ost['log'].numpy(), rtol=0, atol=1e-9)


def test_spin_exchange_sampler_matches_oracle():
    """OppositeSpinExchangeSampler (electron_samplers.py:235-330): the exchange step through the emulated HIP
    psi against the oracle restatement on the same choices; a non-exchange step is the wrapped sampler's."""
    from deepqmc_amd.sampling import MetropolisSampler, OppositeSpinExchangeSampler
    from oracle import sampling as osamp
    h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
    wf = NeuralNetworkWaveFunction(h, 'paulinet', dtype=torch.float64, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    params = wf.init(3, perturb_envelopes=0.1)
    B = 6
    smp = OppositeSpinExchangeSampler(MetropolisSampler(h, wf, tau=0.3), exchange_step_probability=0.5)
    state = smp.init(0, params, B)
    rng = np.random.default_rng(4)
    up, dn, u = rng.integers(0, h.n_up, B), rng.integers(0, h.n_down, B), rng.random(B)
    new, pc, stats = smp.sample(1, state, params, choices=(True, up, dn, u))
    T = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
    p = owf.to_torch(params)
    psi = lambda rr: physics.batch_wave_function(p, wf.spec, rr, T(h.mol.coords), h.n_up, geom.F32_EPS)
    ost = {'r': state['r'].clone(), 'sign': state['psi'].sign.to(torch.float64), 'log': state['psi'].log.clone(),
           'age': state['age'].to(torch.int64), 'tau': 0.3}
    onew, oacc = osamp.spin_exchange_step(psi, ost, h.n_up, torch.as_tensor(up), torch.as_tensor(dn), T(u))
    assert oacc.any() and not oacc.all()                     # the case exercises both branches
    np.testing.assert_array_equal(new['age'].numpy(), onew['age'].numpy())
    np.testing.assert_array_equal(new['r'].numpy(), onew['r'].numpy())
    np.testing.assert_allclose(new['psi'].log.numpy(), onew['log'].numpy(), rtol=1e-11, atol=1e-11)
    np.testing.assert_array_equal(new['psi'].sign.numpy(), onew['sign'].numpy().astype(np.int32))
    assert float(new['tau'][0]) == 0.3                       # no step-size adaptation on exchange steps
    # swapped electrons really are one up and one down electron of the same walker
    moved = (new['r'] != state['r']).any(-1)
    assert ((moved[:, :h.n_up].sum(1) == moved[:, h.n_up:].sum(1))).all()
    new2, _, stats2 = smp.sample(2, new, params, choices=(False, None, None, None))
    assert new2['tau'].item() != 0.3 or stats2['sampling/acceptance'] >= 0


def test_overlap_symmetrisation_known_answers():
    """The known answers of the reference's tests/test_overlap.py (TestSymmetrizeOverlap,
    TestComputeMeanOverlap) for loss.symmetrize_overlap_with_clipped_geometric_mean / compute_mean_overlap."""
    sym = loss.symmetrize_overlap_with_clipped_geometric_mean
    T = lambda a: torch.tensor(a, dtype=torch.float64)
    s06 = 0.06 ** 0.5
    np.testing.assert_allclose(sym(T([[1.0, 0.3], [0.2, 1.0]])).numpy(), [[1.0, s06], [s06, 1.0]], rtol=1e-14)
    y = sym(T([[1.0, -0.4], [0.3, 1.0]]))
    np.testing.assert_allclose(y.numpy(), [[1.0, 0.0], [0.0, 1.0]], atol=0)
    np.testing.assert_allclose(sym(T([[1.0, 2.0], [3.0, 1.0]])).numpy(), [[1.0, 6 ** 0.5], [6 ** 0.5, 1.0]], rtol=1e-14)
    y = sym(T([[1.0, 0.3, -0.5], [0.2, 1.0, 0.4], [0.6, 0.5, 1.0]]))
    np.testing.assert_allclose(y.numpy(), [[1.0, s06, 0.0], [s06, 1.0, 0.2 ** 0.5], [0.0, 0.2 ** 0.5, 1.0]], rtol=1e-14)
    ratio = T([[[[1.0, 1.0], [0.2, 0.4]], [[0.3, 0.5], [1.0, 1.0]]]])
    weight = T([[[1.0, 1.0], [0.8, 1.2]]])
    ov, stats = loss.compute_mean_overlap(ratio, weight)
    np.testing.assert_allclose(float(ov), 0.128, rtol=1e-14)
    np.testing.assert_allclose(stats['overlap/pairwise/mean'][0].numpy(), [[1.0, 0.128 ** 0.5], [0.128 ** 0.5, 1.0]], rtol=1e-14)


def test_multi_geometry_sampler_and_energies():
    """MultiNuclearGeometrySampler (combined_samplers.py:93-214) over two LiH bond lengths x two electronic states:
    only the molecules named by mol_idxs advance, samples carry their molecule index and geometry, and
    compute_local_energy [M,S,B] on the returned PhysicalConfiguration equals per-geometry engines."""
    from deepqmc_amd.engine import Engine
    from deepqmc_amd.sampling import IdleNucleiSampler, MoleculeIdxSampler, MultiNuclearGeometrySampler
    h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
    wf = NeuralNetworkWaveFunction(h, 'paulinet', dtype=torch.float64, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    S, B = 2, 2
    params = [wf.init(s, perturb_envelopes=0.1) for s in range(S)]
    Rs = torch.as_tensor(np.stack([h.mol.coords, h.mol.coords * 1.3, h.mol.coords * 0.8]))
    ms = MultiNuclearGeometrySampler(MultiElectronicStateSampler(DecorrSampler(h, wf, length=2, tau=0.2), S), IdleNucleiSampler())
    state = ms.init(0, params, B, Rs)
    before = [[st['r'].clone() for st in mol] for mol in state['elec']]
    state, pc, stats = ms.sample(1, state, params, [2, 0])
    assert pc.r.shape == (2, S, B, 4, 3) and pc.R.shape == (2, 2, 3) and pc.mol_idx.shape == (2, S, B)
    assert pc.mol_idx[0].unique().tolist() == [2] and pc.mol_idx[1].unique().tolist() == [0]
    assert torch.equal(pc.R[0], Rs[2]) and torch.equal(pc.R[1], Rs[0])
    for s in range(S):
        assert torch.equal(state['elec'][1][s]['r'], before[1][s])              # molecule 1 was not sampled
        assert torch.equal(state['elec'][2][s]['r'], pc.r[0, s])
    assert any(not torch.equal(state['elec'][m][s]['r'], before[m][s]) for m in (0, 2) for s in range(S))
    E, st = loss.compute_local_energy(None, h, wf, params, pc)
    assert E.shape == (2, S, B) and st['hamil/V_loc'].shape == (2, S)
    for k, m in enumerate([2, 0]):
        mol = Molecule(coords=Rs[m].numpy(), charges=h.mol.charges, charge=h.mol.charge, spin=h.mol.spin)
        for s in range(S):
            eng = Engine(wf.spec, MolecularHamiltonian(mol=mol), params[s], dtype=torch.float64, device='cpu', lib=emu_lib(),
                         norm_eps=geom.F32_EPS)
            e_ref, _ = eng.local_energy(pc.r[k, s])
            np.testing.assert_allclose(E[k, s].numpy(), e_ref.numpy(), rtol=1e-12, atol=1e-12)
    ratio, _ = loss.compute_psi_ratio(wf, params, pc)
    assert ratio.shape == (2, S, S, B)
    np.testing.assert_allclose(ratio[:, 0, 0].numpy(), 1.0, rtol=1e-12)
    idx = MoleculeIdxSampler(0, 3, 2)
    assert [idx.sample().tolist() for _ in range(3)] == [[0, 1], [2, 0], [1, 2]]


def test_initial_walkers_follow_each_geometry_and_counters_are_functional():
    """(a) MetropolisSampler.init hands R to the initialiser (electron_samplers.py:86-100), so the
    walkers of every geometry of a MultiNuclearGeometrySampler start around THAT geometry; (b) `sample` leaves the
    caller's update_nuc_counter untouched (functional state)."""
    from deepqmc_amd.sampling import IdleNucleiSampler, MultiNuclearGeometrySampler
    h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
    wf = NeuralNetworkWaveFunction(h, 'paulinet', dtype=torch.float64, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    params = wf.init(0, perturb_envelopes=0.1)
    shift = np.array([[40.0, 0, 0], [40.0, 0, 0]])
    Rs = torch.as_tensor(np.stack([h.mol.coords, h.mol.coords + shift]))
    ms = MultiNuclearGeometrySampler(DecorrSampler(h, wf, length=1, tau=0.2), IdleNucleiSampler(), update_nuc_period=3)
    state = ms.init(0, params, 16, Rs)
    for m in range(2):
        centre = state['elec'][m]['r'].numpy().mean(axis=(0, 1))
        np.testing.assert_allclose(centre, synthetic_walkers(h, 16, seed=0 * 2 + m, R=Rs[m]).mean(axis=(0, 1)), atol=1e-12)
        assert abs(centre[0] - float(Rs[m][:, 0].mean())) < 3.0            # 40 bohr apart: each ensemble sits on its own molecule
        assert np.isfinite(state['elec'][m]['psi'].log.numpy()).all()
    counter0 = state['update_nuc_counter'].copy()
    new_state, _, _ = ms.sample(1, state, params, [0, 1])
    np.testing.assert_array_equal(state['update_nuc_counter'], counter0)
    np.testing.assert_array_equal(new_state['update_nuc_counter'], counter0 + 1)


def test_exchange_indices_are_range_checked_and_stat_vectors_validated():
    from deepqmc_amd.engine import DqmcError
    import pytest
    h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
    wf = NeuralNetworkWaveFunction(h, 'paulinet', dtype=torch.float64, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    params = wf.init(0, perturb_envelopes=0.1)
    eng = wf.engine(params)
    B = 3
    r = torch.as_tensor(synthetic_walkers(h, B, seed=3))
    sign, log = eng.wf_eval(r)
    st = {'r': r.clone(), 'log': log, 'sign': sign, 'age': torch.zeros(B, dtype=torch.int32), 'tau': torch.full((1,), 0.3, dtype=torch.float64)}
    u = torch.full((B,), 0.5, dtype=torch.float64)
    with pytest.raises(DqmcError):
        eng.exchange_step(st, np.array([0, 2, 0]), np.array([0, 0, 0]), u)            # n_up = 2: index 2 is out of range
    with pytest.raises(DqmcError):
        eng.exchange_step(st, np.array([0, 1, 0]), np.array([0, -1, 0]), u)
    eng.exchange_step(st, np.array([0, 1, 0]), np.array([1, 0, 1]), u)
    e = torch.ones(4, dtype=torch.float64)
    with pytest.raises(DqmcError):
        eng.energy_record(e.to(torch.float32))
    with pytest.raises(DqmcError):
        eng.energy_record(torch.ones(8, dtype=torch.float64)[::2])
    with pytest.raises(DqmcError):
        eng.energy_record(e, torch.ones(3, dtype=torch.float64))
    assert eng.energy_record(e)[0] == 4


def test_engine_cache_eviction_keeps_held_engines_alive_and_accepts_tiled_R():
    """an evicted context is dropped, not closed (a caller may hold it); a per-walker tiled R is a
    valid geometry key."""
    h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
    wf = NeuralNetworkWaveFunction(h, 'paulinet', dtype=torch.float64, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    R_tiled = torch.as_tensor(np.broadcast_to(h.mol.coords, (2, 2, 3)).copy())
    assert wf._geometry_key(R_tiled) is None                                     # no nuclear tokens: geometry per call
    wf.max_engines = 2
    tree = wf.init(0, perturb_envelopes=0.1)
    r = torch.as_tensor(synthetic_walkers(h, 2, seed=3))
    # one context per (tree, geometry), as an ansatz with nuclear tokens has: every new geometry evicts the oldest
    geoms = [h.mol.coords * f for f in (1.0, 1.1, 1.2, 1.3)]
    wf._geometry_key = lambda R: np.asarray(R, np.float64).tobytes()
    held = wf.engine(tree, geoms[0])
    ref = held.wf_eval(r)[1].clone()
    for g in geoms[1:]:
        wf.engine(tree, g).wf_eval(r, torch.as_tensor(g))
    assert all(e is not held for _, _, e in wf._engines) and held._ctx is not None
    np.testing.assert_array_equal(held.wf_eval(r)[1].numpy(), ref.numpy())
