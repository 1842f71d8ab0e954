"""Host facade contracts through the SIMT emulator: engine cache keyed by tree identity (no stale weights when
CPython recycles an id), nuclear repulsion recomputed from the R of every call (reference physics.py:112-116),
functional sampler states (reference samplers return new states), rng required with a non-local ECP
(gaussian_type_ecp.py:183)."""
import gc

import numpy as np
import pytest
import torch

from deepqmc_amd import MolecularHamiltonian, Molecule
from deepqmc_amd.engine import DqmcError, Engine
from deepqmc_amd.sampling import DecorrSampler, synthetic_walkers
from deepqmc_amd.types import PhysicalConfiguration
from deepqmc_amd.wf import NeuralNetworkWaveFunction
from oracle import geom
from simt_util import emu_lib


def make(spec='paulinet', **kw):
    h = MolecularHamiltonian(mol=Molecule.from_name('LiH'), **kw)
    wf = NeuralNetworkWaveFunction(h, spec, dtype=torch.float64, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    return h, wf


def test_engine_cache_never_serves_stale_weights():
    """The documented training-loop flow: a NEW parameter tree each iteration while old ones are freed.  An
    id()-keyed cache returned engines holding old weights; identity-checked entries cannot."""
    h, wf = make()
    wf.max_engines = 2
    r = torch.as_tensor(synthetic_walkers(h, 2, seed=3))
    for it in range(8):
        params = wf.init(it, perturb_envelopes=0.1)
        got = wf.apply(params, r).log.numpy()
        fresh = Engine(wf.spec, h, params, dtype=torch.float64, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
        np.testing.assert_array_equal(got, fresh.wf_eval(r)[1].numpy())
        fresh.close()
        del params
        gc.collect()
    assert len(wf._engines) <= 2
    # leaves changed in place are picked up after invalidate()
    params = wf.init(100, perturb_envelopes=0.1)
    a = wf.apply(params, r).log.numpy().copy()
    for mod in params.values():
        for k in mod:
            mod[k] = mod[k] * 1.01
    wf.invalidate(params)
    assert not np.allclose(wf.apply(params, r).log.numpy(), a)


def test_nuclear_repulsion_follows_the_call_geometry():
    h, wf = make()
    params = wf.init(0, perturb_envelopes=0.1)
    eng = wf.engine(params)
    r = torch.as_tensor(synthetic_walkers(h, 2, seed=4))
    R2 = h.mol.coords * 1.5
    e_call, _ = eng.local_energy(PhysicalConfiguration(torch.as_tensor(R2), r, None))
    mol2 = Molecule(coords=R2, charges=h.mol.charges, charge=h.mol.charge, spin=h.mol.spin)
    h2 = MolecularHamiltonian(mol=mol2)
    eng2 = Engine(wf.spec, h2, params, dtype=torch.float64, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    e_built, _ = eng2.local_energy(r)
    np.testing.assert_allclose(e_call.numpy(), e_built.numpy(), rtol=1e-13, atol=1e-13)
    e_orig, _ = eng.local_energy(r)
    assert np.abs(e_orig.numpy() - e_call.numpy()).min() > 1e-3


def test_sampler_states_are_not_aliased():
    h, wf = make()
    params = wf.init(0, perturb_envelopes=0.1)
    sampler = DecorrSampler(h, wf, length=3, tau=0.4)
    s0 = sampler.init(1, params, 4)
    keep = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in s0.items() if k != 'psi'}
    keep_log = s0['psi'].log.clone()
    s1, pc, stats = sampler.sample(2, s0, params)
    assert not torch.equal(s1['r'], s0['r'])                       # something moved ...
    for k, v in keep.items():                                      # ... and the previous state is intact
        assert torch.equal(s0[k], v), k
    assert torch.equal(s0['psi'].log, keep_log)
    with pytest.raises(DqmcError):
        wf.engine(params).mcmc_steps({'r': s1['r'].float(), 'log': s1['psi'].log, 'sign': s1['psi'].sign, 'age': s1['age'],
                                      'tau': s1['tau']}, 1)
    # opt-in in-place states (no clones per call): the same chain, advanced in the caller's tensors
    inplace = DecorrSampler(h, wf, length=3, tau=0.4, in_place=True)
    t0 = inplace.init(1, params, 4)
    r_ptr = t0['r'].data_ptr()
    t1, _, stats_ip = inplace.sample(2, t0, params)
    assert t1['r'].data_ptr() == r_ptr and torch.equal(t1['r'], s1['r']) and torch.equal(t1['psi'].log, s1['psi'].log)
    assert torch.equal(t1['age'], s1['age']) and torch.equal(t1['tau'], s1['tau']) and stats_ip == stats


ECP_TABLES = {'Li': [0, [[-1, [[], [[5.41, 1.0]], [[4.6, -4.6]], [[2.7, 5.41]]]], [0, [[], [], [[1.33, 6.75]]]], [1, [[], [], [[1.25, 0.45]]]]]]}


def test_ecp_needs_rng_and_psi_grad_skips_the_quadrature():
    h, wf = make(ecp_type='synthetic', ecp_tables=ECP_TABLES)
    params = wf.init(0, perturb_envelopes=0.1)
    eng = wf.engine(params)
    r = torch.as_tensor(synthetic_walkers(h, 2, seed=5))
    with pytest.raises(DqmcError):
        eng.local_energy(r, rng=None)
    e, st, g = eng.local_energy(r, rng=3, return_grad=True)
    eng.timing(True)
    eng.timing_reset()
    sign, log, grad = eng.psi_and_grad(r)
    rep = eng.timing_report()
    eng.timing(False)
    assert 'ecp' not in rep                                        # no quadrature walkers were evaluated
    np.testing.assert_allclose(grad.reshape(2, -1).numpy(), g.numpy(), rtol=1e-13)
    s2, l2 = eng.wf_eval(r)
    np.testing.assert_array_equal(sign.numpy(), s2.numpy())
    np.testing.assert_allclose(log.numpy(), l2.numpy(), rtol=1e-12)


def test_f64_refinement_of_ill_conditioned_walkers():
    """float32 context: walkers k_final flags (score = (|lap| + |grad|^2) / max(1, |E_loc|) x conditioning record above
    the threshold) are re-evaluated by the float64 twin; their results equal a float64 engine's (rounded to float32),
    the others stay float32.  Probe off = fixed threshold; the self-calibrating probe is exercised below."""
    import dataclasses
    from deepqmc_amd.spec import paulinet
    h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
    small_net = dataclasses.replace(paulinet(), embedding_dim=32, n_interactions=1, n_determinants=4)      # (keeps the emulation short)
    wf32 = NeuralNetworkWaveFunction(h, small_net, dtype=torch.float32, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    wf64 = NeuralNetworkWaveFunction(h, small_net, dtype=torch.float64, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    params = wf32.init(5, perturb_envelopes=0.1)
    e32, e64 = wf32.engine(params), wf64.engine(params)
    B = 6
    r = torch.as_tensor(synthetic_walkers(h, B, seed=11).astype(np.float32))
    e32.set_option('refine', 0)
    e_plain, st_plain, g_plain = e32.local_energy(r, return_grad=True)
    assert e32.last_refined() == 0
    ratio = ((st_plain['hamil/lap'].abs() + st_plain['hamil/quantum_force']) / e_plain.abs().clamp(min=1.0)).numpy()
    ratio = ratio * np.maximum(1.0, e32.debug_read('kappa', B))
    thr = max(1, int(np.median(ratio)))            # a threshold that splits these walkers
    e32.set_option('refine', 1)
    e32.set_option('refine_probe', 0)
    e32.set_option('refine_thresh', thr)
    e_ref, st_ref, g_ref = e32.local_energy(r, return_grad=True)
    n = e32.last_refined()
    flagged = ratio > thr
    assert n == int(flagged.sum()) and 0 < n < B
    R32 = torch.as_tensor(h.mol.coords, dtype=torch.float32).double()      # the twin sees the float32 context's geometry
    e_d, st_d, g_d = e64.local_energy(PhysicalConfiguration(R32, r.double(), None), return_grad=True)
    np.testing.assert_array_equal(e_ref.numpy()[flagged], e_d.numpy()[flagged].astype(np.float32))
    np.testing.assert_array_equal(e_ref.numpy()[~flagged], e_plain.numpy()[~flagged])
    np.testing.assert_array_equal(g_ref.numpy()[flagged], g_d.numpy()[flagged].astype(np.float32))
    for k in st_ref:
        np.testing.assert_array_equal(st_ref[k].numpy()[flagged], st_d[k].numpy()[flagged].astype(np.float32))
        np.testing.assert_array_equal(st_ref[k].numpy()[~flagged], st_plain[k].numpy()[~flagged])
    # set_params reaches the twin too
    p2 = wf32.init(6, perturb_envelopes=0.1)
    e32.set_params(p2)
    e2, _ = e32.local_energy(r)
    fresh = Engine(wf32.spec, h, p2, dtype=torch.float32, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    fresh.set_option('refine_probe', 0)
    fresh.set_option('refine_thresh', thr)
    e3, _ = fresh.local_energy(r)
    np.testing.assert_array_equal(e2.numpy(), e3.numpy())
    # the self-calibrating probe (library default): the first call evaluates a sample in float64 as well, derives the
    # threshold from the measured error per unit of score and applies it to the same call
    auto = Engine(wf32.spec, h, params, dtype=torch.float32, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    auto.set_option('refine_thresh', 10 ** 6)          # start loose: the whole batch is calibration sample
    auto.set_option('refine_target_e7', 1)             # 1e-7: below float32 resolution -> a low threshold
    e_auto, _ = auto.local_energy(r)
    thr_auto = auto.refine_info()['score_threshold']
    assert auto.refine_info()['error_per_score'] > 0 and thr_auto < 10 ** 6          # the probe ran and moved the threshold
    above = ratio > thr_auto
    assert auto.last_refined() == int(above.sum()) > 0
    np.testing.assert_array_equal(e_auto.numpy()[above], e_d.numpy()[above].astype(np.float32))
    np.testing.assert_array_equal(e_auto.numpy()[~above], e_plain.numpy()[~above])       # sample walkers are not written back
    e_again, _ = auto.local_energy(r)                  # no probe this time: same threshold, same result
    np.testing.assert_array_equal(e_again.numpy(), e_auto.numpy())
    # a smaller calibration sample (option refine_sample): the probe still runs and what a walker gets still depends on its
    # score and the derived threshold alone
    small = Engine(wf32.spec, h, params, dtype=torch.float32, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    small.set_option('refine_sample', 3)
    small.set_option('refine_thresh', 10 ** 6)
    small.set_option('refine_target_e7', 1)
    e_small, _ = small.local_energy(r)
    thr_small = small.refine_info()['score_threshold']
    assert small.refine_info()['error_per_score'] > 0 and thr_small < 10 ** 6
    above_s = ratio > thr_small
    assert small.last_refined() == int(above_s.sum())
    np.testing.assert_array_equal(e_small.numpy()[above_s], e_d.numpy()[above_s].astype(np.float32))
    np.testing.assert_array_equal(e_small.numpy()[~above_s], e_plain.numpy()[~above_s])
    loose = Engine(wf32.spec, h, params, dtype=torch.float32, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    loose.set_option('refine_probe', 0)
    loose.set_option('refine_thresh', 10 ** 9)
    e_l, _ = loose.local_energy(r)
    assert loose.last_refined() == 0
    np.testing.assert_array_equal(e_l.numpy(), e_plain.numpy())


def test_direct_mode_hysteresis_counters_and_scores():
    """The whole-batch float64 ("direct") mode of the refinement (engine.hip: mostly_flagged): a context enters it when more
    than "refine_direct_pct" (60 %) of a batch lies above the score threshold and leaves it only when a later float32 pass
    finds fewer than "refine_direct_exit_pct" (45 %) above it -- a flagged share between the two lines keeps whatever mode
    the context is in.  Also: dqmc_refine_scores returns the scores the decision was taken on, dqmc_refine_counters what ran.
    (A reduced PauliNet keeps the emulation short; thresholds are set by hand, the probe is off.)"""
    import dataclasses
    from deepqmc_amd.params import init_params
    from deepqmc_amd.spec import paulinet
    h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
    spec = dataclasses.replace(paulinet(), embedding_dim=32, n_interactions=1, n_determinants=4)
    params = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=5, perturb_envelopes=0.1)
    B = 20
    r = torch.as_tensor(synthetic_walkers(h, B, seed=21).astype(np.float32))
    eng = Engine(spec, h, params, dtype=torch.float32, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    eng.set_option('refine_probe', 0)
    eng.set_option('refine_direct_calls', 1)
    eng.set_option('refine_direct_backoff', 0)           # (the doubling of the stay is checked at the end)
    eng.set_option('refine_thresh', 10 ** 9)
    e_plain, st0 = eng.local_energy(r)
    score = eng.refine_scores(B)
    expect = ((st0['hamil/lap'].abs() + st0['hamil/quantum_force']) / e_plain.abs().clamp(min=1.0)).numpy() * np.maximum(1.0, eng.debug_read('kappa', B))
    np.testing.assert_allclose(score, expect, rtol=1e-5)
    def above(t):
        return int((score > t).sum())

    def thr_for(n_above):        # an integer threshold (dqmc_set_option takes integers) with exactly n_above scores above it
        for t in range(1, int(score.max()) + 1):
            if above(t) == n_above:
                return t
        raise AssertionError(f'no integer threshold leaves {n_above} scores above it: {np.sort(score)}')
    t_mid, t_hi, t_lo = thr_for(10), thr_for(15), thr_for(8)      # 50 % / 75 % / 40 % of 20 walkers (lines: enter 60 %, exit 45 %)
    c0 = eng.refine_counters()
    # 50 % above, coming from the mixed mode: stays mixed
    eng.set_option('refine_thresh', t_mid)
    eng.local_energy(r)
    assert eng.last_refined() == 10 and eng.refine_info()['direct_f64_calls_left'] == 0
    # 75 %: enters the direct mode (this call evaluates the whole batch in float64, and so does the next one)
    eng.set_option('refine_thresh', t_hi)
    e_dir, _ = eng.local_energy(r)
    assert eng.last_refined() == B and eng.refine_info()['direct_f64_calls_left'] == 1
    eng.set_option('refine_thresh', t_mid)
    eng.local_energy(r)                                  # the direct call
    assert eng.last_refined() == B and eng.refine_info()['direct_f64_calls_left'] == 0
    with pytest.raises(Exception):
        eng.refine_scores(B)                             # no float32 pass ran: no scores
    # 50 % again, but coming FROM the direct mode: above the exit line, so the context stays there
    e_again, _ = eng.local_energy(r)
    assert eng.last_refined() == B and eng.refine_info()['direct_f64_calls_left'] == 1
    np.testing.assert_array_equal(e_again.numpy(), e_dir.numpy())
    eng.local_energy(r)                                  # (direct)
    # 40 %: below the exit line -> back to the mixed mode
    eng.set_option('refine_thresh', t_lo)
    e_mix, _ = eng.local_energy(r)
    assert eng.last_refined() == 8 and eng.refine_info()['direct_f64_calls_left'] == 0
    keep = score <= t_lo
    np.testing.assert_array_equal(e_mix.numpy()[keep], e_plain.numpy()[keep])
    np.testing.assert_array_equal(e_mix.numpy()[~keep], e_dir.numpy()[~keep])
    c1 = eng.refine_counters()
    assert c1['calls'] - c0['calls'] == 6 and c1['direct_f64_calls'] - c0['direct_f64_calls'] == 4 and c1['probe_calls'] == 0
    assert c1['walkers_refined'] - c0['walkers_refined'] == 10 + 4 * B + 8
    # back-off: every float32 look that CONFIRMS the direct mode doubles the stay, "refine_direct_backoff" times at most
    eng.set_option('refine_direct_backoff', 2)
    eng.set_option('refine_thresh', t_hi)
    stays = []
    for _ in range(4):
        eng.local_energy(r)                              # a look (float32 pass + decision)
        left = eng.refine_info()['direct_f64_calls_left']
        stays.append(left)
        for _ in range(left):
            eng.local_energy(r)                          # the direct calls of the stay
    assert stays == [1, 2, 4, 4], stays


@pytest.mark.parametrize('molname,ansatz', [('LiH', 'paulinet'), ('C', 'ferminet')])
def test_evaluate_spin_matches_the_reference_loop(molname, ansatz):
    """`evaluate_spin` (reference physics.py:159-226): all n_up n_down swapped configurations of a walker batch as ONE
    value-only evaluation through the (emulated) HIP engine against the oracle's literal double loop per walker; the carbon
    atom is spin-polarised (4 up / 2 down: a non-zero constant term).  For a determinant-based ansatz <S^2> of a sample
    is not an eigenvalue, so the check is the estimator, not a number."""
    from deepqmc_amd.physics import evaluate_spin, make_stochastic_spin_raising_operator
    from oracle import physics as ophys
    from oracle import wf as owf
    import dataclasses
    from deepqmc_amd.spec import ANSATZES
    h = MolecularHamiltonian(mol=Molecule.from_name(molname))
    spec = dataclasses.replace(ANSATZES[ansatz](), embedding_dim=32, n_interactions=1, n_determinants=2)     # (keeps the emulation short)
    wf = NeuralNetworkWaveFunction(h, spec, dtype=torch.float64, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    params = wf.init(2, perturb_envelopes=0.1)
    B = 3
    r = torch.as_tensor(synthetic_walkers(h, B, seed=6))
    s2 = evaluate_spin(h, wf)(params, r)
    p = owf.to_torch(params)
    T = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
    ref = torch.stack([ophys.evaluate_spin(p, wf.spec, r[b], T(h.mol.coords), h.n_up, h.n_down, geom.F32_EPS) for b in range(B)])
    np.testing.assert_allclose(s2.numpy(), ref.numpy(), rtol=1e-9, atol=1e-10)
    # the stochastic estimator: <S^2> = const + sum over down electrons of (raising-operator value - 1)
    op = make_stochastic_spin_raising_operator(h, wf)
    umd = h.n_up - h.n_down
    tot = torch.full((B,), umd / 2 * (umd / 2 + 1) + h.n_down, dtype=torch.float64)
    for d in range(h.n_up, h.n_up + h.n_down):
        tot = tot + op(params, r, d) - 1.0
    np.testing.assert_allclose(tot.numpy(), s2.numpy(), rtol=1e-10, atol=1e-10)
    per_walker = op(params, PhysicalConfiguration(torch.as_tensor(h.mol.coords), r, None), torch.tensor([h.n_up, h.n_up + h.n_down - 1, h.n_up]))
    np.testing.assert_allclose(per_walker[0].item(), op(params, r[:1], h.n_up)[0].item(), rtol=1e-12)
    np.testing.assert_allclose(per_walker[1].item(), op(params, r[1:2], h.n_up + h.n_down - 1)[0].item(), rtol=1e-12)


def test_engine_is_released_by_reference_count():
    """A context owns device memory by the tens of GB (workspace, float64 twin, captured graphs): it must go when the last
    reference goes, not when the cyclic collector gets round to it (round 4: a `self`-capturing closure kept every Engine
    in a cycle; on the GPU box a run of large tests then ran out of memory)."""
    import gc
    import weakref
    h, wf = make()
    params = wf.init(0, perturb_envelopes=0.1)
    gc.collect()
    gc.disable()
    try:
        eng = Engine(wf.spec, h, params, dtype=torch.float64, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
        eng.local_energy(torch.as_tensor(synthetic_walkers(h, 2, seed=1)))
        ref = weakref.ref(eng)
        del eng
        assert ref() is None
    finally:
        gc.enable()


def test_return_mos_released_engines_and_numpy_spin_input():
    """Boundary details of the reference surface:
      * `ansatz.apply(..., return_mos=True)` (wf/nn_wave_function.py:127,144-145) returns the orbital matrices the determinants
        are taken of, `(orb_up, orb_down)`: their stacked slogdet reproduces log|psi| of a one-determinant, cusp-less
        ansatz, and for K determinants the CI sum of the value path;
      * an engine whose context was released raises a clear "engine closed" error instead of the library's "null argument";
      * an evicted engine that nobody holds is freed by reference count (no getrefcount heuristics), one that a caller
        holds keeps working;
      * the per-walker branch of the stochastic spin-raising operator accepts NumPy walkers and a geometry per walker."""
    import weakref
    from deepqmc_amd.physics import make_stochastic_spin_raising_operator
    h, wf = make()
    params = wf.init(3, perturb_envelopes=0.1)
    B = 3
    r = torch.as_tensor(synthetic_walkers(h, B, seed=4))
    psi = wf.apply(params, r)
    up, dn = wf.apply(params, r, return_mos=True)
    K, N = wf.spec.n_determinants, h.n_elec
    assert up.shape == (B, K, h.n_up, N) and dn.shape == (B, K, h.n_down, N)
    sgn, logdet = torch.linalg.slogdet(torch.cat([up, dn], dim=2))                 # [B, K]
    # orientation (a determinant does not see a transposition): rows are ELECTRONS, columns orbitals -- electron i's row against
    # the oracle's orbital matrix A[k, i, :] of the same walker
    from oracle import wf as owf_
    from oracle import geom as ogeom_
    A_ref, _ = owf_.orbitals(owf_.to_torch(params), wf.spec, torch.as_tensor(r[0], dtype=torch.float64), torch.as_tensor(h.mol.coords, dtype=torch.float64), h.n_up, wf.norm_eps if hasattr(wf, 'norm_eps') else ogeom_.F64_EPS)
    np.testing.assert_allclose(torch.cat([up, dn], dim=2)[0].numpy(), A_ref.detach().numpy(), rtol=1e-9, atol=1e-12)
    # a caller's own choice of the value-path kernels survives the call (it used to be reset to the default)
    eng_ = wf.engine(params)
    eng_.set_option('fused', 2)
    wf.apply(params, r, return_mos=True)
    assert eng_.get_option('fused', 1) == 2
    eng_.set_option('fused', 1)
    eng = wf.engine(params)
    np.testing.assert_allclose(logdet.numpy(), eng.debug_read('logdet', B)[:, :, 0], rtol=1e-10, atol=1e-10)
    np.testing.assert_array_equal(sgn.numpy().astype(np.int32), eng.debug_read('sign_k', B))
    assert torch.equal(wf.apply(params, r).log, psi.log)                           # the value path is back on its default kernels
    # closed engines say so
    held = wf.engine(params)
    wf.release()
    with pytest.raises(DqmcError, match='engine closed'):
        held.wf_eval(r)
    # eviction: freed by reference count when nobody holds the engine, alive while somebody does
    wf.max_engines = 1
    t1, t2, t3 = (wf.init(s, perturb_envelopes=0.1) for s in (1, 2, 3))
    wf._geometry_key = lambda R: None if R is None else np.asarray(R, np.float64).tobytes()      # (one context per (tree, geometry))
    g = [h.mol.coords, h.mol.coords * 1.1, h.mol.coords * 1.2]
    gc.collect(); gc.disable()
    try:
        w1 = weakref.ref(wf.engine(t1, g[0]))
        keep = wf.engine(t2, g[1])                         # evicts the first: nobody holds it -> gone now
        assert w1() is None
        wf.engine(t3, g[2])                                # evicts `keep` from the cache; the caller still holds it
        assert keep._ctx is not None
        keep.wf_eval(r, torch.as_tensor(g[1]))
    finally:
        gc.enable()
    # spin-raising operator: NumPy walkers, a geometry per walker, per-walker electron indices
    h2, wf2 = make()
    p2 = wf2.init(2, perturb_envelopes=0.1)
    op = make_stochastic_spin_raising_operator(h2, wf2)
    r_np = synthetic_walkers(h2, B, seed=6)
    idx = torch.tensor([h2.n_up, h2.n_up + h2.n_down - 1, h2.n_up])
    R_b = np.broadcast_to(h2.mol.coords, (B, h2.n_nuc, 3)).copy()
    a = op(p2, r_np, idx)
    b = op(p2, PhysicalConfiguration(torch.as_tensor(R_b), torch.as_tensor(r_np), None), idx)
    c = torch.stack([op(p2, torch.as_tensor(r_np[k:k + 1]), int(idx[k]))[0] for k in range(B)])
    np.testing.assert_allclose(a.numpy(), c.numpy(), rtol=1e-12)
    np.testing.assert_allclose(b.numpy(), c.numpy(), rtol=1e-12)


def test_float64_tail_of_a_float32_pass():
    """The float64 TAIL of a float32 forward-Laplacian pass (engine.hip, above tail_f64; engine_refine.inl: run_tail): the ops
    from the backflow head on -- the LINEAR ops that write what ORBITALS reads, ORBITALS, SLOGDET, FINAL -- run on the float64 twin
    for every walker of an unchunked pass, reading the float32 head's activations where they lie.  Checked here, through the
    emulator: WHICH kernels run where (timing records of the context and of its twin), that the results are float32-class
    results of the same quantity (against the float64 engine), that dqmc_debug_read finds the tail's buffers in the twin, that
    a chunked pass and a context with the option off keep the plain float32 tail, and that flags / scores come out the same way."""
    import dataclasses
    from deepqmc_amd.params import init_params
    from deepqmc_amd.spec import paulinet
    h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
    spec = dataclasses.replace(paulinet(), embedding_dim=32, n_interactions=1, n_determinants=4)
    params = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=5, perturb_envelopes=0.1)
    B = 16
    r32 = synthetic_walkers(h, B, seed=9).astype(np.float32)
    r = torch.as_tensor(r32)
    e64 = Engine(spec, h, params, dtype=torch.float64, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
    ref, st64, g64 = e64.local_energy(torch.as_tensor(r32.astype(np.float64)), return_grad=True)
    orb64 = e64.debug_read('orbitals', B)

    def engine(tail):
        e = Engine(spec, h, params, dtype=torch.float32, device='cpu', lib=emu_lib(), norm_eps=geom.F32_EPS)
        e.set_option('refine', 0)
        e.set_option('tail_f64', tail)
        return e
    on, off = engine(1), engine(0)
    out = {}
    for name, e in (('on', on), ('off', off)):
        e.timing(True); e.timing_reset()
        out[name] = e.local_energy(r, return_grad=True)
        rep = e.timing_report()
        e.timing(False)
        n32 = {k: rep.get(k, {'launches': 0})['launches'] for k in ('orbitals', 'slogdet', 'final', 'tail')}
        n64 = {k: rep.get('f64.' + k, {'launches': 0})['launches'] for k in ('linear', 'orbitals', 'slogdet', 'final')}
        if name == 'on':
            assert n32 == {'orbitals': 0, 'slogdet': 0, 'final': 0, 'tail': 2}, n32          # widen + narrow; the head stops before the backflow head
            assert n64['orbitals'] == 1 and n64['slogdet'] == 1 and n64['final'] == 1 and n64['linear'] >= 1, n64
        else:
            assert n32 == {'orbitals': 1, 'slogdet': 1, 'final': 1, 'tail': 0} and not any(n64.values()), (n32, n64)
    rel = {k: np.abs(v[0].double().numpy() - ref.numpy()) / np.maximum(1.0, np.abs(ref.numpy())) for k, v in out.items()}
    assert np.median(rel['on']) < 5e-6 and np.median(rel['off']) < 5e-6, (rel['on'], rel['off'])      # both float32-class evaluations
    for key in ('hamil/E_kin', 'hamil/V_el', 'hamil/V_loc'):
        np.testing.assert_allclose(out['on'][1][key].double().numpy(), st64[key].numpy(), rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(out['on'][2].double().numpy(), g64.numpy(), rtol=5e-3, atol=5e-2)
    # the potential terms do not pass through the network: the float64 k_final gives them to float32 OUTPUT rounding
    np.testing.assert_allclose(out['on'][1]['hamil/V_el'].double().numpy(), st64['hamil/V_el'].numpy(), rtol=2e-7)
    # the tail's buffers live in the twin: the Slater matrices read back are float64 values built on a float32 head
    orb_on = on.debug_read('orbitals', B)
    assert np.abs(orb_on[:, :, 0] - orb64[:, :, 0]).max() < 1e-4 * np.abs(orb64[:, :, 0]).max()
    assert (orb_on[:, :, 0].astype(np.float32).astype(np.float64) != orb_on[:, :, 0]).any()      # (float64 values, not float32-representable ones)
    # a chunked pass keeps the float32 tail (the hand-over is for unchunked passes), same results as the option off
    small = engine(1)
    per_walker = 4 * sum(rows * 16 * width for rows, width in small.program.bufs)
    small.set_option('ws_budget_mb', max(1, int(3.5 * per_walker / 2 ** 20)))
    e_c, _ = small.local_energy(r)
    if small.last_chunks()['own'] > 1:
        np.testing.assert_array_equal(e_c.numpy(), out['off'][0].numpy())
    # flags and scores through the twin's k_final: same count, same walkers as with a float32 tail up to round-off in the score
    fl_on, fl_off = engine(1), engine(0)
    for e in (fl_on, fl_off):
        e.set_option('refine', 1); e.set_option('refine_probe', 0); e.set_option('refine_thresh', 10 ** 9)
        e.local_energy(r)
    s_on, s_off = fl_on.refine_scores(B), fl_off.refine_scores(B)
    np.testing.assert_allclose(s_on, s_off, rtol=5e-2)
    thr = int(np.sort(s_on)[B - 4]) + 1
    fl_on.set_option('refine_thresh', thr)
    e_ref, _ = fl_on.local_energy(r)
    assert fl_on.last_refined() == int((s_on > thr).sum())
    above = s_on > thr
    np.testing.assert_allclose(e_ref.numpy()[above], ref.numpy()[above], rtol=3e-7)                   # refined walkers: the full float64 pass (the twin sees the float32-rounded geometry)
    np.testing.assert_array_equal(e_ref.numpy()[~above], out['on'][0].numpy()[~above])                 # kept walkers: float32 head + float64 tail
