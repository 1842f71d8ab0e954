"""Electron samplers: `MetropolisSampler` and `DecorrSampler` with the reference's
`init / sample / update` surface (src/deepqmc/sampling/base.py:14-79,
sampling/electron_samplers.py:32-173,333-357).  State is a dict of device tensors
{'r','psi': Psi(sign, log),'age','tau'}; the propose / evaluate / accept loop runs inside
`dqmc_mcmc_steps` on the GPU.  `rng` is an integer seed for the device Philox generator
(or pass explicit `noise`/`unif` tensors for bit-reproducible parity runs).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .types import PhysicalConfiguration, Psi


def synthetic_walkers(hamil, n: int, seed: int = 1, std: float = 1.0) -> np.ndarray:
    """Atom-centred Gaussian walkers r = R[nucleus] + N(0, std): electrons assigned to nuclei
    in proportion to charge (BASELINE.md "Inputs").  The reference's shell-based initialiser
    (sampling/electron_sample_initializers.py) is out of scope."""
    rng = np.random.default_rng(seed)
    mol = hamil.mol
    centers = np.repeat(np.arange(len(mol.charges)), mol.charges.astype(int))[:hamil.n_elec]
    return mol.coords[centers][None] + std * rng.standard_normal((n, hamil.n_elec, 3))


class MetropolisSampler:
    """electron_samplers.py:32-173 with length-1 decorrelation."""

    length = 1

    def __init__(self, hamil, wf, *, tau: float = 1.0, target_acceptance: Optional[float] = 0.57,
                 max_age: Optional[int] = None, sample_initializer=synthetic_walkers):
        self.hamil, self.wf = hamil, wf
        self.initial_tau, self.target_acceptance, self.max_age = tau, target_acceptance, max_age
        self.sample_initializer = sample_initializer

    def phys_conf(self, R, r) -> PhysicalConfiguration:
        """electron_samplers.py:165-173."""
        n = r.shape[0]
        return PhysicalConfiguration(R, r, torch.zeros(n, dtype=torch.int32, device=r.device))

    def update(self, state, params, R=None):
        """electron_samplers.py:76-84: recompute psi at the current positions."""
        psi = self.wf.apply(params, PhysicalConfiguration(R, state['r'], None) if R is not None else state['r'])
        return {**state, 'psi': psi}

    def init(self, rng, params, n: int, R=None):
        """electron_samplers.py:86-100."""
        eng = self.wf.engine(params)
        r = torch.as_tensor(self.sample_initializer(self.hamil, n, seed=int(rng)), dtype=eng.dtype, device=eng.device)
        state = {'r': r.contiguous(), 'age': torch.zeros(n, dtype=torch.int32, device=eng.device),
                 'tau': torch.full((1,), self.initial_tau, dtype=eng.dtype, device=eng.device)}
        return self.update(state, params, R)

    def sample(self, rng, state, params, R=None, noise=None, unif=None):
        """electron_samplers.py:140-152 / :347-357.  Returns (state, phys_conf, stats)."""
        eng = self.wf.engine(params)
        # the reference's samplers are functional (a new state per call); dqmc_mcmc_steps updates its arguments
        # in place, so it gets copies and the caller's previous state stays intact (rollback, multi-state loops)
        st = {'r': state['r'].clone(), 'log': state['psi'].log.clone(), 'sign': state['psi'].sign.clone(),
              'age': state['age'].clone(), 'tau': state['tau'].clone()}
        stats = eng.mcmc_steps(st, self.length, max_age=self.max_age, target_acceptance=self.target_acceptance,
                               seed=int(rng), noise=noise, unif=unif, R=R)
        state = {'r': st['r'], 'psi': Psi(st['sign'], st['log']), 'age': st['age'], 'tau': st['tau']}
        return state, self.phys_conf(eng.R if R is None else R, state['r']), stats


class DecorrSampler(MetropolisSampler):
    """electron_samplers.py:333-357 chained in front of MetropolisSampler
    (sampling_utils.py:31-54): `length` sub-steps per sample, stats of the last."""

    def __init__(self, hamil, wf, *, length: int = 30, **kw):
        super().__init__(hamil, wf, **kw)
        self.length = length


class MultiElectronicStateSampler:
    """sampling/combined_samplers.py:58-90: one independent Markov chain ensemble per electronic
    state, each with its own parameter set.  The reference vmaps the sampler over the state axis;
    here the states are a host loop over HIP contexts.  States/stats gain a leading list axis."""

    def __init__(self, sampler, n_state: int):
        self.sampler, self.n_state = sampler, n_state

    def init(self, rng, params, electron_batch_size: int, R=None):
        return [self.sampler.init(int(rng) * self.n_state + s, params[s], electron_batch_size, R)
                for s in range(self.n_state)]

    def sample(self, rng, state, params, R=None):
        out = [self.sampler.sample(int(rng) * self.n_state + s, state[s], params[s], R) for s in range(self.n_state)]
        states, pcs, stats = zip(*out)
        r = torch.stack([pc.r for pc in pcs])
        return list(states), PhysicalConfiguration(pcs[0].R, r, torch.zeros(r.shape[:2], dtype=torch.int32, device=r.device)), \
            {k: [st[k] for st in stats] for k in stats[0]}

    def update(self, state, params, R=None):
        return [self.sampler.update(state[s], params[s], R) for s in range(self.n_state)]


class LangevinSampler(MetropolisSampler):
    """Metropolis-adjusted Langevin sampler (sampling/electron_samplers.py:176-232): drift = cleaned grad log|psi|
    (sampling_utils.py:72-101) from one forward-Laplacian pass per proposal, Green's-function ratio in the acceptance.
    Everything -- drift cleaning, proposal, acceptance, state selection, step-size adaptation -- runs in the HIP
    library (`dqmc_langevin_update` / `dqmc_langevin_steps`); the state carries the drift as 'force'."""

    def update(self, state, params, R=None):
        eng = self.wf.engine(params)
        sign, log, force = eng.langevin_update(state['r'], state['tau'], self.hamil.mol.charges, R)
        return {**state, 'psi': Psi(sign, log), 'force': force}

    def sample(self, rng, state, params, R=None, noise=None, unif=None):
        eng = self.wf.engine(params)
        st = {'r': state['r'].clone(), 'log': state['psi'].log.clone(), 'sign': state['psi'].sign.clone(),
              'age': state['age'].clone(), 'tau': state['tau'].clone(), 'force': state['force'].clone()}
        stats = eng.langevin_steps(st, self.length, self.hamil.mol.charges, max_age=self.max_age,
                                   target_acceptance=self.target_acceptance, seed=int(rng), noise=noise, unif=unif, R=R)
        state = {'r': st['r'], 'psi': Psi(st['sign'], st['log']), 'age': st['age'], 'tau': st['tau'], 'force': st['force']}
        return state, self.phys_conf(eng.R if R is None else R, state['r']), stats


class OppositeSpinExchangeSampler:
    """sampling/electron_samplers.py:235-330 chained in front of a Metropolis-type sampler: with probability
    `exchange_step_probability` a step proposes swapping the positions of one random spin-up and one random
    spin-down electron per walker (uniform choice = the reference's default zero logits) and accepts with
    |psi'|^2 / |psi|^2 -- no age override and no step-size adaptation on such steps (:312-321) -- otherwise it
    is an ordinary step of the wrapped sampler.  Swap, psi evaluation, acceptance and state update run in the HIP
    library (`dqmc_exchange_step`); only the random choices are drawn on the host.  `choices` = (is_exchange: bool,
    up_idx[B], down_idx[B], unif[B]) overrides the draws (parity tests)."""

    def __init__(self, sampler: MetropolisSampler, *, exchange_step_probability: float):
        self.sampler = sampler
        self.exchange_step_probability = exchange_step_probability
        self.hamil, self.wf = sampler.hamil, sampler.wf

    def init(self, rng, params, n, R=None):
        return self.sampler.init(rng, params, n, R)

    def update(self, state, params, R=None):
        return self.sampler.update(state, params, R)

    def sample(self, rng, state, params, R=None, choices=None, **kw):
        eng = self.wf.engine(params)
        B, n_up, n_down = state['r'].shape[0], self.hamil.n_up, self.hamil.n_down
        gen = torch.Generator(device='cpu')
        gen.manual_seed(int(rng) * 7919 + 13)
        if choices is None:
            is_ex = bool(torch.rand((), generator=gen) < self.exchange_step_probability)
            up_idx = torch.randint(0, n_up, (B,), generator=gen)
            down_idx = torch.randint(0, n_down, (B,), generator=gen)
            unif = torch.rand(B, generator=gen, dtype=torch.float64)
        else:
            is_ex, up_idx, down_idx, unif = choices
        if not is_ex:
            return self.sampler.sample(rng, state, params, R, **kw)
        st = {'r': state['r'].clone(), 'log': state['psi'].log.clone(), 'sign': state['psi'].sign.clone(),
              'age': state['age'].clone(), 'tau': state['tau']}
        stats = eng.exchange_step(st, up_idx, down_idx, unif, R=R)
        new = {**state, 'r': st['r'], 'psi': Psi(st['sign'], st['log']), 'age': st['age']}
        if 'force' in state:          # a Langevin state: the drift belongs to the positions
            new = self.sampler.update(new, params, R)
        return new, self.sampler.phys_conf(eng.R if R is None else R, new['r']), stats
