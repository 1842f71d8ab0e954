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
        st = {'r': state['r'], 'log': state['psi'].log, 'sign': state['psi'].sign, 'age': state['age'], 'tau': state['tau']}
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
