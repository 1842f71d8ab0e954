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


def synthetic_walkers(hamil, n: int, seed: int = 1, std: float = 1.0, R=None) -> np.ndarray:
    """Atom-centred Gaussian walkers r = R[nucleus] + N(0, std): electrons assigned to nuclei
    in proportion to charge (BASELINE.md "Inputs"), around the geometry `R` when one is given
    (the reference hands R to its initialiser, electron_samplers.py:86-100), else around the
    Hamiltonian's own.  The reference's shell-based initialiser
    (sampling/electron_sample_initializers.py) is out of scope."""
    rng = np.random.default_rng(seed)
    mol = hamil.mol
    centers = np.repeat(np.arange(len(mol.charges)), mol.charges.astype(int))[:hamil.n_elec]
    if R is None:
        coords = mol.coords
    else:
        coords = np.asarray(R.detach().cpu() if hasattr(R, 'detach') else R, np.float64).reshape(-1, len(mol.charges), 3)[0]
    return coords[centers][None] + std * rng.standard_normal((n, hamil.n_elec, 3))


class MetropolisSampler:
    """electron_samplers.py:32-173 with length-1 decorrelation."""

    length = 1

    def __init__(self, hamil, wf, *, tau: float = 1.0, target_acceptance: Optional[float] = 0.57,
                 max_age: Optional[int] = None, sample_initializer=synthetic_walkers, in_place: bool = False):
        """`in_place`: `sample` advances the tensors of the state it is given instead of copies of them (no five
        device-to-device clones per call; the state passed in is then NOT a snapshot of the previous step -- for loops
        that never roll back, e.g. bench.py)."""
        self.hamil, self.wf = hamil, wf
        self.initial_tau, self.target_acceptance, self.max_age = tau, target_acceptance, max_age
        self.sample_initializer = sample_initializer
        self.in_place = in_place

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
        eng = self.wf.engine(params, R)
        kw = {} if R is None else {'R': R}       # initialisers written for round 1 (no R argument) keep working at the default geometry
        r = torch.as_tensor(self.sample_initializer(self.hamil, n, seed=int(rng), **kw), dtype=eng.dtype, device=eng.device)
        state = {'r': r.contiguous(), 'age': torch.zeros(n, dtype=torch.int32, device=eng.device),
                 'tau': torch.full((1,), self.initial_tau, dtype=eng.dtype, device=eng.device)}
        return self.update(state, params, R)

    def sample(self, rng, state, params, R=None, noise=None, unif=None):
        """electron_samplers.py:140-152 / :347-357.  Returns (state, phys_conf, stats)."""
        eng = self.wf.engine(params, R)
        # the reference's samplers are functional (a new state per call); dqmc_mcmc_steps updates its arguments
        # in place, so it gets copies and the caller's previous state stays intact (rollback, multi-state loops)
        if self.in_place:
            st = {'r': state['r'], 'log': state['psi'].log, 'sign': state['psi'].sign, 'age': state['age'], 'tau': state['tau']}
        else:
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


class IdleNucleiSampler:
    """sampling/nuclei_samplers.py:15-37: keeps the nuclei where they are."""

    def __init__(self, charges=None):
        pass

    def init(self, nuc_coords, *args, **kw):
        return {'R': nuc_coords}

    def sample(self, rng, state):
        return state, torch.zeros_like(torch.as_tensor(state['R'])), {}


class MoleculeIdxSampler:
    """sampling/combined_samplers.py:17-56: cycles through the molecule indices in batches of `batch_size`,
    optionally shuffled once or at every pass (NumPy permutation stream instead of the JAX one)."""

    def __init__(self, rng, n_mols: int, batch_size: int, shuffle=False):
        assert shuffle in (False, 'once', 'always')
        self.rng = np.random.default_rng(int(rng))
        self.n_mols, self.batch_size, self.shuffle, self.state = n_mols, batch_size, shuffle, 0
        self._first = self.rng.permutation(n_mols)
        self.permutation = self.new_permutation()

    def new_permutation(self):
        if not self.shuffle:
            return np.arange(self.n_mols)
        return self._first if self.shuffle == 'once' else self.rng.permutation(self.n_mols)

    def sample(self):
        idx = np.arange(self.state, min(self.state + self.batch_size, self.n_mols))
        value = [self.permutation[idx]]
        if len(idx) < self.batch_size:
            self.permutation = self.new_permutation()
            value.append(self.permutation[np.arange(self.batch_size - len(idx))])
        self.state = (self.state + self.batch_size) % self.n_mols
        return np.concatenate(value)


class MultiNuclearGeometrySampler:
    """sampling/combined_samplers.py:93-214: electron sampler states for a set of nuclear geometries; `sample` works
    on the molecules named by `mol_idxs` (gather -> optional nuclear update / electron warp / re-equilibration ->
    electron sampling -> scatter back) and labels the samples with their molecule index.  The molecule axis is a host
    loop: the geometry is an argument of every HIP call (ansatzes with nuclear tokens get one context per geometry),
    the walkers of one (molecule, state) are the GPU batch.  State: {'nuc': [per molecule], 'elec': [per molecule]
    (whatever `elec_sampler.init` returns), 'update_nuc_counter': int array}."""

    def __init__(self, elec_sampler, nuc_sampler=None, warp_elec_fn=None, update_nuc_period=None,
                 elec_equilibration_steps=None):
        self.elec_sampler = elec_sampler
        self.nuc_sampler = nuc_sampler if nuc_sampler is not None else IdleNucleiSampler()
        self.warp_elec_fn = warp_elec_fn
        self.update_nuc_period = update_nuc_period
        self.elec_equilibration_steps = elec_equilibration_steps

    def init(self, rng, params, electron_batch_size: int, R):
        R = torch.as_tensor(R)
        M = R.shape[0]
        return {'nuc': [self.nuc_sampler.init(R[m]) for m in range(M)],
                'elec': [self.elec_sampler.init(int(rng) * M + m, params, electron_batch_size, R[m]) for m in range(M)],
                'update_nuc_counter': np.zeros(M, np.int64)}

    def update_nuc(self, rng, nuc, elec, params):
        """combined_samplers.py:131-160 for one molecule."""
        nuc, dR, stats = self.nuc_sampler.sample(int(rng), nuc)
        if self.warp_elec_fn is not None:
            elec = self.warp_elec_fn(int(rng) + 1, nuc['R'], dR, elec)
        elec = self.elec_sampler.update(elec, params, nuc['R'])
        for i in range(self.elec_equilibration_steps or 0):
            elec = self.elec_sampler.sample(int(rng) * 1009 + i, elec, params, nuc['R'])[0]
        return nuc, elec, stats

    def sample(self, rng, smpl_state, params, mol_idxs):
        mol_idxs = [int(m) for m in np.asarray(mol_idxs).reshape(-1)]
        counter = np.array(smpl_state['update_nuc_counter'], copy=True)      # functional state: the caller's stays as it was
        nuc, elec = list(smpl_state['nuc']), list(smpl_state['elec'])
        pcs, stats = [], []
        for k, m in enumerate(mol_idxs):
            if self.update_nuc_period is not None:
                if counter[m] == self.update_nuc_period - 1:
                    nuc[m], elec[m], _ = self.update_nuc(int(rng) * 7919 + m, nuc[m], elec[m], params)
                    counter[m] = 0
                else:
                    counter[m] += 1
            elec[m], pc, st = self.elec_sampler.sample(int(rng) * len(nuc) + m, elec[m], params, nuc[m]['R'])
            pcs.append(pc)
            stats.append(st)
        r = torch.stack([pc.r for pc in pcs])                                   # [M_batch, (S,) B, N, 3]
        Rb = torch.stack([torch.as_tensor(nuc[m]['R'], dtype=r.dtype, device=r.device) for m in mol_idxs])
        mol_idx = torch.as_tensor(mol_idxs, dtype=torch.int32, device=r.device).reshape((-1,) + (1,) * (r.dim() - 3)).expand(r.shape[:-2])
        new_state = {'nuc': nuc, 'elec': elec, 'update_nuc_counter': counter}
        return new_state, PhysicalConfiguration(Rb, r, mol_idx.contiguous()), {k: [st[k] for st in stats] for k in stats[0]}

    def update(self, smpl_state, params):
        smpl_state = dict(smpl_state)
        smpl_state['elec'] = [self.elec_sampler.update(e, params, n['R']) for e, n in zip(smpl_state['elec'], smpl_state['nuc'])]
        return smpl_state


class LangevinSampler(MetropolisSampler):
    """Metropolis-adjusted Langevin sampler (sampling/electron_samplers.py:176-232): drift = cleaned grad log|psi|
    (sampling_utils.py:72-101) from one forward-Laplacian pass per proposal, Green's-function ratio in the acceptance.
    Everything -- drift cleaning, proposal, acceptance, state selection, step-size adaptation -- runs in the HIP
    library (`dqmc_langevin_update` / `dqmc_langevin_steps`); the state carries the drift as 'force'."""

    def update(self, state, params, R=None):
        eng = self.wf.engine(params, R)
        sign, log, force = eng.langevin_update(state['r'], state['tau'], self.hamil.mol.charges, R)
        return {**state, 'psi': Psi(sign, log), 'force': force}

    def sample(self, rng, state, params, R=None, noise=None, unif=None):
        eng = self.wf.engine(params, R)
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
        eng = self.wf.engine(params, R)
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
