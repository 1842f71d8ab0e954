"""Metropolis / decorrelated sampling and the energy reduction (oracle; test infra only).

Restates reference sampling/electron_samplers.py:102-163 (`_proposal`, `_acc_log_prob`,
`_accept`, `compute_stats`), :347-357 (`DecorrSampler.sample`) and the cross-device
mean/std/min/max of observable.py:474-479 + parallel.py:175-225.  Random numbers are
explicit inputs (`noise` ~ N(0,1) [n_sub,B,N,3], `unif` ~ U[0,1) [n_sub,B]) so the HIP path
can be compared bit-for-bit on the accept decisions.
"""
from __future__ import annotations

import math

import torch

from . import geom
from .physics import batch_wave_function


def metropolis_step(params, spec, state: dict, R, n_up, eps, noise, unif, max_age=None, target_acceptance=0.57,
                    psi_fn=None):
    """One MetropolisSampler.sample (electron_samplers.py:140-152).  `state` holds
    r[B,N,3], sign[B], log[B], age[B] (int), tau (float)."""
    r, tau = state['r'], state['tau']
    r_prop = r + tau * noise                                            # :102-104
    if psi_fn is not None:          # any batched wave function r[B,N,3] -> (sign[B], log[B])
        sign_p, log_p = psi_fn(r_prop)
    else:
        sign_p, log_p = batch_wave_function(params, spec, r_prop, R, n_up, eps)
    log_prob = 2 * (log_p - state['log'])                               # :106-107
    accepted = log_prob > torch.log(unif)                               # :118 (NaN -> False)
    if max_age is not None:
        accepted = accepted | (state['age'] >= max_age)                 # :119-120
    acceptance = accepted.to(torch.float64).sum() / accepted.shape[0]   # :121
    if target_acceptance is not None:                                   # :122-126
        new_tau = tau / (target_acceptance / max(float(acceptance), 0.05))
    else:
        new_tau = tau
    age = torch.where(accepted, torch.zeros_like(state['age']), state['age'] + 1)   # :127-128
    new = {
        'r': torch.where(accepted[:, None, None], r_prop, r),
        'sign': torch.where(accepted, sign_p, state['sign']),
        'log': torch.where(accepted, log_p, state['log']),
        'age': age,
        'tau': new_tau,
    }
    return new, accepted, float(acceptance)


def sampler_stats(state: dict, acceptance: float, eps: float):
    """compute_stats (electron_samplers.py:154-163)."""
    return {
        'sampling/acceptance': acceptance,
        'sampling/tau': state['tau'],
        'sampling/age/mean': float(state['age'].to(torch.float64).mean()),
        'sampling/age/max': int(state['age'].max()),
        'sampling/log_psi/mean': float(state['log'].mean()),
        'sampling/log_psi/std': float(state['log'].std(unbiased=False)),
        'sampling/dists/mean': float(geom.pairwise_self_distance(state['r'], eps=eps).mean()),
    }


def decorr_sample(params, spec, state, R, n_up, eps, noise, unif, max_age=None, target_acceptance=0.57, psi_fn=None):
    """DecorrSampler.sample (electron_samplers.py:347-357): `length` = noise.shape[0]
    Metropolis sub-steps, stats of the last one.  Returns (state, stats, accept[n_sub,B])."""
    acc_hist = []
    acceptance = 0.0
    for k in range(noise.shape[0]):
        state, accepted, acceptance = metropolis_step(
            params, spec, state, R, n_up, eps, noise[k], unif[k], max_age, target_acceptance, psi_fn)
        acc_hist.append(accepted)
    return state, sampler_stats(state, acceptance, eps), torch.stack(acc_hist)


def energy_stats(e_loc: torch.Tensor):
    """EnergyMonitor (observable.py:474-479): mean, population std (two-pass,
    parallel.py:215-225), min, max over the walker axis."""
    mean = e_loc.mean()
    return {
        'local_energy/mean': float(mean),
        'local_energy/std': float(torch.sqrt(((e_loc - mean) ** 2).mean())),
        'local_energy/min': float(e_loc.min()),
        'local_energy/max': float(e_loc.max()),
    }
