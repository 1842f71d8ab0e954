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


# ---- Langevin (MALA), reference sampling/electron_samplers.py:176-232 + sampling_utils.py:72-101 ----

def clean_force(force, r, R, charges, tau):
    """sampling_utils.py:72-101: damp the drift near nuclei (Umrigar-style crossover) and cap its
    length so that one step cannot overshoot the nearest nucleus."""
    z = r[:, :, None, :] - R[None, None, :, :]                         # [B,N,n_nuc,3]
    z2 = (z ** 2).sum(-1)
    idx = z2.argmin(-1)                                                # nearest nucleus
    zn = torch.gather(z, 2, idx[..., None, None].expand(-1, -1, 1, 3)).squeeze(2)
    z2n = torch.gather(z2, 2, idx[..., None]).squeeze(2)
    eps = torch.finfo(force.dtype).eps
    z_unit = zn / torch.linalg.norm(zn, dim=-1, keepdim=True)
    f_unit = force / torch.clamp(torch.linalg.norm(force, dim=-1, keepdim=True), min=eps)
    Z2z2 = charges[idx] ** 2 * z2n
    a = (1 + (f_unit * z_unit).sum(-1)) / 2 + Z2z2 / (10 * (4 + Z2z2))
    av2tau = a * (force ** 2).sum(-1) * tau
    factor = 2 / (torch.sqrt(1 + 2 * av2tau) + 1)
    force = factor[..., None] * force
    norm_factor = torch.clamp(torch.sqrt(z2n) / (tau * torch.clamp(torch.linalg.norm(force, dim=-1), min=eps)), max=1.0)
    return force * norm_factor[..., None]


def langevin_step(psi_force_fn, state, R, charges, noise, unif, max_age=None, target_acceptance=0.57):
    """One LangevinSampler.sample.  psi_force_fn(r[B,N,3]) -> (sign, log, grad log|psi| [B,N,3]).
    `state` additionally carries 'force' (cleaned with the tau of the step that produced it)."""
    r, tau = state['r'], state['tau']
    r_prop = r + tau * state['force'] + math.sqrt(tau) * noise                 # :214-221
    sign_p, log_p, g = psi_force_fn(r_prop)
    force_p = clean_force(g, r_prop, R, charges, tau)                          # :204-208 (tau of the previous iteration)
    log_G = ((state['force'] + force_p) * ((r - r_prop) + tau / 2 * (state['force'] - force_p))).sum(dim=(1, 2))   # :223-232
    log_prob = log_G + 2 * (log_p - state['log'])
    accepted = log_prob > torch.log(unif)
    if max_age is not None:
        accepted = accepted | (state['age'] >= max_age)
    acceptance = accepted.to(torch.float64).sum() / accepted.shape[0]
    new_tau = tau / (target_acceptance / max(float(acceptance), 0.05)) if target_acceptance is not None else tau
    sel = lambda a, b: torch.where(accepted.reshape((-1,) + (1,) * (a.dim() - 1)), a, b)
    new = {'r': sel(r_prop, r), 'sign': sel(sign_p, state['sign']), 'log': sel(log_p, state['log']),
           'force': sel(force_p, state['force']),
           'age': torch.where(accepted, torch.zeros_like(state['age']), state['age'] + 1), 'tau': new_tau}
    return new, accepted, float(acceptance)


# ---- opposite-spin exchange step, reference sampling/electron_samplers.py:235-330 ----

def spin_exchange_step(psi_fn, state, n_up: int, up_idx, down_idx, unif):
    """One exchange step: swap r[b, up_idx[b]] <-> r[b, n_up + down_idx[b]] (:277-288), accept with
    2 (log|psi'| - log|psi|) > log u (:290-293), `_accept` WITHOUT max_age / target_acceptance (:312-313)."""
    r = state['r']
    B = r.shape[0]
    bi = torch.arange(B)
    r_prop = r.clone()
    r_prop[bi, up_idx] = r[bi, n_up + down_idx]
    r_prop[bi, n_up + down_idx] = r[bi, up_idx]
    sign_p, log_p = psi_fn(r_prop)
    accepted = 2 * (log_p - state['log']) > torch.log(unif)
    new = {'r': torch.where(accepted[:, None, None], r_prop, r), 'sign': torch.where(accepted, sign_p, state['sign']),
           'log': torch.where(accepted, log_p, state['log']),
           'age': torch.where(accepted, torch.zeros_like(state['age']), state['age'] + 1), 'tau': state['tau']}
    return new, accepted
