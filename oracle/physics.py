"""Local energy and its parts (oracle; test infrastructure only).

Restates reference physics.py:79-156 (kinetic term, Coulomb terms, the
reverse-forward Laplacian loop) and hamil.py:156-184 (`loc_ene`).  The Laplacian of
log|psi| is available three independent ways: `laplacian_hessian` (torch.func
forward-over-reverse Hessian), `laplacian_loop` (the literal loop of physics.py:144-156)
and `laplacian_fd` (central finite differences).
"""
from __future__ import annotations

from typing import Callable

import torch
from torch.func import grad, hessian, jvp

from . import geom
from . import wf as owf


def nuclear_energy(R: torch.Tensor, ns_valence: torch.Tensor) -> torch.Tensor:
    """physics.py:112-116."""
    if R.shape[0] < 2:
        return R.new_zeros(())
    zz = geom.triu_flat(ns_valence[:, None] * ns_valence)
    i, j = geom.triu_indices(R.shape[0])
    # pairwise_self_distance of the nuclei uses the safe norm with the dtype's eps;
    # nuclei are never coincident so eps is immaterial at the 1e-16 level
    d = geom.pairwise_self_distance(R)
    return (zz / d).sum()


def electronic_potential(r: torch.Tensor, eps: float) -> torch.Tensor:
    """physics.py:119-121 (safe norm)."""
    return (1 / geom.pairwise_self_distance(r, eps=eps)).sum(-1)


def local_potential(r: torch.Tensor, R: torch.Tensor, charges: torch.Tensor) -> torch.Tensor:
    """NuclearCoulombPotential.local_potential, physics.py:131-133 (plain norm)."""
    return -(charges / geom.pairwise_distance(r, R)).sum(dim=(-1, -2))


def laplacian_hessian(f: Callable, x: torch.Tensor):
    g = grad(f)(x)
    h = hessian(f)(x)
    return torch.diagonal(h).sum(), g


def laplacian_loop(f: Callable, x: torch.Tensor):
    """Literal physics.py:144-156: grad, then one JVP of the gradient per coordinate."""
    gf = grad(f)
    df = gf(x)
    eye = torch.eye(x.shape[0], dtype=x.dtype)
    acc = x.new_zeros(())
    for i in range(x.shape[0]):
        _, t = jvp(gf, (x,), (eye[i],))
        acc = acc + t[i]
    return acc, df


def laplacian_fd(f: Callable, x: torch.Tensor, h: float = 1e-4):
    f0 = f(x)
    lap = x.new_zeros(())
    g = torch.zeros_like(x)
    for i in range(x.shape[0]):
        e = torch.zeros_like(x)
        e[i] = h
        fp, fm = f(x + e), f(x - e)
        lap = lap + (fp - 2 * f0 + fm) / h ** 2
        g[i] = (fp - fm) / (2 * h)
    return lap, g


def local_energy(params, spec, r: torch.Tensor, R: torch.Tensor, charges: torch.Tensor, n_up: int,
                 eps: float = geom.F64_EPS, laplacian=laplacian_hessian):
    """`loc_ene` of hamil.py:160-182 for one walker (Coulomb potential, V_nl = 0).
    Returns (E_loc, stats, (sign, log|psi|))."""
    def logpsi(flat):
        return owf.wave_function(params, spec, flat.reshape(-1, 3), R, n_up, eps)[1]

    lap, qf = laplacian(logpsi, r.reshape(-1))
    qf2 = (qf ** 2).sum()
    e_kin = -0.5 * (lap + qf2)                                           # physics.py:108
    e_nuc = nuclear_energy(R, charges)
    v_el = electronic_potential(r, eps)
    v_loc = local_potential(r, R, charges)
    v_nl = r.new_zeros(())
    e_loc = e_kin + v_loc + v_nl + v_el + e_nuc                          # hamil.py:172
    stats = {
        'hamil/V_el': v_el, 'hamil/E_kin': e_kin, 'hamil/V_loc': v_loc, 'hamil/V_nl': v_nl,
        'hamil/lap': lap, 'hamil/quantum_force': qf2,
    }
    return e_loc, stats, qf


def batch_wave_function(params, spec, r: torch.Tensor, R: torch.Tensor, n_up: int, eps: float):
    """vmap(wf) of sampling/electron_samplers.py:76-81 as a plain loop."""
    signs, logs = [], []
    with torch.no_grad():
        for b in range(r.shape[0]):
            s, l = owf.wave_function(params, spec, r[b], R, n_up, eps)
            signs.append(s)
            logs.append(l)
    return torch.stack(signs), torch.stack(logs)


def batch_local_energy(params, spec, r: torch.Tensor, R: torch.Tensor, charges, n_up: int, eps: float,
                       laplacian=laplacian_hessian):
    """compute_local_energy's walker vmap (loss/energy.py:50-57) as a plain loop."""
    es, stats, qfs = [], [], []
    for b in range(r.shape[0]):
        e, st, qf = local_energy(params, spec, r[b], R, charges, n_up, eps, laplacian)
        es.append(e.detach())
        stats.append({k: v.detach() for k, v in st.items()})
        qfs.append(qf.detach())
    out = {k: torch.stack([s[k] for s in stats]) for k in stats[0]}
    return torch.stack(es), out, torch.stack(qfs)


def evaluate_spin(params, spec, r: torch.Tensor, R: torch.Tensor, n_up: int, n_down: int, eps: float) -> torch.Tensor:
    """Reference physics.py:159-226 for ONE walker r[N,3], read literally: the constant term, then for every down
    electron beta and every up electron alpha the ratio of the wave function with the two positions exchanged."""
    umd = n_up - n_down
    s2 = torch.tensor(umd / 2 * (umd / 2 + 1) + n_down, dtype=torch.float64)        # physics.py:168
    s0, l0 = owf.wave_function(params, spec, r, R, n_up, eps)                       # physics.py:171
    for down_idx in range(n_up, n_up + n_down):                                     # physics.py:174-176
        for up_idx in range(n_up):                                                  # physics.py:224-226
            perm = list(range(r.shape[0]))                                          # physics.py:206-211
            perm[down_idx], perm[up_idx] = up_idx, down_idx
            sp, lp = owf.wave_function(params, spec, r[perm], R, n_up, eps)
            s2 = s2 - s0 * sp * torch.exp(lp - l0)                                  # physics.py:218-222
    return s2
