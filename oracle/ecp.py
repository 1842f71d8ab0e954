"""Gaussian-type effective core potentials (oracle; test infra only).

Restates reference ecp/gaussian_type_ecp.py:127-159 (`local_potential`), :161-255
(`nonloc_potential`) and ecp/ecp_utils.py:23-75 (icosahedron quadrature, rotation of the quadrature
onto the electron-nucleus axis, random rotation about it) in float64 PyTorch.

PARITY UNPINNED for this file: the reference's ECP goldens (tests/test_potential/*.npz,
tests/test_hamil/test_local_energy_Molecular_PP_.npz) depend on the bfd / ccECP coefficient tables of
pyscf (`pyscf.gto.basis.load_ecp`, not vendored and not installed here) and on the electron initialiser,
so they cannot be reproduced offline.  The formulas are checked instead by properties (tests/test_ecp.py):
the first quadrature point is the electron itself, the 12-point rule integrates Legendre polynomials up
to degree 5 exactly, an s-type wave function sees only the l = 0 channel, a zero table reduces to the bare
Coulomb potential.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def unit_icosahedron():
    """ecp_utils.py:23-32 -> (points [12,3], polar angles [12])."""
    sph = [[0.0, 0.0], [math.pi, 0.0]]
    for j in range(5):
        sph.append([math.atan(2), math.pi / 5 * 2 * j])
        sph.append([math.pi - math.atan(2), math.pi / 5 * (2 * j - 1)])
    sph = np.array(sph)
    th, ph = sph[:, 0], sph[:, 1]
    pts = np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)], -1)   # sph2cart, :11-20
    return pts, th


def legendre_table(l_max_p1: int, thetas: np.ndarray) -> np.ndarray:
    """P_l(cos theta_q), [12, l_max_p1] (gaussian_type_ecp.py:192-198)."""
    x = np.cos(thetas)
    cols = [np.ones_like(x), x]
    for l in range(2, l_max_p1):
        cols.append(((2 * l - 1) * x * cols[-1] - (l - 1) * cols[-2]) / l)
    return np.stack(cols[:l_max_p1], -1)


def rot_y(t):
    c, s = torch.cos(t), torch.sin(t)
    z, o = torch.zeros_like(t), torch.ones_like(t)
    return torch.stack([torch.stack([c, z, s]), torch.stack([z, o, z]), torch.stack([-s, z, c])])


def rot_z(p):
    c, s = torch.cos(p), torch.sin(p)
    z, o = torch.zeros_like(p), torch.ones_like(p)
    return torch.stack([torch.stack([c, -s, z]), torch.stack([s, c, z]), torch.stack([z, z, o])])


def quadrature_points(r_i: torch.Tensor, R_a: torch.Tensor, phi_random: torch.Tensor, pts: torch.Tensor):
    """ecp_utils.py:35-63 for all 12 points: [12,3]."""
    d = r_i - R_a
    radius = torch.linalg.norm(d)
    theta = torch.arccos(torch.clamp(d[2] / radius, -1.0, 1.0))
    phi = torch.atan2(d[1], d[0])
    rot = rot_z(phi) @ rot_y(theta) @ rot_z(phi_random)
    return radius * (pts @ rot.T) + R_a


def local_potential(r, R, ns_valence, loc_params, ecp_mask):
    """gaussian_type_ecp.py:127-159.  r[N,3]; loc_params[n_nuc,3,2,n_terms] ([.,term,0,.] exponent,
    [.,term,1,.] coefficient; terms r^-1, r^0, r^1)."""
    dists = torch.linalg.norm(r[:, None, :] - R[None, :, :], dim=-1)              # [N, n_nuc]
    v = -(ns_valence[None, :] / dists).sum()
    for a in range(R.shape[0]):
        if not ecp_mask[a]:
            continue
        ra = dists[:, a, None]                                                     # [N,1]
        al, be = loc_params[a, :, 0, :], loc_params[a, :, 1, :]
        v = v + (be[0] / ra * torch.exp(-al[0] * ra ** 2)).sum()
        v = v + (be[1] * torch.exp(-al[1] * ra ** 2)).sum()
        v = v + (be[2] * ra * torch.exp(-al[2] * ra ** 2)).sum()
    return v


def nonloc_potential(r, R, nl_params, psi_fn, phi_random):
    """gaussian_type_ecp.py:161-255.  nl_params[n_nuc,l_max_p1,2,n_terms]; psi_fn(r[M,N,3]) -> (sign[M],
    log[M]); phi_random[n_nl_nuc, N] = the reference's uniform(fold_in(fold_in(rng, j), i), 0, pi/5)."""
    pts_np, th = unit_icosahedron()
    pts = torch.as_tensor(pts_np, dtype=r.dtype)
    nuc_with_nl = [a for a in range(nl_params.shape[0]) if bool((nl_params[a] != 0).any())]     # :121
    if not nuc_with_nl:
        return torch.zeros((), dtype=r.dtype)
    sign0, log0 = psi_fn(r[None])
    l_max_p1 = nl_params.shape[1]
    P = torch.as_tensor(legendre_table(l_max_p1, th), dtype=r.dtype)               # [12, l]
    coef = (torch.arange(l_max_p1, dtype=r.dtype) * 2 + 1) / 12                    # (2l+1)/12
    total = torch.zeros((), dtype=r.dtype)
    for j, a in enumerate(nuc_with_nl):
        d = torch.linalg.norm(r - R[a], dim=-1)                                    # [N]
        v_l = (nl_params[a, None, :, 1, :] * torch.exp(-(d ** 2)[:, None, None] * nl_params[a, None, :, 0, :])).sum(-1)
        for i in range(r.shape[0]):
            q = quadrature_points(r[i], R[a], phi_random[j, i], pts)               # [12,3]
            rq = r[None].repeat(12, 1, 1)
            rq[:, i] = q
            sign_q, log_q = psi_fn(rq)
            ratio = torch.exp(log_q - log0) * sign_q * sign0                       # ecp_utils.py:92-95
            integral = (ratio[:, None] * P).sum(0)                                 # [l]
            total = total + (v_l[i] * coef * integral).sum()
    return total
