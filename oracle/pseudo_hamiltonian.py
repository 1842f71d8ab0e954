"""Pseudo-Hamiltonian (oracle; test infrastructure only).

Restates reference ecp/pseudo_hamiltonian.py in float64 PyTorch, function by function:
`local_potential` (:173-190), `coefficients` (`compute_coefficients_of_differential_operators`, :192-234),
`transformed_laplacian` (`compute_differential_operator_using_laplacian`, :115-146 -- literally the reference's
route: change coordinates to v = Q^-1 r with Q fixed, take the plain Laplacian of v -> log|psi|(Q v)) and
`kinetic_term` (:236-278).  The tabulated radial functions are interpolated like
jax.scipy.interpolate.RegularGridInterpolator(method='linear', fill_value=0.0) on linspace(0, r_max, n_grid) (:95-101).

PARITY UNPINNED for this file: the reference has no test of the pseudo-Hamiltonian (nothing under tests/ imports
it), so there is no golden to compare with.  The formulas are checked by properties (tests/test_pseudo_hamiltonian.py):
a vanishing L^2 table reduces to the ordinary kinetic energy; the coordinate-change Laplacian equals
sum_i tr(A_i Hess_ii) from the full Hessian; the XML tables of the reference (when present) satisfy the
2 v0 = 3 v1 relation the reference's parser relies on (:59-62).
"""
from __future__ import annotations

from typing import Callable

import torch
from torch.func import grad, hessian

from . import geom
from . import physics as ophys
from . import wf as owf


def interp(table: torch.Tensor, r_max: float, x: torch.Tensor) -> torch.Tensor:
    """Linear interpolation on linspace(0, r_max, len(table)); 0 outside the grid."""
    n = table.shape[0]
    t = x * ((n - 1) / r_max)
    inside = (t >= 0) & (t <= n - 1)
    tc = t.clamp(0, n - 1)
    k = tc.detach().floor().long().clamp(max=n - 2)
    f = tc - k
    val = table[k] + f * (table[k + 1] - table[k])
    return torch.where(inside, val, torch.zeros_like(val))


def _ph_dists(r, R, mask):
    idx = torch.nonzero(torch.as_tensor(mask)).reshape(-1)
    diffs = r[:, None, :] - R[None, idx, :]                      # (N, n_ph, 3)
    return idx, diffs, diffs.norm(dim=-1)


def local_potential(r, R, ns_valence, mask, rv_loc, r_max):
    """:173-190: -sum Z_eff / |r - R| over all nuclei + sum rV_loc(d) / d over the PH nuclei."""
    v_coul = -(ns_valence / geom.pairwise_distance(r, R)).sum()
    idx, _, d = _ph_dists(r, R, mask)
    rv = torch.stack([interp(rv_loc[a], r_max, d[:, j]) for j, a in enumerate(idx.tolist())], dim=1)
    return v_coul + (rv / d).sum()


def coefficients(r, R, mask, rv_l2, r_max):
    """:192-234 -> A [N,3,3], b [N,3]."""
    idx, diffs, d = _ph_dists(r, R, mask)
    rv = torch.stack([interp(rv_l2[a], r_max, d[:, j]) for j, a in enumerate(idx.tolist())], dim=1)   # (N, n_ph)
    v = rv / d
    b = (2 * v[..., None] * diffs).sum(-2)
    eye = torch.eye(3, dtype=r.dtype)
    diag = (rv * d)[..., None, None] * eye
    nondiag = v[..., None, None] * diffs[..., :, None] * diffs[..., None, :]
    A = (diag - nondiag).sum(-3) + 0.5 * eye
    return A, b


def transformed_laplacian(logpsi: Callable, Q: torch.Tensor, r: torch.Tensor):
    """:115-146 with the Hessian-trace Laplacian: (Laplacian in v, gradient in v [N,3])."""
    v = torch.linalg.solve_triangular(Q, r[..., None], upper=False)[..., 0]

    def f(v_flat):
        rr = torch.einsum('nxy,ny->nx', Q, v_flat.reshape(-1, 3))
        return logpsi(rr.reshape(-1))

    lap, jac = ophys.laplacian_hessian(f, v.reshape(-1))
    return lap, jac.reshape(-1, 3)


def kinetic_term(logpsi: Callable, r, R, mask, rv_l2, r_max):
    """:236-278 -> (E_kin, Laplacian-like term, quantum-force-like term, grad_v)."""
    A, b = coefficients(r, R, mask, rv_l2, r_max)
    Q = torch.linalg.cholesky(A)                                 # lower: A = Q Q^T
    lap, jac_v = transformed_laplacian(logpsi, Q, r)
    jac_r = torch.linalg.solve_triangular(Q.transpose(-1, -2), jac_v[..., None], upper=True)[..., 0]
    first_order = (b * jac_r).sum()
    qf = (jac_v * jac_v).sum()
    return first_order - (lap + qf), lap, qf, jac_v


def kinetic_term_direct(logpsi: Callable, r, R, mask, rv_l2, r_max):
    """The same operator without the coordinate change: sum_i b_i.grad_i - sum_i tr(A_i Hess_ii) - grad^T A grad
    (an independent route for the property test)."""
    A, b = coefficients(r, R, mask, rv_l2, r_max)
    x = r.reshape(-1)
    g = grad(logpsi)(x).reshape(-1, 3)
    H = hessian(logpsi)(x)
    N = r.shape[0]
    second = sum((A[i] * H[3 * i:3 * i + 3, 3 * i:3 * i + 3]).sum() for i in range(N))
    quad = torch.einsum('na,nab,nb->', g, A, g)
    return (b * g).sum() - second - quad


def local_energy(params, spec, r, R, ns_valence, n_up, mask, rv_loc, rv_l2, r_max, eps=geom.F64_EPS):
    """hamil.py:160-182 with `pot = PseudoHamiltonian` for one walker -> (E_loc, stats, grad_v)."""
    def logpsi(flat):
        return owf.wave_function(params, spec, flat.reshape(-1, 3), R, n_up, eps)[1]

    e_kin, lap, qf, jac_v = kinetic_term(logpsi, r, R, mask, rv_l2, r_max)
    v_loc = local_potential(r, R, ns_valence, mask, rv_loc, r_max)
    v_el = ophys.electronic_potential(r, eps)
    e_nuc = ophys.nuclear_energy(R, ns_valence)
    e_loc = e_kin + v_loc + v_el + e_nuc
    stats = {'hamil/V_el': v_el, 'hamil/E_kin': e_kin, 'hamil/V_loc': v_loc, 'hamil/V_nl': r.new_zeros(()),
             'hamil/lap': lap, 'hamil/quantum_force': qf}
    return e_loc, stats, jac_v
