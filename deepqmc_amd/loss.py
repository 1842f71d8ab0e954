"""Batch-level callers of the hot path: `compute_local_energy` (reference
src/deepqmc/loss/energy.py:19-60) and `compute_psi_ratio` (loss/overlap.py:19-99), with the
reference's `[molecule, electronic state, walker]` batch axes.  The molecule and state axes are
host loops over HIP contexts (one per state's parameter set); the walker axis is the GPU batch.
Only single-geometry batches (M = 1, the `IdleNucleiSampler` case of the BASELINE configs) are
accepted.
"""
from __future__ import annotations

from typing import Sequence

import torch

from .hamil import STAT_KEYS


def _check(r):
    if r.dim() != 5 or r.shape[0] != 1:
        raise ValueError('expected r[M=1, S, B, N, 3]')


def compute_local_energy(rng, hamil, ansatz, params: Sequence, phys_conf_r: torch.Tensor):
    """loss/energy.py:19-60: returns (E_loc[M,S,B], stats{key: [M,S]} = per-(molecule,state)
    means over the walkers, energy.py:59).  `params` = one parameter tree per state."""
    _check(phys_conf_r)
    S = phys_conf_r.shape[1]
    assert len(params) == S
    es, stats = [], {k: [] for k in STAT_KEYS}
    for s in range(S):
        e, st = ansatz.engine(params[s]).local_energy(phys_conf_r[0, s].contiguous(), rng=rng)
        es.append(e)
        for k in STAT_KEYS:
            stats[k].append(st[k].mean())
    return torch.stack(es)[None], {k: torch.stack(v)[None] for k, v in stats.items()}


def compute_psi_ratio(ansatz, params: Sequence, phys_conf_r: torch.Tensor):
    """loss/overlap.py:77-99 + :40-75: R[m, i, j, b] = psi_i(r_b ~ psi_j^2) / psi_j(r_b ~ psi_j^2),
    computed from log-shifted values (shift = mean log|psi| of each state i over all samples)."""
    _check(phys_conf_r)
    S, B = phys_conf_r.shape[1], phys_conf_r.shape[2]
    sign = torch.empty(S, S, B, dtype=torch.float64, device=phys_conf_r.device)
    log = torch.empty(S, S, B, dtype=torch.float64, device=phys_conf_r.device)
    for i in range(S):                       # wave function i ...
        eng = ansatz.engine(params[i])
        for j in range(S):                   # ... on the samples of state j
            sg, lg = eng.wf_eval(phys_conf_r[0, j].contiguous())
            sign[i, j], log[i, j] = sg.double(), lg.double()
    mean_log = log.mean(dim=(1, 2))                                   # overlap.py:92-94
    shifted = log - mean_log[:, None, None]
    diag = torch.diagonal(shifted, dim1=0, dim2=1).permute(1, 0)     # [S(j), B]
    log_ratio = shifted - diag[None]
    sdiag = torch.diagonal(sign, dim1=0, dim2=1).permute(1, 0)
    return (sign * sdiag[None] * torch.exp(log_ratio))[None]


def symmetrize_overlap_with_clipped_geometric_mean(x: torch.Tensor) -> torch.Tensor:
    """loss/overlap.py:102-121: y_ij = sign(x_ij) sqrt(max(0, x_ij x_ji))."""
    return torch.sign(x) * torch.sqrt(torch.clamp(x * x.transpose(-1, -2), min=0.0))


def compute_mean_overlap(psi_ratio: torch.Tensor, weight: torch.Tensor):
    """loss/overlap.py:124-149: weighted mean of the psi ratios over the walkers of ALL ranks (the reference's
    all_device_mean; here one all-reduce of an [M,S,S] tensor), symmetrised; returns (sum_{i<j} S_ij^2 averaged
    over molecules, {'overlap/pairwise/mean': S[M,S,S]}).  psi_ratio [M,S,S,B], weight [M,S,B]."""
    import torch.distributed as dist
    s = (weight[:, None, :, :] * psi_ratio).sum(-1)
    n = torch.tensor(float(psi_ratio.shape[-1]), dtype=s.dtype, device=s.device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(s)
        dist.all_reduce(n)
    symm = symmetrize_overlap_with_clipped_geometric_mean(s / n)
    S = symm.shape[-1]
    iu = torch.triu_indices(S, S, offset=1, device=symm.device)
    loss = (symm[:, iu[0], iu[1]] ** 2).sum(-1).mean()
    return loss, {'overlap/pairwise/mean': symm}
