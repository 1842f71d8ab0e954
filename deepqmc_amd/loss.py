"""Batch-level callers of the hot path: `compute_local_energy` (reference
src/deepqmc/loss/energy.py:19-60) and `compute_psi_ratio` (loss/overlap.py:19-99), with the
reference's `[molecule, electronic state, walker]` batch axes.  The molecule and state axes are
host loops over HIP contexts (one per state's parameter set; the geometry of a molecule is an argument
of the call); the walker axis is the GPU batch.  `phys_conf` is either the electron tensor r[M,S,B,N,3]
(M = 1, the Hamiltonian's geometry) or a `PhysicalConfiguration(R[M,n_nuc,3], r[M,S,B,N,3], mol_idx)` as
`MultiNuclearGeometrySampler.sample` returns it.
"""
from __future__ import annotations

from typing import Sequence

import torch

from .hamil import STAT_KEYS


def _unpack(phys_conf):
    """-> (r[M,S,B,N,3], [R_m or None per molecule])."""
    if hasattr(phys_conf, 'r'):
        r, R = phys_conf.r, phys_conf.R
    else:
        r, R = phys_conf, None
    if r.dim() != 5:
        raise ValueError('expected r[M, S, B, N, 3]')
    M = r.shape[0]
    if R is None:
        if M != 1:
            raise ValueError('M > 1 molecules need their geometries: pass a PhysicalConfiguration(R[M,n_nuc,3], r, mol_idx)')
        return r, [None]
    R = torch.as_tensor(R)
    while R.dim() > 3:            # the samplers tile R over the state / walker axes (electron_samplers.py:165-173)
        R = R[:, 0]
    if R.dim() == 2:
        R = R[None]
    if R.shape[0] != M:
        raise ValueError('R and r disagree on the number of molecules')
    return r, [R[m] for m in range(M)]


def compute_local_energy(rng, hamil, ansatz, params: Sequence, phys_conf_r: torch.Tensor):
    """loss/energy.py:19-60: returns (E_loc[M,S,B], stats{key: [M,S]} = per-(molecule,state)
    means over the walkers, energy.py:59).  `params` = one parameter tree per state."""
    from .types import PhysicalConfiguration
    r, Rs = _unpack(phys_conf_r)
    M, S = r.shape[0], r.shape[1]
    assert len(params) == S
    es, stats = [], {k: [] for k in STAT_KEYS}
    for m in range(M):
        em, sm = [], {k: [] for k in STAT_KEYS}
        for s in range(S):
            rr = r[m, s].contiguous()
            pc = rr if Rs[m] is None else PhysicalConfiguration(Rs[m], rr, None)
            e, st = ansatz.engine(params[s], Rs[m]).local_energy(pc, rng=rng)
            em.append(e)
            for k in STAT_KEYS:
                sm[k].append(st[k].mean())
        es.append(torch.stack(em))
        for k in STAT_KEYS:
            stats[k].append(torch.stack(sm[k]))
    return torch.stack(es), {k: torch.stack(v) for k, v in stats.items()}


def compute_psi_ratio(ansatz, params: Sequence, phys_conf_r: torch.Tensor):
    """loss/overlap.py:77-99 + :40-75: R[m, i, j, b] = psi_i(r_b ~ psi_j^2) / psi_j(r_b ~ psi_j^2),
    computed from log-shifted values (shift = mean log|psi| of each state i over all samples).  Returns the reference's
    tuple `(psi_ratio, stats)`; `stats` is the (empty) dict of compute_wave_function_values, overlap.py:19-50."""
    r, Rs = _unpack(phys_conf_r)
    M, S, B = r.shape[0], r.shape[1], r.shape[2]
    out = []
    for m in range(M):
        sign = torch.empty(S, S, B, dtype=torch.float64, device=r.device)
        log = torch.empty(S, S, B, dtype=torch.float64, device=r.device)
        r_all = r[m].reshape(S * B, *r.shape[3:]).contiguous()      # the samples of ALL states as one batch
        for i in range(S):                       # wave function i on the samples of every state j: ONE evaluation of
            eng = ansatz.engine(params[i], Rs[m])                     # S * B walkers per parameter set instead of S
            sg, lg = eng.wf_eval(r_all, Rs[m])                        # (the reference vmaps over both axes, overlap.py:40-49)
            sign[i], log[i] = sg.double().reshape(S, B), lg.double().reshape(S, B)
        mean_log = log.mean(dim=(1, 2))                                   # overlap.py:92-94
        shifted = log - mean_log[:, None, None]
        diag = torch.diagonal(shifted, dim1=0, dim2=1).permute(1, 0)     # [S(j), B]
        log_ratio = shifted - diag[None]
        sdiag = torch.diagonal(sign, dim1=0, dim2=1).permute(1, 0)
        out.append(sign * sdiag[None] * torch.exp(log_ratio))
    return torch.stack(out), {}


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
        packed = torch.cat([s.reshape(-1), n.reshape(1)])          # sums and walker count in ONE all-reduce
        dist.all_reduce(packed)
        s, n = packed[:-1].reshape(s.shape), packed[-1]
    symm = symmetrize_overlap_with_clipped_geometric_mean(s / n)
    S = symm.shape[-1]
    iu = torch.triu_indices(S, S, offset=1, device=symm.device)
    loss = (symm[:, iu[0], iu[1]] ** 2).sum(-1).mean()
    return loss, {'overlap/pairwise/mean': symm}
