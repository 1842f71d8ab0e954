"""Geometry helpers and edge construction (oracle; test infrastructure only).

Restates reference src/deepqmc/geom/general.py:19-43, utils.py:57-59,79-85 and
gnn/graph.py:17-31,66-159.
"""
from __future__ import annotations

import torch

F32_EPS = float(torch.finfo(torch.float32).eps)
F64_EPS = float(torch.finfo(torch.float64).eps)


def norm(rs: torch.Tensor, safe: bool = False, eps: float = F64_EPS) -> torch.Tensor:
    """utils.py:79-85.  `eps` = finfo(dtype).eps of the *reference's* compute dtype: pass
    F32_EPS when emulating the production f32 path in float64."""
    if safe:
        return torch.sqrt(eps + (rs * rs).sum(-1))
    return torch.linalg.norm(rs, dim=-1)


def triu_indices(n: int):
    i, j = torch.triu_indices(n, n, offset=1)
    return i, j


def triu_flat(x: torch.Tensor) -> torch.Tensor:
    """utils.py:57-59."""
    i, j = triu_indices(x.shape[-1])
    return x[..., i, j]


def pairwise_distance(c1: torch.Tensor, c2: torch.Tensor) -> torch.Tensor:
    """geom/general.py:19-21 (plain norm)."""
    return torch.linalg.norm(c1[..., :, None, :] - c2[..., None, :, :], dim=-1)


def pairwise_diffs(c1: torch.Tensor, c2: torch.Tensor) -> torch.Tensor:
    """geom/general.py:24-27: differences with the squared distance appended."""
    d = c1[..., :, None, :] - c2[..., None, :, :]
    return torch.cat([d, (d ** 2).sum(-1, keepdim=True)], dim=-1)


def pairwise_self_distance(c: torch.Tensor, full: bool = False, eps: float = F64_EPS) -> torch.Tensor:
    """geom/general.py:30-43: safe-norm distances in triu (i<j, row-major) order."""
    n = c.shape[-2]
    i, j = triu_indices(n)
    d = c[..., :, None, :] - c[..., None, :, :]
    dists = norm(d[..., i, j, :], safe=True, eps=eps)
    if full:
        # out-of-place scatter (one-hot matrix) so that the function composes with torch.func.vmap / jacfwd
        P = torch.zeros(len(i), n * n, dtype=c.dtype)
        k = torch.arange(len(i))
        P[k, i * n + j] = 1
        P[k, j * n + i] = 1
        return (dists @ P).reshape(*dists.shape[:-1], n, n)
    return dists


def offdiagonal_sender_idx(n: int) -> torch.Tensor:
    """gnn/graph.py:17-21: [n-1, n] sender index, row k column r -> k + (r <= k)."""
    a = torch.arange(n)
    k = torch.arange(n - 1)[:, None]
    return (a[None, :] <= k).long() + k


def compute_edges(pos_sender: torch.Tensor, pos_receiver: torch.Tensor, filter_diagonal: bool) -> torch.Tensor:
    """gnn/graph.py:23-31: diffs[s, r] = pos_receiver[r] - pos_sender[s]."""
    diffs = pos_receiver[None, :, :] - pos_sender[:, None, :]
    if filter_diagonal:
        n = pos_sender.shape[-2]
        recv = torch.arange(n)[None].expand(n - 1, n)
        send = offdiagonal_sender_idx(n)
        diffs = diffs[send, recv, :]
    return diffs


def molecular_edges(r: torch.Tensor, R: torch.Tensor, n_up: int, edge_types, self_interaction: bool):
    """gnn/graph.py:66-159.  Returns {type: dict of component arrays [n_send, n_recv, 3]}."""
    out = {}
    for typ in edge_types:
        if typ == 'ne':
            out[typ] = {'ne': compute_edges(R, r, False)}
        elif typ == 'same':
            out[typ] = {
                'uu': compute_edges(r[:n_up], r[:n_up], not self_interaction),
                'dd': compute_edges(r[n_up:], r[n_up:], not self_interaction),
            }
        elif typ == 'anti':
            out[typ] = {
                'du': compute_edges(r[n_up:], r[:n_up], False),
                'ud': compute_edges(r[:n_up], r[n_up:], False),
            }
        elif typ == 'up':
            out[typ] = {'up': compute_edges(r[:n_up], r, False)}
        elif typ == 'down':
            out[typ] = {'down': compute_edges(r[n_up:], r, False)}
        else:
            raise ValueError(typ)
    return out


def single_array(comps: dict) -> torch.Tensor:
    """`GraphEdges.single_array` (gnn/graph.py:246-256,300-310): component arrays
    flattened to rows and concatenated in dict order (uu;dd / du;ud)."""
    return torch.cat([c.reshape(-1, c.shape[-1]) for c in comps.values()], dim=0)


def from_single_array(comps: dict, arr: torch.Tensor) -> dict:
    """`update_from_single_array` (gnn/graph.py:258-266,312-319)."""
    out, o = {}, 0
    for k, c in comps.items():
        n = c.shape[0] * c.shape[1]
        out[k] = arr[o:o + n].reshape(c.shape[0], c.shape[1], arr.shape[-1])
        o += n
    return out
