"""Spin expectation value on the value path: `evaluate_spin` and the stochastic spin-raising estimator
(reference src/deepqmc/physics.py:159-239).

    <S^2>(r) = (n_up - n_dn)/2 ((n_up - n_dn)/2 + 1) + n_dn - sum_{beta in down} sum_{alpha in up} P_ab psi(r) / psi(r)

where P_ab swaps the positions of up electron alpha and down electron beta.  The reference walks the n_up n_dn swaps
with two nested fori_loops per walker; here all swapped configurations of a walker batch are ONE value-only evaluation of
B n_up n_dn walkers through the HIP engine (`dqmc_wf_eval`: the fused LDS-resident kernel or the layered MFMA kernels,
whichever the library picks), and the ratios are formed from (sign, log|psi|) as the reference does (physics.py:218-222).
"""
from __future__ import annotations

import torch

from .types import PhysicalConfiguration


def _swapped(r: torch.Tensor, up_idx: torch.Tensor, down_idx: torch.Tensor) -> torch.Tensor:
    """r[B,N,3] -> r'[B,P,N,3] with electrons up_idx[p] and down_idx[p] exchanged in copy p (physics.py:206-214)."""
    B, N, _ = r.shape
    P = up_idx.numel()
    perm = torch.arange(N, device=r.device).repeat(P, 1)
    rows = torch.arange(P, device=r.device)
    perm[rows, up_idx] = down_idx
    perm[rows, down_idx] = up_idx
    return r[:, perm]                                             # [B, P, N, 3]


def _ratios(ansatz, params, phys_conf, up_idx, down_idx):
    r = phys_conf.r if isinstance(phys_conf, PhysicalConfiguration) else phys_conf
    R = phys_conf.R if isinstance(phys_conf, PhysicalConfiguration) else None
    eng = ansatz.engine(params, R)
    r = torch.as_tensor(r, dtype=eng.dtype, device=eng.device)
    B, N = r.shape[0], r.shape[1]
    s0, l0 = eng.wf_eval(r.contiguous(), R)
    rp = _swapped(r, up_idx, down_idx)
    P = rp.shape[1]
    sp, lp = eng.wf_eval(rp.reshape(B * P, N, 3).contiguous(), R)
    sp, lp = sp.reshape(B, P).double(), lp.reshape(B, P).double()
    return s0.double()[:, None] * sp * torch.exp(lp - l0.double()[:, None])          # [B, P]


def evaluate_spin(hamil, ansatz):
    """physics.py:159-183.  Returns `spin(params, phys_conf) -> <S^2>[B]` (float64), batched over the walkers."""
    n_up, n_dn = hamil.n_up, hamil.n_down
    umd = n_up - n_dn

    def evaluate_spin_(params, phys_conf):
        dev = ansatz.engine(params, getattr(phys_conf, 'R', None)).device
        up = torch.arange(n_up, device=dev).repeat(n_dn)                              # alpha fastest, beta slowest: the loop order
        dn = (n_up + torch.arange(n_dn, device=dev)).repeat_interleave(n_up)          #   of physics.py:174-176 / :224-226
        s2 = umd / 2 * (umd / 2 + 1) + n_dn
        return s2 - _ratios(ansatz, params, phys_conf, up, dn).sum(-1)

    return evaluate_spin_


def make_stochastic_spin_raising_operator(hamil, ansatz):
    """physics.py:229-239: 1 - sum_alpha P_{alpha beta} psi / psi for one down electron `down_idx` (an int, or one index
    per walker)."""
    n_up = hamil.n_up

    def evaluate(params, phys_conf, down_idx):
        dev = ansatz.engine(params, getattr(phys_conf, 'R', None)).device
        r = torch.as_tensor(phys_conf.r if isinstance(phys_conf, PhysicalConfiguration) else phys_conf)      # (NumPy input too)
        down_idx = torch.as_tensor(down_idx, device=dev)
        if down_idx.dim() == 0:
            up = torch.arange(n_up, device=dev)
            return 1.0 - _ratios(ansatz, params, phys_conf, up, down_idx.expand(n_up).clone()).sum(-1)
        out = torch.empty(r.shape[0], dtype=torch.float64, device=dev)               # per-walker indices: group equal ones
        for d in torch.unique(down_idx).tolist():
            sel = (down_idx == d).nonzero().flatten()
            rs = r[sel.to(r.device)]
            if isinstance(phys_conf, PhysicalConfiguration):
                Rb = phys_conf.R
                if Rb is not None and torch.as_tensor(Rb).dim() == 3:        # a geometry per walker: the selected walkers' own
                    Rb = torch.as_tensor(Rb)[sel.to(torch.as_tensor(Rb).device)]
                sub = PhysicalConfiguration(Rb, rs, None)
            else:
                sub = rs
            out[sel] = evaluate(params, sub, d)
        return out

    return evaluate
