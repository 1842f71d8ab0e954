"""Walker data parallelism: one process per GPU, walkers pre-sharded, ONE collective per VMC
step.  Replaces the reference's pmean/pmin/pmax/all_device_std chain
(src/deepqmc/parallel.py:175-225, called from observable.py:474-479 and loss/energy.py:74)
by an all-gather of a 56-byte record per rank (RCCL over xGMI; gloo on CPU for tests) and a
local Chan merge -- the messages are latency-bound, so one call instead of five.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_total: int, rank: int, world: int):
    """`electron_batch_size // device_count` split of sampling/sampling_utils.py:253-262;
    divisibility is required as in validate_kwargs.py:42-48."""
    if n_total % world:
        raise ValueError('electron_batch_size must be divisible by the number of GPUs')
    per = n_total // world
    return rank * per, (rank + 1) * per


def all_gather_records(record: np.ndarray, device) -> np.ndarray:
    """All-gather this rank's 7-double energy record; returns [world, 7]."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return np.asarray(record, np.float64).reshape(1, 7)
    t = torch.as_tensor(np.asarray(record, np.float64), device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out).cpu().numpy()


def energy_stats(engine, e_loc, w=None):
    """mean / std / min / max of the local energy over ALL ranks' walkers."""
    rec = engine.energy_record(e_loc, w)
    dev = e_loc.device if (dist.is_initialized() and dist.get_backend() == 'nccl') else 'cpu'
    return engine.merge_energy_records(all_gather_records(rec, dev))
