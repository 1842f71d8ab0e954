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


class _stdout_to_stderr:
    """File descriptor 1 points at stderr while the block runs; C stdio is flushed on both edges so that text buffered by
    a C library inside the block leaves through the redirected descriptor."""

    def __enter__(self):
        import ctypes
        import os
        import sys
        self._libc = ctypes.CDLL(None)
        sys.stdout.flush()
        self._libc.fflush(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import os
        self._libc.fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


class RcclCommunicator:
    """An RCCL communicator of the library's own (`ncclCommInitRank` through ctypes) for
    `dqmc_energy_stats_allgather`: the unique id is created on rank 0 and distributed with one torch.distributed
    broadcast (or used directly for a single rank).  One process per GPU, as everywhere in this package."""

    def __init__(self, rank: int = 0, world: int = 1, device=None):
        import ctypes
        self._ct = ctypes
        self.rccl = ctypes.CDLL('librccl.so')
        self.rank, self.world = rank, world
        uid = (ctypes.c_char * 128)()
        if rank == 0:
            rc = self.rccl.ncclGetUniqueId(ctypes.byref(uid))
            if rc:
                raise RuntimeError(f'ncclGetUniqueId failed: {rc}')
        if world > 1:
            t = torch.tensor(list(bytes(uid)), dtype=torch.uint8, device=device)
            dist.broadcast(t, 0)
            uid = (ctypes.c_char * 128).from_buffer_copy(bytes(t.cpu().tolist()))

        class _Uid(ctypes.Structure):
            _fields_ = [('internal', ctypes.c_char * 128)]
        u = _Uid()
        ctypes.memmove(ctypes.byref(u), uid, 128)
        self.comm = ctypes.c_void_p()
        self.rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _Uid, ctypes.c_int]
        with _stdout_to_stderr():          # (this RCCL build prints a version banner on stdout at communicator creation:
            rc = self.rccl.ncclCommInitRank(ctypes.byref(self.comm), world, u, rank)      # a caller's stdout may be a protocol)
        if rc:
            raise RuntimeError(f'ncclCommInitRank failed: {rc}')

    def count(self) -> int:
        """Ranks of the communicator as RCCL itself reports them (ncclCommCount)."""
        n = self._ct.c_int(0)
        self.rccl.ncclCommCount.argtypes = [self._ct.c_void_p, self._ct.POINTER(self._ct.c_int)]
        rc = self.rccl.ncclCommCount(self.comm, self._ct.byref(n))
        if rc:
            raise RuntimeError(f'ncclCommCount failed: {rc}')
        return int(n.value)

    def close(self):
        if self.comm:
            self.rccl.ncclCommDestroy.argtypes = [self._ct.c_void_p]
            self.rccl.ncclCommDestroy(self.comm)
            self.comm = None


def energy_stats_inlib(engine, e_loc, comm: RcclCommunicator, w=None):
    """`energy_stats` with the collective inside the HIP library (one ncclAllGather on the context's stream)."""
    import ctypes
    engine.check_walker_vector('e_loc', e_loc)
    if w is not None:
        engine.check_walker_vector('w', w, e_loc.shape[0])
    out = (ctypes.c_double * 5)()
    engine._check(engine.lib.dqmc_energy_stats_allgather(engine._ctx, comm.comm, comm.world, e_loc.data_ptr(),
                                                         w.data_ptr() if w is not None else None, e_loc.shape[0], out))
    return dict(zip(('local_energy/mean', 'local_energy/std', 'local_energy/min', 'local_energy/max',
                     'local_energy/weighted_mean'), list(out)))
