"""Wire structs of the hot path.

Mirrors the reference's `Psi(sign, log)` (reference src/deepqmc/types.py:20-27) and
`PhysicalConfiguration(R, r, mol_idx)` (types.py:30-56).  Arrays are torch tensors
(device buffers handed to the C-ABI by raw pointer) or NumPy arrays on the host.
"""
from __future__ import annotations

from typing import Any, NamedTuple


class Psi(NamedTuple):
    """Wave function value: sign in {-1, 0, +1} and log|psi| (types.py:20-27)."""

    sign: Any
    log: Any


class PhysicalConfiguration(NamedTuple):
    """Nuclear coordinates R[..., n_nuc, 3], electron coordinates r[..., N, 3] and the
    molecule index the sample belongs to (types.py:30-56)."""

    R: Any
    r: Any
    mol_idx: Any

    def __len__(self):  # types.py:49-50
        return len(self.r)

    @property
    def batch_shape(self):  # types.py:52-56
        return tuple(self.r.shape[:-2])
