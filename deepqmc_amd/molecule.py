"""Molecule container (host side).

Follows reference src/deepqmc/molecule.py:34-100 (`Molecule` dataclass: coords are
converted to bohr in __post_init__, charges to float) and units.py:17-22
(angstrom->bohr via CODATA constants).  The named geometries are the ones
BASELINE.json's configs need; values are the reference's conf/hamil/mol/*.yaml data
(LiH.yaml, N2.yaml, cyclobutadiene_square.yaml, H2.yaml) restated here, plus a
benzene geometry the reference does not ship (D6h, C-C 1.39 A,
C-H 1.09 A).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

# scipy.constants.angstrom / scipy.constants.value('atomic unit of length')
# (units.py:17-22); CODATA 2018 value, identical digits to what scipy 1.15 returns.
ANGSTROM_TO_BOHR = 1.8897261259077824


def _benzene_angstrom():
    cc, ch = 1.39, 1.09
    coords, charges = [], []
    for ring_r, z in ((cc, 6), (cc + ch, 1)):
        for k in range(6):
            a = math.pi / 3 * k
            coords.append([ring_r * math.cos(a), ring_r * math.sin(a), 0.0])
            charges.append(z)
    return coords, charges


_NAMED = {
    # name: (coords, charges, charge, spin, unit)
    'H2': ([[0.0, 0.0, 0.0], [0.742, 0.0, 0.0]], [1, 1], 0, 0, 'angstrom'),
    'LiH': ([[0.0, 0.0, 0.0], [1.595, 0.0, 0.0]], [3, 1], 0, 0, 'angstrom'),
    'Be': ([[0.0, 0.0, 0.0]], [4], 0, 0, 'angstrom'),
    'C': ([[0.0, 0.0, 0.0]], [6], 0, 2, 'angstrom'),
    'N2': ([[-2.13534, 0.0, 0.0], [2.13534, 0.0, 0.0]], [7, 7], 0, 0, 'angstrom'),
    'H2O': (
        [[0.0, 0.0, 0.0], [0.75695, 0.58588, 0.0], [-0.75695, 0.58588, 0.0]],
        [8, 1, 1], 0, 0, 'angstrom',
    ),
    'cyclobutadiene_square': (
        [
            [0.0, 0.0, 0.0], [2.74199, 0.0, 0.0], [2.74199, 2.74199, 0.0],
            [0.0, 2.74199, 0.0], [-1.44047, -1.44047, 0.0], [4.18246, -1.44047, 0.0],
            [4.18246, 4.18246, 0.0], [-1.44047, 4.18246, 0.0],
        ],
        [6, 6, 6, 6, 1, 1, 1, 1], 0, 0, 'angstrom',
    ),
    'benzene': (*_benzene_angstrom(), 0, 0, 'angstrom'),
}


@dataclass(frozen=True)
class Molecule:
    """coords [n_nuc,3] (stored in bohr), charges [n_nuc] (float), total charge, spin
    (n_up - n_down).  molecule.py:34-83."""

    coords: np.ndarray
    charges: np.ndarray
    charge: int
    spin: int
    unit: str = 'bohr'
    n_atom_types: int = field(init=False, default=0)

    all_names = frozenset(_NAMED)

    def __post_init__(self):
        mult = {'bohr': 1.0, 'angstrom': ANGSTROM_TO_BOHR}[self.unit]
        object.__setattr__(self, 'coords', np.asarray(self.coords, np.float64) * mult)
        object.__setattr__(self, 'charges', np.asarray(self.charges, np.float64))
        object.__setattr__(self, 'unit', 'bohr')
        object.__setattr__(self, 'n_atom_types', len(np.unique(self.charges)))

    def __len__(self):
        return len(self.charges)

    @classmethod
    def from_name(cls, name: str) -> 'Molecule':
        """molecule.py:104-118; raises ValueError for unknown names like the reference."""
        if name not in _NAMED:
            raise ValueError(f'Unknown molecule name: {name}')
        coords, charges, charge, spin, unit = _NAMED[name]
        return cls(coords=coords, charges=charges, charge=charge, spin=spin, unit=unit)
