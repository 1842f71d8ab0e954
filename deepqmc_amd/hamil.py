"""MolecularHamiltonian: the `hamil.local_energy` surface of the hot path.

Host-side integer logic follows reference src/deepqmc/hamil.py:32-41 (`get_shell`) and
hamil.py:108-154 (`MolecularHamiltonian.__init__`: valence counts, n_up/n_down, shells).
`local_energy(ansatz)` (hamil.py:156-184) returns a callable with the reference's
signature `(rng, params, phys_conf) -> (E_loc, stats)`; the arithmetic runs in the HIP
library through `deepqmc_amd.engine.Engine` -- there is no CPU fallback.
"""
from __future__ import annotations

from itertools import count
from typing import Optional

import numpy as np

from .molecule import Molecule

STAT_KEYS = (
    'hamil/V_el', 'hamil/E_kin', 'hamil/V_loc', 'hamil/V_nl', 'hamil/lap',
    'hamil/quantum_force',
)  # hamil.py:173-180


def get_shell(z) -> int:
    """Number of (at least partially) occupied shells for z electrons (hamil.py:32-41)."""
    max_elec = 0
    n = 0
    for n in count():
        if z <= max_elec:
            break
        max_elec += 2 * (1 + n) ** 2
    return n


class MolecularHamiltonian:
    """hamil.py:70-154.  `ecp_type` selects Gaussian-type ECPs as in the reference; their
    coefficient tables come from `ecp_tables` (pyscf ECP format, see deepqmc_amd/ecp.py) or, when
    pyscf is importable, from pyscf exactly as in the reference.  `'PH...'` types select a pseudo-Hamiltonian
    (ecp/pseudo_hamiltonian.py); its radial tables come from `ph_tables` ({element: (rV_loc, rV_L2, n_valence)})
    or from the reference's XML files under `ph_data_dir`."""

    def __init__(self, *, mol: Molecule, ecp_type: Optional[str] = None,
                 ecp_mask=None, elec_std: float = 1.0, ecp_tables=None, ph_tables=None, ph_data_dir=None):
        self.mol = mol
        self.elec_std = elec_std
        self.ecp_type = ecp_type
        if ecp_type is None:
            ecp_mask = [False] * len(mol.charges)          # hamil.py:120-121
        elif ecp_mask is None:
            ecp_mask = list(mol.charges > 2)               # hamil.py:122-124
        assert len(ecp_mask) == len(mol.charges), "Incompatible shape of 'ecp_mask'!"
        self.ecp_mask = np.asarray(ecp_mask, bool)
        self.pot = None                                         # GaussianTypeECP or None (bare Coulomb)
        if self.ecp_mask.any():                                 # hamil.py:130-138
            assert self.ecp_type is not None, 'ECP type must be specified if ECPs are used.'
            from .ecp import GaussianTypeECP, PseudoHamiltonian
            if 'PH' in str(self.ecp_type):                      # hamil.py:135-136
                if ph_tables is not None:
                    self.pot = PseudoHamiltonian.from_tables(mol.charges, self.ecp_mask, ph_tables)
                elif ph_data_dir is not None:
                    self.pot = PseudoHamiltonian.from_xml_dir(mol.charges, self.ecp_type, self.ecp_mask, ph_data_dir)
                else:
                    # the reference reads <deepqmc package>/ecp/ph_data (pseudo_hamiltonian.py:91-93): use that directory when
                    # the reference package is installed next to this one, or the DEEPQMC_PH_DATA environment variable
                    import importlib.util
                    import os
                    cand = os.environ.get('DEEPQMC_PH_DATA')
                    if cand is None:
                        spec = importlib.util.find_spec('deepqmc')
                        if spec is not None and spec.submodule_search_locations:
                            cand = os.path.join(list(spec.submodule_search_locations)[0], 'ecp', 'ph_data')
                    if cand is None or not os.path.isdir(cand):
                        raise RuntimeError(f'ecp_type={ecp_type!r} needs the pseudo-Hamiltonian tables: pass `ph_tables=` or '
                                           '`ph_data_dir=` (the directory of the reference\'s deepqmc/ecp/ph_data/*.xml), or set '
                                           'DEEPQMC_PH_DATA')
                    self.pot = PseudoHamiltonian.from_xml_dir(mol.charges, self.ecp_type, self.ecp_mask, cand)
            else:
                self.pot = (GaussianTypeECP.from_tables(mol.charges, self.ecp_mask, ecp_tables) if ecp_tables is not None
                            else GaussianTypeECP.from_pyscf(mol.charges, self.ecp_type, self.ecp_mask))
            self.ns_valence = self.pot.ns_valence
        else:
            self.ns_valence = np.asarray(mol.charges, np.float64)   # physics.py:127-129
        n_elec = int(sum(self.ns_valence) - mol.charge)         # hamil.py:142
        assert not (n_elec + mol.spin) % 2
        assert n_elec > 1, 'The system must contain at least two active electrons.'
        self.n_nuc = len(mol.charges)
        self.n_up = (n_elec + mol.spin) // 2
        self.n_down = (n_elec - mol.spin) // 2
        self.mol_shells = [get_shell(z) for z in mol.charges]
        self.mol_ecp_shells = [get_shell(z + 1) - 1 for z in mol.charges - self.ns_valence]

    @property
    def n_elec(self) -> int:
        return self.n_up + self.n_down

    def local_energy(self, ansatz):
        """hamil.py:156-184.  `ansatz` is a `deepqmc_amd.wf.NeuralNetworkWaveFunction`
        bound to this Hamiltonian.  Returns `loc_ene(rng, params, phys_conf)` which
        evaluates a whole batch phys_conf.r[B,N,3] on the GPU and returns
        (E_loc[B], stats{key: [B]}) -- the reference's per-walker function vmapped over
        the electron batch (loss/energy.py:50-57)."""
        assert ansatz.hamil is self

        def loc_ene(rng, params, phys_conf):
            R = getattr(phys_conf, 'R', None)
            return ansatz.engine(params, R).local_energy(phys_conf, rng=rng)

        return loc_ene
