"""Gaussian-type effective core potentials: parameter tables (host side).

Mirrors reference ecp/gaussian_type_ecp.py:32-93 (`parse_gaussian_type_ecp_params`): per nucleus the number
of valence electron slots, the local terms `loc[n_nuc, 3, 2, n_terms]` (r^-1, r^0, r^1 Gaussians; index 0 =
exponent, 1 = coefficient) and the non-local channels `nl[n_nuc, l_max+1, 2, n_terms]`, zero padded.  The
reference reads the bfd / ccECP coefficients from pyscf (`pyscf.gto.basis.load_ecp`); pyscf is neither
vendored in the reference nor installed here, so tables are supplied by the caller in pyscf's own format
(`ECP_TABLES`-style dict, see `GaussianTypeECP.from_tables`) or, if pyscf happens to be importable, looked up
exactly as the reference does.  The arithmetic (local and non-local potential) runs in the HIP library
(csrc/kernels_head.hip: k_final, csrc/kernels_ecp.hip).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np

ELEMENTS = ['X', 'H', 'He', 'Li', 'Be', 'B', 'C', 'N', 'O', 'F', 'Ne', 'Na', 'Mg', 'Al', 'Si', 'P', 'S', 'Cl', 'Ar',
            'K', 'Ca', 'Sc', 'Ti', 'V', 'Cr', 'Mn', 'Fe', 'Co', 'Ni', 'Cu', 'Zn', 'Ga', 'Ge', 'As', 'Se', 'Br', 'Kr']


def _pad3(arrays):
    """ecp_utils.py:78-90: zero-pad a list of 3-D arrays to a common shape and stack."""
    arrays = [np.asarray(a, np.float64).reshape(np.asarray(a).shape if np.asarray(a).ndim == 3 else (1, 1, 0))
              for a in arrays]
    shape = np.max([a.shape for a in arrays], axis=0)
    out = np.zeros((len(arrays), *shape))
    for k, a in enumerate(arrays):
        out[k, :a.shape[0], :a.shape[1], :a.shape[2]] = a
    return out


class GaussianTypeECP:
    """ns_valence[n_nuc], loc_params[n_nuc,3,2,n_loc], nl_params[n_nuc,L,2,n_nl] (gaussian_type_ecp.py:112-125)."""

    def __init__(self, ns_valence, loc_params, nl_params, ecp_mask):
        self.ns_valence = np.asarray(ns_valence, np.float64)
        self.loc_params = np.ascontiguousarray(loc_params, np.float64)
        self.nl_params = np.ascontiguousarray(nl_params, np.float64)
        self.ecp_mask = np.asarray(ecp_mask, bool)
        n = len(self.ns_valence)
        assert self.loc_params.ndim == 4 and self.loc_params.shape[:3] == (n, 3, 2)
        assert self.nl_params.ndim == 4 and self.nl_params.shape[0] == n and self.nl_params.shape[2] == 2
        self.nuc_with_nl_pot = np.unique(np.nonzero(self.nl_params)[0])       # :121

    @classmethod
    def from_tables(cls, charges: Sequence[float], ecp_mask: Sequence[bool], tables: Dict[str, list]):
        """`tables[element]` in pyscf's ECP format: [n_core, [[-1, [r^-2.., r^-1 terms, r^0 terms, r^1 terms, ..]],
        [0, [.., .., s-channel r^0 terms, ..]], [1, [...]], ...]] with each term list [[exponent, coefficient], ...]
        -- what `load_ecp(ecp_type, [element])` returns.  Parsing follows gaussian_type_ecp.py:55-93."""
        ns_valence, loc_list, nl_list, n_same = [], [], [], []
        for z, use in zip(charges, ecp_mask):
            if use:
                data = tables[ELEMENTS[int(z)]]
                loc = [list(map(list, t)) for t in data[1][0][1][1:4]]
                loc += [[] for _ in range(3 - len(loc))]
                if len(data[1]) > 1:
                    chans = [np.asarray(di[1][2], np.float64).reshape(-1, 2) for di in data[1][1:]]
                    nl = np.zeros((len(chans), 2, max(len(c) for c in chans)))     # ragged channels: zero padded
                    for l, c in enumerate(chans):
                        nl[l, :, :len(c)] = c.T
                else:
                    nl = np.zeros((1, 1, 0))
                n_same.append(max(len(t) for t in loc))
                n_core = data[0]
            else:
                n_core, loc, nl = 0, [[], [], []], np.zeros((1, 1, 0))
            ns_valence.append(z - n_core)
            loc_list.append(loc)
            nl_list.append(nl)
        pad = max(n_same, default=0)
        loc_arr = np.zeros((len(loc_list), 3, 2, pad))
        for a, loc in enumerate(loc_list):
            for term, pairs in enumerate(loc):
                for k, (alpha, beta) in enumerate(pairs):
                    loc_arr[a, term, 0, k], loc_arr[a, term, 1, k] = alpha, beta
        nl_arr = _pad3(nl_list)
        if nl_arr.shape[2] == 1:        # no nucleus has a non-local part
            nl_arr = np.zeros((len(nl_list), max(nl_arr.shape[1], 1), 2, 0))
        return cls(ns_valence, loc_arr, nl_arr, ecp_mask)

    @classmethod
    def from_pyscf(cls, charges, ecp_type: str, ecp_mask):
        """The reference's own lookup (gaussian_type_ecp.py:55-61); needs pyscf."""
        try:
            from pyscf.gto.basis import load_ecp
        except ImportError as e:
            raise RuntimeError(
                f"ecp_type={ecp_type!r} needs pyscf's coefficient tables, and pyscf is not installed; pass "
                "`ecp_tables=` (pyscf ECP format) to MolecularHamiltonian instead") from e
        tables = {}
        for z, use in zip(charges, ecp_mask):
            if use:
                el = ELEMENTS[int(z)]
                tables[el] = load_ecp(ecp_type, [el])
                assert tables[el], f'Effective core potential of type {ecp_type} not found for {el} atom.'
        return cls.from_tables(charges, ecp_mask, tables)


# ----------------------------------------------------------------------------------------------------------------------
# Pseudo-Hamiltonians (reference ecp/pseudo_hamiltonian.py).  Host side: the radial tables; the arithmetic
# (coefficients A, b per electron, the transformed forward-Laplacian, the local term) runs in the HIP library
# (csrc/kernels_ecp.hip: k_ph_coeffs; pair_feature_lane, k_orbitals, k_final).

# pseudo_hamiltonian.py:18-30 (default file suffix per element, the OPH23 selection)
ELEMENTS_WITH_EXISTING_PH = {15: ('P', 'cc'), 16: ('S', 'cc'), 17: ('Cl', 'cc'), 24: ('Cr', 'cc'), 25: ('Mn', 'hf'),
                             26: ('Fe', 'cc'), 27: ('Co', 'cc'), 28: ('Ni', 'hf'), 29: ('Cu', 'hf'), 30: ('Zn', 'cc')}
PH_GRID, PH_RMAX = 10001, 10.0        # pseudo_hamiltonian.py:95 (rx = linspace(0, 10, 10001))


def parse_ph_xml(xml_file):
    """(r*V_loc + Z_eff, r*V_L2, n_valence) from a QMCPACK-style semilocal pseudopotential file whose s, p, d
    channels were built as a pseudo-Hamiltonian (pseudo_hamiltonian.py:32-70): with s(r), d(r) the r*V tables of the
    l = 0 and l = 2 channels, v0 = s - d, the local function is d + v0 + zval and the L^2 function -v0 / 6."""
    from xml.etree import ElementTree
    root = ElementTree.parse(xml_file).getroot()
    zval = float(root.find('header').attrib['zval'])
    semilocal = root.find('semilocal')
    chan = {}
    for vps in semilocal.findall('vps'):
        chan[vps.attrib['l']] = np.array(vps.find('radfunc').find('data').text.split(), np.float64)
    s, d = chan['s'], chan['d']
    v0 = s - d
    return d + v0 + zval, -v0 / 6.0, zval


class PseudoHamiltonian:
    """ns_valence[n_nuc] and the tables rv_loc / rv_l2 [n_nuc, n_grid] on linspace(0, r_max, n_grid)
    (rows of nuclei outside `ecp_mask` are zero).  pseudo_hamiltonian.py:73-112,159-171."""

    def __init__(self, ns_valence, rv_loc, rv_l2, r_max, ecp_mask):
        self.ns_valence = np.asarray(ns_valence, np.float64)
        self.rv_loc = np.ascontiguousarray(rv_loc, np.float64)
        self.rv_l2 = np.ascontiguousarray(rv_l2, np.float64)
        self.r_max = float(r_max)
        self.ecp_mask = np.asarray(ecp_mask, bool)
        assert self.rv_loc.shape == self.rv_l2.shape == (len(self.ns_valence), self.rv_loc.shape[1])

    @classmethod
    def from_tables(cls, charges, ecp_mask, tables, r_max=PH_RMAX):
        """`tables[element] = (rv_loc[n_grid], rv_l2[n_grid], n_valence)` -- what `parse_ph_xml` returns."""
        n_grid = max((len(tables[ELEMENTS[int(z)]][0]) for z, use in zip(charges, ecp_mask) if use), default=2)
        ns_valence, loc, l2 = [], np.zeros((len(charges), n_grid)), np.zeros((len(charges), n_grid))
        for a, (z, use) in enumerate(zip(charges, ecp_mask)):
            if use:
                t_loc, t_l2, n_val = tables[ELEMENTS[int(z)]]
                assert len(t_loc) == len(t_l2) == n_grid, 'all pseudo-Hamiltonian tables must share one grid'
                loc[a], l2[a] = t_loc, t_l2
                ns_valence.append(float(n_val))
            else:
                ns_valence.append(float(z))
        return cls(ns_valence, loc, l2, r_max, ecp_mask)

    @classmethod
    def from_xml_dir(cls, charges, ecp_type, ecp_mask, ph_data_dir):
        """The reference's lookup (pseudo_hamiltonian.py:73-112): `<ph_data_dir>/<El>.<suffix>.xml`, suffix from
        `ecp_type` ('PHcc' -> 'cc'; bare 'PH' -> the per-element default).  The reference ships these files in
        deepqmc/ecp/ph_data (separately licensed); point `ph_data_dir` at them."""
        import os
        suffix_req = str(ecp_type)[2:] if str(ecp_type).startswith('PH') else str(ecp_type)
        tables = {}
        for z, use in zip(charges, ecp_mask):
            if use:
                z = int(z)
                assert z in ELEMENTS_WITH_EXISTING_PH, \
                    f'Pseudo-Hamiltonian for atomic number {z} not found (probably does not exist!).'
                name, default_suffix = ELEMENTS_WITH_EXISTING_PH[z]
                if name not in tables:
                    tables[name] = parse_ph_xml(os.path.join(ph_data_dir, f'{name}.{suffix_req or default_suffix}.xml'))
        return cls.from_tables(charges, ecp_mask, tables)
