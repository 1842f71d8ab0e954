"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol that
include/dqmc.h declares (no compute calls here); the ctypes binding table matches the header."""
import ctypes
import os
import re
import subprocess

import pytest

from deepqmc_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, 'include', 'dqmc.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(dqmc_[a-z_0-9]+)\s*\(', src)))


def test_binding_table_matches_header():
    assert sorted(n for n, _, _ in _lib.SIGNATURES) == header_symbols()


def test_library_exports_every_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.run(['make', '-C', _lib.CSRC, '-j8'], check=True, capture_output=True)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), name
    _lib.bind(lib)
    assert lib.dqmc_last_error() is not None


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libdqmc_hip.so')
    monkeypatch.setattr(_lib, '_cached', None)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.load()


def test_malformed_program_is_rejected():
    """Error behaviour of the boundary: bad programs return DQMC_E_ARG, never crash."""
    import numpy as np
    import torch
    from deepqmc_amd.engine import DqmcError, Engine
    from deepqmc_amd.hamil import MolecularHamiltonian
    from deepqmc_amd.molecule import Molecule
    from deepqmc_amd.params import init_params
    from deepqmc_amd.spec import paulinet
    from simt_util import emu_lib
    h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
    tree = init_params(paulinet(), 2, 2, 2)
    eng = Engine(paulinet(), h, tree, dtype=torch.float64, device='cpu', lib=emu_lib())
    with pytest.raises(DqmcError, match='unknown option'):
        eng.set_option('no_such_option', 1)
    bad = eng.program
    lin = next(op for op in bad.ops if op.kind == 3)
    lin.i[17] = 10_000                 # destination buffer out of range
    import deepqmc_amd.engine as E
    orig = E.compile_program
    try:
        E.compile_program = lambda *a, **k: bad
        with pytest.raises(DqmcError, match='malformed op'):
            Engine(paulinet(), h, tree, dtype=torch.float64, device='cpu', lib=emu_lib())
    finally:
        E.compile_program = orig
