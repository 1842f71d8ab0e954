"""Helpers for the SIMT-emulation tests (tests/simt): build and load the emulated library."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
_lib_cache = None


def emu_lib():
    global _lib_cache
    if _lib_cache is None:
        from deepqmc_amd import _lib
        out = subprocess.run(['sh', os.path.join(HERE, 'simt', 'build_emu.sh')], check=True, capture_output=True, text=True)
        path = out.stdout.strip().splitlines()[-1]
        _lib_cache = _lib.bind(ctypes.CDLL(path))
    return _lib_cache
