"""ctypes binding of libdqmc_hip.so (C ABI: include/dqmc.h).

There is no fallback: if the HIP library has not been built, `load()` raises.  Build it with
`python -c "import __graft_entry__ as g; g.build()"` (hipcc --offload-arch=gfx950).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_int64, c_size_t, c_uint8, c_uint64, c_void_p

from .program import DqmcBuf, DqmcOp, DqmcSystem

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB_PATH = os.path.join(CSRC, 'libdqmc_hip.so')

# every symbol include/dqmc.h declares: (name, restype, argtypes)
SIGNATURES = [
    ('dqmc_create', c_int, [POINTER(c_void_p), c_int, c_void_p, POINTER(DqmcSystem), POINTER(c_double),
                            POINTER(DqmcBuf), c_int, POINTER(DqmcOp), c_int, POINTER(c_double), c_size_t,
                            POINTER(c_int32), c_size_t]),
    ('dqmc_destroy', None, [c_void_p]),
    ('dqmc_last_error', c_char_p, []),
    ('dqmc_set_weights', c_int, [c_void_p, POINTER(c_double), c_size_t]),
    ('dqmc_wf_eval', c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    ('dqmc_local_energy', c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    ('dqmc_psi_grad', c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    ('dqmc_mcmc_steps', c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                c_int, c_double, c_uint64, c_void_p, c_void_p, c_void_p, POINTER(c_double)]),
    ('dqmc_langevin_update', c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_double), c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    ('dqmc_langevin_steps', c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_double),
                                    c_int, c_int, c_int, c_double, c_uint64, c_void_p, c_void_p, c_void_p, POINTER(c_double)]),
    ('dqmc_exchange_step', c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                   c_void_p, c_void_p, POINTER(c_double)]),
    ('dqmc_set_ecp', c_int, [c_void_p, c_int, POINTER(c_double), c_int, c_int, POINTER(c_double)]),
    ('dqmc_ecp_rotation', c_int, [c_void_p, c_uint64, c_void_p]),
    ('dqmc_set_pseudo_hamiltonian', c_int, [c_void_p, c_int, c_double, POINTER(c_double), POINTER(c_double), POINTER(c_int32)]),
    ('dqmc_energy_stats', c_int, [c_void_p, c_void_p, c_void_p, c_int, POINTER(c_double)]),
    ('dqmc_energy_stats_allgather', c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, POINTER(c_double)]),
    ('dqmc_merge_energy_stats', c_int, [POINTER(c_double), c_int, POINTER(c_double)]),
    ('dqmc_debug_read', c_int, [c_void_p, c_int, POINTER(c_double), c_size_t]),
    ('dqmc_debug_lanes', c_int, [c_void_p]),
    ('dqmc_last_refined', c_int, [c_void_p]),
    ('dqmc_refine_info', c_int, [c_void_p, POINTER(c_double)]),
    ('dqmc_last_chunks', c_int, [c_void_p, POINTER(c_int)]),
    ('dqmc_ecp_counts', c_int, [c_void_p, POINTER(c_int64)]),
    ('dqmc_refine_counters', c_int, [c_void_p, POINTER(c_int64)]),
    ('dqmc_substep_kernel', c_int, [c_void_p, c_char_p, c_size_t]),
    ('dqmc_refine_scores', c_int, [c_void_p, POINTER(c_double), c_int]),
    ('dqmc_set_option', c_int, [c_void_p, c_char_p, c_int]),
    ('dqmc_timing_enable', c_int, [c_void_p, c_int]),
    ('dqmc_timing_reset', c_int, [c_void_p]),
    ('dqmc_timing_get', c_int, [c_void_p, c_char_p, POINTER(c_double), POINTER(c_int64), POINTER(c_double)]),
    ('dqmc_timing_get_executed', c_int, [c_void_p, c_char_p, POINTER(c_double)]),
    ('dqmc_timing_names', c_int, [c_void_p, c_char_p, c_size_t]),
]


def bind(lib: ctypes.CDLL) -> ctypes.CDLL:
    """Attach argtypes/restype for every entry point; raises AttributeError if one is missing."""
    for name, restype, argtypes in SIGNATURES:
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


_cached = None


def load() -> ctypes.CDLL:
    global _cached
    if _cached is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} not found: the HIP library is not built. Run '
                '`python -c "import __graft_entry__ as g; g.build()"` (needs hipcc); there is no CPU fallback.')
        _cached = bind(ctypes.CDLL(LIB_PATH))
    return _cached
