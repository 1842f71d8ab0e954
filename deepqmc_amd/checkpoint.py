"""Reader for the reference's training checkpoints -- the wire format either side of the hot path.

The reference writes `pickle.dump((step, serialize_train_state(state)), f)` (src/deepqmc/log.py:114-128): a
`deepqmc.types.TrainState(sampler, params, opt)` named tuple (types.py:99-104) of JAX arrays, gathered on one device
(log.py:45-55): `params` is the haiku tree `{module_path: {leaf: array}}` whose leaves carry a leading
electronic-state axis `[S, ...]` (wf/base.py:27 `tree_stack`); `sampler['elec']` holds `r [M, S, B, N, 3]`,
`psi.sign / psi.log [M, S, B]`, `age [M, S, B]`, `tau [M, S]` (log.py:51-54).

Neither `deepqmc` nor `jax` exists in this image, so the file is read with a restricted unpickler that

  * rebuilds JAX arrays as NumPy arrays (`jax._src.array._reconstruct_array(fun, args, arr_state, aval_state)` is how
    jax.Array pickles itself: `fun(*args)` IS the NumPy reconstruction),
  * maps `deepqmc.types.TrainState` / `Psi` to the local named tuples,
  * turns every other non-NumPy class (optimiser / KFAC state, haiku containers) into an inert `Opaque` record, and
    refuses anything callable outside that allow-list (a checkpoint is data; no code of it runs).

`load_params(path)` returns the per-state parameter trees in the layout `NeuralNetworkWaveFunction.engine` expects, so a
checkpoint written by the reference on a JAX machine can be evaluated by the HIP path here.
"""
from __future__ import annotations

import io
import pickle
from collections import OrderedDict
from typing import Any, List, NamedTuple, Tuple

import numpy as np

from .types import Psi


class TrainState(NamedTuple):
    """types.py:99-104."""
    sampler: Any
    params: Any
    opt: Any


class Opaque:
    """Placeholder for an object whose class is not available (optimiser state etc.): keeps what pickle gave it."""

    def __init__(self, *args, **kw):
        self.args, self.kw, self.state = args, kw, None

    def __setstate__(self, state):
        self.state = state

    def __call__(self, *a, **k):          # reduce callables that are themselves stubs
        return Opaque(*a, **k)


def _reconstruct_array(fun, args, arr_state, aval_state):
    """jax/_src/array.py `_reconstruct_array`: NumPy value in, device array out -- here the NumPy value is the result."""
    np_value = fun(*args)
    np_value.__setstate__(arr_state)
    return np_value


def _opaque_class(module, name):
    return type(name, (Opaque,), {'__module__': f'opaque.{module}'})


_SAFE_BUILTINS = {'dict', 'list', 'tuple', 'set', 'frozenset', 'complex', 'slice', 'range', 'bytearray', 'bytes', 'int', 'float', 'bool', 'str'}


# Exact (module, name) pairs a NumPy array / scalar / dtype pickle refers to (NumPy 1.x and 2.x module paths).  Nothing
# else of NumPy is reachable: `np.load`, `np.save`, `np.fromfile` ... are callables like any other.
_NUMPY_CORE = ('numpy.core.multiarray', 'numpy._core.multiarray', 'numpy.core.numeric', 'numpy._core.numeric')
_SAFE_NUMPY = ({('numpy', 'ndarray'), ('numpy', 'dtype')}
               | {(m, n) for m in _NUMPY_CORE[:2] for n in ('_reconstruct', 'scalar', '_frombuffer')}
               | {(m, '_frombuffer') for m in _NUMPY_CORE[2:]})


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        # protocol >= 4 resolves dotted names attribute by attribute ('builtins.eval' under any allowed module):
        # no legitimate entry of a checkpoint has one
        if '.' in name:
            raise pickle.UnpicklingError(f'refusing dotted name {module}:{name} in a checkpoint')
        if module == 'numpy' or module.startswith('numpy.'):
            if (module, name) in _SAFE_NUMPY:
                return super().find_class(module, name)
            if module == 'numpy.dtypes' and name.endswith('DType') and name.isidentifier():
                cls = getattr(np.dtypes, name, None)
                if isinstance(cls, type) and issubclass(cls, np.dtype):
                    return cls
            raise pickle.UnpicklingError(f'refusing {module}.{name} in a checkpoint')
        if module in ('jax._src.array', 'jax.interpreters.xla', 'jaxlib.xla_extension') and name == '_reconstruct_array':
            return _reconstruct_array
        if module == 'deepqmc.types' and name == 'TrainState':
            return TrainState
        if module == 'deepqmc.types' and name == 'Psi':
            return Psi
        if module == 'collections' and name in ('OrderedDict', 'defaultdict', 'deque'):
            return super().find_class(module, name)
        if (module, name) == ('_codecs', 'encode'):      # how protocol <= 2 spells a bytes object (str -> bytes, no side effects)
            return super().find_class(module, name)
        if module == 'builtins' and name in _SAFE_BUILTINS:
            return super().find_class(module, name)
        if module in ('builtins', '__builtin__'):
            raise pickle.UnpicklingError(f'refusing {module}.{name} in a checkpoint')
        return _opaque_class(module, name)


def load(path_or_bytes) -> Tuple[int, TrainState]:
    """CheckpointStore.load (log.py:121-128) without deserialisation onto devices: (step, TrainState) of NumPy arrays."""
    if isinstance(path_or_bytes, (bytes, bytearray)):
        f = io.BytesIO(path_or_bytes)
    else:
        f = open(path_or_bytes, 'rb')
    with f:
        step, state = _Unpickler(f).load()
    if not isinstance(state, TrainState):
        state = TrainState(*state)
    return int(step), state


def _to_dict(tree):
    """haiku parameter containers (dict / FlatMapping stubs) -> {module: {leaf: ndarray}}."""
    if isinstance(tree, Opaque):          # e.g. haiku FlatMapping pickled as (cls, (dict,)) or with a dict state
        for cand in list(tree.args) + [tree.state]:
            if isinstance(cand, dict):
                tree = cand
                break
            if isinstance(cand, (tuple, list)) and cand and isinstance(cand[0], dict):
                tree = cand[0]
                break
    if not isinstance(tree, dict):
        raise ValueError(f'cannot interpret {type(tree)} as a parameter tree')
    return OrderedDict((mod, OrderedDict((leaf, np.asarray(v)) for leaf, v in _to_dict(leaves).items()) if isinstance(leaves, (dict, Opaque))
                        else np.asarray(leaves)) for mod, leaves in tree.items())


def split_states(params, n_states: int = None) -> List[OrderedDict]:
    """The reference stacks the parameter trees of the electronic states on a leading axis (wf/base.py:27); the HIP
    path takes one tree per state (one context each).  `n_states` defaults to the leading extent shared by all leaves."""
    tree = _to_dict(params)
    lead = {v.shape[0] if v.ndim else None for m in tree.values() for v in m.values()}
    if n_states is None:
        if len(lead) != 1 or None in lead:
            raise ValueError('leaves do not share a leading electronic-state axis; pass n_states')
        n_states = lead.pop()
    return [OrderedDict((mod, OrderedDict((leaf, np.asarray(v[s], np.float64)) for leaf, v in leaves.items()))
                        for mod, leaves in tree.items()) for s in range(n_states)]


def load_params(path_or_bytes, n_states: int = None):
    """(step, [parameter tree per electronic state]) ready for `NeuralNetworkWaveFunction.engine / apply`."""
    step, state = load(path_or_bytes)
    return step, split_states(state.params, n_states)


def sampler_states(state: TrainState, molecule: int = 0):
    """The electron sampler state of every electronic state of one molecule as the dicts `MetropolisSampler.sample`
    takes ({'r','psi','age','tau'} of NumPy arrays; move them to the device with torch.as_tensor)."""
    elec = state.sampler['elec'] if isinstance(state.sampler, dict) else state.sampler
    r, age, tau, psi = (np.asarray(elec[k]) if k != 'psi' else elec[k] for k in ('r', 'age', 'tau', 'psi'))
    sign, log = (np.asarray(psi.sign), np.asarray(psi.log)) if hasattr(psi, 'sign') else (np.asarray(psi[0]), np.asarray(psi[1]))
    out = []
    for s in range(r.shape[1]):
        out.append({'r': r[molecule, s], 'psi': Psi(sign[molecule, s].astype(np.int32), log[molecule, s]),
                    'age': age[molecule, s].astype(np.int32), 'tau': np.asarray(tau[molecule, s]).reshape(1)})
    return out
