"""Reader of the reference's checkpoint format (log.py:114-128) without jax / deepqmc installed: a file is written
exactly as the reference would -- `pickle.dump((step, TrainState(sampler, params, opt)))` with classes that live in
(fake, test-only) modules named `deepqmc.types`, `jax._src.array` and `kfac_jax`, arrays that pickle themselves the
way jax.Array does -- then the fake modules are removed and the file is read back by deepqmc_amd.checkpoint."""
import pickle
import sys
import types
from collections import namedtuple

import numpy as np
import pytest

from deepqmc_amd import checkpoint
from deepqmc_amd.params import init_params
from deepqmc_amd.spec import paulinet


def write_reference_style_checkpoint(path, trees, r, sign, log, age, tau, step=4200):
    mods = {}

    def fake(name):
        m = types.ModuleType(name)
        mods[name] = m
        sys.modules[name] = m
        return m

    jax_arr = fake('jax._src.array')
    fake('jax._src'); fake('jax')
    dq = fake('deepqmc.types'); fake('deepqmc')
    kf = fake('kfac_jax')

    def _reconstruct_array(fun, args, arr_state, aval_state):      # the writer side never calls it
        raise AssertionError

    _reconstruct_array.__module__ = 'jax._src.array'
    _reconstruct_array.__qualname__ = '_reconstruct_array'
    jax_arr._reconstruct_array = _reconstruct_array

    class ArrayImpl:                                               # pickles like jax.Array (jax/_src/array.py __reduce__)
        def __init__(self, v):
            self.v = np.asarray(v)

        def __reduce__(self):
            fun, args, arr_state = self.v.__reduce__()
            return _reconstruct_array, (fun, args, arr_state, {'weak_type': False})

    TrainState = namedtuple('TrainState', 'sampler params opt')
    TrainState.__module__ = 'deepqmc.types'; TrainState.__qualname__ = 'TrainState'
    dq.TrainState = TrainState
    Psi = namedtuple('Psi', 'sign log')
    Psi.__module__ = 'deepqmc.types'; Psi.__qualname__ = 'Psi'
    dq.Psi = Psi

    class OptState:                                                # stands for the KFAC optimiser state
        def __init__(self):
            self.damping = ArrayImpl(np.float32(1e-3))
    OptState.__module__ = 'kfac_jax'; OptState.__qualname__ = 'OptState'
    kf.OptState = OptState

    S = len(trees)
    stacked = {mod: {leaf: ArrayImpl(np.stack([np.asarray(t[mod][leaf], np.float32) for t in trees])) for leaf in trees[0][mod]}
               for mod in trees[0]}
    sampler = {'elec': {'r': ArrayImpl(r), 'psi': Psi(ArrayImpl(sign), ArrayImpl(log)), 'age': ArrayImpl(age), 'tau': ArrayImpl(tau)},
               'nuc': {'R': ArrayImpl(np.zeros((1, 2, 3), np.float32))}, 'update_nuc_counter': ArrayImpl(np.int32(0))}
    try:
        with open(path, 'wb') as f:
            pickle.dump((step, TrainState(sampler, stacked, OptState())), f)
    finally:
        for name in mods:
            sys.modules.pop(name, None)
    assert 'jax' not in sys.modules and 'deepqmc' not in sys.modules
    return S


def test_reads_reference_checkpoint(tmp_path):
    spec = paulinet()
    trees = [init_params(spec, 2, 2, 2, seed=s, perturb_envelopes=0.1) for s in range(3)]
    rng = np.random.default_rng(0)
    M, S, B, N = 1, 3, 5, 4
    r = rng.standard_normal((M, S, B, N, 3)).astype(np.float32)
    sign = np.sign(rng.standard_normal((M, S, B))).astype(np.float32)
    log = rng.standard_normal((M, S, B)).astype(np.float32)
    age = rng.integers(0, 4, (M, S, B)).astype(np.int32)
    tau = rng.random((M, S)).astype(np.float32)
    path = tmp_path / 'chkpt-4200.pt'
    write_reference_style_checkpoint(path, trees, r, sign, log, age, tau)
    raw = open(path, 'rb').read()
    assert b'jax._src.array' in raw and b'deepqmc.types' in raw     # the file really names the reference's classes
    step, state = checkpoint.load(path)
    assert step == 4200 and isinstance(state, checkpoint.TrainState)
    assert isinstance(state.opt, checkpoint.Opaque)                 # optimiser state: carried, not interpreted
    step2, per_state = checkpoint.load_params(path)
    assert step2 == 4200 and len(per_state) == 3
    for s in range(3):
        assert list(per_state[s]) == list(trees[s])                 # module order and names survive
        for mod in trees[s]:
            for leaf in trees[s][mod]:
                np.testing.assert_array_equal(per_state[s][mod][leaf], np.asarray(trees[s][mod][leaf], np.float32).astype(np.float64))
    states = checkpoint.sampler_states(state)
    assert len(states) == 3
    np.testing.assert_array_equal(states[1]['r'], r[0, 1])
    np.testing.assert_array_equal(states[2]['psi'].log, log[0, 2])
    np.testing.assert_array_equal(states[0]['age'], age[0, 0])
    assert states[0]['tau'].shape == (1,) and states[0]['tau'][0] == tau[0, 0]


def test_loaded_parameters_drive_the_engine(tmp_path):
    """A checkpoint's parameters compile into the same layer program as the tree they came from."""
    from deepqmc_amd.program import compile_program
    spec = paulinet()
    tree = init_params(spec, 2, 2, 2, seed=7, perturb_envelopes=0.1)
    path = tmp_path / 'chkpt-1.pt'
    z = np.zeros((1, 1, 2))
    write_reference_style_checkpoint(path, [tree], np.zeros((1, 1, 2, 4, 3)), z, z, z.astype(np.int32), np.zeros((1, 1)))
    _, (loaded,) = checkpoint.load_params(path)
    a = compile_program(spec, loaded, 2, 2, 2)
    f32 = {m: {k: np.asarray(v, np.float32).astype(np.float64) for k, v in lv.items()} for m, lv in tree.items()}
    b = compile_program(spec, f32, 2, 2, 2)
    np.testing.assert_array_equal(a.weights, b.weights)


def test_refuses_code(tmp_path):
    import os
    evil = pickle.dumps((1, (None, {'m': {'w': np.zeros(1)}}, None)))
    step, st = checkpoint.load(evil)                                # plain tuples are fine
    assert step == 1

    class X:
        def __reduce__(self):
            return (eval, ('1+1',))
    with pytest.raises(pickle.UnpicklingError):
        checkpoint.load(pickle.dumps((1, X())))
    y = pickle.dumps((1, (None, None, os.getcwd)))                  # a foreign callable becomes an inert record, never called
    _, st = checkpoint.load(y)
    assert isinstance(st.opt, type) or isinstance(st.opt, checkpoint.Opaque) or st.opt is not os.getcwd


def _stack_global(module, name, arg):
    """Protocol-4 pickle of (1, <module:name>(arg)) written by hand: STACK_GLOBAL resolves dotted names."""
    import pickletools  # noqa: F401  (documentation of the opcodes used)
    def u(b):
        b = b.encode()
        return b'\x8c' + bytes([len(b)]) + b
    return (b'\x80\x04' + b'K\x01' + u(module) + u(name) + b'\x93' + u(arg) + b'\x85' + b'R' + b'\x86' + b'.')


@pytest.mark.parametrize('module', ['numpy._core.numeric', 'numpy.core.numeric', 'numpy._core.multiarray', 'numpy'])
def test_refuses_dotted_names_under_allowed_modules(module):
    """STACK_GLOBAL ('numpy._core.numeric', 'builtins.eval') used to come back as eval."""
    payload = _stack_global(module, 'builtins.eval', '1+1')
    with pytest.raises(pickle.UnpicklingError):
        checkpoint.load(payload)


@pytest.mark.parametrize('name', ['save', 'load', 'fromfile', 'savetxt', 'loadtxt', 'memmap'])
def test_refuses_numpy_callables(tmp_path, name):
    """the getattr(np, name) fallback exposed every NumPy top-level callable."""
    target = tmp_path / 'x.npy'
    payload = _stack_global('numpy', name, str(target))
    with pytest.raises(pickle.UnpicklingError):
        checkpoint.load(payload)
    assert not target.exists()


def test_numpy_scalars_dtypes_and_protocol5_arrays_still_load():
    obj = (3, (None, {'m': {'w': np.arange(6, dtype=np.float32).reshape(2, 3), 's': np.float64(2.5), 'i': np.int32(7),
                            'f': np.asfortranarray(np.ones((2, 2)))}}, None))
    for proto in (2, 4, 5):
        step, st = checkpoint.load(pickle.dumps(obj, protocol=proto))
        assert step == 3
        np.testing.assert_array_equal(st.params['m']['w'], obj[1][1]['m']['w'])
        assert st.params['m']['s'] == 2.5 and st.params['m']['i'] == 7
