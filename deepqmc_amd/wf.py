"""`NeuralNetworkWaveFunction`: the ansatz surface (`init` / `apply`) of the reference
(src/deepqmc/types.py:107-150, wf/nn_wave_function.py:42-173) backed by the HIP engine.

    ansatz = NeuralNetworkWaveFunction(hamil, 'paulinet')
    params = ansatz.init(seed)                 # haiku-shaped parameter tree (NumPy leaves)
    psi    = ansatz.apply(params, phys_conf)   # Psi(sign[B], log[B]) on the GPU

`apply` evaluates a whole batch (the reference vmaps its single-walker apply,
sampling/electron_samplers.py:76-81).  One engine (HIP context) is cached per parameter
tree identity; pass a new tree after an optimiser step and its weights are uploaded.
"""
from __future__ import annotations

import torch

from .engine import Engine
from .hamil import MolecularHamiltonian
from .params import init_params
from .spec import ANSATZES, AnsatzSpec
from .types import PhysicalConfiguration, Psi


class NeuralNetworkWaveFunction:
    def __init__(self, hamil: MolecularHamiltonian, spec='paulinet', *, dtype=torch.float32, device='cuda',
                 norm_eps=None, lib=None):
        self.hamil = hamil
        if spec == 'transpsiformer':
            self.spec: AnsatzSpec = ANSATZES[spec](hamil.mol.charges)      # atom types of the nuclei embedding
        else:
            self.spec = ANSATZES[spec]() if isinstance(spec, str) else spec
        self.dtype, self.device, self.norm_eps, self._lib = dtype, device, norm_eps, lib
        self._engines = {}       # id(params) -> Engine, most recently used last (one per electronic state)
        self.max_engines = 8

    def init(self, rng=0, phys_conf=None, **kw):
        """types.py:121-133.  `rng`: an integer seed (the JAX key stream is not reproduced)."""
        h = self.hamil
        return init_params(self.spec, h.n_up, h.n_down, h.n_nuc, seed=int(rng), **kw)

    def engine(self, params) -> Engine:
        """One HIP context per live parameter tree (e.g. per electronic state, the leading `S` axis of
        the reference's params, wf/base.py:27); the least recently used context is re-targeted with
        `set_params` once `max_engines` trees are alive."""
        key = id(params)
        eng = self._engines.pop(key, None)
        if eng is None:
            if len(self._engines) >= self.max_engines:
                old_key = next(iter(self._engines))
                eng = self._engines.pop(old_key)
                eng.set_params(params)
            else:
                eng = Engine(self.spec, self.hamil, params, dtype=self.dtype, device=self.device,
                             norm_eps=self.norm_eps, lib=self._lib)
        self._engines[key] = eng
        return eng

    def apply(self, params, phys_conf, return_mos: bool = False) -> Psi:
        """types.py:135-150 (batched)."""
        if return_mos:
            raise NotImplementedError('return_mos is a pretraining hook, outside the hot path')
        r = phys_conf.r if isinstance(phys_conf, PhysicalConfiguration) else phys_conf
        R = phys_conf.R if isinstance(phys_conf, PhysicalConfiguration) else None
        sign, log = self.engine(params).wf_eval(r, R)
        return Psi(sign, log)

    __call__ = apply
