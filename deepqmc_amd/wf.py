"""`NeuralNetworkWaveFunction`: the ansatz surface (`init` / `apply`) of the reference
(src/deepqmc/types.py:107-150, wf/nn_wave_function.py:42-173) backed by the HIP engine.

    ansatz = NeuralNetworkWaveFunction(hamil, 'paulinet')
    params = ansatz.init(seed)                 # haiku-shaped parameter tree (NumPy leaves)
    psi    = ansatz.apply(params, phys_conf)   # Psi(sign[B], log[B]) on the GPU

`apply` evaluates a whole batch (the reference vmaps its single-walker apply,
sampling/electron_samplers.py:76-81).  One engine (HIP context) is kept per live parameter
tree OBJECT: the cache holds a reference to the tree and matches it with `is`, so a recycled
`id()` can never alias an engine holding other weights.  Pass a new tree object after an
optimiser step and its weights are uploaded into the least recently used context; leaves
mutated in place are not detected -- call `invalidate(params)` after doing that.
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
        self._engines = []       # [(params tree, Engine)], most recently used last (one per electronic state)
        self.max_engines = 8
        self._evictions = 0

    def init(self, rng=0, phys_conf=None, **kw):
        """types.py:121-133.  `rng`: an integer seed (the JAX key stream is not reproduced)."""
        h = self.hamil
        return init_params(self.spec, h.n_up, h.n_down, h.n_nuc, seed=int(rng), **kw)

    def _geometry_key(self, R):
        """Ansatzes with nuclear tokens fold the geometry into the program: one context per (tree, geometry).
        Everything else takes R per call and shares one context across geometries."""
        if R is None or not self.spec.nuclei_tokens:
            return None
        import numpy as np
        R = np.asarray(R.detach().cpu() if hasattr(R, 'detach') else R, np.float64).reshape(-1, self.hamil.n_nuc, 3)
        if R.shape[0] > 1 and not np.array_equal(R, np.broadcast_to(R[:1], R.shape)):
            raise ValueError('an ansatz with nuclear tokens takes ONE geometry per call (per-walker geometries differ)')
        R = np.ascontiguousarray(R[0])
        return None if np.array_equal(R, self.hamil.mol.coords) else R.tobytes()

    def engine(self, params, R=None) -> Engine:
        """One HIP context per live parameter tree (e.g. per electronic state, the leading `S` axis of
        the reference's params, wf/base.py:27).  The cache entry keeps the tree alive and is matched by
        identity; once `max_engines` trees are alive the least recently used context is re-targeted
        with `set_params` (no new allocation)."""
        gkey = self._geometry_key(R)
        for k, (tree, key, eng) in enumerate(self._engines):
            if tree is params and key == gkey:
                self._engines.append(self._engines.pop(k))
                return eng
        reuse = [k for k, (_, key, _) in enumerate(self._engines) if key == gkey]
        if len(self._engines) >= self.max_engines and reuse:
            _, _, eng = self._engines.pop(reuse[0])
            eng.set_params(params)
        else:
            if len(self._engines) >= self.max_engines:
                # the evicted context leaves the cache; its device memory (workspace, float64 twin, captured graphs) goes back
                # when its LAST reference goes -- now, if the cache held the only one (Engine keeps no reference cycle:
                # tests/test_facade_emu.py::test_engine_is_released_by_reference_count), otherwise when the caller that
                # still holds it drops it.  Nothing is closed under a caller's feet.
                self._engines.pop(0)
                self._evictions += 1
                if self._evictions == 4 * self.max_engines:
                    import warnings
                    warnings.warn(f'NeuralNetworkWaveFunction: {self._evictions} HIP contexts evicted from a cache of '
                                  f'{self.max_engines}; raise `max_engines` to (geometries x states) in use')
            import numpy as np
            R0 = None if gkey is None else np.frombuffer(gkey, np.float64).reshape(self.hamil.n_nuc, 3)
            eng = Engine(self.spec, self.hamil, params, dtype=self.dtype, device=self.device,
                         norm_eps=self.norm_eps, lib=self._lib, R=R0)
        self._engines.append((params, gkey, eng))
        return eng

    def release(self):
        """Close every cached HIP context (device workspaces, twins, captured graphs) NOW, whoever still refers to them:
        an `Engine` obtained from `engine()` before this call raises a clear "engine closed" error afterwards."""
        while self._engines:
            self._engines.pop()[2].close()

    def invalidate(self, params=None):
        """Re-upload the weights of `params` (all cached trees if None) after their leaves were changed in place."""
        for tree, _, eng in self._engines:
            if params is None or tree is params:
                eng.set_params(tree)

    def apply(self, params, phys_conf, return_mos: bool = False) -> Psi:
        """types.py:135-150 (batched)."""
        r = phys_conf.r if isinstance(phys_conf, PhysicalConfiguration) else phys_conf
        R = phys_conf.R if isinstance(phys_conf, PhysicalConfiguration) else None
        eng = self.engine(params, R)
        if return_mos:
            # wf/nn_wave_function.py:127,144-145: the (backflow-transformed) molecular orbitals the determinants are taken of,
            # `(orb_up, orb_down)` -- the pretraining hook.  They are the Slater-matrix buffer of the value path: read back
            # through the debug interface (a host copy: this is not a hot path) as [B, K, n_up | n_down, n_orb].
            if not self.spec.full_determinant:
                raise NotImplementedError('return_mos (wf/nn_wave_function.py:144): only full-determinant programs expose '
                                          'their orbital matrices; spin-factorised ones store block-diagonal matrices')
            import numpy as np
            prev_fused = eng.get_option('fused', 1)
            eng.set_option('fused', 0)              # every activation buffer stays readable (include/dqmc.h: "fused")
            try:
                eng.wf_eval(r, R)
                B = torch.as_tensor(r).shape[0]
                N, K = self.hamil.n_elec, self.spec.n_determinants
                A = eng.debug_read('orbitals', B)[:, :, 0, :N * N].reshape(B, K, N, N)     # rows = electrons, columns = orbitals
            finally:
                eng.set_option('fused', prev_fused)      # (the caller's own choice, not the default)
            A = torch.as_tensor(np.ascontiguousarray(A), dtype=self.dtype, device=self.device)
            return A[:, :, :self.hamil.n_up], A[:, :, self.hamil.n_up:]
        sign, log = eng.wf_eval(r, R)
        return Psi(sign, log)

    __call__ = apply
