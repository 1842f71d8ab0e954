"""Engine: one HIP context (`dqmc_ctx`) = one ansatz + parameter set on one GPU.

Host glue only: compiles the layer program (program.py), hands it to the C ABI
(include/dqmc.h) and moves raw device pointers of torch tensors across it.  torch is used
for device memory and streams, nothing else.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .hamil import STAT_KEYS, MolecularHamiltonian
from .program import DqmcSystem, Program, compile_program
from .spec import AnsatzSpec
from .types import PhysicalConfiguration

F32_EPS = float(np.finfo(np.float32).eps)
F64_EPS = float(np.finfo(np.float64).eps)
SAMPLER_STAT_KEYS = ('sampling/acceptance', 'sampling/tau', 'sampling/age/mean', 'sampling/age/max',
                     'sampling/log_psi/mean', 'sampling/log_psi/std', 'sampling/dists/mean')


class DqmcError(RuntimeError):
    pass


def nuclear_energy(coords: np.ndarray, charges: np.ndarray) -> float:
    """sum_{a<b} Z_a Z_b / R_ab (reference physics.py:112-116).  Host-side helper (reports, tests); the
    library recomputes the term on the device from the R of every call."""
    e = 0.0
    for a in range(len(charges)):
        for b in range(a + 1, len(charges)):
            e += charges[a] * charges[b] / float(np.linalg.norm(coords[a] - coords[b]))
    return e


class Engine:
    def __init__(self, spec: AnsatzSpec, hamil: MolecularHamiltonian, params, *, dtype=torch.float32,
                 device='cuda', norm_eps: Optional[float] = None, lib=None, R=None, compiler=None):
        """`R` [n_nuc,3]: the geometry this context is compiled for (default: the Hamiltonian's).  It matters only
        for ansatzes with nuclear tokens, whose nuclear stream is folded into the program; every other ansatz
        takes the geometry per call.  `compiler(params) -> Program` replaces the default `compile_program(spec, ...)`
        (e.g. program_featurewise.compile_featurewise for the reference's test ansatz family); `spec` may then be
        any object with `n_determinants` and `nuclei_tokens`."""
        self.lib = lib if lib is not None else _lib.load()
        self.spec, self.hamil = spec, hamil
        self.dtype = dtype
        self.device = torch.device(device)
        if dtype not in (torch.float32, torch.float64):
            raise ValueError('dtype must be float32 or float64')
        # eps under the safe norm is the compute dtype's machine eps (reference utils.py:79-85)
        self.norm_eps = norm_eps if norm_eps is not None else (F32_EPS if dtype == torch.float32 else F64_EPS)
        self.R0 = np.asarray(hamil.mol.coords if R is None else R, np.float64).reshape(hamil.n_nuc, 3)
        R0_, eps_ = self.R0, self.norm_eps      # (no `self` in the closure: a cycle would leave the context -- tens of GB of
        self._compile = compiler if compiler is not None else (       # workspace -- to the cyclic collector instead of the reference count)
            lambda p_: compile_program(spec, p_, hamil.n_up, hamil.n_down, hamil.n_nuc, R=R0_, eps=eps_))
        self.program: Program = self._compile(params)
        self.N = hamil.n_up + hamil.n_down
        sysd = DqmcSystem(hamil.n_up, hamil.n_down, hamil.n_nuc, spec.n_determinants,
                          0 if dtype == torch.float32 else 1, 0, self.norm_eps, 0.0)
        p = self.program
        self._bufs, self._ops = p.c_bufs(), p.c_ops()
        w = np.ascontiguousarray(p.weights, np.float64)
        it = np.ascontiguousarray(p.itable if p.itable.size else np.zeros(1, np.int32), np.int32)
        ch = np.ascontiguousarray(hamil.ns_valence, np.float64)
        stream = 0
        dev_index = 0
        if self.device.type == 'cuda':
            dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
            stream = torch.cuda.current_stream(self.device).cuda_stream
        ctx = ctypes.c_void_p()
        rc = self.lib.dqmc_create(ctypes.byref(ctx), dev_index, ctypes.c_void_p(stream), ctypes.byref(sysd),
                                  ch.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), self._bufs, len(p.bufs),
                                  self._ops, len(p.ops), w.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), w.size,
                                  it.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), p.itable.size)
        self._ctx = ctx
        self._check(rc)
        self._ecp_phi = None
        pot = getattr(hamil, 'pot', None)
        dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)) if a.size else None
        if pot is not None and hasattr(pot, 'rv_l2'):       # pseudo-Hamiltonian tables -> device
            mask = np.ascontiguousarray(pot.ecp_mask, np.int32)
            self._check(self.lib.dqmc_set_pseudo_hamiltonian(self._ctx, pot.rv_loc.shape[1], pot.r_max, dp(pot.rv_loc), dp(pot.rv_l2),
                                                             mask.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))))
        elif pot is not None:                               # Gaussian-type ECP tables -> device
            self._check(self.lib.dqmc_set_ecp(self._ctx, pot.loc_params.shape[3], dp(pot.loc_params),
                                              pot.nl_params.shape[1], pot.nl_params.shape[3], dp(pot.nl_params)))
        self.R = torch.as_tensor(self.R0, dtype=dtype, device=self.device).contiguous()

    # ------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != 0:
            if getattr(self, '_ctx', None) is None:
                raise DqmcError('engine closed: its HIP context was destroyed (Engine.close / NeuralNetworkWaveFunction.release); '
                                'obtain a new one with NeuralNetworkWaveFunction.engine(params)')
            raise DqmcError(f'dqmc error {rc}: {self.lib.dqmc_last_error().decode()}')

    def close(self):
        if getattr(self, '_ctx', None):
            self.lib.dqmc_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _t(self, x, dtype=None):
        return torch.as_tensor(x, dtype=dtype or self.dtype, device=self.device).contiguous()

    def _R(self, R):
        if R is None:
            return self.R
        R = self._t(R)
        if R.dim() == 3:      # reference tiles R per walker (electron_samplers.py:165-173)
            R = R[0].contiguous()
        if self.spec.nuclei_tokens and not torch.equal(R, self.R):
            raise DqmcError('this ansatz folds the nuclear stream at the geometry of its Hamiltonian; '
                            'build a new engine for a different R')
        return R

    def set_params(self, params):
        """New parameter tree after an optimiser step (same structure)."""
        prog = self._compile(params)
        w = np.ascontiguousarray(prog.weights, np.float64)
        self._check(self.lib.dqmc_set_weights(self._ctx, w.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), w.size))

    # ---- ansatz.apply ---------------------------------------------------
    def wf_eval(self, r, R=None):
        r = self._t(r)
        B = r.shape[0]
        assert r.shape[1:] == (self.N, 3)
        logpsi = torch.empty(B, dtype=self.dtype, device=self.device)
        sign = torch.empty(B, dtype=torch.int32, device=self.device)
        self._check(self.lib.dqmc_wf_eval(self._ctx, r.data_ptr(), self._R(R).data_ptr(), B, logpsi.data_ptr(),
                                          sign.data_ptr()))
        return sign, logpsi

    # ---- hamil.local_energy ---------------------------------------------
    def local_energy(self, phys_conf, rng=None, return_grad=False, ecp_phi=None):
        """`rng` (an int seed or None) keys the random quadrature rotation of the non-local ECP term
        (hamil.py:166, gaussian_type_ecp.py:221); `ecp_phi` [B, n_ecp_nl, N] overrides it with explicit
        angles (parity tests).  Both are ignored without an ECP."""
        if isinstance(phys_conf, PhysicalConfiguration):
            r, R = phys_conf.r, phys_conf.R
        else:
            r, R = phys_conf, None
        r = self._t(r)
        B = r.shape[0]
        assert r.shape[1:] == (self.N, 3)
        if getattr(getattr(self.hamil, 'pot', None), 'nl_params', np.zeros(0)).size:
            if rng is None and ecp_phi is None:      # gaussian_type_ecp.py:183 `assert rng is not None`
                raise DqmcError('a Hamiltonian with a non-local ECP needs `rng` (the seed of the quadrature rotation)')
            if ecp_phi is None and not isinstance(rng, (int, np.integer)):
                raise DqmcError('`rng` must be an integer seed')
            self._ecp_phi = self._t(ecp_phi) if ecp_phi is not None else None      # keep alive during the call
            if self._ecp_phi is not None and self._ecp_phi.shape[0] != B:
                raise DqmcError('ecp_phi must have one row per walker')
            seed = int(rng) & (2 ** 64 - 1) if isinstance(rng, (int, np.integer)) else 0
            self._check(self.lib.dqmc_ecp_rotation(self._ctx, seed,
                                                   self._ecp_phi.data_ptr() if self._ecp_phi is not None else None))
        e = torch.empty(B, dtype=self.dtype, device=self.device)
        st = torch.empty(6, B, dtype=self.dtype, device=self.device)
        grad = torch.empty(B, 3 * self.N, dtype=self.dtype, device=self.device) if return_grad else None
        self._check(self.lib.dqmc_local_energy(self._ctx, r.data_ptr(), self._R(R).data_ptr(), B, e.data_ptr(),
                                               st.data_ptr(), grad.data_ptr() if return_grad else None, None, None))
        stats = {k: st[i] for i, k in enumerate(STAT_KEYS)}
        return (e, stats, grad) if return_grad else (e, stats)

    def psi_and_grad(self, r, R=None):
        """(sign, log|psi|, grad log|psi| [B,N,3]) in one forward-Laplacian pass -- what the reference's
        LangevinSampler obtains with value_and_grad (electron_samplers.py:193-201).  No potential terms and
        no ECP quadrature run (dqmc_psi_grad)."""
        r = self._t(r)
        B = r.shape[0]
        grad = torch.empty(B, 3 * self.N, dtype=self.dtype, device=self.device)
        logpsi = torch.empty(B, dtype=self.dtype, device=self.device)
        sign = torch.empty(B, dtype=torch.int32, device=self.device)
        self._check(self.lib.dqmc_psi_grad(self._ctx, r.data_ptr(), self._R(R).data_ptr(), B, logpsi.data_ptr(),
                                           sign.data_ptr(), grad.data_ptr()))
        return sign, logpsi, grad.reshape(B, self.N, 3)

    # ---- sampler ----------------------------------------------------------
    def mcmc_steps(self, state: Dict[str, torch.Tensor], n_sub: int, *, max_age=None, target_acceptance=0.57,
                   seed: int = 0, noise=None, unif=None, R=None, want_stats=True, return_accept=False):
        """In-place Metropolis sub-steps on state {'r','log','sign','age','tau'} (device tensors)."""
        r, B = state['r'], state['r'].shape[0]
        want = {'r': self.dtype, 'log': self.dtype, 'tau': self.dtype, 'sign': torch.int32, 'age': torch.int32}
        for k, dt in want.items():     # raw device pointers cross the C ABI: the layout must be exactly the documented one
            t = state[k]
            if t.dtype != dt or not t.is_contiguous() or t.device.type != self.device.type:
                raise DqmcError(f"sampler state '{k}' must be a contiguous {dt} tensor on {self.device}")
        if r.shape != (B, self.N, 3) or any(state[k].shape[0] != B for k in ('log', 'sign', 'age')):
            raise DqmcError('sampler state shapes do not match [B, N, 3] / [B]')
        acc = torch.empty(n_sub, B, dtype=torch.uint8, device=self.device) if return_accept else None
        stats = (ctypes.c_double * 7)()
        if noise is not None:
            noise, unif = self._t(noise), self._t(unif)
            assert noise.shape == (n_sub, B, self.N, 3) and unif.shape == (n_sub, B)
        rc = self.lib.dqmc_mcmc_steps(
            self._ctx, r.data_ptr(), state['log'].data_ptr(), state['sign'].data_ptr(), state['age'].data_ptr(),
            state['tau'].data_ptr(), self._R(R).data_ptr(), B, n_sub, -1 if max_age is None else int(max_age),
            -1.0 if target_acceptance is None else float(target_acceptance), int(seed),
            noise.data_ptr() if noise is not None else None, unif.data_ptr() if unif is not None else None,
            acc.data_ptr() if return_accept else None, stats if want_stats else None)
        self._check(rc)
        out = dict(zip(SAMPLER_STAT_KEYS, list(stats))) if want_stats else {}
        return (out, acc) if return_accept else out

    def _check_state(self, state, keys):
        B = state['r'].shape[0]
        want = {'r': self.dtype, 'log': self.dtype, 'tau': self.dtype, 'force': self.dtype, 'sign': torch.int32, 'age': torch.int32}
        for k in keys:     # raw device pointers cross the C ABI: the layout must be exactly the documented one
            t = state[k]
            if t.dtype != want[k] or not t.is_contiguous() or t.device.type != self.device.type:
                raise DqmcError(f"sampler state '{k}' must be a contiguous {want[k]} tensor on {self.device}")
        if state['r'].shape != (B, self.N, 3) or any(state[k].shape[0] != B for k in keys if k not in ('r', 'tau')):
            raise DqmcError('sampler state shapes do not match [B, N, 3] / [B]')
        return B

    def _molz(self, charges):
        z = np.ascontiguousarray(charges, np.float64)
        assert z.shape == (self.hamil.n_nuc,)
        return z, z.ctypes.data_as(ctypes.POINTER(ctypes.c_double))

    def langevin_update(self, r, tau, mol_charges, R=None):
        """LangevinSampler._update (electron_samplers.py:193-208): (sign, log|psi|, cleaned drift [B,N,3])."""
        r = self._t(r)
        B = r.shape[0]
        tau = self._t(tau).reshape(1)
        z, zp = self._molz(mol_charges)
        logpsi = torch.empty(B, dtype=self.dtype, device=self.device)
        sign = torch.empty(B, dtype=torch.int32, device=self.device)
        force = torch.empty(B, self.N, 3, dtype=self.dtype, device=self.device)
        self._check(self.lib.dqmc_langevin_update(self._ctx, r.data_ptr(), self._R(R).data_ptr(), zp, B, tau.data_ptr(),
                                                  logpsi.data_ptr(), sign.data_ptr(), force.data_ptr()))
        return sign, logpsi, force

    def langevin_steps(self, state, n_sub, mol_charges, *, max_age=None, target_acceptance=0.57, seed=0, noise=None, unif=None,
                       R=None, want_stats=True, return_accept=False):
        """In-place Langevin sub-steps on state {'r','log','sign','age','tau','force'} (dqmc_langevin_steps)."""
        B = self._check_state(state, ('r', 'log', 'sign', 'age', 'tau', 'force'))
        acc = torch.empty(n_sub, B, dtype=torch.uint8, device=self.device) if return_accept else None
        stats = (ctypes.c_double * 7)()
        if noise is not None:
            noise, unif = self._t(noise), self._t(unif)
            assert noise.shape == (n_sub, B, self.N, 3) and unif.shape == (n_sub, B)
        z, zp = self._molz(mol_charges)
        self._check(self.lib.dqmc_langevin_steps(
            self._ctx, state['r'].data_ptr(), state['log'].data_ptr(), state['sign'].data_ptr(), state['age'].data_ptr(),
            state['force'].data_ptr(), state['tau'].data_ptr(), self._R(R).data_ptr(), zp, B, n_sub,
            -1 if max_age is None else int(max_age), -1.0 if target_acceptance is None else float(target_acceptance), int(seed),
            noise.data_ptr() if noise is not None else None, unif.data_ptr() if unif is not None else None,
            acc.data_ptr() if return_accept else None, stats if want_stats else None))
        out = dict(zip(SAMPLER_STAT_KEYS, list(stats))) if want_stats else {}
        return (out, acc) if return_accept else out

    def exchange_step(self, state, up_idx, down_idx, unif, R=None, return_accept=False):
        """One opposite-spin exchange step, in place on state {'r','log','sign','age','tau'} (dqmc_exchange_step)."""
        B = self._check_state(state, ('r', 'log', 'sign', 'age', 'tau'))
        for name, idx, n in (('up_idx', up_idx, self.hamil.n_up), ('down_idx', down_idx, self.hamil.n_down)):
            host = idx if isinstance(idx, np.ndarray) else (idx if isinstance(idx, torch.Tensor) and idx.device.type == 'cpu' else None)
            if host is not None and len(host) and (int(host.min()) < 0 or int(host.max()) >= n):     # device tensors: clamped in the kernel
                raise DqmcError(f'{name} must lie in [0, {n})')
        up = torch.as_tensor(up_idx, device=self.device).to(torch.int32).contiguous()
        dn = torch.as_tensor(down_idx, device=self.device).to(torch.int32).contiguous()
        u = self._t(unif)
        assert up.shape == dn.shape == u.shape == (B,)
        acc = torch.empty(B, dtype=torch.uint8, device=self.device) if return_accept else None
        stats = (ctypes.c_double * 7)()
        self._check(self.lib.dqmc_exchange_step(self._ctx, state['r'].data_ptr(), state['log'].data_ptr(), state['sign'].data_ptr(),
                                                state['age'].data_ptr(), state['tau'].data_ptr(), self._R(R).data_ptr(), B,
                                                up.data_ptr(), dn.data_ptr(), u.data_ptr(), acc.data_ptr() if return_accept else None,
                                                stats))
        out = dict(zip(SAMPLER_STAT_KEYS, list(stats)))
        return (out, acc) if return_accept else out

    def check_walker_vector(self, name, t, n=None):
        """A [B] tensor about to cross the C ABI as a raw pointer: dtype, layout and device must be the context's."""
        if not isinstance(t, torch.Tensor) or t.dtype != self.dtype or not t.is_contiguous() or t.device.type != self.device.type \
                or t.dim() != 1 or (n is not None and t.shape[0] != n):
            raise DqmcError(f"'{name}' must be a contiguous 1-d {self.dtype} tensor on {self.device}"
                            + (f' of length {n}' if n is not None else ''))
        return t

    def energy_record(self, e_loc, w=None):
        self.check_walker_vector('e_loc', e_loc)
        if w is not None:
            self.check_walker_vector('w', w, e_loc.shape[0])
        rec = (ctypes.c_double * 7)()
        self._check(self.lib.dqmc_energy_stats(self._ctx, e_loc.data_ptr(), w.data_ptr() if w is not None else None,
                                               e_loc.shape[0], rec))
        return np.array(list(rec))

    def merge_energy_records(self, records: np.ndarray):
        records = np.ascontiguousarray(records, np.float64).reshape(-1, 7)
        out = (ctypes.c_double * 5)()
        self._check(self.lib.dqmc_merge_energy_stats(records.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                     records.shape[0], out))
        return dict(zip(('local_energy/mean', 'local_energy/std', 'local_energy/min', 'local_energy/max',
                         'local_energy/weighted_mean'), list(out)))

    # ---- debug / timing ---------------------------------------------------
    def debug_read(self, name_or_idx, B: int):
        TP = self.lib.dqmc_debug_lanes(self._ctx)
        if name_or_idx == 'logdet':
            shape, idx = (B, self.spec.n_determinants, TP), -1
        elif name_or_idx == 'sign_k':
            shape, idx = (B, self.spec.n_determinants), -2
        elif name_or_idx == 'kappa':      # conditioning record per walker (kernels_head.hip), last Laplacian-mode call
            shape, idx = (B,), -4
        else:
            idx = self.program.buf_names[name_or_idx] if isinstance(name_or_idx, str) else int(name_or_idx)
            rows, width = self.program.bufs[idx]
            shape = (B, rows, TP, width)
        out = np.empty(shape, np.float64)
        self._check(self.lib.dqmc_debug_read(self._ctx, idx, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), out.size))
        return out

    def last_refined(self) -> int:
        """Walkers the last local-energy call re-evaluated in float64 (dqmc_last_refined)."""
        return int(self.lib.dqmc_last_refined(self._ctx))

    def refine_info(self) -> dict:
        """State of the float64 refinement after the last local-energy call (dqmc_refine_info)."""
        out = (ctypes.c_double * 4)()
        self._check(self.lib.dqmc_refine_info(self._ctx, out))
        # ('error_per_score': the scale m of the exponential error model, float32 error = m x score x xi)
        return {'mode': int(out[0]), 'score_threshold': out[1], 'error_per_score': out[2], 'direct_f64_calls_left': int(out[3])}

    def refine_counters(self) -> dict:
        """Running counts of the context (dqmc_refine_counters): local-energy calls, calls evaluated in float64 whole,
        calibration probes, walkers re-evaluated in float64."""
        out = (ctypes.c_int64 * 4)()
        self._check(self.lib.dqmc_refine_counters(self._ctx, out))
        return {'calls': int(out[0]), 'direct_f64_calls': int(out[1]), 'probe_calls': int(out[2]), 'walkers_refined': int(out[3])}

    def substep_kernel(self) -> str:
        """Name of the plan-specialised sub-step kernel bound to this context's program (dqmc_substep_kernel), '' if none."""
        buf = ctypes.create_string_buffer(128)
        rc = self.lib.dqmc_substep_kernel(self._ctx, buf, 128)
        if rc < 0:
            self._check(rc)
        return buf.value.decode()

    def refine_scores(self, n: int) -> np.ndarray:
        """Error-predictor scores of the first n walkers of the last float32 forward-Laplacian pass (dqmc_refine_scores)."""
        out = np.empty(int(n), np.float64)
        self._check(self.lib.dqmc_refine_scores(self._ctx, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), int(n)))
        return out

    def last_chunks(self) -> dict:
        """Walker chunks the last local-energy call was split into (dqmc_last_chunks): the context's own pass and its
        float64 twin's."""
        out = (ctypes.c_int * 2)()
        self._check(self.lib.dqmc_last_chunks(self._ctx, out))
        return {'own': int(out[0]), 'twin': int(out[1])}

    def ecp_counts(self) -> dict:
        """(nucleus, electron) pairs of the last mixed-precision ECP quadrature by class (dqmc_ecp_counts)."""
        out = (ctypes.c_int64 * 3)()
        self._check(self.lib.dqmc_ecp_counts(self._ctx, out))
        return {'f32': int(out[0]), 'f64': int(out[1]), 'dropped': int(out[2])}

    def set_option(self, name: str, value: int):
        self._check(self.lib.dqmc_set_option(self._ctx, name.encode(), int(value)))
        self.__dict__.setdefault('_options', {})[name] = int(value)      # (what the caller chose: apply(return_mos=True) restores it)

    def get_option(self, name: str, default: int) -> int:
        """The value last set through set_option (the library has no getter), else `default`."""
        return self.__dict__.get('_options', {}).get(name, default)

    def timing(self, enable=True):
        self._check(self.lib.dqmc_timing_enable(self._ctx, int(enable)))

    def timing_reset(self):
        self._check(self.lib.dqmc_timing_reset(self._ctx))

    def timing_report(self):
        buf = ctypes.create_string_buffer(1024)
        self._check(self.lib.dqmc_timing_names(self._ctx, buf, 1024))
        out = {}
        for nm in filter(None, buf.value.decode().split(',')):
            ms, n, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
            self._check(self.lib.dqmc_timing_get(self._ctx, nm.encode(), ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)))
            ex = ctypes.c_double()
            self._check(self.lib.dqmc_timing_get_executed(self._ctx, nm.encode(), ctypes.byref(ex)))
            out[nm] = {'ms': ms.value, 'launches': n.value, 'flops': fl.value, 'flops_executed': ex.value}
        return out
