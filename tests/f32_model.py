"""CPU model of the float32 build's rounding (test infrastructure; not a product path).

`Interp32` runs the layer program exactly like oracle/program_interp.py but stores every
activation buffer and the weights in float32 and multiplies in float32 (BLAS sgemm ~ the
v_mfma_f32_16x16x4_f32 FMA chain), while the pieces the HIP library keeps in double stay in
double: pair features and envelopes are evaluated in double and rounded on store, slogdet /
CI sum / E_loc run in double on the float32 Slater matrices.  It predicts the error
distribution of the f32 build against the f64 oracle without a GPU, and lets precision
policies be compared: `hi` names buffers that are kept in double (and whose producing ops run
in double), which is how the "which stage needs more bits" question is answered.
"""
import numpy as np

from oracle.program_interp import Interp, lanes


class Interp32(Interp):
    def __init__(self, prog, charges, norm_eps, hi=(), split_hi=()):
        super().__init__(prog, charges, norm_eps)
        self.hi_idx = {prog.buf_names[n] for n in hi if n in prog.buf_names}
        self.w64 = self.w
        self.w32 = self.w.astype(np.float32)

    def run(self, r, R, laplacian):
        p, N = self.p, self.N
        r = np.asarray(r, np.float64)
        R = np.asarray(R, np.float64)
        B = r.shape[0]
        self.T, self.TP = lanes(N, laplacian)
        self.lap = laplacian
        self.bufs = [np.zeros((B, rows, self.TP, width), np.float64 if k in self.hi_idx else np.float32)
                     for k, (rows, width) in enumerate(p.bufs)]
        self.r, self.R, self.B = r, R, B
        out = None
        for op in p.ops:
            out = getattr(self, f'op_{op.kind}')(op) or out
        return out

    def op_3(self, op):       # LINEAR in the precision of its destination
        dst = op.i[17]
        hi = dst in self.hi_idx
        self.w = self.w64 if hi else self.w32
        saved = None
        if not hi:
            # inputs that live in double (hi producers) are rounded when a float32 op reads them
            saved = {}
            for p_ in range(op.i[0]):
                src = op.i[1 + 4 * p_]
                if self.bufs[src].dtype == np.float64:
                    saved[src] = self.bufs[src]
                    self.bufs[src] = self.bufs[src].astype(np.float32)
        super().op_3(op)
        if saved:
            for k, v in saved.items():
                self.bufs[k] = v
        self.w = self.w64

    def op_8(self, op):       # ORBITALS: double arithmetic, rounded on store (k_orbitals)
        bf = op.i[0]
        keep = self.bufs[bf]
        self.bufs[bf] = keep.astype(np.float64)
        super().op_8(op)
        self.bufs[bf] = keep

    def op_9(self, op):       # SLOGDET: double arithmetic on the stored Slater matrices
        orb = op.i[0]
        keep = self.bufs[orb]
        self.bufs[orb] = keep.astype(np.float64)
        super().op_9(op)
        self.bufs[orb] = keep

    def op_10(self, op):
        jas = op.i[0]
        keep = None
        if jas >= 0:
            keep = self.bufs[jas]
            self.bufs[jas] = keep.astype(np.float64)
        out = super().op_10(op)
        if keep is not None:
            self.bufs[jas] = keep
        return out


def error_profile(e, ref):
    rel = np.abs(e - ref) / np.maximum(1.0, np.abs(ref))
    q = lambda x: float(np.quantile(rel, x))
    return {'median': q(0.5), 'p90': q(0.9), 'p99': q(0.99), 'max': float(rel.max()),
            'frac_within_1e-5': float((rel < 1e-5).mean())}, rel


class InterpAcc(Interp32):
    """Interp32 with the ACCUMULATION of the linear layers modelled as the matrix pipe does it (round 5): the products of a
    k-step of 4 are summed exactly and added to a float32 accumulator that is rounded after every step (v_mfma_f32_16x16x4_f32;
    the bf16 path rounds once per 32 k and product pass).  `chunk` = 0: one accumulator chain over the whole K of the layer (the
    kernels until round 5); `chunk` = 16 / 32: a FRESH accumulator per chunk, chunk results added to the running sum
    (kernel_linear.hip: FRESH / BF_FRESH); `comp`: the running sum exact (what a compensated outer sum would give; chunk = 4
    with comp = exact accumulation).  `exact_ops`: indices of LINEAR ops that accumulate exactly whatever the others do.
    It predicted what the MI355X then measured: error scale m x 0.6-0.7 for chunks of 16-32 (device: x 0.70 LiH, x 0.65 N2)."""

    def __init__(self, prog, charges, norm_eps, hi=(), chunk=0, comp=False, exact_ops=()):
        super().__init__(prog, charges, norm_eps, hi=hi)
        self.chunk, self.comp, self.exact_ops = chunk, comp, set(exact_ops)

    def run(self, r, R, laplacian):
        self._op_index = {id(op): k for k, op in enumerate(self.p.ops)}
        return super().run(r, R, laplacian)

    def op_3(self, op):
        import math
        from deepqmc_amd import program as P
        i = op.i
        dst, dr0, dc0, nrows, nout, woff, boff, act, res, rr0, rnorm = i[17:28]
        if dst in self.hi_idx:
            return super().op_3(op)
        chunk, comp = (4, True) if self._op_index[id(op)] in self.exact_ops else (self.chunk, self.comp)
        f32 = np.float32
        nout_p = P.pad4(nout)
        run = np.zeros((self.B, nrows, self.TP, nout_p))
        part = np.zeros_like(run)
        kcount, o = 0, woff
        w = self.w32.astype(np.float64)
        for p_ in range(i[0]):
            src, r0, K, bc = i[1 + 4 * p_:5 + 4 * p_]
            Kp = P.pad4(K)
            W = w[o:o + Kp * nout_p].reshape(Kp, nout_p)
            o += Kp * nout_p
            X = self.bufs[src][:, r0:r0 + (1 if bc else nrows), :, :Kp].astype(np.float32).astype(np.float64)
            for k0 in range(0, Kp, 4):
                part = (part + X[..., k0:k0 + 4] @ W[k0:k0 + 4]).astype(f32).astype(np.float64)
                kcount += 4
                if chunk and kcount % chunk == 0:
                    run = run + part if comp else (run + part).astype(f32).astype(np.float64)
                    part = np.zeros_like(run)
        acc = part if not chunk else (run + part if comp else (run + part).astype(f32).astype(np.float64))
        if boff >= 0:
            acc[:, :, 0, :] += self.w32[boff:boff + nout_p].astype(np.float64)
        y = self._ll(acc)
        v = y[..., 0]
        if act == 1:
            t = np.tanh(v)
            y = self._chain(y, t, 1 - t * t, -2 * t * (1 - t * t))
        elif act == 2:
            s = 1 / (1 + np.exp(-v))
            y = self._chain(y, v * s, s * (1 + v * (1 - s)), s * (1 - s) * (2 + v * (1 - 2 * s)))
        elif act == 3:
            s = 1 / (1 + np.exp(-v))
            y = self._chain(y, np.logaddexp(0.0, v) - math.log(2.0), s, s * (1 - s))
        elif act == 4:
            t = np.tanh(v / 4)
            y = self._chain(y, 1 + 2 * t, (1 - t * t) / 2, -t * (1 - t * t) / 4)
        y = self._ll(y)
        if res >= 0:
            y = (self.bufs[res][:, rr0:rr0 + nrows, :, dc0:dc0 + nout_p] + y) * (1 / math.sqrt(2.0) if rnorm else 1.0)
        self.bufs[dst][:, dr0:dr0 + nrows, :, dc0:dc0 + nout_p] = y
