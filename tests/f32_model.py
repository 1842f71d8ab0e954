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
