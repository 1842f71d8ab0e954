"""NumPy float64 interpreter of the layer program (oracle; test infrastructure only).

Executes the `dqmc_op` list of deepqmc_amd/program.py with the forward-Laplacian rules of
SURVEY.md appendix C (what folx.ForwardLaplacianOperator computes for the reference,
conf/hamil/qc_forward_laplacian.yaml:7-10), buffer by buffer in the device layout
real[B][rows][TP][width].  It is the per-op reference the HIP kernels are compared with
(through dqmc_debug_read) and is itself checked against the autograd oracle
(oracle/physics.py) in tests/test_program_interp.py.
"""
from __future__ import annotations

import math

import numpy as np

from deepqmc_amd import program as P


def lanes(N: int, laplacian: bool):
    if not laplacian:
        return 1, 1
    T = 3 * N + 2
    return T, (T + 15) // 16 * 16


class Interp:
    def __init__(self, prog: P.Program, charges, norm_eps: float):
        self.p = prog
        self.N = prog.n_up + prog.n_down
        self.charges = np.asarray(charges, np.float64)
        self.eps = norm_eps
        self.w = prog.weights
        self.it = prog.itable

    # ------------------------------------------------------------------
    def run(self, r, R, laplacian: bool):
        p, N = self.p, self.N
        r = np.asarray(r, np.float64)
        R = np.asarray(R, np.float64)
        B = r.shape[0]
        self.T, self.TP = lanes(N, laplacian)
        T, TP = self.T, self.TP
        self.lap = laplacian
        self.bufs = [np.zeros((B, rows, TP, width)) for rows, width in p.bufs]
        self.r, self.R, self.B = r, R, B
        out = None
        for op in p.ops:
            out = getattr(self, f'op_{op.kind}')(op) or out
        return out

    # ---- helpers: distance with lanes -------------------------------
    def _dist_lanes(self, d, recv, send):
        """rho = sqrt(eps + d.d) and d components as lane arrays [B, TP] given the electron
        indices the difference depends on (+1 for recv, -1 for send; send < 0: nucleus)."""
        B, TP, T = self.B, self.TP, self.T
        rho = np.sqrt(self.eps + (d * d).sum(-1))                   # [B]
        f_rho = np.zeros((B, TP))
        f_d = np.zeros((B, 3, TP))
        f_rho[:, 0] = rho
        f_d[:, :, 0] = d
        if self.lap:
            g = d / rho[:, None]                                    # d rho / d r_recv
            lap1 = 3 / rho - (d * d).sum(-1) / rho ** 3             # Laplacian w.r.t. one end
            n_ends = 0
            for idx, sgn in ((recv, 1.0), (send, -1.0)):
                if idx < 0:
                    continue
                n_ends += 1
                for a in range(3):
                    f_rho[:, 1 + 3 * idx + a] += sgn * g[:, a]
                    f_d[:, a, 1 + 3 * idx + a] += sgn
            if recv == send:            # self edge: d == 0 identically, no dependence on r
                f_rho[:, 1:] = 0
                f_d[:, :, 1:] = 0
                n_ends = 0
            f_rho[:, T - 1] = n_ends * lap1
        return f_rho, f_d

    def _feat4(self, d, recv, send, log_rescale):
        """[|d|, dx, dy, dz] (optionally log-rescaled) as [B, TP, 4]."""
        f_rho, f_d = self._dist_lanes(d, recv, send)
        out = np.zeros((self.B, self.TP, 4))
        if not log_rescale:
            out[:, :, 0] = f_rho
            out[:, :, 1:] = f_d.transpose(0, 2, 1)
            return out
        # s(rho) = log1p(rho)/rho; features: rho*s = log1p(rho) and d*s
        rho = f_rho[:, 0]
        s = np.log1p(rho) / rho
        s1 = (1 / (1 + rho) - s) / rho                               # s'
        s2 = (-1 / (1 + rho) ** 2 - 2 * s1) / rho                    # s''
        out[:, :, 0] = self._chain(f_rho, np.log1p(rho), 1 / (1 + rho), -1 / (1 + rho) ** 2)
        f_s = self._chain(f_rho, s, s1, s2)
        for a in range(3):
            out[:, :, 1 + a] = self._prod(f_d[:, a], f_s)
        return out

    def _chain(self, fx, y, d1, d2):
        """Elementwise y = phi(x) on lane arrays [..., TP] (lane axis last)."""
        T = self.T
        out = np.zeros_like(fx)
        out[..., 0] = y
        if self.lap:
            J = fx[..., 1:T - 1]
            out[..., 1:T - 1] = d1[..., None] * J
            out[..., T - 1] = d1 * fx[..., T - 1] + d2 * (J * J).sum(-1)
        return out

    def _prod(self, fa, fb):
        """Elementwise product on lane arrays [..., TP]."""
        T = self.T
        out = np.zeros_like(fa)
        out[..., 0] = fa[..., 0] * fb[..., 0]
        if self.lap:
            Ja, Jb = fa[..., 1:T - 1], fb[..., 1:T - 1]
            out[..., 1:T - 1] = Ja * fb[..., :1] + fa[..., :1] * Jb
            out[..., T - 1] = fa[..., T - 1] * fb[..., 0] + fa[..., 0] * fb[..., T - 1] + 2 * (Ja * Jb).sum(-1)
        return out

    # lane-last <-> buffer layout [B, rows, TP, width]
    @staticmethod
    def _ll(x):          # [..., TP, W] -> [..., W, TP]
        return np.swapaxes(x, -1, -2)

    # ---- ops -----------------------------------------------------------
    def op_1(self, op):  # FEAT_EN
        dst, logr, spin = op.i[0], op.i[1], op.i[2]
        buf = self.bufs[dst]
        n_nuc = self.R.shape[0]
        for i in range(self.N):
            for a in range(n_nuc):
                d = self.r[:, i] - self.R[a]
                buf[:, i, :, 4 * a:4 * a + 4] = self._feat4(d, i, -1, logr)
            if spin:
                buf[:, i, 0, 4 * n_nuc] = 1.0 if i < self.p.n_up else -1.0

    def op_2(self, op):  # FEAT_EE
        dst, tab, n, logr = op.i[:4]
        buf = self.bufs[dst]
        for k in range(n):
            rc, sd = self.it[tab + 2 * k], self.it[tab + 2 * k + 1]
            d = self.r[:, rc] - (self.r[:, sd] if sd >= 0 else self.R[-1 - sd])     # sd < 0: nucleus -1 - sd
            buf[:, k] = self._feat4(d, rc, sd if sd >= 0 else -1, logr)

    def op_3(self, op):  # LINEAR
        i = op.i
        npieces = i[0]
        dst, dr0, dc0, nrows, nout, woff, boff, act, res, rr0, rnorm = i[17:28]
        nout_p = P.pad4(nout)
        acc = np.zeros((self.B, nrows, self.TP, nout_p))
        o = woff
        for p_ in range(npieces):
            src, r0, K, bc = i[1 + 4 * p_:5 + 4 * p_]
            Kp = P.pad4(K)
            W = self.w[o:o + Kp * nout_p].reshape(Kp, nout_p)
            o += Kp * nout_p
            X = self.bufs[src][:, r0:r0 + (1 if bc else nrows), :, :Kp]
            acc += X @ W                                               # bcast over rows
        if boff >= 0:
            acc[:, :, 0, :] += self.w[boff:boff + nout_p]
        y = self._ll(acc)                                              # [B,rows,W,TP]
        v = y[..., 0]
        if act == 1:
            t = np.tanh(v)
            y = self._chain(y, t, 1 - t * t, -2 * t * (1 - t * t))
        elif act == 2:
            s = 1 / (1 + np.exp(-v))
            d1 = s * (1 + v * (1 - s))
            d2 = s * (1 - s) * (2 + v * (1 - 2 * s))
            y = self._chain(y, v * s, d1, d2)
        elif act == 3:        # shifted softplus (hkext.py:13-19)
            s = 1 / (1 + np.exp(-v))
            y = self._chain(y, np.logaddexp(0.0, v) - math.log(2.0), s, s * (1 - s))
        elif act == 4:        # 1 + 2 tanh(x/4) (wf/nn_wave_function.py:17)
            t = np.tanh(v / 4)
            y = self._chain(y, 1 + 2 * t, (1 - t * t) / 2, -t * (1 - t * t) / 4)
        y = self._ll(y)
        if res >= 0:
            y = (self.bufs[res][:, rr0:rr0 + nrows, :, dc0:dc0 + nout_p] + y) * (1 / math.sqrt(2.0) if rnorm else 1.0)
        self.bufs[dst][:, dr0:dr0 + nrows, :, dc0:dc0 + nout_p] = y

    def op_4(self, op):  # SPIN_MEAN
        src, dst, n_up = op.i[:3]
        x = self.bufs[src]
        self.bufs[dst][:, 0] = x[:, :n_up].mean(1)
        self.bufs[dst][:, 1] = x[:, n_up:].mean(1)

    def op_5(self, op):  # CONV
        we, hx, dst, c0, tab, S, W = op.i[:7]
        out = self.bufs[dst]
        for i in range(self.N):
            acc = np.zeros((self.B, W, self.TP))
            for s in range(S):
                row, snd = self.it[tab + 2 * (i * S + s)], self.it[tab + 2 * (i * S + s) + 1]
                if row < 0:
                    continue
                a = self._ll(self.bufs[we][:, row, :, :W])
                b_ = self._ll(self.bufs[hx][:, snd if snd >= 0 else -1 - snd, :, :W])
                acc += self._prod(a, b_)
            out[:, i, :, c0:c0 + W] = self._ll(acc)

    def op_6(self, op):  # EDGE_SUM
        eb, div, dst, c0, tab, S, W = op.i[:7]
        out = self.bufs[dst]
        for i in range(self.N):
            acc = np.zeros((self.B, self.TP, W))
            for s in range(S):
                row = self.it[tab + 2 * (i * S + s)]
                if row >= 0:
                    acc += self.bufs[eb][:, row, :, :W]
            out[:, i, :, c0:c0 + W] = acc / div

    def op_7(self, op):  # ROW_SUM
        src, dst = op.i[:2]
        self.bufs[dst][:, 0] = self.bufs[src].sum(1)

    def op_8(self, op):  # ORBITALS
        bf, dst, o_pu, o_pd, o_zu, o_zd = op.i[:6]
        n_env = max(op.i[6], 1)                       # envelopes per nucleus (table column = nuc * n_env + e)
        N, K, n_up = self.N, self.p.spec.n_determinants, self.p.n_up
        n_nuc = self.R.shape[0] * n_env
        out = self.bufs[dst]
        for i in range(N):
            o_pi, o_z = (o_pu, o_zu) if i < n_up else (o_pd, o_zd)
            pi = self.w[o_pi:o_pi + K * N * n_nuc].reshape(K * N, n_nuc)
            ze = self.w[o_z:o_z + K * N * n_nuc].reshape(K * N, n_nuc)
            env = np.zeros((self.B, K * N, self.TP))
            for a in range(n_nuc):
                f_rho, _ = self._dist_lanes(self.r[:, i] - self.R[a // n_env], i, -1)   # [B,TP]
                z = np.abs(ze[:, a])                                                 # [KN]
                ex = np.exp(-z[None] * f_rho[:, :1])                                 # [B,KN]
                f = np.broadcast_to(f_rho[:, None, :], (self.B, K * N, self.TP))
                env += pi[None, :, a, None] * self._chain(f, ex, -z[None] * ex, z[None] ** 2 * ex)
            b_ = self._ll(self.bufs[bf][:, i, :, :K * N])                            # [B,KN,TP]
            A = self._prod(env, b_).reshape(self.B, K, N, self.TP)                   # [B,K,mu,TP]
            out[:, :, :, i * N:(i + 1) * N] = A.transpose(0, 1, 3, 2)

    def op_9(self, op):  # SLOGDET
        orb = op.i[0]
        N, K, T = self.N, self.p.spec.n_determinants, self.T
        A = self.bufs[orb][..., :N * N].reshape(self.B, K, self.TP, N, N)
        A0 = A[:, :, 0]
        sign, logdet = np.linalg.slogdet(A0)
        self.sign_k = sign
        self.logdet = np.zeros((self.B, K, self.TP))
        self.logdet[:, :, 0] = logdet
        if self.lap:
            inv = np.linalg.inv(A0)                                                  # [B,K,N,N]
            M = np.einsum('bkij,bkcjl->bkcil', inv, A[:, :, 1:T])                    # incl. L lane
            tr = np.trace(M, axis1=-2, axis2=-1)                                     # [B,K,T-1]
            tr2 = np.einsum('bkcij,bkcji->bkc', M[:, :, :T - 2], M[:, :, :T - 2])
            self.logdet[:, :, 1:T - 1] = tr[:, :, :T - 2]
            self.logdet[:, :, T - 1] = tr[:, :, T - 2] - tr2.sum(-1)

    def op_10(self, op):  # FINAL
        jas, cc_off, cusp_kind, al_off = op.i[:4]
        s_same, s_anti = op.f[0], op.f[1]
        N, K, T, n_up = self.N, self.p.spec.n_determinants, self.T, self.p.n_up
        c = self.w[cc_off:cc_off + K] if cc_off >= 0 else np.ones(K)
        x = self.logdet
        shift = x[:, :, 0].max(1)
        shift = np.where(np.isinf(shift), 0.0, shift)
        pt = c[None] * self.sign_k * np.exp(x[:, :, 0] - shift[:, None])             # [B,K]
        psi = pt.sum(1)
        f = np.zeros((self.B, self.TP))
        f[:, 0] = np.log(np.abs(psi)) + shift
        sign = np.sign(psi)
        if self.lap:
            pk = pt / psi[:, None]
            Jk = x[:, :, 1:T - 1]
            J = (pk[:, :, None] * Jk).sum(1)
            f[:, 1:T - 1] = J
            f[:, T - 1] = (pk * (x[:, :, T - 1] + (Jk * Jk).sum(-1))).sum(1) - (J * J).sum(-1)
        # cusps
        if cusp_kind:
            a_same, a_anti = self.w[al_off], self.w[al_off + 1]
            for i in range(N):
                for j in range(i + 1, N):
                    same = (i < n_up) == (j < n_up)
                    sc, al = (s_same, a_same) if same else (s_anti, a_anti)
                    f_rho, _ = self._dist_lanes(self.r[:, i] - self.r[:, j], i, j)
                    rho = f_rho[:, 0]
                    if cusp_kind == 1:
                        g, g1, g2 = -sc / (al * (1 + al * rho)), sc / (1 + al * rho) ** 2, -2 * sc * al / (1 + al * rho) ** 3
                    else:
                        g, g1, g2 = -sc * al ** 2 / (al + rho), sc * al ** 2 / (al + rho) ** 2, -2 * sc * al ** 2 / (al + rho) ** 3
                    f += self._chain(f_rho, g, g1, g2)
        if jas >= 0:
            f += self.bufs[jas][:, 0, :, 0]
        self.logpsi_lanes = f
        res = {'sign': sign.astype(np.int32), 'log': f[:, 0].copy()}
        if self.lap:
            lap, grad = f[:, T - 1], f[:, 1:T - 1]
            qf2 = (grad ** 2).sum(-1)
            e_kin = -0.5 * (lap + qf2)
            dn = np.linalg.norm(self.r[:, :, None] - self.R[None, None], axis=-1)
            v_loc = -(self.charges[None, None] / dn).sum((1, 2))
            v_el = np.zeros(self.B)
            for i in range(N):
                for j in range(i + 1, N):
                    d = self.r[:, i] - self.r[:, j]
                    v_el += 1 / np.sqrt(self.eps + (d * d).sum(-1))
            e_nuc = 0.0
            for a in range(len(self.charges)):
                for b_ in range(a + 1, len(self.charges)):
                    e_nuc += self.charges[a] * self.charges[b_] / np.linalg.norm(self.R[a] - self.R[b_])
            res.update({'e_loc': e_kin + v_loc + v_el + e_nuc, 'grad': grad.copy(),
                        'stats': np.stack([v_el, e_kin, v_loc, np.zeros(self.B), lap, qf2])})
        return res

    def op_12(self, op):  # CONST rows
        dst, off = op.i[:2]
        rows, width = self.p.bufs[dst]
        self.bufs[dst][:, :, 0, :] = self.w[off:off + rows * width].reshape(rows, width)[None]

    def op_11(self, op):  # ATTENTION (forward-Laplacian softmax attention, appendix C)
        qb, kb, vb, dst, H, hd = op.i[:6]
        n_const, kc_off, vc_off = op.i[6:9]          # constant (nuclear-token) key / value rows, value lane only
        T = self.T
        sc = 1 / math.sqrt(hd)
        for h in range(H):
            sl = slice(h * hd, (h + 1) * hd)
            q, k, v = (self.bufs[b_][..., sl] for b_ in (qb, kb, vb))              # [B,N,TP,hd]
            if n_const:
                ext = []
                for off, arr in ((kc_off, k), (vc_off, v)):
                    c = np.zeros((self.B, n_const, self.TP, hd))
                    c[:, :, 0] = self.w[off:off + n_const * H * hd].reshape(n_const, H * hd)[None, :, sl]
                    ext.append(np.concatenate([c, arr], axis=1))
                k, v = ext
            S = np.einsum('bitd,bjd->bijt', q, k[:, :, 0]) * sc                     # lanes of q
            S[..., 1:] += np.einsum('bid,bjtd->bijt', q[:, :, 0], k[:, :, 1:]) * sc
            if self.lap:
                S[..., T - 1] += 2 * sc * np.einsum('bicd,bjcd->bij', q[:, :, 1:T - 1], k[:, :, 1:T - 1])
            m = S[..., 0].max(-1, keepdims=True)
            e = np.exp(S[..., 0] - m)
            Pv = e / e.sum(-1, keepdims=True)                                       # [B,i,j]
            Pl = np.zeros_like(S)
            Pl[..., 0] = Pv
            if self.lap:
                dS = S[..., 1:T - 1]
                mc = (Pv[..., None] * dS).sum(2, keepdims=True)
                dP = Pv[..., None] * (dS - mc)
                Pl[..., 1:T - 1] = dP
                LS = S[..., T - 1]
                Pl[..., T - 1] = (dP * (dS - mc)).sum(-1) + Pv * (
                    LS - (Pv * LS).sum(-1, keepdims=True) - (dP * dS).sum((2, 3))[:, :, None])
            out = np.einsum('bijt,bjd->bitd', Pl, v[:, :, 0])
            out[:, :, 1:] += np.einsum('bij,bjtd->bitd', Pv, v[:, :, 1:])
            if self.lap:
                out[:, :, T - 1] += 2 * np.einsum('bijc,bjcd->bid', Pl[..., 1:T - 1], v[:, :, 1:T - 1])
            if self.TP > T:
                out[:, :, T:] = 0
            self.bufs[dst][..., sl] = out
