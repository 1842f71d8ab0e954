"""Compile (AnsatzSpec, parameter tree) into the flat layer program of include/dqmc.h.

The reference builds its forward pass by tracing haiku modules
(wf/nn_wave_function.py:127-173 -> wf/omni.py:157-178 -> gnn/electron_gnn.py:403-432);
here the same dataflow is emitted once, on the host, as a list of `dqmc_op` records over
per-walker activation buffers plus one contiguous weight buffer and one int table
(edge orderings of gnn/graph.py:17-31,132-139).  The HIP library executes the list.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .params import attention_feature_name, GNN, OMNI, WF, layer_dims, layer_name
from .spec import AnsatzSpec, MLPSpec

OP_FEAT_EN, OP_FEAT_EE, OP_LINEAR, OP_SPIN_MEAN, OP_CONV, OP_EDGE_SUM, OP_ROW_SUM = 1, 2, 3, 4, 5, 6, 7
OP_ORBITALS, OP_SLOGDET, OP_FINAL, OP_ATTENTION, OP_CONST = 8, 9, 10, 11, 12
ACT = {None: 0, 'tanh': 1, 'silu': 2, 'ssp': 3, 'mult_tanh': 4}
N_OP_I = 28


def pad4(n: int) -> int:
    return (n + 3) // 4 * 4


class DqmcBuf(ctypes.Structure):
    _fields_ = [('rows', ctypes.c_int32), ('width', ctypes.c_int32)]


class DqmcOp(ctypes.Structure):
    _fields_ = [('kind', ctypes.c_int32), ('i', ctypes.c_int32 * N_OP_I), ('f', ctypes.c_float * 4)]


class DqmcSystem(ctypes.Structure):
    _fields_ = [('n_up', ctypes.c_int32), ('n_down', ctypes.c_int32), ('n_nuc', ctypes.c_int32),
                ('n_det', ctypes.c_int32), ('dtype', ctypes.c_int32), ('reserved', ctypes.c_int32),
                ('norm_eps', ctypes.c_double), ('e_nuc', ctypes.c_double)]


@dataclass
class Op:
    kind: int
    i: List[int]
    f: List[float] = field(default_factory=lambda: [0.0] * 4)
    note: str = ''


def edge_pairs(n_up: int, n_down: int, edge_types: Sequence[str], self_interaction: bool):
    """(recv, send) of every edge row, in `single_array` order, per edge type
    (gnn/graph.py:17-31,132-139,246-335).  Returns {type: [(recv, send), ...]}."""
    N = n_up + n_down

    def block(senders, receivers, mask_self):
        ns, nr = len(senders), len(receivers)
        out = []
        if mask_self:
            assert ns == nr
            for k in range(ns - 1):                 # row k of the [n-1, n] edge array
                for rr in range(nr):
                    out.append((receivers[rr], senders[k + (1 if rr <= k else 0)]))
        else:
            for s in range(ns):
                for rr in range(nr):
                    out.append((receivers[rr], senders[s]))
        return out

    up, dn, al = list(range(n_up)), list(range(n_up, N)), list(range(N))
    res = {}
    for t in edge_types:
        if t == 'same':
            res[t] = block(up, up, not self_interaction) + block(dn, dn, not self_interaction)
        elif t == 'anti':
            res[t] = block(dn, up, False) + block(up, dn, False)     # du then ud
        elif t == 'up':
            res[t] = block(up, al, False)
        elif t == 'down':
            res[t] = block(dn, al, False)
        else:
            raise ValueError(t)
    return res


@dataclass
class Program:
    spec: AnsatzSpec
    n_up: int
    n_down: int
    n_nuc: int
    bufs: List[Tuple[int, int]]
    ops: List[Op]
    weights: np.ndarray            # float64
    itable: np.ndarray             # int32
    buf_names: Dict[str, int]
    weight_slots: List[tuple]      # (offset, module, leaf, kind, meta) for repacking
    flops_per_walker: float        # F_lin = 2 * sum rows*in*out over Linear layers (unpadded)

    def c_bufs(self):
        arr = (DqmcBuf * len(self.bufs))()
        for k, (r, w) in enumerate(self.bufs):
            arr[k].rows, arr[k].width = r, w
        return arr

    def c_ops(self):
        arr = (DqmcOp * len(self.ops))()
        for k, op in enumerate(self.ops):
            arr[k].kind = op.kind
            ii = list(op.i) + [0] * (N_OP_I - len(op.i))
            for j in range(N_OP_I):
                arr[k].i[j] = int(ii[j])
            for j in range(4):
                arr[k].f[j] = float(op.f[j])
        return arr


class _Builder:
    def __init__(self, spec, params, n_up, n_down, n_nuc):
        self.spec, self.params = spec, params
        self.n_up, self.n_down, self.n_nuc = n_up, n_down, n_nuc
        self.N = n_up + n_down
        self.bufs: List[Tuple[int, int]] = []
        self.names: Dict[str, int] = {}
        self.ops: List[Op] = []
        self.w: List[np.ndarray] = []
        self.w_len = 0
        self.slots: List[tuple] = []
        self.it: List[int] = []
        self.flops = 0.0

    # -- allocation helpers --
    def buf(self, name, rows, width):
        self.bufs.append((max(rows, 1), pad4(width)))     # an empty edge segment (e.g. one electron per spin) keeps a dummy row
        self.names[name] = len(self.bufs) - 1
        return len(self.bufs) - 1

    def push_w(self, arr, slot):
        off = self.w_len
        a = np.asarray(arr, np.float64).reshape(-1)
        n = pad4(a.size)
        buf = np.zeros(n)
        buf[:a.size] = a
        self.w.append(buf)
        self.w_len += n
        self.slots.append((off,) + slot)
        return off

    def push_table(self, ints):
        off = len(self.it)
        self.it += [int(x) for x in ints]
        return off

    def pack_linear(self, module, piece_K: Sequence[int]):
        """W [sum pad4(K_p)][pad4(Nout)] with the rows of each concat piece padded."""
        p = self.params[module]
        W = np.asarray(p['w'], np.float64)
        assert W.shape[0] == sum(piece_K), (module, W.shape, piece_K)
        nout = W.shape[1]
        blocks, o = [], 0
        for K in piece_K:
            blk = np.zeros((pad4(K), pad4(nout)))
            blk[:K, :nout] = W[o:o + K]
            blocks.append(blk)
            o += K
        w_off = self.push_w(np.concatenate(blocks, 0), (module, 'w', 'linear_w', tuple(piece_K)))
        b_off = -1
        if 'b' in p:
            b = np.zeros(pad4(nout))
            b[:nout] = np.asarray(p['b'], np.float64)
            b_off = self.push_w(b, (module, 'b', 'linear_b', (nout,)))
        return w_off, b_off, nout

    def linear(self, module, pieces, dst, dst_r0, dst_col0, nrows, act, res=-1, res_r0=0, res_scale=1.0, note=''):
        """pieces: [(src buf, r0, K, bcast)]."""
        assert 1 <= len(pieces) <= 4
        w_off, b_off, nout = self.pack_linear(module, [p[2] for p in pieces])
        i = [len(pieces)]
        for p in pieces:
            i += list(p)
        i += [0] * (17 - len(i))
        i += [dst, dst_r0, dst_col0, nrows, nout, w_off, b_off, ACT[act], res, res_r0, int(res_scale != 1.0)]
        if nrows == 0:          # empty segment: the weights stay in the buffer (parameter order), no op
            return nout
        self.ops.append(Op(OP_LINEAR, i, [0, 0, 0, 0], note or module))
        self.flops += 2.0 * nrows * sum(p[2] for p in pieces) * nout
        return nout

    def mlp(self, prefix, mspec: MLPSpec, pieces, in_dim, out_dim, dst, dst_r0, dst_col0, nrows, tmp_rows,
            res=-1, res_r0=0, res_scale=1.0):
        """hkext.MLP as a chain of LINEAR ops; hidden activations go to fresh buffers with
        `tmp_rows` rows per walker (row offsets preserved).  The residual applies to the
        last layer only (it wraps the whole MLP, hkext.py:130-137)."""
        for step in self.mlp_steps(prefix, mspec, pieces, in_dim, out_dim, dst, dst_r0, dst_col0, nrows, tmp_rows,
                                   res, res_r0, res_scale):
            step()

    def mlp_steps(self, prefix, mspec: MLPSpec, pieces, in_dim, out_dim, dst, dst_r0, dst_col0, nrows, tmp_rows,
                  res=-1, res_r0=0, res_scale=1.0):
        """The LINEAR ops of one MLP as a list of thunks, so that independent MLPs can be emitted
        in lockstep (ops of equal depth adjacent => one dependency level in the fused kernel).
        Hidden buffers hold just the `nrows` rows of the segment (`tmp_rows` is unused)."""
        dims = mspec.dims(in_dim, out_dim)
        state = {'cur': pieces}
        steps = []
        for k, dim in enumerate(dims):
            last = k == len(dims) - 1
            act = mspec.layer_act(k, len(dims))

            def step(k=k, dim=dim, last=last, act=act):
                if last:
                    self.linear(f'{prefix}/linear_{k}', state['cur'], dst, dst_r0, dst_col0, nrows, act, res, res_r0, res_scale)
                else:
                    hb = self.buf(f'{prefix}/hidden_{k}', nrows, dim)
                    self.linear(f'{prefix}/linear_{k}', state['cur'], hb, 0, 0, nrows, act)
                    state['cur'] = [(hb, 0, dim, 0)]
            steps.append(step)
        return steps

    @staticmethod
    def lockstep(*chains):
        for k in range(max(len(c) for c in chains)):
            for c in chains:
                if k < len(c):
                    c[k]()


def compile_program(spec: AnsatzSpec, params, n_up: int, n_down: int, n_nuc: int, R=None, eps=None) -> Program:
    """`R` [n_nuc,3] and the safe-norm `eps` are needed only by ansatzes with nuclear tokens, whose
    electron-independent nuclear stream is folded into constants here (nuclear_stream.py)."""
    b = _Builder(spec, params, n_up, n_down, n_nuc)
    nuc_kv, nuc_zetas = None, None
    if spec.nuclei_tokens:
        if R is None or eps is None:
            raise ValueError(f'{spec.name}: the nuclear geometry R and eps are needed to compile nuclear tokens')
        from .nuclear_stream import fold
        nuc_kv, nuc_zetas = fold(params, spec, R, eps)
    N, K, D, E = n_up + n_down, spec.n_determinants, spec.embedding_dim, spec.two_particle_dim
    S2 = 1.0 / np.sqrt(2.0)

    # ---- input features ----
    d0, rows = layer_dims(spec, n_nuc)
    x = b.buf('x_feat', N, d0)
    b.ops.append(Op(OP_FEAT_EN, [x, int(spec.emb_log_rescale), int(spec.emb_use_spin)], note='electron-nucleus features'))
    x_dim = d0
    if spec.emb_project:
        x1 = b.buf('x0', N, D)
        b.linear(f'{GNN}/~/electron_embedding/linear', [(x, 0, d0, 0)], x1, 0, 0, N, None)
        x, x_dim = x1, D
    else:
        b.names['x0'] = x

    pairs = edge_pairs(n_up, n_down, spec.edge_types, spec.self_interaction)
    seg, n_edge_rows = {}, 0                      # edge type -> (row offset, n rows)
    for t in spec.edge_types:
        seg[t] = (n_edge_rows, len(pairs[t]))
        n_edge_rows += len(pairs[t])
    e, e_dim = -1, 0
    conv_tab: Dict[str, Tuple[int, int]] = {}
    if spec.edge_types:
        flat = [v for t in spec.edge_types for pr in pairs[t] for v in pr]
        tab = b.push_table(flat)
        e = b.buf('e0', n_edge_rows, 4)
        e_dim = 4
        b.ops.append(Op(OP_FEAT_EE, [e, tab, n_edge_rows, int(spec.edge_log_rescale)], note='electron-electron edge features'))
        for t in spec.edge_types:                 # per-receiver (edge row, sender) lists
            per = [[] for _ in range(N)]
            for k, (rc, sd) in enumerate(pairs[t]):
                per[rc].append((seg[t][0] + k, sd))
            S = max(len(p) for p in per)
            flat = []
            for p in per:
                p = p + [(-1, -1)] * (S - len(p))
                flat += [v for pr in p for v in pr]
            conv_tab[t] = (b.push_table(flat), S)

    # ---- interaction layers ----
    for l, row in enumerate(rows):
        ln = layer_name(l)
        if spec.layer_kind == 'attention':
            uf = f'{ln}/~/{attention_feature_name(spec)}'
            H = spec.num_heads
            hd = x_dim // H
            q, k_, v = (b.buf(f'l{l}/{nm}', N, H * hd) for nm in ('q', 'k', 'v'))
            for nm, dst in (('query', q), ('key', k_), ('value', v)):
                b.linear(f'{uf}/multi_head_attention/{nm}', [(x, 0, x_dim, 0)], dst, 0, 0, N, None)
            att = b.buf(f'l{l}/att', N, H * hd)
            n_const, kc_off, vc_off = 0, 0, 0
            if nuc_kv is not None:       # nuclear tokens: constant extra key / value rows (no derivative lanes)
                n_const = n_nuc
                kc_off = b.push_w(nuc_kv[l][0], ('', '', 'const', ()))
                vc_off = b.push_w(nuc_kv[l][1], ('', '', 'const', ()))
            b.ops.append(Op(OP_ATTENTION, [q, k_, v, att, H, hd, n_const, kc_off, vc_off], note=f'layer {l} attention'))
            # attention FLOPs (QK^T and PV), algorithmic: 2 * 2 * N*(N + n_const)*D per walker
            b.flops += 4.0 * N * (N + n_const) * H * hd
            a2 = b.buf(f'l{l}/attended', N, x_dim)
            b.linear(f'{uf}/multi_head_attention/linear', [(att, 0, H * hd, 0)], a2, 0, 0, N, None, res=x, res_scale=1.0)
            xn = b.buf(f'x{l + 1}', N, x_dim)
            b.mlp(f'{uf}/mlp', spec.attn_mlp, [(a2, 0, x_dim, 0)], x_dim, x_dim, xn, 0, 0, N, N, res=a2, res_scale=1.0)
            x = xn
            continue
        pieces = []
        mean = -1
        cbuf, c_col = -1, 0
        n_conv = sum(1 for u in spec.update_features if u.startswith('conv_') or u.startswith('edge_'))
        c_width = sum(E if u.startswith('conv_') else e_dim for u in spec.update_features
                      if u.startswith('conv_') or u.startswith('edge_'))
        if n_conv:
            cbuf = b.buf(f'l{l}/agg', N, c_width)
        for uf in spec.update_features:
            if uf == 'residual':
                pieces.append((x, 0, x_dim, 0))
            elif uf in ('node_up', 'node_down'):
                if mean < 0:
                    mean = b.buf(f'l{l}/mean', 2, x_dim)     # the SPIN_MEAN op is emitted right before g
                pieces.append((mean, 0 if uf == 'node_up' else 1, x_dim, 1))
            elif uf.startswith('conv_'):
                t = uf[5:]
                base = f'{ln}/~/convolution_electron_update_feature/~single_edge_type_update'
                r0, nr = seg[t]
                we = b.names.get(f'l{l}/we')
                if we is None:
                    we = b.buf(f'l{l}/we', n_edge_rows, E)
                hx = b.buf(f'l{l}/hx_{t}', N, E)
                b.lockstep(
                    b.mlp_steps(f'{base}/w_{t}', spec.w, [(e, r0, e_dim, 0)], e_dim, E, we, r0, 0, nr, n_edge_rows),
                    b.mlp_steps(f'{base}/h_{t}', spec.h, [(x, 0, x_dim, 0)], x_dim, E, hx, 0, 0, N, N))
                tab, S = conv_tab[t]
                b.ops.append(Op(OP_CONV, [we, hx, cbuf, c_col, tab, S, E], note=f'layer {l} conv_{t}'))
                c_col += E
            elif uf.startswith('edge_'):
                t = uf[5:]
                tab, S = conv_tab[t]
                n_send = n_up if t == 'up' else n_down
                b.ops.append(Op(OP_EDGE_SUM, [e, max(n_send, 1), cbuf, c_col, tab, S, e_dim], [0, 0, 0, 0],
                                note=f'layer {l} edge_{t} mean'))
                c_col += e_dim
            else:
                raise ValueError(uf)
        if cbuf >= 0:
            pieces.append((cbuf, 0, c_width, 0))
        assert sum(p[2] for p in pieces) == row['cat']
        if mean >= 0:
            b.ops.append(Op(OP_SPIN_MEAN, [x, mean, n_up], note=f'layer {l} spin means'))
        xn = b.buf(f'x{l + 1}', N, D)
        resid = spec.electron_residual_normalize is not None and x_dim == D
        b.mlp(f'{ln}/~/g', spec.g, pieces, row['cat'], D, xn, 0, 0, N, N,
              res=x if resid else -1, res_scale=(S2 if spec.electron_residual_normalize else 1.0) if resid else 1.0)
        if spec.deep_features and not row['last']:
            en = b.buf(f'e{l + 1}', n_edge_rows, E)
            resid_e = spec.two_particle_residual_normalize is not None and e_dim == E
            b.mlp(f'{ln}/~/u', spec.u, [(e, 0, e_dim, 0)], e_dim, E, en, 0, 0, n_edge_rows, n_edge_rows,
                  res=e if resid_e else -1,
                  res_scale=(S2 if spec.two_particle_residual_normalize else 1.0) if resid_e else 1.0)
            e, e_dim = en, E
        x, x_dim = xn, D

    # ---- heads ----
    jas = -1
    if spec.jastrow is not None:
        xs = b.buf('x_sum', 1, x_dim)
        b.ops.append(Op(OP_ROW_SUM, [x, xs], note='Jastrow: sum over electrons'))
        jas = b.buf('jastrow', 1, 1)
        b.mlp(f'{OMNI}/~/Jastrow/~/mlp', spec.jastrow, [(xs, 0, x_dim, 0)], x_dim, 1, jas, 0, 0, 1, 1)
    assert spec.full_determinant, 'only full determinants are compiled'
    bf = b.buf('backflow', N, K * N)
    b.mlp(f'{OMNI}/~/Backflow/~/mlp', spec.backflow, [(x, 0, x_dim, 0)], x_dim, K * N, bf, 0, 0, n_up, N)
    b.mlp(f'{OMNI}/~/Backflow_1/~/mlp', spec.backflow, [(x, n_up, x_dim, 0)], x_dim, K * N, bf, n_up, 0, n_down, N)
    n_env = 1
    if spec.envelope == 'simplified':
        # wf/env.py:110-226 with pi = 1 and orbital-independent exponents, expanded to the general
        # [K*N orbitals][n_nuc * n_env] table the ORBITALS kernel reads: zeta[(k,mu)][(nuc,e)] = zetas[nuc,k,e]
        n_env = spec.n_envelope_per_nucleus
        offs = [b.push_w(np.ones((K * N, n_nuc * n_env)), ('', '', 'const', ())) for _ in range(2)]
        for spin in ('up', 'down'):
            z = np.transpose(nuc_zetas[spin], (1, 0, 2)).reshape(K, 1, n_nuc * n_env)
            offs.append(b.push_w(np.broadcast_to(z, (K, N, n_nuc * n_env)), ('', '', 'const', ())))
    else:
        env = params[f'{WF}/~/exponential_envelopes']
        offs = [b.push_w(env[nm], (f'{WF}/~/exponential_envelopes', nm, 'raw', ())) for nm in
                ('pi_up', 'pi_down', 'zetas_up', 'zetas_down')]
    orb = b.buf('orbitals', K, N * N)
    b.ops.append(Op(OP_ORBITALS, [bf, orb] + offs + [n_env], note='Slater matrices = envelope * backflow'))
    b.ops.append(Op(OP_SLOGDET, [orb], note='slogdet + derivative traces'))
    cc_off = -1
    if spec.conf_coeff == 'linear':
        cc_off = b.push_w(params[f'{WF}/~/conf_coeff']['w'], (f'{WF}/~/conf_coeff', 'w', 'raw', ()))
    cusp_kind = {None: 0, 'deepqmc': 1, 'psiformer': 2}[spec.cusp]
    if spec.cusp is not None and spec.cusp_trainable_alpha:
        cm = params[f'{WF}/~/electronic_cusp_asymptotic']
        al_off = b.push_w(np.array([float(cm['same_alpha']), float(cm['anti_alpha'])]),
                          (f'{WF}/~/electronic_cusp_asymptotic', ('same_alpha', 'anti_alpha'), 'alphas', ()))
    else:
        al_off = b.push_w(np.array([spec.cusp_alpha, spec.cusp_alpha]), ('', '', 'const', ()))
    b.ops.append(Op(OP_FINAL, [jas, cc_off, cusp_kind, al_off], [spec.cusp_same_scale, spec.cusp_anti_scale, 0, 0],
                    note='CI sum + cusp + Jastrow (+ potentials, E_loc)'))
    return Program(spec, n_up, n_down, n_nuc, b.bufs, b.ops, np.concatenate(b.w), np.asarray(b.it, np.int32),
                   b.names, b.slots, b.flops)
