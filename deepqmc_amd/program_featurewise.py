"""Layer program of the reference's *featurewise-convolution* ansatz family -- the configuration of the reference's
own test suite (reference tests/conf/ansatz.yaml), whose psi / Laplacian / local-energy / sampler goldens
(tests/test_wf/*.npz, tests/test_hamil/test_local_energy_Molecular_.npz, tests/test_sampling/*.npz) are thereby
comparable DIRECTLY with the HIP path.

What distinguishes it from conf/ansatz/default.yaml (compile_program):
  * electron and nuclear embeddings are learned constants (hk.Embed; `positional_embeddings: false`,
    gnn/electron_gnn.py:497-503,596-625) -> CONST ops;
  * `update_rule: featurewise` (gnn/electron_gnn.py:242-259): one convolution per edge type same / anti / ne -- the
    'ne' senders are the constant nuclear embeddings -- each through its own g MLP, summed onto the residual
    (`normalize: false`);
  * shifted-softplus MLPs (hkext.py:13-19) for Jastrow and backflow; the default multiplicative backflow activation
    1 + 2 tanh(x/4) (wf/nn_wave_function.py:14-33) on the backflow output;
  * spin-restricted per-shell envelopes with shared exponents (wf/env.py:27-47,57-75);
  * `full_determinant: false` (wf/nn_wave_function.py:136-156): psi = sum_k c_k det(A_k^up) det(A_k^down).  The two
    spin blocks are laid out as ONE block-diagonal N x N matrix per determinant (zero envelope weights off the
    blocks), whose pivoted LU visits the blocks one after the other: sign and log|det| are the product / sum of the
    blocks', every derivative trace is block-wise too.
Parameters come as the haiku tree of the reference (module names as in tests/test_wf/test_grad_psi.npz).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Sequence

import numpy as np

from .params import GNN, OMNI, WF, layer_name
from .program import (OP_CONST, OP_CONV, OP_FEAT_EE, OP_FINAL, OP_ORBITALS, OP_ROW_SUM, OP_SLOGDET, Op, Program, _Builder, edge_pairs)
from .spec import AnsatzSpec, MLPSpec


@dataclass(frozen=True)
class FeaturewiseSpec:
    """tests/conf/ansatz.yaml (keys cited per field)."""
    n_determinants: int = 2                      # n_determinants
    embedding_dim: int = 8                       # omni_factory.embedding_dim
    two_particle_dim: int = 8                    # gnn_factory.two_particle_stream_dim
    edge_types: Sequence[str] = ('same', 'anti', 'ne')
    w: MLPSpec = MLPSpec(('log', 1), False, False, 'tanh')       # update_features[0].w_factory
    h: MLPSpec = MLPSpec(('log', 1), True, False, 'tanh')        # update_features[0].h_factory
    g: MLPSpec = MLPSpec(('log', 1), True, False, 'tanh')        # layer_factory.subnet_factory
    jastrow: MLPSpec = MLPSpec(('log', 3), 'not_last', True, 'ssp')
    backflow: MLPSpec = MLPSpec(('log', 3), True, True, 'ssp')
    cusp_same_scale: float = 0.25
    cusp_anti_scale: float = 0.5
    cusp_alpha: float = 10.0
    name: str = 'featurewise'

    def as_ansatz_spec(self) -> AnsatzSpec:
        return AnsatzSpec(name=self.name, n_determinants=self.n_determinants, full_determinant=False,
                          embedding_dim=self.embedding_dim, two_particle_dim=self.two_particle_dim, n_interactions=1)


def compile_featurewise(fs: FeaturewiseSpec, params, n_up: int, n_down: int, n_nuc: int, mol_shells: Sequence[int]) -> Program:
    """mol_shells: occupied shells per nucleus (MolecularHamiltonian.mol_shells, hamil.py:32-41,147)."""
    N, K, D, E = n_up + n_down, fs.n_determinants, fs.embedding_dim, fs.two_particle_dim
    P = {m: dict(v) for m, v in params.items()}
    b = _Builder(fs.as_ansatz_spec(), P, n_up, n_down, n_nuc)
    ln = layer_name(0)
    conv = f'{ln}/~/convolution_electron_update_feature/~single_edge_type_update'

    # ---- constant embeddings (gnn/electron_gnn.py:497-503, 596-625) ----
    el = np.asarray(P[f'{GNN}/~/electron_embedding/ElectronicEmbedding']['embeddings'], np.float64).reshape(-1, D)
    nuc = np.asarray(P[f'{GNN}/~/nuclei_embedding/~/embed']['embeddings'], np.float64).reshape(n_nuc, D)
    x0 = b.buf('x0', N, D)
    t_el = np.zeros((N, b.bufs[x0][1]))
    t_el[:, :D] = el[0]
    b.ops.append(Op(OP_CONST, [x0, b.push_w(t_el, (f'{GNN}/~/electron_embedding/ElectronicEmbedding', 'embeddings', 'const', ()))],
                    note='electron embedding (constant)'))
    xn = b.buf('nuc', n_nuc, D)
    b.ops.append(Op(OP_CONST, [xn, b.push_w(np.pad(nuc, ((0, 0), (0, b.bufs[xn][1] - D))),
                                            (f'{GNN}/~/nuclei_embedding/~/embed', 'embeddings', 'const', ()))],
                    note='nuclear embedding (constant)'))

    # ---- edges: same / anti (electron senders), ne (nuclear senders, row = nucleus * N + electron) ----
    pairs = edge_pairs(n_up, n_down, ('same', 'anti'), False)
    pairs['ne'] = [(i, -1 - a) for a in range(n_nuc) for i in range(N)]
    seg, n_rows = {}, 0
    for t in fs.edge_types:
        seg[t] = (n_rows, len(pairs[t]))
        n_rows += len(pairs[t])
    tab = b.push_table([v for t in fs.edge_types for pr in pairs[t] for v in pr])
    e = b.buf('e0', n_rows, 4)
    b.ops.append(Op(OP_FEAT_EE, [e, tab, n_rows, 0], note='edge features (same, anti, ne)'))
    we = b.buf('l0/we', n_rows, E)
    x, res = x0, x0
    for k_t, t in enumerate(fs.edge_types):
        r0, nr = seg[t]
        per = [[] for _ in range(N)]
        for k, (rc, sd) in enumerate(pairs[t]):
            per[rc].append((r0 + k, sd))
        S = max(len(p_) for p_ in per)
        flat = [v for p_ in per for pr in (p_ + [(-1, -1)] * (S - len(p_))) for v in pr]
        ctab = b.push_table(flat)
        b.mlp(f'{conv}/w_{t}', fs.w, [(e, r0, 4, 0)], 4, E, we, r0, 0, nr, n_rows)
        send, n_send = (xn, n_nuc) if t == 'ne' else (x0, N)
        hx = b.buf(f'l0/hx_{t}', n_send, E)
        b.mlp(f'{conv}/h_{t}', fs.h, [(send, 0, D, 0)], D, E, hx, 0, 0, n_send, n_send)
        agg = b.buf(f'l0/agg_{t}', N, E)
        b.ops.append(Op(OP_CONV, [we, hx, agg, 0, ctab, S, E], note=f'conv_{t}'))
        # featurewise update: x = x0 + sum_t g_t(conv_t)  (residual, normalize: false), accumulated through the residual input
        out = b.buf('x1' if k_t == len(fs.edge_types) - 1 else f'l0/acc_{t}', N, D)
        b.mlp(f'{ln}/~/g_conv_{t}', fs.g, [(agg, 0, E, 0)], E, D, out, 0, 0, N, N, res=res, res_scale=1.0)
        res = out
    x = res

    # ---- heads ----
    xs = b.buf('x_sum', 1, D)
    b.ops.append(Op(OP_ROW_SUM, [x, xs], note='Jastrow: sum over electrons'))
    jas = b.buf('jastrow', 1, 1)
    b.mlp(f'{OMNI}/~/Jastrow/~/mlp', fs.jastrow, [(xs, 0, D, 0)], D, 1, jas, 0, 0, 1, 1)
    bf = b.buf('backflow', N, K * N)
    for spin, mod, r0, n_s, col0 in (('up', f'{OMNI}/~/Backflow/~/mlp', 0, n_up, 0), ('down', f'{OMNI}/~/Backflow_1/~/mlp', n_up, n_down, n_up)):
        dims = fs.backflow.dims(D, K * n_s)
        last = f'{mod}/linear_{len(dims) - 1}'
        W, bias = np.asarray(P[last]['w'], np.float64), np.asarray(P[last]['b'], np.float64)
        Wf, bf_ = np.zeros((W.shape[0], K * N)), np.zeros(K * N)
        for k in range(K):                                   # orbital mu' of this spin -> column k*N + col0 + mu'
            Wf[:, k * N + col0:k * N + col0 + n_s] = W[:, k * n_s:(k + 1) * n_s]
            bf_[k * N + col0:k * N + col0 + n_s] = bias[k * n_s:(k + 1) * n_s]
        P[last] = {'w': Wf, 'b': bf_}
        cur, d_in = [(x, r0, D, 0)], D
        for k, dim in enumerate(dims):
            if k < len(dims) - 1:
                hb = b.buf(f'{mod}/hidden_{k}', n_s, dim)
                b.linear(f'{mod}/linear_{k}', cur, hb, 0, 0, n_s, fs.backflow.activation)
                cur = [(hb, 0, dim, 0)]
            else:                                            # last layer + BackflowOp's 1 + 2 tanh(x/4)
                b.linear(last, cur, bf, r0, 0, n_s, 'mult_tanh')
    # ---- envelopes: one exponential per occupied shell, centred on its nucleus; spin restricted, shared exponents ----
    n_env = int(max(mol_shells))
    env = P[f'{WF}/~/exponential_envelopes']
    pi, zetas = np.asarray(env['pi'], np.float64), np.asarray(env['zetas'], np.float64)
    shell_col = [a * n_env + s for a in range(n_nuc) for s in range(int(mol_shells[a]))]      # table column of shell j
    assert pi.shape == (K * N, len(shell_col)) and zetas.shape == (len(shell_col),)
    L = n_nuc * n_env
    tabs = {}
    for spin, lo, hi in (('up', 0, n_up), ('down', n_up, N)):
        t_pi, t_ze = np.zeros((K * N, L)), np.ones((K * N, L))
        for k in range(K):
            for mu in range(lo, hi):
                t_pi[k * N + mu, shell_col] = pi[k * N + mu]
        t_ze[:, shell_col] = zetas[None]
        tabs[spin] = (t_pi, t_ze)
    offs = [b.push_w(tabs['up'][0], ('', '', 'const', ())), b.push_w(tabs['down'][0], ('', '', 'const', ())),
            b.push_w(tabs['up'][1], ('', '', 'const', ())), b.push_w(tabs['down'][1], ('', '', 'const', ()))]
    orb = b.buf('orbitals', K, N * N)
    b.ops.append(Op(OP_ORBITALS, [bf, orb] + offs + [n_env], note='block-diagonal Slater matrices = envelope * backflow'))
    b.ops.append(Op(OP_SLOGDET, [orb], note='slogdet + derivative traces'))
    cc_off = b.push_w(P[f'{WF}/~/conf_coeff']['w'], (f'{WF}/~/conf_coeff', 'w', 'raw', ()))
    al_off = b.push_w(np.array([fs.cusp_alpha, fs.cusp_alpha]), ('', '', 'const', ()))
    b.ops.append(Op(OP_FINAL, [jas, cc_off, 1, al_off], [fs.cusp_same_scale, fs.cusp_anti_scale, 0, 0], note='CI sum + cusp + Jastrow'))
    return Program(fs.as_ansatz_spec(), n_up, n_down, n_nuc, b.bufs, b.ops, np.concatenate(b.w), np.asarray(b.it, np.int32),
                   b.names, b.slots, b.flops)
