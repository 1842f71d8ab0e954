"""Parameter tree of the wave function: names, shapes, initialisation.

The reference keeps parameters in a haiku tree `{module_path: {'w'|'b'|...: array}}`
(key list visible in reference tests/test_wf/test_grad_psi.npz).  This
module rebuilds that tree for an `AnsatzSpec` so that a parameter dict produced by the
reference (e.g. unpickled from a checkpoint) can be fed to the HIP engine unchanged.

Initial values follow each YAML's init family (reference src/deepqmc/hkext.py:68-81):
  'default'  -> VarianceScaling(1, fan_in, truncated_normal), zero biases
  'ferminet' -> VarianceScaling(1, fan_in, normal) weights, VarianceScaling(1, fan_out, normal) biases
drawn from NumPy `default_rng(seed)` (the JAX PRNG stream itself is not reproduced here).
Envelope pi/zeta start at one (`init_to_ones: true`, wf/env.py:77-92).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np

from .spec import AnsatzSpec, MLPSpec

WF = 'neural_network_wave_function'
OMNI = f'{WF}/~/omni_net'
GNN = f'{OMNI}/~/electron_gnn'
_TRUNC_STD = 0.87962566103423978  # std of a standard normal truncated to [-2, 2]


def layer_name(i: int) -> str:
    return f'{GNN}/~/electron_gnn_layer' + (f'_{i}' if i else '')


def mlp_entries(prefix: str, spec: MLPSpec, in_dim: int, out_dim: int) -> List[Tuple[str, str, tuple, str]]:
    """[(module, leaf, shape, init)] of an hkext.MLP (hkext.py:83-113)."""
    dims = spec.dims(in_dim, out_dim)
    out = []
    d = in_dim
    for i, dim in enumerate(dims):
        mod = f'{prefix}/linear_{i}'
        out.append((mod, 'w', (d, dim), f'{spec.init}_w'))
        if spec.layer_bias(i, len(dims)):
            out.append((mod, 'b', (dim,), f'{spec.init}_b'))
        d = dim
    return out


def layer_dims(spec: AnsatzSpec, n_nuc: int):
    """Per-layer (node_in, edge_in, concat_dim, node_out, edge_out) widths."""
    d0 = 4 * n_nuc + (1 if spec.emb_use_spin else 0)
    d = spec.embedding_dim if spec.emb_project else d0
    e = 4 if spec.edge_types else 0
    E, D = spec.two_particle_dim, spec.embedding_dim
    rows = []
    for l in range(spec.n_interactions):
        last = l == spec.n_interactions - 1
        cat = 0
        for uf in spec.update_features:
            if uf in ('residual', 'node_up', 'node_down'):
                cat += d
            elif uf.startswith('conv_'):
                cat += E
            elif uf.startswith('edge_'):
                cat += e
            elif uf == 'attention':
                cat += d
        e_out = E if (spec.deep_features and not last) else e
        rows.append(dict(d_in=d, e_in=e, cat=cat, d_out=D, e_out=e_out, last=last))
        d, e = D, e_out
    return d0, rows


NUC_EMB = f'{GNN}/~/nuclei_embedding'
NUC_HEAD = f'{OMNI}/~/nuclear_gnn_head'
NUC_EDGE_MLP = MLPSpec((32,), True, True, 'silu', 'ferminet')       # electron_gnn.py:474-482


def nuc_embed_mlp(D: int) -> MLPSpec:
    return MLPSpec((D,), True, True, 'silu', 'ferminet')            # electron_gnn.py:483-491


def attention_feature_name(spec: AnsatzSpec) -> str:
    return 'combined_node_attention_update_feature' if spec.nuclei_tokens else 'node_attention_electron_update_feature'


def param_entries(spec: AnsatzSpec, n_up: int, n_down: int, n_nuc: int):
    """Ordered [(module, leaf, shape, init)] for the whole ansatz."""
    N, K, D, E = n_up + n_down, spec.n_determinants, spec.embedding_dim, spec.two_particle_dim
    ent: List[Tuple[str, str, tuple, str]] = []
    n_env = n_nuc  # one shell per nucleus: per_shell false (wf/env.py:27-32)
    env = f'{WF}/~/exponential_envelopes'
    if spec.envelope == 'exponential':
        for leaf in ('pi_up', 'pi_down', 'zetas_up', 'zetas_down'):
            ent.append((env, leaf, (K * N, n_env), 'ones'))
    if spec.conf_coeff == 'linear':
        ent.append((f'{WF}/~/conf_coeff', 'w', (K, 1), 'ones'))
    if spec.cusp is not None and spec.cusp_trainable_alpha:
        cm = f'{WF}/~/electronic_cusp_asymptotic'
        ent.append((cm, 'same_alpha', (), 'alpha'))
        ent.append((cm, 'anti_alpha', (), 'alpha'))
    d0, rows = layer_dims(spec, n_nuc)
    if spec.nuclei_tokens:        # nuclei embedding: nn edge features + one-hot atom type -> edge_mlp -> sum -> embed_mlp
        ent += mlp_entries(f'{NUC_EMB}/edge_mlp', NUC_EDGE_MLP, 4 + n_nuc, 32)
        ent += mlp_entries(f'{NUC_EMB}/embed_mlp', nuc_embed_mlp(D), 32, D)
    if spec.emb_project:
        ent.append((f'{GNN}/~/electron_embedding/linear', 'w', (d0, D), 'hk_linear_w'))
    for l, row in enumerate(rows):
        ln = layer_name(l)
        if spec.layer_kind == 'attention':
            uf = f'{ln}/~/{attention_feature_name(spec)}'
            hd = row['d_in'] // spec.num_heads
            for nm in ('query', 'key', 'value'):
                ent.append((f'{uf}/multi_head_attention/{nm}', 'w', (row['d_in'], spec.num_heads * hd), 'ferminet_w'))
            ent.append((f'{uf}/multi_head_attention/linear', 'w', (spec.num_heads * hd, row['d_in']), 'ferminet_w'))
            ent += mlp_entries(f'{uf}/mlp', spec.attn_mlp, row['d_in'], row['d_in'])
            continue
        for uf in spec.update_features:
            if uf.startswith('conv_'):
                typ = uf[5:]
                base = f'{ln}/~/convolution_electron_update_feature/~single_edge_type_update'
                ent += mlp_entries(f'{base}/w_{typ}', spec.w, row['e_in'], E)
                ent += mlp_entries(f'{base}/h_{typ}', spec.h, row['d_in'], E)
        ent += mlp_entries(f'{ln}/~/g', spec.g, row['cat'], D)
        if spec.deep_features and not row['last']:
            ent += mlp_entries(f'{ln}/~/u', spec.u, row['e_in'], E)
    if spec.envelope == 'simplified':   # NuclearGNNHead (omni.py:181-211): one GLU readout per spin + bias (init 2)
        n_z = K * spec.n_envelope_per_nucleus
        for glu in ('zetas_readout_glu', 'zetas_readout_glu_1'):
            for lin in ('W', 'V'):
                ent.append((f'{NUC_HEAD}/{glu}/{lin}', 'w', (D, n_z), 'hk_linear_w'))
                ent.append((f'{NUC_HEAD}/{glu}/{lin}', 'b', (n_z,), 'default_b'))
        for spin in ('up', 'down'):
            ent.append((NUC_HEAD, f'zetas_bias_{spin}', (n_nuc, K, spec.n_envelope_per_nucleus), 'twos'))
    if spec.jastrow is not None:
        ent += mlp_entries(f'{OMNI}/~/Jastrow/~/mlp', spec.jastrow, D, 1)
    n_orb_up, n_orb_down = (N, N) if spec.full_determinant else (n_up, n_down)
    ent += mlp_entries(f'{OMNI}/~/Backflow/~/mlp', spec.backflow, D, n_orb_up * K)
    ent += mlp_entries(f'{OMNI}/~/Backflow_1/~/mlp', spec.backflow, D, n_orb_down * K)
    return ent


def _draw(rng: np.random.Generator, shape, init: str, spec: AnsatzSpec):
    if init == 'ones':
        return np.ones(shape)
    if init == 'twos':
        return 2.0 * np.ones(shape)
    if init == 'alpha':
        return np.asarray(spec.cusp_alpha, np.float64)
    if init in ('default_b',):
        return np.zeros(shape)
    fan_in = shape[0] if len(shape) >= 1 else 1
    if init == 'default_w':
        std = np.sqrt(1.0 / fan_in) / _TRUNC_STD
        x = rng.standard_normal(shape)
        bad = np.abs(x) > 2
        while bad.any():  # rejection sampling of the truncated normal
            x[bad] = rng.standard_normal(int(bad.sum()))
            bad = np.abs(x) > 2
        return x * std
    if init == 'hk_linear_w':  # hk.Linear default: TruncatedNormal(1/sqrt(fan_in))
        x = np.clip(rng.standard_normal(shape), -2, 2)
        return x / np.sqrt(fan_in)
    if init == 'ferminet_w':
        return rng.standard_normal(shape) * np.sqrt(1.0 / fan_in)
    if init == 'ferminet_b':  # VarianceScaling(fan_out) on a 1-D shape: fan = shape[0]
        return rng.standard_normal(shape) * np.sqrt(1.0 / shape[0])
    raise ValueError(init)


def init_params(spec: AnsatzSpec, n_up: int, n_down: int, n_nuc: int, seed: int = 0,
                perturb_envelopes: float = 0.0) -> Dict[str, Dict[str, np.ndarray]]:
    """Synthetic parameter tree (float64 NumPy leaves).  `perturb_envelopes` > 0 adds
    N(0, s) noise to pi/zeta so that parity tests do not run on the degenerate all-ones
    envelope (all K determinants would share one envelope matrix)."""
    rng = np.random.default_rng(seed)
    tree: Dict[str, Dict[str, np.ndarray]] = OrderedDict()
    for mod, leaf, shape, init in param_entries(spec, n_up, n_down, n_nuc):
        val = _draw(rng, shape, init, spec)
        if perturb_envelopes and ((init == 'ones' and 'envelopes' in mod) or init == 'twos'):
            val = val + perturb_envelopes * rng.standard_normal(shape)
        tree.setdefault(mod, OrderedDict())[leaf] = np.asarray(val, np.float64)
    return tree


def n_params(tree) -> int:
    return int(sum(v.size for m in tree.values() for v in m.values()))
