"""Nuclear stream of the TransPsiformer, folded at program-compile time (host, float64 NumPy).

With `elec_to_nuc: false` (conf/ansatz/transpsiformer.yaml:102, gnn/update_features.py:428-434) the nuclear
tokens attend only to nuclear tokens: their embeddings at every layer, and the envelope exponents the
NuclearGNNHead reads out of them (wf/omni.py:181-211), depend on the geometry R and on the parameters but
not on the electrons.  They are therefore constants of a compiled layer program -- like packed weights --
and are evaluated once per parameter set here (n_nuc^2 D^2-class work, ~1 MFLOP) instead of once per walker
and derivative lane on the GPU.  What the device needs per layer are the key / value projections of the
nuclear tokens (extra rows of the attention of the electron queries) and, at the end, zetas.

Follows gnn/electron_gnn.py:435-537 (NucleiEmbedding with nn edge features), update_features.py:385-451
(CombinedNodeAttentionUpdateFeature restricted to the nuclear rows), hkext.py:83-113,165-202 (MLP, GLU).
"""
from __future__ import annotations

import numpy as np

from .params import NUC_EDGE_MLP, NUC_EMB, NUC_HEAD, attention_feature_name, layer_dims, layer_name, nuc_embed_mlp
from .spec import AnsatzSpec, MLPSpec


def _act(name):
    if name is None:
        return lambda x: x
    if name == 'tanh':
        return np.tanh
    if name == 'silu':
        return lambda x: x / (1.0 + np.exp(-x))
    raise ValueError(name)


def _mlp(params, prefix: str, spec: MLPSpec, x, out_dim: int):
    dims = spec.dims(x.shape[-1], out_dim)
    for i in range(len(dims)):
        p = params[f'{prefix}/linear_{i}']
        x = x @ np.asarray(p['w'], np.float64)
        if 'b' in p:
            x = x + np.asarray(p['b'], np.float64)
        x = _act(spec.layer_act(i, len(dims)))(x)
    return x


def nuclei_embedding(params, spec: AnsatzSpec, R: np.ndarray, eps: float) -> np.ndarray:
    n_nuc = R.shape[0]
    d = R[None, :, :] - R[:, None, :]                           # diffs[s, r] = R[r] - R[s] (gnn/graph.py:23-31)
    rho = np.sqrt(eps + (d ** 2).sum(-1))
    s = np.log1p(rho) / rho
    f = np.concatenate([(rho * s)[..., None], d * s[..., None]], -1)
    onehot = np.zeros((n_nuc, n_nuc))
    onehot[np.arange(n_nuc), list(spec.nuc_types)] = 1.0
    f = np.concatenate([f, np.broadcast_to(onehot[:, None, :], (n_nuc, n_nuc, n_nuc))], -1)
    e = _mlp(params, f'{NUC_EMB}/edge_mlp', NUC_EDGE_MLP, f, 32)
    return _mlp(params, f'{NUC_EMB}/embed_mlp', nuc_embed_mlp(spec.embedding_dim), e.sum(0), spec.embedding_dim)


def fold(params, spec: AnsatzSpec, R, eps: float):
    """-> (per layer [(K_nuc[n_nuc, D], V_nuc[n_nuc, D])], zetas {'up','down'} [n_nuc, K, n_env] or None)."""
    R = np.asarray(R, np.float64)
    n_nuc, H = R.shape[0], spec.num_heads
    assert len(spec.nuc_types) == n_nuc, 'spec.nuc_types must list one atom type per nucleus (transpsiformer(charges))'
    h = nuclei_embedding(params, spec, R, eps)
    _, rows = layer_dims(spec, n_nuc)
    kv = []
    for l in range(len(rows)):
        uf = f'{layer_name(l)}/~/{attention_feature_name(spec)}'
        W = {nm: np.asarray(params[f'{uf}/multi_head_attention/{nm}']['w'], np.float64)
             for nm in ('query', 'key', 'value', 'linear')}
        hd = h.shape[-1] // H
        q, k, v = ((h @ W[nm]) for nm in ('query', 'key', 'value'))
        kv.append((k.copy(), v.copy()))
        qh, kh, vh = (a.reshape(n_nuc, H, hd) for a in (q, k, v))
        logits = np.einsum('thd,Thd->htT', qh, kh) / np.sqrt(hd)
        w = np.exp(logits - logits.max(-1, keepdims=True))
        w = w / w.sum(-1, keepdims=True)
        att = h + np.einsum('htT,Thd->thd', w, vh).reshape(n_nuc, H * hd) @ W['linear']
        h = att + _mlp(params, f'{uf}/mlp', spec.attn_mlp, att, h.shape[-1])
    zetas = None
    if spec.envelope == 'simplified':
        K, ne = spec.n_determinants, spec.n_envelope_per_nucleus
        mu = h.mean(-1, keepdims=True)
        ln = (h - mu) / np.sqrt(((h - mu) ** 2).mean(-1, keepdims=True) + 1e-5)     # hk.LayerNorm(-1, False, False)
        zetas = {}
        for spin, glu in (('up', 'zetas_readout_glu'), ('down', 'zetas_readout_glu_1')):
            Wp, Vp = params[f'{NUC_HEAD}/{glu}/W'], params[f'{NUC_HEAD}/{glu}/V']
            gate = 1.0 / (1.0 + np.exp(-(ln @ np.asarray(Wp['w'], np.float64) + np.asarray(Wp['b'], np.float64))))
            lin = ln @ np.asarray(Vp['w'], np.float64) + np.asarray(Vp['b'], np.float64)
            zetas[spin] = (gate * lin).reshape(n_nuc, K, ne) + np.asarray(params[NUC_HEAD][f'zetas_bias_{spin}'], np.float64)
    return kv, zetas
