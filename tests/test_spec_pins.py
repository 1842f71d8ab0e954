"""Independent pins for the parts of the oracle that no reference golden reaches.

1. `deepqmc_amd/spec.py` is a hand transcription of the reference's ansatz YAMLs that the HIP path AND the oracle
   consume, so "HIP == oracle" cannot catch a mistyped width or flag.  Here every field of the four BASELINE
   ansatzes is compared with a machine extraction of the YAMLs (tests/golden/ansatz_yaml.json, made by
   tests/golden/make_yaml_golden.py from /root/reference/src/deepqmc/conf/ansatz/*.yaml), the `'log'` width rule is
   checked against `hkext.py:85-91` read literally, and the parameter count of LiH / psiformer against the number the
   reference itself logs (1 610 498, doc/examples/ground_state_lih.ipynb cell 3).
2. The attention stack of the oracle (`hk.MultiHeadAttention`, `hk.LayerNorm(-1, False, False)`, `hkext.GLU`:
   restated from the dm-haiku documentation and hkext.py:165-202) is compared with PyTorch's independent
   implementations of the same published algorithms: `torch.nn.MultiheadAttention` / `F.scaled_dot_product_attention`
   with the weights transplanted, `F.layer_norm`, `F.glu`; and with formulas written out literally here.
"""
import json
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from deepqmc_amd import spec as S
from deepqmc_amd.hamil import MolecularHamiltonian
from deepqmc_amd.molecule import Molecule
from deepqmc_amd.params import init_params, n_params
from oracle import wf as owf

HERE = os.path.dirname(os.path.abspath(__file__))
YAML = json.load(open(os.path.join(HERE, 'golden', 'ansatz_yaml.json')))
GNN = 'omni_factory.gnn_factory'
LAYER = GNN + '.layer_factory'


def yaml_mlp(flat, prefix):
    """hkext.MLP arguments at `prefix` of a flattened YAML -> MLPSpec (None when the key holds no MLP)."""
    if flat.get(prefix + '._target_') != 'deepqmc.hkext.MLP':
        return None
    n = flat[prefix + '.hidden_layers.#']
    hidden = tuple(flat[f'{prefix}.hidden_layers.{i}'] for i in range(n))
    act = flat.get(prefix + '.activation._target_', flat.get(prefix + '.activation'))
    act = {None: None, 'jax.numpy.tanh': 'tanh', 'jax.nn.silu': 'silu', 'jax.nn.swish': 'silu'}[act]
    return S.MLPSpec(hidden, flat[prefix + '.bias'], flat[prefix + '.last_linear'], act, flat[prefix + '.init'])


def residual_flag(flat, key):
    """ResidualConnection(normalize=...) -> its `normalize`; false / null -> None (no residual)."""
    if flat.get(key + '._target_') == 'deepqmc.hkext.ResidualConnection':
        return flat[key + '.normalize']
    assert flat.get(key) in (False, None)
    return None


@pytest.mark.parametrize('yaml_name,spec_fn', [('default', S.paulinet), ('ferminet', S.ferminet), ('psiformer', S.psiformer),
                                               ('transpsiformer', lambda: S.transpsiformer([6, 6, 1, 1]))])
def test_spec_matches_reference_yaml(yaml_name, spec_fn):
    y, sp = YAML[yaml_name], spec_fn()
    # ---- NeuralNetworkWaveFunction level
    assert y['_target_'] == 'deepqmc.wf.NeuralNetworkWaveFunction'
    assert sp.n_determinants == y['n_determinants'] and sp.full_determinant == y['full_determinant']
    assert y['backflow_transform'] == 'mult' and y['backflow_op.mult_act'] == '${eval:"lambda x: x"}' and y['cusp_nuclei'] is False
    assert sp.conf_coeff == {'haiku.Linear': 'linear', 'deepqmc.hkext.SumPool': 'sum'}[y['conf_coeff._target_']]
    if sp.conf_coeff == 'linear':
        assert y['conf_coeff.with_bias'] is False and y['conf_coeff.w_init._target_'] == 'jax.numpy.ones'
    if y.get('cusp_electrons') is False:
        assert sp.cusp is None
    else:
        kind = {'deepqmc.wf.cusp.DeepQMCCusp': 'deepqmc', 'deepqmc.wf.cusp.PsiformerCusp': 'psiformer'}[y['cusp_electrons.cusp_function._target_']]
        assert (sp.cusp, sp.cusp_same_scale, sp.cusp_anti_scale, sp.cusp_alpha, sp.cusp_trainable_alpha) == (
            kind, y['cusp_electrons.same_scale'], y['cusp_electrons.anti_scale'], y['cusp_electrons.alpha'], y['cusp_electrons.trainable_alpha'])
    # ---- envelopes
    if y['envelope._target_'] == 'deepqmc.wf.env.ExponentialEnvelopes':
        assert sp.envelope == 'exponential' and sp.n_envelope_per_nucleus == 1
        # the variant the kernels implement: isotropic, one shell per nucleus, per-orbital exponent, spin-unrestricted, ones init
        assert (y['envelope.isotropic'], y['envelope.per_shell'], y['envelope.per_orbital_exponent'], y['envelope.spin_restricted'],
                y['envelope.init_to_ones'], y['envelope.softplus_zeta']) == (True, False, True, False, True, False)
    else:
        assert y['envelope._target_'] == 'deepqmc.wf.env.SimplifiedNucleusDependentEnvelopes' and sp.envelope == 'simplified'
        assert sp.n_envelope_per_nucleus == y['envelope.n_envelope_per_nucleus']
        assert y['envelope.per_orbital_exponent'] is False and y['envelope.fixed_pi'] is True
    # ---- OmniNet
    assert sp.embedding_dim == y['omni_factory.embedding_dim']
    assert sp.jastrow == yaml_mlp(y, 'omni_factory.jastrow_factory.subnet_factory')
    if sp.jastrow is not None:
        assert y['omni_factory.jastrow_factory.sum_first'] is True
    assert sp.backflow == yaml_mlp(y, 'omni_factory.backflow_factory.subnet_factory')
    assert sp.nuclei_tokens == (y.get('omni_factory.nuclear_gnn_head._target_') == 'deepqmc.wf.omni.NuclearGNNHead')
    # ---- ElectronGNN
    assert sp.n_interactions == y[GNN + '.n_interactions'] and sp.two_particle_dim == y[GNN + '.two_particle_stream_dim']
    assert sp.self_interaction == y[GNN + '.self_interaction']
    emb = GNN + '.electron_embedding'
    assert y[emb + '.positional_embeddings.ne.features.#'] == 2
    assert y[emb + '.positional_embeddings.ne.features.0._target_'].endswith('DistancePowerEdgeFeature')
    assert y[emb + '.positional_embeddings.ne.features.0.powers.#'] == 1 and y[emb + '.positional_embeddings.ne.features.0.powers.0'] == 1
    assert y[emb + '.positional_embeddings.ne.features.1._target_'].endswith('DifferenceEdgeFeature')
    assert sp.emb_log_rescale == bool(y.get(emb + '.positional_embeddings.ne.features.0.log_rescale', False)) \
        == bool(y.get(emb + '.positional_embeddings.ne.features.1.log_rescale', False))
    assert sp.emb_use_spin == y[emb + '.use_spin'] and sp.emb_project == y[emb + '.project_to_embedding_dim']
    if y.get(GNN + '.edge_features') is None and not any(k.startswith(GNN + '.edge_features.') for k in y):
        assert sp.edge_types == ()
    else:
        types = sorted({k[len(GNN + '.edge_features.'):].split('.')[0] for k in y if k.startswith(GNN + '.edge_features.')})
        assert sorted(sp.edge_types) == types
        for t in types:
            ef = f'{GNN}.edge_features.{t}'
            assert y[ef + '.features.#'] == 2 and y[ef + '.features.0.powers.0'] == 1
            assert sp.edge_log_rescale == bool(y.get(ef + '.features.0.log_rescale', False))
    # ---- layer
    assert y[LAYER + '.update_rule'] == 'concatenate'
    assert sp.electron_residual_normalize == residual_flag(y, LAYER + '.electron_residual')
    assert sp.two_particle_residual_normalize == residual_flag(y, LAYER + '.two_particle_residual')
    feats = []
    for i in range(y[LAYER + '.update_features.#']):
        uf = f'{LAYER}.update_features.{i}'
        kind = y[uf + '._target_'].rsplit('.', 1)[1]
        if kind == 'ResidualElectronUpdateFeature':
            feats.append('residual')
        elif kind == 'NodeSumElectronUpdateFeature':
            assert y[uf + '.normalize'] is True
            feats += [f'node_{y[f"{uf}.node_types.{j}"]}' for j in range(y[uf + '.node_types.#'])]
        elif kind == 'ConvolutionElectronUpdateFeature':
            assert y[uf + '.normalize'] is False
            feats += [f'conv_{y[f"{uf}.edge_types.{j}"]}' for j in range(y[uf + '.edge_types.#'])]
            assert sp.w == yaml_mlp(y, uf + '.w_factory') and sp.h == yaml_mlp(y, uf + '.h_factory')
        elif kind == 'EdgeSumElectronUpdateFeature':
            assert y[uf + '.normalize'] is True
            feats += [f'edge_{y[f"{uf}.edge_types.{j}"]}' for j in range(y[uf + '.edge_types.#'])]
        elif kind in ('NodeAttentionElectronUpdateFeature', 'CombinedNodeAttentionUpdateFeature'):
            feats.append('attention')
            assert sp.layer_kind == 'attention' and sp.num_heads == y[uf + '.num_heads']
            assert sp.attn_mlp == yaml_mlp(y, uf + '.mlp_factory')
            assert y[uf + '.attention_residual.normalize'] is False and y[uf + '.mlp_residual.normalize'] is False
            assert sp.nuclei_tokens == (kind == 'CombinedNodeAttentionUpdateFeature')
            if sp.nuclei_tokens:
                assert y[uf + '.elec_to_nuc'] is False
        else:
            raise AssertionError(kind)
    assert tuple(feats) == sp.update_features
    if sp.layer_kind == 'message_passing':
        sub = yaml_mlp(y, LAYER + '.subnet_factory')            # electron_gnn.py: the default for every label ...
        g = yaml_mlp(y, LAYER + '.subnet_factory_by_lbl.g') or sub        # ... unless overridden per label
        assert sp.g == g and sp.u == sub and y[LAYER + '.deep_features'] == 'shared'
        if sp.w is None:
            assert not any('.w_factory' in k for k in y)
    else:
        assert y[LAYER + '.subnet_factory._target_'] == 'deepqmc.hkext.Identity' and y[LAYER + '.deep_features'] is False
        assert sp.g is None and sp.u is None and sp.w is None and sp.h is None
    if sp.nuclei_tokens:
        ne = GNN + '.nuclei_embedding'
        assert (y[ne + '.embedding_dim'], y[ne + '.atom_type_embedding'], y[ne + '.subnet_type']) == (sp.embedding_dim, True, 'mlp')
        assert y[ne + '.edge_features.features.0.log_rescale'] is True and y[ne + '.edge_features.features.1.log_rescale'] is True
        assert y['omni_factory.nuclear_gnn_head.one_particle_parameters.zetas.0'] == '${ansatz.n_determinants}'
        assert y['omni_factory.nuclear_gnn_head.one_particle_parameters.zetas.1'] == '${ansatz.envelope.n_envelope_per_nucleus}'
    else:
        assert y[GNN + '.nuclei_embedding'] is None


def test_log_width_rule_literal():
    """hkext.py:85-91: qs = [k / n_hidden ...]; dims = [round(in ** (1 - q) * out ** q) for q in qs] -- the examples
    SURVEY section 8 lists (4 -> [11, 32], 8 -> [16, 32], 128 -> [64, 32]) and a sweep."""
    m2, m1 = S.MLPSpec(('log', 2)), S.MLPSpec(('log', 1))
    assert m2.dims(4, 32) == [11, 32] and m2.dims(8, 32) == [16, 32] and m2.dims(128, 32) == [64, 32]
    assert m1.dims(448, 128) == [128] and S.MLPSpec((64, 48)).dims(7, 5) == [64, 48, 5]
    for n_hidden in (1, 2, 3):
        for i, o in ((3, 256), (256, 256), (832, 256), (17, 1), (5, 40)):
            qs = [k / n_hidden for k in range(1, n_hidden + 1)]
            assert S.MLPSpec(('log', n_hidden)).dims(i, o) == [round(i ** (1 - q) * o ** q) for q in qs]


def test_parameter_counts_against_the_reference_log():
    """The reference prints the parameter count at start-up: LiH / psiformer = 1 610 498 (ground_state_lih.ipynb)."""
    h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
    assert n_params(init_params(S.psiformer(), h.n_up, h.n_down, h.n_nuc)) == 1610498
    # closed form read off the reference modules (4 layers: Q, K, V, O 256 x 256 without bias + MLP 256 -> 256 -> 256 with
    # bias, update_features.py:262-286; embedding 4 n_nuc + 1 -> 256 WITHOUT bias, electron_gnn.py:618-619; one backflow
    # 256 -> K N per spin without bias, wf/omni.py:146-153; pi and zeta [K N, n_nuc] per spin, wf/env.py; 2 cusp alphas)
    N, K, nn, D = 4, 16, 2, 256
    closed = 4 * (4 * D * D + 2 * (D * D + D)) + (4 * nn + 1) * D + 2 * D * K * N + 2 * (2 * K * N * nn) + 2
    assert closed == 1610498
    # SURVEY section 8 "Config -> concrete sizes"
    for name, spec, n, lo, hi in (('LiH', S.paulinet(), None, 190e3, 205e3), ('N2', S.ferminet(), None, 750e3, 790e3),
                                  ('benzene', S.psiformer(), None, 1.9e6, 2.0e6)):
        hh = MolecularHamiltonian(mol=Molecule.from_name(name))
        assert lo < n_params(init_params(spec, hh.n_up, hh.n_down, hh.n_nuc)) < hi, name


# ------------------------------------------------------------------------------------------------------------------
def _mha_params(D, seed):
    g = torch.Generator().manual_seed(seed)
    return {f'p/{nm}': {'w': torch.randn(D, D, generator=g, dtype=torch.float64) / math.sqrt(D)} for nm in ('query', 'key', 'value', 'linear')}


@pytest.mark.parametrize('T,D,H', [(4, 32, 4), (9, 64, 4), (7, 24, 3)])
def test_attention_matches_torch_multihead_attention(T, D, H):
    """oracle.wf.attention (hk.MultiHeadAttention, with_bias=False, key_size = D / H) == torch.nn.MultiheadAttention
    with the same projections: haiku's Linear is x @ W with W[in, out], torch's in_proj is x @ W^T; both split the
    projected vector into H contiguous chunks of D / H; logits are scaled by 1 / sqrt(key_size)."""
    p = _mha_params(D, 0)
    h = torch.randn(T, D, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    got = owf.attention(p, 'p', h, H)
    mha = torch.nn.MultiheadAttention(D, H, bias=False, dtype=torch.float64)
    with torch.no_grad():
        mha.in_proj_weight.copy_(torch.cat([p['p/query']['w'].T, p['p/key']['w'].T, p['p/value']['w'].T]))
        mha.out_proj.weight.copy_(p['p/linear']['w'].T)
    ref, _ = mha(h[:, None], h[:, None], h[:, None], need_weights=False)            # (tokens, batch = 1, features)
    np.testing.assert_allclose(got.numpy(), ref[:, 0].detach().numpy(), rtol=1e-12, atol=1e-12)


def test_masked_attention_matches_sdpa_and_literal_softmax():
    """The TransPsiformer mask (update_features.py:428-434: nuclei do not see electrons) through
    F.scaled_dot_product_attention (boolean mask: True = attend) and through a softmax written out by hand with the
    -1e30 fill haiku uses."""
    T, n_nuc, D, H = 11, 4, 32, 4
    hd = D // H
    p = _mha_params(D, 3)
    h = torch.randn(T, D, dtype=torch.float64, generator=torch.Generator().manual_seed(4))
    mask = torch.ones(T, T, dtype=torch.bool)
    mask[:n_nuc, n_nuc:] = False
    got = owf.attention(p, 'p', h, H, mask)
    q, k, v = ((h @ p[f'p/{n}']['w']).reshape(T, H, hd).transpose(0, 1) for n in ('query', 'key', 'value'))       # [H, T, hd]
    sd = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)                    # scale defaults to 1 / sqrt(hd)
    ref = sd.transpose(0, 1).reshape(T, D) @ p['p/linear']['w']
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-12, atol=1e-12)
    out = torch.zeros(T, H, hd, dtype=torch.float64)
    for hh in range(H):
        for t in range(T):
            logits = torch.stack([(q[hh, t] * k[hh, s]).sum() / math.sqrt(hd) if mask[t, s] else torch.tensor(-1e30, dtype=torch.float64)
                                  for s in range(T)])
            w = torch.exp(logits - logits.max())
            out[t, hh] = (w / w.sum()) @ v[hh]
    np.testing.assert_allclose(got.numpy(), (out.reshape(T, D) @ p['p/linear']['w']).numpy(), rtol=1e-12, atol=1e-12)
    # nuclear rows must not depend on the electrons at all
    h2 = h.clone()
    h2[n_nuc:] += 1.0
    np.testing.assert_array_equal(owf.attention(p, 'p', h2, H, mask)[:n_nuc].numpy(), got[:n_nuc].numpy())


def test_nuclear_head_is_layernorm_glu_bias_read_literally():
    """wf/omni.py:181-211 + hkext.py:165-202: zetas = sigmoid(W LN(h) + b_W) * (V LN(h) + b_V) + bias with
    LN = hk.LayerNorm(-1, False, False): (x - mean) / sqrt(biased variance + 1e-5), no scale / offset.  Checked
    against the formula written out and against torch's F.glu (first half * sigmoid(second half))."""
    from deepqmc_amd.params import NUC_HEAD
    spec = S.transpsiformer([6, 6, 1, 1])
    K, ne, D, n_nuc = spec.n_determinants, spec.n_envelope_per_nucleus, 64, 4
    g = torch.Generator().manual_seed(7)
    R = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    params = {NUC_HEAD: {'zetas_bias_up': R(n_nuc, K, ne), 'zetas_bias_down': R(n_nuc, K, ne)}}
    for glu in ('zetas_readout_glu', 'zetas_readout_glu_1'):
        for lin in ('W', 'V'):
            params[f'{NUC_HEAD}/{glu}/{lin}'] = {'w': R(D, K * ne), 'b': R(K * ne)}
    h = 3.0 * R(n_nuc, D) + 0.7
    got = owf.nuclear_head(params, spec, h)
    mean = h.mean(-1, keepdim=True)
    var = ((h - mean) ** 2).mean(-1, keepdim=True)                  # jnp.var: biased
    ln = (h - mean) / torch.sqrt(var + 1e-5)
    np.testing.assert_allclose(ln.numpy(), F.layer_norm(h, (D,), eps=1e-5).numpy(), rtol=1e-12, atol=1e-12)
    for spin, glu in (('up', 'zetas_readout_glu'), ('down', 'zetas_readout_glu_1')):
        W, V = params[f'{NUC_HEAD}/{glu}/W'], params[f'{NUC_HEAD}/{glu}/V']
        a, b = ln @ V['w'] + V['b'], ln @ W['w'] + W['b']
        lit = (1.0 / (1.0 + torch.exp(-b))) * a
        np.testing.assert_allclose(lit.numpy(), F.glu(torch.cat([a, b], -1), -1).numpy(), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(got[spin].numpy(), (lit.reshape(n_nuc, K, ne) + params[NUC_HEAD][f'zetas_bias_{spin}']).numpy(),
                                   rtol=1e-12, atol=1e-12)


def test_host_fold_of_the_nuclear_stream_uses_the_same_head():
    """deepqmc_amd/nuclear_stream.py (the product's parameter-only constant fold) against the oracle's nuclear head and
    nuclei embedding on C4H4: the zeta table the kernels receive is the pinned formula's output."""
    from deepqmc_amd import nuclear_stream
    from oracle import geom
    mol = Molecule.from_name('cyclobutadiene_square')
    h = MolecularHamiltonian(mol=mol)
    spec = S.transpsiformer(mol.charges)
    tree = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=5, perturb_envelopes=0.1)
    kv, zetas = nuclear_stream.fold(tree, spec, mol.coords, geom.F32_EPS)
    p = owf.to_torch(tree)
    r = torch.as_tensor(mol.coords[np.repeat(np.arange(h.n_nuc), mol.charges.astype(int))[:h.n_elec]] + 0.3)
    _, xn = owf.gnn(p, spec, r, torch.as_tensor(mol.coords), h.n_up, geom.F32_EPS)
    z = owf.nuclear_head(p, spec, xn)
    for spin in ('up', 'down'):
        np.testing.assert_allclose(np.asarray(zetas[spin]), z[spin].numpy(), rtol=1e-10, atol=1e-10)
